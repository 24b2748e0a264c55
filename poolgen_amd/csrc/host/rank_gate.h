// rank_gate.h -- the go / abort agreement of the CLI's rank threads (`poolgen ... --n-gpus N`, host/main.cpp).
#pragma once
#include <condition_variable>
#include <exception>
#include <mutex>
#include <stdexcept>

// A rank thread must never enter a collective (ncclCommInitRank, the all-reduce) that another rank will not reach: every
// rank reports at the gate whether its part up to here succeeded, waits for the others, and all of them go on or none does.
struct RankAborted : std::runtime_error {
    RankAborted() : std::runtime_error("another rank failed before the collective; this rank stops too") {}
};
class RankGate {
    std::mutex mu;
    std::condition_variable cv;
    const int n;
    int arrived = 0, generation = 0;
    bool all_ok = true, verdict = true;
public:
    explicit RankGate(int n_ranks) : n(n_ranks) {}
    bool arrive(bool ok) { // true on every rank iff every rank arrived with ok
        std::unique_lock<std::mutex> lk(mu);
        all_ok = all_ok && ok;
        const int gen = generation;
        if (++arrived == n) {
            verdict = all_ok;
            arrived = 0; all_ok = true; ++generation;
            cv.notify_all();
            return verdict;
        }
        cv.wait(lk, [&] { return generation != gen; });
        return verdict;
    }
    // runs `part` (the rank's work since the last gate), then the gate: rethrows the rank's own error, or RankAborted when
    // it was another rank that failed
    template <typename F> void pass(F part) {
        std::exception_ptr mine;
        try { part(); } catch (...) { mine = std::current_exception(); }
        const bool go = arrive(!mine);
        if (mine) std::rethrow_exception(mine);
        if (!go) throw RankAborted();
    }
};

