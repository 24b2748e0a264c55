import sys, time, torch, os
sys.path.insert(0, ".")
from poolgen_amd import Engine, synth
eng = Engine(0)
for n in [int(x) for x in sys.argv[1:]] or (250, 224, 500):
    p = 4_000_000 if n < 400 else 2_000_000
    G = synth.genotype_matrix(p, n, "cuda")
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        S = eng.kinship_partial(G, n)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(n, p, "kinship ms %.2f" % (dt * 1e3), "TFLOP/s algorithmic %.1f" % (2.0 * n * n * p / dt / 1e12))
    del G
