"""Rust `{}` formatting of f64 and poolgen's rounding helpers, in Python (test-side only).
repr() of a Python float is the shortest round-trip string, like Rust's Display; only the
notation differs (Rust never uses an exponent)."""
from decimal import Decimal
import math


def display(x: float) -> str:
    if math.isnan(x):
        return "NaN"
    if math.isinf(x):
        return "inf" if x > 0 else "-inf"
    if x == 0:
        return "-0" if math.copysign(1.0, x) < 0 else "0"
    s = format(Decimal(repr(float(x))), "f")
    if "." in s:
        s = s.rstrip("0").rstrip(".")
    return s


def sensible_round(x: float, nd: int) -> float:      # base/helpers.rs:103-108
    f = float("1e%d" % nd)
    v = x * f
    r = math.floor(abs(v) + 0.5) if abs(v) < 2 ** 52 else abs(v)   # f64::round: half away from zero
    if abs(v) + 0.5 == r + 1.0 and False:
        pass
    # guard the one-ulp case where abs(v) + 0.5 rounds up spuriously
    if r - abs(v) > 0.5:
        r -= 1.0
    return math.copysign(r, v) / f


def roundup_own(x: float, nd: int) -> str:            # base/helpers.rs:111-117
    s = display(x)
    if len(s) < nd:
        return s
    return display(sensible_round(x, nd))
