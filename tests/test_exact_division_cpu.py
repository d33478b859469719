"""The lowering kernels take integer codes from a quotient built with multiply-adds instead of an IEEE divide
(csrc/wc_int_policy.h: IntPolicyT::finish + quant_n_div).  This test restates those two routines in exact rational
arithmetic (every fp32 operation = the exact result rounded to nearest even, fma = one rounding) and checks, without a
GPU, that

  * `finish` returns the correctly rounded reciprocal RN(1/s) whether it starts from RN(1/s) itself or from the float
    next to it on either side (v_rcp_f32 plus one Newton step delivers one of the two floats around 1/s), and
  * `quant_n_div`'s q2 is the correctly rounded quotient RN(w/s), the value the reference's divide produces
    (weight_lowering.py:678-682), for random full-mantissa operands, bf16 weights, exact ties and near-ties.
"""
import random
from fractions import Fraction

import numpy as np

F32_MIN_EXP = -126


def rn32(x: Fraction) -> Fraction:
    """x rounded to the nearest fp32 value (ties to even), as an exact Fraction.  Normal and subnormal range."""
    if x == 0:
        return Fraction(0)
    sign = -1 if x < 0 else 1
    a = -x if x < 0 else x
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1  # 2^e <= a < 2^(e+1)
    e = max(e, F32_MIN_EXP)
    ulp = Fraction(2) ** (e - 23)
    q, r = divmod(a, ulp)
    q = int(q)
    if r * 2 > ulp or (r * 2 == ulp and (q & 1)):
        q += 1
    return sign * q * ulp


def fma32(a, b, c):
    return rn32(a * b + c)


def f32(x) -> Fraction:
    return Fraction(float(np.float32(x)))


def ulp_of(y: Fraction) -> Fraction:
    a = abs(y)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2) ** e > a:
        e -= 1
    return Fraction(2) ** (max(e, F32_MIN_EXP) - 23)


def finish_recip(s: Fraction, r_start: Fraction) -> Fraction:
    """IntPolicyT::finish from `r_start` (within one ulp of 1/s): the exact residual and the half-ulp comparison."""
    r = r_start
    e = fma32(-s, r, Fraction(1))
    if abs(1 / s - r) <= ulp_of(r) * (Fraction(1, 2) + Fraction(1, 1 << 20)):
        # what v_rcp_f32 + one Newton step can deliver (half an ulp + 2^-46): there the residual is exact, so the
        # comparison below is an exact statement about |1/s - r|
        assert e == 1 - s * r
    half_ulp_s = abs(s) * ulp_of(r) / 2  # ldexp(|s|, frexp_exp(r) - 25): exact scaling by a power of two
    if abs(e) > half_ulp_s:  # the device adds +-1 to the bit pattern: the next float in magnitude
        mag, step = abs(r), ulp_of(r)
        if e > 0:
            mag += step
        else:
            is_pow2 = mag.numerator & (mag.numerator - 1) == 0 and mag.denominator & (mag.denominator - 1) == 0
            mag -= step / 2 if is_pow2 else step
        r = mag if r > 0 else -mag
    return r


def quotient(w: Fraction, s: Fraction, r: Fraction) -> Fraction:
    """quant_n_div: q0 = RN(w r); two fma corrections."""
    q = rn32(w * r)
    res = fma32(q, -s, w)
    q = fma32(res, r, q)
    res = fma32(q, -s, w)
    return fma32(res, r, q)


def _rand_f32(rng, lo_exp, hi_exp) -> Fraction:
    m = rng.getrandbits(23) | (1 << 23)
    e = rng.randint(lo_exp, hi_exp)
    v = Fraction(m) * Fraction(2) ** (e - 23)
    return -v if rng.random() < 0.5 else v


def test_reciprocal_is_correctly_rounded_from_any_neighbour():
    rng = random.Random(1)
    for _ in range(3000):
        s = _rand_f32(rng, -60, 60)
        exact = rn32(1 / s)
        u = ulp_of(exact)
        for start in (exact, exact + u, exact - u):
            if ulp_of(start) != u:  # the neighbour sits in another binade: v_rcp + Newton does not land there
                continue
            assert finish_recip(s, start) == exact, (float(s), float(start))


def test_quotient_matches_the_ieee_divide():
    rng = random.Random(2)
    cases = []
    for _ in range(4000):  # full-mantissa operands, quotients up to a few hundred (INT8) and far below one
        s = _rand_f32(rng, -40, 40)
        w = s * Fraction(rng.uniform(-300.0, 300.0))
        cases.append((rn32(w), s))
    for _ in range(2000):  # bf16 weights (8-bit significands): exact ties w/s = k + 1/2 and their neighbours
        s8 = Fraction(rng.randint(128, 255)) * Fraction(2) ** rng.randint(-20, 0)
        k = rng.randint(-8, 7)
        w = s8 * (Fraction(k) + Fraction(1, 2))
        cases.append((rn32(w), rn32(s8)))
        cases.append((rn32(w) + ulp_of(w), rn32(s8)))
        cases.append((rn32(w) - ulp_of(w), rn32(s8)))
    for _ in range(1000):  # tiny weights against large scales (quotients far below 1/2)
        cases.append((_rand_f32(rng, -90, -50), _rand_f32(rng, -30, 30)))
    for w, s in cases:
        r = rn32(1 / s)
        want = rn32(w / s)
        got = quotient(w, s, r)
        assert got == want, (float(w), float(s), float(got), float(want))
        # and the code the reference derives from it: rne(RN(w/s))
        assert round(got) == round(want)


def test_numpy_division_agrees_with_the_rational_model():
    """Sanity of the model itself: rn32(w / s) is what numpy's float32 divide returns."""
    rng = random.Random(3)
    for _ in range(2000):
        w, s = _rand_f32(rng, -20, 20), _rand_f32(rng, -20, 20)
        assert f32(np.float32(float(w)) / np.float32(float(s))) == rn32(w / s)
