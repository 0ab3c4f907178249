"""
The arithmetic behind csrc/mtm_device_util.hip.h::quotient_as_float, restated in numpy (CPU suite, always on).

The IEEE-division epilogues of the score kernel store (float)(num / t) - the value OpenCV's common_matchTemplate stores
 - without dividing: q0 = num * RN(RN(1 / sq) * RN(1 / templ_norm)), and only where q0 lies within 32
ulp(double) of a float32 rounding boundary (or, in the general form, is tiny) the division itself.  float64 multiplication and
division are IEEE operations on both sides, so the claim "an unflagged q0 rounds to the same float as the true quotient" can be
checked here; the GPU test (tests/test_gpu_last_segments.py::test_quotient_without_division) checks the compiled function.
"""
import numpy as np


def needs_division(q0, tiny=True):
    bits = q0.view(np.uint64)
    lo = (bits & np.uint64(0x1FFFFFFF)).astype(np.int64)
    near = np.abs(lo - 0x10000000) <= 32
    if not tiny:
        return near
    mag = np.abs(q0)
    return near | ((mag > 0) & (mag < 2.0 ** -120))


def operands(rng, n):
    e_w = (rng.integers(1, 2 ** 44, n) >> rng.integers(0, 32, n)).astype(np.float64) + 1.0
    e_t = (rng.integers(1, 2 ** 44, n) >> rng.integers(0, 32, n)).astype(np.float64) + 1.0
    sq, tn = np.sqrt(e_w), np.sqrt(e_t)
    return sq, tn, sq * tn, (1.0 / sq) * (1.0 / tn)


def test_unflagged_quotients_round_like_the_division():
    rng = np.random.default_rng(1)
    n = 2_000_000
    sq, tn, tt, rr = operands(rng, n)
    # random numerators, integer-valued for half of them
    num = (rng.random(n) * 2.4 - 1.2) * tt
    num[::2] = np.rint(num[::2])
    q0, qr = num * rr, num / tt
    flag = needs_division(q0, tiny=False)
    assert np.array_equal(q0[~flag].astype(np.float32).view(np.uint32), qr[~flag].astype(np.float32).view(np.uint32))
    assert flag.mean() < 1e-5
    # the two float64 quotients stay within the 6 ulp the source states
    ok = (np.abs(qr) > 2.0 ** -1000) & (np.abs(qr) < 2.0 ** 1000)
    d = np.abs(np.abs(q0[ok]).view(np.int64) - np.abs(qr[ok]).view(np.int64))
    assert d.max() <= 6, int(d.max())


def test_quotients_placed_on_rounding_boundaries_are_all_flagged_and_right():
    rng = np.random.default_rng(2)
    n = 1_000_000
    sq, tn, tt, rr = operands(rng, n)
    fb = rng.integers(0x35800000, 0x3F800000, n).astype(np.uint32)           # floats in [2^-20, 1)
    lo_f, hi_f = fb.view(np.float32).astype(np.float64), (fb + np.uint32(1)).view(np.float32).astype(np.float64)
    num = (0.5 * (lo_f + hi_f)) * tt                                           # the quotient straddles the boundary ...
    num = (num.view(np.int64) + rng.integers(-4, 5, n)).view(np.float64)       # ... a few ulp either side of it
    num[::3] = -num[::3]
    q0, qr = num * rr, num / tt
    flag = needs_division(q0, tiny=False)
    assert flag.all()
    # and what the kernel stores for them is the division's float by construction; here: the margin of 32 ulp really is needed
    # for SOME of them (the plain product would have rounded to the other float)
    wrong = q0.astype(np.float32).view(np.uint32) != qr.astype(np.float32).view(np.uint32)
    assert 0.05 < wrong.mean() < 0.95


def test_tiny_quotients_are_flagged_by_the_general_form_only():
    q = np.array([2.0 ** -121, -2.0 ** -130, 2.0 ** -149 * 1.5, 2.0 ** -119, 0.0, -0.0, 0.25], dtype=np.float64)
    # (significands chosen away from the boundary pattern: only the magnitude decides)
    q = (q.view(np.uint64) & ~np.uint64(0x1FFFFFFF)).view(np.float64)
    assert needs_division(q, tiny=True).tolist() == [True, True, True, False, False, False, False]
    assert not needs_division(q, tiny=False).any()
