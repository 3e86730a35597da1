"""CPU model of bilateral_upsample_r8 (miniengineao_amd/csrc/meao_kernels.hip): the UNORM8 code taken from the
uncorrected-reciprocal estimate must equal the code of the correctly rounded chain whenever the estimate is further
than the margin from a rounding boundary -- here with ADVERSARIAL reciprocals (every v_rcp_f32 replaced by the correctly
rounded reciprocal moved by -1, 0 or +1 ulp, the worst the device check meao_selftest(4) allows), so the test does not
depend on how the hardware's reciprocal happens to err.  Also reports how much of the proven bound random operands use."""
import numpy as np
import pytest

F = np.float32
MARGIN = F(2.0 ** -10)


def fma(a, b, c):
    # float64 holds the product of two binary32 exactly; the sum rounds once to 53 bits and once to 24 (a double rounding
    # 2^-29 ulp events apart from a true fma: far below anything this test resolves)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F)


def rcp_exact(x):
    return (1.0 / x.astype(np.float64)).astype(F)


def nudge(x, ulps):
    return (x.view(np.int32) + ulps.astype(np.int32)).view(F)


def unorm8(q):
    s = np.clip(q, F(0), F(1)) * F(255)
    return (s + F(0.5)).astype(np.uint32)


def exact_chain(hd, hi_ao, d, a, tol, noise):
    ks = [F(9), F(3), F(1), F(3)]
    w = [(k.astype(np.float64) / (np.abs(hd - d[i]) + tol).astype(np.float64)).astype(F) for i, k in enumerate(ks)]
    total = (((w[0] + w[1]) + w[2]) + w[3]) + noise
    s = a[0] * w[0]
    for i in (1, 2, 3):
        s = fma(a[i], w[i], s)
    p = hi_ao * (s + noise)
    return (p.astype(np.float64) / total.astype(np.float64)).astype(F)


def estimate_chain(hd, hi_ao, d, a, tol, noise, rng, paired=False):
    """paired: bilateral_upsample_r8<PAIRED> -- the reciprocals of a tap pair from ONE (adversarial) reciprocal of their product,
    1/x0 = x1 * rcp(x0 * x1): three reciprocals per texel instead of five."""
    n = hd.shape[0]
    x = [np.abs(hd - d[i]) + tol for i in range(4)]
    if paired:
        r01 = nudge(rcp_exact(x[0] * x[1]), rng.integers(-1, 2, n))
        r23 = nudge(rcp_exact(x[2] * x[3]), rng.integers(-1, 2, n))
        r = [x[1] * r01, x[0] * r01, x[3] * r23, x[2] * r23]
    else:
        r = [nudge(rcp_exact(x[i]), rng.integers(-1, 2, n)) for i in range(4)]
    # round 6: the constant factors folded into fused multiply-adds (the estimate need not follow the reference's operation order)
    nine, three = np.full(n, 9, F), np.full(n, 3, F)
    total = fma(nine, r[0], fma(three, r[1] + r[3], r[2] + noise))
    s = fma(nine, a[0] * r[0], fma(three, fma(a[3], r[3], a[1] * r[1]), fma(a[2], r[2], noise)))
    q = (hi_ao * s) * nudge(rcp_exact(total), rng.integers(-1, 2, n))
    return fma(np.clip(q, F(0), F(1)), np.full(n, 255, F), np.full(n, F(0.5) + MARGIN, F))


def operands(rng, n):
    # (tolerance and depth differences span the whole range the exact mode accepts: products of two x = |dHi - dLo| + tol run
    # from 2^-88 to 2^42, the extremes the paired form has to survive)
    hd = (2.0 ** rng.integers(-12, 0, n) * (1 + rng.random(n))).astype(F)
    d = [np.maximum(hd * (1 + (rng.random(n) - 0.5) * 2.0 ** rng.integers(-23, 2, n)), 2.0 ** -24).astype(F) for _ in range(4)]
    a = [np.where(rng.integers(0, 4, n) == 0, (rng.integers(0, 256, n) / 255.0), rng.random(n)).astype(F) for _ in range(4)]
    hi_ao = np.where(rng.integers(0, 2, n) == 0, 1.0, rng.integers(0, 256, n) / 255.0).astype(F)
    tol = (2.0 ** rng.integers(-44, 21, n)).astype(F)
    noise = (2.0 ** rng.integers(-30, 51, n)).astype(F)
    return hd, hi_ao, d, a, tol, noise


@pytest.mark.parametrize("paired,bound", [(False, 5.1e-4), (True, 5.7e-4)])
def test_estimate_code_equals_exact_code_outside_the_margin(paired, bound):
    """bound: 31 u * 255 (five reciprocals) / 35 u * 255 (three: 5u instead of 3u per weight reciprocal) + the conversion
    roundings, in codes -- both below the margin of 2^-10 = 9.8e-4."""
    rng = np.random.default_rng(20260925 + paired)
    worst, near, total = 0.0, 0, 0
    for _ in range(8):
        ops = operands(rng, 1 << 18)
        want = unorm8(exact_chain(*ops))
        v = estimate_chain(*ops, rng, paired=paired)
        safe = (v - np.floor(v)) >= F(2) * MARGIN           # the kernel's test: fract(v~ + margin) >= 2 margins
        got = v.astype(np.uint32)
        assert np.array_equal(got[safe], want[safe])
        # distance of the estimate from the exact chain's scaled value, in codes
        q = exact_chain(*ops).astype(np.float64)
        exact_scaled = np.clip(q, 0, 1) * 255.0 + 0.5
        worst = max(worst, float(np.max(np.abs((v.astype(np.float64) - float(MARGIN)) - exact_scaled))))
        near += int(np.count_nonzero(~safe)); total += safe.size
    assert worst < bound < float(MARGIN)                    # the proven bound, and the margin above it
    assert 0.5 * 2.0 ** -9 < near / total < 2.0 * 2.0 ** -9  # the exact path is taken about once in 512 texels
