"""The two inequalities S1's lazy form rests on ("S1 lazy form"), checked on the CPU with a numpy restatement of
the device helpers (fast-plaid_amd/csrc/fp_kernels.hip: mono16 / unmono16 / s1_u2 / s1_lower16, and the per-column slack of
k_probe_tau / k_lz_exact) -- no GPU needed:

  (1) the kernel stores s = h(x + u), u = w + kappa |x|, where the reference's value is t = h(c) for some chain result
      c in [x - u, x + u]:        s1_lower16(s) <= t <= s                       (what the probe's threshold and every bound stage use)
  (2) s - s1_lower16(s) <= ulp16(s) + 2 u2(|s|)  with ulp16 clamped at 2^-14    (the selection's slack per column maximum)

h = round-to-nearest-even to fp16.  The restatement follows the device code line by line in fp32 arithmetic."""
import numpy as np

KAPPA = np.float32(2.0 ** -20)


def mono16(h):
    h = np.where((h & 0x7FFF) == 0, 0, h).astype(np.uint32)
    return np.where(h & 0x8000, (~h) & 0xFFFF, h | 0x8000).astype(np.uint32)


def unmono16(k):
    k = k.astype(np.uint32)
    return np.where(k & 0x8000, k ^ 0x8000, (~k) & 0xFFFF).astype(np.uint16)


def s1_u2(s_abs, w, kappa):
    return (np.float32(2.0) * (w + np.float32(1.01) * kappa * s_abs) + np.float32(1e-30)).astype(np.float32)


def f16_bits_to_f32(b):
    return b.astype(np.uint16).view(np.float16).astype(np.float32)


def s1_lower16(stored, w, kappa):
    k = mono16(stored)
    kp = np.where(k > 0, k - 1, 0).astype(np.uint32)
    kp = np.where(kp == 0x7FFF, 0x7FFE, kp)
    sf, pf = f16_bits_to_f32(stored), f16_bits_to_f32(unmono16(kp))
    mid = (np.float32(0.5) * sf + np.float32(0.5) * pf).astype(np.float32)
    lo = (mid - s1_u2(np.abs(sf), w, kappa)).astype(np.float32)
    lo = (lo - np.abs(lo) * np.float32(2.4e-7) - np.float32(1e-37)).astype(np.float32)
    out = lo.astype(np.float16).view(np.uint16)
    return np.where((w == 0) | (k == 0), stored, out).astype(np.uint16)


def _samples(rng, n):
    # scores of unit-ish vectors (|x| up to ~1.2), with a heavy share near zero where fp16 is finest and the window spans
    # several fp16 steps, plus larger values (unnormalised queries); windows of queries of norm 1e-3 .. 30 (w0 = 2^-21.5)
    mag = np.concatenate([rng.uniform(0, 1.2, n // 3), 10.0 ** rng.uniform(-9, 0, n // 3), rng.uniform(1.0, 40.0, n - 2 * (n // 3))])
    x = (mag * rng.choice([-1.0, 1.0], n)).astype(np.float32)
    w = (2.0 ** -21.5 * 10.0 ** rng.uniform(-3, 1.5, n)).astype(np.float32)
    return x, w


def test_stored_upper_candidate_brackets_the_reference_value():
    rng = np.random.default_rng(7)
    x, w = _samples(rng, 2_000_000)
    u = (KAPPA * np.abs(x) + w).astype(np.float32)          # fma in the kernel: at most one rounding less
    s = (x + u).astype(np.float32).astype(np.float16)        # the stored upper candidate
    s_bits = s.view(np.uint16)
    lo_bits = s1_lower16(s_bits, w, KAPPA)
    lo = f16_bits_to_f32(lo_bits).astype(np.float64)
    for c in ((x.astype(np.float64) - u), x.astype(np.float64), (x.astype(np.float64) + u)):   # the extremes suffice: h is monotone
        t = c.astype(np.float32).astype(np.float16).astype(np.float64)
        assert np.all(t <= s.astype(np.float64)), "the stored value must not lie below a possible reference value"
        bad = np.nonzero(lo > t)[0]
        assert bad.size == 0, (x[bad[:3]], w[bad[:3]], s[bad[:3]], lo[bad[:3]], t[bad[:3]])


def test_lower_bound_is_monotone_in_the_stored_value():
    # (the probe lowers its threshold with it: a larger stored maximum must never give a smaller bound)
    keys = np.arange(1, 0x10000, dtype=np.uint32)
    keys = keys[(keys != 0x7FFF)]
    bits = unmono16(keys)
    vals = f16_bits_to_f32(bits)
    ok = np.isfinite(vals)
    bits, vals = bits[ok], vals[ok]
    for w in (np.float32(1e-9), np.float32(3.4e-7), np.float32(1e-5)):
        lo = f16_bits_to_f32(s1_lower16(bits, np.full(bits.shape, w, np.float32), KAPPA))
        order = np.argsort(vals, kind="stable")
        assert np.all(np.diff(lo[order]) >= 0)
        assert np.all(lo <= vals)


def test_column_slack_covers_the_gap_to_the_lower_bound():
    rng = np.random.default_rng(11)
    n = 1_000_000
    s = np.abs(np.concatenate([rng.uniform(0, 1.2, n // 2), 10.0 ** rng.uniform(-9, 1.5, n - n // 2)])).astype(np.float16)
    w = (2.0 ** -21.5 * 10.0 ** rng.uniform(-3, 1.5, n)).astype(np.float32)
    bits = s.view(np.uint16)
    lo = f16_bits_to_f32(s1_lower16(bits, w, KAPPA)).astype(np.float64)
    e = (bits & 0x7C00).astype(np.uint32)
    e = np.where(e < 0x2C00, 0x2C00, e) - 0x2800
    ulp = f16_bits_to_f32(e.astype(np.uint16))
    slack = (ulp + np.float32(2.0) * s1_u2(np.abs(s.astype(np.float32)), w, KAPPA)).astype(np.float64)
    gap = s.astype(np.float64) - lo
    bad = np.nonzero(gap > slack * (1 + 1e-6))[0]
    assert bad.size == 0, (s[bad[:3]], w[bad[:3]], gap[bad[:3]], slack[bad[:3]])
