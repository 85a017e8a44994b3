"""The proof obligations behind the corridor checker's span certificates (csrc/vigo_corridor.hip, SpanConst),
checked on the CPU against the oracle's own samples — a numpy restatement of the kernel's constants, independent of the
device code:

  interval  every sample's float position (oracle sampler, exact-power chain, PS.cpp:1026-1056 + pose2Octomap) lies in
            [(float)(p - R), (float)(p + R)], p = the fast form at ONE clock value of the span, R = 2 E + L * dt;
  keys      the reference's expressions from a pose's float to a lattice point's voxel key (PO.cpp:548-560, octomap
            coordToKey) are monotone in that float: keys at the interval's ends bracket every sample's key, and where the
            ends agree, every sample agrees;
  counts    the lattice count (int)((xmax - xmin) / map_res) of every sample lies in the per-segment range [nlo, nhi],
            and equals nhi exactly when the computed difference reaches the dividing line thr.

No GPU: these are statements about floating-point arithmetic, and they must hold for the kernel to be allowed to skip
samples at all."""
import numpy as np
import pytest

import oracle_lib as ol
from trajectory_planner_amd import synth

U40 = 1.0 + 2.0 ** -40


def bernstein_abs_max_of_derivative(c, Tu):
    deg = len(c) - 1
    best = 0.0
    for i in range(deg):
        bi, ratio, pw = c[1], 1.0, 1.0
        for k in range(1, i + 1):
            ratio *= (i - k + 1) / (deg - k)
            pw *= Tu
            bi += ratio * ((k + 1) * c[k + 1]) * pw
        best = max(best, abs(bi))
    return best


def segment_constants(c, n, dT, box_a, map_res):
    """per axis: E, base, lipd, nlo, nhi, thr — the arithmetic of k_corridor's prologue"""
    deg = len(c) - 1
    Tu = (n - 1) * dT * (1.0 + 2.0 ** -20)
    Tm = abs(Tu)
    A = sum(abs(c[d]) * Tm ** d for d in range(deg + 1))
    A1 = sum((d + 1) * abs(c[d + 1]) * Tm ** d for d in range(deg))
    E = 2.0 ** -46 * A + 2.0 ** -1000
    L = bernstein_abs_max_of_derivative(c, Tu) + 2.0 ** -40 * A1
    drift = n * 2.0 ** -52 * (Tm + abs(dT))
    base = (2.0 * E + L * drift) * U40
    lipd = L * abs(dT) * U40
    h = box_a / 2
    Mx = A * (1.0 + 2.0 ** -20) + abs(h)
    dl = 2.0 ** -50 * (Mx + abs(h))
    ql, qh = (box_a - dl) / map_res, (box_a + dl) / map_res
    ql -= abs(ql) * 2.0 ** -50
    qh += abs(qh) * 2.0 ** -50
    nlo, nhi = int(ql), int(qh)
    thr = -1.0
    if nhi == nlo + 1:
        cnt = lambda d: int(d / map_res)
        cc = nhi * map_res
        for _ in range(8):
            if cnt(np.nextafter(cc, -np.inf)) >= nhi:
                cc = float(np.nextafter(cc, -np.inf))
        while cnt(cc) < nhi:
            cc = float(np.nextafter(cc, np.inf))
        assert cnt(cc) >= nhi > cnt(float(np.nextafter(cc, -np.inf)))
        thr = cc
    return Tu, E, base, lipd, nlo, nhi, thr


def fast_form(c, t):
    x, pw = 0.0, 1.0
    for d in range(len(c)):
        x += c[d] * pw
        pw *= t
    return x


def keys_of(f, h, i, map_res, rf):
    """lattice point i of an axis from the pose's float: (float)(f - h + i * map_res), floor(rf * q) (vectorised)"""
    q = (f.astype(np.float64) - h + i * map_res).astype(np.float32)
    return q, np.floor(rf * q.astype(np.float64)).astype(np.int64)


@pytest.mark.parametrize("seed,n,box,map_res,res", [(1, 6000, (0.4, 0.4, 0.2), 0.2, 0.1), (2, 10000, (0.6, 0.3, 0.2), 0.1, 0.1),
                                                    (3, 3000, (0.55, 0.47, 0.23), 0.2, 0.05), (4, 9000, (0.4, 0.4, 0.2), 0.2, 0.1)])
def test_interval_keys_and_counts_of_every_span(seed, n, box, map_res, res):
    S = 10
    coeffs, n_samp, delT, dur = synth.make_corridor_segments(100 + seed, S, extent_lo=(-4, -4, 0.6), extent_hi=(4, 4, 1.6), n_samples=n)
    if seed == 4:
        coeffs[:, :, 1:] *= 0.05                       # slow segments: long stretches with constant keys
        coeffs[:5, 0, 0] = np.round(coeffs[:5, 0, 0] / res) * res - box[0] / 2 + 1e-7       # a lattice point riding a voxel face
    rf = 1.0 / res
    with ol.pow_mode(True):                            # the exact-power chain: what the device's samples are, by construction
        f_all = ol.poly_sample(coeffs, n_samp, delT, n, f32=True)         # [S, n, 3] float32
    rng = np.random.default_rng(seed)
    spans = 0
    agree = 0
    for s in range(S):
        dT = float(delT[s])
        for a in range(3):
            c = [float(x) for x in coeffs[s, a]]
            Tu, E, base, lipd, nlo, nhi, thr = segment_constants(c, n, dT, box[a], map_res)
            f = f_all[s, :, a]
            h = box[a] / 2
            # counts: every sample's count in [nlo, nhi], and == nhi exactly where the difference reaches thr
            lo64, hi64 = f.astype(np.float64) - h, f.astype(np.float64) + h
            diff = hi64 - lo64
            num = (diff / map_res).astype(np.int64)
            assert num.min() >= nlo and num.max() <= nhi, (s, a, nlo, nhi, int(num.min()), int(num.max()))
            assert nhi - nlo <= 1
            assert np.array_equal(num == nhi, diff >= thr)
            for length in (64, 16, 4):
                for k0 in rng.integers(0, n - length, size=12):
                    k0 = int(k0)
                    cidx = k0 + length // 2
                    hs = max(cidx - k0, k0 + length - 1 - cidx)
                    ts = min(max(cidx * dT, 0.0), Tu)
                    p = fast_form(c, ts)
                    R = (base + lipd * hs) * U40
                    flo, fhi = np.float32(p - R), np.float32(p + R)
                    fk = f[k0:k0 + length]
                    # interval: the oracle's float of every sample of the span
                    assert flo <= fk.min() and fk.max() <= fhi, (s, a, length, k0, float(flo), float(fk.min()), float(fk.max()), float(fhi))
                    # keys: monotone, so bracketed by the ends; equal ends pin every sample
                    for i in range(nhi + 1):
                        _, k_lo = keys_of(np.array([flo]), h, i, map_res, rf)
                        _, k_hi = keys_of(np.array([fhi]), h, i, map_res, rf)
                        _, kk = keys_of(fk, h, i, map_res, rf)
                        assert k_lo[0] <= kk.min() and kk.max() <= k_hi[0]
                        if k_lo[0] == k_hi[0]:
                            assert (kk == k_lo[0]).all()
                            agree += 1
                    spans += 1
    assert spans == S * 3 * 3 * 12
    assert agree > spans            # most lattice points of most spans have one key at both ends: the certificates have something to decide
