"""Not gpu: the arithmetic of the trace engine's conservative fp32 pre-test (csrc/trace_fast.hpp: tri_pre_reject) restated in numpy
float32 and held against the reference's own triangle test (MTPC/sceneManagement.cpp:316-338 + the t_x of pathTracing.cpp:347) in
float64 on every (ray, triangle) pair of a shipped scene: a triangle the reference accepts with a positive t may never be rejected.
The same property is checked on the GPU by the self-check build (tests/test_gpu_parity.py::test_pre_test_rejects_no_candidate); this
one pins the formulas and their constants where no GPU is needed."""
import numpy as np
import pytest

from conftest import SCENES

F = np.float32


def _fma(a, b, c):
    """one rounding: the products of two float32 values are exact in float64, the sum rounds once to float32 (what v_fma_f32 does, up to
    the double rounding of cases that cannot matter for a bound with a third of slack)"""
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64) + np.asarray(c, dtype=np.float64)).astype(F)


def _reference_test(v1, v2, v3, nrm, o, d):
    with np.errstate(all="ignore"):
        t = ((v1 - o) * nrm).sum(1) / (nrm * d).sum(1)
        p = o + d * t[:, None]

        def dirk(a, b):
            return (np.cross(b - a, p - a) * nrm).sum(1)
        d1, d2, d3 = dirk(v1, v2), dirk(v2, v3), dirk(v3, v1)
        accept = (d1 * d2 >= 0) & (d1 * d3 >= 0) & (d2 * d3 >= 0)
        tk = (p[:, 0] - o[0]) / d[0]
    return accept & (tk > 0), tk


def _pre_test(V0, E1, E2, a1, a2, S, o, d, limit, margin):
    """tri_pre_reject, operation for operation; returns the rejection mask"""
    O, D = o.astype(F), d.astype(F)
    dm, omax = np.abs(D).max(), np.abs(O).max()
    in_range = omax <= F(4) * F(S) and 1e-6 <= S <= 1e6 and 1e-6 <= dm <= 1e6
    eta4 = F(2.0 ** -20) * (omax + F(S)) * F(1.0001) if in_range else F(np.inf)
    tv = O - V0
    px = _fma(D[1], E2[:, 2], -(D[2] * E2[:, 1])); py = _fma(D[2], E2[:, 0], -(D[0] * E2[:, 2])); pz = _fma(D[0], E2[:, 1], -(D[1] * E2[:, 0]))
    det = _fma(E1[:, 2], pz, _fma(E1[:, 1], py, E1[:, 0] * px))
    u = _fma(tv[:, 2], pz, _fma(tv[:, 1], py, tv[:, 0] * px))
    qx = _fma(tv[:, 1], E1[:, 2], -(tv[:, 2] * E1[:, 1])); qy = _fma(tv[:, 2], E1[:, 0], -(tv[:, 0] * E1[:, 2])); qz = _fma(tv[:, 0], E1[:, 1], -(tv[:, 1] * E1[:, 0]))
    v = _fma(D[2], qz, _fma(D[1], qy, D[0] * qx))
    tq = _fma(E2[:, 2], qz, _fma(E2[:, 1], qy, E2[:, 0] * qx))
    T = np.abs(tv).max(1)
    base = _fma(T, F(2.0 ** -18), eta4)
    da1, da2 = dm * a1, dm * a2
    Eu, Ev, X, Etq = da2 * base, da1 * base, da2 * a1, (a1 * a2) * base
    Dt = np.abs(det)
    neg = det < 0
    U, V, TQ = np.where(neg, -u, u), np.where(neg, -v, v), np.where(neg, -tq, tq)
    with np.errstate(all="ignore"):
        clear = Dt > F(2.0 ** -9) * X
        rej = (U < -Eu) | (V < -Ev) | ((U + V) - Dt > (Eu + Ev) + F(2.0 ** -19) * X)
        rej |= (TQ + Etq < -((F(margin) * Dt) * F(1.002))) | (TQ - Etq > (F(limit) * Dt) * F(1.002))
    return clear & rej


@pytest.mark.parametrize("name", ["cornell-box", "veach-mis"])
def test_pre_test_never_rejects_what_the_reference_accepts(mcpt, name):
    sc = mcpt.Scene(SCENES, name, width=64, height=36)
    g, _, _ = sc.faces()
    sc.close()
    v1, v2, v3, nrm = g[:, 0:3], g[:, 3:6], g[:, 6:9], g[:, 24:27]
    S = float(np.abs(g[:, :9]).max())
    e1, e2, e3 = v2 - v1, v3 - v1, (v3 - v1) - (v2 - v1)
    V0, E1, E2 = v1.astype(F), e1.astype(F), e2.astype(F)
    a1 = np.nextafter((np.abs(e1).sum(1) * (1 + 2.0 ** -20)).astype(F), F(np.inf))          # rounded up, like k_build_pre
    a2 = np.nextafter((np.abs(e2).sum(1) * (1 + 2.0 ** -20)).astype(F), F(np.inf))
    ln = np.stack([np.linalg.norm(e, axis=1) for e in (e1, e2, e3)])
    off = ~((ln.max(0) < np.inf) & (ln.min(0) >= 2.0 ** -16 * S) & (ln.min(0) >= 2.0 ** -12 * ln.max(0)))
    a1 = np.where(off, F(np.inf), a1)                                                          # pre-test switched off for these
    rng = np.random.default_rng(5)
    pts = g[:, :9].reshape(-1, 3)
    lo, hi = pts.min(0), pts.max(0)
    n_rays = 150
    o = lo + (hi - lo) * rng.random((n_rays, 3))
    d = rng.normal(size=(n_rays, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o[::3] = v1[rng.integers(len(v1), size=len(o[::3]))] + 0.01 * d[::3]       # rays that leave a surface, like shadow and bounce rays
    # ... and rays aimed at points ON edges and AT vertices of triangles: there a weight is zero up to rounding, the reference's
    # products decide by their last bits, and only the error bounds keep the fp32 test from deciding differently (with the bounds
    # set to zero this test fails at once)
    tri = rng.integers(len(v1), size=n_rays)
    s_e = rng.random(n_rays)[:, None]
    on_edge = v1[tri] + s_e * (v2[tri] - v1[tri])
    on_edge[::2] = v3[tri[::2]]                                                 # every other one: a vertex
    o_e = lo + (hi - lo) * rng.random((n_rays, 3))
    d_e = on_edge - o_e
    d_e /= np.linalg.norm(d_e, axis=1, keepdims=True)
    o, d = np.vstack([o, o_e]), np.vstack([d, d_e])
    n_rays = len(o)
    cands = rejected = 0
    for i in range(n_rays):
        cand, tk = _reference_test(v1, v2, v3, nrm, o[i], d[i])
        rmax = 1.0 / np.abs(d[i]).min()
        scale = max(S, np.abs(o[i]).max())
        margin = 1.0000001e-9 * scale * rmax if rmax <= 1e6 else np.inf
        # (a) no leader yet: nothing may be rejected for distance; (b) the closest candidate leads: only what lies beyond it may go
        rej = _pre_test(V0, E1, E2, a1, a2, S, o[i], d[i], np.inf, margin)
        assert not (rej & cand).any(), (name, i, np.nonzero(rej & cand)[0][:5])
        if cand.any():
            best = tk[cand].min()
            limit = np.nextafter(F((best + best * 2.0 ** -47) + margin), F(np.inf))
            rej2 = _pre_test(V0, E1, E2, a1, a2, S, o[i], d[i], limit, margin)
            wrong = rej2 & cand & (tk <= best)
            assert not wrong.any(), (name, i, np.nonzero(wrong)[0][:5])
        cands += int(cand.sum()); rejected += int(rej.sum())
    assert cands > 50 and rejected > 0.9 * n_rays * len(g)       # the test rejects nearly everything, and there were candidates to lose


def _ru32(x):
    """float64 array -> float32 rounded toward +inf (the device's __double2float_ru)"""
    f = x.astype(np.float32)
    low = f.astype(np.float64) < x
    return np.where(low, np.nextafter(f, np.float32(np.inf)), f).astype(np.float32)


def test_pool_engine_recomputed_pads_and_margin_are_never_smaller():
    """csrc/trace_pool.hpp does not carry the culling pads and the pruning margin of a ray in its LDS slot: every step recomputes them in
    fp32 from the floats it has at hand (pad_of, margin_of).  Culling is only safe with pads / margins at least as large as the ones
    the voting engine derives in fp64 (trace_fast.hpp: make_rayf; trace_persistent.hpp: margin_f), so: the fp32 formulas, restated
    here operation for operation in numpy float32, against the fp64 ones on two million random rays over eight decades of scene size."""
    rng = np.random.default_rng(12)
    n = 2_000_000
    absmax = 10.0 ** rng.uniform(-3, 5, n)
    o = (rng.random((n, 3)) * 2 - 1) * absmax[:, None] * rng.choice([1.0, 3.9], n)[:, None]
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[::7, 0] *= 1e-5                                        # near-axis-parallel directions: large reciprocals
    rcp = 1.0 / d
    f32 = np.float32
    # the voting engine (fp64, rounded up to float)
    pad64 = _ru32(16.0 * 2.0 ** -24 * (np.abs(o) + 3.0 * absmax[:, None]) * np.abs(rcp) * 1.0000002)
    rmax = np.abs(rcp).max(axis=1)
    scale = np.maximum(absmax, np.abs(o).max(axis=1))
    margin64 = np.where(rmax <= 1e6, _ru32(1.0000001e-9 * scale * rmax), f32(np.inf))
    # the pool engine (fp32): of = (float)o, rf = (float)rcp, s3f = ru(3 absmax), absmax_f = ru(absmax)
    of, rf = o.astype(f32), rcp.astype(f32)
    s3f, absmax_f = _ru32(3.0 * absmax), _ru32(absmax)
    c_pad = f32(2.0 ** -20) * f32(1.00001)
    pad32 = c_pad * ((np.abs(of) + s3f[:, None]) * np.abs(rf))
    rmax_f = np.abs(rf).max(axis=1)
    scale_f = np.maximum(absmax_f, np.abs(of).max(axis=1))
    c_m = f32(1.0000001e-9) * f32(1.00001)
    margin32 = np.where(rmax_f <= f32(0.99e6), c_m * (scale_f * rmax_f), f32(np.inf))
    assert pad32.dtype == np.float32 and margin32.dtype == np.float32
    assert (pad32 >= pad64).all(), int((pad32 < pad64).sum())
    assert (margin32 >= margin64).all(), int((margin32 < margin64).sum())
    # ... and not wastefully larger: within 0.01 %
    fin = np.isfinite(margin64) & np.isfinite(margin32)
    assert fin.sum() > n // 2
    assert (pad32 <= pad64 * f32(1.0001)).all() and (margin32[fin] <= margin64[fin] * f32(1.0001)).all()


def test_bench_quotes_only_profiles_of_the_loaded_build(tmp_path):
    """bench.committed_profile: counters under profiles/ are quoted only when the file carries the build id asked for."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    (prof / "r08_final_hbm_traffic.json").write_text(json.dumps({"bytes_per_launch": 2.0, "build_id": "aaaaaaaaaaaaaaaa"}))
    (prof / "r09_final_hbm_traffic.json").write_text(json.dumps({"bytes_per_launch": 1.0, "build_id": "0123456789abcdef"}))
    old = bench.ROOT
    bench.ROOT = str(tmp_path)
    try:
        f, j, why = bench.committed_profile("r*_final_hbm_traffic.json", "ffffffffffffffff")
        assert f is None and j is None and "another build" in why
        f, j, why = bench.committed_profile("r*_final_hbm_traffic.json", "aaaaaaaaaaaaaaaa")
        assert j["bytes_per_launch"] == 2.0 and why is None and f.endswith("r08_final_hbm_traffic.json")
        f, j, why = bench.committed_profile("r*_final_nothing.json", "aaaaaaaaaaaaaaaa")
        assert j is None and "no committed profile" in why
    finally:
        bench.ROOT = old
