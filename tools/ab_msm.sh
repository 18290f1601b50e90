export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# quick MSM timing for the four groups (device-resident inputs); prints one line per (curve, log2n)
python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import bench_msm
from lambda_elliptic_curves_amd import msm, _lib
import ctypes as C
d = C.c_int(0); _lib.lib().lw_hip_init(C.byref(d), 1)
base = bench_msm.synth_points_bls12381_g1(1 << 12, 1)
for name, crv, L in (("bls12381_g1", msm.BLS12381Curve, 24), ("bls12381_g1", msm.BLS12381Curve, 20)):
    n = 1 << L
    rng = np.random.default_rng(1)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    t_pts = torch.from_numpy(base.view(np.int64)).cuda().repeat(n // (1 << 12), 1)
    t_sc = torch.from_numpy(sc.view(np.int64)).cuda()
    msm.msm_device(crv, t_sc, t_pts, n); torch.cuda.synchronize()
    _lib.profile_begin(); t0 = time.perf_counter()
    for _ in range(3): msm.msm_device(crv, t_sc, t_pts, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3; prof = _lib.profile_end()
    print(name, L, round(dt * 1e3, 3), {k: round(v[1] / max(v[0], 1), 3) for k, v in prof.items()})
    t0 = time.perf_counter(); srs = msm.Srs(crv, t_points=t_pts, n=n); torch.cuda.synchronize(); t_create = time.perf_counter() - t0
    srs.msm_device(t_sc, n); torch.cuda.synchronize()
    _lib.profile_begin(); t0 = time.perf_counter()
    for _ in range(3): srs.msm_device(t_sc, n)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3; prof = _lib.profile_end()
    print(name, L, "affine SRS:", round(dt * 1e3, 3), "create", round(t_create * 1e3, 1), {k: round(v[1] / max(v[0], 1), 3) for k, v in prof.items() if "accumulate" in k})
    srs.close()
PY
