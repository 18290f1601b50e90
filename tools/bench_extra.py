#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (not the contract bench): other fields, directions, sizes and curves,
device-resident, HIP-event timed through the library's own profiling hooks."""
import json
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from lambda_elliptic_curves_amd import _lib, fft, msm
    from tools import inputs as util
    from tools.synth import distinct_points
    out = {}

    def time_ntt(tag, fld, name, L, batch=1, inverse=False, reps=10):
        n = (1 << L) * batch
        a = util.rand_elems(name, n, 1)
        t_in = torch.from_numpy(a.view(np.int32 if a.dtype == np.uint32 else np.int64)).cuda()
        t_out = torch.empty_like(t_in)
        for _ in range(2):
            fft.ntt_device(fld, t_in, t_out, L, inverse=inverse, batch=batch)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fft.ntt_device(fld, t_in, t_out, L, inverse=inverse, batch=batch)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        out[tag] = {"ms": dt * 1e3, "elements_per_s": n / dt, "GB_per_s_algorithmic": 2 * n * fld.elem_bytes / dt / 1e9}
        print(tag, out[tag], flush=True)

    fp = util.field_pairs()
    for L in (20, 22, 24, 26):
        time_ntt(f"stark252_fwd_2^{L}", fp["stark252"][0], "stark252", L, reps=5 if L == 26 else 10)
    time_ntt("stark252_inv_2^24", fp["stark252"][0], "stark252", 24, inverse=True)
    time_ntt("fr381_fwd_2^24", fp["fr381"][0], "fr381", 24)
    time_ntt("babybear_u32_4x2^24", fp["babybear_u32"][0], "babybear_u32", 24, batch=4)
    time_ntt("babybear_u64_4x2^24", fp["babybear_u64"][0], "babybear_u64", 24, batch=4)
    time_ntt("babybear_ext4_2^24", fp["babybear_ext4"][0], "babybear_ext4", 24)
    time_ntt("stark252_64cols_2^18", fp["stark252"][0], "stark252", 18, batch=64)

    def time_msm(tag, cname, L, reps=2):
        crv, oid = util.curve_pairs()[cname]
        n = 1 << L
        rng = np.random.default_rng(5)
        sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2)
        tp = distinct_points(crv, n)          # G, 2G, ..., nG: all distinct, Z != 1
        ts = torch.from_numpy(sc.view(np.int64)).cuda()
        msm.msm_device(crv, ts, tp, n)
        t0 = time.perf_counter()
        for _ in range(reps):
            msm.msm_device(crv, ts, tp, n)
        dt = (time.perf_counter() - t0) / reps
        out[tag] = {"ms": dt * 1e3, "points_per_s": n / dt}
        print(tag, out[tag], flush=True)

    for L in (16, 20, 22, 24):
        time_msm(f"bls12_381_g1_2^{L}", "bls12_381_g1", L)
    time_msm("bn254_g1_2^23", "bn254_g1", 23)      # BASELINE config 5: 2^26 points over 8 GPUs = 2^23 per GPU
    time_msm("bn254_g1_2^24", "bn254_g1", 24)
    time_msm("bn254_g2_2^22", "bn254_g2", 22)
    time_msm("bn254_g2_2^23", "bn254_g2", 23)
    time_msm("bls12_381_g2_2^20", "bls12_381_g2", 20)
    # device-resident affine SRS (lw_hip_srs_*): one-off normalisation, then mixed-addition MSMs
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 1 << 24
    tp = distinct_points(crv, n)
    ts = torch.from_numpy((np.random.default_rng(5).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2)).view(np.int64)).cuda()
    t0 = time.perf_counter()
    srs = msm.Srs(crv, t_points=tp, n=n)
    t_create = time.perf_counter() - t0
    srs.msm_device(ts, n)
    t0 = time.perf_counter()
    for _ in range(2):
        srs.msm_device(ts, n)
    dt = (time.perf_counter() - t0) / 2
    out["bls12_381_g1_2^24_affine_srs"] = {"ms": dt * 1e3, "points_per_s": n / dt, "srs_create_ms": t_create * 1e3}
    print("bls12_381_g1_2^24_affine_srs", out["bls12_381_g1_2^24_affine_srs"], flush=True)
    srs.close()
    del tp, ts
    torch.cuda.empty_cache()

    # host-buffer entry points (what a drop-in caller without device residency pays): PCIe copies included
    a = util.rand_elems("stark252", 1 << 24, 2)
    fft.evaluate_fft(fp["stark252"][0], a)
    t0 = time.perf_counter()
    for _ in range(3):
        fft.evaluate_fft(fp["stark252"][0], a)
    dt = (time.perf_counter() - t0) / 3
    out["stark252_fwd_2^24_host_buffers"] = {"ms": dt * 1e3, "elements_per_s": (1 << 24) / dt}
    print("stark252_fwd_2^24_host_buffers", out["stark252_fwd_2^24_host_buffers"], flush=True)
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 1 << 22
    pts = distinct_points(crv, n).cpu().numpy().view(np.uint64)
    sc = np.random.default_rng(6).integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    msm.msm(crv, sc, pts)
    t0 = time.perf_counter()
    for _ in range(2):
        msm.msm(crv, sc, pts)
    dt = (time.perf_counter() - t0) / 2
    out["bls12_381_g1_2^22_host_buffers"] = {"ms": dt * 1e3, "points_per_s": n / dt}
    print("bls12_381_g1_2^22_host_buffers", out["bls12_381_g1_2^22_host_buffers"], flush=True)
    # (the CPU context rows — column-parallel NTT and window-parallel MSM on all host cores — are bench.py's cpu_all_cores)
    json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/bench_extra.json", "w"), indent=1)


if __name__ == "__main__":
    main()
