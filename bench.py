#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X NTT + MSM hot path (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input resident in HBM:
  * NTT leg (the `value`): one forward Stark252 NTT of 2^log2n elements (Polynomial::evaluate_fft backend arm),
  * MSM leg (reported under "msm"): one BLS12-381 G1 Pippenger MSM of 2^msm_log2n DISTINCT points.
N > 1 (launched by torch.distributed.run, one rank per GPU), weak scaling — per-GPU work is fixed:
  * NTT leg: ONE transform of N_gpus * 2^log2n elements block-distributed over the ranks, through the library's own
    RCCL communicator (lw_hip_comm_init + lw_hip_ntt_sharded_device, csrc/comm.hip: four-step with all-to-all);
    --dist-mode independent instead gives every rank its own 2^log2n transform (STARK columns, no collective);
  * MSM leg: every rank holds 2^msm_log2n (scalar, point) pairs; lw_hip_msm_sharded_device all-gathers the partial sums.
  torch.distributed is used for the rendezvous, the barriers and the max-over-ranks only.

After the timed region rank 0 (N = 1) runs the CPU oracle on the SAME inputs: single-threaded evaluate_fft at 2^log2n
(the `cpu_baseline`), compared byte for byte with the timed GPU output ("bit_exact"), plus the all-cores context rows
(column-parallel NTT, window-parallel MSM over the whole MSM input, also compared).  A mismatch exits non-zero.

Prints ONE JSON line on rank 0.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.3 TB/s achievable)


def rand_field_elems(n, seed, top_bits=59):
    """Synthetic field elements: uniformly random canonical residues < 2^251 < p."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    a[:, 0] &= np.uint64((1 << top_bits) - 1)
    return a


def latest_profile(pattern):
    """Newest committed profiles/rNN_<pattern> (per-round naming)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + pattern)))
    return files[-1] if files else None


def measured_rate(label, default):
    """(rate in products/s, source) of a `RATE <label> <x> Gmul/s` line in the newest committed profiles/rNN_microbench.txt."""
    import re
    path = latest_profile("microbench.txt")
    if path:
        m = re.search(r"RATE %s\s+([0-9.]+) Gmul/s" % re.escape(label), open(path).read())
        if m:
            return float(m.group(1)) * 1e9, "profiles/%s (%s, all CUs)" % (os.path.basename(path), label)
    return default, "default (no profiles/rNN_microbench.txt)"


def pmc_counter(kernel_substr, counter):
    """Launch-weighted mean of a counter for a kernel from the newest committed rocprofv3 counter summary, or None."""
    import csv
    path = latest_profile("pmc_summary.csv")
    if not path:
        return None
    try:
        rows = [r for r in csv.DictReader(open(path)) if r["counter"] == counter and kernel_substr in r["kernel"]]
        tot = sum(float(r["mean_per_launch"]) * int(r["launches"]) for r in rows)
        cnt = sum(int(r["launches"]) for r in rows)
        return tot / cnt if cnt else None
    except Exception:
        return None


def traffic_entry(leg, log2n):
    """(bytes per launch of the leg's dominant kernel, source) from profiles/traffic_latest.json, or (None, None)."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    try:
        tj = json.load(open(tpath))
        ent = tj.get(leg) or (tj if leg == "ntt" and "hbm_bytes_per_launch" in tj else None)
        if ent and ent.get("log2n") == log2n:
            return ent.get("hbm_bytes_per_launch"), ent.get("source", "profiles/traffic_latest.json")
    except Exception:
        pass
    return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2n", type=int, default=24, help="NTT size (Stark252)")
    ap.add_argument("--msm-log2n", type=int, default=24, help="MSM size (BLS12-381 G1)")
    ap.add_argument("--workload", choices=["ntt", "msm", "all", "cfg4", "cfg5"], default="all",
                    help="all = the headline (BASELINE configs 2 + 3); cfg4 / cfg5 = BASELINE's multi-GPU configurations (bench_cfg.py), "
                         "runnable at any --gpus N (N = 1: through the library's 1-rank communicator)")
    ap.add_argument("--cfg5-log2n", type=int, default=26, help="total points of --workload cfg5")
    ap.add_argument("--dist-mode", choices=["sharded", "independent"], default="sharded")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (and with them the bit-exact checks)")
    ap.add_argument("--no-host-path", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run", file=sys.stderr)
        if world == 1 and args.gpus > 1:
            sys.exit(2)
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the HIP path has no CPU fallback", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from lambda_elliptic_curves_amd import _lib, fft
    from lambda_elliptic_curves_amd import distributed as D
    import ctypes as C
    dev = C.c_int(local_rank)
    rc = _lib.lib().lw_hip_init(C.byref(dev), 1)
    if rc:
        raise RuntimeError(_lib.last_error())

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks_ok(ok):
        """Collective decision: True only if every rank says True (a per-rank fallback would leave the others blocked
        in a collective and change the measured workload silently)."""
        if world == 1:
            return ok
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    # library-owned RCCL communicator for the sharded legs (the data path never goes through torch.distributed)
    comm, comm_error = None, None
    if world > 1 and args.dist_mode == "sharded":
        try:
            comm = D.HipComm.from_torch_dist()
        except Exception as e:
            comm_error = "%s: %s" % (type(e).__name__, str(e)[:160])
        if not all_ranks_ok(comm is not None):
            if comm is not None:
                comm.close()
            comm = None
            comm_error = comm_error or "another rank failed to create the communicator"

    result = {}
    mismatch = []
    # ------------------------------------------------------------------ BASELINE configs 4 and 5 (bench_cfg.py)
    if args.workload in ("cfg4", "cfg5"):
        import bench_cfg

        def comm_factory():
            if comm is not None or world > 1:
                return comm, comm_error
            try:   # N = 1: a 1-rank communicator, so that the sharded entry points (and RCCL) are what is timed
                return D.HipComm(D.HipComm.unique_id(), 0, 1), None
            except Exception as e:
                return None, "%s: %s" % (type(e).__name__, str(e)[:160])
        fn = bench_cfg.run_cfg4 if args.workload == "cfg4" else bench_cfg.run_cfg5
        result, ok = fn(args, world, rank, barrier, max_over_ranks, all_ranks_ok, comm_factory)
        comm = None   # closed by the leg
        if rank == 0:
            print(json.dumps(result))
        if world > 1:
            dist.destroy_process_group()
        if not ok:
            print("bench.py: BIT-EXACTNESS FAILURE in %s" % args.workload, file=sys.stderr)
            sys.exit(1)
        return
    # ------------------------------------------------------------------ NTT leg
    if args.workload in ("ntt", "all"):
        L = args.log2n
        n = 1 << L
        fld = fft.Stark252PrimeField
        host = rand_field_elems(n, 0x5EED0000 + L + 1000 * rank)
        t_in = torch.from_numpy(host.view(np.int64)).cuda().view(n, 4)
        t_out = torch.empty_like(t_in)
        dist_mode = "single" if world == 1 else ("sharded" if comm is not None else "independent")
        sharded_error = comm_error
        Lt = L + (world.bit_length() - 1 if dist_mode == "sharded" else 0)

        def step():
            if dist_mode == "sharded":
                return D.ntt_sharded(fld, t_in, Lt, comm)
            fft.ntt_device(fld, t_in, t_out, L)
            return t_out

        ok, err = True, None
        try:
            for _ in range(args.warmup):
                last = step()
            torch.cuda.synchronize()
        except Exception as e:
            ok, err = False, "%s: %s" % (type(e).__name__, str(e)[:160])
        if not all_ranks_ok(ok):
            if dist_mode != "sharded":
                raise RuntimeError("NTT warm-up failed: %s" % err)
            # every rank switches together; the line below is labelled as a different workload
            sharded_error = err or "another rank failed in the sharded warm-up"
            dist_mode, Lt = "independent", L
            for _ in range(args.warmup):
                last = step()
        barrier()
        _lib.profile_begin()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            last = step()
        barrier()
        dt = time.perf_counter() - t0
        prof = _lib.profile_end()
        dt = max_over_ranks(dt)
        ms_per_step = dt * 1e3 / args.steps
        value = world * n * args.steps / dt
        # dominant kernel = the LDS-tiled NTT pass (all passes of one transform are launches of it)
        launches = sum(v[0] for k, v in prof.items() if k.startswith("ntt_pass_kernel"))
        total_ms = sum(v[1] for k, v in prof.items() if k.startswith("ntt_pass_kernel"))
        passes = max(launches // max(args.steps, 1), 1)
        avg_launch_ms = total_ms / max(launches, 1)
        n_local_ntt = n                                          # the local transform is 2^log2n in every mode
        alg_bytes_per_launch = 2.0 * n_local_ntt * 32 / max(passes, 1)     # 2*N*B per transform, spread over its passes
        achieved = alg_bytes_per_launch / (avg_launch_ms * 1e-3) / 1e9
        traffic, traffic_src = traffic_entry("ntt", L)
        mul_peak, mul_peak_src = measured_rate("fe_mul Stark252", 182.5e9)
        result.update({
            "metric": "NTT elems/sec (Stark252 radix-2, 2^%d, forward, bit-exact vs CPU: see 'bit_exact') "
                      "[+ MSM G1 point-adds/sec under 'msm']" % L,
            "value": value, "unit": "elements/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32x8 (256-bit Montgomery, 32-bit limbs)", "data": "synthetic",
            "config": {"workload": ("Stark252 NTT 2^%d forward, inputs resident in HBM" % L) if world == 1 else
                                   ("Stark252 NTT, 2^%d elements per GPU x %d GPUs, %s" % (L, world, dist_mode)),
                       "field": "Stark252", "log2n": L, "log2n_total": Lt, "passes": passes, "parallelism": dist_mode,
                       "transport": ("library-owned RCCL communicator (lw_hip_comm_init), all-to-all = ncclSend/ncclRecv groups"
                                     if dist_mode == "sharded" else None),
                       "sharded_error": sharded_error},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "ntt_pass_kernel", "avg_launch_ms": avg_launch_ms, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch,
                         "note": "256-bit NTT is integer-VALU bound (see 'valu'); HBM fraction reported as required"},
            "valu": {"modmul_per_s": (n // 2) * L * args.steps * world / dt, "unit": "Montgomery products/s",
                     "peak_measured": mul_peak * world, "peak_source": mul_peak_src,
                     "frac": (n // 2) * L * args.steps / dt / mul_peak,
                     "valu_busy_pmc": pmc_counter("ntt_pass_kernel", "VALUBusy"),
                     "note": "the bound that applies: products/s against the measured product rate of the MAC pipe; "
                             "valu_busy_pmc = rocprofv3 VALUBusy (%) from the newest profiles/rNN_pmc_summary.csv"},
            "kernel_times_ms": {k: {"launches": v[0], "avg_ms": v[1] / max(v[0], 1)} for k, v in prof.items()},
        })
        # ---- CPU baseline on the same input + byte comparison with the timed output (rank 0, N = 1 only)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O   # checker / baseline only — never on the product path
            gpu_out = last.cpu().numpy().view(np.uint64).reshape(n, 4)
            t0 = time.perf_counter()
            ref = O.evaluate_fft(O.F_STARK252, host)
            dtc = time.perf_counter() - t0
            ntt_exact = bool(np.array_equal(gpu_out, ref))
            if not ntt_exact:
                mismatch.append("NTT 2^%d output differs from the oracle" % L)
            del gpu_out, ref
            result["cpu_baseline"] = {"value": n / dtc, "unit": "elements/s", "cores": 1, "kind": "port",
                                      "sample": "one Stark252 evaluate_fft of 2^%d elements (the timed GPU input), oracle/lw_oracle.c "
                                                "(C restatement of the reference's single-threaded CPU path, twiddles regenerated "
                                                "per call), %.1f s" % (L, dtc),
                                      "host_cores_available": os.cpu_count()}
            result["bit_exact"] = {"ntt": ntt_exact,
                                   "ntt_check": "timed GPU output of the last step == oracle evaluate_fft on the same 2^%d input, "
                                                "byte for byte" % L}
            # context row C2: all host cores, one column per thread, as provers/stark/src/trace.rs:186-190
            from concurrent.futures import ThreadPoolExecutor
            thr = max(1, min(os.cpu_count() or 1, 64))
            Lc = 20
            cols = [rand_field_elems(1 << Lc, 0xC0150000 + k) for k in range(thr)]
            t0 = time.perf_counter()
            with ThreadPoolExecutor(thr) as ex:
                list(ex.map(lambda a: O.evaluate_fft(O.F_STARK252, a), cols))
            dta = time.perf_counter() - t0
            result["cpu_all_cores"] = {"ntt": {"value": thr * (1 << Lc) / dta, "unit": "elements/s", "cores": thr, "kind": "port",
                                               "sample": "%d Stark252 columns of 2^%d, one oracle evaluate_fft per thread "
                                                         "(column-parallel, trace.rs:186-190), %.1f s" % (thr, Lc, dta)}}
            del cols
        # ---- the drop-in call with host slices (PCIe inclusive; never the `value`)
        if rank == 0 and world == 1 and not args.no_host_path:
            hp = {}
            fft.ntt(fld, host)
            t0 = time.perf_counter()
            for _ in range(2):
                fft.ntt(fld, host)          # a fresh result buffer per call, as Polynomial::evaluate_fft returns a new Vec
            dth = (time.perf_counter() - t0) / 2
            keep = [fft.ntt(fld, host)]
            t0 = time.perf_counter()
            for _ in range(2):
                keep.append(fft.ntt(fld, host))   # fresh result buffers again, but alive past the loop: the call alone, without the free
            dtk = (time.perf_counter() - t0) / 2
            del keep
            out_h = np.empty_like(host)
            args_h = (fld.field, fld.layout, _lib.DIR_FORWARD, host.ctypes.data_as(C.c_void_p), out_h.ctypes.data_as(C.c_void_p), L, 1, 0, None)
            _lib.lib().lw_hip_ntt(*args_h)
            t0 = time.perf_counter()
            for _ in range(3):
                _lib.lib().lw_hip_ntt(*args_h)   # caller-owned buffers reused across calls
            dtr = (time.perf_counter() - t0) / 3
            # the result in a pinned buffer of the library's pool (lw_hip_result_acquire): a new buffer object per call, as a
            # Vec-shaped result is, but recycled resident memory underneath
            with fft.ResultBuffer(fld, n) as rb:
                fft.ntt(fld, host, out=rb.array)
            t0 = time.perf_counter()
            for _ in range(3):
                with fft.ResultBuffer(fld, n) as rb:
                    fft.ntt(fld, host, out=rb.array)
            dtp = (time.perf_counter() - t0) / 3
            hp["ntt"] = {"ms": dth * 1e3, "elements_per_s": n / dth, "ms_fresh_result_call_only": dtk * 1e3, "ms_reused_buffers": dtr * 1e3,
                         "ms_pooled_result": dtp * 1e3,
                         "what": "lw_hip_ntt on host buffers, Stark252 2^%d: H2D + transform + D2H (%d MiB each way); 'ms' = new "
                                 "numpy result buffer per call (huge pages requested, populated by helper threads while the upload "
                                 "and kernels run, downloaded chunk by chunk behind the populate front; includes freeing the "
                                 "previous result: munmap of 512 MiB alone is 19-39 ms on this host, profiles/r03_thp_populate.txt), "
                                 "'ms_fresh_result_call_only' = the same with the results kept alive (no free inside the loop), "
                                 "'ms_reused_buffers' = same caller buffers every call, 'ms_pooled_result' = result "
                                 "in a pinned buffer acquired from / released to the library's pool per call" % (L, n * 32 >> 20)}
            result["host_path"] = hp
        del t_in, t_out, host, last
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ MSM leg
    if args.workload in ("msm", "all"):
        from bench_msm import run_msm_leg
        msm_res = run_msm_leg(args, world, rank, barrier, max_over_ranks, all_ranks_ok, comm, comm_error)
        result["msm"] = msm_res
        be = msm_res.pop("bit_exact", None)
        if be is not None:
            result.setdefault("bit_exact", {}).update(be)
            if not all(v for k, v in be.items() if k in ("msm", "msm_prefix")):
                mismatch.append("MSM result differs from the oracle")
        ca = msm_res.pop("cpu_all_cores", None)
        if ca is not None:
            result.setdefault("cpu_all_cores", {})["msm"] = ca
        hp = msm_res.pop("host_path", None)
        if hp is not None:
            result.setdefault("host_path", {})["msm"] = hp

    if rank == 0:
        print(json.dumps(result))
    if comm is not None:
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    if mismatch:
        print("bench.py: BIT-EXACTNESS FAILURE: " + "; ".join(mismatch), file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
