#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs over bench.py (FETCH_SIZE, WRITE_SIZE; separate passes, MI355X_MICROARCH.md §HBM)
into profiles/traffic_latest.json: HBM bytes per launch of the NTT pass kernel ("ntt") and HBM bytes of all MSM
kernels of one MSM step ("msm").
gfx950 corrections: counters are in KiB; FETCH_SIZE reads exactly 1/2 of a wide coalesced stream -> doubled.  The
guide calibrates that factor for 16-byte-per-lane STREAMING reads and says to calibrate other patterns on a known byte
count: the MSM accumulation gathers one 128-byte-aligned 128-byte row per item (c = 16: 268 M items at 2^24 = 34.4 GB of
lines + 1.07 GB of indices) and its raw FETCH_SIZE read 35.5 GB when this was calibrated (38-39 GB with the ordered piece
dispatch; c = 20: 218 M items, 28.8 GB expected, 32.6-33.8 GB read), i.e. the counter is within 15 % of the line count for
whole-line gathers and nowhere near half of it — so kernels whose name contains "msm_accumulate" take factor 1, everything
else factor 2.
With the two bench lines (plain run, run under rocprofv3 --kernel-trace --stats) it also records how much slower the
kernels ran under the profiler, so that profiles/*kernel_stats.csv can be compared with the unprofiled bench line.
usage: parse_pmc.py FETCH_DIR WRITE_DIR NTT_LOG2N MSM_LOG2N MSM_STEPS_TIMED OUT.json [BENCH.json BENCH_UNDER_ROCPROF.json]"""
import csv
import glob
import json
import sys


def per_kernel(dirpath, counter):
    tot, cnt = {}, {}
    for f in glob.glob(dirpath + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            k = row["Kernel_Name"]
            tot[k] = tot.get(k, 0.0) + float(row["Counter_Value"])
            cnt[k] = cnt.get(k, 0) + 1
    return {k: (tot[k] / cnt[k], cnt[k]) for k in tot}


def main():
    fetch_dir, write_dir, ntt_log2n, msm_log2n, msm_steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), sys.argv[6]
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fe) | set(wr)):
        factor = 1 if "msm_accumulate" in k else 2      # see the module docstring
        f = fe.get(k, (0, 0))[0] * 1024 * factor    # KiB -> bytes, gfx950 correction
        w = wr.get(k, (0, 0))[0] * 1024
        rows[k.split("(")[0][:120]] = {"fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w, "launches": fe.get(k, wr.get(k))[1]}
    ntt = [v for k, v in rows.items() if "ntt_pass_kernel" in k]
    per_launch = sum(v["hbm_bytes"] * v["launches"] for v in ntt) / max(sum(v["launches"] for v in ntt), 1)
    # MSM: all kernels of the lw::msm_* / ec_* family; bench.py ran warm-up (2) + timed (msm_steps) MSMs in the profiled process
    msm = {k: v for k, v in rows.items() if "lw::msm_" in k}
    msm_total = sum(v["hbm_bytes"] * v["launches"] for v in msm.values())
    n_msm = 2 + msm_steps
    ratio = None
    if len(sys.argv) > 8:
        try:
            plain, prof = (json.loads(open(p).read().strip().splitlines()[-1]) for p in (sys.argv[7], sys.argv[8]))
            ratio = {"ntt_ms_per_step": {"plain": plain["ms_per_step"], "under_rocprofv3": prof["ms_per_step"],
                                         "ratio": prof["ms_per_step"] / plain["ms_per_step"]},
                     "ntt_pass_avg_launch_ms": {"plain": plain["roofline"]["avg_launch_ms"], "under_rocprofv3": prof["roofline"]["avg_launch_ms"],
                                                "ratio": prof["roofline"]["avg_launch_ms"] / plain["roofline"]["avg_launch_ms"]},
                     "msm_ms_per_step": {"plain": plain["msm"]["ms_per_step"], "under_rocprofv3": prof["msm"]["ms_per_step"],
                                         "ratio": prof["msm"]["ms_per_step"] / plain["msm"]["ms_per_step"]},
                     "what": "the same bench.py command with and without rocprofv3 --kernel-trace --stats in one gpurun call: kernels "
                             "run slower under the profiler, so kernel_stats.csv averages are compared with the line printed inside "
                             "the profiled process (bench_under_rocprof.json), not with the plain one"}
        except Exception as e:   # noqa: BLE001
            ratio = {"error": str(e)[:200]}
    src = "profiles/traffic_latest.json (tools/profile_all.sh, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over bench.py)"
    json.dump({"ntt": {"log2n": ntt_log2n, "hbm_bytes_per_launch": per_launch, "source": src},
               "msm": {"log2n": msm_log2n, "hbm_bytes_per_launch": msm_total / n_msm, "msms_in_profile": n_msm, "source": src,
                       "what": "HBM bytes of all MSM kernels of one MSM (sum over kernels of mean bytes x launches / MSMs run)"},
               "profiled_vs_unprofiled": ratio,
               "kernels": rows,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs; KiB units; FETCH_SIZE doubled (gfx950 streaming reads) except for the whole-line gathers of msm_accumulate_kernel, calibrated at factor 1 on their known line count"},
              open(out, "w"), indent=1)
    print(json.dumps({"ntt_hbm_bytes_per_launch": per_launch, "msm_hbm_bytes_per_msm": msm_total / n_msm}))


if __name__ == "__main__":
    main()
