#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc runs (FETCH_SIZE, WRITE_SIZE; separate passes, MI355X_MICROARCH.md §HBM) into
profiles/traffic_latest.json: HBM bytes per launch of the NTT pass kernel.
gfx950 corrections: counters are in KiB; FETCH_SIZE reads exactly 1/2 of a wide coalesced stream -> doubled."""
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
    fetch_dir, write_dir, log2n, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    fe, wr = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
    rows = {}
    for k in sorted(set(fe) | set(wr)):
        f = fe.get(k, (0, 0))[0] * 1024 * 2      # KiB -> bytes, x2 gfx950 correction
        w = wr.get(k, (0, 0))[0] * 1024
        rows[k] = {"fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes": f + w, "launches": fe.get(k, wr.get(k))[1]}
    ntt = [v for k, v in rows.items() if "ntt_pass_kernel" in k]
    per_launch = sum(v["hbm_bytes"] * v["launches"] for v in ntt) / max(sum(v["launches"] for v in ntt), 1)
    json.dump({"log2n": log2n, "hbm_bytes_per_launch": per_launch, "source": out, "kernels": rows,
               "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate runs; KiB units; FETCH_SIZE doubled (gfx950)"},
              open(out, "w"), indent=1)
    print(json.dumps({"hbm_bytes_per_launch": per_launch}))


if __name__ == "__main__":
    main()
