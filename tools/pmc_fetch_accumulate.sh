export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# usage: pmc_fetch_accumulate.sh TAG : FETCH_SIZE (KiB, raw) of the MSM accumulate kernel for the current environment
export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d gpurun_out/pf_$1 -o pmc -- python3 bench.py --workload msm --no-cpu-baseline --no-host-path --steps 2 --warmup 1 > /dev/null 2>&1
python3 - <<PY
import csv, glob
tot = {}
for f in glob.glob("gpurun_out/pf_$1/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "msm_accumulate" in r["Kernel_Name"] and "true" in r["Kernel_Name"]:
            tot.setdefault(r["Kernel_Name"][:60], []).append(float(r["Counter_Value"]))
for k, v in tot.items(): print("$1", k, "mean raw GiB-ish: %.2f GB over %d launches" % (sum(v) / len(v) * 1024 / 1e9, len(v)))
PY
rm -rf gpurun_out/pf_$1
