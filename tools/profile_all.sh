#!/bin/bash
# Regenerates everything under profiles/ that bench.py's numbers are checked against.  Run on the GPU box from the
# repository root (gpurun -- 'bash tools/profile_all.sh TAG'); outputs land in gpurun_out/prof_TAG/ and are copied to
# profiles/ by hand afterwards.  Counter passes are separate runs with --kernel-trace only (no other trace domains).
set -e -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="bench.py --no-cpu-baseline --steps 5 --warmup 2"
python3 bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $B > $OUT/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o pmc -- python3 $B --workload ntt > $OUT/pmc_$c.log 2>&1
done
for c in VALUBusy LDSBankConflict MeanOccupancyPerCU; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o pmc -- python3 $B --msm-log2n 22 > $OUT/pmc_$c.log 2>&1 || echo "counter $c failed" >> $OUT/failed.txt
done
python3 tools/parse_pmc.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE 24 $OUT/traffic_latest.json
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_VALUBusy $OUT/pmc_LDSBankConflict $OUT/pmc_MeanOccupancyPerCU > $OUT/pmc_summary.csv
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
# keep the merged-back payload small: the per-dispatch traces are not needed once summarised
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls -la $OUT
