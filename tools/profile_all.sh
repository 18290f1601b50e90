#!/bin/bash
export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# Regenerates everything under profiles/ that bench.py's numbers are checked against.  Run on the GPU box from the
# repository root (gpurun -- 'bash tools/profile_all.sh TAG'); outputs land in gpurun_out/prof_TAG/ and the summaries
# are copied to profiles/ by hand afterwards.  Counter passes are separate runs with --kernel-trace only.
set -e -o pipefail
TAG=${1:-r03}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
# the profiled command is bench.py with its default step counts (20 timed NTT steps after 5 warm-up ones; the MSM leg
# clamps itself to 5 + 2), minus the CPU legs
B="bench.py --no-cpu-baseline --no-host-path --steps 20 --warmup 5"
python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
echo "bench done" ; tail -c 600 $OUT/bench.err || true
# same command under the profiler: the JSON it prints (live HIP events) next to rocprofv3's own kernel stats
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $B > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_wl -o stats -- python3 tools/profile_workloads.py all > $OUT/stats_wl.log 2>&1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE VALUBusy LDSBankConflict MeanOccupancyPerCU; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmc_$c -o pmc -- python3 $B > $OUT/pmc_$c.log 2>&1 || echo "counter $c failed (bench)" >> $OUT/failed.txt
  echo "pmc $c bench done"
done
for c in FETCH_SIZE WRITE_SIZE VALUBusy LDSBankConflict; do
  rocprofv3 --kernel-trace --output-format csv --pmc $c -d $OUT/pmcwl_$c -o pmc -- python3 tools/profile_workloads.py ntt > $OUT/pmcwl_$c.log 2>&1 || echo "counter $c failed (workloads)" >> $OUT/failed.txt
  echo "pmc $c workloads done"
done
python3 tools/parse_pmc.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE 24 24 5 $OUT/traffic_latest.json $OUT/bench.json $OUT/bench_under_rocprof.json
python3 tools/pmc_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/pmc_VALUBusy $OUT/pmc_LDSBankConflict $OUT/pmc_MeanOccupancyPerCU > $OUT/pmc_summary.csv
python3 tools/pmc_summary.py $OUT/pmcwl_FETCH_SIZE $OUT/pmcwl_WRITE_SIZE $OUT/pmcwl_VALUBusy $OUT/pmcwl_LDSBankConflict > $OUT/pmc_summary_workloads.csv
python3 tools/kernel_medians.py $OUT/stats > $OUT/kernel_medians.csv
python3 tools/kernel_medians.py $OUT/stats_wl > $OUT/kernel_medians_workloads.csv
find $OUT/stats -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT/stats_wl -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_workloads.csv \;
# keep the merged-back payload small: the per-dispatch traces are not needed once summarised
find $OUT -name "*kernel_trace.csv" -delete
find $OUT -name "*counter_collection.csv" -delete
find $OUT -name "*.db" -delete
ls -la $OUT
