export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# MSM_LOG2N=L selects the size (default 24)
# usage: ab_msm_env.sh "VAR=val" "VAR2=val2" ... : one MSM bench per setting ("-" = defaults)
for kv in "$@"; do if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi; env $pre python bench.py --steps 10 --warmup 2 --workload msm --msm-log2n ${MSM_LOG2N:-24} --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('$kv', round(m['ms_per_step'],2), {k:round(v['avg_ms'],3) for k,v in m['kernel_times_ms'].items()})"; done
