export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# usage: ab_ntt_env.sh "VAR=val" ... : one NTT bench per setting ("-" = defaults)
for kv in "$@"; do if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi; env $pre python bench.py --workload ntt --no-cpu-baseline --no-host-path --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$kv', round(j['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in j['kernel_times_ms'].items()})"; done
