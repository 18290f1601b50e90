export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# usage: ab_ntt_sizes.sh "L L ..." "VAR=val|-" ... : Stark252 NTT per size and setting
Ls=$1; shift
for kv in "$@"; do if [ "$kv" = "-" ]; then pre=""; else pre="$kv"; fi; for L in $Ls; do env $pre python bench.py --workload ntt --log2n $L --no-cpu-baseline --no-host-path --steps 30 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$kv 2^$L', round(j['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in j['kernel_times_ms'].items()})"; done; done
