export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# Needs an ablation build of the library: make -C lambda_elliptic_curves_amd/csrc clean all ABLATION=1 (rebuild without it afterwards)
# A/B helper: LW_HIP_NTT_DBG values to compare are the arguments (default: 0 0 0)
for d in ${@:-0 0 0}; do LW_HIP_NTT_DBG=$d python bench.py --workload ntt --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('dbg=$d', round(j['ms_per_step'],4), {k:round(v['avg_ms'],4) for k,v in j['kernel_times_ms'].items()})"; done
