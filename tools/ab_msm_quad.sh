export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: bucket-reduce levels with eight lanes per group and every addition spread over a quad (LW_HIP_MSM_QUAD = log2 of the
# widest level in lanes that takes them, 0 = none), DESIGN 4.4
for L in ${@:-16 20 22}; do for q in 0 16 18 20 0 18; do LW_HIP_MSM_QUAD=$q python bench.py --steps 10 --warmup 3 --workload msm --msm-log2n $L --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('2^$L quad=$q', round(m['ms_per_step'],3), {k:round(v['avg_ms']*v['launches']/m['steps'],3) for k,v in m['kernel_times_ms'].items() if 'group_sum' in k or 'combine' in k})"; done; done
