export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: buckets per running-sum group (LW_HIP_MSM_GLOG = log2, default 3) with the quad kernels on the chain-bound levels
for L in ${@:-16 20 24}; do for g in 2 3 4 2 3 4; do LW_HIP_MSM_GLOG=$g python bench.py --steps 8 --warmup 3 --workload msm --msm-log2n $L --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('2^$L glog=$g', round(m['ms_per_step'],3), {k:round(v['avg_ms']*v['launches']/m['steps'],3) for k,v in m['kernel_times_ms'].items() if 'group_sum' in k or 'combine' in k})"; done; done
