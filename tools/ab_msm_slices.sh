export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: window slices of the MSM (csrc/msm_core.cuh run()): LW_HIP_MSM_SLICES=2 = two slices with side-stream overlap from 2^22 points, default = one slice
for L in ${@:-24}; do for sl in 1 2 1 2; do LW_HIP_MSM_SLICES=$sl python bench.py --steps 5 --warmup 2 --workload msm --msm-log2n $L --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('2^$L slices=$sl', round(m['ms_per_step'],2), {k:round(v['avg_ms']*v['launches']/m['steps'],3) for k,v in m['kernel_times_ms'].items()})"; done; done
