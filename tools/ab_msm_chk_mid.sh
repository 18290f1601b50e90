export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: run length of the batch inversion (LW_HIP_MSM_CHK) at the sizes where the normalisation is the longer of the two
# concurrent pre-phases (2^21 .. 2^23)
for L in ${@:-21 22 23}; do for chk in 32 64 128 32 64 128; do LW_HIP_MSM_CHK=$chk python bench.py --steps 8 --warmup 2 --workload msm --msm-log2n $L --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('2^$L chk=$chk', round(m['ms_per_step'],3), {k:round(v['avg_ms']*v['launches']/m['steps'],3) for k,v in m['kernel_times_ms'].items() if 'affine' in k or 'digits' in k or 'accumulate_kernel' == k})"; done; done
