export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: piece length of the accumulation (LW_HIP_MSM_CH) on small MSMs, where a work-item's chain IS the kernel time
for L in ${@:-10 12 14 16}; do for ch in 4 8 16 4 8 16; do LW_HIP_MSM_CH=$ch python bench.py --steps 10 --warmup 3 --workload msm --msm-log2n $L --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('2^$L ch=$ch', round(m['ms_per_step'],3), {k:round(v['avg_ms']*v['launches']/m['steps'],3) for k,v in m['kernel_times_ms'].items() if 'accumulate' in k})"; done; done
