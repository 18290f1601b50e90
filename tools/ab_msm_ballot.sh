export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: ranks inside the coarse sort from LDS atomics (default) or from wave-wide ballots (LW_HIP_MSM_BALLOT=1), DESIGN 4.4
for b in 0 1 0 1; do LW_HIP_MSM_BALLOT=$b python bench.py --steps 5 --warmup 2 --workload msm --msm-log2n ${1:-24} --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('ballot=$b', round(m['ms_per_step'],2), {k:round(v['avg_ms']*v['launches']/m['steps'],3) for k,v in m['kernel_times_ms'].items() if 'coarse' in k})"; done
for b in 0 1; do echo "ballot=$b"; LW_HIP_MSM_BALLOT=$b python tools/ab_msm_skew.py ${1:-24}; done
