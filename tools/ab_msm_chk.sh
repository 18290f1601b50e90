export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
for chk in 128 64 96; do LW_HIP_MSM_CHK=$chk python bench.py --steps 10 --warmup 2 --workload msm --no-cpu-baseline --no-host-path 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read()); m=j['msm']; print('chk=$chk', round(m['ms_per_step'],2), {k:round(v['avg_ms'],3) for k,v in m['kernel_times_ms'].items()})"; done
