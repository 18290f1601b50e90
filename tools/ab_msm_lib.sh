export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# usage: ab_msm_lib.sh : MSM bench (MSM_LOG2N) with lib/liblw_hip.so, then with lib/liblw_old.so swapped in, same box
bash tools/ab_msm_env.sh - | cut -c1-420
cp lambda_elliptic_curves_amd/lib/liblw_hip.so /tmp/liblw_keep.so
cp lambda_elliptic_curves_amd/lib/liblw_old.so lambda_elliptic_curves_amd/lib/liblw_hip.so
echo OLD; bash tools/ab_msm_env.sh - | cut -c1-420
cp /tmp/liblw_keep.so lambda_elliptic_curves_amd/lib/liblw_hip.so
echo NEW; bash tools/ab_msm_env.sh - | cut -c1-420
