export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# usage: ab_msm_csweep.sh "L L ..." "c c ...": MSM wall time (tools/ab_msm_sizes.py) for every window width
for c in $2; do LW_HIP_MSM_C=$c python tools/ab_msm_sizes.py $1 2>/dev/null | tr '\n' ' '; echo; done
