export LW_HIP_TUNING=1   # the library reads its A/B switches only with this set
# A/B: Merkle levels per launch (LW_HIP_MERKLE_FUSE: 1 = a launch per level, default 4), DESIGN 4.6
for f in 1 4 1 4 6; do echo "fuse=$f"; LW_HIP_MERKLE_FUSE=$f python tools/ab_merkle.py 2>/dev/null | tail -4; LW_HIP_MERKLE_FUSE=$f python tools/fri_chain.py 5 2>/dev/null | tail -1; done
