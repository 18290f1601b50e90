"""Synthetic inputs for the tools/ scripts, in the reference's memory layout, WITHOUT the CPU checker (tests/util.py pulls in
oracle/, which only tests/, smoke() and bench.py's cpu_baseline leg may use)."""
import numpy as np

P_BABYBEAR = 2013265921
P_STARK252 = 0x800000000000011000000000000000000000000000000000000000000000001
P_FR381 = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001


def field_pairs():
    """name -> (product Field, None): the shape of tests/util.field_pairs() without the checker's field ids"""
    from lambda_elliptic_curves_amd import fft
    return {"stark252": (fft.Stark252PrimeField, None), "fr381": (fft.FrField, None),
            "babybear_u64": (fft.Babybear31PrimeField, None), "babybear_u32": (fft.Babybear31PrimeFieldU32, None),
            "babybear_ext4": (fft.Degree4BabyBearExtensionField, None)}


def curve_pairs():
    from lambda_elliptic_curves_amd import msm
    return {"bls12_381_g1": (msm.BLS12381Curve, None), "bn254_g1": (msm.BN254Curve, None),
            "bn254_g2": (msm.BN254TwistCurve, None), "bls12_381_g2": (msm.BLS12381TwistCurve, None)}


def rand_elems(name, n, seed):
    """n canonical residues (any value < p is a valid Montgomery-form element)"""
    rng = np.random.default_rng(seed)
    if name in ("stark252", "fr381"):
        a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
        a[:, 0] &= np.uint64((1 << (59 if name == "stark252" else 62)) - 1)
        return a
    if name == "babybear_u32":
        return rng.integers(0, P_BABYBEAR, size=n, dtype=np.uint32)
    if name == "babybear_u64":
        return rng.integers(0, P_BABYBEAR, size=n, dtype=np.uint64)
    if name == "babybear_ext4":
        return rng.integers(0, P_BABYBEAR, size=(n, 4), dtype=np.uint64)
    raise KeyError(name)


def offset_elem(name, h):
    """Coset offset h (small canonical int) as one domain-field element in memory (Montgomery) form"""
    if name == "babybear_u32":
        return np.array([h * (1 << 32) % P_BABYBEAR], dtype=np.uint32)
    if name in ("babybear_u64", "babybear_ext4"):
        return np.array([h * (1 << 64) % P_BABYBEAR], dtype=np.uint64)
    p = P_STARK252 if name == "stark252" else P_FR381
    m = h * (1 << 256) % p
    return np.array([(m >> (64 * (3 - k))) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)
