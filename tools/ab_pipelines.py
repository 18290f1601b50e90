#!/usr/bin/env python3
"""The SURVEY 8(f) pipelines through their host entry points (every layer / vector crosses PCIe both ways) and through the
device-resident ones (only challenges in and roots / one point out): wall time per call and the library's kernel times.
  * Groth16: calculate_h_coefficients for 2^20 gates, then the MSM of h against an SRS prefix (prover.rs:68-72,97-101)
  * FRI: the whole commit phase from 2^20 coefficients, blow-up 2, down to the constant (fri/mod.rs:22-75)"""
import hashlib
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from lambda_elliptic_curves_amd import _lib, fft, groth16, merkle, msm
from tools.synth import distinct_points

P_STARK = 0x800000000000011000000000000000000000000000000000000000000000001


def elems256(n, top_bits, seed):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    a[:, 0] &= np.uint64((1 << top_bits) - 1)
    return a


def stark_mont(v):
    m = v * (1 << 256) % P_STARK
    return np.array([(m >> (64 * (3 - k))) & ((1 << 64) - 1) for k in range(4)], dtype=np.uint64)


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    _lib.profile_begin()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    prof = _lib.profile_end()
    return dt * 1e3, sum(v[1] for v in prof.values()) / reps


# ---- Groth16 h + MSM
g = 1 << 20
l, r, o = (elems256(g, 62, s) for s in (1, 2, 3))
srs_n = 1 << 20
srs = msm.Srs(msm.BLS12381Curve, t_points=distinct_points(msm.BLS12381Curve, srs_n), n=srs_n)
host = lambda: srs.msm_fr(groth16.calculate_h_coefficients(l, r, o, g, strip=False)[:srs_n])
t_l, t_r, t_o = (torch.from_numpy(x.view(np.int64)).cuda() for x in (l, r, o))
t_h = torch.empty((2 * g, 4), dtype=torch.int64, device="cuda")
dev = lambda: srs.msm_fr_device(groth16.calculate_h_coefficients_device(t_l, t_r, t_o, g, g, t_out=t_h), srs_n)
wh, kh = timed(host)
wd, kd = timed(dev)
print("groth16 h (2^20 gates) + MSM of 2^20 coefficients: host entry points %.2f ms wall (%.2f ms kernels)   device-resident %.2f ms wall (%.2f ms kernels)"
      % (wh, kh, wd, kd), flush=True)
srs.close()

# ---- FRI commit phase
L = 20
n = 1 << L
fld = fft.Stark252PrimeField
co = elems256(n, 59, 5)
offs = [stark_mont(pow(3, 1 << k, P_STARK)) for k in range(L + 1)]


def zeta_of(state):
    return stark_mont(int.from_bytes(hashlib.sha256(state).digest()[:31], "big"))


def fri_host():
    poly, dom, state = co, 2 * n, b"t"
    for k in range(1, L + 1):
        dom //= 2
        poly, ev, root = merkle.fri_layer(fld, poly, zeta_of(state), offs[k], dom)
        state = hashlib.sha256(state + root).digest()
        if poly.shape[0] == 0:
            break
    return poly


t_co = torch.from_numpy(co.view(np.int64)).cuda()


def fri_dev():
    st = {"s": b"t"}

    def sample():
        return zeta_of(st["s"])

    def absorb(root):
        st["s"] = hashlib.sha256(st["s"] + root).digest()
    return merkle.fri_commit_phase_device(fld, L + 1, t_co, n, sample, absorb, lambda k: offs[k], 2 * n)


wh, kh = timed(fri_host, 2)
wd, kd = timed(fri_dev, 2)
print("FRI commit phase, 2^20 coefficients, blow-up 2, %d layers: host entry points %.2f ms wall (%.2f ms kernels)   device-resident %.2f ms wall (%.2f ms kernels)"
      % (L, wh, kh, wd, kd), flush=True)
