#!/usr/bin/env python3
"""Per-kernel times of the Groth16 h-coefficient pipeline (2^20 gates) and one FRI commit-phase layer (2^20 coefficients,
blow-up 4) through their host entry points (kernel times from the library's own events; the wall time includes PCIe)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lambda_elliptic_curves_amd import _lib, groth16, merkle
from tests import util
n = 1 << 20
l, r, o = (util.rand_elems("fr381", n, s) for s in (1, 2, 3))
groth16.calculate_h_coefficients(l, r, o, n)
_lib.profile_begin()
t0 = time.perf_counter()
groth16.calculate_h_coefficients(l, r, o, n)
dt = time.perf_counter() - t0
prof = _lib.profile_end()
print("groth16 h 2^20 gates: %.2f ms wall" % (dt * 1e3), {k: (v[0], round(v[1], 3)) for k, v in prof.items()}, "kernel total %.3f ms" % sum(v[1] for v in prof.values()), flush=True)
fld = util.field_pairs()["stark252"][0]
co = util.rand_elems("stark252", n, 5)
zeta = util.rand_elems("stark252", 1, 6)[0]
off = util.offset_elem("stark252", 3)
merkle.fri_layer(fld, co, zeta, off, 2 * n)
_lib.profile_begin()
t0 = time.perf_counter()
merkle.fri_layer(fld, co, zeta, off, 2 * n)
dt = time.perf_counter() - t0
prof = _lib.profile_end()
print("fri layer 2^20 -> domain 2^21: %.2f ms wall" % (dt * 1e3), {k: (v[0], round(v[1], 3)) for k, v in prof.items()}, "kernel total %.3f ms" % sum(v[1] for v in prof.values()), flush=True)
