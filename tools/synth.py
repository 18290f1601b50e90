#!/usr/bin/env python3
"""Valid, DISTINCT synthetic points for the secondary benchmarks and profiling drivers, built without the CPU checker:
{1..n}G grown from the public generator constants with the library's own batched addition
(lw_hip_ec_add_outer_device).  The sums come out of the complete projective addition, so Z != 1."""
import ctypes as C

import numpy as np


def seed_points():
    """(G, 2G, 3G, 4G) rows in the reference layout for each group, from the public generator constants via the
    library's own batched addition (identity + G, G + G, ...)."""
    import torch
    from lambda_elliptic_curves_amd import _lib, msm
    R384, R256 = 1 << 384, 1 << 256
    P381 = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    P254 = 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47

    def limbs(v, words):
        return [(v >> (64 * (words - 1 - j))) & ((1 << 64) - 1) for j in range(words)]

    def fp(v, p, R, words):
        return limbs(v * R % p, words)

    gens = {
        "BLS12381Curve": (P381, R384, 6, [(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,),
                                           (0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1,), (1,)]),
        "BN254Curve": (P254, R256, 4, [(1,), (2,), (1,)]),
        "BN254TwistCurve": (P254, R256, 4, [
            (0x1800deef121f1e76426a00665e5c4479674322d4f75edadd46debd5cd992f6ed, 0x198e9393920d483a7260bfb731fb5d25f1aa493335a9e71297e485b7aef312c2),
            (0x12c85ea5db8c6deb4aab71808dcb408fe3d1e7690c43d37b4ce6cc0166fa7daa, 0x090689d0585ff075ec9e99ad690c3395bc4b313370b38ef355acdadcd122975b),
            (1, 0)]),
        "BLS12381TwistCurve": (P381, R384, 6, [
            (0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8, 0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
            (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801, 0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be),
            (1, 0)]),
    }
    out = {}
    for crv in (msm.BLS12381Curve, msm.BN254Curve, msm.BN254TwistCurve, msm.BLS12381TwistCurve):
        p, R, w, coords = gens[crv.name]
        row = []
        for comp in coords:
            for v in comp:
                row += fp(v, p, R, w)
        g = np.array([row], dtype=np.uint64)
        cur = torch.from_numpy(g.view(np.int64)).cuda()
        rows = [cur]
        for _ in range(3):   # 2G, 3G, 4G
            nxt = torch.empty_like(cur)
            rc = _lib.lib().lw_hip_ec_add_outer_device(crv.curve, C.c_void_p(rows[-1].data_ptr()), 1, C.c_void_p(cur.data_ptr()), 1,
                                                       C.c_void_p(nxt.data_ptr()), None)
            assert rc == 0
            torch.cuda.synchronize()
            rows.append(nxt)
        out[crv.name] = torch.cat(rows).cpu().numpy().view(np.uint64)
    return out




_SEEDS = None


def distinct_points(crv, n):
    """torch int64 tensor [n, point_words] on the GPU: the points G, 2G, ..., nG of the curve's group."""
    import torch
    from lambda_elliptic_curves_amd import _lib
    global _SEEDS
    if _SEEDS is None:
        _SEEDS = seed_points()
    cur = torch.from_numpy(_SEEDS[crv.name].view(np.int64)).cuda()
    while cur.shape[0] < n:      # {1..m}G -> {1..2m}G by adding mG to every row
        m = cur.shape[0]
        out = torch.empty((m, crv.point_words), dtype=torch.int64, device="cuda")
        last = cur[m - 1:m].contiguous()
        rc = _lib.lib().lw_hip_ec_add_outer_device(crv.curve, C.c_void_p(cur.data_ptr()), m, C.c_void_p(last.data_ptr()), 1,
                                                   C.c_void_p(out.data_ptr()), None)
        assert rc == 0, _lib.last_error()
        torch.cuda.synchronize()
        cur = torch.cat([cur, out])
    return cur[:n].contiguous()
