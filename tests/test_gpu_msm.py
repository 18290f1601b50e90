"""GPU parity: HIP Pippenger MSM (through the C ABI) vs the CPU oracle's restatement of msm::pippenger::msm.
Equality is on the canonical affine image, as the reference's own PartialEq is up to projective scaling
(math/src/elliptic_curve/point.rs:57-63).  Mirrors math/src/msm/pippenger.rs:204-233."""
import numpy as np
import pytest

from oracle import bigint_def as D
from oracle import oracle as O
from tests import util

pytestmark = pytest.mark.gpu
CURVES = ["bls12_381_g1", "bn254_g1", "bn254_g2", "bls12_381_g2"]


def aff(oid, p):
    return O.point_to_affine_ints(oid, p)


@pytest.mark.parametrize("name", CURVES)
@pytest.mark.parametrize("n", [1, 2, 3, 17, 30, 100, 1000])
def test_msm_matches_reference_algorithm_small(name, n):
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()[name]
    scalars, points = util.msm_case(oid, n, 100 + n)
    got = msm.msm(crv, scalars, points)
    exp = O.msm(oid, scalars, points)
    assert aff(oid, got) == aff(oid, exp)
    assert O.ec_eq(oid, got, exp)               # the reference's own equality
    # output is normalised: Z == 1 (Montgomery one)
    w = crv.coord_words
    bf = O.CURVE_BASE_FIELD[oid]
    assert O.limbs_to_int(got[2 * w:2 * w + O.FIELD_WORDS[bf]]) == O.field_params(bf)["one"]


@pytest.mark.parametrize("name,n", [("bls12_381_g1", 1 << 12), ("bls12_381_g1", 1 << 16), ("bn254_g1", 1 << 14),
                                    ("bn254_g2", 1 << 11), ("bls12_381_g2", 1 << 10)])
def test_msm_matches_reference_algorithm_medium(name, n):
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()[name]
    scalars, points = util.msm_case(oid, n, 7 + n)
    got = msm.msm(crv, scalars, points)
    exp = O.parallel_msm_with(oid, scalars, points, max(2, O.optimum_window_size(n)), 16)
    assert aff(oid, got) == aff(oid, exp)


def test_msm_vs_affine_bigint_definition():
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    cd = D.BLS12_381_G1
    rng = np.random.default_rng(3)
    mult = [int(x) for x in rng.integers(1, 1 << 40, size=20)]
    g = util.generator(oid)
    pts = np.stack([O.ec_mul(oid, g, m, 1) for m in mult])
    ks = [int.from_bytes(rng.bytes(32), "big") for _ in mult]
    got = msm.msm(crv, O.ints_to_array(ks, 4), pts)
    total = sum(k * m for k, m in zip(ks, mult))
    assert aff(oid, got) == cd.tup(cd.mul(total, cd.gen))


@pytest.mark.parametrize("name", ["bls12_381_g1", "bn254_g1"])
def test_msm_edge_cases(name):
    from lambda_elliptic_curves_amd import errors, msm
    crv, oid = util.curve_pairs()[name]
    g = util.generator(oid)
    neutral = O.ec_neutral(oid)
    # empty input -> neutral element (pippenger.rs:102)
    out = msm.msm(crv, np.zeros((0, 4), np.uint64), np.zeros((0, crv.point_words), np.uint64))
    assert aff(oid, out) is None and np.array_equal(out, neutral)
    with pytest.raises(errors.LengthMismatch):      # pippenger.rs:25-27
        msm.msm(crv, np.zeros((2, 4), np.uint64), np.stack([g]))
    r = D.P_FR381 if name == "bls12_381_g1" else D.P_FR254
    p = O.ec_mul(oid, g, 12345, 1)
    # duplicates, P and -P in one bucket, identity inputs, zero scalars, scalar = r-1, scalar = 2^256-1
    pts = np.stack([p, p, O.ec_neg(oid, p), neutral, g, g, p])
    ks = [7, 7, 7, 99, r - 1, 0, (1 << 256) - 1]
    got = msm.msm(crv, O.ints_to_array(ks, 4), pts)
    exp = O.msm(oid, O.ints_to_array(ks, 4), pts)
    assert aff(oid, got) == aff(oid, exp)
    # everything cancels -> neutral
    pts = np.stack([p, O.ec_neg(oid, p)])
    got = msm.msm(crv, O.ints_to_array([5, 5], 4), pts)
    assert aff(oid, got) is None


def test_msm_skewed_scalars_all_equal():
    # every point lands in the same bucket of every window: the segmented reduction needs extra rounds
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 20000
    _, points = util.msm_case(oid, n, 5)
    k = 0x1234567890abcdef1234567890abcdef1234567890abcdef1234567890abcdef
    scalars = np.tile(O.int_to_limbs(k, 4), (n, 1))
    got = msm.msm(crv, scalars, points)
    acc = O.ec_neutral(oid)          # sum of points, then one scalar multiplication
    for i in range(n):
        acc = O.ec_add(oid, acc, points[i])
    assert aff(oid, got) == aff(oid, O.ec_mul(oid, acc, k))
    # all points equal, distinct scalars (P+P doubling inside every bucket chain)
    n2 = 3000
    scalars2, _ = util.msm_case(oid, n2, 6)
    pts2 = np.tile(points[0], (n2, 1))
    got2 = msm.msm(crv, scalars2, pts2)
    tot = sum(O.array_to_ints(scalars2))
    assert aff(oid, got2) == aff(oid, O.ec_mul(oid, points[0], tot, 5))


def test_msm_device_resident_large_linearity():
    # 2^20 points: MSM(k, P) + MSM(k', P) == MSM(k + k', P) (size-independent property)
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 1 << 20
    s1, points = util.msm_case(oid, n, 11)
    rng = np.random.default_rng(13)
    s2 = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    s1[:, 0] &= np.uint64((1 << 62) - 1)      # keep k + k' < 2^256
    ssum = np.zeros_like(s1)
    carry = np.zeros(n, dtype=np.uint64)
    for limb in (3, 2, 1, 0):
        a, b = s1[:, limb], s2[:, limb]
        t = a + b
        c1 = (t < a).astype(np.uint64)
        t2 = t + carry
        c2 = (t2 < t).astype(np.uint64)
        ssum[:, limb] = t2
        carry = c1 + c2
    tp = torch.from_numpy(points.view(np.int64)).cuda()
    outs = []
    for s in (s1, s2, ssum):
        ts = torch.from_numpy(s.view(np.int64)).cuda()
        outs.append(msm.msm_device(crv, ts, tp, n))
    lhs = O.ec_add(oid, outs[0], outs[1])
    assert aff(oid, lhs) == aff(oid, outs[2])


@pytest.mark.parametrize("name,fr", [("bls12_381_g1", O.F_FR381), ("bn254_g1", O.F_FR254), ("bn254_g2", O.F_FR254)])
def test_msm_over_montgomery_scalars_equals_representative_then_msm(name, fr):
    # what groth16/kzg callers do: scalars.map(representative) then msm (provers/groth16/src/prover.rs:69-85)
    from lambda_elliptic_curves_amd import msm
    from oracle import bigint_def as D
    crv, oid = util.curve_pairs()[name]
    n = 500
    _, points = util.msm_case(oid, n, 31)
    r = D.P_FR381 if fr == O.F_FR381 else D.P_FR254
    rng = np.random.default_rng(9)
    ks = [int.from_bytes(rng.bytes(32), "big") % r for _ in range(n)]
    ks[0], ks[1] = 0, r - 1
    canon = O.ints_to_array(ks, 4)
    mont = O.ints_to_array([O.to_mont(fr, k) for k in ks], 4)
    got = msm.msm_fr(crv, mont, points)
    assert aff(oid, got) == aff(oid, O.msm(oid, canon, points))
    assert aff(oid, got) == aff(oid, msm.msm(crv, canon, points))


def test_msm_above_2_24_points_splits_additively():
    # N > 2^24 takes the 64-bit sort items (index no longer fits the packed 32-bit form) and BASELINE's full size
    # sits exactly on the boundary: MSM over 2^24 + 4096 pairs must equal MSM(first 2^24) + MSM(last 4096), and the
    # 2^24 part runs the packed path.  BN254 G1 (96-byte points) keeps the buffers small.
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bn254_g1"]
    base_n, n_lo, n_hi = 1 << 12, 1 << 24, 1 << 12
    _, base = util.msm_case(oid, base_n, 21)
    n = n_lo + n_hi
    rng = np.random.default_rng(22)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    tp = torch.from_numpy(base.view(np.int64)).cuda().repeat(n // base_n, 1)
    ts = torch.from_numpy(sc.view(np.int64)).cuda()
    whole = msm.msm_device(crv, ts, tp, n)
    lo = msm.msm_device(crv, ts[:n_lo], tp[:n_lo], n_lo)
    hi = msm.msm_device(crv, ts[n_lo:].contiguous(), tp[n_lo:].contiguous(), n_hi)
    assert aff(oid, O.ec_add(oid, lo, hi)) == aff(oid, whole)
    # and the small tail agrees with the oracle's Pippenger outright
    want = O.msm(oid, sc[n_lo:], np.ascontiguousarray(np.tile(base, (n // base_n, 1))[n_lo:]))
    assert aff(oid, hi) == aff(oid, want)


def test_msm_large_input_with_identity_rows():
    # From 2^22 points the library normalises the inputs (batch inversion over strided runs of 128) before the
    # accumulation; identity rows (z = 0) must drop out of every run.  Replacing them by any other point with a zero
    # scalar gives the same sum.
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bn254_g1"]
    n, base_n = 1 << 22, 1 << 12
    _, base = util.msm_case(oid, base_n, 31)
    rng = np.random.default_rng(32)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    pts = np.tile(base, (n // base_n, 1))
    hole = rng.random(n) < 0.2
    hole[[0, 1, n - 1]] = True
    with_id = pts.copy()
    with_id[hole] = O.ec_neutral(oid)
    sc_zeroed = sc.copy()
    sc_zeroed[hole] = 0
    got = msm.msm_device(crv, torch.from_numpy(sc.view(np.int64)).cuda(), torch.from_numpy(with_id.view(np.int64)).cuda(), n)
    ref = msm.msm_device(crv, torch.from_numpy(sc_zeroed.view(np.int64)).cuda(), torch.from_numpy(pts.view(np.int64)).cuda(), n)
    assert aff(oid, got) == aff(oid, ref)
    # anchor the pair to the oracle on a slice small enough for the CPU
    m = 1 << 12
    small = msm.msm(crv, sc[:m], with_id[:m])
    assert aff(oid, small) == aff(oid, O.msm(oid, sc[:m], with_id[:m]))


@pytest.mark.parametrize("name,log_n", [("bls12_381_g1", 21), ("bn254_g2", 19)])
def test_msm_host_buffers_large_equals_device_resident(name, log_n):
    """Host-buffer calls of normalised sizes upload the points in chunks of 2^20 under the sort and normalise each chunk as it
    arrives (csrc/msm.hip msm_device, h_points): a size that is not a multiple of the chunk, with identity rows at the chunk
    seams, must give the device-resident call's result; a prefix is anchored to the oracle."""
    import torch
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()[name]
    n, base_n = (1 << log_n) + 777, 1 << 11
    _, base = util.msm_case(oid, base_n, 61)
    rng = np.random.default_rng(62)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64)
    pts = np.tile(base, (n // base_n + 1, 1))[:n].copy()
    for i in (0, (1 << 20) - 1, 1 << 20, (1 << 20) + 1, n - 1):
        if i < n:
            pts[i] = O.ec_neutral(oid)
    host = msm.msm(crv, sc, pts)
    dev = msm.msm_device(crv, torch.from_numpy(sc.view(np.int64)).cuda(), torch.from_numpy(pts.view(np.int64)).cuda(), n)
    assert aff(oid, host) == aff(oid, dev)
    m = 1 << 11
    assert aff(oid, msm.msm(crv, sc[:m], pts[:m])) == aff(oid, O.msm(oid, sc[:m], pts[:m]))


@pytest.mark.parametrize("name", CURVES)
def test_batched_group_law_outer_addition(name):
    # lw_hip_ec_add_outer_device == IsGroup::operate_with (short_weierstrass/point.rs:171-207) pair by pair, including
    # P + P, P + (-P) and the identity on either side
    import ctypes as C
    import torch
    from lambda_elliptic_curves_amd import _lib
    crv, oid = util.curve_pairs()[name]
    _, pts = util.msm_case(oid, 9, 77)
    rows = np.concatenate([pts[:5], O.ec_neutral(oid)[None, :]])
    cols = np.stack([pts[5], pts[0], O.ec_neg(oid, pts[1]), O.ec_neutral(oid)])
    tr, tc = torch.from_numpy(rows.view(np.int64)).cuda(), torch.from_numpy(cols.view(np.int64)).cuda()
    out = torch.empty((len(rows) * len(cols), crv.point_words), dtype=torch.int64, device="cuda")
    rc = _lib.lib().lw_hip_ec_add_outer_device(crv.curve, C.c_void_p(tr.data_ptr()), len(rows), C.c_void_p(tc.data_ptr()), len(cols),
                                               C.c_void_p(out.data_ptr()), None)
    assert rc == 0, _lib.last_error()
    torch.cuda.synchronize()
    got = out.cpu().numpy().view(np.uint64)
    for j in range(len(cols)):
        for i in range(len(rows)):
            assert aff(oid, got[j * len(rows) + i]) == aff(oid, O.ec_add(oid, rows[i], cols[j])), (i, j)


@pytest.mark.parametrize("name,n", [("bls12_381_g1", 3000), ("bn254_g1", 1500), ("bn254_g2", 400), ("bls12_381_g2", 300)])
def test_msm_every_window_width_matches_oracle(name, n, monkeypatch):
    """The device recodes scalars into SIGNED c-bit digits (msm.hip: |d| <= 2^(c-1), W = ceil(257/c) windows, bucket j =
    multiplier j + 1) where the reference uses unsigned ones (pippenger.rs:76-81); the sum is the same group element for
    every c.  LW_HIP_MSM_C forces the width (read per call): all widths the sort supports — narrow and wide items, coarse /
    fine key splits, windows that start at bit 256 — against the oracle on scalars that exercise the carry chain: random
    256-bit values (not reduced), 2^256 - 1, 2^255, r - 1, digits equal to exactly half a window, zeros and ones."""
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()[name]
    _, points = util.msm_case(oid, n, 900 + n)
    rng = np.random.default_rng(n)
    r = D.P_FR381 if name.startswith("bls") else D.P_FR254
    ks = [int.from_bytes(rng.bytes(32), "big") for _ in range(n)]
    special = [(1 << 256) - 1, 1 << 255, r - 1, 0, 1, 2, (1 << 255) - 1, 1 << 128, (1 << 128) - 1]
    for c in range(3, 21):       # every digit of the scalar = 2^(c-1): the largest digit that does not carry; and 2^(c-1) + 1: the smallest that does
        special.append(sum((1 << (c - 1)) << (c * w) for w in range(256 // c)))
        special.append(sum(((1 << (c - 1)) + 1) << (c * w) for w in range(256 // c)))
    for i, v in enumerate(special):
        ks[(i * 7) % n] = v & ((1 << 256) - 1)
    scalars = O.ints_to_array(ks, 4)
    exp = aff(oid, O.parallel_msm_with(oid, scalars, points, 8, 16))
    for c in range(3, 21):
        monkeypatch.setenv("LW_HIP_MSM_C", str(c))
        got = msm.msm(crv, scalars, points)
        assert aff(oid, got) == exp, f"window width {c}"


@pytest.mark.parametrize("name,n", [("bls12_381_g1", 5000), ("bn254_g1", 3000), ("bn254_g2", 700), ("bls12_381_g2", 500)])
def test_msm_bucket_reduce_pair_and_quad_kernels_agree_with_oracle(name, n, monkeypatch):
    """The running sums over the buckets (pippenger.rs:85-98) are hierarchical; a level runs on two lanes per group or — where
    the chain of dependent additions is all there is — on eight, with every complete addition spread over a quad of lanes
    (msm_group_sum_quad_kernel, ec.cuh pt_add_quad: doublings as p + p, the identity as an operand, partial last groups).
    LW_HIP_MSM_QUAD = log2 of the widest level in lanes that takes the quad kernels (read per call): none, the default, all
    of them; window widths whose bucket count is and is not a multiple of the group size."""
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()[name]
    scalars, points = util.msm_case(oid, n, 7100 + n)
    scalars[5] = 0                       # empty contribution
    scalars[6:9] = scalars[9:12]         # repeated scalars on different points
    exp = aff(oid, O.parallel_msm_with(oid, scalars, points, 8, 16))
    for c in (5, 8, 11, 16):
        monkeypatch.setenv("LW_HIP_MSM_C", str(c))
        for q in ("0", "18", "30"):
            monkeypatch.setenv("LW_HIP_MSM_QUAD", q)
            assert aff(oid, msm.msm(crv, scalars, points)) == exp, f"c = {c}, quad = {q}"
        # the accumulation itself on one lane per piece or on four (msm_accumulate_quad_kernel: signed entries, the ordered
        # dispatch, multi-round partial sums, empty keys), LW_HIP_MSM_ACCQ read per call; short pieces force several rounds
        monkeypatch.setenv("LW_HIP_MSM_CH", "4")
        for aq in ("0", "30"):
            monkeypatch.setenv("LW_HIP_MSM_ACCQ", aq)
            assert aff(oid, msm.msm(crv, scalars, points)) == exp, f"c = {c}, accq = {aq}"
        monkeypatch.delenv("LW_HIP_MSM_ACCQ")
        monkeypatch.delenv("LW_HIP_MSM_CH")


def test_msm_wide_windows_at_scale_match_oracle(monkeypatch):
    """c = 20 (the width used from 2^23 points, 64-bit items, 512 coarse bins x 1024 keys) and c = 19 on 2^17 points: buckets of
    ~0.25 items, the short-top-window path with multi-round partial sums, and the ordered piece dispatch."""
    from lambda_elliptic_curves_amd import msm
    crv, oid = util.curve_pairs()["bls12_381_g1"]
    n = 1 << 17
    scalars, points = util.msm_case(oid, n, 4242, threads=util.host_threads())
    exp = aff(oid, O.parallel_msm_with(oid, scalars, points, 14, util.host_threads()))
    for c in (19, 20, 13):
        monkeypatch.setenv("LW_HIP_MSM_C", str(c))
        assert aff(oid, msm.msm(crv, scalars, points)) == exp, f"window width {c}"
