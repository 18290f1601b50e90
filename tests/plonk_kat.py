"""Restatement of the PLONK prover's rounds 1-3 for its reference test circuit (provers/plonk/src/prover.rs:311-531,
test_utils/circuit_1.rs, test_utils/utils.rs) over an abstract set of polynomial operations, so the same flow runs on
the CPU oracle and on the HIP path.  Test infrastructure: it exists to reach the commitments the reference's own tests
hard-code (prover.rs:760-836).  Everything between the FFT / MSM calls is plain integer arithmetic mod r."""
from oracle import bigint_def as D

R = D.P_FR381
K1 = 7                                             # ORDER_R_MINUS_1_ROOT_UNITY (test_utils/utils.rs:26)
PERMUTATION = [11, 3, 0, 1, 2, 4, 6, 10, 5, 8, 7, 9]  # circuit_1.rs:27
N = 4
BETA = 0x0bdda7414bdf5bf42b77cbb3af4a82f32ec7622dd6c71575bede021e6e4609d4
GAMMA = 0x58f6690d9b36e62e4a0aef27612819288df2a3ff5bf01597cf06779503f51583
ALPHA = 0x583cfb0df2ef98f2131d717bc6aadd571c5302597c135cab7c00435817bf6e50


def strip(c):
    c = list(c)
    while c and c[-1] % R == 0:
        c.pop()
    return c


def rounds_1_to_3(ops, omega):
    """ops: interp(evals) -> coeffs; eval_offset(coeffs, domain_size, offset) -> evals; interp_offset(evals, offset) ->
    coeffs; commit(coeffs) -> affine (x, y) or None.  All values canonical integers mod r.  Returns the commitments."""
    x, e = 2, 2
    y = x * e % R
    wa, wb, wc = [x, y, x, y], [x, x, e, y], [x, x, y, x]          # test_witness_1 (circuit_1.rs:91-115)
    domain = [pow(omega, i, R) for i in range(N)]
    identity = [pow(omega, row, R) * pow(K1, col, R) % R for col in range(3) for row in range(N)]
    permuted = [identity[PERMUTATION[i]] for i in range(3 * N)]
    s1l, s2l, s3l = permuted[:4], permuted[4:8], permuted[8:]
    neg1 = R - 1
    ql, qr, qo = ops.interp([neg1, neg1, 0, 1]), ops.interp([0, 0, 0, neg1]), ops.interp([0, 0, neg1, 0])
    qm, qc = ops.interp([0, 0, 1, 0]), ops.interp([0, 0, 0, 0])
    s1, s2, s3 = ops.interp(s1l), ops.interp(s2l), ops.interp(s3l)
    out = {}
    # round 1 (zero blinding: TestRandomFieldGenerator)
    p_a, p_b, p_c = strip(ops.interp(wa)), strip(ops.interp(wb)), strip(ops.interp(wc))
    out["a_1"], out["b_1"], out["c_1"] = ops.commit(p_a), ops.commit(p_b), ops.commit(p_c)
    # round 2: the permutation accumulator
    k2 = K1 * K1 % R
    lp = lambda w, eta: (w + BETA * eta + GAMMA) % R
    zc = [1]
    for i in range(N - 1):
        num = lp(wa[i], domain[i]) * lp(wb[i], domain[i] * K1 % R) * lp(wc[i], domain[i] * k2 % R) % R
        den = lp(wa[i], s1l[i]) * lp(wb[i], s2l[i]) * lp(wc[i], s3l[i]) % R
        zc.append(zc[-1] * num % R * pow(den, -1, R) % R)
    p_z = strip(ops.interp(zc))
    out["z_1"] = ops.commit(p_z)
    # round 3: the quotient, in evaluation form on the coset K1 * <w_16>
    z_x_omega = strip([c * domain[i % N] % R for i, c in enumerate(p_z)])
    l1 = ops.interp([1, 0, 0, 0])
    p_pi = ops.interp([2, 4, 0, 0])
    degree = 4 * N
    ev = lambda p: ops.eval_offset(strip(p), degree, K1)
    a_e, b_e, c_e = ev(p_a), ev(p_b), ev(p_c)
    ql_e, qr_e, qm_e, qo_e, qc_e, pi_e = ev(ql), ev(qr), ev(qm), ev(qo), ev(qc), ev(p_pi)
    x_e, z_e, zw_e = ev([0, 1]), ev(p_z), ev(z_x_omega)
    s1_e, s2_e, s3_e, l1_e = ev(s1), ev(s2), ev(s3), ev(l1)
    zh_e = ev([neg1, 0, 0, 0, 1])
    c_eval = []
    for i in range(degree):
        a, b, c = a_e[i], b_e[i], c_e[i]
        cons = (a * b * qm_e[i] + a * ql_e[i] + b * qr_e[i] + c * qo_e[i] + qc_e[i] + pi_e[i]) % R
        f = (a + x_e[i] * BETA + GAMMA) * (b + x_e[i] * BETA * K1 + GAMMA) * (c + x_e[i] * BETA * k2 + GAMMA) % R
        g = (a + s1_e[i] * BETA + GAMMA) * (b + s2_e[i] * BETA + GAMMA) * (c + s3_e[i] * BETA + GAMMA) % R
        p1 = (g * zw_e[i] - f * z_e[i]) % R
        p2 = (z_e[i] - 1) * l1_e[i] % R
        p = ((p2 * ALPHA + p1) * ALPHA + cons) % R
        c_eval.append(p * pow(zh_e[i], -1, R) % R)
    t = strip(ops.interp_offset(c_eval, K1))
    t = t + [0] * (3 * (N + 2) - len(t))
    out["t_lo_1"] = ops.commit(strip(t[:N + 2]))
    out["t_mid_1"] = ops.commit(strip(t[N + 2:2 * (N + 2)]))
    out["t_hi_1"] = ops.commit(strip(t[2 * (N + 2):3 * (N + 2)]))
    return out
