/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.
 * Include-template: short-Weierstrass homogeneous-projective group law and Pippenger MSM,
 * exactly as the reference's CPU path (a = 0 curves keep the general-a formula's shape).
 *
 * Parameters (macros):
 *   EC_NAME(x)            name mangler
 *   FE_T                  base-field element type (struct of u64 words)
 *   FE_MUL(r,a,b) FE_ADD(r,a,b) FE_SUB(r,a,b) FE_NEG(r,a)  (pointers, may alias)
 *   FE_IS_ZERO(a) FE_EQ(a,b) FE_SET_ZERO(r) FE_SET_ONE(r) FE_INV(r,a)
 *   EC_A_IS_ZERO          1 if curve a == 0 (all curves in scope)
 *
 * Follows:
 *   math/src/elliptic_curve/short_weierstrass/point.rs:54-89    double
 *   math/src/elliptic_curve/short_weierstrass/point.rs:154-214  neutral_element / operate_with / neg
 *   math/src/elliptic_curve/point.rs:41-63                      to_affine / PartialEq
 *   math/src/cyclic_group.rs:17-29                              operate_with_self
 *   math/src/msm/pippenger.rs:18-103                            msm / optimum_window_size / msm_with
 *   math/src/msm/pippenger.rs:109-161                           parallel_msm_with
 *   math/src/msm/naive.rs:33-49                                 naive msm
 */

typedef struct { FE_T x, y, z; } EC_NAME(pt);

static void EC_NAME(neutral)(EC_NAME(pt) *r) { /* point.rs:156-162 */
    FE_SET_ZERO(&r->x); FE_SET_ONE(&r->y); FE_SET_ZERO(&r->z);
}
static int EC_NAME(is_neutral)(const EC_NAME(pt) *p) { return FE_IS_ZERO(&p->z); } /* :164-167 */

/* point.rs:54-89 */
static void EC_NAME(double)(EC_NAME(pt) *r, const EC_NAME(pt) *p) {
    if (EC_NAME(is_neutral)(p)) { *r = *p; return; }
    FE_T px = p->x, py = p->y, pz = p->z;
    FE_T px_square, three_px_square, w, w_square, s, s_square, s_cube, t, eight_s_cube;
    FE_T b, eight_b, four_b, h, hs, pys_square, eight_pys_square, xp, yp;
    FE_MUL(&px_square, &px, &px);
    FE_ADD(&three_px_square, &px_square, &px_square);
    FE_ADD(&three_px_square, &three_px_square, &px_square);
    /* w = a*pz*pz + 3px^2 ; a == 0 for every curve in scope, a*pz*pz == 0 exactly */
    w = three_px_square;
    FE_MUL(&w_square, &w, &w);
    FE_MUL(&s, &py, &pz);
    FE_MUL(&s_square, &s, &s);
    FE_MUL(&s_cube, &s, &s_square);
    FE_ADD(&t, &s_cube, &s_cube);        /* 2 s^3 */
    FE_ADD(&t, &t, &t);                  /* 4 s^3 */
    FE_ADD(&eight_s_cube, &t, &t);       /* 8 s^3 */
    FE_MUL(&b, &px, &py);
    FE_MUL(&b, &b, &s);
    FE_ADD(&t, &b, &b);                  /* 2b */
    FE_ADD(&four_b, &t, &t);             /* 4b */
    FE_ADD(&eight_b, &four_b, &four_b);  /* 8b */
    FE_SUB(&h, &w_square, &eight_b);
    FE_MUL(&hs, &h, &s);
    FE_MUL(&pys_square, &py, &py);
    FE_MUL(&pys_square, &pys_square, &s_square);
    FE_ADD(&t, &pys_square, &pys_square);
    FE_ADD(&t, &t, &t);
    FE_ADD(&eight_pys_square, &t, &t);
    FE_ADD(&xp, &hs, &hs);
    FE_SUB(&t, &four_b, &h);
    FE_MUL(&yp, &w, &t);
    FE_SUB(&yp, &yp, &eight_pys_square);
    r->x = xp; r->y = yp; r->z = eight_s_cube;
}

/* point.rs:171-207 */
static void EC_NAME(add)(EC_NAME(pt) *r, const EC_NAME(pt) *p, const EC_NAME(pt) *q) {
    if (EC_NAME(is_neutral)(q)) { *r = *p; return; }
    if (EC_NAME(is_neutral)(p)) { *r = *q; return; }
    FE_T px = p->x, py = p->y, pz = p->z, qx = q->x, qy = q->y, qz = q->z;
    FE_T u1, u2, v1, v2;
    FE_MUL(&u1, &qy, &pz);
    FE_MUL(&u2, &py, &qz);
    FE_MUL(&v1, &qx, &pz);
    FE_MUL(&v2, &px, &qz);
    if (FE_EQ(&v1, &v2)) {
        if (!FE_EQ(&u1, &u2) || FE_IS_ZERO(&py)) { EC_NAME(neutral)(r); return; }
        EC_NAME(pt) pc = *p;
        EC_NAME(double)(r, &pc);
        return;
    }
    FE_T u, v, w, u_square, v_square, v_cube, v_square_v2, a, t, xp, yp, zp;
    FE_SUB(&u, &u1, &u2);
    FE_SUB(&v, &v1, &v2);
    FE_MUL(&w, &pz, &qz);
    FE_MUL(&u_square, &u, &u);
    FE_MUL(&v_square, &v, &v);
    FE_MUL(&v_cube, &v, &v_square);
    FE_MUL(&v_square_v2, &v_square, &v2);
    FE_MUL(&a, &u_square, &w);
    FE_SUB(&a, &a, &v_cube);
    FE_ADD(&t, &v_square_v2, &v_square_v2);
    FE_SUB(&a, &a, &t);
    FE_MUL(&xp, &v, &a);
    FE_SUB(&t, &v_square_v2, &a);
    FE_MUL(&yp, &u, &t);
    FE_MUL(&t, &v_cube, &u2);
    FE_SUB(&yp, &yp, &t);
    FE_MUL(&zp, &v_cube, &w);
    r->x = xp; r->y = yp; r->z = zp;
}

static void EC_NAME(neg)(EC_NAME(pt) *r, const EC_NAME(pt) *p) { /* :210-213 */
    r->x = p->x; FE_NEG(&r->y, &p->y); r->z = p->z;
}

/* elliptic_curve/point.rs:41-54 */
static void EC_NAME(to_affine)(EC_NAME(pt) *r, const EC_NAME(pt) *p) {
    if (FE_IS_ZERO(&p->z)) { EC_NAME(neutral)(r); return; }
    FE_T inv_z;
    FE_INV(&inv_z, &p->z);
    FE_MUL(&r->x, &p->x, &inv_z);
    FE_MUL(&r->y, &p->y, &inv_z);
    FE_SET_ONE(&r->z);
}

/* elliptic_curve/point.rs:57-63 */
static int EC_NAME(eq)(const EC_NAME(pt) *p, const EC_NAME(pt) *q) {
    FE_T a, b, c, d;
    FE_MUL(&a, &p->x, &q->z);
    FE_MUL(&b, &p->z, &q->x);
    FE_MUL(&c, &p->y, &q->z);
    FE_MUL(&d, &q->y, &p->z);
    return FE_EQ(&a, &b) && FE_EQ(&c, &d);
}

/* cyclic_group.rs:17-29; exponent as MS-first limbs */
static void EC_NAME(mul)(EC_NAME(pt) *r, const EC_NAME(pt) *p, const u64 *k, int nl) {
    EC_NAME(pt) result, base = *p;
    EC_NAME(neutral)(&result);
    u64 e[8];
    for (int i = 0; i < nl; i++) e[i] = k[i];
    for (;;) {
        int zero = 1;
        for (int i = 0; i < nl; i++) if (e[i]) zero = 0;
        if (zero) break;
        if (e[nl - 1] & 1) EC_NAME(add)(&result, &result, &base);
        for (int i = nl - 1; i > 0; i--) e[i] = (e[i] >> 1) | (e[i - 1] << 63);
        e[0] >>= 1;
        EC_NAME(pt) b2 = base;
        EC_NAME(add)(&base, &b2, &b2);
    }
    *r = result;
}

/* (k >> shift).limbs[NUM_LIMBS-1]  — logical shift of an MS-first limb array, then the low limb */
static inline u64 EC_NAME(shr_low_limb)(const u64 *k, int nl, size_t shift) {
    size_t limb = shift / 64, bit = shift % 64;
    if (limb >= (size_t)nl) return 0;
    /* little index: position p (0 = least significant) lives at k[nl-1-p] */
    u64 lo = k[nl - 1 - limb] >> bit;
    if (bit && limb + 1 < (size_t)nl) lo |= k[nl - 2 - limb] << (64 - bit);
    return lo;
}

/* naive.rs:33-49 */
static void EC_NAME(msm_naive)(const u64 *cs, int nl, const EC_NAME(pt) *pts, size_t n, EC_NAME(pt) *out) {
    EC_NAME(pt) acc;
    EC_NAME(neutral)(&acc);
    for (size_t i = 0; i < n; i++) {
        EC_NAME(pt) t;
        EC_NAME(mul)(&t, &pts[i], cs + i * nl, nl);
        EC_NAME(add)(&acc, &acc, &t);
    }
    *out = acc;
}

/* one window of pippenger.rs:69-98: scatter + running-sum reduce; buckets must be neutral on entry
   and are neutral again on exit */
static void EC_NAME(window_sum)(const u64 *cs, int nl, const EC_NAME(pt) *pts, size_t n, size_t shift,
                                 size_t n_buckets, EC_NAME(pt) *buckets, EC_NAME(pt) *out) {
    for (size_t i = 0; i < n; i++) {
        u64 m = EC_NAME(shr_low_limb)(cs + i * nl, nl, shift) & (u64)n_buckets;
        if (m != 0) EC_NAME(add)(&buckets[m - 1], &buckets[m - 1], &pts[i]);
    }
    EC_NAME(pt) m_run, g;
    int have_g = 0;
    EC_NAME(neutral)(&m_run);
    for (size_t b = n_buckets; b-- > 0;) {
        EC_NAME(add)(&m_run, &m_run, &buckets[b]);
        EC_NAME(neutral)(&buckets[b]);
        if (!have_g) { g = m_run; have_g = 1; }
        else EC_NAME(add)(&g, &g, &m_run);
    }
    if (!have_g) EC_NAME(neutral)(&g);
    *out = g;
}

/* pippenger.rs:42-103 */
static int EC_NAME(msm_with)(const u64 *cs, int nl, const EC_NAME(pt) *pts, size_t n, size_t window_size,
                              EC_NAME(pt) *out) {
    if (window_size < 2) window_size = 2;
    if (window_size > 32) window_size = 32;
    size_t num_windows = (64 * (size_t)nl - 1) / window_size + 1;
    size_t n_buckets = ((size_t)1 << window_size) - 1;
    EC_NAME(pt) *buckets = (EC_NAME(pt) *)malloc(n_buckets * sizeof(EC_NAME(pt)));
    if (!buckets) return -1;
    for (size_t b = 0; b < n_buckets; b++) EC_NAME(neutral)(&buckets[b]);
    EC_NAME(pt) t;
    int have_t = 0;
    u64 pow2[1] = { 1ull << window_size };
    for (size_t w = num_windows; w-- > 0;) {
        EC_NAME(pt) g;
        EC_NAME(window_sum)(cs, nl, pts, n, w * window_size, n_buckets, buckets, &g);
        if (!have_t) { t = g; have_t = 1; }
        else {
            EC_NAME(pt) t2;
            EC_NAME(mul)(&t2, &t, pow2, 1);
            EC_NAME(add)(&t, &t2, &g);
        }
    }
    if (!have_t) EC_NAME(neutral)(&t);
    free(buckets);
    *out = t;
    return 0;
}

/* pippenger.rs:34-40 */
static size_t EC_NAME(optimum_window_size)(size_t data_length) {
    size_t lg = 0;
    if (data_length) lg = 63 - (size_t)__builtin_clzll((u64)data_length);
    return (lg * 4) / 5;
}

/* pippenger.rs:18-32 (length check is done by the caller who owns both lengths) */
static int EC_NAME(msm)(const u64 *cs, int nl, const EC_NAME(pt) *pts, size_t n, EC_NAME(pt) *out) {
    return EC_NAME(msm_with)(cs, nl, pts, n, EC_NAME(optimum_window_size)(n), out);
}

/* pippenger.rs:109-161 — one task per window, private buckets, shift by operate_with_self(1<<shift) */
typedef struct {
    const u64 *cs; int nl; const EC_NAME(pt) *pts; size_t n; size_t window_size;
    size_t num_windows; size_t n_buckets; EC_NAME(pt) *results; volatile long *next; int rc;
} EC_NAME(par_job);

static void *EC_NAME(par_worker)(void *arg) {
    EC_NAME(par_job) *j = (EC_NAME(par_job) *)arg;
    EC_NAME(pt) *buckets = (EC_NAME(pt) *)malloc(j->n_buckets * sizeof(EC_NAME(pt)));
    if (!buckets) { j->rc = -1; return NULL; }
    for (size_t b = 0; b < j->n_buckets; b++) EC_NAME(neutral)(&buckets[b]);
    for (;;) {
        long w = __sync_fetch_and_add(j->next, 1);
        if (w >= (long)j->num_windows) break;
        size_t shift = (size_t)w * j->window_size;
        EC_NAME(pt) g;
        EC_NAME(window_sum)(j->cs, j->nl, j->pts, j->n, shift, j->n_buckets, buckets, &g);
        /* window_item.operate_with_self(1 << shift) */
        u64 e[8];
        for (int i = 0; i < j->nl; i++) e[i] = 0;
        if (shift < 64 * (size_t)j->nl) e[j->nl - 1 - shift / 64] = 1ull << (shift % 64);
        EC_NAME(mul)(&j->results[w], &g, e, j->nl);
    }
    free(buckets);
    return NULL;
}

static int EC_NAME(parallel_msm_with)(const u64 *cs, int nl, const EC_NAME(pt) *pts, size_t n,
                                       size_t window_size, int threads, EC_NAME(pt) *out) {
    size_t num_windows = (64 * (size_t)nl - 1) / window_size + 1;
    size_t n_buckets = ((size_t)1 << window_size) - 1;
    EC_NAME(pt) *results = (EC_NAME(pt) *)malloc(num_windows * sizeof(EC_NAME(pt)));
    if (!results) return -1;
    volatile long next = 0;
    if (threads < 1) threads = 1;
    if ((size_t)threads > num_windows) threads = (int)num_windows;
    EC_NAME(par_job) job = { cs, nl, pts, n, window_size, num_windows, n_buckets, results, &next, 0 };
    pthread_t th[64];
    if (threads > 64) threads = 64;
    for (int t = 0; t < threads; t++) pthread_create(&th[t], NULL, EC_NAME(par_worker), &job);
    for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    EC_NAME(pt) acc;
    EC_NAME(neutral)(&acc);
    for (size_t w = 0; w < num_windows; w++) EC_NAME(add)(&acc, &acc, &results[w]);
    free(results);
    *out = acc;
    return job.rc;
}
