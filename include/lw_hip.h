/*
 * lw_hip.h — C ABI of the MI355X (gfx950) NTT + MSM backend for lambdaworks.
 *
 * This is the drop-in boundary for the reference's data-parallel prover hot path.  Each entry point cites
 * the reference interface it replaces (paths relative to the lambdaworks tree, v0.11.0):
 *
 *   - lw_hip_ntt / lw_hip_ntt_device        <->  evaluate_fft_cuda / interpolate_fft_cuda
 *                                                (math/src/fft/gpu/cuda/polynomial.rs:16-49), i.e. the backend arm
 *                                                of Polynomial::evaluate_fft / interpolate_fft
 *                                                (math/src/fft/polynomial.rs:54-62,103-110)
 *   - lw_polynomial_evaluate_fft            <->  Polynomial::evaluate_fft / evaluate_offset_fft
 *                                                (math/src/fft/polynomial.rs:25-68,74-82)
 *   - lw_polynomial_interpolate_fft         <->  Polynomial::interpolate_fft / interpolate_offset_fft
 *                                                (math/src/fft/polynomial.rs:87-127)
 *   - lw_hip_gen_twiddles                   <->  gen_twiddles (math/src/fft/gpu/cuda/ops.rs:45-66), get_twiddles
 *                                                (math/src/fft/cpu/roots_of_unity.rs:66-75)
 *   - lw_hip_bitrev_permutation             <->  bitrev_permutation (math/src/fft/gpu/cuda/ops.rs:68-77),
 *                                                in_place_bit_reverse_permute (math/src/fft/cpu/bit_reversing.rs:2-9)
 *   - lw_hip_msm / lw_hip_msm_device        <->  msm::pippenger::msm (math/src/msm/pippenger.rs:18-32)
 *   - lw_hip_init / lw_hip_shutdown         <->  CudaState::new (math/src/fft/gpu/cuda/state.rs:29-38); the
 *                                                reference builds and drops device state on every call, this
 *                                                library keeps one context (twiddle caches, scratch, streams)
 *   - error codes                           <->  FFTError (math/src/fft/errors.rs:12-20), MSMError
 *                                                (math/src/msm/naive.rs:7-9), CudaError
 *                                                (gpu/src/cuda/abstractions/errors.rs:3-21)
 *
 * Data crosses the boundary bit-for-bit as the reference keeps it in memory (no conversion, as
 * math/src/gpu/cuda/field/element.rs:30-42): a field element is UnsignedInteger{limbs:[u64;N]} with
 * limbs[0] MOST significant, in Montgomery form; a projective point is X,Y,Z consecutive; an Fp2
 * coordinate is [c0,c1]; MSM scalars are canonical (non-Montgomery) UnsignedInteger<4>.
 *
 * All functions return 0 on success or a negative lw_status_t; lw_hip_last_error() gives a thread-local
 * message.  No exceptions or panics cross the ABI.  The caller owns every buffer; nothing is retained.
 * There is NO CPU fallback: without a usable gfx950 device every compute entry point fails with
 * LW_ERR_NO_DEVICE.
 */
#ifndef LW_HIP_H
#define LW_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    LW_FIELD_STARK252 = 0,      /* field_name() == "stark256" (stark_252_prime_field.rs:26-28) */
    LW_FIELD_BLS12_381_FR = 1,  /* bls12_381/default_types.rs:25-30 */
    LW_FIELD_BABYBEAR = 2       /* field_name() == "babybear31" (babybear.rs:33-35, babybear_u32.rs:21-23) */
} lw_field_t;

typedef enum {
    LW_LAYOUT_U64_LIMBS_MS_FIRST = 0, /* MontgomeryBackendPrimeField<_,4>: 4 x u64, R = 2^256 */
    LW_LAYOUT_BABYBEAR_U32_R32 = 1,   /* U32MontgomeryBackendPrimeField: one u32, R = 2^32 (babybear_u32.rs:6) */
    LW_LAYOUT_BABYBEAR_U64_R64 = 2,   /* MontgomeryBackendPrimeField<_,1>: one u64, R = 2^64 (babybear.rs:19-20) */
    LW_LAYOUT_EXT4_INTERLEAVED = 3    /* Degree4BabyBearExtensionField values: 4 x u64 (R = 2^64) per element,
                                         domain in the base field (quartic_babybear.rs:16-19,155-166) */
} lw_layout_t;

typedef enum { LW_DIR_FORWARD = 0, LW_DIR_INVERSE = 1 /* scaled by N^-1 */ } lw_dir_t;

typedef enum {
    LW_CURVE_BLS12_381_G1 = 0, /* 3 x 6 u64 per point */
    LW_CURVE_BN254_G1 = 1,     /* 3 x 4 u64 */
    LW_CURVE_BN254_G2 = 2,     /* 3 x 2 x 4 u64 */
    LW_CURVE_BLS12_381_G2 = 3  /* 3 x 2 x 6 u64 */
} lw_curve_t;

typedef enum {
    LW_OK = 0,
    LW_ERR_INPUT_NOT_POW2 = -1,  /* FFTError::InputError */
    LW_ERR_ORDER_TOO_LARGE = -2, /* FFTError::OrderError */
    LW_ERR_ROOT_OF_UNITY = -3,   /* FFTError::RootOfUnityError / FieldError::RootOfUnityError */
    LW_ERR_LENGTH_MISMATCH = -4, /* MSMError::LengthMismatch */
    LW_ERR_NO_DEVICE = -5,       /* CudaError::DeviceNotFound */
    LW_ERR_ALLOC = -6,           /* CudaError::AllocateMemory */
    LW_ERR_LAUNCH = -7,          /* CudaError::Launch / FunctionError */
    LW_ERR_COMM = -8,            /* RCCL unavailable / no communicator / collective failed (multi-GPU entry points) */
    LW_ERR_BAD_ARG = -9,
    LW_ERR_INV_ZERO = -10        /* FieldError::InvZeroError (zero coset offset) */
} lw_status_t;

typedef struct {
    double last_ntt_ms;      /* host wall time of the last lw_hip_ntt* call */
    double last_msm_ms;
    uint64_t ntt_calls, msm_calls;
    uint64_t twiddle_bytes;  /* device bytes held by twiddle caches */
    uint64_t scratch_bytes;
} lw_timings_t;

/* Per-kernel device timing (HIP events recorded on the launch stream around every kernel this library
 * launches between begin and end).  Instrumentation only; the reference's analogue is the `instruments`
 * feature's per-round timers (provers/stark/src/prover.rs:884-1049). */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
} lw_kernel_time_t;
typedef struct {
    int n;
    lw_kernel_time_t k[32];
} lw_profile_t;

/* ---- context ---- */
/* One context per process, bound to ONE device: NULL,0 -> the calling thread's current device; n_devices > 1 ->
 * LW_ERR_BAD_ARG (multi-GPU jobs run one process per GPU and shard through lw_hip_comm_init below).  Calling it again
 * with another device id releases every cached table / workspace of the old device first; destroy lw_srs_t handles
 * before doing that.  Every entry point binds the context's device for the calling thread for the duration of the call
 * (the HIP current device is per thread) and restores the caller's afterwards.
 *
 * Stream contract of the *_device entry points: work is enqueued on `hip_stream` and the call returns without waiting
 * for it, except where a result is handed back through a host pointer (the MSM's out_point_host, out_root): those
 * synchronise the stream before returning.
 *
 * Threads: the library keeps LW_LANES (4) independent sets of scratch, staging buffers, workspaces and side streams
 * ("lanes", csrc/context.h).  A call takes the first lane that is free, so calls from different host threads — the
 * reference's rayon loop over columns (provers/stark/src/trace.rs:186-190), an NTT caller beside an MSM caller — run
 * concurrently: one caller's download overlaps another's upload and kernels (host-buffer entry points run on their lane's
 * own stream).  More concurrent callers than lanes wait for a lane.  A single-threaded caller always gets lane 0 and sees
 * a one-context library; lanes allocate lazily, so memory grows only with the concurrency actually used.  Within a lane,
 * a call on a different stream than the lane's previous one first waits (hipStreamWaitEvent) for that call's work, so
 * calls may be issued from any stream or thread in any order.  The twiddle tables are shared by all lanes and rebuilt,
 * when a larger transform arrives, with every other call out of the library.  The multi-GPU entry points (one
 * communicator per process) always run on lane 0.  Callers that want column parallelism inside ONE call still pass
 * `batch`. */
int lw_hip_init(const int *device_ids, int n_devices);
void lw_hip_shutdown(void);
int lw_hip_device_count(void);
const char *lw_hip_last_error(void);
int lw_hip_get_timings(lw_timings_t *out);
int lw_hip_profile_begin(void);             /* start recording kernel events */
int lw_hip_profile_end(lw_profile_t *out);  /* synchronise, stop, report */
size_t lw_hip_field_elem_bytes(lw_field_t field, lw_layout_t layout);
size_t lw_hip_curve_point_bytes(lw_curve_t curve);

/* Result buffers for the host-buffer entry points.  Polynomial::evaluate_fft returns a NEW Vec on every call
 * (math/src/fft/polynomial.rs:37-38, fft/gpu/cuda/state.rs:61-68,197-202): a 512 MiB result that has never been touched costs
 * 131072 first-touch page faults when the device-to-host copy lands in it — several times the copy itself.  A caller that
 * can hold its result in a library buffer asks for one here: pinned, resident memory from a small pool (reused across
 * calls, released with lw_hip_result_release or at lw_hip_shutdown), into which lw_hip_ntt / lw_polynomial_evaluate_fft
 * copy at the PCIe rate.  Any other `out` pointer keeps working: large fresh buffers are populated (huge pages where the
 * host grants them) by helper threads while the upload and the kernels run, and downloaded chunk by chunk behind that. */
int lw_hip_result_acquire(size_t bytes, void **out_ptr);
int lw_hip_result_release(void *ptr);

/* ---- NTT backend seam ----
 * `in` holds `batch` transforms of 2^log2n elements, `batch_stride_elems` apart (0 -> dense).  Forward:
 * natural-order coefficients -> natural-order evaluations at w^i.  Inverse: evaluations -> coefficients,
 * already multiplied by N^-1.  coset_offset (one domain-field element in the same layout's base word, or
 * NULL): forward evaluates on offset*w^i, inverse divides the result by offset^i.  `in` may alias `out`.
 * log2n > TWO_ADICITY -> LW_ERR_ROOT_OF_UNITY; log2n > 63 -> LW_ERR_ORDER_TOO_LARGE. */
int lw_hip_ntt(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *in, void *out, uint32_t log2n,
               uint32_t batch, size_t batch_stride_elems, const void *coset_offset_or_null);

/* Same, on device-resident buffers; `hip_stream` is a hipStream_t (NULL = default stream); asynchronous. */
int lw_hip_ntt_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                      uint32_t log2n, uint32_t batch, size_t batch_stride_elems, const void *coset_offset_or_null,
                      void *hip_stream);

/* Low-degree extension on device buffers: the forward transform of 2^log2_coeffs coefficients (batch blocks, dense)
 * zero-padded to 2^log2n, i.e. Polynomial::evaluate_fft / evaluate_offset_fft with blowup_factor / domain_size
 * (math/src/fft/polynomial.rs:30-38) as the STARK prover calls it (provers/stark/src/prover.rs:150-167), without
 * materialising the padding: the log2n - log2_coeffs stages that would only replicate the block are skipped.
 * All fields and layouts (BASELINE config 4, the BabyBear STARK LDE, is this call with batch = 4); d_out must not alias
 * d_coeffs. */
int lw_hip_ntt_lde_device(lw_field_t field, lw_layout_t layout, const void *d_coeffs, uint32_t log2_coeffs, void *d_out,
                          uint32_t log2n, uint32_t batch, const void *coset_offset_or_null, void *hip_stream);

/* RootsConfig (math/src/field/traits.rs): 0 Natural, 1 NaturalInversed, 2 BitReverse, 3 BitReverseInversed.
 * Writes 2^order / 2 domain-field elements (host buffer, the layout's base word type).  order > 63 ->
 * LW_ERR_ORDER_TOO_LARGE; order > TWO_ADICITY -> LW_ERR_ROOT_OF_UNITY; order 0 -> nothing written. */
int lw_hip_gen_twiddles(lw_field_t field, lw_layout_t layout, uint64_t order, int config, void *out);
/* get_powers_of_primitive_root(order, count, config) and, with offset != NULL (config 0 only),
 * get_powers_of_primitive_root_coset(order, count, offset) (math/src/fft/cpu/roots_of_unity.rs:13-61): out[i] =
 * [offset *] w^(+-i), w the primitive 2^order-th root, as domain-field elements in the layout's base word (host
 * buffer).  The bit-reversed configurations return next_power_of_two(count) entries, bit-reverse permuted, as the
 * reference does; *out_len receives the number of entries (call with out == NULL to query it).  count 0 -> nothing.
 * order > TWO_ADICITY -> LW_ERR_ROOT_OF_UNITY. */
int lw_hip_gen_powers(lw_field_t field, lw_layout_t layout, uint64_t order, size_t count, int config, const void *offset_or_null,
                      void *out, size_t *out_len);
/* out[i] = in[bitrev(i)] over n = 2^k elements of the layout (host buffers, may alias). */
int lw_hip_bitrev_permutation(lw_field_t field, lw_layout_t layout, const void *in, void *out, size_t n);

/* Cross-shard step of the multi-GPU NTT (no reference counterpart: the reference has no multi-device path).
 * A 2^log2n_total vector is block-distributed over G = 2^log2_shards GPUs (M = N/G elements each).  After the
 * first all-to-all this rank holds x[j1][j2] for j1 = 0..G-1 and its slice j2 in [j2_begin, j2_begin+slice_len),
 * chunk j1 starting chunk_stride_elems*j1 elements into d_in.  Writes, with the same chunking over k1,
 *     y[k1][j2] = w_N^(j2*k1) * sum_j1 w_G^(j1*k1) * x[j1][j2]        (inverse: inverse roots, times G^-1).
 * The second all-to-all then gives rank k1 the row y[k1][0..M), whose local M-point NTT (lw_hip_ntt_device) is
 * X[k1 + G*k2].  See lambda_elliptic_curves_amd/distributed.py for the full exchange schedule. */
int lw_hip_ntt_cross_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in, void *d_out,
                            uint32_t log2n_total, uint32_t log2_shards, uint64_t j2_begin, uint64_t slice_len,
                            uint64_t chunk_stride_elems, uint32_t batch, uint64_t batch_stride_elems, void *hip_stream);

/* ---- Multi-GPU: one process per GPU, library-owned RCCL communicator over xGMI (no reference counterpart: the
 * reference has no multi-device path).  Rank 0 calls lw_hip_comm_unique_id and hands the 128 bytes to the other
 * processes out of band (exactly ncclGetUniqueId's contract); every process then calls lw_hip_comm_init (collective)
 * after lw_hip_init(&device, 1).  The communicator lives in the library context and is released by
 * lw_hip_comm_shutdown / lw_hip_shutdown.  nranks in {1, 2, 4, 8}.  Every failure of this path (RCCL not loadable,
 * no communicator, a failing collective) is LW_ERR_COMM. */
#define LW_HIP_COMM_ID_BYTES 128
int lw_hip_comm_unique_id(uint8_t *out_id /* LW_HIP_COMM_ID_BYTES */);
int lw_hip_comm_init(const uint8_t *unique_id, int rank, int nranks);
int lw_hip_comm_shutdown(void);
int lw_hip_comm_info(int *rank, int *nranks);

/* One transform of 2^log2n_total elements (or `batch` of them) block-distributed over the communicator's G ranks: this
 * rank holds elements [rank*M, (rank+1)*M) of the natural-order vector, M = 2^log2n_total / G; batch entries are M
 * elements apart (dense).  Collective: every rank calls it with the same arguments.  natural_output != 0: d_out_local
 * receives this rank's block of the natural-order result, i.e. the concatenation over ranks is byte-identical to
 * lw_hip_ntt_device on the concatenated input (Polynomial::evaluate_fft / interpolate_fft semantics);
 * natural_output == 0: the cyclic shard X[rank + G*k], k = 0..M-1 (one all-to-all fewer).  d_out_local may alias
 * d_in_local.  Schedule: all-to-all, cross-shard step (lw_hip_ntt_cross_device), all-to-all, local M-point NTT
 * [, all-to-all, interleave] — see csrc/comm.hip. */
int lw_hip_ntt_sharded_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_local, void *d_out_local,
                              uint32_t log2n_total, uint32_t batch, int natural_output, void *hip_stream);
/* The same schedule with G = 2^log2_shards virtual ranks walked on ONE device (exchanges are device-to-device
 * copies): d_in_full / d_out_full hold the whole vectors (batch entries 2^log2n_total apart), virtual rank g owning
 * block g.  With natural_output == 0 block g of d_out_full receives X[g + G*k].  Used to parity-test the exchange
 * schedule on a one-GPU box; needs no communicator. */
int lw_hip_ntt_sharded_selftest_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_full, void *d_out_full,
                                       uint32_t log2n_total, uint32_t log2_shards, uint32_t batch, int natural_output,
                                       void *hip_stream);
/* Self-test hook: the same run cut after step `stop_after` of the schedule (1 exchange A, 2 cross step, 3 exchange C,
 * 4 local NTT, 5 exchange E, 6 interleave): block g of d_out_full receives the batch x M elements virtual rank g holds at
 * that point.  tests/test_gpu_distributed.py checks the Python transliteration of the schedule
 * (lambda_elliptic_curves_amd/distributed.py, the one the world-size-2 gloo test drives) against it step by step. */
int lw_hip_ntt_sharded_selftest_steps_device(lw_field_t field, lw_layout_t layout, lw_dir_t dir, const void *d_in_full, void *d_out_full,
                                             uint32_t log2n_total, uint32_t log2_shards, uint32_t batch, int natural_output,
                                             int stop_after, void *hip_stream);
/* msm over pairs sharded across the ranks: every rank passes its n_local (scalar, point) pairs (n_local may differ per
 * rank, 0 allowed).  Schedule (csrc/comm.hip msm_sharded_run; the north star's "bucket all-reduce"): all ranks agree on the
 * window width from the largest shard; each accumulates its pairs into the full bucket array [W][2^(c-1)]; an all-to-all
 * hands rank g the bucket range g of every window from everyone (W x 2^(c-1) x point bytes per rank and MSM: 654 MB for
 * BN254 G1 at c = 20); rank g adds the G contributions and runs the running sums over its slice only, so the bucket reduce —
 * ~4.7 ms per MSM at c = 20 on one GPU, whatever N — costs 1/G per rank; an all-gather of 2 W points per rank carries the
 * per-slice sums and every rank folds the result.  Every rank receives the sum over all ranks' pairs, normalised like
 * lw_hip_msm. */
int lw_hip_msm_sharded_device(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n_local,
                              void *out_point_host, void *hip_stream);

/* The same run with G = 2^log2_shards virtual ranks walked on ONE device (virtual rank g owns the pairs [g n / G, (g+1) n / G));
 * parity-tests the exchange and the per-slice running sums on a one-GPU box, needs no communicator. */
int lw_hip_msm_sharded_selftest_device(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n_total,
                                       uint32_t log2_shards, void *out_point_host, void *hip_stream);

/* ---- Polynomial FFT API (host buffers, reference semantics) ----
 * evaluate: len = max(coeff_len, domain_size).next_power_of_two() * blowup_factor where coeff_len is
 * taken after stripping trailing zero coefficients (Polynomial::new); writes *out_len elements.  Call with
 * out == NULL to query *out_len.  Zero polynomial -> *out_len zeros.  Non-power-of-two len ->
 * LW_ERR_INPUT_NOT_POW2. */
int lw_polynomial_evaluate_fft(lw_field_t field, lw_layout_t layout, const void *coeffs, size_t n_coeffs,
                               size_t blowup_factor, size_t domain_size, const void *offset_or_null, void *out,
                               size_t out_capacity_elems, size_t *out_len);
/* interpolate: n must be a power of two; writes all n coefficients and reports in *coeff_len the length
 * after Polynomial::new would strip trailing zeros. */
int lw_polynomial_interpolate_fft(lw_field_t field, lw_layout_t layout, const void *evals, size_t n,
                                  const void *offset_or_null, void *out_coeffs, size_t *coeff_len);

/* ---- STARK commitment (SURVEY 8f "next" #1) ----
 * The commitment interpolate_and_commit_main makes right after the LDE (provers/stark/src/prover.rs:229-244): each of
 * the n_cols columns (2^log2n FieldElements, natural order, column-major) is bit-reverse permuted (bit_reverse != 0,
 * :232-234), rows are formed (columns2rows) and BatchedMerkleTree<BatchKeccak256Backend> is built
 * (crypto/src/merkle_tree/merkle.rs:31-56; field_element_vector.rs:41-58; utils.rs:44-72).  The permutation and the
 * transposition are folded into the leaf hash, nothing is copied.  nodes: (2*2^log2n - 1) x 32 bytes, root first,
 * leaves last — the reference's `nodes` vector.  n_cols = 1 is the FRI layer tree (Keccak256Backend). */
int lw_stark_commit_columns(lw_field_t field, const void *columns, uint32_t n_cols, uint32_t log2n, int bit_reverse,
                            uint8_t *out_root /* 32 bytes */, uint8_t *out_nodes_or_null);
int lw_stark_commit_columns_device(lw_field_t field, const void *d_columns, uint32_t n_cols, uint64_t col_stride_elems,
                                   uint32_t log2n, int bit_reverse, void *d_nodes, uint8_t *out_root_or_null, void *hip_stream);

/* The same commitment for any layout whose elements have an AsBytes in the reference — in particular the BabyBear columns
 * of BASELINE config 4 (the STARK LDE): U32MontgomeryBackendPrimeField hashes value().to_be_bytes(), 4 bytes per element
 * (u32_montgomery_backend_prime_field.rs:258-262), the u64-limb BabyBear its one limb big-endian
 * (montgomery_backed_prime_fields.rs:367-373).  LW_LAYOUT_EXT4_INTERLEAVED is rejected (no AsBytes in the reference). */
int lw_stark_commit_columns_layout_device(lw_field_t field, lw_layout_t layout, const void *d_columns, uint32_t n_cols,
                                          uint64_t col_stride_elems, uint32_t log2n, int bit_reverse, void *d_nodes,
                                          uint8_t *out_root_or_null, void *hip_stream);

/* One layer of the FRI commit phase (SURVEY 8f "next" #4), the loop body of commit_phase
 * (provers/stark/src/fri/mod.rs:44-58): p' = 2 * fold_polynomial(p, zeta) (fri/fri_functions.rs:7-30), then
 * new_fri_layer(p', coset_offset, domain_size) (fri/mod.rs:115-141): evaluation on the coset, bit-reverse permuted,
 * Merkle tree over pairs of consecutive evaluations.  The caller (transcript owner) passes the already squared offset
 * and halved domain.  out_poly: ceil(n/2) coefficients (+ stripped length); out_evaluation: domain_size elements
 * (bit-reversed order, as FriLayer stores it); out_nodes: (domain_size - 1) x 32 bytes, root first. */
int lw_stark_fri_layer(lw_field_t field, const void *coeffs, size_t n_coeffs, const void *zeta, const void *coset_offset,
                       size_t domain_size, void *out_poly, size_t *out_poly_len, void *out_evaluation, uint8_t *out_root,
                       uint8_t *out_nodes_or_null);

/* The same layer with every large object resident in HBM (commit_phase's loop, provers/stark/src/fri/mod.rs:44-58, keeps
 * current_poly and the layers; only the challenge and the 32-byte root go through the transcript): d_coeffs holds
 * n_coeffs coefficients; d_out_poly receives p' as a zero-padded block of max(2, next_power_of_two(ceil(n_coeffs/2)))
 * coefficients — the next layer's d_coeffs (pass ceil(n_coeffs/2) or the block length, trailing zeros change nothing);
 * d_out_evaluation_or_null: domain_size elements, bit-reversed order; d_nodes_or_null: (domain_size - 1) x 32 bytes, root
 * first; out_root_or_null: host, 32 bytes (synchronises the stream when given).  zeta and coset_offset are host
 * pointers.  d_nodes_or_null == NULL: fold only — the last step of commit_phase (mod.rs:61-63), whose constant
 * coefficient is the value sent to the verifier; domain_size and coset_offset are then ignored. */
int lw_stark_fri_layer_device(lw_field_t field, const void *d_coeffs, size_t n_coeffs, const void *zeta, const void *coset_offset,
                              size_t domain_size, void *d_out_poly, void *d_out_evaluation_or_null, void *d_nodes_or_null,
                              uint8_t *out_root_or_null, void *hip_stream);

/* ---- Groth16 quotient (SURVEY 8f "next" #3) ----
 * QuadraticArithmeticProgram::calculate_h_coefficients (provers/groth16/src/qap.rs:15-39) once the variable
 * polynomials L, R, O have been accumulated: n_coeffs <= num_gates BLS12-381 FrElements each, num_gates a power of two.
 * Writes 2*num_gates coefficients of h (and the stripped length, as Polynomial::new would leave it). */
int lw_groth16_h_coefficients(const void *l_coeffs, const void *r_coeffs, const void *o_coeffs, size_t n_coeffs, size_t num_gates,
                              void *out_h, size_t *coeff_len);

/* Device-resident form: d_l / d_r / d_o hold n_coeffs FrElements each, d_out_h receives 2*num_gates coefficients (trailing
 * zeros included) and stays in HBM for the MSM that consumes it — Prover::prove feeds h straight into
 * msm(h.representative(), z_powers_of_tau_g1[..h.len()]) (provers/groth16/src/prover.rs:68-72,97-101), which is
 * lw_hip_msm_srs_fr_device on d_out_h.  coeff_len_or_null (host): the stripped length; asking for it synchronises. */
int lw_groth16_h_coefficients_device(const void *d_l, const void *d_r, const void *d_o, size_t n_coeffs, size_t num_gates, void *d_out_h,
                                     size_t *coeff_len_or_null, void *hip_stream);

/* ---- MSM ----
 * scalars: n x 4 u64, canonical integers, MS limb first (callers pass .representative()).
 * points: n projective points, not necessarily normalised (Z != 1 allowed), identity = (0:1:0).
 * out_point: one projective point (same layout); only its affine image is canonical. */
int lw_hip_msm(lw_curve_t curve, const uint64_t *scalars, size_t n_scalars, const void *points, size_t n_points,
               void *out_point);
int lw_hip_msm_device(lw_curve_t curve, const uint64_t *d_scalars, const void *d_points, size_t n,
                      void *out_point_host, void *hip_stream);
/* Same, but the scalars are FrElements of the curve's scalar field as they sit in memory (Montgomery form): the
 * `.representative()` map every reference caller runs on the CPU first (provers/groth16/src/prover.rs:69-78,
 * crypto/src/commitments/kzg.rs:159-163) is done on the device (SURVEY 8f "next" #2). */
int lw_hip_msm_fr(lw_curve_t curve, const uint64_t *fr_elements, size_t n_scalars, const void *points, size_t n_points,
                  void *out_point);
int lw_hip_msm_fr_device(lw_curve_t curve, const uint64_t *d_fr_elements, const void *d_points, size_t n,
                         void *out_point_host, void *hip_stream);

/* Batched group law, IsGroup::operate_with (math/src/elliptic_curve/short_weierstrass/point.rs:171-207) on device
 * buffers: d_out[j*m + i] = d_rows[i] + d_cols[j] (projective points, reference layout; the sums are generally not
 * normalised, Z != 1).  bench.py builds its 2^24 distinct input points with it from two short runs. */
int lw_hip_ec_add_outer_device(lw_curve_t curve, const void *d_rows, size_t m, const void *d_cols, size_t k, void *d_out,
                               void *hip_stream);

/* Fixed point set cached on the device in affine form (SURVEY 8f "next" #2, second half).  Every reference caller
 * multiplies against a structured reference string it built once: KZG commits with
 * msm(&coefficients, &srs.powers_main_group[..coefficients.len()]) (crypto/src/commitments/kzg.rs:159-163), Groth16 with
 * the proving key's l_tau_g1 / r_tau_g1 / ... vectors (provers/groth16/src/prover.rs:69-85).  lw_hip_srs_create uploads
 * the projective points once, normalises them on the device (the to_affine of short_weierstrass/point.rs:91-129; the
 * identity is kept as a marked row) and keeps the affine rows resident; lw_hip_msm_srs then runs the same Pippenger
 * with mixed additions over the first n_scalars points.  Results are identical to lw_hip_msm on the same inputs.
 * n_scalars may be any length <= the SRS length (the KZG call shape); longer is LW_ERR_LENGTH_MISMATCH.
 * Memory: sets of 2^19 points and more keep 13 window-shifted affine copies (2^(20 w) P_i, w = 0..12) so that all windows
 * of an MSM share one bucket set — 13 x n x {128, 64, 128, 192} bytes for BLS12-381 G1 / BN254 G1 / BN254 G2 /
 * BLS12-381 G2 (27 GiB at 2^24 BLS12-381 G1 points), built once in lw_hip_srs_create (0.6 s at 2^24) and only while it
 * fits a quarter of the free device memory; LW_HIP_SRS_FOLD=0 keeps a single copy.  Results do not depend on it. */
typedef struct lw_srs lw_srs_t;
int lw_hip_srs_create(lw_curve_t curve, const void *points, size_t n_points, lw_srs_t **out_srs);
int lw_hip_srs_create_device(lw_curve_t curve, const void *d_points, size_t n_points, void *hip_stream, lw_srs_t **out_srs);
int lw_hip_srs_destroy(lw_srs_t *srs);
int lw_hip_msm_srs(const lw_srs_t *srs, const uint64_t *scalars, size_t n_scalars, void *out_point);
int lw_hip_msm_srs_device(const lw_srs_t *srs, const uint64_t *d_scalars, size_t n_scalars, void *out_point_host,
                          void *hip_stream);
/* scalars as stored FrElements (Montgomery form), see lw_hip_msm_fr */
int lw_hip_msm_srs_fr(const lw_srs_t *srs, const uint64_t *fr_elements, size_t n_scalars, void *out_point);
int lw_hip_msm_srs_fr_device(const lw_srs_t *srs, const uint64_t *d_fr_elements, size_t n_scalars, void *out_point_host,
                             void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif
