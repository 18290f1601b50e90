"""Host-side mirror of the STARK prover's column commitment (provers/stark/src/prover.rs:229-244):
bit-reverse-permute the LDE columns, columns2rows, BatchedMerkleTree<BatchKeccak256Backend>::build — one device pipeline."""
import ctypes as C

import numpy as np

from . import _lib as L
from .errors import check


def commit_columns(field, columns, bit_reverse=True, return_nodes=False):
    """columns: (n_cols, N, 4) uint64 natural-order LDE columns.  Returns the 32-byte root (and the reference's `nodes`
    array, root first, when return_nodes)."""
    cols = np.ascontiguousarray(columns, dtype=np.uint64)
    n_cols, n = cols.shape[0], cols.shape[1]
    log2n = n.bit_length() - 1
    if n == 0 or (1 << log2n) != n:
        from .errors import InputError
        raise InputError(f"Input length is {n}, which is not a power of two")
    root = np.zeros(32, np.uint8)
    nodes = np.zeros((2 * n - 1, 32), np.uint8) if return_nodes else None
    check(L.lib().lw_stark_commit_columns(field.field, cols.ctypes.data_as(C.c_void_p), n_cols, log2n, 1 if bit_reverse else 0,
                                          root.ctypes.data_as(C.c_void_p),
                                          nodes.ctypes.data_as(C.c_void_p) if return_nodes else None))
    return (root.tobytes(), nodes) if return_nodes else root.tobytes()


def commit_columns_device(field, t_columns, n_cols, log2n, t_nodes, bit_reverse=True, stream=None):
    """Device-resident: t_columns holds n_cols dense columns of 2^log2n elements, t_nodes (2*2^log2n - 1) * 32 bytes."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    root = np.zeros(32, np.uint8)
    check(L.lib().lw_stark_commit_columns_device(field.field, C.c_void_p(t_columns.data_ptr()), n_cols, 0, log2n,
                                                 1 if bit_reverse else 0, C.c_void_p(t_nodes.data_ptr()),
                                                 root.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))
    return root.tobytes()


def commit_columns_layout_device(field, t_columns, n_cols, log2n, t_nodes, bit_reverse=True, stream=None):
    """commit_columns_device for any layout with an AsBytes in the reference (BabyBear u32 / u64-limb columns: the STARK LDE
    of BASELINE config 4)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    root = np.zeros(32, np.uint8)
    check(L.lib().lw_stark_commit_columns_layout_device(field.field, field.layout, C.c_void_p(t_columns.data_ptr()), n_cols, 0, log2n,
                                                        1 if bit_reverse else 0, C.c_void_p(t_nodes.data_ptr()),
                                                        root.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))
    return root.tobytes()


def fri_layer(field, coeffs, zeta, coset_offset, domain_size, return_nodes=False):
    """One layer of commit_phase (provers/stark/src/fri/mod.rs:44-58): returns (p' = 2*fold(p, zeta) coefficients,
    bit-reversed evaluation of p' on coset_offset * <w_domain>, Merkle root [, nodes])."""
    a = np.ascontiguousarray(coeffs, dtype=np.uint64).reshape(-1, 4)
    z = np.ascontiguousarray(zeta, dtype=np.uint64).reshape(4)
    off = np.ascontiguousarray(coset_offset, dtype=np.uint64).reshape(4)
    n_out = (a.shape[0] + 1) // 2
    poly = np.zeros((n_out, 4), np.uint64)
    ev = np.zeros((domain_size, 4), np.uint64)
    root = np.zeros(32, np.uint8)
    nodes = np.zeros((domain_size - 1, 32), np.uint8) if return_nodes else None
    plen = C.c_size_t(0)
    vp = lambda x: x.ctypes.data_as(C.c_void_p) if x is not None else None
    check(L.lib().lw_stark_fri_layer(field.field, vp(a), a.shape[0], vp(z), vp(off), domain_size, vp(poly), C.byref(plen), vp(ev),
                                     vp(root), vp(nodes)))
    out = (poly[:plen.value], ev, root.tobytes())
    return out + (nodes,) if return_nodes else out


def fri_layer_device(field, t_coeffs, n_coeffs, zeta, coset_offset, domain_size, want_evaluation=True, stream=None, buffers=None):
    """One layer of commit_phase with everything large resident in HBM: returns (t_poly, n_out, t_evaluation, t_nodes, root);
    t_poly is the zero-padded block holding p' = 2*fold(p, zeta) (the next layer's input), n_out = ceil(n_coeffs/2).
    buffers: (t_poly, t_evaluation_or_None, t_nodes) allocated by the caller (a prover allocates its layers once)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    z = np.ascontiguousarray(zeta, dtype=np.uint64).reshape(4)
    off = np.ascontiguousarray(coset_offset, dtype=np.uint64).reshape(4)
    n_out = (n_coeffs + 1) // 2
    blk = max(2, 1 << (n_out - 1).bit_length())
    if buffers is not None:
        t_poly, t_ev, t_nodes = buffers
        want_evaluation = t_ev is not None
    else:
        t_poly = torch.empty((blk, 4), dtype=torch.int64, device=t_coeffs.device)
        t_ev = torch.empty((domain_size, 4), dtype=torch.int64, device=t_coeffs.device) if want_evaluation else None
        t_nodes = torch.empty(((domain_size - 1) * 4,), dtype=torch.int64, device=t_coeffs.device)
    root = np.zeros(32, np.uint8)
    check(L.lib().lw_stark_fri_layer_device(field.field, C.c_void_p(t_coeffs.data_ptr()), n_coeffs, z.ctypes.data_as(C.c_void_p),
                                            off.ctypes.data_as(C.c_void_p), domain_size, C.c_void_p(t_poly.data_ptr()),
                                            C.c_void_p(t_ev.data_ptr()) if want_evaluation else None, C.c_void_p(t_nodes.data_ptr()),
                                            root.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))
    return t_poly, n_out, t_ev, t_nodes, root.tobytes()


def fri_fold_device(field, t_coeffs, n_coeffs, zeta, stream=None):
    """2 * fold_polynomial(p, zeta) alone (the last step of commit_phase, fri/mod.rs:61-63): returns (t_poly, n_out)."""
    import torch
    if stream is None:
        stream = torch.cuda.current_stream().cuda_stream
    z = np.ascontiguousarray(zeta, dtype=np.uint64).reshape(4)
    n_out = (n_coeffs + 1) // 2
    blk = max(2, 1 << (n_out - 1).bit_length())
    t_poly = torch.empty((blk, 4), dtype=torch.int64, device=t_coeffs.device)
    check(L.lib().lw_stark_fri_layer_device(field.field, C.c_void_p(t_coeffs.data_ptr()), n_coeffs, z.ctypes.data_as(C.c_void_p),
                                            None, 0, C.c_void_p(t_poly.data_ptr()), None, None, None, C.c_void_p(stream)))
    return t_poly, n_out


def fri_commit_phase_device(field, number_layers, t_p0, n_coeffs, sample_zeta, append_root, coset_offset_sq, domain_size,
                            want_evaluations=True):
    """commit_phase (provers/stark/src/fri/mod.rs:22-75) with the polynomial, every layer's evaluation and tree in HBM.
    The transcript stays with the caller: sample_zeta() -> FieldElement (4 x u64), append_root(bytes);
    coset_offset_sq(k) -> the k-times squared coset offset as a FieldElement (the caller owns field arithmetic on scalars).
    Returns (t_last_poly, layers) with layers = [(t_evaluation, t_nodes, root, domain_size)]."""
    import torch
    t_poly, n = t_p0, n_coeffs
    layers = []
    # every layer's buffers up front (sizes are known: the polynomial halves, the domain halves), as a prover would
    bufs, nn, dd = [], n_coeffs, domain_size
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(1, number_layers):
        nn, dd = (nn + 1) // 2, dd // 2
        blk = max(2, 1 << (nn - 1).bit_length())
        bufs.append((torch.empty((blk, 4), dtype=torch.int64, device=t_p0.device),
                     torch.empty((dd, 4), dtype=torch.int64, device=t_p0.device) if want_evaluations else None,
                     torch.empty(((dd - 1) * 4,), dtype=torch.int64, device=t_p0.device)))
    for k in range(1, number_layers):
        zeta = sample_zeta()
        domain_size //= 2
        t_poly, n, t_ev, t_nodes, root = fri_layer_device(field, t_poly, n, zeta, coset_offset_sq(k), domain_size,
                                                          want_evaluation=want_evaluations, stream=stream, buffers=bufs[k - 1])
        layers.append((t_ev, t_nodes, root, domain_size))
        append_root(root)
    t_last, _ = fri_fold_device(field, t_poly, n, sample_zeta())
    return t_last, layers
