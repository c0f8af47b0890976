"""ctypes binding of include/literalkg_hip.h.  No fallback: a missing library is an error."""
import ctypes as C
import os

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "liblkg_hip.so")
_lib = None

i64, i32, f32, vp, u64 = C.c_int64, C.c_int32, C.c_float, C.c_void_p, C.c_uint64

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/literalkg_hip.h
PROTOTYPES = {
    "lkg_version": [],
    "lkg_last_error": [],
    "lkg_preload": [],
    "lkg_csr_build": [i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lkg_csr_transpose": [i64, i64, i64, vp, vp, vp, vp, vp],
    "lkg_row_partition": [i64, vp, i32, vp],
    "lkg_csr_coo_indices_i64": [i64, i64, vp, vp, vp, vp],
    "lkg_csr_build_device_workspace": [i64, i64],
    "lkg_csr_build_device": [i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp],
    "lkg_csr_transpose_device_workspace": [i64, i64],
    "lkg_csr_transpose_device": [i64, i64, i64, vp, vp, vp, vp, vp, vp, i64, vp],
    "lkg_triples_count": [C.c_char_p, vp],
    "lkg_triples_read": [C.c_char_p, i64, vp, vp, vp, vp],
    "lkg_triples_dedup": [i64, vp, vp, vp, vp, vp],
    "lkg_laplacian_f32": [i64, i64, i64, vp, vp, vp, vp, i32, vp],
    "lkg_laplacian_device_f32": [i64, i64, i32, vp, vp, vp, vp, i32, vp, vp, vp],
    "lkg_spmm_csr_f32": [i64, i32, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i32, i32, vp],
    "lkg_spmm_csr_fused_f32": [i64, i32, vp, vp, vp, vp, i64, vp, i64, vp, i64, vp, i64, vp, vp, i64, vp, i64, vp, vp,
                               vp, vp, vp, i32, i32, vp, i64, vp, i64, vp],
    "lkg_csr_extract_rows": [i64, vp, vp, vp, vp, vp, vp, vp, vp],
    "lkg_spmm_csr_scatter_bwd_f32": [i64, i32, vp, vp, vp, vp, i64, vp, i64, vp],
    "lkg_edge_softmax_f32": [i64, i64, i32, vp, vp, vp, vp, vp, vp, vp, i32, i64, i64, vp, i64, vp, i64, vp, vp, vp,
                             i32, i32, i32, vp],
    "lkg_permute_f32": [i64, vp, i64, vp, vp, vp],
    "lkg_csr_check_i32": [i64, vp, i64, vp, i64, i64, vp, vp],
    "lkg_transe_score_fwd_f32": [i64, i32, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
    "lkg_loss_reduce_f32": [i64, vp, vp, f32, vp, vp],
    "lkg_transe_score_bwd_f32": [i64, i32, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, f32, vp, vp, i64, vp, i64, vp],
    "lkg_group_by_key_i64": [i64, i32, vp, vp, vp, vp, vp],
    "lkg_gather_rows_f32": [i64, i32, vp, i64, vp, vp, vp, i64, vp],
    "lkg_scatter_add_rows_f32": [i64, i32, vp, i64, vp, vp, vp, i64, vp],
    "lkg_fill_rows_f32": [i64, i32, vp, vp, i64, f32, vp, i32, vp],
    "lkg_gather_i64": [i64, vp, vp, vp, vp],
    "lkg_gather_rows_range_f32": [i64, i32, vp, i64, vp, i64, i64, vp, i64, vp],
    "lkg_scatter_add_rows_range_f32": [i64, i32, vp, i64, vp, i64, i64, vp, i64, vp],
    "lkg_sample_kg_batch": [i64, i32, u64, vp, i64, vp, vp, vp, vp, i64, i64, vp, vp, vp, vp, vp],
    "lkg_grouped_gemm_f32": [i32, i32, vp, i64, i32, i32, i64, i64, i64, f32, vp, i64, vp, i64, i64, i32, f32, vp, i64,
                             i64, vp],
    "lkg_dense_score_fwd_f32": [i64, i32, i32, vp, vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp],
    "lkg_dense_score_bwd_f32": [i64, i32, i32, vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, vp, vp, vp, vp, i64, vp, i64,
                                vp],
    "lkg_check_grouped_i64": [i64, i32, vp, vp, vp, vp, vp],
    "lkg_sanitize_ids_i64": [i64, vp, i64, i64, vp, vp, vp],
    "lkg_expand_groups_i32": [i64, i32, i32, vp, vp, vp, vp, vp],
    "lkg_act_layernorm_fwd_f32": [i64, i32, vp, i64, f32, vp, vp, f32, vp, i64, vp, i64, f32, vp, vp, f32, u64, vp],
    "lkg_act_layernorm_bwd_f32": [i64, i32, vp, i64, f32, vp, vp, vp, i64, vp, vp, vp, i64, vp, i64, f32, vp, i64, vp,
                                  vp, f32, u64, vp, vp, i32, vp, i64, vp],
    "lkg_dot_score_fwd_f32": [i64, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp],
    "lkg_dot_score_bwd_f32": [i64, i32, vp, i64, vp, vp, vp, vp, vp, f32, vp, vp, i64, vp],
    "lkg_relu_batchnorm_fwd_f32": [i64, i32, vp, i64, vp, vp, f32, i32, f32, vp, vp, vp, i64, vp, vp, vp],
    "lkg_relu_batchnorm_bwd_f32": [i64, i32, vp, i64, vp, vp, vp, i32, vp, i64, vp, i64, vp, vp, vp],
    "lkg_gate_blend_fwd_f32": [i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp],
    "lkg_gate_blend_bwd_f32": [i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, vp],
    "lkg_gate_blend_bwd_stats_f32": [i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64, i32, vp, vp,
                                     i64, i32, vp, i64, vp, vp],
    "lkg_row_absmax_f32": [i64, i32, vp, i64, vp, i32, vp],
    "lkg_gemm_tall_workspace": [i32, i32, vp, i32],
    "lkg_gemm_tall_f32": [i64, i32, i32, vp, vp, vp, vp, i32, vp, vp, i32, f32, f32, vp, i64, vp, i32, vp, i64, vp, i64,
                          vp, i64, vp, i64, vp],
    "lkg_linear_act_layernorm_workspace": [i32, i32, vp],
    "lkg_linear_act_layernorm_fwd_f32": [i64, i32, i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, f32, vp, i64, vp, i64, f32,
                                         vp, vp, f32, u64, vp, i64, vp],
    "lkg_gemm_workspace": [i32, i64, i64, i64],
    "lkg_gemm_f32": [i32, i32, i64, i64, i64, f32, vp, i64, vp, i64, f32, vp, i64, vp, vp, i64, vp],
    "lkg_gemm_f64acc_f32": [i32, i32, i64, i64, i64, vp, i64, vp, i64, vp, i64, vp],
    "lkg_colsum_f32": [i64, i32, vp, i64, vp, vp],
    "lkg_col_absmax_f32": [i64, i32, vp, i64, vp, vp],
    "lkg_gemm_longk_ok": [i64, i64, i64, vp, i64, vp, i64],
    "lkg_gemm_longk_f32": [i64, i64, i64, vp, i64, vp, i64, vp, i64, vp],
    "lkg_gemm_smallm_ok": [i64, i64, i64, vp, i64, vp, i64],
    "lkg_gemm_skinny_ok": [i64, i64, i64, vp, i64, vp, i64],
    "lkg_gemm_skinny_f32": [i64, i64, i64, vp, i64, vp, i64, i32, f32, vp, i64, vp, vp],
    "lkg_gemm_smallm_f32": [i64, i64, i64, vp, i64, vp, i64, vp, i64, vp],
    "lkg_gemm_wgrad_f32": [i64, i64, i64, vp, i64, vp, vp, i64, vp, vp, i64, vp],
    "lkg_narrow_layer_bwd_ok": [i64, i32, i32, vp, i64, vp, i64, vp, i64, vp, i64, vp, i64],
    "lkg_narrow_layer_bwd_workspace": [i64],
    "lkg_narrow_layer_bwd_f32": [i64, i32, i32, vp, i64, vp, i64, vp, i64, f32, vp, vp, i64, vp, vp, vp, i64, vp, i64, f32, f32,
                                 u64, vp, vp, i64, vp, vp, vp, vp, vp, i64, vp],
    "lkg_colsum_weighted_f32": [i64, i32, vp, i64, vp, i64, i32, vp, vp, i64, vp],
    "lkg_eltwise_f32": [i32, i64, i32, vp, i64, vp, i64, f32, f32, vp, i64, vp],
    "lkg_bi_mix_fwd_f32": [i64, i32, vp, i64, vp, i64, vp, i64, f32, vp, i64, vp, i64, vp],
    "lkg_bi_mix_bwd_f32": [i64, i32, vp, i64, vp, i64, vp, i64, vp, i64, i32, f32, vp, vp, vp, vp],
    "lkg_adam_step_f32": [i64, vp, vp, vp, vp, f32, f32, f32, f32, f32, i64, vp],
}
_RESTYPE = {"lkg_last_error": C.c_char_p, "lkg_csr_build_device_workspace": C.c_int64,
            "lkg_gemm_tall_workspace": C.c_int64, "lkg_gemm_workspace": C.c_int64,
            "lkg_linear_act_layernorm_workspace": C.c_int64, "lkg_narrow_layer_bwd_workspace": C.c_int64,
            "lkg_csr_transpose_device_workspace": C.c_int64}


class LkgError(RuntimeError):
    pass


def lib_path():
    return _LIB_PATH


def load():
    """Load the library once.  Raises (loudly) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        raise LkgError(
            f"{_LIB_PATH} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `python -m literalkg_amd.build`). "
            "literalkg_amd has no CPU fallback.")
    lib = C.CDLL(_LIB_PATH)
    for name, argtypes in PROTOTYPES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, C.c_int)
    _lib = lib
    return lib


_preloaded = False


def preload():
    """Load the library's code objects on the current device once per process (include/literalkg_hip.h: lkg_preload), as soon
    as torch has a HIP context -- so that no later call pays for it inside a timed region."""
    global _preloaded
    if _preloaded:
        return
    import torch
    if not torch.cuda.is_initialized():
        return                       # (host-only entry points -- structure build on the CPU, file parsing -- need no device code)
    _preloaded = True
    lib = load()
    if lib.lkg_preload() != 0:
        msg = lib.lkg_last_error()
        raise LkgError(f"lkg_preload failed: {msg.decode() if msg else '?'}")


def call(name, *args):
    lib = load()
    if not _preloaded:
        preload()
    rc = getattr(lib, name)(*args)
    if rc != 0:
        msg = lib.lkg_last_error()
        raise LkgError(f"{name} failed ({rc}): {msg.decode() if msg else '?'}")


_empty_stand_ins = {}


def ptr(t):
    """Raw pointer of a tensor / numpy array (None -> NULL).  An EMPTY tensor (a rank without rows, a structure without
    entries) has a null data pointer of its own; the entry points take NULL to mean "operand absent", so an empty operand is
    handed over as the address of a small per-device stand-in buffer instead -- never dereferenced, its extent being zero."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        p = t.data_ptr()
        if p == 0 and t.numel() == 0:
            key = str(t.device)
            if key not in _empty_stand_ins:
                import torch
                _empty_stand_ins[key] = torch.zeros(64, dtype=torch.float32, device=t.device)
            p = _empty_stand_ins[key].data_ptr()
        return p
    return t.ctypes.data
