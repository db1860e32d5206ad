"""ctypes binding of libchaorec_hip.so (include/chaorec_hip.h).

The HIP library IS the product's compute path: there is no CPU or eager-PyTorch fallback.
`load()` raises if the shared object is missing, and every op in chaorec_amd.ops raises on
non-CUDA tensors.
"""
import ctypes
import os
import shutil
import subprocess

_CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB_PATH = os.path.join(_CSRC, "libchaorec_hip.so")
SOURCES = ["spmm.hip", "bpr.hip", "score_topk.hip", "gemm.hip", "gemm_bf16x3.hip", "metrics.hip", "graph_dropout.hip",
           "rowops.hip", "feature_adam.hip", "exchange.hip"]
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off",
               "-mllvm", "-amdgpu-mfma-vgpr-form"]   # MFMA results in VGPRs: no v_accvgpr_read per compared score

_lib = None
ABI_VERSION = 16

c_i64p = ctypes.c_void_p
c_ptr = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/chaorec_hip.h one to one
SIGNATURES = {
    "chaorec_abi_version": (ctypes.c_int, []),
    "chaorec_last_error": (ctypes.c_char_p, []),
    "chaorec_spmm_csr_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_int32, ctypes.c_float, c_ptr, ctypes.c_float, c_ptr, c_ptr,
                                            ctypes.c_float, c_ptr, ctypes.c_int32, c_ptr]),
    "chaorec_spmm_csr_mean_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_int32, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_float, c_ptr,
                                                 ctypes.c_int32, c_ptr]),
    "chaorec_spmm_csr_adam_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64,
                                                 ctypes.c_int32, ctypes.c_float, c_ptr, ctypes.c_float, c_ptr,
                                                 ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_float,
                                                 ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                 ctypes.c_int32, c_ptr, ctypes.c_int64, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_spmm_csr_rowsparse_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64,
                                                      ctypes.c_int32, ctypes.c_float, c_ptr, ctypes.c_float, c_ptr,
                                                      ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int32, c_ptr]),
    "chaorec_expand_row_bits": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr, c_ptr,
                                               ctypes.c_int64, c_ptr]),
    "chaorec_zero_rows_by_bits_f32": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr, c_ptr]),
    "chaorec_or_words_u32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int64, c_ptr]),
    "chaorec_rows_list_from_bits": (ctypes.c_int, [c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_rows_mean_by_bits_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_float, c_ptr, ctypes.c_int64, ctypes.c_int32,
                                                     c_ptr, c_ptr]),
    "chaorec_spmm_csr_rowlist_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32,
                                                    ctypes.c_float, c_ptr, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr,
                                                    ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_float, c_ptr, c_ptr,
                                                    ctypes.c_int64, ctypes.c_int32, c_ptr]),
    "chaorec_batch_rows": (ctypes.c_int, [c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                          ctypes.c_uint64, ctypes.c_uint64, c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr,
                                          c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_bpr_fwd_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int64,
                                               ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, c_ptr, c_ptr, c_ptr,
                                               c_ptr, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                               c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                               ctypes.c_float, ctypes.c_float, c_ptr, c_ptr]),
    "chaorec_bpr_fwd_bwd_at_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int64,
                                                  ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, c_ptr, c_ptr, c_ptr,
                                                  c_ptr, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_float,
                                                  c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr,
                                                  c_ptr, c_ptr, ctypes.c_float, ctypes.c_float, c_ptr, c_ptr, ctypes.c_int64,
                                                  c_ptr]),
    "chaorec_bpr_finalize_steps_f32": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                      ctypes.c_int32, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr,
                                                      c_ptr, c_ptr]),
    "chaorec_bpr_finalize_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_ptr, c_ptr,
                                                c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_float, ctypes.c_float, c_ptr,
                                                c_ptr]),
    "chaorec_spmm_rows_per_wave": (ctypes.c_int, [ctypes.c_int32]),
    "chaorec_spmm_long_threshold": (ctypes.c_int, []),
    "chaorec_spmm_schedule_len": (ctypes.c_int64, [ctypes.c_int64, ctypes.c_int32]),
    "chaorec_spmm_build_schedule": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr,
                                                   ctypes.c_int64]),
    "chaorec_bpr_fwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int32,
                                           ctypes.c_int32, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_bpr_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int32,
                                           c_ptr, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_bpr_bwd_ordered_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int32,
                                                   c_ptr, ctypes.c_float, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_sample_negatives": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int32,
                                                ctypes.c_uint64, ctypes.c_uint64, c_ptr, ctypes.c_int64, c_ptr, c_ptr]),
    "chaorec_draw_batch": (ctypes.c_int, [c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int32, ctypes.c_int64,
                                          ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, c_ptr, c_ptr, c_ptr, c_ptr,
                                          ctypes.c_int64, c_ptr]),
    "chaorec_exchange_pull_sum_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, c_ptr, c_ptr]),
    "chaorec_exchange_pull_gather_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int64, c_ptr, c_ptr]),
    "chaorec_rows_copy_by_bits_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr, c_ptr]),
    "chaorec_exchange_pull_sum_rows_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                                          ctypes.c_int32, c_ptr, c_ptr, c_ptr]),
    "chaorec_exchange_pull_gather_rows_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                             c_ptr, c_ptr, c_ptr]),
    "chaorec_frontier_pack_f32": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_frontier_unpack_f32": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_shift_cat_i64": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr, c_ptr]),
    "chaorec_score_topk_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                            ctypes.c_int32]),
    "chaorec_score_topk_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                              c_ptr, c_ptr, ctypes.c_float, ctypes.c_int32, ctypes.c_int64,
                                              c_ptr, c_ptr, c_ptr, ctypes.c_size_t, ctypes.c_int32, c_ptr]),
    "chaorec_score_topk_hinted_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                     c_ptr, c_ptr, ctypes.c_float, ctypes.c_int32, ctypes.c_int64,
                                                     c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr, c_ptr,
                                                     ctypes.c_int32, ctypes.c_int32, c_ptr, c_ptr]),
    "chaorec_score_topk_stats": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                                c_ptr, c_ptr]),
    "chaorec_rank_metrics_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int32]),
    "chaorec_rank_metrics_f64": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int64, c_ptr, c_ptr, c_ptr, ctypes.c_int64,
                                                c_ptr, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    "chaorec_gemm_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "chaorec_gemm_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                        ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, c_ptr,
                                        ctypes.c_size_t, c_ptr]),
    "chaorec_gemm_nt_bf16x3_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "chaorec_gemm_nt_bf16x3": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, c_ptr,
                                              ctypes.c_size_t, c_ptr]),
    "chaorec_adam_step_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_float,
                                             ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                             ctypes.c_int32, c_ptr, c_ptr]),
    "chaorec_edge_dropout_norm": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_float,
                                                 ctypes.c_uint64, ctypes.c_uint64, c_ptr, ctypes.c_uint32, c_ptr, c_ptr,
                                                 c_ptr, c_ptr, c_ptr]),
    "chaorec_row_cosine_scale_fwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr]),
    "chaorec_row_cosine_scale_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64,
                                                        ctypes.c_int32, c_ptr]),
    "chaorec_bpr_fwd_drawn_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr, c_ptr, ctypes.c_int64,
                                                 ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint64, c_ptr, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_ptr, c_ptr, c_ptr,
                                                 c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_reduce_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64]),
    "chaorec_colsum_f32": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_ptr, c_ptr,
                                          ctypes.c_size_t, c_ptr]),
    "chaorec_sum_f32": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_float, c_ptr, c_ptr, ctypes.c_size_t, c_ptr]),
    "chaorec_weighted_sample_workspace_bytes": (ctypes.c_size_t, []),
    "chaorec_weighted_sample_keep": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_uint64,
                                                    ctypes.c_uint64, c_ptr, c_ptr, ctypes.c_size_t, c_ptr, c_ptr,
                                                    c_ptr]),
    "chaorec_weighted_sample_keys": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint64, c_ptr,
                                                    c_ptr, c_ptr]),
    "chaorec_leaky_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_float, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_mul_pair_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_rows_mean_f32": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_float, c_ptr, ctypes.c_int64, c_ptr]),
    "chaorec_adam_lowrank_strips": (ctypes.c_int32, [ctypes.c_int32]),
    "chaorec_adam_bias_table": (ctypes.c_int, [c_ptr, ctypes.c_int32, ctypes.c_float, ctypes.c_float, c_ptr]),
    "chaorec_adam_lowrank_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32,
                                                ctypes.c_int32, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                ctypes.c_float, ctypes.c_float, ctypes.c_int32, c_ptr, ctypes.c_int32,
                                                c_ptr, c_ptr, ctypes.c_int32, c_ptr, c_ptr, ctypes.c_int32,
                                                ctypes.c_int32, c_ptr]),
    "chaorec_gemm_tn_bf16x3_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "chaorec_gemm_tn_bf16x3": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, c_ptr, ctypes.c_size_t,
                                              c_ptr]),
    "chaorec_gemm_nn_bf16x3_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "chaorec_gemm_nn_bf16x3": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32, c_ptr,
                                              ctypes.c_size_t, c_ptr]),
    "chaorec_bpr_multi_fwd_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_bpr_multi_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_int32,
                                                 ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_bpr_multi_bwd_ordered_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int32, c_ptr, c_ptr, c_ptr, ctypes.c_int32,
                                                         ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
    "chaorec_gemm_nt_bf16x3_dual": (ctypes.c_int, [c_ptr] * 7 + [ctypes.c_int64] * 9 + [ctypes.c_int32, ctypes.c_int32, c_ptr]),
    "chaorec_gemm_nn_bf16x3_dual_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "chaorec_gemm_nn_bf16x3_dual": (ctypes.c_int, [c_ptr] * 5 + [ctypes.c_int64] * 9 + [c_ptr, ctypes.c_size_t, c_ptr]),
    "chaorec_gemm_tn_bf16x3_dual_workspace_bytes": (ctypes.c_size_t, [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]),
    "chaorec_gemm_tn_bf16x3_dual": (ctypes.c_int, [c_ptr] * 5 + [ctypes.c_int64] * 9 + [c_ptr, ctypes.c_size_t, c_ptr]),
    "chaorec_edge_dot_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32, c_ptr]),
    "chaorec_leaky_cat_add_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int32,
                                                 ctypes.c_int32, ctypes.c_float, c_ptr]),
    "chaorec_leaky_split_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_int64,
                                                   ctypes.c_int32, ctypes.c_int32, ctypes.c_float, c_ptr]),
    "chaorec_normalize_rows_fwd_f32": (ctypes.c_int, [c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64, ctypes.c_int32,
                                                      ctypes.c_float, c_ptr, c_ptr, c_ptr]),
    "chaorec_normalize_rows_bwd_f32": (ctypes.c_int, [c_ptr, c_ptr, c_ptr, ctypes.c_int64, ctypes.c_int64,
                                                      ctypes.c_int32, ctypes.c_float, c_ptr, c_ptr]),
    "chaorec_adam_multi_max": (ctypes.c_int32, []),
    "chaorec_adam_multi_f32": (ctypes.c_int, [ctypes.c_int32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, ctypes.c_float,
                                              ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                              ctypes.c_int32, c_ptr, c_ptr]),
    "chaorec_unique_rows": (ctypes.c_int, [c_ptr, ctypes.c_int64, ctypes.c_int64, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr]),
}


def _experiment_flags():
    """(extra, dropped) hipcc flags of an experiment build (tools/*_variants.py): CHAOREC_EXTRA_HIPCC_FLAGS /
    CHAOREC_DROP_HIPCC_FLAGS (e.g. "-mllvm -amdgpu-mfma-vgpr-form")."""
    return (os.environ.get("CHAOREC_EXTRA_HIPCC_FLAGS", "").split(), os.environ.get("CHAOREC_DROP_HIPCC_FLAGS", "").split())


def _experiment_tag():
    extra, drop = _experiment_flags()
    if not extra and not drop:
        return ""
    import hashlib
    return hashlib.sha1((" ".join(extra) + "|" + " ".join(drop)).encode()).hexdigest()[:10]


def current_lib_path():
    """The product library -- or, only while an experiment's flags are in the environment, that experiment's own file."""
    tag = _experiment_tag()
    return os.path.join(_CSRC, "exp", f"libchaorec_hip_{tag}.so") if tag else LIB_PATH


def build(force=False, verbose=False):
    """Cross-compile the HIP kernels for gfx950 into csrc/libchaorec_hip.so (no GPU needed): one object per source
    file (compiled in parallel, re-compiled only when the file or a header changed), then one link."""
    from concurrent.futures import ThreadPoolExecutor
    srcs = [os.path.join(_CSRC, s) for s in SOURCES]
    hdrs = [os.path.join(_CSRC, f) for f in sorted(os.listdir(_CSRC)) if f.endswith((".h", ".hpp"))]
    hdrs += [os.path.join(os.path.dirname(_CSRC), "..", "include", "chaorec_hip.h"), os.path.abspath(__file__)]
    extra, drop = _experiment_flags()
    lib_path = current_lib_path()
    if not force and os.path.exists(lib_path) and all(os.path.getmtime(lib_path) >= os.path.getmtime(d) for d in srcs + hdrs):
        return lib_path
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    # an experiment build (extra / dropped compiler flags) has its own objects and its own library: it can never
    # replace, or be mistaken for, the product build
    tag = _experiment_tag()
    objdir = os.path.join(_CSRC, "build" + (("_exp_" + tag) if tag else ""))
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(lib_path), exist_ok=True)
    cflags = [f for f in HIPCC_FLAGS if f != "-shared" and f not in drop] + extra

    def compile_one(src):
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        if (not force and os.path.exists(obj)
                and all(os.path.getmtime(obj) >= os.path.getmtime(d) for d in [src] + hdrs)):
            return obj
        cmd = [hipcc] + cflags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(srcs), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, srcs))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


def ensure_built():
    """Entry points that own a whole run (bench.py, smoke(), the test session) call this first: compile the library
    if the file is not there (a fresh checkout; hipcc cross-compiles without a GPU).  load() itself never builds and
    never falls back: a missing library is an error for every op."""
    if not os.path.exists(current_lib_path()):
        build(verbose=True)
    return current_lib_path()


def load():
    global _lib
    if _lib is not None:
        return _lib
    lib_path = current_lib_path()
    if not os.path.exists(lib_path):
        raise RuntimeError(
            f"{lib_path} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or chaorec_amd._lib.build()).  chaorec_amd has no CPU fallback.")
    # torch first: it brings its own HIP runtime (libamdhip64 of its ROCm build); a process in which THIS library pulled
    # in the system's copy before torch was imported ended with "no ROCm-capable device is detected" at the first launch
    # (python __graft_entry__.py smoke: build() loads the library, smoke() then imported torch)
    import torch  # noqa: F401
    lib = ctypes.CDLL(lib_path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here = header/library drift
        fn.restype = res
        fn.argtypes = args
    if lib.chaorec_abi_version() != ABI_VERSION:
        raise RuntimeError("libchaorec_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().chaorec_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
