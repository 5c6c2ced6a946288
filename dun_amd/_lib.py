"""ctypes binding of libdun_hip.so (the C ABI declared in include/dun_hip.h).

There is no CPU fallback: if the library is missing, or a call is made without a GPU tensor, this raises.
torch is used only for device memory and the current HIP stream.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DUN_HIP_LIB") or os.path.join(_HERE, "libdun_hip.so")

DUN_OK = 0
OBJ_VI, OBJ_MLL = 0, 1
RES_OBJECTIVE, RES_E_LOGLIKE, RES_KL, RES_LOG_MARG, RES_DLOG_STD, RES_HEADER = 0, 1, 2, 3, 4, 8

EXPORTS = [
    "dun_abi_version", "dun_last_error", "dun_ll_num_partials", "dun_ll_gauss", "dun_ll_softmax", "dun_ll_reduce",
    "dun_depth_finalize", "dun_depth_finalize_partials", "dun_elbo_gauss_workspace_doubles", "dun_elbo_gauss",
    "dun_static_init", "dun_elbo_gauss_static_max_doubles", "dun_elbo_gauss_static", "dun_predict_gauss", "dun_predict_softmax", "dun_predict_num_partials", "dun_gemm_f32", "dun_gemm_colstat_tiles", "dun_gemm_precision",
    "dun_fc_workspace_floats", "dun_fc_obj_num_partials", "dun_fc_head_is_fused", "dun_fc_forward", "dun_fc_backward", "dun_fc_input_layer", "dun_fc_block_linear",
    "dun_fc_block_norm", "dun_fc_block_apply", "dun_fc_head", "dun_fc_head_depth", "dun_fc_bwd_begin", "dun_fc_bwd_block_reduce", "dun_fc_bwd_block_apply", "dun_fc_bwd_block_dx", "dun_fc_bwd_block_dw",
    "dun_fc_bwd_end", "dun_head_forward", "dun_head_backward_workspace_floats", "dun_head_backward", "dun_sgd_step",
    "dun_sgd_step_multi", "dun_sgd_step_multi_gated", "dun_label_count_bad", "dun_predict_gauss_step", "dun_fc_small_net_mode", "dun_fc_sync_bytes", "dun_fc_sync_status",
    "dun_fc_objective_in_forward", "dun_fc_head_grads_in_forward", "dun_copy_pair",
    "dun_copy_pair_indirect", "dun_predict_gauss_step_signal", "dun_gemm_profile_begin", "dun_gemm_profile_end",
]

SGD_MAX_SEGMENTS = 24


def res_floats(D):
    return RES_HEADER + 5 * D


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
                ("lda", C.c_int), ("ldb", C.c_int), ("ldc", C.c_int),
                ("a_kcontig", C.c_int), ("b_kcontig", C.c_int),
                ("bias", C.c_void_p), ("relu", C.c_int),
                ("R", C.c_void_p), ("ldr", C.c_int),
                ("U", C.c_void_p), ("V", C.c_void_p), ("O", C.c_int), ("ldv", C.c_int),
                ("colstats", C.c_void_p), ("splitk_ws", C.c_void_p), ("splitk_ws_floats", C.c_size_t),
                ("bnz", C.c_void_p), ("bn_mean", C.c_void_p), ("bn_rstd", C.c_void_p), ("bnstats", C.c_void_p),
                ("ldz", C.c_int),
                ("ev_mean", C.c_void_p), ("ev_scale", C.c_void_p), ("ev_shift", C.c_void_p), ("ev_head_w", C.c_void_p),
                ("ev_head_part", C.c_void_p)]


class GemmProfileRec(C.Structure):
    _fields_ = [(k, C.c_int) for k in ("M", "N", "K", "a_kcontig", "b_kcontig", "epilogue", "kernel", "ksplit")] + \
               [("ms", C.c_float)]


class FcTensors(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("w_in", "b_in", "w_out", "b_out", "w", "b", "gamma", "beta")]


class FcCtx(C.Structure):
    _fields_ = [("B", C.c_int), ("din", C.c_int), ("W", C.c_int), ("O", C.c_int), ("L", C.c_int),
                ("residual", C.c_int), ("train", C.c_int), ("depth", C.c_int),
                ("bn_eps", C.c_float), ("bn_momentum", C.c_float),
                ("p", FcTensors),
                ("running_mean", C.c_void_p), ("running_var", C.c_void_p),
                ("x", C.c_void_p), ("H", C.c_void_p), ("Z", C.c_void_p),
                ("bn_mean", C.c_void_p), ("bn_rstd", C.c_void_p), ("bn_sums", C.c_void_p),
                ("act", C.c_void_p), ("ws", C.c_void_p), ("ws_floats", C.c_size_t),
                ("dact", C.c_void_p), ("g", FcTensors), ("G", C.c_void_p), ("dU", C.c_void_p),
                ("obj_y", C.c_void_p), ("obj_w", C.c_void_p), ("obj_w_is_logits", C.c_int),
                ("obj_log_std", C.c_void_p), ("obj_inv_batch", C.c_float), ("obj_dact", C.c_void_p),
                ("obj_partials", C.c_void_p), ("obj_dwpart", C.c_void_p), ("obj_dbpart", C.c_void_p),
                ("head_grads_from_forward", C.c_int),
                ("num_batches_tracked", C.c_void_p), ("sync", C.c_void_p), ("sync_bytes", C.c_size_t),
                ("obj_fin_result", C.c_void_p), ("obj_fin_sums", C.c_void_p), ("obj_fin_q_logits", C.c_void_p),
                ("obj_fin_logprior", C.c_void_p), ("obj_fin_prior_ll", C.c_void_p),
                ("obj_fin_N_train", C.c_double), ("obj_fin_batch", C.c_double), ("obj_fin_beta", C.c_double),
                ("obj_fin_eps", C.c_double), ("obj_fin_objective", C.c_int), ("obj_fin_defer", C.c_int)]


class SgdSegment(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("buf", C.c_void_p), ("n", C.c_size_t)]


_lib = None


def _declare(lib):
    vp, i, f, d, sz = C.c_void_p, C.c_int, C.c_float, C.c_double, C.c_size_t
    sig = {
        "dun_abi_version": (i, []),
        "dun_last_error": (C.c_char_p, []),
        "dun_ll_num_partials": (i, [i, i]),
        "dun_predict_num_partials": (i, [i, i]),
        "dun_ll_gauss": (i, [vp, vp, vp, vp, i, i, i, f, vp, vp, vp, vp]),
        "dun_ll_softmax": (i, [vp, vp, vp, i, i, i, i, f, vp, vp, vp, vp]),
        "dun_ll_reduce": (i, [vp, i, i, vp, vp]),
        "dun_depth_finalize": (i, [vp, i, vp, vp, vp, vp, i, d, d, d, d, i, vp, vp]),
        "dun_depth_finalize_partials": (i, [vp, i, vp, i, vp, vp, vp, vp, i, d, d, d, d, i, vp, vp]),
        "dun_elbo_gauss_workspace_doubles": (sz, [i, i]),
        "dun_elbo_gauss": (i, [vp, vp, vp, vp, vp, vp, i, i, d, d, d, vp, vp, vp, sz, vp]),
        "dun_static_init": (i, []),
        "dun_elbo_gauss_static_max_doubles": (sz, []),
        "dun_elbo_gauss_static": (i, [vp, vp, vp, vp, vp, vp, i, i, d, d, d, vp, vp, vp]),
        "dun_predict_gauss": (i, [vp, vp, i, i, vp, vp, vp, vp, vp, vp, vp]),
        "dun_predict_softmax": (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp]),
        "dun_gemm_f32": (i, [C.POINTER(GemmDesc), vp]),
        "dun_gemm_precision": (i, [i]),
        "dun_gemm_colstat_tiles": (i, [i, i]),
        "dun_fc_workspace_floats": (sz, [i, i, i, i, i]),
        "dun_fc_obj_num_partials": (i, [i, i]),
        "dun_fc_head_is_fused": (i, [C.POINTER(FcCtx)]),
        "dun_fc_forward": (i, [C.POINTER(FcCtx), vp]),
        "dun_fc_backward": (i, [C.POINTER(FcCtx), vp]),
        "dun_fc_input_layer": (i, [C.POINTER(FcCtx), vp]),
        "dun_fc_block_linear": (i, [C.POINTER(FcCtx), i, vp]),
        "dun_fc_block_norm": (i, [C.POINTER(FcCtx), i, d, vp]),
        "dun_fc_block_apply": (i, [C.POINTER(FcCtx), i, vp]),
        "dun_fc_head": (i, [C.POINTER(FcCtx), vp]),
        "dun_fc_head_depth": (i, [C.POINTER(FcCtx), i, vp]),
        "dun_fc_bwd_begin": (i, [C.POINTER(FcCtx), vp]),
        "dun_fc_bwd_block_reduce": (i, [C.POINTER(FcCtx), i, vp]),
        "dun_fc_bwd_block_apply": (i, [C.POINTER(FcCtx), i, d, vp]),
        "dun_fc_bwd_block_dx": (i, [C.POINTER(FcCtx), i, d, vp]),
        "dun_fc_bwd_block_dw": (i, [C.POINTER(FcCtx), i, vp]),
        "dun_fc_bwd_end": (i, [C.POINTER(FcCtx), vp]),
        "dun_head_forward": (i, [vp, vp, vp, i, i, i, vp, vp]),
        "dun_head_backward_workspace_floats": (sz, [i, i, i]),
        "dun_head_backward": (i, [vp, vp, vp, i, i, i, vp, vp, vp, vp, sz, vp]),
        "dun_sgd_step": (i, [vp, vp, vp, sz, f, f, f, vp]),
        "dun_sgd_step_multi": (i, [C.POINTER(SgdSegment), i, f, f, f, vp]),
        "dun_sgd_step_multi_gated": (i, [C.POINTER(SgdSegment), i, f, f, f, vp, vp]),
        "dun_label_count_bad": (i, [vp, C.c_size_t, i, vp, vp]),
        "dun_predict_gauss_step": (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
        "dun_fc_small_net_mode": (i, [i]),
        "dun_fc_sync_bytes": (sz, [i, i]),
        "dun_fc_sync_status": (i, [C.POINTER(FcCtx), vp]),
        "dun_fc_objective_in_forward": (i, [C.POINTER(FcCtx)]),
        "dun_fc_head_grads_in_forward": (i, [C.POINTER(FcCtx)]),
        "dun_copy_pair": (i, [vp, vp, sz, vp, vp, sz, vp]),
        "dun_gemm_profile_begin": (i, [i]),
        "dun_gemm_profile_end": (i, [C.POINTER(GemmProfileRec), i]),
        "dun_copy_pair_indirect": (i, [vp, sz, vp, sz, vp, vp]),
        "dun_predict_gauss_step_signal": (i, [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def lib():
    """The loaded library; raises if it has not been built (python -m dun_amd.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "dun_amd: %s is missing. Build it with `python -m dun_amd.build` (hipcc, gfx950); "
                "there is no CPU fallback for the DUN kernels." % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        _declare(l)
        _lib = l
    return _lib


def check(rc, what=""):
    if rc != DUN_OK:
        msg = lib().dun_last_error()
        raise RuntimeError("dun_hip %s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def stream_ptr(device=None):
    if device is not None and torch.device(device).type != "cuda":
        raise RuntimeError("dun_amd kernels need GPU tensors (got device %s); there is no CPU path" % device)
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    """Device pointer of a tensor the kernels may dereference: contiguous, on a GPU."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("dun_amd kernels need GPU tensors (got a %s tensor); there is no CPU path" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("dun_amd kernels need contiguous tensors")
    return t.data_ptr()


def f32(t):
    if t.dtype != torch.float32:
        raise RuntimeError("dun_amd kernels compute in float32 (got %s)" % t.dtype)
    return ptr(t)


def out_ptr(t):
    """Pointer of an OUTPUT tensor a kernel writes a few scalars to: device memory, or pinned host memory (mapped into
    the device's address space at the same address: a captured step leaves its returned scalars there directly)."""
    if t.is_cuda:
        return ptr(t)
    if not t.is_pinned() or not t.is_contiguous():
        raise RuntimeError("dun_amd kernels write to GPU tensors or contiguous pinned host tensors")
    return t.data_ptr()

