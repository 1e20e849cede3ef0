"""ctypes binding of libeasydgl_hip.so (C ABI declared in include/easydgl_hip.h).

The product path has NO fallback: importing this module without the built library raises, and
every op raises if handed a non-GPU tensor."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_long, c_uint32, c_void_p

# PyTorch-ROCm bundles its own libamdhip64; it must be in the process BEFORE our library is loaded so that both bind to
# ONE HIP runtime (same device context, streams and allocations).  Loading libeasydgl_hip.so first pulls in /opt/rocm's
# copy and every launch on torch memory then fails with "no ROCm-capable device is detected".
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
# (EDGL_LIB_PATH: development override — tools/ load instrumented builds of the same library through it)
LIB_PATH = os.environ.get("EDGL_LIB_PATH") or os.path.join(_HERE, "libeasydgl_hip.so")

F32, BF16 = 0, 1
EPI_BIAS, EPI_GELU, EPI_SAVE_PRE, EPI_MUL_DGELU, EPI_ACCUM, EPI_OUT_F32, EPI_RELU = 1, 2, 4, 8, 16, 32, 64
MAU_CAUSAL, MAU_NO_DIAG, MAU_DIAG_ZERO = 1, 2, 4
MAU_NO_SKIP = 8      # host-side hint (include/easydgl_hip.h): the unskipped BiMAU kernels
TATTN_CAUSAL = 1

P, I, F, L, U32, I64 = c_void_p, c_int, c_float, c_long, c_uint32, c_int64

# name -> (restype, argtypes) — must list EVERY symbol include/easydgl_hip.h declares
SIGNATURES = {
    "edgl_last_error": (c_char_p, []),
    "edgl_version": (I, []),
    "edgl_profile_next": (I, [I, P, P]),
    "edgl_rng_advance": (I, [P, P]),
    "edgl_mask_random": (I, [P, I, I, I, I64, P, U32, P, P, P, P]),
    "edgl_mask_last": (I, [P, I, I, I64, P, P]),
    "edgl_encode_fwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I64, F, F, P, U32, P, P, P, I, P]),
    "edgl_encode_bwd_workspace": (L, [I, I, I]),
    "edgl_encode_bwd": (I, [P, P, P, I, I, I, I, I, F, P, U32, P, P, P, P, I, P]),
    "edgl_encode_bwd_add": (I, [P, P, P, P, P, I, I, I, I, I, F, P, U32, P, P, P, P, I, P]),
    "edgl_encode_fwd_ct": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I64, F, F, P, U32, P, P, P, I, I, I, P]),
    "edgl_encode_fwd_prep": (I, [P, P, P, P, P, P, P, I, I, I, I, I, I64, F, F, P, U32, P, P, P, I, I, P, I, P, P, P, P, P, P, I, P]),
    "edgl_encode_bwd_add_ct": (I, [P, P, P, P, P, I, I, I, I, I, F, P, U32, P, P, P, P, I, I, P]),
    "edgl_encode_bwd_label_fused": (I, [I, I]),
    "edgl_encode_bwd_add_label": (I, [P, P, P, P, P, I, I, I, I, I, F, P, U32, P, P, P, P, I, P, P, P, P, I, P, I, P]),
    "edgl_embed_pos_fwd": (I, [P, P, P, P, P, I, I, I, I, F, F, P, U32, P, P, P, I, P]),
    "edgl_embed_pos_bwd": (I, [P, P, I, I, I, I, F, P, U32, P, P, I, P]),
    "edgl_embed_pos_fwd_ct": (I, [P, P, P, P, P, I, I, I, I, F, F, P, U32, P, P, P, I, I, P]),
    "edgl_embed_pos_bwd_ct": (I, [P, P, I, I, I, I, F, P, U32, P, P, I, I, P]),
    "edgl_gemm": (I, [P, P, P, I, I, I, I, I, I, I, I, P, P, I, I, P, I, P]),
    "edgl_colsum": (I, [P, I, I, I, P, I, P, I, I, P]),
    "edgl_gemm_dw_workspace": (L, [I, I, I, I]),
    "edgl_gemm_dw": (I, [P, P, P, P, I, I, I, I, I, I, P, I, P]),
    "edgl_gemm_dw_defer": (I, [I, P]),
    "edgl_bimau_pack_bytes": (L, [I, I, I, I]),
    "edgl_bimau_pack": (I, [P, P, P, P, I, I, I, P, I, P]),
    "edgl_bimau_saved_bytes": (L, [I, I, I, I, I]),
    "edgl_bimau_mark_group": (I, [I, I, I]),
    "edgl_bimau_fwd": (I, [P, P, I, P, P, P, P, I, I, I, I, I, F, P, U32, P, P, P, I, I, P]),
    "edgl_bimau_fwd_zr": (I, [P, P, I, P, P, P, P, I, I, I, I, I, F, P, U32, P, P, P, P, I, I, P]),
    "edgl_bimau_dropbits_bytes": (L, [I, I, I]),
    "edgl_bimau_dropbits": (I, [I, I, I, F, P, U32, P, P]),
    "edgl_bimau_fwd_db": (I, [P, P, I, P, P, P, P, I, I, I, I, I, F, P, U32, P, F, P, P, P, P, I, I, P]),
    "edgl_bimau_bwd_db": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, U32, P, F, P, P, P, P, P, P, I, I, P]),
    "edgl_bimau_job_order": (I, [P, I, I, P, P]),
    "edgl_bimau_fwd_ord": (I, [P, P, I, P, P, P, P, I, I, I, I, I, F, P, U32, P, F, P, P, P, P, P, I, I, P]),
    "edgl_bimau_bwd_ord": (I, [P, P, P, P, P, P, P, P, I, P, F, P, P, P, I, I, I, I, I, F, P, U32, P, F, P, P, P, P, P, P, P, I, I, P]),
    "edgl_bimau_bwd_workspace": (L, [I, I, I, I, I, I]),
    "edgl_bimau_bwd": (I, [P, P, P, P, P, P, P, P, P, I, I, I, I, I, F, P, U32, P, P, P, P, P, P, I, I, P]),
    "edgl_add_layernorm_fwd": (I, [P, P, I, P, P, I, I, I, F, P, U32, P, I, P, P, I, P]),
    "edgl_add_layernorm_bwd": (I, [P, P, I, P, P, P, I, I, I, F, P, U32, P, I, P, P, P, P, P, P, I, P]),
    "edgl_add_layernorm_bwd_act": (I, [P, P, I, P, P, P, I, I, I, F, P, U32, P, I, P, P, P, P, P, P, P, I, P]),
    "edgl_add_layernorm_fwd_ct": (I, [P, P, I, P, P, I, I, I, F, P, U32, P, I, P, P, I, I, I, P]),
    "edgl_add_layernorm_bwd_act_ct": (I, [P, P, I, P, P, P, I, I, I, F, P, U32, P, I, P, P, P, P, P, P, P, I, I, I, P]),
    "edgl_score_chunks": (I, [I, I]),
    "edgl_compact_rows": (I, [P, P, I, I, P, P, P, P, P, I, P]),
    "edgl_compact_scan": (I, [P, I, P, P, P, P]),
    "edgl_compact_scan_labels": (I, [P, I, P, P, P, P, P]),
    "edgl_compact_gather": (I, [P, P, P, I, I, P, P, I, P]),
    "edgl_scatter_rows": (I, [P, P, I, I, P, I, P]),
    "edgl_score_lse_fwd": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P, P, I, P]),
    "edgl_ce_loss_fwd": (I, [P, P, P, I, P, P, P]),
    "edgl_ce_loss_fwd_add": (I, [P, P, P, I, P, P, P, P, P]),
    "edgl_ce_loss_fwd_add_w": (I, [P, P, P, I, P, P, P, P, P, P]),
    "edgl_score_bwd_workspace": (L, [I, I, I, I, I]),
    "edgl_score_ce_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, I, P]),
    "edgl_score_flash_workspace": (L, [I, I, I, I, I]),
    "edgl_score_flash_fwd": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P, I, P]),
    "edgl_score_prepare_table": (I, [P, I, I, I, I, I, P, I, P]),
    "edgl_score_flash_fwd_pre": (I, [P, P, P, P, I, I, I, I, I, P, P, P, P, I, I, P]),
    "edgl_score_flash_fwd_coef": (I, [P, P, P, P, I, I, I, P, P, P, P, P, I, P]),
    "edgl_score_flash_fwd_coef_w": (I, [P, P, P, P, I, I, I, P, P, P, P, P, P, I, P]),
    "edgl_score_flash_fwd_rows_w": (I, [P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, I, P]),
    "edgl_score_ce_nparts": (I, [I, I]),
    "edgl_score_flash_fwd_rows_wp": (I, [P, P, P, P, I, I, I, P, P, P, P, P, P, P, P, P, I, P]),
    "edgl_ce_loss_parts": (I, [P, I, P, P, P, P, P]),
    "edgl_score_flash_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, I, P]),
    "edgl_score_flash_bwd_ex": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, P, P, P, P, I, I, P]),
    "edgl_score_flash_label_term": (I, [P, P, P, P, I, I, I, I, I, P, P, P, I, P]),
    "edgl_reduce_defer": (I, [I, P]),
    "edgl_reduce_flush": (I, [P]),
    "edgl_mask_topk": (I, [P, I, I, I, P, I, I, P, P, P]),
    "edgl_score_topk_fused_supported": (I, [I, I, I, I, I, I]),
    "edgl_score_topk_fused_workspace": (L, [I, I, I, I, I]),
    "edgl_score_topk_fused": (I, [P, P, P, P, I, I, I, I, I, I, I, P, P, P, I, P]),
    "edgl_topk_merge": (I, [P, P, I, I, I, P, P, P]),
    "edgl_rank_metrics": (I, [P, I, I, P, P, P]),
    "edgl_tpp_workspace": (I, []),
    "edgl_tpp_fwd": (I, [P, P, P, P, P, I, I, I, I, I, F, P, P, I, P]),
    "edgl_tpp_bwd": (I, [P, P, P, P, P, I, I, I, I, I, F, P, P, P, P]),
    "edgl_tpp_fwd_bwd": (I, [P, P, P, P, P, I, I, I, I, I, F, P, P, I, P, P]),
    "edgl_tpp_norm": (I, [P, P, I, I, I, P, P]),
    "edgl_tpp_fwd_bwd_ex": (I, [P, P, P, P, P, I, I, I, I, I, F, P, P, I, P, I, P]),
    "edgl_tpp_rows_workspace": (L, [I, I, I]),
    "edgl_tpp_fwd_bwd_rows": (I, [P, P, P, P, P, I, I, I, I, I, F, P, P, I, P, P]),
    "edgl_tpp_prep_bytes": (L, [I, I, I]),
    "edgl_tpp_prep": (I, [P, P, P, P, I, I, I, I, P, P]),
    "edgl_bimau_bwd_tpp": (I, [P, P, P, P, P, P, P, I, P, F, P, P, P, I, I, I, I, I, F, P, U32, P, F, P, P, P, P, P, P, I, I, P]),
    "edgl_tpp_finish_parts": (I, [P, I, F, I, P, I, I, I, P, P, I, P]),
    "edgl_tpp_finish_parts_n": (I, [P, I, F, I, P, P, I, P]),
    "edgl_dp_counts": (I, [P, P, I, I, I, P, P]),
    "edgl_adam_step": (I, [P, P, P, P, L, F, F, F, F, P, F, P, I, P, P]),
    "edgl_step_begin": (I, [P, P, F, F, F, P]),
    "edgl_adam_apply": (I, [P, P, P, P, L, F, F, F, P, F, P, I, P, P]),
    "edgl_adam_l2_parts": (I, [L]),
    "edgl_adam_apply_l2p": (I, [P, P, P, P, L, F, F, F, P, F, P, I, P, P, P]),
    "edgl_adam_next_tickets": (I, [L]),
    "edgl_adam_apply_l2p_next": (I, [P, P, P, P, L, F, F, F, P, F, P, I, P, P, P, F, P, P]),
    "edgl_l2_from_parts": (I, [P, I, F, P, I, P]),
    "edgl_l2_loss": (I, [P, P, I, F, P, I, P, P]),
    "edgl_cast": (I, [P, P, L, I, P]),
    "edgl_cast_back": (I, [P, P, L, I, I, P]),
    "edgl_add": (I, [P, P, P, L, I, P]),
    "edgl_add_cols": (I, [P, I, P, P, I, L, I, I, P]),
    "edgl_dropout": (I, [P, P, L, F, P, U32, I, P]),
    "edgl_ff_tail": (I, [P, P, P, L, I, F, P, U32, P, I, P]),
    "edgl_relu_bwd": (I, [P, P, P, L, I, P]),
    "edgl_gelu_bwd": (I, [P, P, P, L, I, P]),
    "edgl_tattn_saved_bytes": (L, [I, I, I, I]),
    "edgl_tattn_fwd": (I, [P, I, P, I, P, I, P, I, P, I, I, I, I, I, F, F, P, U32, P, I, P, I, I, P]),
    "edgl_tattn_bwd": (I, [P, I, P, I, P, I, P, P, I, P, I, I, I, I, I, F, F, P, U32, P, I, P, I, P, I, I, I, P]),
    "edgl_timefn_fwd": (I, [P, I, P, I, P, P, P, P, P, I, I, I, I, F, P, P, P, I, P]),
    "edgl_timefn_bwd_workspace": (L, [I]),
    "edgl_timefn_bwd": (I, [P, I, P, P, P, P, P, I, I, I, I, F, P, I, P, I, P, P, P, I, P]),
    "edgl_mask_rows": (I, [P, P, P, L, I, I, P]),
    "edgl_tiattn_bucket_elems": (L, [I, I, I, I]),
    "edgl_tiattn_fwd": (I, [P, I, P, I, P, I, P, I, P, P, P, P, I, I, I, I, I, F, F, I, F, P, U32, P, I, P, P, I, I, P]),
    "edgl_tiattn_bwd": (I, [P, I, P, I, P, I, P, P, P, P, I, P, I, P, P, I, I, I, I, F, F, I, F, P, U32, P, I, P, I, P, I, P, P, P,
                            I, I, P]),
    "edgl_add_pos2": (I, [P, P, P, I, I, I, P, I, P]),
    "edgl_tail_pack_elems": (L, [I]),
    "edgl_tail_supported": (I, [I, I, I]),
    "edgl_tail_variant": (I, [I]),
    "edgl_adam_apply_ex": (I, [P, P, P, P, L, F, F, F, P, F, P, I, P, P, P, L, L, L, L, P, L, L, L, I, P, P, P, F, P]),
    "edgl_score_flash_slab_info": (I, [I, I, I, I, I, P]),
    "edgl_tail_pack": (I, [P, P, P, P, I, P, P]),
    "edgl_tail_fwd": (I, [P, P, I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, U32, U32, P, I, I, P, P, P, P, P, P, P, P, P, P, P,
                          P, P, I, P]),
    "edgl_tail_fwd_ct": (I, [P, P, I, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, U32, U32, P, I, I, P, P, P, P, P, P, P, P, P, P, P,
                             P, P, I, I, I, P]),
    "edgl_tail_bwd_ct": (I, [P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, U32, U32, I, P, P, I, P, P, P, P, P, P, P, P,
                             P, P, P, P, P, P, P, I, I, I, P]),
    "edgl_tail_bwd_workspace": (L, [I, I]),
    "edgl_tail_bwd": (I, [P, I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, I, I, F, P, U32, U32, I, P, P, I, P, P, P, P, P, P, P, P,
                          P, P, P, P, P, P, P, I, P]),
    "edgl_embedding_fwd": (I, [P, L, P, I, I, I, F, P, I, P]),
    "edgl_embedding_bwd": (I, [P, L, P, I, I, I, F, P, I, P]),
    "edgl_time_sinusoid": (I, [P, L, P, I, P, I, P]),
    "edgl_time_function_fwd": (I, [P, L, P, P, I, P, I, P]),
    "edgl_time_function_bwd_workspace": (L, [I]),
    "edgl_time_function_bwd": (I, [P, L, P, P, I, P, P, P, P, I, P]),
}


class EdglError(RuntimeError):
    pass


def _load() -> ctypes.CDLL:
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m easydgl_amd.build` (hipcc --offload-arch=gfx950). "
            "easydgl_amd has no CPU / PyTorch fallback by design.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError here == ABI drift: fail loudly
        fn.restype = res
        fn.argtypes = args
    return lib


_cdll = _load()


class _Profiler:
    """Optional per-call HIP-event bracketing (torch events on the current stream — the stream every kernel
    of this library is launched on).  Disabled -> zero overhead beyond one attribute test."""

    def __init__(self):
        self.enabled = False
        self.only = None          # set of C function names to bracket (None = all)
        self.records = {}         # name -> list[(start_event, end_event)]

    def start(self, only=None):
        self.enabled, self.only, self.records = True, (set(only) if only else None), {}

    def stop(self):
        self.enabled = False

    def summary(self):
        """name -> (calls, total_ms); call after torch.cuda.synchronize()."""
        out = {}
        for name, evs in self.records.items():
            out[name] = (len(evs), sum(a.elapsed_time(b) for a, b in evs))
        return out


profiler = _Profiler()


class _LibProxy:
    def __init__(self, cdll):
        self._cdll = cdll
        self._wrapped = {}

    def __getattr__(self, name):
        fn = getattr(self._cdll, name)
        if not name.startswith("edgl_"):
            return fn

        def call(*args, _fn=fn, _name=name):
            if profiler.enabled and (profiler.only is None or _name in profiler.only):
                import torch
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                rc = _fn(*args)
                b.record()
                profiler.records.setdefault(_name, []).append((a, b))
                return rc
            return _fn(*args)

        self.__dict__[name] = call
        return call


lib = _LibProxy(_cdll)


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib.edgl_last_error()
        raise EdglError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")
