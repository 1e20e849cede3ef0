"""torch.autograd.Function wrappers over the C ABI (include/easydgl_hip.h).

PyTorch is plumbing here: it owns device memory, the stream and the autograd tape; every op body is
a call into libeasydgl_hip.so.  There is no CPU path — handing a CPU tensor to any op raises."""
from __future__ import annotations

import ctypes
import math
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from . import _lib
from ._lib import BF16, F32, check, lib

_DT = {torch.float32: F32, torch.bfloat16: BF16}


def _ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.EdglError("easydgl_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if not t.is_contiguous():
        raise _lib.EdglError("easydgl_amd ops need contiguous tensors")
    return t.data_ptr()


def _vptr(t: torch.Tensor):
    """Pointer of a strided VIEW (a column block of a wider activation buffer); the row stride is passed separately."""
    if not t.is_cuda:
        raise _lib.EdglError("easydgl_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")
    if t.stride(-1) != 1:
        raise _lib.EdglError("easydgl_amd ops need unit stride along the last axis")
    return t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _code(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise _lib.EdglError(f"unsupported activation dtype {t.dtype}")


@dataclass
class Drop:
    """Dropout description handed to a kernel: rate, device RNG state (int64[2]: seed, step), stream id."""
    rate: float = 0.0
    state: Optional[torch.Tensor] = None
    stream_id: int = 0

    @property
    def active(self) -> bool:
        return self.rate > 0.0

    def ptr(self):
        return _ptr(self.state) if self.active else None


NO_DROP = Drop()
MAU_CAUSAL, MAU_NO_DIAG, MAU_DIAG_ZERO = _lib.MAU_CAUSAL, _lib.MAU_NO_DIAG, _lib.MAU_DIAG_ZERO


def make_rng_state(device, seed: int = 9876) -> torch.Tensor:
    """Device RNG state of the counter-based dropout generator: int64[2] = (seed, step)."""
    return torch.tensor([int(seed), 0], device=device, dtype=torch.int64)


def rng_advance(state: torch.Tensor) -> None:
    check(lib.edgl_rng_advance(_ptr(state), _stream()), "edgl_rng_advance")


# ------------------------------------------------------------------------------------------------
# plain helpers (no autograd)
# ------------------------------------------------------------------------------------------------
def gemm(A, Bm, M, N, K, lda, ldb, a_kc, b_kc, out_dtype, bias=None, aux=None, flags=0, splitk=1, ws=None,
         out=None, code=None):
    code = _code(A) if code is None else code
    if out is None:
        out = torch.empty((M, N), device=A.device, dtype=out_dtype)
    if out.dtype == torch.float32:
        flags |= _lib.EPI_OUT_F32
    if splitk > 1 and ws is None:
        ws = torch.empty(splitk * M * N, device=A.device, dtype=torch.float32)
    check(lib.edgl_gemm(_ptr(A), _ptr(Bm), _ptr(out), M, N, K, lda, ldb, out.stride(0), int(a_kc), int(b_kc),
                        _ptr(bias), _ptr(aux), flags, splitk, _ptr(ws), code, _stream()), "edgl_gemm")
    return out


def colsum(X, M, N):
    out = torch.empty(N, device=X.device, dtype=torch.float32)
    ws = torch.empty(256 * N, device=X.device, dtype=torch.float32)
    check(lib.edgl_colsum(_ptr(X), M, N, X.stride(0) if X.dim() == 2 else N, _ptr(out), 0, _ptr(ws),
                          int(X.dtype == torch.float32), _code(X) if X.dtype != torch.float32 else F32, _stream()),
          "edgl_colsum")
    return out


def cast_to(src_f32: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(src_f32.shape, device=src_f32.device, dtype=dtype)
    check(lib.edgl_cast(_ptr(src_f32), _ptr(out), src_f32.numel(), _DT[dtype], _stream()), "edgl_cast")
    return out


def dense_grads(x2, dz, M, K, N):
    code = _code(x2)
    # (dW, db) in one buffer, as they sit in the flat gradient arena: the library then reduces both with one launch
    both = torch.empty((K + 1) * N, device=x2.device, dtype=torch.float32)
    dw, db = both[:K * N].view(K, N), both[K * N:]
    ws = torch.empty(lib.edgl_gemm_dw_workspace(M, K, N, code), device=x2.device, dtype=torch.float32)
    check(lib.edgl_gemm_dw(_ptr(x2), _ptr(dz), _ptr(dw), _ptr(db), M, K, N, x2.stride(0), dz.stride(0), 0, _ptr(ws), code,
                           _stream()), "edgl_gemm_dw")
    return dw, db


def _splitk_for(m_out: int, n_out: int, kc: int) -> int:
    tiles = ((m_out + 127) // 128) * ((n_out + 127) // 128)
    return int(max(1, min(kc // 256, max(1, 320 // tiles))))


# ------------------------------------------------------------------------------------------------
# K1 encode
# ------------------------------------------------------------------------------------------------
class EncodeFn(torch.autograd.Function):
    """EasyDGL.py:70-95.  item_c is the compute copy of the item table (the f32 master itself, or its
    bf16 shadow); gradients flow to the masters."""

    @staticmethod
    def forward(ctx, item_master, pos_tab, mark_emb, item_c, ids, ts, mark_table, tscale, mask_id, time_scale,
                drop: Drop, act_dtype, pad=(0, 0)):
        """pad = (dh_pad, dh_true) of a channel-padded model (model/base.py), (0, 0) otherwise."""
        B, T = ids.shape
        I, C = item_c.shape
        E = mark_table.shape[1]
        x0 = torch.empty((B, T, 3 * C), device=ids.device, dtype=act_dtype)
        spans = torch.empty((B, T), device=ids.device, dtype=torch.float32)
        marks = torch.empty((B, T, E), device=ids.device, dtype=torch.uint8)
        check(lib.edgl_encode_fwd_ct(_ptr(ids), _ptr(ts), _ptr(item_c), _ptr(pos_tab), _ptr(mark_emb), _ptr(mark_table),
                                     _ptr(tscale), B, T, C, E, I, int(mask_id), float(time_scale), float(drop.rate),
                                     drop.ptr(), drop.stream_id, _ptr(x0), _ptr(spans), _ptr(marks), int(pad[0]), int(pad[1]),
                                     _DT[act_dtype], _stream()), "edgl_encode_fwd")
        ctx.save_for_backward(ids, marks)
        ctx.c_true = (C // pad[0] * pad[1]) if pad[0] else 0
        ctx.meta = (B, T, C, E, I, drop, item_master.shape, pos_tab.shape, mark_emb.shape)
        ctx.mark_non_differentiable(spans, marks)
        return x0, spans, marks

    @staticmethod
    def backward(ctx, dx0, _ds, _dm):
        ids, marks = ctx.saved_tensors
        B, T, C, E, I, drop, ishape, pshape, mshape = ctx.meta
        dx0 = dx0.contiguous()
        d_item = torch.zeros(ishape, device=dx0.device, dtype=torch.float32)
        d_pos = torch.zeros(pshape, device=dx0.device, dtype=torch.float32)
        d_mark = torch.empty(mshape, device=dx0.device, dtype=torch.float32)
        ws = torch.empty(lib.edgl_encode_bwd_workspace(B, T, C), device=dx0.device, dtype=torch.float32)
        check(lib.edgl_encode_bwd_add_ct(_ptr(ids), _ptr(marks), _ptr(dx0), None, None, B, T, C, E, I, float(drop.rate), drop.ptr(),
                                         drop.stream_id, _ptr(d_item), _ptr(d_pos), _ptr(d_mark), _ptr(ws), int(ctx.c_true),
                                         _code(dx0), _stream()), "edgl_encode_bwd")
        return (d_item, d_pos, d_mark) + (None,) * 10


class EmbedPosFn(torch.autograd.Function):
    """CTSMA.py:48-58: X0 = dropout(concat(item[ids]*sqrt(C), pos[0..T))), spans, marks (edgl_embed_pos_fwd/bwd)."""

    @staticmethod
    def forward(ctx, item_master, pos_tab, item_c, ids, ts, mark_table, time_scale, drop: Drop, act_dtype, c_true=0):
        """c_true: the true width of a channel-padded model (0: = C) — coding.py:62-63's sqrt(num_units)."""
        B, T = ids.shape
        I, C = item_c.shape
        E = mark_table.shape[1]
        x0 = torch.empty((B, T, 2 * C), device=ids.device, dtype=act_dtype)
        spans = torch.empty((B, T), device=ids.device, dtype=torch.float32)
        marks = torch.empty((B, T, E), device=ids.device, dtype=torch.uint8)
        check(lib.edgl_embed_pos_fwd_ct(_ptr(ids), _ptr(ts), _ptr(item_c), _ptr(pos_tab), _ptr(mark_table), B, T, C, E,
                                        float(time_scale), float(drop.rate), drop.ptr(), drop.stream_id, _ptr(x0), _ptr(spans),
                                        _ptr(marks), int(c_true), _DT[act_dtype], _stream()), "edgl_embed_pos_fwd")
        ctx.save_for_backward(ids)
        ctx.c_true = int(c_true)
        ctx.meta = (B, T, C, I, drop, item_master.shape, pos_tab.shape)
        ctx.mark_non_differentiable(spans, marks)
        return x0, spans, marks

    @staticmethod
    def backward(ctx, dx0, _ds, _dm):
        (ids,) = ctx.saved_tensors
        B, T, C, I, drop, ishape, pshape = ctx.meta
        dx0 = dx0.contiguous()
        d_item = torch.empty(ishape, device=dx0.device, dtype=torch.float32)
        d_pos = torch.zeros(pshape, device=dx0.device, dtype=torch.float32)
        check(lib.edgl_embed_pos_bwd_ct(_ptr(ids), _ptr(dx0), B, T, C, I, float(drop.rate), drop.ptr(), drop.stream_id,
                                        _ptr(d_item), _ptr(d_pos), ctx.c_true, _code(dx0), _stream()), "edgl_embed_pos_bwd")
        return (d_item, d_pos) + (None,) * 8


# ------------------------------------------------------------------------------------------------
# K2/K4 dense
# ------------------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """tf.layers.dense: y = act(x @ W + b), W [in, out] (Appendix A).  w_c is W in the activation dtype."""

    @staticmethod
    def forward(ctx, x, w_master, bias, w_c, gelu):
        """`gelu`: False / True (erf-GELU, EasyDGL.py:19-32) / "relu" (FeedForward inner layer, Base.py:73)."""
        K, N = w_c.shape
        x2 = x.reshape(-1, K).contiguous()
        M = x2.shape[0]
        relu = gelu == "relu"
        gelu = gelu is True
        flags = _lib.EPI_BIAS | (_lib.EPI_GELU | _lib.EPI_SAVE_PRE if gelu else 0) | (_lib.EPI_RELU if relu else 0)
        pre = torch.empty((M, N), device=x.device, dtype=x.dtype) if gelu else None
        y = gemm(x2, w_c, M, N, K, K, N, True, False, x.dtype, bias=bias, aux=pre, flags=flags)
        ctx.save_for_backward(x2, w_c, y if relu else pre)
        ctx.meta = (M, N, K, "relu" if relu else gelu, x.shape)
        return y.reshape(x.shape[:-1] + (N,))

    @staticmethod
    def backward(ctx, dy):
        x2, w_c, pre = ctx.saved_tensors
        M, N, K, gelu, xshape = ctx.meta
        dz = dy.reshape(M, N).contiguous()
        if gelu == "relu":
            out = torch.empty_like(dz)
            check(lib.edgl_relu_bwd(_ptr(dz), _ptr(pre), _ptr(out), dz.numel(), _code(dz), _stream()), "edgl_relu_bwd")
            dz = out
        elif gelu:
            out = torch.empty_like(dz)
            check(lib.edgl_gelu_bwd(_ptr(dz), _ptr(pre), _ptr(out), dz.numel(), _code(dz), _stream()), "edgl_gelu_bwd")
            dz = out
        # dX[M,K] = dz[M,N] . W^T : B(kk=n, nn=k) = W[k][n] -> stored [K rows][N contiguous] = k-contiguous operand
        dx = gemm(dz, w_c, M, K, N, N, N, True, True, dz.dtype)
        # dW[K,N] = X^T . dz and db = colsum(dz): contraction over the rows of two row-major operands
        dw, db = dense_grads(x2, dz, M, K, N)
        return dx.reshape(xshape), dw, db, None, None


class DualLinearFn(torch.autograd.Function):
    """Q = dense(xq, C) and K | V | T_ = dense(xk, 3C) written as the column blocks of ONE [.., 4C] operand — MAU.__call__'s four
    projections (temporal.py:352-355) in the layout the fused attention kernel reads, without a concatenation copy (forward) or
    slice copies (backward): the GEMMs take the row stride 4C of the shared buffer."""

    @staticmethod
    def forward(ctx, xq, xk, wq_master, bq, wq_c, wk_master, bk, wk_c):
        Kq, C = wq_c.shape
        Kk, C3 = wk_c.shape
        xq2, xk2 = xq.reshape(-1, Kq).contiguous(), xk.reshape(-1, Kk).contiguous()
        M = xq2.shape[0]
        out = torch.empty((M, C + C3), device=xq.device, dtype=xq.dtype)
        code, st = _code(xq2), _stream()
        for x2, w, b, K, N, col in ((xq2, wq_c, bq, Kq, C, 0), (xk2, wk_c, bk, Kk, C3, C)):
            check(lib.edgl_gemm(_ptr(x2), _ptr(w), _vptr(out[:, col:]), M, N, K, K, N, out.stride(0), 1, 0, _ptr(b), None, _lib.EPI_BIAS, 1,
                                None, code, st), "edgl_gemm")
        ctx.save_for_backward(xq2, xk2, wq_c, wk_c)
        ctx.meta = (M, C, C3, Kq, Kk, xq.shape, xk.shape)
        return out.reshape(xq.shape[:-1] + (C + C3,))

    @staticmethod
    def backward(ctx, dy):
        xq2, xk2, wq_c, wk_c = ctx.saved_tensors
        M, C, C3, Kq, Kk, qshape, kshape = ctx.meta
        dz = dy.reshape(M, C + C3).contiguous()
        code, st = _code(dz), _stream()
        res = []
        for x2, w, K, N, col in ((xq2, wq_c, Kq, C, 0), (xk2, wk_c, Kk, C3, C)):
            dzv = dz[:, col:col + N]                                   # column block, row stride 4C
            dx = torch.empty((M, K), device=dz.device, dtype=dz.dtype)
            check(lib.edgl_gemm(_vptr(dzv), _ptr(w), _ptr(dx), M, K, N, dz.stride(0), N, K, 1, 1, None, None, 0, 1, None, code, st),
                  "edgl_gemm")
            both = torch.empty((K + 1) * N, device=dz.device, dtype=torch.float32)
            ws = torch.empty(lib.edgl_gemm_dw_workspace(M, K, N, code), device=dz.device, dtype=torch.float32)
            check(lib.edgl_gemm_dw(_ptr(x2), _vptr(dzv), _ptr(both), both.data_ptr() + 4 * K * N, M, K, N, K, dz.stride(0), 0, _ptr(ws), code,
                                   st), "edgl_gemm_dw")
            res.append((dx, both[:K * N].view(K, N), both[K * N:]))
        (dxq, dwq, dbq), (dxk, dwk, dbk) = res
        return dxq.reshape(qshape), dxk.reshape(kshape), dwq, dbq, None, dwk, dbk, None


# ------------------------------------------------------------------------------------------------
# K3 BiMAU
# ------------------------------------------------------------------------------------------------
class BiMAUFn(torch.autograd.Function):
    """Fused attention of BiMAU.__call__ (temporal.py:413-447) given qkvt = dense(x)."""

    @staticmethod
    def forward(ctx, qkvt, resid, W1, b1, w, scaling, ids, spans, marks, H, drop: Drop, flags: int = 0, qk_scale: float = 0.0):
        """qk_scale: the score scale (0 = 1 / sqrt(head dim)); a channel-padded model passes 1 / sqrt(true head dim)."""
        B, T, C4 = qkvt.shape
        C = C4 // 4
        E = w.shape[0]
        code = _code(qkvt)
        # the kernel reads `ids` as B*T 64-bit words: the reference's [h*B,T,T] float mask handed through unchanged would be read
        # as garbage ids without any error (module/temporal.py key_ids_from_masks converts it)
        if ids.dtype != torch.int64 or tuple(ids.shape) != (B, T) or not ids.is_contiguous() or ids.device != qkvt.device:
            raise _lib.EdglError(f"BiMAU: key ids / mask must be a contiguous int64 [B={B}, T={T}] tensor on {qkvt.device}, got "
                                 f"{tuple(ids.shape)} {ids.dtype}")
        pack = torch.empty(lib.edgl_bimau_pack_bytes(C, H, E, code), device=qkvt.device, dtype=torch.uint8)
        check(lib.edgl_bimau_pack(_ptr(W1), _ptr(b1), _ptr(w), _ptr(scaling), C, H, E, _ptr(pack), code, _stream()),
              "edgl_bimau_pack")
        out = torch.empty((B, T, C), device=qkvt.device, dtype=qkvt.dtype)
        lam = torch.empty((H * B, T, E), device=qkvt.device, dtype=torch.float32)
        if resid.stride(-1) != 1 or resid.stride(0) != T * resid.stride(1):
            raise _lib.EdglError("BiMAU residual must be a row-strided view of a contiguous [B,T,*] tensor")
        need_grad = any(ctx.needs_input_grad)
        # head dims >= 64 run as scores phase -> intensity kernel -> values phase and hand H rows / z through `saved`
        need_saved = need_grad or (C // H) >= 64
        saved = torch.empty(lib.edgl_bimau_saved_bytes(B, T, C, H, code), device=qkvt.device, dtype=torch.uint8) if need_saved else None
        check(lib.edgl_bimau_fwd_db(_ptr(qkvt), resid.data_ptr(), resid.stride(1), _ptr(ids), _ptr(spans), _ptr(marks),
                                    _ptr(pack), B, T, C, H, E, float(drop.rate), drop.ptr(), drop.stream_id, None, float(qk_scale),
                                    _ptr(out), _ptr(lam), _ptr(saved), None, int(flags), code, _stream()), "edgl_bimau_fwd")
        if need_grad:
            ctx.save_for_backward(qkvt, ids, spans, marks, pack, lam, saved)
        ctx.qk_scale = float(qk_scale)
        ctx.meta = (B, T, C, H, E, drop, code, W1.shape, b1.shape, w.shape, scaling.shape, int(flags))
        return out, lam

    @staticmethod
    def backward(ctx, d_out, d_lam):
        qkvt, ids, spans, marks, pack, lam, saved = ctx.saved_tensors
        B, T, C, H, E, drop, code, s1, s2, s3, s4, flags = ctx.meta
        d_out = d_out.contiguous()
        dev = d_out.device
        d_qkvt = torch.empty_like(qkvt)
        # dW1 | db1 | dw | dscaling in one buffer (the flat-arena order): the library reduces all four with one launch
        n1, n2, n3, n4 = (int(np.prod(sh)) for sh in (s1, s2, s3, s4))
        both = torch.empty(n1 + n2 + n3 + n4, device=dev, dtype=torch.float32)
        dW1, db1 = both[:n1].view(s1), both[n1:n1 + n2].view(s2)
        dw, dsc = both[n1 + n2:n1 + n2 + n3].view(s3), both[n1 + n2 + n3:].view(s4)
        ws = torch.empty(lib.edgl_bimau_bwd_workspace(B, T, C, H, E, code), device=dev, dtype=torch.uint8)
        dl = d_lam.contiguous() if d_lam is not None else None
        check(lib.edgl_bimau_bwd_db(_ptr(qkvt), _ptr(ids), _ptr(spans), _ptr(marks), _ptr(pack), _ptr(d_out), _ptr(dl),
                                    _ptr(lam), _ptr(saved), B, T, C, H, E, float(drop.rate), drop.ptr(), drop.stream_id, None,
                                    ctx.qk_scale, _ptr(d_qkvt), _ptr(dW1), _ptr(db1), _ptr(dw), _ptr(dsc), _ptr(ws), flags, code,
                                    _stream()), "edgl_bimau_bwd")
        return d_qkvt, d_out, dW1, db1, dw, dsc, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# K4-LN
# ------------------------------------------------------------------------------------------------
class AddLayerNormFn(torch.autograd.Function):
    """y = layernorm_joint(dropout(x) + resid) (Base.py:12-67; EasyDGL.py:114-116,126-128,139), optionally
    emitting only the rows gather_pos [B,Mg] (EasyDGL.py:142-146)."""

    @staticmethod
    def forward(ctx, x, resid, gamma, beta, drop: Drop, gather_pos, pad=(0, 0)):
        B, T, C = x.shape
        code = _code(x)
        stats = torch.empty((B, 2), device=x.device, dtype=torch.float32)
        Mg = 0 if gather_pos is None else gather_pos.shape[1]
        y = torch.empty((B * Mg, C) if gather_pos is not None else (B, T, C), device=x.device, dtype=x.dtype)
        rptr, ld = (None, 0)
        if resid is not None:
            if resid.stride(-1) != 1 or resid.stride(0) != T * resid.stride(1):
                raise _lib.EdglError("layernorm residual must be a row-strided view of a contiguous [B,T,*] tensor")
            rptr, ld = resid.data_ptr(), resid.stride(1)
        check(lib.edgl_add_layernorm_fwd_ct(_ptr(x), rptr, ld, _ptr(gamma), _ptr(beta), B, T, C, float(drop.rate),
                                            drop.ptr(), drop.stream_id, _ptr(gather_pos), Mg, _ptr(y), _ptr(stats),
                                            int(pad[0]), int(pad[1]), code, _stream()), "edgl_add_layernorm_fwd")
        ctx.save_for_backward(x, resid, gamma, stats, gather_pos)
        ctx.pad = (int(pad[0]), int(pad[1]))
        ctx.meta = (B, T, C, drop, code, Mg)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, resid, gamma, stats, gather_pos = ctx.saved_tensors
        B, T, C, drop, code, Mg = ctx.meta
        dy = dy.contiguous()
        dsum = torch.empty_like(x)
        dxd = torch.empty_like(x) if drop.active else None
        both = torch.empty(2 * C, device=x.device, dtype=torch.float32)   # dbeta | dgamma adjacent: one reduction launch
        db, dg = both[:C], both[C:]
        ws = torch.empty(B * 2 * C, device=x.device, dtype=torch.float32)
        rptr, ld = (None, 0) if resid is None else (resid.data_ptr(), resid.stride(1))
        check(lib.edgl_add_layernorm_bwd_act_ct(_ptr(x), rptr, ld, _ptr(gamma), _ptr(stats), _ptr(dy), B, T, C,
                                                float(drop.rate), drop.ptr(), drop.stream_id, _ptr(gather_pos), Mg, None, None,
                                                _ptr(dsum), _ptr(dxd), _ptr(dg), _ptr(db), _ptr(ws), ctx.pad[0], ctx.pad[1], code,
                                                _stream()), "edgl_add_layernorm_bwd")
        dx = dxd if drop.active else dsum
        return dx, (dsum if resid is not None else None), dg, db, None, None, None


# ------------------------------------------------------------------------------------------------
# K5 scoring + CE
# ------------------------------------------------------------------------------------------------
def compact_rows(rows, labels):
    """Weighted rows (label != 0) first: returns rows_c, labels_c, perm, inv, nvalid (all on device)."""
    R, C = rows.shape
    dev = rows.device
    perm = torch.empty(R, device=dev, dtype=torch.int32)
    inv = torch.empty(R, device=dev, dtype=torch.int32)
    nvalid = torch.empty(1, device=dev, dtype=torch.int32)
    rows_c = torch.empty_like(rows)
    labels_c = torch.empty_like(labels)
    check(lib.edgl_compact_rows(_ptr(rows), _ptr(labels), R, C, _ptr(perm), _ptr(inv), _ptr(nvalid), _ptr(rows_c),
                                _ptr(labels_c), _code(rows), _stream()), "edgl_compact_rows")
    return rows_c, labels_c, perm, inv, nvalid


def score_lse(rows, table_c, out_bias, labels, i0, i1, want_logits=False, nvalid=None, want_lse=True):
    R, C = rows.shape
    I = table_c.shape[0]
    dev = rows.device
    lse = torch.empty(R, device=dev, dtype=torch.float32) if want_lse else None     # (None: the logits tile only — evaluation)
    lab_logit = torch.zeros(R, device=dev, dtype=torch.float32) if labels is not None or want_lse else None
    logits = torch.empty((R, i1 - i0), device=dev, dtype=torch.float32) if want_logits else None
    ws = torch.empty(2 * R * lib.edgl_score_chunks(R, i1 - i0), device=dev, dtype=torch.float32)
    check(lib.edgl_score_lse_fwd(_ptr(rows), _ptr(table_c), _ptr(out_bias), _ptr(labels), R, C, I, i0, i1, _ptr(nvalid),
                                 _ptr(lse), _ptr(lab_logit), _ptr(logits), _ptr(ws), _code(rows), _stream()),
          "edgl_score_lse_fwd")
    return lse, lab_logit, logits


VP_FLASH = os.environ.get("EDGL_VP_FLASH", "1") != "0"      # vocab-parallel loss through the flash / strip kernels (0: the two-pass kernels)


def vocab_parallel_ce(rows, table_c, out_bias, labels, group=None):
    """parallel.vocab_parallel_ce with the HIP scoring kernels: every rank holds the same rows [R, C] / labels [R] and scores them
    against its row shard of the (replicated, un-scaled, tied) item table over the item range [i0, i1) with the GLOBAL log-sum-exp —
    the flash form (edgl_score_flash_fwd / _bwd: the strip kernels at the bf16 widths they cover; the row finish takes the gathered
    sums) or, with EDGL_VP_FLASH=0, the two-pass kernels edgl_score_lse_fwd / edgl_score_ce_bwd.  Returns (loss, d_rows [R, C] f32, d_table [I, C] f32 — zero outside the
    rank's shard —, d_bias [I - 1] f32 — zero outside it —, (i0, i1)).  EasyDGL.py:149-155,177-185."""
    from . import parallel
    rows = rows.contiguous()
    labels = labels.reshape(-1).contiguous()
    R, C = rows.shape
    I = table_c.shape[0]
    dev, code, st = rows.device, _code(rows), _stream()
    d_table = torch.zeros((I, C), device=dev, dtype=torch.float32)
    d_bias = torch.zeros(I - 1, device=dev, dtype=torch.float32)
    if VP_FLASH:
        # the flash form over the rank's item range (the strip kernels at bf16 C = 128 / 256 / 512): ONE sweep gives the range's
        # log-sum-exp and the unnormalised row gradients, the backward call finishes them with the GLOBAL log-sum-exp (the label row
        # leaves on the rank that owns the label) and runs the table pass over the range — weighted rows first, as the training step
        rows_c, lab_c, _perm, inv, nvalid = compact_rows(rows, labels)
        state = {}

        def lse_flash(i0, i1):
            ws = torch.empty(int(lib.edgl_score_flash_workspace(R, C, I, i1 - i0, code)), device=dev, dtype=torch.float32)
            lse = torch.empty(R, device=dev, dtype=torch.float32)
            lab = torch.zeros(R, device=dev, dtype=torch.float32)
            check(lib.edgl_score_flash_fwd(_ptr(rows_c), _ptr(table_c), _ptr(out_bias), _ptr(lab_c), R, C, I, i0, i1, _ptr(nvalid), _ptr(lse),
                                           _ptr(lab), _ptr(ws), code, st), "edgl_score_flash_fwd")
            state["ws"] = ws
            own = (lab_c >= max(i0, 1)) & (lab_c < i1)
            live = torch.arange(R, device=dev) < nvalid.to(torch.long)
            # rows behind the weighted ones carry label 0 (weight 0): a finite log-sum-exp of their own keeps the gathered sums finite
            return torch.where(live, lse, torch.zeros_like(lse)), torch.where(own, lab, torch.full_like(lab, float("-inf")))

        def grad_flash(i0, i1, lse, coef):
            d_rows = torch.empty_like(rows_c)
            check(lib.edgl_score_flash_bwd(_ptr(rows_c), _ptr(table_c), _ptr(out_bias), _ptr(lab_c), _ptr(lse.contiguous()), _ptr(coef.contiguous()),
                                           None, R, C, I, i0, i1, _ptr(nvalid), _ptr(d_rows), _ptr(d_table), _ptr(d_bias), _ptr(state["ws"]), code, st),
                  "edgl_score_flash_bwd")
            return d_rows

        def ce_hip(lse, lab, labels_c):      # loss + coefficients of the merged sums in one launch (EasyDGL.py:177-185)
            loss = torch.empty(1, device=dev, dtype=torch.float32)
            coef = torch.empty(R, device=dev, dtype=torch.float32)
            check(lib.edgl_ce_loss_fwd(_ptr(lse), _ptr(lab), _ptr(labels_c), R, _ptr(loss), _ptr(coef), st), "edgl_ce_loss_fwd")
            return loss.reshape(()), coef

        loss, d_rows_c, (i0, i1) = parallel.vocab_parallel_ce(lab_c, I, lse_flash, grad_flash, group, ce_local=ce_hip)
        return loss, d_rows_c[inv.long()], d_table, d_bias, (i0, i1)

    def lse_local(i0, i1):
        lse, lab, _ = score_lse(rows, table_c, out_bias, labels, i0, i1)
        own = (labels >= i0) & (labels < i1)
        return lse, torch.where(own, lab, torch.full_like(lab, float("-inf")))

    def grad_local(i0, i1, lse, coef):
        d_rows = torch.empty_like(rows)
        ws = torch.empty(int(lib.edgl_score_bwd_workspace(R, C, I, i1 - i0, code)), device=dev, dtype=torch.float32)
        check(lib.edgl_score_ce_bwd(_ptr(rows), _ptr(table_c), _ptr(out_bias), _ptr(labels), _ptr(lse), _ptr(coef), None, R, C, I,
                                    i0, i1, None, _ptr(d_rows), _ptr(d_table), _ptr(d_bias), _ptr(ws), code, st), "edgl_score_ce_bwd")
        return d_rows

    loss, d_rows, (i0, i1) = parallel.vocab_parallel_ce(labels, I, lse_local, grad_local, group)
    return loss, d_rows, d_table, d_bias, (i0, i1)


class ScoreCEFn(torch.autograd.Function):
    """EasyDGL.py:149-155,177-185 without the [R, I] logits tensor.  Rows with label 0 carry weight 0
    (EasyDGL.py:180); they are compacted away before scoring, which changes neither loss nor gradients.
    When a gradient is wanted the forward is the flash form (edgl_score_flash_fwd): the same sweep over the item table that
    gives the row log-sum-exp also accumulates the row gradients, and the backward only finishes them."""

    @staticmethod
    def forward(ctx, rows, table_master, out_bias, table_c, labels):
        rows, labels, _perm, inv, nvalid = compact_rows(rows.contiguous(), labels.reshape(-1).contiguous())
        R, C = rows.shape
        I = table_c.shape[0]
        flash = any(ctx.needs_input_grad)
        ws = None
        if flash:
            code = _code(rows)
            ws = torch.empty(int(lib.edgl_score_flash_workspace(R, C, I, I, code)), device=rows.device, dtype=torch.float32)
            lse = torch.empty(R, device=rows.device, dtype=torch.float32)
            lab_logit = torch.zeros(R, device=rows.device, dtype=torch.float32)
            check(lib.edgl_score_flash_fwd(_ptr(rows), _ptr(table_c), _ptr(out_bias), _ptr(labels), R, C, I, 0, I, _ptr(nvalid),
                                           _ptr(lse), _ptr(lab_logit), _ptr(ws), code, _stream()), "edgl_score_flash_fwd")
        else:
            lse, lab_logit, _ = score_lse(rows, table_c, out_bias, labels, 0, I, nvalid=nvalid)
        loss = torch.empty(1, device=rows.device, dtype=torch.float32)
        coef = torch.empty(R, device=rows.device, dtype=torch.float32)
        check(lib.edgl_ce_loss_fwd(_ptr(lse), _ptr(lab_logit), _ptr(labels), R, _ptr(loss), _ptr(coef), _stream()),
              "edgl_ce_loss_fwd")
        ctx.save_for_backward(rows, table_c, out_bias, labels, lse, coef, inv, nvalid, ws)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        rows, table_c, out_bias, labels, lse, coef, inv, nvalid, ws = ctx.saved_tensors
        R, C = rows.shape
        I = table_c.shape[0]
        dev = rows.device
        g = g.reshape(1).to(torch.float32).contiguous()
        d_rows_c = torch.empty_like(rows)
        d_rows = torch.empty_like(rows)
        d_table = torch.empty((I, C), device=dev, dtype=torch.float32)
        d_bias = torch.empty(I - 1, device=dev, dtype=torch.float32)
        check(lib.edgl_score_flash_bwd(_ptr(rows), _ptr(table_c), _ptr(out_bias), _ptr(labels), _ptr(lse), _ptr(coef),
                                       _ptr(g), R, C, I, 0, I, _ptr(nvalid), _ptr(d_rows_c), _ptr(d_table), _ptr(d_bias),
                                       _ptr(ws), _code(rows), _stream()), "edgl_score_flash_bwd")
        check(lib.edgl_scatter_rows(_ptr(d_rows_c), _ptr(inv), R, C, _ptr(d_rows), _code(rows), _stream()),
              "edgl_scatter_rows")
        return d_rows, d_table, d_bias, None, None


class ScoreLogitsFn(torch.autograd.Function):
    """Materialised logits [R, I] (EasyDGL.py:149-151) for API parity with the reference's __call__."""

    @staticmethod
    def forward(ctx, rows, table_master, out_bias, table_c):
        I = table_c.shape[0]
        _, _, logits = score_lse(rows, table_c, out_bias, None, 0, I, want_logits=True)
        ctx.save_for_backward(rows, table_c)
        return logits

    @staticmethod
    def backward(ctx, dl):
        rows, table_c = ctx.saved_tensors
        R, C = rows.shape
        I = table_c.shape[0]
        dl = dl.contiguous()
        dl[:, 0] = 0  # column 0 is the constant -1000 (Base.py:110) over a constant zero row
        dlc = dl if rows.dtype == torch.float32 else cast_to(dl, rows.dtype)
        tab = table_c.clone()
        tab[0] = 0  # coding.py:56-57
        d_rows = gemm(dlc, tab, R, C, I, I, C, True, False, rows.dtype)
        d_table = gemm(dlc, rows, I, C, R, I, C, False, False, torch.float32, splitk=_splitk_for(I, C, R))
        d_table[0] = 0
        d_bias = dl.sum(0)[1:]
        return d_rows, d_table, d_bias, None


# ------------------------------------------------------------------------------------------------
# K8 TPP regulariser, l2
# ------------------------------------------------------------------------------------------------
class TppFn(torch.autograd.Function):
    """ct_reg/h * MAU.biased_likelihood on the gathered intensities (EasyDGL.py:157-175)."""

    @staticmethod
    def forward(ctx, lam, masked_pos, labels, ts_raw, mark_table, H, coef):
        HB, T, E = lam.shape
        B, M = labels.shape if masked_pos is None else masked_pos.shape
        sums = torch.empty(lib.edgl_tpp_workspace(), device=lam.device, dtype=torch.float32)
        reg = torch.empty(1, device=lam.device, dtype=torch.float32)
        check(lib.edgl_tpp_fwd(_ptr(lam), _ptr(masked_pos), _ptr(labels), _ptr(ts_raw), _ptr(mark_table), B, T, H, E, M,
                               float(coef), _ptr(sums), _ptr(reg), 0, _stream()), "edgl_tpp_fwd")
        ctx.save_for_backward(lam, masked_pos, labels, ts_raw, mark_table, sums)
        ctx.meta = (B, T, H, E, M, coef)
        return reg.reshape(())

    @staticmethod
    def backward(ctx, g):
        lam, masked_pos, labels, ts_raw, mark_table, sums = ctx.saved_tensors
        B, T, H, E, M, coef = ctx.meta
        g = g.reshape(1).to(torch.float32).contiguous()
        d_lam = torch.empty_like(lam)
        check(lib.edgl_tpp_bwd(_ptr(lam), _ptr(masked_pos), _ptr(labels), _ptr(ts_raw), _ptr(mark_table), B, T, H, E, M,
                               float(coef), _ptr(sums), _ptr(g), _ptr(d_lam), _stream()), "edgl_tpp_bwd")
        return d_lam, None, None, None, None, None, None


_SEG_CACHE = {}


def _whole_segment(n: int, device) -> torch.Tensor:
    """[0, n] on the device, built once per (n, device): no host-to-device copy inside a step (HIP graph capture forbids it)."""
    key = (n, str(device))
    if key not in _SEG_CACHE:
        _SEG_CACHE[key] = torch.tensor([0, n], device=device, dtype=torch.int64)
    return _SEG_CACHE[key]


class L2Fn(torch.autograd.Function):
    """l2_reg * sum(w^2)/2 over one raw embedding table (coding.py:34-40)."""

    @staticmethod
    def forward(ctx, w, l2):
        seg = _whole_segment(w.numel(), w.device)
        out = torch.empty(1, device=w.device, dtype=torch.float32)
        ws = torch.empty(1024, device=w.device, dtype=torch.float32)
        check(lib.edgl_l2_loss(_ptr(w), _ptr(seg), 1, float(l2), _ptr(out), 0, _ptr(ws), _stream()), "edgl_l2_loss")
        ctx.save_for_backward(w)
        ctx.l2 = l2
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return w * (g * ctx.l2), None


# ------------------------------------------------------------------------------------------------
# K6/K7 evaluation
# ------------------------------------------------------------------------------------------------
def mask_topk(logits: torch.Tensor, i0: int, seen: Optional[torch.Tensor], K: int):
    R, n = logits.shape
    val = torch.empty((R, K), device=logits.device, dtype=torch.float32)
    idx = torch.empty((R, K), device=logits.device, dtype=torch.int32)
    T = 0 if seen is None else seen.shape[1]
    check(lib.edgl_mask_topk(_ptr(logits), R, n, i0, _ptr(seen), T, K, _ptr(val), _ptr(idx), _stream()), "edgl_mask_topk")
    return val, idx


def topk_merge(cand_val: torch.Tensor, cand_idx: torch.Tensor):
    S, R, K = cand_val.shape
    val = torch.empty((R, K), device=cand_val.device, dtype=torch.float32)
    idx = torch.empty((R, K), device=cand_val.device, dtype=torch.int32)
    check(lib.edgl_topk_merge(_ptr(cand_val), _ptr(cand_idx), S, R, K, _ptr(val), _ptr(idx), _stream()), "edgl_topk_merge")
    return val, idx


EVAL_TILE_BYTES = 64 << 20   # logits staging tile of score_topk: [R, chunk] f32, sized to stay L2 / MALL resident
TOPK_REG_ITEMS = (256 * 80 - 8) // 8 * 8   # longest row of the register form of edgl_mask_topk (csrc/k_score.hip), a multiple of 8
EVAL_FUSED = os.environ.get("EDGL_EVAL_FUSED", "1") != "0"   # scoring + seen mask + top-K without a logits tile (k_eval_topk.hip)
EVAL_GEMM = True             # full item chunks of the chunked evaluation path score through edgl_gemm (tests switch it off)


EVAL_FUSED_SCRATCH_BYTES = int(os.environ.get("EDGL_EVAL_FUSED_SCRATCH_BYTES", str(512 << 20)))
_FUSED_EVAL_WS = {}


def _fused_eval_workspace(device, nbytes: int) -> torch.Tensor:
    """Grow-only workspace of the fused evaluation scoring, one per device and stream (the launches of one stream are ordered, so
    consecutive calls may share it): no allocation per call."""
    key = (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _FUSED_EVAL_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), device=device, dtype=torch.uint8)
        _FUSED_EVAL_WS[key] = ws
    return ws


def score_topk(rows, table_c, out_bias, seen, K, i0, i1):
    """Sequential.eval's scoring + seen-mask + top-K (Base.py:150-181) over the item range [i0, i1) WITHOUT the [R, I]
    logits tensor of the reference: the range is walked in chunks whose [R, chunk] f32 logits tile is bounded by
    EVAL_TILE_BYTES (2 GB per batch at |items| = 1M otherwise); every chunk gives a local top-K with global ids and the
    merge kernel orders the candidates by (value desc, id asc) — the tie rule of tf.nn.top_k.  Returns (val, idx) [R, K]."""
    R = rows.shape[0]
    n = i1 - i0
    # the fused form (csrc/k_eval_topk.hip): no logits tile at all — two sweeps on the matrix pipe, candidates above a per-row bound,
    # one ranking launch; item ranges of <= 262 144 per call, longer catalogues in chunks + the merge kernel
    if EVAL_FUSED and K <= 128 and rows.dtype == torch.bfloat16 and i0 % 8 == 0:
        if seen is not None:
            # the sweep reads a row's seen ids as 16-byte pairs of int64 at a row stride of T (csrc/k_eval_topk.hip eval_sweep_kernel)
            if seen.dtype != torch.int64:
                raise _lib.EdglError(f"score_topk: seen ids must be int64 (got {seen.dtype})")
            seen = seen.contiguous()
            if seen.data_ptr() % 16 != 0:
                seen = seen.clone()                     # (an offset view of a larger buffer: a fresh allocation is 256-byte aligned)
        T = 0 if seen is None else seen.shape[1]
        C, code, st = rows.shape[1], _code(rows), _stream()
        fchunk = 131072 if C == 256 else 262144      # (64 item slices of <= 2048 / 4096 items: the seen bitmap's LDS)
        starts = list(range(i0, i1, fchunk))
        # (a short last chunk joins its neighbour's range when that keeps both callable: the fused form wants >= 4096 items)
        if len(starts) > 1 and (i1 - starts[-1]) < 4096:
            starts[-1] = max(starts[-2] + 8, (i1 - 4096) // 8 * 8)
        bounds = [(lo, (starts[j + 1] if j + 1 < len(starts) else i1)) for j, lo in enumerate(starts)]
        # The plan reserves an [R, items] f32 scratch for the exact fallback of rows whose candidate list overflows (almost never
        # touched): bounded by walking the rows in blocks, so that a large evaluation batch on a large catalogue holds at most
        # EVAL_FUSED_SCRATCH_BYTES of it — the unfused path it replaces is bounded by EVAL_TILE_BYTES the same way
        span = max(hi - lo for lo, hi in bounds)
        rb = R if R * span * 4 <= EVAL_FUSED_SCRATCH_BYTES else max(128, (EVAL_FUSED_SCRATCH_BYTES // (span * 4)) // 128 * 128)
        rblocks = [(r0, min(R, r0 + rb)) for r0 in range(0, R, rb)]
        rsizes = sorted({r1 - r0 for r0, r1 in rblocks})
        if all(lib.edgl_score_topk_fused_supported(rn, C, hi - lo, T, K, code) for lo, hi in bounds for rn in rsizes) \
                and (len(bounds) == 1 or K <= 512):
            wsb = max(int(lib.edgl_score_topk_fused_workspace(rn, C, hi - lo, T, K)) for lo, hi in bounds for rn in rsizes)
            ws = _fused_eval_workspace(rows.device, wsb)
            cval = torch.empty((len(bounds), R, K), device=rows.device, dtype=torch.float32)
            cidx = torch.empty((len(bounds), R, K), device=rows.device, dtype=torch.int32)
            for j, (lo, hi) in enumerate(bounds):
                for r0, r1 in rblocks:
                    check(lib.edgl_score_topk_fused(_ptr(rows[r0:r1]), _ptr(table_c), _ptr(out_bias),
                                                    None if seen is None else _ptr(seen[r0:r1]), T, r1 - r0, C, table_c.shape[0], lo, hi, K,
                                                    _ptr(cval[j, r0:r1]), _ptr(cidx[j, r0:r1]), _ptr(ws), code, st), "edgl_score_topk_fused")
            return _merge_lists(cval, cidx, R, K)
    chunk = max(1024, (EVAL_TILE_BYTES // (4 * R)) // 8 * 8)
    if K <= 128 and chunk > TOPK_REG_ITEMS >= 1024:      # rows the top-K kernel keeps in registers: one read of the tile instead of five
        chunk = TOPK_REG_ITEMS // 128 * 128              # (a multiple of 128: full chunks take the tiled GEMM below)
    if n <= chunk:
        _, _, logits = score_lse(rows, table_c, out_bias, None, i0, i1, want_logits=True, want_lse=False)
        return mask_topk(logits, i0, seen, K)
    starts = list(range(i0, i1, chunk))                  # chunk starts stay multiples of 8 (i0 is one)
    if len(starts) > 1 and K > 512:
        raise _lib.EdglError(f"score_topk: K={K} > 512 on the chunked path (the merge kernel orders up to 1024 candidates per row: "
                             f"two K-lists at least)")
    # One logits tile, one scoring workspace and the candidate lists of ALL chunks are allocated once: per chunk the loop is two
    # library calls (at 1 M items and 49 chunks the allocations and wrappers of the per-chunk form kept the GPU waiting for the host)
    C, I, dev, code, st = rows.shape[1], table_c.shape[0], rows.device, _code(rows), _stream()
    logits = torch.empty((R, chunk), device=dev, dtype=torch.float32)
    ws = torch.empty(2 * R * lib.edgl_score_chunks(R, chunk), device=dev, dtype=torch.float32)
    cval = torch.empty((len(starts), R, K), device=dev, dtype=torch.float32)
    cidx = torch.empty((len(starts), R, K), device=dev, dtype=torch.int32)
    T = 0 if seen is None else seen.shape[1]
    esz = rows.element_size()
    # bias of item z at element z (item 0, the padding item, never read through it): 16-byte aligned for every chunk start
    bias_z = torch.cat([out_bias.new_zeros(1), out_bias]) if (code == BF16 and EVAL_GEMM and len(starts) > 2) else None
    for j, lo in enumerate(starts):
        hi = min(i1, lo + chunk)
        if bias_z is not None and lo >= 1 and (hi - lo) % 128 == 0 and C % 32 == 0:
            # a full chunk behind the padding item: logits = rows . table[lo:hi]^T + bias as a plain GEMM with f32 output (the
            # weights-resident strip kernel against the scoring kernel's 45-85 us at 512 x 20 352 x 128 / 256); bias of item z is
            # out_bias[z - 1] (EasyDGL.py:149-151), the pad logit -1000 belongs to item 0 — never in these chunks
            check(lib.edgl_gemm(_ptr(rows), ctypes.c_void_p(table_c.data_ptr() + lo * C * esz), _ptr(logits), R, hi - lo, C, C, C,
                                hi - lo, 1, 1, ctypes.c_void_p(bias_z.data_ptr() + lo * 4), None,
                                _lib.EPI_BIAS | _lib.EPI_OUT_F32, 1, None, code, st), "edgl_gemm")
        else:
            check(lib.edgl_score_lse_fwd(_ptr(rows), _ptr(table_c), _ptr(out_bias), None, R, C, I, lo, hi, None, None, None,
                                         _ptr(logits), _ptr(ws), code, st), "edgl_score_lse_fwd")  # logits [R, hi - lo], row stride hi - lo
        check(lib.edgl_mask_topk(_ptr(logits), R, hi - lo, lo, _ptr(seen), T, K, _ptr(cval[j]), _ptr(cidx[j]), st), "edgl_mask_topk")
    return _merge_lists(cval, cidx, R, K)


def _merge_lists(cval, cidx, R, K):
    """[S, R, K] candidate lists -> the global top-K by (value desc, index asc): rounds of the merge kernel (<= 1024 candidates each)."""
    dev, st = cval.device, _stream()
    fan = max(2, 1024 // K)                              # the merge kernel takes up to 1024 candidates per row
    while cval.shape[0] > 1:
        n = cval.shape[0]
        ng = (n + fan - 1) // fan
        nval = torch.empty((ng, R, K), device=dev, dtype=torch.float32)
        nidx = torch.empty((ng, R, K), device=dev, dtype=torch.int32)
        for g in range(ng):
            a, b = g * fan, min(n, (g + 1) * fan)
            if b - a == 1:
                nval[g].copy_(cval[a]); nidx[g].copy_(cidx[a])
            else:
                check(lib.edgl_topk_merge(_ptr(cval[a:b]), _ptr(cidx[a:b]), b - a, R, K, _ptr(nval[g]), _ptr(nidx[g]), st),
                      "edgl_topk_merge")
        cval, cidx = nval, nidx
    return cval[0], cidx[0]


def rank_metrics(topk_idx: torch.Tensor, label: torch.Tensor, metrics: torch.Tensor) -> None:
    R, K = topk_idx.shape
    check(lib.edgl_rank_metrics(_ptr(topk_idx), R, K, _ptr(label), _ptr(metrics), _stream()), "edgl_rank_metrics")


def adam_step(param, grad, m, v, lr, state, l2, seg, shadow, beta1=0.9, beta2=0.999, eps=1e-8):
    check(lib.edgl_adam_step(_ptr(param), _ptr(grad), _ptr(m), _ptr(v), param.numel(), float(lr), beta1, beta2, eps,
                             _ptr(state), float(l2), _ptr(seg), 0 if seg is None else seg.numel() // 2, _ptr(shadow),
                             _stream()), "edgl_adam_step")


class DropoutFn(torch.autograd.Function):
    """tf.layers.dropout as a standalone op (FeedForward, Base.py:80,83): counter-based mask, regenerated in the backward."""

    @staticmethod
    def forward(ctx, x, drop: Drop):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.edgl_dropout(_ptr(x), _ptr(y), x.numel(), float(drop.rate), drop.ptr(), drop.stream_id, _code(x), _stream()),
              "edgl_dropout")
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        d = ctx.drop
        check(lib.edgl_dropout(_ptr(dy), _ptr(dx), dy.numel(), float(d.rate), d.ptr(), d.stream_id, _code(dy), _stream()),
              "edgl_dropout")
        return dx, None


def dropout(x, drop: Drop):
    return x if not drop.active else DropoutFn.apply(x, drop)


class AddFn(torch.autograd.Function):
    """out = a + b (edgl_add); the gradient passes to both."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        check(lib.edgl_add(_ptr(a), _ptr(b), _ptr(out), a.numel(), _code(a), _stream()), "edgl_add")
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return AddFn.apply(a, b)


# ------------------------------------------------------------------------------------------------
# config 5 baselines: item embedding alone, row mask, K11 attention with the time feature map
# ------------------------------------------------------------------------------------------------
class EmbedFn(torch.autograd.Function):
    """x = dropout(item[ids] * sqrt(C)) with the zero-padded row 0 (TGAT.py:49-56, TiSASREC.py:52-65): edgl_embed_pos_* with
    no position table."""

    @staticmethod
    def forward(ctx, item_master, item_c, ids, drop: Drop, act_dtype, c_true=0):
        B, T = ids.shape
        I, C = item_c.shape
        x0 = torch.empty((B, T, C), device=ids.device, dtype=act_dtype)
        check(lib.edgl_embed_pos_fwd_ct(_ptr(ids), None, _ptr(item_c), None, None, B, T, C, 0, 1.0, float(drop.rate), drop.ptr(),
                                        drop.stream_id, _ptr(x0), None, None, int(c_true), _DT[act_dtype], _stream()),
              "edgl_embed_pos_fwd")
        ctx.c_true = int(c_true)
        ctx.save_for_backward(ids)
        ctx.meta = (B, T, C, I, drop, item_master.shape)
        return x0

    @staticmethod
    def backward(ctx, dx0):
        (ids,) = ctx.saved_tensors
        B, T, C, I, drop, ishape = ctx.meta
        dx0 = dx0.contiguous()
        d_item = torch.empty(ishape, device=dx0.device, dtype=torch.float32)
        check(lib.edgl_embed_pos_bwd_ct(_ptr(ids), _ptr(dx0), B, T, C, I, float(drop.rate), drop.ptr(), drop.stream_id,
                                        _ptr(d_item), None, ctx.c_true, _code(dx0), _stream()), "edgl_embed_pos_bwd")
        return d_item, None, None, None, None, None


class MaskRowsFn(torch.autograd.Function):
    """y = x * (ids != 0)[..., None]  (`seqs_outs *= seqs_masks`, TGAT.py:70)."""

    @staticmethod
    def forward(ctx, x, ids):
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib.edgl_mask_rows(_ptr(x), _ptr(ids), _ptr(y), ids.numel(), x.shape[-1], _code(x), _stream()), "edgl_mask_rows")
        ctx.save_for_backward(ids)
        return y

    @staticmethod
    def backward(ctx, dy):
        (ids,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib.edgl_mask_rows(_ptr(dy), _ptr(ids), _ptr(dx), ids.numel(), dy.shape[-1], _code(dy), _stream()), "edgl_mask_rows")
        return dx, None


def mask_rows(x, ids):
    return MaskRowsFn.apply(x, ids)


class TfAttnFn(torch.autograd.Function):
    """TfMultiHeadAttention after its dense layers (temporal.py:139-184): q [B,T,C], kv [B,T,2C] (K | V column blocks),
    resid = the queries; time feature map (edgl_timefn_*) + masked causal attention (edgl_tattn_*).  `violations`: device
    int32 counter of positions whose timestamps decrease (see include/easydgl_hip.h)."""

    @staticmethod
    def forward(ctx, q, kv, resid, pos_tab, omega, phi, ids, ts, H, time_scale, drop: Drop, violations, qk_scale=0.0):
        """qk_scale: the score scale (0 = 1 / sqrt(head dim)); a channel-padded model passes 1 / sqrt(true head dim)."""
        B, T, C = q.shape
        dh = C // H
        q, kv, resid = q.contiguous(), kv.contiguous(), resid.contiguous()
        code = _code(q)
        qx = torch.empty((B, T, 3 * C), device=q.device, dtype=q.dtype)
        kx = torch.empty_like(qx)
        check(lib.edgl_timefn_fwd(_ptr(q), C, _ptr(kv), 2 * C, _ptr(pos_tab), _ptr(ts), _ptr(ids), _ptr(omega), _ptr(phi), B, T, C,
                                  H, float(time_scale), _ptr(qx), _ptr(kx), _ptr(violations), code, _stream()), "edgl_timefn_fwd")
        out = torch.empty((B, T, C), device=q.device, dtype=q.dtype)
        need = any(ctx.needs_input_grad)
        saved = torch.empty(int(lib.edgl_tattn_saved_bytes(B, T, H, dh)), device=q.device, dtype=torch.uint8) if need else None
        scale = float(qk_scale) if qk_scale else 1.0 / float(dh) ** 0.5                  # temporal.py:153
        check(lib.edgl_tattn_fwd(_ptr(qx), 3 * C, _ptr(kx), 3 * C, _vptr(kv[:, :, C:]), 2 * C, _ptr(resid), C, _ptr(ids), B, T, H,
                                 3 * dh, dh, scale, float(drop.rate), drop.ptr(), drop.stream_id, _ptr(out), C, _ptr(saved),
                                 _lib.TATTN_CAUSAL, code, _stream()), "edgl_tattn_fwd")
        ctx.save_for_backward(q, kv, qx, kx, omega, phi, ids, ts, saved)
        ctx.meta = (B, T, C, H, dh, scale, float(time_scale), drop, pos_tab.shape)
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, kv, qx, kx, omega, phi, ids, ts, saved = ctx.saved_tensors
        B, T, C, H, dh, scale, time_scale, drop, pshape = ctx.meta
        d_out = d_out.contiguous()
        code = _code(q)
        d_qx, d_kx = torch.empty_like(qx), torch.empty_like(kx)
        d_kv = torch.empty_like(kv)
        check(lib.edgl_tattn_bwd(_ptr(qx), 3 * C, _ptr(kx), 3 * C, _vptr(kv[:, :, C:]), 2 * C, _ptr(ids), _ptr(d_out), C, _ptr(saved),
                                 B, T, H, 3 * dh, dh, scale, float(drop.rate), drop.ptr(), drop.stream_id, _ptr(d_qx), 3 * C,
                                 _ptr(d_kx), 3 * C, _vptr(d_kv[:, :, C:]), 2 * C, _lib.TATTN_CAUSAL, code, _stream()), "edgl_tattn_bwd")
        d_q = torch.empty_like(q)
        both = torch.empty(2 * C, device=q.device, dtype=torch.float32)   # d_omega | d_phi adjacent: one reduction launch
        d_omega, d_phi = both[:C], both[C:]
        ws = torch.empty(int(lib.edgl_timefn_bwd_workspace(C)), device=q.device, dtype=torch.float32)
        check(lib.edgl_timefn_bwd(_ptr(q), C, _ptr(ts), _ptr(omega), _ptr(phi), _ptr(d_qx), _ptr(d_kx), B, T, C, H, time_scale,
                                  _ptr(d_q), C, _ptr(d_kv), 2 * C, _ptr(d_omega), _ptr(d_phi), _ptr(ws), code, _stream()),
              "edgl_timefn_bwd")
        # d(pos_tab)[t] = sum_b dK[b,t]  (K + pos enters the score, temporal.py:143,148): column sums of d_kv seen as [B, T*2C]
        d_pos = colsum(d_kv.view(B, T * 2 * C), B, T * 2 * C).view(T, 2 * C)[:, :C].contiguous()
        if pshape[0] != T:
            full = torch.zeros(pshape, device=q.device, dtype=torch.float32)
            full[:T] = d_pos
            d_pos = full
        return d_q, d_kv, d_out, d_pos, d_omega, d_phi, None, None, None, None, None, None, None


class TiAttnFn(torch.autograd.Function):
    """TiMultiHeadAttention after its dense layers (temporal.py:45-104): q [B,T,C], kv [B,T,2C] (K | V), resid = the queries,
    position tables posK/posV f32 [>=T, C], interval tables (masters f32 + compute-dtype copies) [timelen, C]."""

    @staticmethod
    def forward(ctx, q, kv, resid, posK, posV, ktime, vtime, ktime_c, vtime_c, ids, ts, H, time_scale, timelen, drop: Drop,
                qk_scale=0.0):
        B, T, C = q.shape
        dh = C // H
        q, kv, resid = q.contiguous(), kv.contiguous(), resid.contiguous()
        code = _code(q)
        kvp = torch.empty_like(kv)
        check(lib.edgl_add_pos2(_ptr(kv), _ptr(posK), _ptr(posV), B, T, C, _ptr(kvp), code, _stream()), "edgl_add_pos2")
        out = torch.empty((B, T, C), device=q.device, dtype=q.dtype)
        need = any(ctx.needs_input_grad)
        saved = wbuf = None
        if need:
            saved = torch.empty(int(lib.edgl_tattn_saved_bytes(B, T, H, dh)), device=q.device, dtype=torch.uint8)
            wbuf = torch.empty(int(lib.edgl_tiattn_bucket_elems(B, T, H, timelen)), device=q.device, dtype=q.dtype)
        scale = float(qk_scale) if qk_scale else 1.0 / float(dh) ** 0.5
        rows = ktime_c.shape[0]
        check(lib.edgl_tiattn_fwd(_ptr(q), C, _ptr(kvp), 2 * C, _vptr(kvp[:, :, C:]), 2 * C, _ptr(resid), C, _ptr(ids), _ptr(ts),
                                  _ptr(ktime_c), _ptr(vtime_c), rows, B, T, H, dh, scale, float(time_scale), int(timelen),
                                  float(drop.rate), drop.ptr(), drop.stream_id, _ptr(out), C, _ptr(saved), _ptr(wbuf),
                                  _lib.TATTN_CAUSAL, code, _stream()), "edgl_tiattn_fwd")
        ctx.save_for_backward(q, kvp, ktime_c, vtime_c, ids, ts, saved, wbuf)
        ctx.meta = (B, T, C, H, dh, scale, float(time_scale), int(timelen), drop, posK.shape, posV.shape, rows)
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, kvp, ktime_c, vtime_c, ids, ts, saved, wbuf = ctx.saved_tensors
        B, T, C, H, dh, scale, time_scale, timelen, drop, pk_shape, pv_shape, rows = ctx.meta
        d_out = d_out.contiguous()
        code = _code(q)
        d_q, d_kv = torch.empty_like(q), torch.empty_like(kvp)
        dgbuf = torch.empty_like(wbuf)   # binned score gradients
        d_kt = torch.empty((rows, C), device=q.device, dtype=torch.float32)
        d_vt = torch.empty_like(d_kt)
        check(lib.edgl_tiattn_bwd(_ptr(q), C, _ptr(kvp), 2 * C, _vptr(kvp[:, :, C:]), 2 * C, _ptr(ids), _ptr(ts), _ptr(ktime_c),
                                  _ptr(vtime_c), rows, _ptr(d_out), C, _ptr(saved), _ptr(wbuf), B, T, H, dh, scale, time_scale,
                                  timelen, float(drop.rate), drop.ptr(), drop.stream_id, _ptr(d_q), C, _ptr(d_kv), 2 * C,
                                  _vptr(d_kv[:, :, C:]), 2 * C, _ptr(dgbuf), _ptr(d_kt), _ptr(d_vt), _lib.TATTN_CAUSAL, code,
                                  _stream()), "edgl_tiattn_bwd")
        d_pos = colsum(d_kv.view(B, T * 2 * C), B, T * 2 * C).view(T, 2 * C)     # the tables enter K and V additively
        d_pk = torch.zeros(pk_shape, device=q.device, dtype=torch.float32)
        d_pv = torch.zeros(pv_shape, device=q.device, dtype=torch.float32)
        d_pk[:T] = d_pos[:, :C]
        d_pv[:T] = d_pos[:, C:]
        return d_q, d_kv, d_out, d_pk, d_pv, d_kt, d_vt, None, None, None, None, None, None, None, None, None



class FFTailFn(torch.autograd.Function):
    """FeedForward tail: (dropout(a) + b) * (ids != 0) in one kernel (edgl_ff_tail); ids None = no row mask."""

    @staticmethod
    def forward(ctx, a, b, ids, drop: Drop):
        a, b = a.contiguous(), b.contiguous()
        out = torch.empty_like(a)
        C = a.shape[-1]
        rows = a.numel() // C
        check(lib.edgl_ff_tail(_ptr(a), _ptr(b), _ptr(ids), rows, C, float(drop.rate), drop.ptr(), drop.stream_id, _ptr(out),
                               _code(a), _stream()), "edgl_ff_tail")
        ctx.ids, ctx.drop, ctx.meta = ids, drop, (rows, C)
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        rows, C = ctx.meta
        d, ids = ctx.drop, ctx.ids
        da = torch.empty_like(g)
        check(lib.edgl_ff_tail(_ptr(g), None, _ptr(ids), rows, C, float(d.rate), d.ptr(), d.stream_id, _ptr(da), _code(g),
                               _stream()), "edgl_ff_tail")
        if ids is None:
            db = g
        else:
            db = torch.empty_like(g)
            check(lib.edgl_mask_rows(_ptr(g), _ptr(ids), _ptr(db), rows, C, _code(g), _stream()), "edgl_mask_rows")
        return da, db, None, None


def ff_tail(a, b, ids, drop: Drop):
    return FFTailFn.apply(a, b, ids, drop)


# ------------------------------------------------------------------------------------------------
# operator-level forms of src/module/coding.py (called on their own, outside the fused model kernels)
# ------------------------------------------------------------------------------------------------
class EmbeddingFn(torch.autograd.Function):
    """Embedding.__call__ (coding.py:60-64): scale * table[ids] with the zero-padded row 0 (edgl_embedding_fwd/bwd).
    table_c is the compute copy of the table (the f32 master, or its bf16 shadow); the gradient goes to the master."""

    @staticmethod
    def forward(ctx, table_master, table_c, ids, zero_pad, scale):
        ids = ids.contiguous()
        rows, C = table_c.shape
        out = torch.empty(tuple(ids.shape) + (C,), device=ids.device, dtype=table_c.dtype)
        check(lib.edgl_embedding_fwd(_ptr(ids), ids.numel(), _ptr(table_c), rows, C, int(bool(zero_pad)), float(scale), _ptr(out),
                                     _code(table_c), _stream()), "edgl_embedding_fwd")
        ctx.save_for_backward(ids)
        ctx.meta = (rows, C, int(bool(zero_pad)), float(scale))
        return out

    @staticmethod
    def backward(ctx, d_out):
        (ids,) = ctx.saved_tensors
        rows, C, zero_pad, scale = ctx.meta
        d_out = d_out.contiguous()
        d_table = torch.empty((rows, C), device=d_out.device, dtype=torch.float32)
        check(lib.edgl_embedding_bwd(_ptr(ids), ids.numel(), _ptr(d_out), rows, C, zero_pad, scale, _ptr(d_table), _code(d_out),
                                     _stream()), "edgl_embedding_bwd")
        return d_table, None, None, None, None


def time_sinusoid(x: torch.Tensor, tscale: torch.Tensor, num_units: int, dtype=torch.float32) -> torch.Tensor:
    """TimeSinusoidCoding.code (coding.py:137-149): x [B,T] -> [B,T,C], sin on even / cos on odd channels."""
    x = x.to(torch.float32).contiguous()
    out = torch.empty(tuple(x.shape) + (num_units,), device=x.device, dtype=dtype)
    check(lib.edgl_time_sinusoid(_ptr(x), x.numel(), _ptr(tscale), num_units, _ptr(out), _DT[dtype], _stream()),
          "edgl_time_sinusoid")
    return out


class TimeFunctionFn(torch.autograd.Function):
    """TimeFunctionCoding.code (coding.py:113-122): cos(x[..., None] * basis_freq + phase)."""

    @staticmethod
    def forward(ctx, x, freq, phase, dtype):
        x = x.to(torch.float32).contiguous()
        C = freq.shape[0]
        out = torch.empty(tuple(x.shape) + (C,), device=x.device, dtype=dtype)
        check(lib.edgl_time_function_fwd(_ptr(x), x.numel(), _ptr(freq), _ptr(phase), C, _ptr(out), _DT[dtype], _stream()),
              "edgl_time_function_fwd")
        ctx.save_for_backward(x, freq, phase)
        return out

    @staticmethod
    def backward(ctx, d_out):
        x, freq, phase = ctx.saved_tensors
        C = freq.shape[0]
        d_out = d_out.contiguous()
        both = torch.empty(2 * C, device=x.device, dtype=torch.float32)
        ws = torch.empty(int(lib.edgl_time_function_bwd_workspace(C)), device=x.device, dtype=torch.float32)
        check(lib.edgl_time_function_bwd(_ptr(x), x.numel(), _ptr(freq), _ptr(phase), C, _ptr(d_out), _ptr(both[:C]),
                                         _ptr(both[C:]), _ptr(ws), _code(d_out), _stream()), "edgl_time_function_bwd")
        return None, both[:C], both[C:], None
