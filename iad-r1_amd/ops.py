"""Functional wrappers over the C ABI (iadr1_amd.hip): allocate outputs with torch (device memory is
torch's job, arithmetic is not) and launch on torch's current HIP stream.  2-D tensors are row-major
views; a non-unit inner stride is an error, an outer stride is passed through as `ld`."""
from __future__ import annotations

import ctypes
import os

import numpy as np
import torch

from . import hip

BF16, F32 = torch.bfloat16, torch.float32


def _ld(x: torch.Tensor) -> int:
    assert x.dim() == 2 and x.stride(1) == 1, f"need a row-major 2-D view, got shape {tuple(x.shape)} stride {x.stride()}"
    return x.stride(0)


_FUSE_SWIGLU = os.environ.get("IADR1_FUSE_SWIGLU", "1") != "0"


def h2d(a, device):
    """Host array / CPU tensor -> device tensor through pinned memory, without blocking.  A pageable host-to-device copy synchronises the stream: the host
    would wait for everything it has enqueued so far, lose its run-ahead, and the GPU would idle while it catches up (index plans, targets, advantages are
    uploaded in the middle of enqueued phases).  A/B on the 3B bench step, three alternating pairs: 1272.0 / 1265.6 / 1268.4 ms against 1274.0 / 1270.4 / 1269.6
    with the pageable form."""
    t = torch.from_numpy(a) if not isinstance(a, torch.Tensor) else a
    if torch.device(device).type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def gemm_nt(a, b, bias=None, out=None, out_dtype=BF16, accumulate=False, act=0):
    """out[M,N] (+)= act(a[M,K] @ b[N,K]^T + bias)"""
    M, K = a.shape
    N, K2 = b.shape
    assert K == K2 and a.dtype == BF16 and b.dtype == BF16
    if out is None:
        assert not accumulate
        out = torch.empty(M, N, dtype=out_dtype, device=a.device)
    mode = 0 if out.dtype == BF16 else (2 if accumulate else 1)
    assert not (accumulate and out.dtype != F32)
    ks = splitk_slices(M, N, K) if (accumulate and bias is None and act == 0) else 1
    if ks > 1:
        ws = _splitk_workspace(ks * M * N, a.device)
        hip.call("gemm_nt_splitk_acc_bf16", a, b, out, ws, M, N, K, _ld(a), _ld(b), _ld(out), ks)
        return out
    hip.call("gemm_nt_bf16", a, b, out, bias, M, N, K, _ld(a), _ld(b), _ld(out), mode, act)
    return out


_GEMM_TN = os.environ.get("IADR1_GEMM_TN", "auto")   # 0: never; 1: the split-K shapes only; 2: every shape the 256 x 256 kernel takes; auto: 2 where the caller asks for it
                                                     # (the PA-SFT engine), else 1 -- profiles/r05_gemm_tn.txt: PA-SFT 358.9 / 354.7 / 351.3 ms, SC-GRPO 1205.7 / 1198.9 / 1207.6


def gemm_tn_acc(dy, x, out, wide=False):
    """out[N, K] (fp32) += dy[T, N]^T @ x[T, K] -- a weight gradient from the two operands as the backward holds them (include/iadr1_hip.h iadr1_gemm_tn_acc_bf16: no
    transposed copies) where the 256 x 256 kernel is the one gemm_nt would run on the transposed copies; otherwise (small / ragged shapes, IADR1_GEMM_TN=0) the
    transposes + gemm_nt(accumulate).  Same bits either way.  wide: also the shapes with enough output tiles for an un-split launch (gate|up, lm_head: the TN
    kernel runs at 88 % of the NT kernel's rate, so there it pays only where the transposes are not hidden under another stream's GEMMs)."""
    T, N = dy.shape
    T2, K = x.shape
    assert T == T2 and dy.dtype == BF16 and x.dtype == BF16 and out.dtype == F32 and tuple(out.shape) == (N, K)
    if _GEMM_TN != "0" and N >= 256 and K >= 256 and N % 8 == 0 and K % 8 == 0 and _ld(dy) % 8 == 0 and _ld(x) % 8 == 0 and _ld(out) % 4 == 0:
        Tp = (T + 7) // 8 * 8                      # (what the transposed form contracts over: rows padded to 8)
        ks = splitk_slices(N, K, Tp)
        tiles = ((N + 255) // 256) * ((K + 255) // 256)
        if ks > 1:
            hip.call("gemm_tn_acc_bf16", dy, x, out, _splitk_workspace(ks * N * K, dy.device), N, K, T, _ld(dy), _ld(x), _ld(out), ks)
            return out
        if (_GEMM_TN == "2" or (_GEMM_TN == "auto" and wide)) and N >= 512 and K >= 512 and tiles >= 192:
            hip.call("gemm_tn_acc_bf16", dy, x, out, None, N, K, T, _ld(dy), _ld(x), _ld(out), 1)
            return out
    return gemm_nt(transpose(dy, pad_rows_to=8), transpose(x, pad_rows_to=8), out=out, accumulate=True)


_SPLITK = os.environ.get("IADR1_GEMM_SPLITK", "1") != "0"
_splitk_ws = {}


def splitk_slices(M, N, K) -> int:
    """K slices for an accumulate-mode GEMM (weight gradient) whose 256 x 256 output tiles do not fill 256 CUs twice over: enough slices for >= 512 blocks, each
    at least 1024 deep (16 K tiles: the prologue / epilogue of a tile stay below ~10 %); 1 = the plain kernel.  IADR1_GEMM_SPLITK=0 switches it off (A/B)."""
    if not _SPLITK or M < 512 or N < 512 or N % 4 or K < 4096:
        return 1
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    if tiles >= 448:
        return 1
    ks = min((512 + tiles - 1) // tiles, K // 1024)
    while ks > 1 and (ks - 1) * (((K + ks - 1) // ks + 63) // 64 * 64) >= K:      # no empty last slice
        ks -= 1
    return max(ks, 1)


def _splitk_workspace(n_floats, device):
    """fp32 partial tiles of the split-K weight-gradient GEMMs: one grow-only buffer per (device, stream) -- launches on a stream are ordered, so are its users."""
    key = (str(device), torch.cuda.current_stream().cuda_stream)
    w = _splitk_ws.get(key)
    if w is None or w.numel() < n_floats:
        w = _splitk_ws[key] = torch.empty(n_floats, dtype=F32, device=device)
    return w


def pack_weight(w, out=None):
    """[N,K] row-major -> decode-packed fragment order (flat [N*K])."""
    N, K = w.shape
    o = out if out is not None else torch.empty(N * K, dtype=BF16, device=w.device)
    hip.call("pack_weight_bf16", w, _ld(w), o, N, K)
    return o


def pack_gateup(w, out=None):
    """[2I,K] gate|up -> decode-packed with gate/up tiles interleaved (for gemm_skinny(..., swiglu=True))."""
    N2, K = w.shape
    o = out if out is not None else torch.empty(N2 * K, dtype=BF16, device=w.device)
    hip.call("pack_gateup_bf16", w, _ld(w), o, N2 // 2, K)
    return o


def pack_weight_fp8(w, out=None, out_scale=None, gateup=False):
    """[N,K] bf16 -> (FP8 e4m3 decode pack: flat uint8 [N*K], fp32 scale per output row [N]) for gemm_skinny (include/iadr1_hip.h iadr1_pack_weight_fp8).
    gateup: w is a gate|up matrix, tiles interleaved for the fused-SwiGLU mode."""
    N, K = w.shape
    out = out if out is not None else torch.empty(N * K, dtype=torch.uint8, device=w.device)
    out_scale = out_scale if out_scale is not None else torch.empty(N, dtype=F32, device=w.device)
    assert out.dtype == torch.uint8 and out.numel() == N * K and out_scale.dtype == F32 and out_scale.numel() == N
    hip.call("pack_weight_fp8", w, _ld(w), out, out_scale, N, K, N // 2 if gateup else 0)
    return out, out_scale


class SideOut(ctypes.Structure):
    """include/iadr1_hip.h iadr1_side_out_t: where a decode-step kernel also writes its rows of the training arena.  Built once per (kernel, layer)
    by the rollout (the pointers are static) and handed to the four entry points that take `side`."""
    _fields_ = [("p0", ctypes.c_void_p), ("ld0", ctypes.c_longlong), ("p1", ctypes.c_void_p), ("ld1", ctypes.c_longlong), ("p2", ctypes.c_void_p),
                ("ld2", ctypes.c_longlong), ("step", ctypes.c_void_p), ("base", ctypes.c_longlong), ("seq_stride", ctypes.c_longlong),
                ("mark", ctypes.c_void_p), ("mark_mul", ctypes.c_uint), ("mark_add", ctypes.c_uint)]

    @staticmethod
    def make(step, base, seq_stride, p0=None, p1=None, p2=None, ld1=None, mark=None, mark_mul=0, mark_add=0):
        """p0 / p1 are [rows, width] tensors (ld = row stride) except where the header says otherwise (ld1 overrides, e.g. the [Hq][T] log-sum-exp).
        mark: the decode step's progress word (uint32 / int32 device tensor; rmsnorm_fwd only): *mark = *step * mark_mul + mark_add when the kernel starts."""
        ld = lambda t: 0 if t is None else (t.stride(0) if t.dim() > 1 else 1)
        ptr = lambda t: None if t is None else t.data_ptr()
        so = SideOut(ptr(p0), ld(p0), ptr(p1), ld(p1) if ld1 is None else ld1, ptr(p2), ld(p2), step.data_ptr(), int(base), int(seq_stride),
                     ptr(mark), int(mark_mul), int(mark_add))
        so._keep = (step, p0, p1, p2, mark)      # the struct holds raw device pointers: keep their owners alive with it
        return so


def _side(so):
    return None if so is None else ctypes.addressof(so)


class PackedAct:
    """bf16 activations [M, K] in the decode-packed layout (MFMA B-fragment order, rows padded to 64; C ABI: leading
    dimension 0).  Producers: pack_act, rmsnorm_fwd(out=PackedAct), attn_decode(out=PackedAct), gemm_skinny(swiglu, out=PackedAct)."""

    def __init__(self, M, K, device):
        assert K % 32 == 0
        self.M, self.K = M, K
        self.buf = torch.zeros((M + 63) // 64 * 64 * K, dtype=BF16, device=device)

    @property
    def shape(self):
        return (self.M, self.K)

    def unpack(self):
        Mp = self.buf.numel() // self.K
        return self.buf.view(Mp // 64, self.K // 32, 4, 4, 16, 8).permute(0, 2, 4, 1, 3, 5).reshape(Mp, self.K)[: self.M]


def pack_act(x, out=None):
    M, K = x.shape
    out = out if out is not None else PackedAct(M, K, x.device)
    hip.call("pack_act_bf16", x, _ld(x), out.buf, M, K)
    return out


def _xarg(x):
    return (x.buf, 0) if isinstance(x, PackedAct) else (x, _ld(x))


def gemm_skinny(x, wp, N, bias=None, out=None, out_dtype=BF16, ksplit=1, swiglu=False, side=None):
    """out[M,N] = x[M,K] @ W[N,K]^T + bias with W given decode-packed (`pack_weight`); HBM-bound weight stream.
    x may be a PackedAct.  ksplit > 1: `out` is fp32 [ksplit, M, N] partial slabs to be summed by rmsnorm_fwd(x32=...)."""
    M, K = x.shape
    xb, ldx = _xarg(x)
    dev = xb.device
    if isinstance(wp, tuple):       # (FP8 pack, per-row scales): ops.pack_weight_fp8 -- the opt-in FP8 weight stream of the rollout
        w8, wscale = wp
        assert w8.dtype == torch.uint8 and w8.numel() == N * K and wscale.numel() == N and side is None
        if swiglu:
            out = out if out is not None else torch.empty(M, N // 2, dtype=BF16, device=dev)
            ob, ldo = _xarg(out)
            hip.call("gemm_skinny_fp8w", xb, w8, wscale, ob, None, M, N, K, ldx, ldo, 3, 1)
            return out
        if out is None:
            out = torch.empty((ksplit, M, N) if ksplit > 1 else (M, N), dtype=F32 if ksplit > 1 else out_dtype, device=dev)
        if out.dim() == 3:
            assert out.dtype == F32 and out.shape[0] == ksplit and out.is_contiguous() and bias is None
            mode, ldy = 2, N
        else:
            mode, ldy = (1 if out.dtype == F32 else 0), _ld(out)
        hip.call("gemm_skinny_fp8w", xb, w8, wscale, out, bias, M, N, K, ldx, ldy, mode, ksplit)
        return out
    w = wp
    assert wp.numel() == N * K
    if swiglu:
        if out is None:
            out = torch.empty(M, N // 2, dtype=BF16, device=dev)
        ob, ldo = _xarg(out)
        hip.call("gemm_skinny_bf16", xb, w, ob, None, M, N, K, ldx, K, ldo, 3, 1, _side(side))
        return out
    if out is None:
        out = torch.empty((ksplit, M, N) if ksplit > 1 else (M, N), dtype=F32 if ksplit > 1 else out_dtype, device=dev)
    if out.dim() == 3:
        assert out.dtype == F32 and out.shape[0] == ksplit and out.is_contiguous() and bias is None
        mode, ldy = 2, N
    else:
        mode, ldy = (1 if out.dtype == F32 else 0), _ld(out)
    hip.call("gemm_skinny_bf16", xb, w, out, bias, M, N, K, ldx, K, ldy, mode, ksplit, None)
    return out


def pack_qkv_rope(w, bias, Hq, Hkv, D, out=None, out_bias=None):
    """q|k|v weight [N, K] (+ bias [N]) -> decode-packed with rotary partners sharing a tile (for gemm_qkv_rope_kv)."""
    N, K = w.shape
    out = out if out is not None else torch.empty(N * K, dtype=BF16, device=w.device)
    out_bias = out_bias if out_bias is not None else torch.empty(N, dtype=BF16, device=w.device)
    hip.call("pack_qkv_rope_bf16", w, _ld(w), bias, out, out_bias, Hq, Hkv, D, K)
    return out, out_bias


def gemm_qkv_rope_kv(x, wp, bias_p, q_out, cos, sin, slot, kcache, vcache, Hq, Hkv, D, side=None):
    """Decode step: q_out[:, :Hq*D] = rope(x.Wq^T + bq); K / V rows of the new token appended to the paged cache."""
    M, K = x.shape
    xb, ldx = _xarg(x)
    hip.call("gemm_qkv_rope_kv_bf16", xb, wp, bias_p, q_out, cos, sin, slot, kcache, vcache, M, Hq, Hkv, D, K, ldx, _ld(q_out), _side(side))
    return q_out


def transpose(x, out=None, pad_rows_to=1):
    """out[C, Rp] = x[R, C]^T; Rp = R rounded up to `pad_rows_to` (extra columns zero) so the result can be a
    K-contiguous GEMM operand (K must be a multiple of 8)."""
    R, C = x.shape
    if out is None:
        Rp = (R + pad_rows_to - 1) // pad_rows_to * pad_rows_to
        out = torch.empty(C, Rp, dtype=BF16, device=x.device)
        if Rp != R:
            out[:, R:].zero_()
    hip.call("transpose_bf16", x, _ld(x), out, _ld(out), R, C)
    return out


def rmsnorm_fwd(x, w, eps, res=None, res_out=None, x32=None, xbias=None, want_rstd=False, out=None, side=None):
    """x: bf16 [T,H]  or  x32: fp32 [nsplit,T,H] partial slabs (summed in the kernel)."""
    if x is not None:
        T, H = x.shape
        ldx, nsplit = _ld(x), 0
    else:
        x32 = x32 if x32.dim() == 3 else x32.unsqueeze(0)
        nsplit, T, H = x32.shape
        assert x32.is_contiguous()
        ldx = H
    dev = (x if x is not None else x32).device
    y = out if out is not None else torch.empty(T, H, dtype=BF16, device=dev)
    rstd = torch.empty(T, dtype=F32, device=dev) if want_rstd else None
    ldr = _ld(res) if res is not None else (_ld(res_out) if res_out is not None else H)
    yb, ldy = _xarg(y)
    hip.call("rmsnorm_fwd", x, x32, nsplit, xbias, res, res_out, w, yb, rstd, T, H, ldx, ldr, ldy, float(eps), _side(side))
    return y, rstd


def _ws_floats(nbytes, device):
    """fp32 scratch of the ordered two-stage reductions (norm gains, bias column sums): the grow-only per-(device, stream) buffer of the split-K GEMMs."""
    return _splitk_workspace((int(nbytes) + 3) // 4, device)


def rmsnorm_bwd(dy, x, w, rstd, dres=None, dw=None, out=None):
    T, H = x.shape
    dx = out if out is not None else torch.empty(T, H, dtype=BF16, device=x.device)
    assert _ld(dy) == _ld(x) == _ld(dx) and (dres is None or _ld(dres) == _ld(x))
    ws = _ws_floats(hip.lib().iadr1_rmsnorm_bwd_workspace_bytes(T, H), x.device) if dw is not None else None
    hip.call("rmsnorm_bwd", dy, x, w, rstd, dres, dx, dw, ws, T, H, _ld(x))
    return dx


def rope_(x, cos, sin, nheads, D, backward=False):
    T = x.shape[0]
    assert cos.dtype == F32 and cos.shape == (T, D // 2) and cos.is_contiguous() and sin.is_contiguous()
    hip.call("rope_inplace", x, _ld(x), cos, sin, T, nheads, D, 1 if backward else 0)
    return x


def gemm_swiglu_fused(x, w_gu, gu, a):
    """The single-launch form (include/iadr1_hip.h iadr1_gemm_swiglu_bf16); gu may be None.  Shape requirements are the caller's (gemm_swiglu)."""
    T, K = x.shape
    I = w_gu.shape[0] // 2
    hip.call("gemm_swiglu_bf16", x, w_gu, gu, a, T, I, K, _ld(x), _ld(w_gu), _ld(gu) if gu is not None else 0, _ld(a))


def gemm_swiglu_rows(x, w_gu, gu, a, n_blocks: int, block: int, block_stride: int):
    """include/iadr1_hip.h iadr1_gemm_swiglu_rows_bf16: the fused gate|up + SwiGLU contraction over n_blocks row blocks of `block` rows, `block_stride` rows apart,
    of x / gu / a (all addressed from their first row; gu may be None)."""
    K = x.shape[1]
    I = w_gu.shape[0] // 2
    hip.call("gemm_swiglu_rows_bf16", x, w_gu, gu, a, n_blocks * block, I, K, _ld(x), _ld(w_gu), _ld(gu) if gu is not None else 0, _ld(a), block, block_stride)


def gemm_swiglu(x, w_gu, gu_out=None, a_out=None, keep_gu=True):
    """(gu, a) with gu = x @ w_gu^T ([T, 2I], None when keep_gu is False and the fused launch ran) and a = swiglu(gu) ([T, I]).  One launch when the
    shape allows (T % 256 == 0, I % 128 == 0, enough tiles for the 256x256 kernel), gemm_nt + swiglu_fwd otherwise -- same bits either way."""
    T, K = x.shape
    I = w_gu.shape[0] // 2
    a = a_out if a_out is not None else torch.empty(T, I, dtype=BF16, device=x.device)
    if _FUSE_SWIGLU and T % 256 == 0 and I % 128 == 0 and (T // 256) * (I // 128) >= 192 and _ld(x) % 8 == 0:
        gu = (gu_out if gu_out is not None else torch.empty(T, 2 * I, dtype=BF16, device=x.device)) if keep_gu else None
        gemm_swiglu_fused(x, w_gu, gu, a)
        return gu, a
    gu = gemm_nt(x, w_gu, out=gu_out)
    return gu, swiglu_fwd(gu, out=a)


def swiglu_fwd(gu, out=None):
    T, I2 = gu.shape
    a = out if out is not None else torch.empty(T, I2 // 2, dtype=BF16, device=gu.device)
    hip.call("swiglu_fwd", gu, _ld(gu), a, _ld(a), T, I2 // 2)
    return a


def swiglu_bwd(da, gu, out=None):
    T, I2 = gu.shape
    dgu = out if out is not None else torch.empty(T, I2, dtype=BF16, device=gu.device)
    hip.call("swiglu_bwd", da, _ld(da), gu, _ld(gu), dgu, _ld(dgu), T, I2 // 2)
    return dgu


def gelu_fwd(z):
    a = torch.empty_like(z)
    hip.call("gelu_fwd", z, a, z.numel())
    return a


def gelu_bwd(da, z):
    dz = torch.empty_like(z)
    hip.call("gelu_bwd", da, z, dz, z.numel())
    return dz


def quick_gelu_fwd(z):
    a = torch.empty_like(z)
    hip.call("quick_gelu_fwd", z, a, z.numel())
    return a


def quick_gelu_bwd(da, z):
    dz = torch.empty_like(z)
    hip.call("quick_gelu_bwd", da, z, dz, z.numel())
    return dz


def layernorm_fwd(x, w, b, eps, res=None, res_out=None, want_stats=False):
    T, H = x.shape
    y = torch.empty(T, H, dtype=BF16, device=x.device)
    mean = torch.empty(T, dtype=F32, device=x.device) if want_stats else None
    rstd = torch.empty(T, dtype=F32, device=x.device) if want_stats else None
    ldr = _ld(res) if res is not None else H
    hip.call("layernorm_fwd", x, res, res_out, w, b, y, mean, rstd, T, H, _ld(x), ldr, _ld(y), float(eps))
    return y, mean, rstd


def layernorm_bwd(dy, x, w, mean, rstd, dres=None, dw=None, db=None):
    T, H = x.shape
    dx = torch.empty(T, H, dtype=BF16, device=x.device)
    ws = _ws_floats(hip.lib().iadr1_layernorm_bwd_workspace_bytes(T, H), x.device) if dw is not None else None
    hip.call("layernorm_bwd", dy, x, w, mean, rstd, dres, dx, dw, db, ws, T, H, _ld(x))
    return dx


def colsum_acc(dy, out32):
    T, N = dy.shape
    hip.call("colsum_acc", dy, _ld(dy), out32, _ws_floats(hip.lib().iadr1_colsum_workspace_bytes(T, N), dy.device), T, N)
    return out32


def embed_fwd(ids, img_index, E, img, out=None):
    T = ids.numel()
    H = E.shape[1]
    o = out if out is not None else torch.empty(T, H, dtype=BF16, device=E.device)
    hip.call("embed_fwd", ids, img_index, E, img, o, T, H)
    return o


def scatter_plan(dst_rows, device):
    """Host side of rows_scatter_acc: dst_rows[t] >= 0 = destination row of source row t (< 0: dropped) -> (rows int64, ptr int32, idx int32, U) with the device
    arrays padded to the number of source rows: the distinct destination rows in ascending order, and for each the source rows that feed it, ascending (np.unique /
    stable argsort: integer work).  Padded, because the number of DISTINCT token ids changes from step to step: arrays of a fixed size per batch shape come out of the
    caching allocator's free list instead of asking the device for a new block now and then (no allocator traffic in the step)."""
    import numpy as np
    d = np.asarray(dst_rows).reshape(-1)
    src = np.flatnonzero(d >= 0)
    if src.size == 0:
        return None
    order = np.argsort(d[src], kind="stable")
    rows, counts = np.unique(d[src], return_counts=True)
    T, U = d.size, len(rows)
    idx = np.zeros(T, dtype=np.int32)
    idx[: src.size] = src[order]
    rows_p = np.zeros(T, dtype=np.int64)
    rows_p[:U] = rows
    ptr = np.full(T + 1, src.size, dtype=np.int32)
    ptr[0] = 0
    np.cumsum(counts, out=ptr[1: U + 1])
    return h2d(rows_p, device), h2d(ptr, device), h2d(idx, device), U


def rows_scatter_acc(src, plan, dst32):
    """dst32[rows[u]] += sum of src[idx[k]] over the CSR list of u, in list order (include/iadr1_hip.h iadr1_rows_scatter_acc): deterministic scatter-add."""
    if plan is None:
        return dst32
    rows, ptr, idx, U = plan
    T, H = src.shape
    assert src.is_contiguous() and dst32.dtype == F32 and dst32.is_contiguous() and dst32.shape[-1] == H
    hip.call("rows_scatter_acc", src, rows, ptr, idx, dst32, U, H)
    return dst32


def rows_gather_sum(src, ptr, idx, T, weights=None):
    """out[t] = sum_{k in ptr[t]..ptr[t+1]} weights[k] * src[idx[k]] (include/iadr1_hip.h iadr1_rows_gather_sum)."""
    H = src.shape[1]
    assert src.is_contiguous() and ptr.dtype == torch.int32 and idx.dtype == torch.int32 and ptr.numel() == T + 1
    assert weights is None or (weights.dtype == F32 and weights.numel() == idx.numel())
    out = torch.empty(T, H, dtype=BF16, device=src.device)
    hip.call("rows_gather_sum", src, ptr, idx, weights, out, T, H)
    return out


def gelu_tanh_fwd(z):
    a = torch.empty_like(z)
    hip.call("gelu_tanh_fwd", z, a, z.numel())
    return a


def gelu_tanh_bwd(da, z):
    dz = torch.empty_like(z)
    hip.call("gelu_tanh_bwd", da, z, dz, z.numel())
    return dz


def cast_f32_to_bf16(x, cpad=None):
    R, C = x.shape
    cpad = cpad or C
    out = torch.empty(R, cpad, dtype=BF16, device=x.device)
    hip.call("cast_f32_to_bf16", x, _ld(x), out, _ld(out), R, C, cpad)
    return out


def f32_bias_to_bf16(x32, bias, out=None):
    R, C = x32.shape
    o = out if out is not None else torch.empty(R, C, dtype=BF16, device=x32.device)
    hip.call("f32_bias_to_bf16", x32, bias, o, R, C)
    return o


class Segments:
    """Explicit attention segments on the flat token axis (device int32 arrays)."""

    def __init__(self, starts, ends, device, prefix=None):
        """prefix: optional [nseg][4] rows {prefix_start, prefix_len, child_first, child_count} (shared-prefix attention,
        include/iadr1_hip.h iadr1_attn_fwd); segments must then be non-empty."""
        self.n = len(starts)
        self.max_len = max(e - s for s, e in zip(starts, ends)) if self.n else 0
        # rows the (disjoint) segments hold and the range they span: callers skip zero-filling outputs when every row belongs to a segment
        self.rows = int(sum(e - s for s, e in zip(starts, ends)))
        self.lo, self.hi = (int(min(starts)), int(max(ends))) if self.n else (0, 0)
        # (through pinned memory, h2d above: a tensor built from a Python list directly on the device is a pageable copy -- the host would wait for everything enqueued
        # so far; measured on the co-scheduled 3B step: text_plan_shared blocked for the whole 24 ms tail of the shadow pass)
        self.start = h2d(np.asarray(starts, dtype=np.int32).reshape(-1), device)
        self.end = h2d(np.asarray(ends, dtype=np.int32).reshape(-1), device)
        self.start_host, self.end_host = [int(z) for z in starts], [int(z) for z in ends]      # (host copies: reading the device arrays back would drain the stream)
        self.prefix = None
        # launch hint (include/iadr1_hip.h nseg_head / max_seqlen_tail): a leading run of long segments followed by shorter ones
        self.n_head, self.max_tail = 0, 0
        if prefix is not None:
            assert len(prefix) == self.n and all(e > s for s, e in zip(starts, ends)), "shared-prefix segments must be non-empty"
            self.prefix = h2d(np.ascontiguousarray(np.asarray(prefix, dtype=np.int32).reshape(self.n, 4)), device)
            parents = [i for i, pr in enumerate(prefix) if pr[3] > 0]
            nh = (max(parents) + 1) if parents else 0
            if 0 < nh < self.n and parents == list(range(nh)):
                self.n_head, self.max_tail = nh, max(e - s for s, e in zip(starts[nh:], ends[nh:]))

    def covers(self, r0, r1):
        """True when every row of [r0, r1) lies in a segment (no left / post-EOS padding rows)."""
        return self.rows == r1 - r0 and self.lo >= r0 and self.hi <= r1

    @staticmethod
    def from_cu(cu, device):
        return Segments(list(cu[:-1]), list(cu[1:]), device)


def attn_fwd(q, k, v, seg: Segments, Hq, Hkv, D, causal, scale, out=None, want_lse=True):
    T = q.shape[0]
    o = out if out is not None else torch.zeros(T, Hq * D, dtype=BF16, device=q.device)
    lse = torch.empty(Hq, T, dtype=F32, device=q.device) if want_lse else None
    hip.call("attn_fwd", q, k, v, o, lse, seg.start, seg.end, seg.prefix, seg.n, seg.max_len, seg.n_head, seg.max_tail, T, Hq, Hkv, D, _ld(q), _ld(k), _ld(v), _ld(o), 1 if causal else 0, float(scale))
    return o, lse


_DKV_WS = {}


def _dkv_workspace(n, device):
    t = _DKV_WS.get(device)
    if t is None or t.numel() < n:
        _DKV_WS[device] = t = torch.empty(n, dtype=F32, device=device)
    return t


def attn_bwd(q, k, v, o, dout, lse, seg: Segments, Hq, Hkv, D, causal, scale, dq, dk, dv):
    T = q.shape[0]
    delta = torch.empty(Hq, T, dtype=F32, device=q.device)
    group = Hq // Hkv
    hs = 1
    if seg.prefix is not None:   # shared-prefix segments: split the long query loops of the prefix blocks over the q heads of the group
        hs = 4 if group % 4 == 0 else (group if 1 < group <= 8 else 1)
    ws = _dkv_workspace(hs * T * Hkv * 2 * D, q.device) if hs > 1 else None
    hip.call("attn_bwd", q, k, v, o, dout, lse, delta, dq, dk, dv, ws, hs, seg.start, seg.end, seg.prefix, seg.n, seg.max_len, seg.n_head, seg.max_tail, T, Hq, Hkv, D,
             _ld(q), _ld(k), _ld(v), _ld(o), _ld(dout), _ld(dq), _ld(dk), _ld(dv), 1 if causal else 0, float(scale))


def attn_decode(q, kcache, vcache, block_table, ctx_len, Hq, Hkv, D, scale, out=None, side=None, seqs_per_group=0):
    """seqs_per_group: sequences [g*n, (g+1)*n) share prompt pages (placement hint: one XCD per prompt group; same results)."""
    B = q.shape[0]
    o = out if out is not None else torch.empty(B, Hq * D, dtype=BF16, device=q.device)
    ob, ldo = _xarg(o)
    hip.call("attn_decode", q, kcache, vcache, block_table, ctx_len, ob, B, Hq, Hkv, D, block_table.shape[1], _ld(q), ldo, float(scale), int(seqs_per_group), _side(side))
    return o


def attn_decode_group_ws(B, G, Hq, Hkv, D, chunks, device):
    """fp32 workspace of attn_decode_group(chunks > 1)."""
    rows = G * (Hq // Hkv)
    rmax = 16 if rows <= 16 else (32 if rows <= 32 else 64)
    return torch.empty((B // G) * Hkv * chunks * rmax * (D + 2), dtype=F32, device=device)


def attn_decode_group(q, kcache, vcache, block_table, ctx_len, shared_pages, G, Hq, Hkv, D, scale, out=None, side=None, chunks=1, ws=None):
    """Decode attention with the full prompt pages of every group of G sequences read once (include/iadr1_hip.h iadr1_attn_decode_group); chunks > 1 splits
    the shared pages over that many blocks per (group, kv head) (two launches, partial states through `ws`)."""
    B = q.shape[0]
    o = out if out is not None else torch.empty(B, Hq * D, dtype=BF16, device=q.device)
    ob, ldo = _xarg(o)
    assert shared_pages.dtype == torch.int32 and shared_pages.numel() == B // G
    if chunks > 1 and ws is None:
        ws = attn_decode_group_ws(B, G, Hq, Hkv, D, chunks, q.device)
    hip.call("attn_decode_group", q, kcache, vcache, block_table, ctx_len, shared_pages, ob, B, G, Hq, Hkv, D, block_table.shape[1], _ld(q), ldo, float(scale), int(chunks), ws,
             _side(side))
    return o


def kv_store(k, v, slot, kcache, vcache, Hkv, D):
    hip.call("kv_store", k, _ld(k), v, _ld(v), slot, kcache, vcache, k.shape[0], Hkv, D)


def logprob_rows(logits32, targets, want_lse=True):
    R, V = logits32.shape
    logp = torch.empty(R, dtype=F32, device=logits32.device)
    lse = torch.empty(R, dtype=F32, device=logits32.device) if want_lse else None
    hip.call("logprob_rows", logits32, _ld(logits32), targets, logp, lse, R, V)
    return logp, lse


def dlogits_rows(logits32, targets, lse, g, out=None):
    R, V = logits32.shape
    dl = out if out is not None else torch.empty(R, V, dtype=BF16, device=logits32.device)
    hip.call("dlogits_rows", logits32, _ld(logits32), targets, lse, g, dl, _ld(dl), R, V)
    return dl


def linear_logprob(h, w, targets, logp=None, lse=None, ws=None):
    """logp[r] = log_softmax(h[r] @ w^T)[targets[r]] (0 for targets < 0) and lse[r], without materialising the [R, V] logits
    (include/iadr1_hip.h iadr1_linear_logprob_fwd).  ws: uint8 workspace of linear_logprob_ws_bytes(R, V) bytes (allocated when None)."""
    R, K = h.shape
    V, K2 = w.shape
    assert K == K2 and h.dtype == BF16 and w.dtype == BF16 and targets.dtype == torch.int64 and targets.numel() == R
    logp = logp if logp is not None else torch.empty(R, dtype=F32, device=h.device)
    lse = lse if lse is not None else torch.empty(R, dtype=F32, device=h.device)
    need = linear_logprob_ws_bytes(R, V)
    if ws is None:
        ws = torch.empty(need, dtype=torch.uint8, device=h.device)
    assert ws.numel() * ws.element_size() >= need and ws.is_contiguous()
    hip.call("linear_logprob_fwd", h, w, targets, logp, lse, ws, R, V, K, _ld(h), _ld(w))
    return logp, lse


def linear_logprob_ws_bytes(R, V):
    return int(hip.lib().iadr1_linear_logprob_workspace_bytes(R, V))


def linear_logprob_dlogits(h, w, targets, lse, g, out=None):
    """dl[R, V] (bf16) = g[r] * (onehot(targets[r]) - softmax(h[r] @ w^T)) from recomputed logits (iadr1_linear_logprob_dlogits)."""
    R, K = h.shape
    V = w.shape[0]
    dl = out if out is not None else torch.empty(R, V, dtype=BF16, device=h.device)
    hip.call("linear_logprob_dlogits", h, w, targets, lse, g, dl, _ld(dl), R, V, K, _ld(h), _ld(w))
    return dl


def grpo_loss(logp, ref_logp, adv, mask, beta, n_total_rows=None):
    N, C = logp.shape
    dev = logp.device
    dlogp = torch.empty(N, C, dtype=F32, device=dev)
    kl = torch.empty(N, C, dtype=F32, device=dev)
    row_loss = torch.empty(N, dtype=F32, device=dev)
    row_kl = torch.empty(N, dtype=F32, device=dev)
    hip.call("grpo_loss", logp, ref_logp, adv, mask, float(beta), int(n_total_rows or N), dlogp, kl, row_loss, row_kl, N, C)
    return dlogp, kl, row_loss, row_kl


_sample_ws = {}


def sample(logits32, temperature, top_k, top_p, seed, step, suppress_token=-1, step_ptr=None, out=None, seed_ptr=None):
    B, V = logits32.shape
    o = out if out is not None else torch.empty(B, dtype=torch.int64, device=logits32.device)
    key = (B, logits32.device)
    ws = _sample_ws.get(key)
    if ws is None:
        ws = torch.empty(hip.lib().iadr1_sample_workspace_bytes(B), dtype=torch.uint8, device=logits32.device)
        _sample_ws[key] = ws
    hip.call("sample_topk_topp", logits32, _ld(logits32), o, ws, B, V, float(temperature), int(top_k), float(top_p), int(suppress_token), int(seed), int(step), step_ptr, seed_ptr)
    return o


def rope_kv_store(qkv, cos, sin, slot, kcache, vcache, Hq, Hkv, D):
    hip.call("rope_kv_store", qkv, _ld(qkv), cos, sin, slot, kcache, vcache, qkv.shape[0], Hq, Hkv, D)


def rope_table(pos, inv_freq, cos, sin):
    B, half = cos.shape
    hip.call("rope_table", pos, inv_freq, cos, sin, B, half)


def decode_advance(sampled, cur_tok, out_tokens, pos, ctx_len, slot, block_table, finished, step, eos, pad, all_done=None, inv_freq=None, cos=None, sin=None):
    """all_done: int32[1] <- every sequence finished; inv_freq + cos / sin [B, half]: also write the rotary table of the bumped positions (the next step's)."""
    B, C = out_tokens.shape
    hip.call("decode_advance", sampled, cur_tok, out_tokens, C, pos, ctx_len, slot, block_table, block_table.shape[1], finished, step, int(eos), int(pad), B,
             all_done, inv_freq, cos if inv_freq is not None else None, sin if inv_freq is not None else None, cos.shape[1] if inv_freq is not None else 0)
