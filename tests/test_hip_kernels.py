"""Per-kernel parity of the HIP library (through the C ABI) against plain torch fp32 on the same
bf16-rounded inputs.  GPU only."""
import math

import numpy as np
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

import iadr1_amd  # noqa: E402,F401
from iadr1_amd import ops  # noqa: E402

DEV = "cuda"
BF, F32 = torch.bfloat16, torch.float32


def rnd(*shape, scale=1.0, seed=0, dtype=BF):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(DEV)


def close(got, ref, rtol, atol, what=""):
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        i = int(torch.argmax(err - tol))
        idx = np.unravel_index(i, tuple(ref.shape)) if ref.dim() else ()
        raise AssertionError(f"{what}: {int(bad.sum())}/{ref.numel()} out of tol; worst at {idx}: got {got.flatten()[i].item():.6g} ref {ref.flatten()[i].item():.6g} (max|err| {err.max().item():.4g}, max|ref| {ref.abs().max().item():.4g})")


# ------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 192), (100, 200, 72), (1000, 520, 1216), (333, 324, 160), (64, 1280, 3456), (4096, 2048, 2048), (4096, 4096, 1024), (3000, 5000, 520)])
@pytest.mark.parametrize("mode", ["bf16", "bf16_bias", "bf16_bias_gelu", "f32", "f32_acc"])
def test_gemm_nt(M, N, K, mode):
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.5)
    bias = rnd(N, seed=3) if "bias" in mode else None
    ref = a.float() @ b.float().t()
    if bias is not None:
        ref = ref + bias.float()
    if "gelu" in mode:
        ref = torch.nn.functional.gelu(ref)
    if mode.startswith("bf16"):
        out = ops.gemm_nt(a, b, bias=bias, act=1 if "gelu" in mode else 0)
        close(out, ref, 1e-2, 1e-2 * math.sqrt(K) * 0.5, f"gemm {mode} {M}x{N}x{K}")
    elif mode == "f32":
        out = ops.gemm_nt(a, b, out_dtype=F32)
        close(out, ref, 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"gemm f32 {M}x{N}x{K}")
    else:
        base = rnd(M, N, seed=4, dtype=F32)
        out = base.clone()
        ops.gemm_nt(a, b, out=out, accumulate=True)
        close(out, ref + base.float(), 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"gemm f32 acc {M}x{N}x{K}")


@pytest.mark.parametrize("R,V,K", [(20, 1000, 64), (300, 152064 // 8, 256), (513, 4096 + 128, 128), (1024, 151936, 512), (4096, 32064, 256)])
def test_linear_logprob_matches_gemm_then_logprob_rows(R, V, K):
    """iadr1_linear_logprob_fwd / _dlogits (logits never stored) against the two-step form they replace (gemm_nt fp32 logits -> logprob_rows / dlogits_rows)
    and against fp32 torch log_softmax; ragged row / vocabulary tiles, ignored rows (target -100), targets in the first and the last column."""
    h, w = rnd(R, K, seed=1), rnd(V, K, seed=2, scale=0.6)
    gen = torch.Generator(device="cpu").manual_seed(3)
    tg = torch.randint(0, V, (R,), generator=gen)
    tg[0], tg[1], tg[R // 2] = 0, V - 1, -100
    tg = tg.to(DEV)
    g = rnd(R, seed=4, dtype=F32)
    g[R // 2] = 0.0
    logits = ops.gemm_nt(h, w, out_dtype=F32)
    lp0, lse0 = ops.logprob_rows(logits, tg)
    lp, lse = ops.linear_logprob(h, w, tg)
    close(lse, lse0, 1e-6, 2e-5, f"linear_logprob lse {R}x{V}x{K}")
    close(lp, lp0, 1e-6, 2e-5, f"linear_logprob logp {R}x{V}x{K}")
    assert float(lp[R // 2]) == 0.0
    ref = torch.log_softmax(h.float() @ w.float().t(), -1)
    keep = tg >= 0
    close(lp[keep], ref[keep].gather(1, tg[keep][:, None])[:, 0], 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"linear_logprob vs torch {R}x{V}x{K}")
    lp2, lse2 = ops.linear_logprob(h, w, tg)      # deterministic
    assert torch.equal(lp, lp2) and torch.equal(lse, lse2)
    if V % 8 == 0:
        dl0 = ops.dlogits_rows(logits, tg, lse0, g)
        dl = ops.linear_logprob_dlogits(h, w, tg, lse0, g)
        assert torch.equal(dl, dl0), f"dlogits {R}x{V}x{K}: {(dl.float() - dl0.float()).abs().max().item()}"


@pytest.mark.parametrize("M,N,K", [(2048, 2048, 20480), (2560, 2048, 20480), (2048, 11008, 20480), (6912, 1280, 8192), (1280, 1280, 8192), (1300, 516, 4096), (3840, 1280, 8200)])
def test_gemm_nt_splitk_accumulate(M, N, K):
    """Weight-gradient shapes with few 256 x 256 output tiles run as split-K (iadr1_gemm_nt_splitk_acc_bf16: K slices -> fp32 partial tiles -> reduce launch).
    Against fp32 torch, against the un-split kernel (fp32 summation order differs, nothing else), run-to-run bit-stable; ragged tiles, a K that is not a
    multiple of the slice (2560 x 2048 x 20480 -> 7 slices of 2944), strided C."""
    ks = ops.splitk_slices(M, N, K)
    assert ks >= 2, (M, N, K, ks)
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.5)
    ref = a.float() @ b.float().t()
    base = rnd(M, N + 8, seed=4, dtype=F32)
    out = base.clone()
    ops.gemm_nt(a, b, out=out[:, 4: 4 + N], accumulate=True)
    tol = 1e-3 * math.sqrt(K) * 0.1
    close(out[:, 4: 4 + N], ref + base[:, 4: 4 + N], 1e-4, tol, f"gemm split-K {ks} acc {M}x{N}x{K}")
    assert torch.equal(out[:, :4], base[:, :4]) and torch.equal(out[:, 4 + N:], base[:, 4 + N:])
    out2 = base.clone()
    ops.gemm_nt(a, b, out=out2[:, 4: 4 + N], accumulate=True)
    assert torch.equal(out, out2)                                           # no atomics: bit-reproducible
    plain = base.clone()
    ops.hip.call("gemm_nt_bf16", a, b, plain[:, 4: 4 + N], None, M, N, K, K, K, N + 8, 2, 0)
    close(out[:, 4: 4 + N], plain[:, 4: 4 + N], 1e-5, 1e-4 * math.sqrt(K) * 0.1, "split-K vs plain kernel")
    assert ops.splitk_slices(22016, 2048, 20480) == 1 and ops.splitk_slices(2048, 2048, 2048) == 1 and ops.splitk_slices(256, 2048, 20480) == 1


def test_gemm_nt_strided_views():
    # operands / outputs that are column slices of wider buffers (the fused qkv / gate|up layouts)
    big_a, big_b = rnd(300, 512, seed=5), rnd(260, 512, seed=6)
    a, b = big_a[:, 128:384], big_b[:, 64:320]
    outbuf = torch.zeros(300, 520, dtype=BF, device=DEV)
    ops.gemm_nt(a, b, out=outbuf[:, 256:516])
    close(outbuf[:, 256:516], a.float() @ b.float().t(), 1e-2, 0.1, "gemm strided")
    assert float(outbuf[:, :256].abs().sum()) == 0 and float(outbuf[:, 516:].abs().sum()) == 0


@pytest.mark.parametrize("M,N,K", [(64, 2048, 2048), (64, 2560, 2048), (64, 640, 256), (8, 256, 512), (100, 1008, 1056), (64, 22016, 2048), (64, 2048, 11008),
                                   (64, 3584, 18944), (24, 3584, 18944)])          # the 7B down projection: 8 K slices of 74 steps -> persistent split kernel <8, 10>
def test_gemm_skinny(M, N, K):
    x, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(N, seed=3)
    wp = ops.pack_weight(w)
    ref = x.float() @ w.float().t()
    y = ops.gemm_skinny(x, wp, N, out_dtype=F32)
    close(y, ref, 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"skinny f32 {M}x{N}x{K}")
    yb = ops.gemm_skinny(x, wp, N, bias=bias)
    close(yb, ref + bias.float(), 1e-2, 1e-2 * math.sqrt(K) * 0.3, f"skinny bf16+bias {M}x{N}x{K}")
    for ks in (1, 2, 8):
        part = ops.gemm_skinny(x, wp, N, out=torch.full((ks, M, N), 7.0, dtype=F32, device=DEV), ksplit=ks)
        close(part.sum(0), ref, 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"skinny split-K {ks} {M}x{N}x{K}")


@pytest.mark.parametrize("M,I,K", [(64, 11008, 2048), (64, 512, 256), (7, 192, 96),
                                   (64, 18944, 3584), (24, 18944, 3584), (64, 14336, 4096)])     # 7B / Mistral widths: the one-shot 64-column kernel
def test_gemm_skinny_fused_swiglu(M, I, K):
    x, w = rnd(M, K, seed=1), rnd(2 * I, K, seed=2, scale=0.3)
    a = ops.gemm_skinny(x, ops.pack_gateup(w), 2 * I, swiglu=True)
    gu = (x.float() @ w.float().t()).to(BF)
    ref = torch.nn.functional.silu(gu[:, :I].float()).to(BF).float() * gu[:, I:].float()
    close(a, ref, 2e-2, 2e-2 * math.sqrt(K) * 0.3, f"skinny swiglu {M}x{I}x{K}")


@pytest.mark.parametrize("M,N,K", [(64, 2048, 2048), (64, 2560, 2048), (8, 256, 512), (100, 1008, 1056), (64, 22016, 2048), (64, 2048, 11008), (64, 3584, 18944)])
def test_gemm_skinny_decode_packed_x(M, N, K):
    """X in the decode-packed layout (C ABI: ldx == 0) gives the same numbers as row-major X: bit-exact where the K split
    over waves is the same, fp32-rounding-close otherwise."""
    x, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(N, seed=3)
    wp = ops.pack_weight(w)
    xp = ops.pack_act(x)
    assert torch.equal(xp.unpack(), x)
    assert xp.buf.numel() == (M + 63) // 64 * 64 * K
    y0, y1 = ops.gemm_skinny(x, wp, N, out_dtype=F32), ops.gemm_skinny(xp, wp, N, out_dtype=F32)
    close(y1, y0, 1e-5, 1e-4, "packed X f32")
    close(ops.gemm_skinny(xp, wp, N, bias=bias), ops.gemm_skinny(x, wp, N, bias=bias), 1e-2, 1e-2, "packed X bf16+bias")
    for ks in (2, 8):
        p0 = ops.gemm_skinny(x, wp, N, out=torch.zeros((ks, M, N), dtype=F32, device=DEV), ksplit=ks)
        p1 = ops.gemm_skinny(xp, wp, N, out=torch.zeros((ks, M, N), dtype=F32, device=DEV), ksplit=ks)
        close(p1.sum(0), p0.sum(0), 1e-5, 1e-4, f"packed X split-K {ks}")


@pytest.mark.parametrize("M,I,K", [(64, 11008, 2048), (64, 512, 256), (7, 192, 96), (64, 18944, 3584)])
def test_decode_packed_producers(M, I, K):
    """RMSNorm and the fused-SwiGLU epilogue write the decode-packed layout directly (ldy == 0): same values as row-major."""
    x, w = rnd(M, K, seed=1), rnd(2 * I, K, seed=2, scale=0.3)
    wpk = ops.pack_gateup(w)
    a0 = ops.gemm_skinny(x, wpk, 2 * I, swiglu=True)
    a1 = ops.gemm_skinny(ops.pack_act(x), wpk, 2 * I, swiglu=True, out=ops.PackedAct(M, I, DEV))
    close(a1.unpack(), a0, 1e-2, 1e-2, "swiglu packed out")
    g, res = (1 + 0.1 * rnd(K, seed=4).float()).to(BF), rnd(M, K, seed=5)
    y0, _ = ops.rmsnorm_fwd(x, g, 1e-6, res=res, res_out=torch.empty_like(x))
    y1, _ = ops.rmsnorm_fwd(x, g, 1e-6, res=res, res_out=torch.empty_like(x), out=ops.PackedAct(M, K, DEV))
    assert torch.equal(y1.unpack(), y0)


@pytest.mark.parametrize("R,C", [(64, 64), (100, 200), (4096, 2560), (37, 8)])
def test_transpose(R, C):
    x = rnd(R, ((C + 7) // 8) * 8, seed=3)[:, :C]
    out = ops.transpose(x)
    assert torch.equal(out, x.t().contiguous())


# ------------------------------------------------------------------------------------------------ norm
def _rms_ref(x, w, eps):
    xf = x.float()
    n = (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).to(BF)
    return (w * n).to(BF)


@pytest.mark.parametrize("T,H", [(5, 256), (300, 2048), (130, 1280), (64, 160), (33, 3584)])
def test_rmsnorm_fwd_bwd(T, H):
    x, w, res = rnd(T, H, seed=1), (1 + 0.1 * rnd(H, seed=2).float()).to(BF), rnd(T, H, seed=3)
    y, rstd = ops.rmsnorm_fwd(x, w, 1e-6, want_rstd=True)
    assert torch.equal(y, _rms_ref(x, w, 1e-6)) or (y.float() - _rms_ref(x, w, 1e-6).float()).abs().max() <= 2 ** -6 * y.float().abs().max()
    close(rstd, torch.rsqrt(x.float().pow(2).mean(-1) + 1e-6), 1e-5, 1e-6, "rstd")
    # fused residual
    ro = torch.empty_like(x)
    y2, _ = ops.rmsnorm_fwd(x, w, 1e-6, res=res, res_out=ro)
    s = (x.float() + res.float()).to(BF)
    assert torch.equal(ro, s)
    close(y2, _rms_ref(s, w, 1e-6), 2e-2, 1e-3, "rmsnorm+res")
    # fp32 partial-slab input (decode path: split-K skinny GEMM output summed inside the norm)
    x32 = torch.stack([x.float() * 0.25, x.float() * 0.5, x.float() * 0.25])
    y3, _ = ops.rmsnorm_fwd(None, w, 1e-6, res=res, res_out=ro, x32=x32)
    close(y3, y2, 2e-2, 1e-3, "rmsnorm x32")
    # backward vs autograd (fp32)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    yy = wr * (xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6))
    dy = rnd(T, H, seed=4)
    yy.backward(dy.float())
    dw = torch.zeros(H, dtype=F32, device=DEV)
    dres = rnd(T, H, seed=5)
    dx = ops.rmsnorm_bwd(dy, x, w, rstd, dres=dres, dw=dw)
    close(dx, xr.grad + dres.float(), 2e-2, 2e-2, "rmsnorm dx")
    close(dw, wr.grad, 1e-2, 1e-2 * math.sqrt(T), "rmsnorm dw")


@pytest.mark.parametrize("T,H", [(5, 160), (300, 1280), (131, 256)])
def test_layernorm_fwd_bwd(T, H):
    """nn.LayerNorm(eps=1e-6) with bias (Qwen2-VL vision tower) vs torch fp32."""
    x, res = rnd(T, H, seed=1), rnd(T, H, seed=3)
    w, b = (1 + 0.1 * rnd(H, seed=2).float()).to(BF), (0.1 * rnd(H, seed=6).float()).to(BF)
    ro = torch.empty_like(x)
    y, mean, rstd = ops.layernorm_fwd(x, w, b, 1e-6, res=res, res_out=ro, want_stats=True)
    s = (x.float() + res.float()).to(BF)
    assert torch.equal(ro, s)
    sr = s.float().requires_grad_(True)
    wr, br = w.float().requires_grad_(True), b.float().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(sr, (H,), wr, br, 1e-6)
    close(y, ref, 2e-2, 1e-2, "layernorm fwd")
    close(mean, s.float().mean(-1), 1e-5, 1e-5, "mean")
    close(rstd, torch.rsqrt(s.float().var(-1, unbiased=False) + 1e-6), 1e-4, 1e-5, "rstd")
    y0, _, _ = ops.layernorm_fwd(s, w, b, 1e-6)
    assert torch.equal(y0, y)
    dy, dres = rnd(T, H, seed=4), rnd(T, H, seed=5)
    ref.backward(dy.float())
    dw, db = torch.zeros(H, dtype=F32, device=DEV), torch.zeros(H, dtype=F32, device=DEV)
    dx = ops.layernorm_bwd(dy, s, w, mean, rstd, dres=dres, dw=dw, db=db)
    close(dx, sr.grad + dres.float(), 2e-2, 2e-2, "layernorm dx")
    close(dw, wr.grad, 1e-2, 1e-2 * math.sqrt(T), "layernorm dw")
    close(db, br.grad, 1e-2, 1e-2 * math.sqrt(T), "layernorm db")


def test_quick_gelu_fwd_bwd():
    z = (rnd(50, 640, seed=1).float() * 3).to(BF)
    zr = z.float().requires_grad_(True)
    ref = zr * torch.sigmoid(1.702 * zr)
    close(ops.quick_gelu_fwd(z), ref, 1e-2, 1e-2, "quick_gelu")
    da = rnd(50, 640, seed=2)
    ref.backward(da.float())
    close(ops.quick_gelu_bwd(da, z), zr.grad, 1e-2, 1e-2, "quick_gelu bwd")


# ------------------------------------------------------------------------------------------------ rope / act
@pytest.mark.parametrize("D,nh", [(128, 3), (80, 4)])
def test_rope(D, nh):
    T = 77
    width = (nh + 2) * D
    x = rnd(T, width, seed=1)
    ang = torch.rand(T, D // 2, generator=torch.Generator().manual_seed(2)) * 6.28
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    ref = x.float().clone()
    h = ref[:, : nh * D].view(T, nh, D)
    c2, s2 = torch.cat([cos, cos], -1).unsqueeze(1), torch.cat([sin, sin], -1).unsqueeze(1)
    rot = torch.cat([-h[..., D // 2:], h[..., : D // 2]], -1)
    ref[:, : nh * D] = (h * c2 + rot * s2).reshape(T, nh * D)
    y = x.clone()
    ops.rope_(y, cos, sin, nh, D)
    close(y, ref, 1e-2, 1e-2, "rope fwd")
    assert torch.equal(y[:, nh * D:], x[:, nh * D:])
    # backward = transpose: applying it to the forward output restores the input (orthogonal map)
    ops.rope_(y, cos, sin, nh, D, backward=True)
    close(y, x, 2e-2, 2e-2, "rope bwd∘fwd")


def test_swiglu_gelu_colsum():
    T, I = 70, 328
    gu = rnd(T, 2 * I, seed=1)
    a = ops.swiglu_fwd(gu)
    g, u = gu[:, :I].float(), gu[:, I:].float()
    close(a, torch.nn.functional.silu(g) * u, 1e-2, 1e-2, "swiglu fwd")
    da = rnd(T, I, seed=2)
    gr, ur = g.clone().requires_grad_(True), u.clone().requires_grad_(True)
    (torch.nn.functional.silu(gr) * ur).backward(da.float())
    dgu = ops.swiglu_bwd(da, gu)
    close(dgu[:, :I], gr.grad, 1e-2, 1e-2, "swiglu dgate")
    close(dgu[:, I:], ur.grad, 1e-2, 1e-2, "swiglu dup")
    z = rnd(T, 640, seed=3)
    close(ops.gelu_fwd(z), torch.nn.functional.gelu(z.float()), 1e-2, 1e-2, "gelu fwd")
    zr = z.float().requires_grad_(True)
    torch.nn.functional.gelu(zr).backward(rnd(T, 640, seed=4).float())
    close(ops.gelu_bwd(rnd(T, 640, seed=4), z), zr.grad, 1e-2, 1e-2, "gelu bwd")
    dy = rnd(1000, 520, seed=5)
    out = torch.ones(520, dtype=F32, device=DEV)
    ops.colsum_acc(dy, out)
    close(out, 1 + dy.float().sum(0), 1e-4, 1e-2, "colsum")


def test_embed():
    V, H, T = 100, 64, 50
    E, img = rnd(V, H, seed=1), rnd(7, H, seed=2)
    ids = torch.randint(0, V, (T,), generator=torch.Generator().manual_seed(3)).to(DEV)
    idx = torch.full((T,), -1, dtype=torch.int32)
    idx[5:12] = torch.arange(7, dtype=torch.int32)
    idx[30:33] = torch.tensor([1, 2, 3], dtype=torch.int32)  # the same image rows used twice (G copies share an image)
    idx = idx.to(DEV)
    out = ops.embed_fwd(ids, idx, E, img)
    ref = E[ids].clone()
    m = idx >= 0
    ref[m] = img[idx[m].long()]
    assert torch.equal(out, ref)
    dx = rnd(T, H, seed=4)
    dE, dimg = torch.zeros(V, H, dtype=F32, device=DEV), torch.zeros(7, H, dtype=F32, device=DEV)
    # backward = ordered scatter-add over host-built CSR plans (no atomics): which token rows feed which embedding row / image row
    ids_h, idx_h = ids.cpu().numpy(), idx.cpu().numpy().astype(np.int64)
    ops.rows_scatter_acc(dx, ops.scatter_plan(np.where(idx_h < 0, ids_h, -1), DEV), dE)
    ops.rows_scatter_acc(dx, ops.scatter_plan(idx_h, DEV), dimg)
    dE2, dimg2 = torch.zeros_like(dE), torch.zeros_like(dimg)
    ops.rows_scatter_acc(dx, ops.scatter_plan(np.where(idx_h < 0, ids_h, -1), DEV), dE2)
    ops.rows_scatter_acc(dx, ops.scatter_plan(idx_h, DEV), dimg2)
    assert torch.equal(dE, dE2) and torch.equal(dimg, dimg2)          # bit-reproducible
    rE, rI = torch.zeros_like(dE), torch.zeros_like(dimg)
    rE.index_add_(0, ids[~m], dx.float()[~m])
    rI.index_add_(0, idx[m].long(), dx.float()[m])
    close(dE, rE, 1e-5, 1e-5, "dE")
    close(dimg, rI, 1e-5, 1e-5, "dimg")


# ------------------------------------------------------------------------------------------------ attention
def _attn_ref(q, k, v, segs, Hq, Hkv, D, causal, scale):
    T = q.shape[0]
    qf, kf, vf = q.float().view(T, Hq, D), k.float().view(T, Hkv, D), v.float().view(T, Hkv, D)
    out = torch.zeros(T, Hq, D, device=q.device)
    lse = torch.full((Hq, T), float("-inf"), device=q.device)
    grp = Hq // Hkv
    for s, e in segs:
        qs, ks, vs = qf[s:e].transpose(0, 1), kf[s:e].transpose(0, 1).repeat_interleave(grp, 0), vf[s:e].transpose(0, 1).repeat_interleave(grp, 0)
        sc = (qs @ ks.transpose(1, 2)) * scale
        if causal:
            n = e - s
            sc = sc.masked_fill(~torch.ones(n, n, dtype=torch.bool, device=q.device).tril(), float("-inf"))
        lse[:, s:e] = torch.logsumexp(sc, -1)
        out[s:e] = (torch.softmax(sc, -1) @ vs).transpose(0, 1)
    return out.view(T, Hq * D), lse


ATTN_CASES = [
    # D, Hq, Hkv, causal, segments
    (128, 4, 2, True, [(0, 200), (205, 333), (340, 341), (400, 477)]),
    (128, 2, 1, True, [(3, 131)]),
    (128, 16, 2, True, [(0, 768), (768, 1536)]),
    (80, 2, 2, False, [(0, 64), (64, 128), (128, 176), (176, 192)]),
    (80, 3, 3, False, [(0, 300), (300, 364), (364, 1388)]),
    (128, 2, 2, False, [(0, 100)]),
    # head geometries of BASELINE configs 4 / 5 and the LLaVA decoders (REF sc_grpo_trainer.py:116-137): Qwen2.5-VL-7B / Qwen2-7B 28:4 (group 7),
    # LLaMA-7B MHA 32:32, Mistral-7B 32:8
    (128, 28, 4, True, [(0, 200), (205, 333), (340, 341), (400, 777)]),
    (128, 32, 32, True, [(0, 130), (130, 131), (140, 397)]),
    (128, 32, 8, True, [(5, 261), (261, 300), (300, 556)]),
]


@pytest.mark.parametrize("D,Hq,Hkv,causal,segs", ATTN_CASES)
def test_attn_fwd_bwd(D, Hq, Hkv, causal, segs):
    T = segs[-1][1] + 5
    W = (Hq + 2 * Hkv) * D
    qkv = rnd(T, W, seed=1, scale=1.0)
    q, k, v = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    scale = D ** -0.5
    seg = ops.Segments([s for s, _ in segs], [e for _, e in segs], DEV)
    o, lse = ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, causal, scale)
    ref_o, ref_lse = _attn_ref(q, k, v, segs, Hq, Hkv, D, causal, scale)
    inside = torch.zeros(T, dtype=torch.bool, device=DEV)
    for s, e in segs:
        inside[s:e] = True
    close(o[inside], ref_o[inside], 2e-2, 2e-2, "attn o")
    close(lse[:, inside], ref_lse[:, inside], 1e-3, 1e-2, "attn lse")
    # backward vs autograd of the fp32 reference
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    ro, _ = _attn_ref(qr, kr, vr, segs, Hq, Hkv, D, causal, scale)
    do = rnd(T, Hq * D, seed=2)
    do[~inside] = 0
    ro.backward(do.float())
    dqkv = torch.zeros(T, W, dtype=BF, device=DEV)
    ops.attn_bwd(q, k, v, o, do, lse, seg, Hq, Hkv, D, causal, scale, dqkv[:, : Hq * D], dqkv[:, Hq * D: (Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:])
    mag = lambda t: float(t.abs().max())
    close(dqkv[:, : Hq * D][inside], qr.grad[inside], 3e-2, 2e-2 * mag(qr.grad), "attn dq")
    close(dqkv[:, Hq * D: (Hq + Hkv) * D][inside], kr.grad[inside], 3e-2, 2e-2 * mag(kr.grad), "attn dk")
    close(dqkv[:, (Hq + Hkv) * D:][inside], vr.grad[inside], 3e-2, 2e-2 * mag(vr.grad), "attn dv")


PREFIX_CASES = [
    # Hq, Hkv, prompts [(start, end)], completions per prompt [[(start, end), ...], ...]
    (4, 2, [(3, 103), (110, 174)], [[(200, 230), (232, 233), (240, 337)], [(340, 404), (404, 450)]]),
    (16, 2, [(0, 512)], [[(512 + 256 * i, 512 + 256 * i + n) for i, n in enumerate([256, 17, 200, 64])]]),
    (2, 1, [(0, 37)], [[(40, 41)]]),
    # group 7 (28:4: Qwen2.5-VL-7B, LLaVA-OneVision-7B), MHA (LLaVA-1.5) and 32:8 (LLaVA-NeXT-Mistral) in the shared-prefix form
    (28, 4, [(0, 160), (160, 224)], [[(256 + 64 * i, 256 + 64 * i + n) for i, n in enumerate([64, 9, 50])], [(480, 544), (544, 545)]]),
    (32, 32, [(2, 98)], [[(100, 140), (140, 141), (150, 214)]]),
    (32, 8, [(0, 128)], [[(128, 192), (192, 230)]]),
]


@pytest.mark.parametrize("Hq,Hkv,prompts,comps", PREFIX_CASES)
def test_attn_shared_prefix_fwd_bwd(Hq, Hkv, prompts, comps):
    """Shared-prefix attention (seg_prefix): a completion segment's keys are its prompt's tokens (all visible) + its own tokens
    (causal).  Reference: per (prompt, completion) pair, plain causal attention over the concatenated row, fp32 autograd --
    exactly what the reference's [B*G, P+C] batch computes; prompt outputs / gradients from the prompt's own causal pass plus
    the key/value gradients of every completion."""
    D, scale = 128, 128 ** -0.5
    segs = list(prompts) + [c for cs in comps for c in cs]
    T = max(e for _, e in segs) + 3
    W = (Hq + 2 * Hkv) * D
    qkv = rnd(T, W, seed=1)
    q, k, v = qkv[:, : Hq * D], qkv[:, Hq * D: (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D:]
    prefix, first = [], len(prompts)
    for cs in comps:
        prefix.append([0, 0, first, len(cs)])
        first += len(cs)
    for (ps, pe), cs in zip(prompts, comps):
        prefix += [[ps, pe - ps, 0, 0] for _ in cs]
    seg = ops.Segments([s_ for s_, _ in segs], [e for _, e in segs], DEV, prefix=prefix)
    o, lse = ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, True, scale)
    do = rnd(T, Hq * D, seed=2)
    inside = torch.zeros(T, dtype=torch.bool, device=DEV)
    for s_, e in segs:
        inside[s_:e] = True
    do[~inside] = 0
    dqkv = torch.zeros(T, W, dtype=BF, device=DEV)
    ops.attn_bwd(q, k, v, o, do, lse, seg, Hq, Hkv, D, True, scale, dqkv[:, : Hq * D], dqkv[:, Hq * D: (Hq + Hkv) * D], dqkv[:, (Hq + Hkv) * D:])
    # reference
    qr, kr, vr = (t.float().clone().requires_grad_(True) for t in (q, k, v))
    grp = Hq // Hkv
    ref_o = torch.zeros(T, Hq * D, device=DEV)
    ref_lse = torch.full((Hq, T), float("-inf"), device=DEV)
    loss = 0.0

    def causal_rows(idx):  # attention of the token rows `idx` (one concatenated sequence), causal
        n = len(idx)
        qs = qr[idx].view(n, Hq, D).transpose(0, 1)
        ks = kr[idx].view(n, Hkv, D).transpose(0, 1).repeat_interleave(grp, 0)
        vs = vr[idx].view(n, Hkv, D).transpose(0, 1).repeat_interleave(grp, 0)
        sc = (qs @ ks.transpose(1, 2)) * scale
        sc = sc.masked_fill(~torch.ones(n, n, dtype=torch.bool, device=DEV).tril(), float("-inf"))
        return (torch.softmax(sc, -1) @ vs).transpose(0, 1).reshape(n, Hq * D), torch.logsumexp(sc, -1)

    for (ps, pe), cs in zip(prompts, comps):
        pidx = torch.arange(ps, pe, device=DEV)
        po, pl = causal_rows(pidx)
        ref_o[ps:pe], ref_lse[:, ps:pe] = po.detach(), pl.detach()
        loss = loss + (po * do[ps:pe].float()).sum()
        for cs_, ce in cs:
            idx = torch.cat([pidx, torch.arange(cs_, ce, device=DEV)])
            co, cl = causal_rows(idx)
            n = ce - cs_
            ref_o[cs_:ce], ref_lse[:, cs_:ce] = co[-n:].detach(), cl[:, -n:].detach()
            loss = loss + (co[-n:] * do[cs_:ce].float()).sum()
    loss.backward()
    close(o[inside], ref_o[inside], 2e-2, 2e-2, "prefix attn o")
    close(lse[:, inside], ref_lse[:, inside], 1e-3, 1e-2, "prefix attn lse")
    mag = lambda t: float(t.abs().max())
    close(dqkv[:, : Hq * D][inside], qr.grad[inside], 3e-2, 2e-2 * mag(qr.grad), "prefix attn dq")
    close(dqkv[:, Hq * D: (Hq + Hkv) * D][inside], kr.grad[inside], 3e-2, 2e-2 * mag(kr.grad), "prefix attn dk")
    close(dqkv[:, (Hq + Hkv) * D:][inside], vr.grad[inside], 3e-2, 2e-2 * mag(vr.grad), "prefix attn dv")


@pytest.mark.parametrize("Hq,Hkv,B,lens", [(16, 2, 5, [1, 33, 64, 517, 767]), (2, 1, 3, [7, 32, 100]), (28, 4, 2, [300, 31])])
def test_decode_attention_and_kv_store(Hq, Hkv, B, lens):
    D = 128
    maxp = (max(lens) + 31) // 32
    npages = B * maxp + 3
    kc = torch.zeros(npages, Hkv, 32, D, dtype=BF, device=DEV)
    vc = torch.zeros(npages, Hkv, D, 32, dtype=BF, device=DEV)
    perm = torch.randperm(npages, generator=torch.Generator().manual_seed(0))[: B * maxp].view(B, maxp).to(torch.int32)
    ks, vs, slots = [], [], []
    for b, n in enumerate(lens):
        ks.append(rnd(n, Hkv * D, seed=10 + b))
        vs.append(rnd(n, Hkv * D, seed=20 + b))
        pos = torch.arange(n)
        slots.append(perm[b, pos // 32].long() * 32 + pos % 32)
    kall, vall, slot = torch.cat(ks), torch.cat(vs), torch.cat(slots).to(DEV)
    slot_pad = torch.cat([slot, torch.tensor([-1], device=DEV)])  # a skipped (padding) row
    kall2, vall2 = torch.cat([kall, rnd(1, Hkv * D, seed=99)]), torch.cat([vall, rnd(1, Hkv * D, seed=98)])
    ops.kv_store(kall2, vall2, slot_pad, kc, vc, Hkv, D)
    q = rnd(B, Hq * D, seed=5)
    o = ops.attn_decode(q, kc, vc, perm.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), Hq, Hkv, D, D ** -0.5)
    op = ops.attn_decode(q, kc, vc, perm.to(DEV), torch.tensor(lens, dtype=torch.int32, device=DEV), Hq, Hkv, D, D ** -0.5, out=ops.PackedAct(B, Hq * D, DEV))
    assert torch.equal(op.unpack(), o)          # decode-packed output (ldo == 0): same values, MFMA-fragment order
    if B == len(lens) and Hq == 16:
        # fused decode-step variant: rope(q,k) + cache append of one new token per sequence == rope_ + kv_store
        qkv = rnd(B, (Hq + 2 * Hkv) * D, seed=7)
        ang = torch.rand(B, D // 2, generator=torch.Generator().manual_seed(3)) * 6.28
        cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
        new_slot = torch.stack([perm[b, lens[b] // 32].long() * 32 + lens[b] % 32 for b in range(B)]).to(DEV)
        kc1, vc1, kc2, vc2 = kc.clone(), vc.clone(), kc.clone(), vc.clone()
        a = qkv.clone()
        ops.rope_(a, cos, sin, Hq + Hkv, D)
        ops.kv_store(a[:, Hq * D: (Hq + Hkv) * D], a[:, (Hq + Hkv) * D:], new_slot, kc1, vc1, Hkv, D)
        bq = qkv.clone()
        ops.rope_kv_store(bq, cos, sin, new_slot, kc2, vc2, Hq, Hkv, D)
        assert torch.equal(bq[:, : Hq * D], a[:, : Hq * D]) and torch.equal(kc1, kc2) and torch.equal(vc1, vc2)
    for b, n in enumerate(lens):
        qf = q[b].float().view(Hq, 1, D)
        kf = ks[b].float().view(n, Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        vf = vs[b].float().view(n, Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        ref = (torch.softmax(qf @ kf.transpose(1, 2) * D ** -0.5, -1) @ vf).reshape(Hq * D)
        close(o[b], ref, 2e-2, 2e-2, f"decode b={b} n={n}")


@pytest.mark.parametrize("Hq,Hkv,M", [(16, 2, 64), (2, 1, 5), (28, 4, 33)])
def test_fused_qkv_rope_kv_equals_gemm_then_rope_kv_store(Hq, Hkv, M):
    """Decode-step fusion (iadr1_gemm_qkv_rope_kv_bf16) == skinny GEMM + bias, then rope_kv_store: q values and both cache pages."""
    D, K = 128, 256
    N = (Hq + 2 * Hkv) * D
    x, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(N, seed=3)
    ang = torch.rand(M, D // 2, generator=torch.Generator().manual_seed(3)) * 6.28
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    npages = M + 2
    slot = (torch.arange(M) * 32 + (torch.arange(M) * 7) % 32 + 32).to(DEV)
    slot[M // 2] = -1                                                        # a finished sequence: no cache write
    kc0, vc0 = rnd(npages, Hkv, 32, D, seed=4), rnd(npages, Hkv, D, 32, seed=5)
    # unfused
    qkv = ops.gemm_skinny(x, ops.pack_weight(w), N, bias=bias)
    kc1, vc1 = kc0.clone(), vc0.clone()
    ops.rope_kv_store(qkv, cos, sin, slot, kc1, vc1, Hq, Hkv, D)
    # fused, X both row-major and decode-packed
    wp, bp = ops.pack_qkv_rope(w, bias, Hq, Hkv, D)
    for xin in (x, ops.pack_act(x)):
        kc2, vc2 = kc0.clone(), vc0.clone()
        q2 = torch.zeros(M, N, dtype=BF, device=DEV)
        ops.gemm_qkv_rope_kv(xin, wp, bp, q2, cos, sin, slot, kc2, vc2, Hq, Hkv, D)
        close(q2[:, : Hq * D], qkv[:, : Hq * D], 1e-2, 1e-2, "fused q")       # fp32 rotary may contract differently: <= 1 bf16 ulp
        close(kc2, kc1, 1e-2, 1e-2, "fused K cache")
        assert torch.equal(vc2, vc1)                                          # V is a plain copy of the bf16 projection


def test_fused_v_tile_side_rows_at_the_7b_head_geometry():
    """28 q + 4 k + 4 v heads = 288 column tiles on 256 CUs: the K-head blocks of iadr1_gemm_qkv_rope_kv_bf16 take their V tile along (one round of blocks).  The
    training-arena rows written through `side` (roped q | roped k | v) and both cache pages equal the unfused reference; V is bit-equal."""
    Hq, Hkv, D, K, M = 28, 4, 128, 3584, 64
    N = (Hq + 2 * Hkv) * D
    x, w, bias = rnd(M, K, seed=4), rnd(N, K, seed=5, scale=0.1), rnd(N, seed=6)
    ang = torch.rand(M, D // 2, generator=torch.Generator().manual_seed(3)) * 6.28
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    slot = (torch.arange(M) * 32 + 5).to(DEV)
    kc, vc = torch.zeros(M + 2, Hkv, 32, D, dtype=BF, device=DEV), torch.zeros(M + 2, Hkv, D, 32, dtype=BF, device=DEV)
    step = torch.tensor([3], dtype=torch.int32, device=DEV)
    T, base, stride = 40 + M * 8, 20, 8
    rows = base + torch.arange(M, device=DEV) * stride + 3
    sq = torch.full((T, N), float("nan"), dtype=BF, device=DEV)
    wp, bp = ops.pack_qkv_rope(w, bias, Hq, Hkv, D)
    q_out = torch.zeros(M, N, dtype=BF, device=DEV)
    ops.gemm_qkv_rope_kv(ops.pack_act(x), wp, bp, q_out, cos, sin, slot, kc, vc, Hq, Hkv, D, side=ops.SideOut.make(step, base, stride, p0=sq))
    ref = ops.gemm_skinny(x, ops.pack_weight(w), N, bias=bias)
    kc1, vc1 = torch.zeros_like(kc), torch.zeros_like(vc)
    ops.rope_kv_store(ref.clone(), cos, sin, slot, kc1, vc1, Hq, Hkv, D)
    ops.rope_(ref, cos, sin, Hq + Hkv, D)
    assert torch.equal(sq[rows][:, : Hq * D], q_out[:, : Hq * D])
    close(sq[rows], ref, 1e-2, 1e-2, "side q|k|v rows")
    assert torch.equal(sq[rows][:, (Hq + Hkv) * D:], ref[:, (Hq + Hkv) * D:]) and torch.equal(vc, vc1)
    close(kc, kc1, 1e-2, 1e-2, "K cache")
    untouched = torch.ones(T, dtype=torch.bool, device=DEV)
    untouched[rows] = False
    assert bool(torch.isnan(sq[untouched].float()).all())


# ------------------------------------------------------------------------------------------------ losses
def test_logprob_dlogits_grpo():
    R, V = 37, 1288
    lg = rnd(R, V, seed=1, scale=3.0, dtype=F32)
    tg = torch.randint(0, V, (R,), generator=torch.Generator().manual_seed(2)).to(DEV)
    tg[3] = -100
    logp, lse = ops.logprob_rows(lg, tg)
    lp = torch.log_softmax(lg, -1)
    ref = lp.gather(1, tg.clamp(min=0).view(-1, 1)).squeeze(1)
    ref[3] = 0
    close(logp, ref, 1e-5, 1e-5, "logp")
    close(lse, torch.logsumexp(lg, -1), 1e-6, 1e-5, "lse")
    g = rnd(R, seed=3, dtype=F32)
    dl = ops.dlogits_rows(lg, tg, lse, g)
    oh = torch.zeros(R, V, device=DEV)
    oh[torch.arange(R, device=DEV)[tg >= 0], tg[tg >= 0]] = 1
    close(dl, g.view(-1, 1) * (oh - torch.softmax(lg, -1)), 1e-2, 1e-4, "dlogits")
    N, C = 8, 12
    p = (rnd(N, C, seed=4, dtype=F32) * 0.3 - 2).requires_grad_(True)
    r = p.detach() + rnd(N, C, seed=5, dtype=F32) * 0.1
    adv = rnd(N, seed=6, dtype=F32)
    mask = (torch.rand(N, C, generator=torch.Generator().manual_seed(7)) > 0.3).int().to(DEV)
    mask[:, 0] = 1
    kl = torch.exp(r - p) - (r - p) - 1
    ptl = -(torch.exp(p - p.detach()) * adv.unsqueeze(1) - 0.04 * kl)
    loss = ((ptl * mask).sum(1) / mask.sum(1)).mean()
    loss.backward()
    dlogp, klo, row_loss, row_kl = ops.grpo_loss(p.detach(), r, adv, mask, 0.04)
    close(dlogp, p.grad, 1e-4, 1e-7, "dlogp")
    close(klo, kl.detach(), 1e-4, 1e-7, "kl")
    close(row_loss.mean(), loss.detach(), 1e-5, 1e-7, "loss")
    close(row_kl.mean(), ((kl * mask).sum(1) / mask.sum(1)).mean().detach(), 1e-5, 1e-7, "mean kl")


def test_adamw_matches_torch():
    from iadr1_amd import hip
    n = 10007
    w0 = rnd(n, seed=1, dtype=F32)
    master, m, v = w0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pb = torch.empty(n, dtype=BF, device=DEV)
    wt = w0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([wt], lr=1e-2, weight_decay=0.1)
    for step in range(1, 4):
        g = rnd(n, seed=10 + step, dtype=F32) * 3
        wt.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([wt], 1.0)
        opt.step()
        grad = g.clone()
        norm2, scratch = torch.zeros(1, device=DEV), torch.zeros(2048, device=DEV)
        hip.call("sumsq", grad, n, scratch, norm2)
        norm2b = torch.zeros(1, device=DEV)
        hip.call("sumsq", grad, n, scratch, norm2b)
        assert torch.equal(norm2, norm2b)   # deterministic: replicas must agree bit for bit
        close(norm2[0], (g * g).sum(), 1e-4, 1e-3, "sumsq")
        hip.call("adamw_flat", master, m, v, grad, pb, n, 1e-2, 0.9, 0.999, 1e-8, 0.1, step, 1.0, norm2, 1.0)
        assert float(grad.abs().sum()) == 0.0
        close(master, wt.detach(), 1e-5, 1e-6, f"adamw step {step}")
        assert torch.equal(pb, master.to(BF))


# ------------------------------------------------------------------------------------------------ sampler
@pytest.mark.parametrize("V", [5000, 640, 151936])
def test_sampler_greedy_and_topk_topp(V):
    from oracle import sampler as osamp
    B = 16
    lg = rnd(B, V, seed=1, scale=2.5, dtype=F32)
    lg[2, 17] = lg[2, V - 100] = 50.0  # tie: lowest index wins
    out = ops.sample(lg, 0.0, 50, 0.9, seed=1, step=0)
    assert out.tolist() == lg.argmax(-1).tolist() and out[2].item() == 17
    sup = int(lg[5].argmax())
    out2 = ops.sample(lg, 0.0, 50, 0.9, seed=1, step=0, suppress_token=sup)
    assert out2[5].item() != sup
    lgc = lg.cpu().numpy()
    mism = 0
    for step in range(20 if V < 100000 else 3):
        got = ops.sample(lg, 0.9, 50, 0.9, seed=1234567890123, step=step).tolist()
        for b in range(B):
            ids, _ = osamp.candidates(lgc[b], 0.9, 50, 0.9)
            assert got[b] in set(ids.tolist()), "sampled token outside the top-k/top-p candidate set"
            ref, margin = osamp.sample_row(lgc[b], 0.9, 50, 0.9, 1234567890123, b, step)
            if margin > 1e-4:
                mism += got[b] != ref
    assert mism == 0
    # device-resident step counter gives the same stream as the argument
    sp = torch.tensor([7], dtype=torch.int32, device=DEV)
    assert torch.equal(ops.sample(lg, 0.9, 50, 0.9, seed=5, step=0, step_ptr=sp), ops.sample(lg, 0.9, 50, 0.9, seed=5, step=7))
    sd = torch.tensor([1234567890123], dtype=torch.int64, device=DEV)     # device-resident seed (graph replays): same stream as the scalar
    assert torch.equal(ops.sample(lg, 0.9, 50, 0.9, seed=0, step=3, seed_ptr=sd), ops.sample(lg, 0.9, 50, 0.9, seed=1234567890123, step=3))


def test_sampler_against_transformers_warpers():
    """sample.hip against the THIRD-PARTY golden of tools/make_golden_sampler.py (transformers' Temperature -> TopK -> TopP warpers on fixed logits;
    the reference's vLLM SamplingParams(temperature, top_p=0.9, top_k=50), REF sc_grpo_trainer.py:353-358): every draw lies in the kept set, every
    kept token with expected count >= 5 is drawn, and the empirical frequencies of 10 240 draws per row are within 3 sigma of the renormalised
    probabilities (+ a chi-square bound on the whole row).  A wrong candidate set or a sampler drawing from the un-renormalised distribution fails."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "sampler_hf.npz"))
    B, STEPS = 64, 160
    N = B * STEPS
    for i in range(len(g["vocab"])):
        t, k, p = (float(z) for z in g["settings"][i])
        V = int(g["vocab"][i])
        lg = torch.from_numpy(g["logits"][i, :V]).to(DEV).repeat(B, 1).contiguous()
        n = int(g["count"][i])
        ids, pr = g["ids"][i, :n], g["probs"][i, :n].astype(np.float64)
        draws = torch.stack([ops.sample(lg, t, int(k), p, seed=977 + i, step=s) for s in range(STEPS)]).flatten().cpu().numpy()
        assert set(np.unique(draws).tolist()) <= set(ids.tolist()), f"row {i}: a draw outside transformers' kept set"
        cnt = np.array([(draws == a).sum() for a in ids], dtype=np.float64)
        sig = np.sqrt(N * pr * (1 - pr))
        z = np.abs(cnt - N * pr) / np.maximum(sig, 1.0)
        assert z.max() < 3.0 + 1.0 * (n > 20), f"row {i}: frequency off by {z.max():.2f} sigma (token {int(ids[int(z.argmax())])})"   # 3 sigma; 4 when > 20 tokens are tested at once
        assert (cnt[N * pr >= 5] > 0).all(), f"row {i}: a kept token is never drawn"
        if n > 1:
            chi2 = float((((cnt - N * pr) ** 2) / (N * pr)).sum())
            assert chi2 < (n - 1) + 4.5 * math.sqrt(2 * (n - 1)) + 4, f"row {i}: chi2 {chi2:.1f} for {n - 1} dof"


@pytest.mark.gpu
@pytest.mark.parametrize("M,I,K,keep", [(2048, 3072, 256, True), (2048, 3072, 256, False), (4096, 1664, 512, True), (768, 11008, 2048, True)])
def test_gemm_swiglu_is_bit_identical_to_gemm_then_swiglu(M, I, K, keep):
    """Fused gate|up GEMM + SwiGLU epilogue (iadr1_gemm_swiglu_bf16) vs the two separate launches, bit for bit, incl. the stored gate|up matrix."""
    if not ops._FUSE_SWIGLU:
        pytest.skip("IADR1_FUSE_SWIGLU=0")
    g = torch.Generator(device="cpu").manual_seed(M + I + K)
    x = (torch.randn(M, K, generator=g) * 0.7).to(torch.bfloat16).cuda()
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5 * 2.0).to(torch.bfloat16).cuda()
    gu_ref = ops.gemm_nt(x, w)
    a_ref = ops.swiglu_fwd(gu_ref)
    assert (M // 256) * (I // 128) >= 192          # the fused kernel is what runs
    gu_buf = torch.full((M, 2 * I), float("nan"), dtype=torch.bfloat16, device="cuda")
    gu, a = ops.gemm_swiglu(x, w, gu_out=gu_buf, keep_gu=keep)
    assert torch.equal(a.view(torch.int16), a_ref.view(torch.int16))
    if keep:
        assert gu.data_ptr() == gu_buf.data_ptr() and torch.equal(gu.view(torch.int16), gu_ref.view(torch.int16))
    else:
        assert gu is None and bool(torch.isnan(gu_buf.float()).all())     # nothing was written
    # shapes the fused kernel does not take fall back to the two launches
    x2 = x[:300].contiguous()
    gu2, a2 = ops.gemm_swiglu(x2, w)
    assert torch.equal(a2.view(torch.int16), ops.swiglu_fwd(ops.gemm_nt(x2, w)).view(torch.int16)) and gu2 is not None


@pytest.mark.gpu
@pytest.mark.parametrize("nseq,block,stride,I,K,keep", [(64, 32, 256, 1664, 512, True), (32, 16, 48, 3072, 256, True), (64, 16, 40, 11008, 2048, False)])
def test_gemm_swiglu_rows_equals_the_plain_launch_on_gathered_rows(nseq, block, stride, I, K, keep):
    """iadr1_gemm_swiglu_rows_bf16 (row blocks of a sequence-major arena) vs iadr1_gemm_swiglu_bf16 on the same rows gathered into a dense matrix: bit-equal gate|up
    and SwiGLU rows at the mapped positions, every other row of the outputs untouched."""
    g = torch.Generator(device="cpu").manual_seed(nseq + block + I)
    base = 24
    T = base + nseq * stride + 8
    X = (torch.randn(T, K, generator=g) * 0.7).to(torch.bfloat16).cuda()
    w = (torch.randn(2 * I, K, generator=g) * K ** -0.5 * 2.0).to(torch.bfloat16).cuda()
    off = 8                                             # the chunk starts 8 rows into every sequence
    rows = (base + off + torch.arange(nseq)[:, None] * stride + torch.arange(block)[None, :]).reshape(-1).cuda()
    xg = X[rows].contiguous()
    gu_ref = torch.empty(rows.numel(), 2 * I, dtype=torch.bfloat16, device="cuda")
    a_ref = torch.empty(rows.numel(), I, dtype=torch.bfloat16, device="cuda")
    ops.gemm_swiglu_fused(xg, w, gu_ref, a_ref)
    GU = torch.full((T, 2 * I), float("nan"), dtype=torch.bfloat16, device="cuda")
    A = torch.full((T, I), float("nan"), dtype=torch.bfloat16, device="cuda")
    ops.gemm_swiglu_rows(X[base + off:], w, GU[base + off:] if keep else None, A[base + off:], nseq, block, stride)
    assert torch.equal(A[rows].view(torch.int16), a_ref.view(torch.int16))
    other = torch.ones(T, dtype=torch.bool, device="cuda"); other[rows] = False
    assert bool(torch.isnan(A[other].float()).all())
    if keep:
        assert torch.equal(GU[rows].view(torch.int16), gu_ref.view(torch.int16)) and bool(torch.isnan(GU[other].float()).all())
    else:
        assert bool(torch.isnan(GU.float()).all())
    with pytest.raises(RuntimeError):
        ops.gemm_swiglu_rows(X[base:], w, None, A[base:], nseq, 24, stride)        # not a power of two


@pytest.mark.gpu
@pytest.mark.parametrize("N,K,T", [(2560, 2048, 12288), (22016, 2048, 4096), (2048, 11008, 20480), (2568, 2048, 4100), (22024, 2056, 1000), (1280, 1280, 8192)])
def test_gemm_tn_acc_is_bit_identical_to_the_transposed_nt_form(N, K, T):
    """iadr1_gemm_tn_acc_bf16 (weight gradient straight from row-major dY [T, N] and X [T, K]: hardware transpose reads out of LDS) vs the two transposes +
    gemm_nt(accumulate) -- plain and split-K launches, ragged T (not a multiple of 64 / of 8), N / K that are not multiples of 256: the same bits, accumulated
    onto a non-zero gradient buffer; row strides larger than the width (views of wider buffers)."""
    g = torch.Generator(device="cpu").manual_seed(N + K + T)
    dyb = (torch.randn(T, N + 8, generator=g) * 0.5).to(torch.bfloat16).cuda()
    xb = (torch.randn(T, K + 16, generator=g) * 0.5).to(torch.bfloat16).cuda()
    dy, x = dyb[:, :N], xb[:, :K]
    base = torch.randn(N, K, generator=g).cuda()
    want = ops.gemm_nt(ops.transpose(dy, pad_rows_to=8), ops.transpose(x, pad_rows_to=8), out=base.clone(), accumulate=True)
    got = base.clone()
    calls = []
    orig, mode = ops.hip.call, ops._GEMM_TN
    ops.hip.call = lambda name, *a: (calls.append(name), orig(name, *a))[1]
    ops._GEMM_TN = "2"                                               # every shape the 256 x 256 kernel takes
    try:
        ops.gemm_tn_acc(dy, x, got)
    finally:
        ops.hip.call, ops._GEMM_TN = orig, mode
    assert calls == ["gemm_tn_acc_bf16"], calls                     # the TN kernel is what ran (no transposes)
    assert torch.equal(got.view(torch.int32), want.view(torch.int32))
    ref = base.double() + dy.double().t() @ x.double()
    assert float((got.double() - ref).abs().max()) < 2e-2 * float(ref.abs().max())


@pytest.mark.gpu
def test_decode_side_outputs_equal_the_main_outputs():
    """iadr1_side_out_t (`side` argument of the four decode-step entry points): every kernel that carries side outputs writes, at row
    base + s * stride + *step of row-major training buffers, exactly what its main (decode-packed / paged) output holds; without `side` nothing else
    is written (the struct is an explicit per-call argument, there is no armed state)."""
    step = torch.tensor([3], dtype=torch.int32, device=DEV)
    base, stride, T = 10, 7, 10 + 7 * 64 + 8
    row = lambda s: base + s * stride + 3
    rows = torch.tensor([row(s) for s in range(64)], device=DEV)
    nan = lambda *shape, dtype=BF: torch.full(shape, float("nan"), dtype=dtype, device=DEV)
    # RMSNorm over split-K slabs + residual (the decode form): residual rows, normalised rows, rstd
    H = 2048
    slabs, res, g = rnd(2, 64, H, seed=1, dtype=F32), rnd(64, H, seed=2), rnd(H, seed=3)
    sx, sy, sr = nan(T, H), nan(T, H), nan(T, dtype=F32)
    res_out, y = torch.empty_like(res), ops.PackedAct(64, H, DEV)
    _, rstd = ops.rmsnorm_fwd(None, g, 1e-6, res=res, res_out=res_out, x32=slabs, out=y, want_rstd=True, side=ops.SideOut.make(step, base, stride, p0=sx, p1=sy, p2=sr))
    assert torch.equal(sx[rows], res_out) and torch.equal(sy[rows], y.unpack()) and torch.equal(sr[rows], rstd)
    untouched = torch.ones(T, dtype=torch.bool, device=DEV); untouched[rows] = False
    assert bool(torch.isnan(sx[untouched].float()).all())
    sy2 = nan(T, H)
    ops.rmsnorm_fwd(None, g, 1e-6, res=res, res_out=res_out, x32=slabs, out=y)     # no `side`: no side output
    assert bool(torch.isnan(sy2.float()).all())
    # q|k|v projection + rotary + cache append: the roped rows in [q | k | v] order
    Hq, Hkv, D, K = 16, 2, 128, 256
    N = (Hq + 2 * Hkv) * D
    x, w, bias = rnd(64, K, seed=4), rnd(N, K, seed=5, scale=0.3), rnd(N, seed=6)
    ang = torch.rand(64, D // 2, generator=torch.Generator().manual_seed(3)) * 6.28
    cos, sin = ang.cos().to(DEV), ang.sin().to(DEV)
    slot = (torch.arange(64) * 32 + 5).to(DEV)
    kc, vc = torch.zeros(66, Hkv, 32, D, dtype=BF, device=DEV), torch.zeros(66, Hkv, D, 32, dtype=BF, device=DEV)
    wp, bp = ops.pack_qkv_rope(w, bias, Hq, Hkv, D)
    q_out, sq = torch.zeros(64, N, dtype=BF, device=DEV), nan(T, N)
    ops.gemm_qkv_rope_kv(ops.pack_act(x), wp, bp, q_out, cos, sin, slot, kc, vc, Hq, Hkv, D, side=ops.SideOut.make(step, base, stride, p0=sq))
    ref = ops.gemm_skinny(x, ops.pack_weight(w), N, bias=bias)
    ops.rope_(ref, cos, sin, Hq + Hkv, D)
    assert torch.equal(sq[rows][:, : Hq * D], q_out[:, : Hq * D])
    close(sq[rows], ref, 1e-2, 1e-2, "side q|k|v rows")
    assert torch.equal(sq[rows][:, (Hq + Hkv) * D:], ref[:, (Hq + Hkv) * D:])       # V: a plain copy of the bf16 projection
    # paged attention: output rows + log-sum-exp
    lens = [1 + (37 * s) % 200 for s in range(64)]
    maxp = (max(lens) + 31) // 32
    kc = torch.zeros(64 * maxp + 1, Hkv, 32, D, dtype=BF, device=DEV)
    vc = torch.zeros(64 * maxp + 1, Hkv, D, 32, dtype=BF, device=DEV)
    table = (torch.arange(64 * maxp).view(64, maxp) + 1).to(torch.int32).to(DEV)
    ks, vs = [rnd(n, Hkv * D, seed=100 + s) for s, n in enumerate(lens)], [rnd(n, Hkv * D, seed=300 + s) for s, n in enumerate(lens)]
    slots = torch.cat([table[s, torch.arange(n) // 32].long().cpu() * 32 + torch.arange(n) % 32 for s, n in enumerate(lens)]).to(DEV)
    ops.kv_store(torch.cat(ks), torch.cat(vs), slots, kc, vc, Hkv, D)
    q = rnd(64, Hq * D, seed=7)
    so, sl = nan(T, Hq * D), nan(Hq, T, dtype=F32)
    o = ops.attn_decode(q, kc, vc, table, torch.tensor(lens, dtype=torch.int32, device=DEV), Hq, Hkv, D, D ** -0.5, out=ops.PackedAct(64, Hq * D, DEV),
                        side=ops.SideOut.make(step, base, stride, p0=so, p1=sl, ld1=T))
    assert torch.equal(so[rows], o.unpack())
    for s in (0, 17, 63):
        kf = ks[s].float().view(lens[s], Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        lse = torch.logsumexp(q[s].float().view(Hq, 1, D) @ kf.transpose(1, 2) * D ** -0.5, -1).view(Hq)
        close(sl[:, row(s)], lse, 2e-3, 2e-3, f"side lse s={s}")
    # persistent fused-SwiGLU gate|up projection: pre-activation rows [gate | up] and the activation rows
    if os.environ.get("IADR1_SKINNY_PERS", "1") == "0":
        return
    I, K = 8192, 1024
    x, w = rnd(64, K, seed=8), rnd(2 * I, K, seed=9, scale=K ** -0.5 * 2)
    sg, sa = nan(T, 2 * I), nan(T, I)
    a = ops.gemm_skinny(ops.pack_act(x), ops.pack_gateup(w), 2 * I, swiglu=True, out=ops.PackedAct(64, I, DEV), side=ops.SideOut.make(step, base, stride, p0=sg, p1=sa))
    assert torch.equal(sa[rows], a.unpack())
    gu_ref = ops.gemm_nt(x, w)
    close(sg[rows], gu_ref, 1e-2, 1e-2, "side gate|up rows")
    assert torch.equal(ops.swiglu_fwd(sg[rows].contiguous()), sa[rows])              # the activation is the SwiGLU of exactly the stored pre-activations


# ------------------------------------------------------------------------------------------------ FP8 weight stream of the rollout (opt-in)
def _fp8_reference(w, scale=None):
    """torch restatement of iadr1_pack_weight_fp8: per-row scale amax / 448, round-to-nearest-even to OCP e4m3; returns (fp8 values [N,K], scales, dequantised fp32).
    `scale`: quantise with these scales (the device's: its amax / 448 may differ from torch's in the last bit, which moves exact ties like 84 -> 80 | 88)."""
    wf = w.float()
    own = wf.abs().amax(1).clamp_min(1e-30) / 448.0
    scale = own if scale is None else scale
    q = (wf / scale[:, None]).to(torch.float8_e4m3fn)
    return q, own, q.float() * scale[:, None]


def _unpack_fp8(buf, N, K, gateup=False):
    t = buf.view(N // 16, K // 64, 4, 16, 2, 8).permute(0, 3, 1, 4, 2, 5).reshape(N, K)      # [tile][dstep][g][n%16][h][8] -> [n][k]
    if gateup:                                                                                 # tiles 2q / 2q+1 = gate / up rows of the same 16 columns
        I = N // 2
        t = t.view(N // 32, 2, 16, K)
        t = torch.cat([t[:, 0].reshape(I, K), t[:, 1].reshape(I, K)], 0)
    return t


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,packed", [(64, 2048, 11008, True), (64, 151936, 2048, True), (8, 256, 512, False), (33, 3584, 18944, True), (100, 640, 256, False),
                                          (40, 16384, 1024, True), (64, 17920, 1536, True), (64, 16384, 2048, False)])
def test_fp8_weight_gemm(M, N, K, packed):
    """FP8 (e4m3, per-row scale) decode weights: the pack holds exactly torch's round-to-nearest e4m3 of w / scale, and the GEMM on it equals the GEMM on the
    dequantised weights at the bf16 kernels' own tolerance -- the kernel adds no error to the format's.  fp32 / bf16+bias / split-K slab outputs.  Shapes of
    every kernel form: persistent X-resident (K = 1024 / 1536 / 2048, N >= 16384), persistent split-K (2048 x 11008, 8 slices), one-shot wide (the rest)."""
    x, w, bias = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=0.3), rnd(N, seed=3)
    w8, sc = ops.pack_weight_fp8(w)
    q, scale, deq = _fp8_reference(w, scale=sc)
    assert torch.equal(_unpack_fp8(w8, N, K), q.view(torch.uint8)) and torch.allclose(sc, scale, rtol=1e-6, atol=0)
    ref = x.float() @ deq.t()
    xin = ops.pack_act(x) if packed else x
    close(ops.gemm_skinny(xin, (w8, sc), N, out_dtype=F32), ref, 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"fp8w f32 {M}x{N}x{K}")
    close(ops.gemm_skinny(xin, (w8, sc), N, bias=bias), ref + bias.float(), 1e-2, 1e-2 * math.sqrt(K) * 0.3, f"fp8w bf16+bias {M}x{N}x{K}")
    for ks in (2, 8):
        part = ops.gemm_skinny(xin, (w8, sc), N, out=torch.full((ks, M, N), 7.0, dtype=F32, device=DEV), ksplit=ks)
        close(part.sum(0), ref, 1e-4, 1e-3 * math.sqrt(K) * 0.1, f"fp8w split-K {ks}")
    # against the ORIGINAL weights: the format's error, 3 mantissa bits per weight -> a few percent of the output's scale
    full = x.float() @ w.float().t()
    rel = ((ref - full).norm() / full.norm()).item()
    assert rel < 0.04, rel


@pytest.mark.gpu
@pytest.mark.parametrize("M,I,K", [(64, 11008, 2048), (64, 512, 256), (7, 192, 128), (64, 18944, 3584)])
def test_fp8_weight_gemm_fused_swiglu(M, I, K):
    x, w = rnd(M, K, seed=1), rnd(2 * I, K, seed=2, scale=0.3)
    w8, sc = ops.pack_weight_fp8(w, gateup=True)
    q, scale, deq = _fp8_reference(w, scale=sc)
    assert torch.equal(_unpack_fp8(w8, 2 * I, K, gateup=True), q.view(torch.uint8))
    gu = (x.float() @ deq.t()).to(BF)
    ref = torch.nn.functional.silu(gu[:, :I].float()).to(BF).float() * gu[:, I:].float()
    a = ops.gemm_skinny(ops.pack_act(x), (w8, sc), 2 * I, swiglu=True, out=ops.PackedAct(M, I, DEV)).unpack()
    close(a, ref, 2e-2, 2e-2 * math.sqrt(K) * 0.3, f"fp8w swiglu {M}x{I}x{K}")


@pytest.mark.gpu
@pytest.mark.parametrize("Hq,Hkv,G,prompts,extra", [(16, 2, 8, [512, 777], [0, 5, 130, 31, 64, 1, 255, 200]), (32, 32, 8, [832], [3, 60, 0, 128, 7, 33, 90, 255]),
                                                    (32, 8, 8, [3184, 40], [1, 2, 3, 4, 5, 6, 7, 8]), (28, 4, 8, [100], [9, 19, 29, 39, 49, 59, 69, 79]),
                                                    (2, 1, 4, [7, 64, 31], [0, 1, 32, 40]), (16, 16, 3, [300], [5, 6, 7])])
def test_decode_attention_group_equals_per_sequence(Hq, Hkv, G, prompts, extra):
    """iadr1_attn_decode_group (one block per (prompt group, kv head): the group's full prompt pages read once for all of its query rows, then every sequence's
    private pages, merged) against iadr1_attn_decode (one block per (sequence, kv head)) on the rollout's page layout -- shared full prompt pages, private pages for the
    prompt remainder + completion -- and against fp32 torch; GQA groups of 8 / 4 / 7 / 2, MHA, prompts shorter than a page (no shared page), 1..4 column tiles."""
    D, PAGE = 128, 32
    Bp = len(prompts)
    B = Bp * G
    ctx = [prompts[b // G] + extra[b % G] + 1 for b in range(B)]                    # prompt + generated so far (+ the current token)
    n_shared = [p // PAGE for p in prompts]
    maxp = max((c_ + PAGE - 1) // PAGE for c_ in ctx) + 1
    bt = np.zeros((B, maxp), dtype=np.int32)
    nxt = 1
    shared = []
    for b in range(Bp):
        shared.append(list(range(nxt, nxt + n_shared[b])))
        nxt += n_shared[b]
    for r in range(B):
        priv = (ctx[r] + PAGE - 1) // PAGE - n_shared[r // G]
        pages = shared[r // G] + list(range(nxt, nxt + priv))
        nxt += priv
        bt[r, : len(pages)] = pages
    kc = torch.zeros(nxt + 1, Hkv, 32, D, dtype=BF, device=DEV)
    vc = torch.zeros(nxt + 1, Hkv, D, 32, dtype=BF, device=DEV)
    keys, vals = {}, {}
    for b in range(Bp):                                                              # prompt K/V: identical for the sequences of a group
        pk, pv = rnd(prompts[b], Hkv * D, seed=100 + b), rnd(prompts[b], Hkv * D, seed=200 + b)
        for s_ in range(G):
            r = b * G + s_
            n_own = ctx[r] - prompts[b]
            keys[r] = torch.cat([pk, rnd(n_own, Hkv * D, seed=300 + r)])
            vals[r] = torch.cat([pv, rnd(n_own, Hkv * D, seed=400 + r)])
            pos = torch.arange(ctx[r])
            slot = torch.from_numpy(bt[r])[pos // PAGE].long() * PAGE + pos % PAGE
            ops.kv_store(keys[r], vals[r], slot.to(DEV), kc, vc, Hkv, D)            # (shared pages are written G times with the same values)
    q = rnd(B, Hq * D, seed=5)
    btd, ctxd = torch.from_numpy(bt).to(DEV), torch.tensor(ctx, dtype=torch.int32, device=DEV)
    sp = torch.tensor(n_shared, dtype=torch.int32, device=DEV)
    o_seq = ops.attn_decode(q, kc, vc, btd, ctxd, Hq, Hkv, D, D ** -0.5)
    o_grp = ops.attn_decode_group(q, kc, vc, btd, ctxd, sp, G, Hq, Hkv, D, D ** -0.5)
    close(o_grp, o_seq, 1e-2, 4e-3, "group vs per-sequence decode attention")        # same arithmetic, different fp32 summation order: a bf16 ulp
    op = ops.attn_decode_group(q, kc, vc, btd, ctxd, sp, G, Hq, Hkv, D, D ** -0.5, out=ops.PackedAct(B, Hq * D, DEV))
    assert torch.equal(op.unpack(), o_grp)
    for chunks in (2, 5):       # shared pages split over `chunks` blocks per (group, kv head): two launches, partial states merged in chunk order
        o_ch = ops.attn_decode_group(q, kc, vc, btd, ctxd, sp, G, Hq, Hkv, D, D ** -0.5, chunks=chunks)
        close(o_ch, o_seq, 1e-2, 4e-3, f"chunked group attention ({chunks})")
        assert torch.equal(o_ch, ops.attn_decode_group(q, kc, vc, btd, ctxd, sp, G, Hq, Hkv, D, D ** -0.5, chunks=chunks))       # reproducible
    for r in (0, B // 2, B - 1):
        n = ctx[r]
        qf = q[r].float().view(Hq, 1, D)
        kf = keys[r].float().view(n, Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        vf = vals[r].float().view(n, Hkv, D).transpose(0, 1).repeat_interleave(Hq // Hkv, 0)
        ref = (torch.softmax(qf @ kf.transpose(1, 2) * D ** -0.5, -1) @ vf).reshape(Hq * D)
        close(o_grp[r], ref, 2e-2, 2e-2, f"group decode r={r} n={n}")
    # side outputs: the attention rows and log-sum-exp land in the training arena exactly as the per-sequence kernel writes them
    step = torch.tensor([3], dtype=torch.int32, device=DEV)
    T = B * 8 + 10
    arena_o, arena_l = torch.zeros(T, Hq * D, dtype=BF, device=DEV), torch.zeros(Hq, T, dtype=F32, device=DEV)
    so = ops.SideOut.make(step, 2, 8, p0=arena_o, p1=arena_l, ld1=T)
    ops.attn_decode_group(q, kc, vc, btd, ctxd, sp, G, Hq, Hkv, D, D ** -0.5, side=so)
    rows = 2 + torch.arange(B, device=DEV) * 8 + 3
    assert torch.equal(arena_o[rows], o_grp)
    arena_o2, arena_l2 = torch.zeros_like(arena_o), torch.zeros_like(arena_l)
    ops.attn_decode(q, kc, vc, btd, ctxd, Hq, Hkv, D, D ** -0.5, side=ops.SideOut.make(step, 2, 8, p0=arena_o2, p1=arena_l2, ld1=T))
    close(arena_l[:, rows], arena_l2[:, rows], 1e-5, 1e-5, "log-sum-exp side output")


@pytest.mark.gpu
@pytest.mark.parametrize("Hq,Hkv,G,Bp", [(16, 2, 8, 8), (28, 4, 8, 3), (2, 1, 4, 11), (16, 16, 2, 9)])
def test_decode_attention_group_placement_hint_does_not_change_results(Hq, Hkv, G, Bp):
    """seqs_per_group re-orders the blocks of iadr1_attn_decode (one XCD per prompt group); outputs are bit-identical with and without it, for group counts that are and are
    not multiples of the 8 XCDs."""
    D, B = 128, Bp * G
    lens = [40 + 13 * (b % 7) + 64 * (b // G % 3) for b in range(B)]
    maxp = (max(lens) + 31) // 32
    kc = rnd(B * maxp + 2, Hkv * 32 * D, seed=1).view(-1, Hkv, 32, D)
    vc = rnd(B * maxp + 2, Hkv * D * 32, seed=2).view(-1, Hkv, D, 32)
    bt = torch.randperm(B * maxp + 2, generator=torch.Generator().manual_seed(0))[: B * maxp].view(B, maxp).to(torch.int32).to(DEV)
    ctx = torch.tensor(lens, dtype=torch.int32, device=DEV)
    q = rnd(B, Hq * D, seed=5)
    a = ops.attn_decode(q, kc, vc, bt, ctx, Hq, Hkv, D, D ** -0.5)
    b_ = ops.attn_decode(q, kc, vc, bt, ctx, Hq, Hkv, D, D ** -0.5, seqs_per_group=G)
    assert torch.equal(a, b_)
