"""ORACLE (test infrastructure, not product): fp32 CPU restatement of the Qwen2.5-VL forward that
the reference's hot path calls (`model(**inputs).logits`,
/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:505).

The arithmetic lives in a third-party dependency of the reference that is NOT under /root/reference:
`transformers==4.51.3` (requirements.txt:205).  This file restates the published algorithm of
`transformers/models/qwen2_5_vl/modeling_qwen2_5_vl.py` (line numbers below are for the installed
5.15.0, cited as TF:) and is pinned against golden vectors captured from that implementation by
tools/make_golden.py (tests/golden/logps_padded.npz, sc_grpo_g*.npz, greedy.npz, sft.npz,
vision_index.json) -- see tests/test_oracle_*.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Plain torch ops on CPU tensors, autograd for gradients; no HIP, no transformers import.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# index helpers (pure integer work; bit-exact targets)
# ----------------------------------------------------------------------------------------------
def vision_window_index(grid_thw, spatial_merge_size=2, window_size=112, patch_size=14):
    """TF:vision_utils.py:130-185 get_vision_window_index.  Returns (window_index [N/4] long,
    cu_window_seqlens list[int] in units of patches, consecutive duplicates removed)."""
    win = window_size // spatial_merge_size // patch_size
    unit = spatial_merge_size**2
    index_out, cu, base = [], [0], 0
    for t, h, w in grid_thw:
        lh, lw = h // spatial_merge_size, w // spatial_merge_size
        # NOTE: pads a full extra window when the grid is already a multiple (as the reference does)
        ph, pw = win - lh % win, win - lw % win
        nh, nw = (lh + ph) // win, (lw + pw) // win
        idx = torch.full((t, lh + ph, lw + pw), -100, dtype=torch.long)
        idx[:, :lh, :lw] = torch.arange(t * lh * lw).view(t, lh, lw)
        idx = idx.view(t, nh, win, nw, win).permute(0, 1, 3, 2, 4).reshape(t, nh * nw, win * win)
        counts = (idx != -100).sum(-1).reshape(-1)
        flat = idx.reshape(-1)
        index_out.append(flat[flat != -100] + base)
        for c in (counts.cumsum(0) * unit + cu[-1]).tolist():
            cu.append(c)
        base += t * lh * lw
    dedup = [cu[0]]
    for c in cu[1:]:
        if c != dedup[-1]:
            dedup.append(c)
    return torch.cat(index_out), dedup


def vision_position_ids(grid_thw, spatial_merge_size=2):
    """TF:vision_utils.py:81-127: (h, w) index per patch in merge-block-major order -> [N, 2]."""
    out = []
    m = spatial_merge_size
    for t, h, w in grid_thw:
        hp = torch.arange(h).view(h, 1).expand(h, w)
        wp = torch.arange(w).view(1, w).expand(h, w)
        hp = hp.reshape(h // m, m, w // m, m).transpose(1, 2).flatten()
        wp = wp.reshape(h // m, m, w // m, m).transpose(1, 2).flatten()
        out.append(torch.stack([hp, wp], -1).repeat(t, 1))
    return torch.cat(out, 0)


def vision_cu_seqlens(grid_thw):
    """TF:vision_utils.py:41-65: one full-attention segment per frame."""
    cu = [0]
    for t, h, w in grid_thw:
        for _ in range(t):
            cu.append(cu[-1] + h * w)
    return cu


def mrope_position_ids(input_ids, attention_mask, grid_thw, image_token_id, spatial_merge_size=2):
    """TF:modeling_qwen2_5_vl.py:944-1062 get_rope_index (images only).  [3,B,S] long + deltas [B,1].
    Padded slots keep position 0; text runs count 1-D; an image run gets (t,h,w) = (st, st+row,
    st+col) on the merged grid and the next text resumes at st + max(h,w)/merge."""
    B, S = input_ids.shape
    pos = torch.zeros(3, B, S, dtype=torch.long)
    deltas = []
    gi = 0
    for b in range(B):
        keep = attention_mask[b].bool()
        ids = input_ids[b][keep].tolist()
        chunks, cur, i = [], 0, 0
        while i < len(ids):
            j = i
            is_img = ids[i] == image_token_id
            while j < len(ids) and (ids[j] == image_token_id) == is_img:
                j += 1
            if not is_img:
                n = j - i
                chunks.append(torch.arange(n).view(1, -1).expand(3, -1) + cur)
                cur += n
            else:
                t, h, w = grid_thw[gi]
                gi += 1
                lh, lw = h // spatial_merge_size, w // spatial_merge_size
                tt = torch.arange(t).view(t, 1, 1).expand(t, lh, lw).reshape(-1)
                hh = torch.arange(lh).view(1, lh, 1).expand(t, lh, lw).reshape(-1)
                ww = torch.arange(lw).view(1, 1, lw).expand(t, lh, lw).reshape(-1)
                assert j - i == t * lh * lw, "image-pad run does not match grid"
                chunks.append(torch.stack([tt, hh, ww]) + cur)
                cur += max(h, w) // spatial_merge_size
            i = j
        p = torch.cat(chunks, 1)
        pos[:, b, keep] = p
        deltas.append(int(p.max()) + 1 - len(ids))
    return pos, torch.tensor(deltas).view(-1, 1)


# ----------------------------------------------------------------------------------------------
# float ops
# ----------------------------------------------------------------------------------------------
def rmsnorm(x, w, eps):
    """TF:modeling_qwen2_5_vl.py:74-79."""
    var = x.float().pow(2).mean(-1, keepdim=True)
    return w * (x.float() * torch.rsqrt(var + eps)).to(x.dtype)


def rotate_half(x):
    h = x.shape[-1] // 2
    return torch.cat([-x[..., h:], x[..., :h]], -1)


def _segment_attention(q, k, v, cu, scale):
    """Non-causal softmax attention inside each [cu[i], cu[i+1]) segment (TF::225-291 eager path).
    q,k,v: [N, H, d]."""
    out = torch.empty_like(q)
    for a, b in zip(cu[:-1], cu[1:]):
        qs, ks, vs = (z[a:b].transpose(0, 1) for z in (q, k, v))  # [H, n, d]
        p = torch.softmax((qs @ ks.transpose(1, 2)) * scale, -1, dtype=torch.float32).to(q.dtype)
        out[a:b] = (p @ vs).transpose(0, 1)
    return out


class Qwen25VLOracle:
    """Functional fp32 model over a {checkpoint-name: tensor} dict."""

    def __init__(self, cfg: dict, weights: dict, requires_grad=False, dtype=torch.float32, copy: bool = True):
        """requires_grad: bool, or a collection of checkpoint names (gradients for those tensors only).  copy=False adopts the given tensors
        (full-size models: 15 GB of fp32 for the 3B shapes -- tests/test_hip_model.py full-depth parity)."""
        self.cfg = cfg
        self.w = {}
        for k, a in weights.items():
            t = torch.as_tensor(a).to(dtype)
            if copy:
                t = t.clone()
            t.requires_grad_(requires_grad if isinstance(requires_grad, bool) else k in requires_grad)
            self.w[k] = t
        if cfg.get("tie_word_embeddings", False):
            self.w["lm_head.weight"] = self.w["model.embed_tokens.weight"]

    def parameters(self):
        """(name, leaf tensor) pairs; the tied lm_head alias is not repeated."""
        tied = self.cfg.get("tie_word_embeddings", False)
        for k, t in self.w.items():
            if tied and k == "lm_head.weight":
                continue
            yield k, t

    # -- Qwen2-VL vision tower: TF:models/qwen2_vl/modeling_qwen2_vl.py:650-720 (blocks ::418-447, MLP ::293-301,
    #    merger ::270-290): LayerNorm blocks, fc1 -> QuickGELU -> fc2, full attention per image, LayerNorm merger --------
    def _visual_qwen2vl(self, pixel_values, grid_thw, return_last_hidden=False):
        v = self.cfg["vision"]
        w = self.w
        vh, nh = v["hidden_size"], v["num_heads"]
        d = vh // nh
        m2 = v["spatial_merge_size"] ** 2
        x = pixel_values.to(w["visual.patch_embed.proj.weight"].dtype) @ w["visual.patch_embed.proj.weight"].reshape(vh, -1).t()
        n = x.shape[0]
        cu = vision_cu_seqlens(grid_thw)
        pos = vision_position_ids(grid_thw, v["spatial_merge_size"])
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d // 2, 2, dtype=torch.float32) / (d // 2)))
        rot = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)
        emb = torch.cat([rot, rot], -1)
        cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)
        ln = lambda z, name: F.layer_norm(z, (vh,), w[name + ".weight"], w[name + ".bias"], 1e-6)
        for i in range(v["depth"]):
            b = f"visual.blocks.{i}."
            h = ln(x, b + "norm1")
            qkv = (h @ w[b + "attn.qkv.weight"].t() + w[b + "attn.qkv.bias"]).view(n, 3, nh, d)
            q, k, val = qkv[:, 0], qkv[:, 1], qkv[:, 2]
            q = (q.float() * cos + rotate_half(q.float()) * sin).to(x.dtype)
            k = (k.float() * cos + rotate_half(k.float()) * sin).to(x.dtype)
            a = _segment_attention(q, k, val, cu, d**-0.5).reshape(n, vh)
            x = x + (a @ w[b + "attn.proj.weight"].t() + w[b + "attn.proj.bias"])
            z = ln(x, b + "norm2") @ w[b + "mlp.fc1.weight"].t() + w[b + "mlp.fc1.bias"]
            x = x + ((z * torch.sigmoid(1.702 * z)) @ w[b + "mlp.fc2.weight"].t() + w[b + "mlp.fc2.bias"])
        h = ln(x, "visual.merger.ln_q").view(-1, vh * m2)
        h = F.gelu(h @ w["visual.merger.mlp.0.weight"].t() + w["visual.merger.mlp.0.bias"])
        merged = h @ w["visual.merger.mlp.2.weight"].t() + w["visual.merger.mlp.2.bias"]
        return (merged, x) if return_last_hidden else merged

    # -- vision tower: TF::408-471 ---------------------------------------------------------------
    def visual(self, pixel_values, grid_thw, return_last_hidden=False):
        v = self.cfg["vision"]
        if v.get("arch") == "qwen2_vl":
            return self._visual_qwen2vl(pixel_values, grid_thw, return_last_hidden)
        w = self.w
        vh, nh = v["hidden_size"], v["num_heads"]
        d = vh // nh
        m2 = v["spatial_merge_size"] ** 2
        x = pixel_values.to(w["visual.patch_embed.proj.weight"].dtype) @ w["visual.patch_embed.proj.weight"].reshape(vh, -1).t()
        n = x.shape[0]
        win_idx, cu_win = vision_window_index(grid_thw, v["spatial_merge_size"], v["window_size"], v["patch_size"])
        cu_full = vision_cu_seqlens(grid_thw)
        x = x.view(n // m2, m2, vh)[win_idx].reshape(n, vh)
        pos = vision_position_ids(grid_thw, v["spatial_merge_size"])
        inv_freq = 1.0 / (10000.0 ** (torch.arange(0, d // 2, 2, dtype=torch.float32) / (d // 2)))
        rot = (pos.unsqueeze(-1).float() * inv_freq).flatten(1)  # [N, d/2]
        rot = rot.view(n // m2, m2, -1)[win_idx].reshape(n, -1)
        emb = torch.cat([rot, rot], -1)
        cos, sin = emb.cos().unsqueeze(1), emb.sin().unsqueeze(1)  # [N,1,d]
        for i in range(v["depth"]):
            b = f"visual.blocks.{i}."
            h = rmsnorm(x, w[b + "norm1.weight"], 1e-6)
            qkv = (h @ w[b + "attn.qkv.weight"].t() + w[b + "attn.qkv.bias"]).view(n, 3, nh, d)
            q, k, val = qkv[:, 0], qkv[:, 1], qkv[:, 2]
            q = (q.float() * cos + rotate_half(q.float()) * sin).to(x.dtype)
            k = (k.float() * cos + rotate_half(k.float()) * sin).to(x.dtype)
            cu = cu_full if i in v["fullatt_block_indexes"] else cu_win
            a = _segment_attention(q, k, val, cu, d**-0.5).reshape(n, vh)
            x = x + (a @ w[b + "attn.proj.weight"].t() + w[b + "attn.proj.bias"])
            h = rmsnorm(x, w[b + "norm2.weight"], 1e-6)
            g = h @ w[b + "mlp.gate_proj.weight"].t() + w[b + "mlp.gate_proj.bias"]
            u = h @ w[b + "mlp.up_proj.weight"].t() + w[b + "mlp.up_proj.bias"]
            x = x + ((F.silu(g) * u) @ w[b + "mlp.down_proj.weight"].t() + w[b + "mlp.down_proj.bias"])
        h = rmsnorm(x, w["visual.merger.ln_q.weight"], 1e-6).view(-1, vh * m2)
        h = F.gelu(h @ w["visual.merger.mlp.0.weight"].t() + w["visual.merger.mlp.0.bias"])
        h = h @ w["visual.merger.mlp.2.weight"].t() + w["visual.merger.mlp.2.bias"]
        merged = h[torch.argsort(win_idx)]
        return (merged, x) if return_last_hidden else merged

    # -- text decoder: TF::790-873, layer ::708-757 ------------------------------------------------
    def _rope_cos_sin(self, position_ids):
        t = self.cfg["text"]
        hd = t["hidden_size"] // t["num_attention_heads"]
        inv_freq = 1.0 / (t["rope_theta"] ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
        freqs = position_ids.float().unsqueeze(-1) * inv_freq  # [3,B,S,hd/2]
        emb = torch.cat([freqs, freqs], -1)
        cos, sin = emb.cos(), emb.sin()
        sec = list(t["mrope_section"]) * 2
        cos = torch.cat([c[i % 3] for i, c in enumerate(cos.split(sec, -1))], -1)  # [B,S,hd]  TF::590-596
        sin = torch.cat([c[i % 3] for i, c in enumerate(sin.split(sec, -1))], -1)
        return cos, sin

    def embed(self, input_ids, image_embeds=None):
        x = self.w["model.embed_tokens.weight"][input_ids]
        if image_embeds is not None:
            mask = input_ids == self.cfg["image_token_id"]
            assert int(mask.sum()) == image_embeds.shape[0], "image tokens != image features"
            x = x.masked_scatter(mask.unsqueeze(-1), image_embeds.to(x.dtype))  # TF::1210-1215
        return x

    def text_model(self, x, attention_mask, position_ids, return_hidden=False):
        t = self.cfg["text"]
        w = self.w
        B, S, H = x.shape
        nh, nkv = t["num_attention_heads"], t["num_key_value_heads"]
        hd = H // nh
        cos, sin = self._rope_cos_sin(position_ids)
        cos, sin = cos.unsqueeze(1).to(x.dtype), sin.unsqueeze(1).to(x.dtype)
        causal = torch.ones(S, S, dtype=torch.bool).tril()
        allowed = causal.view(1, 1, S, S) & attention_mask.bool().view(B, 1, 1, S)
        hiddens = [x]
        for i in range(t["num_hidden_layers"]):
            b = f"model.layers.{i}."
            h = rmsnorm(x, w[b + "input_layernorm.weight"], t["rms_norm_eps"])
            q = (h @ w[b + "self_attn.q_proj.weight"].t() + w[b + "self_attn.q_proj.bias"]).view(B, S, nh, hd).transpose(1, 2)
            k = (h @ w[b + "self_attn.k_proj.weight"].t() + w[b + "self_attn.k_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
            v = (h @ w[b + "self_attn.v_proj.weight"].t() + w[b + "self_attn.v_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
            q = q * cos + rotate_half(q) * sin
            k = k * cos + rotate_half(k) * sin
            k = k.repeat_interleave(nh // nkv, 1)
            v = v.repeat_interleave(nh // nkv, 1)
            s = (q @ k.transpose(2, 3)) * hd**-0.5
            s = s.masked_fill(~allowed, torch.finfo(s.dtype).min)
            p = torch.softmax(s, -1, dtype=torch.float32).to(x.dtype)
            a = (p @ v).transpose(1, 2).reshape(B, S, H)
            x = x + a @ w[b + "self_attn.o_proj.weight"].t()
            h = rmsnorm(x, w[b + "post_attention_layernorm.weight"], t["rms_norm_eps"])
            g = h @ w[b + "mlp.gate_proj.weight"].t()
            u = h @ w[b + "mlp.up_proj.weight"].t()
            x = x + (F.silu(g) * u) @ w[b + "mlp.down_proj.weight"].t()
            hiddens.append(x)
        x = rmsnorm(x, w["model.norm.weight"], t["rms_norm_eps"])
        return (x, hiddens) if return_hidden else x

    def hidden_states(self, input_ids, attention_mask, pixel_values=None, image_grid_thw=None, return_hidden=False):
        grids = [tuple(int(z) for z in g) for g in image_grid_thw] if image_grid_thw is not None else []
        img = self.visual(pixel_values, grids) if pixel_values is not None else None
        x = self.embed(input_ids, img)
        pos, _ = mrope_position_ids(input_ids, attention_mask, grids, self.cfg["image_token_id"], self.cfg["vision"]["spatial_merge_size"])
        return self.text_model(x, attention_mask, pos, return_hidden)

    def logits(self, input_ids, attention_mask, pixel_values=None, image_grid_thw=None):
        h = self.hidden_states(input_ids, attention_mask, pixel_values, image_grid_thw)
        return h @ self.w["lm_head.weight"].t()

    # -- the hot path's consumer: sc_grpo_trainer.py:505-514 -----------------------------------------
    def per_token_logps(self, input_ids, attention_mask, pixel_values=None, image_grid_thw=None):
        """log_softmax(logits[:, :-1]) gathered at input_ids[:, 1:]  -> [B, S-1]; no temperature."""
        lg = self.logits(input_ids, attention_mask, pixel_values, image_grid_thw)[:, :-1]
        lp = torch.log_softmax(lg.float(), -1)
        return lp.gather(-1, input_ids[:, 1:].unsqueeze(-1)).squeeze(-1)

    # -- PA-SFT loss: TF:loss/loss_utils.py:32-71 (shift, ignore_index=-100, mean over kept) --------
    def sft_loss(self, input_ids, attention_mask, labels, pixel_values=None, image_grid_thw=None, num_items_in_batch=None):
        lg = self.logits(input_ids, attention_mask, pixel_values, image_grid_thw).float()
        shift = F.pad(labels, (0, 1), value=-100)[:, 1:]
        if num_items_in_batch is None:
            return F.cross_entropy(lg.view(-1, lg.shape[-1]), shift.reshape(-1), ignore_index=-100, reduction="mean")
        return F.cross_entropy(lg.view(-1, lg.shape[-1]), shift.reshape(-1), ignore_index=-100, reduction="sum") / num_items_in_batch

    # -- greedy rollout (full recompute per step; the bit-exact token-id target) --------------------
    @torch.no_grad()
    def greedy_generate(self, prompt_ids, prompt_mask, pixel_values, image_grid_thw, max_new_tokens, eos_token_id=None, pad_token_id=0):
        ids, mask = prompt_ids.clone(), prompt_mask.clone()
        done = torch.zeros(ids.shape[0], dtype=torch.bool)
        for _ in range(max_new_tokens):
            lg = self.logits(ids, mask, pixel_values, image_grid_thw)[:, -1]
            nxt = lg.argmax(-1)
            if eos_token_id is not None:
                nxt = torch.where(done, torch.full_like(nxt, pad_token_id), nxt)
                done |= nxt == eos_token_id
            ids = torch.cat([ids, nxt.view(-1, 1)], 1)
            mask = torch.cat([mask, torch.ones_like(nxt).view(-1, 1)], 1)
        return ids


    # -- greedy rollout with a key/value cache: the arithmetic HF `generate` / vLLM actually run (prefill once, then one new token per step
    #    attending to cached K/V: TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:641-689 with `past_key_values`, positions for later tokens = cache
    #    length + rope_delta, TF::1165-1176).  Same values as greedy_generate up to fp32 summation order; tests/test_oracle_model.py checks the
    #    token ids against it and against the HF golden.  Used by bench.py's cpu_baseline leg (the full-recompute form is quadratically slower) ----
    @torch.no_grad()
    def _layers_cached(self, x, allowed, cos, sin, cache):
        """x: [B, Sn, H] new tokens; allowed: [B, 1, Sn, Sk] bool over (cached + new) keys; cache: list of [k, v] per layer (appended in place)."""
        t, w = self.cfg["text"], self.w
        B, Sn, H = x.shape
        nh, nkv = t["num_attention_heads"], t["num_key_value_heads"]
        hd = H // nh
        cos, sin = cos.unsqueeze(1).to(x.dtype), sin.unsqueeze(1).to(x.dtype)
        for i in range(t["num_hidden_layers"]):
            b = f"model.layers.{i}."
            h = rmsnorm(x, w[b + "input_layernorm.weight"], t["rms_norm_eps"])
            q = (h @ w[b + "self_attn.q_proj.weight"].t() + w[b + "self_attn.q_proj.bias"]).view(B, Sn, nh, hd).transpose(1, 2)
            k = (h @ w[b + "self_attn.k_proj.weight"].t() + w[b + "self_attn.k_proj.bias"]).view(B, Sn, nkv, hd).transpose(1, 2)
            v = (h @ w[b + "self_attn.v_proj.weight"].t() + w[b + "self_attn.v_proj.bias"]).view(B, Sn, nkv, hd).transpose(1, 2)
            q = q * cos + rotate_half(q) * sin
            k = k * cos + rotate_half(k) * sin
            if cache[i] is None:
                cache[i] = [k, v]
            else:
                cache[i] = [torch.cat([cache[i][0], k], 2), torch.cat([cache[i][1], v], 2)]
            kk = cache[i][0].repeat_interleave(nh // nkv, 1)
            vv = cache[i][1].repeat_interleave(nh // nkv, 1)
            sc = (q @ kk.transpose(2, 3)) * hd**-0.5
            sc = sc.masked_fill(~allowed, torch.finfo(sc.dtype).min)
            pr = torch.softmax(sc, -1, dtype=torch.float32).to(x.dtype)
            a = (pr @ vv).transpose(1, 2).reshape(B, Sn, H)
            x = x + a @ w[b + "self_attn.o_proj.weight"].t()
            h = rmsnorm(x, w[b + "post_attention_layernorm.weight"], t["rms_norm_eps"])
            x = x + (F.silu(h @ w[b + "mlp.gate_proj.weight"].t()) * (h @ w[b + "mlp.up_proj.weight"].t())) @ w[b + "mlp.down_proj.weight"].t()
        return rmsnorm(x, w["model.norm.weight"], t["rms_norm_eps"])

    @torch.no_grad()
    def prefill_cached(self, prompt_ids, prompt_mask, pixel_values, image_grid_thw):
        """-> (logits of the last prompt position [B, V], state) with state = {cache, mask, rope_deltas}."""
        grids = [tuple(int(z) for z in g) for g in image_grid_thw] if image_grid_thw is not None else []
        img = self.visual(pixel_values, grids) if pixel_values is not None else None
        x = self.embed(prompt_ids, img)
        pos, deltas = mrope_position_ids(prompt_ids, prompt_mask, grids, self.cfg["image_token_id"], self.cfg["vision"]["spatial_merge_size"])
        B, S = prompt_ids.shape
        allowed = torch.ones(S, S, dtype=torch.bool).tril().view(1, 1, S, S) & prompt_mask.bool().view(B, 1, 1, S)
        cos, sin = self._rope_cos_sin(pos)
        cache = [None] * self.cfg["text"]["num_hidden_layers"]
        h = self._layers_cached(x, allowed, cos, sin, cache)
        return h[:, -1] @ self.w["lm_head.weight"].t(), {"cache": cache, "mask": prompt_mask.clone(), "deltas": torch.as_tensor(deltas).view(-1).long()}

    @torch.no_grad()
    def decode_step_cached(self, tokens, state):
        """One new token per sequence -> logits [B, V]; K/V appended to state["cache"]."""
        B = tokens.shape[0]
        n_real = state["mask"].sum(1)                                  # tokens attended so far (left padding excluded)
        pos = (n_real + state["deltas"]).view(1, B, 1).expand(3, B, 1)  # text token: the same position on all three M-RoPE axes
        state["mask"] = torch.cat([state["mask"], torch.ones(B, 1, dtype=state["mask"].dtype)], 1)
        allowed = state["mask"].bool().view(B, 1, 1, -1)
        cos, sin = self._rope_cos_sin(pos)
        x = self.w["model.embed_tokens.weight"][tokens].view(B, 1, -1)
        h = self._layers_cached(x, allowed, cos, sin, state["cache"])
        return h[:, -1] @ self.w["lm_head.weight"].t()

    @torch.no_grad()
    def greedy_generate_cached(self, prompt_ids, prompt_mask, pixel_values, image_grid_thw, max_new_tokens, eos_token_id=None, pad_token_id=0):
        """Same contract as greedy_generate (returns [B, P + max_new_tokens] ids)."""
        lg, st = self.prefill_cached(prompt_ids, prompt_mask, pixel_values, image_grid_thw)
        ids = prompt_ids.clone()
        done = torch.zeros(ids.shape[0], dtype=torch.bool)
        for it in range(max_new_tokens):
            nxt = lg.argmax(-1)
            if eos_token_id is not None:
                nxt = torch.where(done, torch.full_like(nxt, pad_token_id), nxt)
                done |= nxt == eos_token_id
            ids = torch.cat([ids, nxt.view(-1, 1)], 1)
            if it + 1 < max_new_tokens:
                lg = self.decode_step_cached(nxt, st)
        return ids


def flops_per_sequence(cfg: dict, S: int, n_patches: int, logits_positions: int | None = None) -> dict:
    """Algorithmic forward FLOPs of one sequence (GEMM 2*params*tokens + attention), SURVEY section 8(d)."""
    t, v = cfg["text"], cfg["vision"]
    h, inter, L = t["hidden_size"], t["intermediate_size"], t["num_hidden_layers"]
    hd = h // t["num_attention_heads"]
    kvd = hd * t["num_key_value_heads"]
    per_layer = h * (h + 2 * kvd) + h * h + 3 * h * inter
    llm_gemm = 2.0 * per_layer * L * S
    llm_attn = 4.0 * S * S * h / 2 * L
    head = 2.0 * h * t["vocab_size"] * (logits_positions if logits_positions is not None else S)
    vh, vi = v["hidden_size"], v["intermediate_size"]
    kpe = v["in_channels"] * v["temporal_patch_size"] * v["patch_size"] ** 2
    vit_layer = 4 * vh * vh + 3 * vh * vi
    vit_gemm = 2.0 * n_patches * (kpe * vh + vit_layer * v["depth"]) + 2.0 * (n_patches / 4) * ((4 * vh) ** 2 + 4 * vh * v["out_hidden_size"])
    return {"llm_gemm": llm_gemm, "llm_attn": llm_attn, "lm_head": head, "vit": vit_gemm}
