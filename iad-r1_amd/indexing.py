"""Host-side integer index plans for the Qwen2.5-VL hot path (numpy, no torch autograd, no GPU):
ViT window permutation + attention segments, 2-D rotary positions, M-RoPE position ids.

Behaviour follows the pinned dependency of the reference (transformers; cited as TF: with 5.15.0 line
numbers): TF:vision_utils.py:130-185 (window index), :81-127 (vision position ids), :41-65 (cu_seqlens),
TF:models/qwen2_5_vl/modeling_qwen2_5_vl.py:944-1062 (get_rope_index).  Checked bit-exactly against
tests/golden/vision_index.json (captured from that implementation)."""
from __future__ import annotations

import numpy as np


def vision_window_index(grid_thw, spatial_merge_size=2, window_size=112, patch_size=14):
    """-> (window_index [sum t*h*w/m^2] int64, cu_window_seqlens list[int] in patches)."""
    win = window_size // spatial_merge_size // patch_size
    unit = spatial_merge_size**2
    parts, cu, base = [], [0], 0
    for t, h, w in grid_thw:
        lh, lw = h // spatial_merge_size, w // spatial_merge_size
        ph, pw = win - lh % win, win - lw % win  # a full extra window when already aligned (kept: it only yields empty windows)
        nh, nw = (lh + ph) // win, (lw + pw) // win
        idx = np.full((t, lh + ph, lw + pw), -1, dtype=np.int64)
        idx[:, :lh, :lw] = np.arange(t * lh * lw).reshape(t, lh, lw)
        idx = idx.reshape(t, nh, win, nw, win).transpose(0, 1, 3, 2, 4).reshape(t * nh * nw, win * win)
        counts = (idx >= 0).sum(1)
        flat = idx.reshape(-1)
        parts.append(flat[flat >= 0] + base)
        for c in np.cumsum(counts) * unit + cu[-1]:
            if c != cu[-1]:
                cu.append(int(c))
        base += t * lh * lw
    return np.concatenate(parts), cu


def vision_position_ids(grid_thw, spatial_merge_size=2):
    """-> [N, 2] (row, col) of every patch, patches in merge-block-major order."""
    m = spatial_merge_size
    out = []
    for t, h, w in grid_thw:
        hp = np.broadcast_to(np.arange(h)[:, None], (h, w)).reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
        wp = np.broadcast_to(np.arange(w)[None, :], (h, w)).reshape(h // m, m, w // m, m).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], -1), (t, 1)))
    return np.concatenate(out, 0)


def vision_cu_seqlens(grid_thw):
    cu = [0]
    for t, h, w in grid_thw:
        for _ in range(t):
            cu.append(cu[-1] + h * w)
    return cu


def mrope_position_ids(input_ids: np.ndarray, attention_mask: np.ndarray, grid_thw, image_token_id: int, spatial_merge_size=2):
    """-> (pos [3,B,S] int64, rope_deltas [B] int64).  Padded slots keep 0."""
    B, S = input_ids.shape
    pos = np.zeros((3, B, S), dtype=np.int64)
    deltas = np.zeros(B, dtype=np.int64)
    gi = 0
    for b in range(B):
        keep = attention_mask[b].astype(bool)
        ids = input_ids[b][keep]
        n = len(ids)
        is_img = ids == image_token_id
        # run boundaries
        edges = np.flatnonzero(np.diff(is_img.astype(np.int8))) + 1
        starts = np.concatenate([[0], edges])
        ends = np.concatenate([edges, [n]])
        chunks, cur = [], 0
        for s, e in zip(starts, ends):
            if not is_img[s]:
                ln = e - s
                chunks.append(np.broadcast_to(np.arange(ln) + cur, (3, ln)))
                cur += ln
            else:
                t, h, w = grid_thw[gi]
                gi += 1
                lh, lw = h // spatial_merge_size, w // spatial_merge_size
                if e - s != t * lh * lw:
                    raise ValueError(f"row {b}: {e - s} image-pad tokens but grid {t}x{h}x{w} needs {t * lh * lw}")
                tt = np.repeat(np.arange(t), lh * lw)
                hh = np.tile(np.repeat(np.arange(lh), lw), t)
                ww = np.tile(np.arange(lw), t * lh)
                chunks.append(np.stack([tt, hh, ww]) + cur)
                cur += max(h, w) // spatial_merge_size
        p = np.concatenate(chunks, 1) if chunks else np.zeros((3, 0), dtype=np.int64)
        pos[:, b, keep] = p
        deltas[b] = (int(p.max()) + 1 if n else 0) - n
    return pos, deltas


def mrope_component_of_channel(mrope_section, half_dim):
    """Which of (t,h,w) drives rotary frequency j (TF::590-596: sections [16,24,24] over 64 freqs)."""
    comp = np.concatenate([np.full(n, i % 3) for i, n in enumerate(mrope_section)])
    assert len(comp) == half_dim, (mrope_section, half_dim)
    return comp
