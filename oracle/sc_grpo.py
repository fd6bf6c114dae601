"""ORACLE (test infrastructure, not product): restatement of the arithmetic of
`SCGRPOTrainer.compute_loss`, /root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:586-819,
with the model forward supplied by oracle.qwen25vl.  Pinned by tests/golden/sc_grpo_g{4,8}.npz,
which were captured from the reference's own compute_loss (tools/make_golden.py).
"""
from __future__ import annotations

import torch


def right_pad(rows, pad_value):
    """trl/trl/trainer/utils.py:418-478 `pad` for 1-D rows, padding_side='right'."""
    m = max(len(r) for r in rows)
    out = torch.full((len(rows), m), pad_value, dtype=torch.long)
    for i, r in enumerate(rows):
        out[i, : len(r)] = torch.as_tensor(r, dtype=torch.long)
    return out


def eos_completion_mask(completion_ids, eos_token_id):
    """sc_grpo_trainer.py:722-726: keep everything up to and including the first EOS."""
    is_eos = completion_ids == eos_token_id
    n, c = is_eos.shape
    eos_idx = torch.full((n,), c, dtype=torch.long)
    has = is_eos.any(1)
    eos_idx[has] = is_eos.int().argmax(1)[has]
    return (torch.arange(c).expand(n, -1) <= eos_idx.unsqueeze(1)).int()


def group_advantages(rewards, G):
    """sc_grpo_trainer.py:787-793: per-group mean, UNBIASED std, eps 1e-4; groups are interleaved."""
    r = rewards.view(-1, G)
    mean = r.mean(1).repeat_interleave(G)
    std = r.std(1).repeat_interleave(G)
    return (rewards - mean) / (std + 1e-4), std


def grpo_loss(logps, ref_logps, advantages, completion_mask, beta):
    """sc_grpo_trainer.py:746,796-798,816.  Returns (loss, per_token_kl, mean_kl)."""
    kl = torch.exp(ref_logps - logps) - (ref_logps - logps) - 1
    ptl = torch.exp(logps - logps.detach()) * advantages.unsqueeze(1)
    ptl = -(ptl - beta * kl)
    m = completion_mask
    loss = ((ptl * m).sum(1) / m.sum(1)).mean()
    mean_kl = ((kl * m).sum(1) / m.sum(1)).mean()
    return loss, kl, mean_kl


def sc_grpo_step(policy, ref, prompt_ids, prompt_mask, pixel_values, image_grid_thw, completions, rewards_per_func, G, beta, eos_token_id, pad_token_id,
                 max_prompt_length=None, interleaved=False, images_per_prompt=None, rotate_right_padded_rows=False):
    """One micro-step for B prompts x G completions.  Default: TILE order for tensors as the reference (sc_grpo_trainer.py:625-628; equal to
    interleaved at B=1, the only batch size its scripts use).  interleaved=True: prompt-major order (p0 x G, p1 x G, ...) everywhere -- the
    consistent reading of SURVEY.md Appendix B.1 the engine uses for B > 1; `completions` / `rewards_per_func` are then prompt-major too.
    max_prompt_length: the left truncation of sc_grpo_trainer.py:630-634 (ids and mask only).
    rotate_right_padded_rows: the llava branches of `_get_per_token_logps` (sc_grpo_trainer.py:502-504 -> :516-567): rows that end in padding and carry
    no left padding are rotated before the model runs, while the log-probs are still sliced and masked at the un-rotated columns.
    `completions` = list of id lists, `rewards_per_func` = [B*G, n_funcs] already evaluated on the decoded strings."""
    grids_p = [tuple(int(z) for z in g) for g in image_grid_thw]
    if max_prompt_length is not None:
        prompt_ids, prompt_mask = prompt_ids[:, -max_prompt_length:], prompt_mask[:, -max_prompt_length:]
    if interleaved:
        B = prompt_ids.shape[0]
        ipp = images_per_prompt or [1] * B
        p_ids, p_mask = prompt_ids.repeat_interleave(G, 0), prompt_mask.repeat_interleave(G, 0)
        n_patch = [g[0] * g[1] * g[2] for g in grids_p]
        blocks, grids, k, r = [], [], 0, 0
        for b in range(B):
            n = sum(n_patch[k: k + ipp[b]])
            blocks += [pixel_values[r: r + n]] * G
            grids += grids_p[k: k + ipp[b]] * G
            k, r = k + ipp[b], r + n
        pv = torch.cat(blocks, 0)
    else:
        rep = lambda t: t.repeat(G, *[1] * (t.dim() - 1))
        p_ids, p_mask, pv = rep(prompt_ids), rep(prompt_mask), rep(pixel_values)
        grids = grids_p * G
    comp = right_pad(completions, pad_token_id)
    cmask = eos_completion_mask(comp, eos_token_id)
    ids = torch.cat([p_ids, comp], 1)
    mask = torch.cat([p_mask, cmask], 1)
    P = p_ids.shape[1]
    m_ids, m_mask = ids, mask
    if rotate_right_padded_rows:
        from .llava_ov import ensure_left_padding
        m_ids, m_mask = ensure_left_padding(ids, mask, pad_token_id)
    logps = policy.per_token_logps(m_ids, m_mask, pv, grids)[:, P - 1:]
    with torch.no_grad():
        ref_logps = ref.per_token_logps(m_ids, m_mask, pv, grids)[:, P - 1:]
    rewards = rewards_per_func.sum(1)
    adv, std = group_advantages(rewards, G)
    loss, kl, mean_kl = grpo_loss(logps, ref_logps, adv, cmask, beta)
    metrics = {
        "completion_length": cmask.sum(1).float().mean().item(),
        "reward": rewards.mean().item(),
        "reward_std": std.mean().item(),
        "kl": mean_kl.item(),
    }
    return {"loss": loss, "logps": logps, "ref_logps": ref_logps, "kl": kl, "advantages": adv, "completion_mask": cmask, "ids": ids, "mask": mask, "metrics": metrics}
