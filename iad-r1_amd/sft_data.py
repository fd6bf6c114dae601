"""PA-SFT batch construction: how one multi-turn conversation becomes (input_ids, labels) under a token budget.

Host-side integer logic, no device work.  Follows the reference's LLaMA-Factory fork:
  * `turn_budget`       <- infer_seqlen, train/stage_sft/llamafactory/data/processors/processor_utils.py:51-65
  * `supervised_labels` <- _encode_supervised_example, train/stage_sft/llamafactory/data/processors/supervised.py:33-87
Pinned bit-exactly against those two functions by tests/golden/sft_data.json (tools/make_golden_sft_data.py)."""
from typing import List, Sequence, Tuple

IGNORE_INDEX = -100


def turn_budget(source_len: int, target_len: int, budget: int) -> Tuple[int, int]:
    """How many prompt / answer tokens of one turn survive when only `budget` tokens are left.

    A short answer (less than half the budget) is kept whole and the prompt gives way; otherwise a short prompt is kept whole and the answer gives way;
    when both are long they share the budget in proportion to their lengths (answer share rounded down, prompt gets the remainder)."""
    if 2 * target_len < budget:
        target_cap = budget
    elif 2 * source_len < budget:
        target_cap = budget - source_len
    else:
        target_cap = int(budget * (target_len / (source_len + target_len)))
    keep_target = target_len if target_len < target_cap else target_cap
    room = budget - keep_target
    keep_source = min(source_len, room if room > 0 else 0)
    return keep_source, keep_target


def supervised_labels(turns: Sequence[Tuple[Sequence[int], Sequence[int]]], cutoff_len: int, eos_token_id: int = None, train_on_prompt: bool = False,
                      mask_history: bool = False, efficient_eos: bool = False) -> Tuple[List[int], List[int]]:
    """turns = [(prompt_ids, answer_ids)] per conversation turn  ->  (input_ids, labels), both at most cutoff_len long.

    Turns are admitted in order (newest first under `mask_history`, so old turns are the ones dropped) until the budget is used up, each trimmed by
    `turn_budget` against what is left.  Prompt tokens are labelled IGNORE_INDEX unless `train_on_prompt`; with `mask_history` only the last turn's answer
    is supervised.  `efficient_eos` templates omit the per-turn eos from the rendered text: the first prompt position of every turn is labelled eos and
    one eos is appended at the end (a slot reserved for it up front)."""
    used = 1 if efficient_eos else 0
    pieces = []     # (ids, labels) per admitted turn, admission order
    order = list(turns)[::-1] if mask_history else list(turns)
    for n, (src, tgt) in enumerate(order):
        if used >= cutoff_len:
            break
        ks, kt = turn_budget(len(src), len(tgt), cutoff_len - used)
        src, tgt = list(src[:ks]), list(tgt[:kt])
        used += ks + kt
        if train_on_prompt:
            src_lab = list(src)
        else:
            src_lab = [IGNORE_INDEX] * ks
            if efficient_eos and ks:
                src_lab[0] = eos_token_id
        tgt_lab = [IGNORE_INDEX] * kt if (mask_history and n) else list(tgt)
        pieces.append((src + tgt, src_lab + tgt_lab))
    if mask_history:
        pieces.reverse()
    ids = [t for p in pieces for t in p[0]]
    labels = [t for p in pieces for t in p[1]]
    if efficient_eos:
        ids.append(eos_token_id)
        labels.append(eos_token_id)
    return ids, labels
