#!/usr/bin/env python3
"""Golden for the reference's ABLATION reward functions (/root/reference/train/stage_rl/reward.py:107-347: consistency_reward_cot, format_consistency_reward_cot,
accuracy_reward_cot_wo_type, accuracy_reward_cot_wo_location, format_reward_cot_base, accuracy_reward_cot_base, wo_format).  None of them is registered by the entry
point (grpo_ad.py:126-129 registers `accuracy` and `format` only), but they are part of the reward module's surface: a user who swaps one into `reward_funcs_registry`
finds it here under the same name.  Imports the reference's reward module with the recipe of tools/make_golden.py and runs every function on that tool's case table
plus a table with <description> tags (the `_cot` variants count three tags).  Run here (build container) only."""
import contextlib
import io
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

NAMES = ["consistency_reward_cot", "format_consistency_reward_cot", "accuracy_reward_cot_wo_type", "accuracy_reward_cot_wo_location", "format_reward_cot_base",
         "accuracy_reward_cot_base"]


def extra_cases():
    comps, sols = [], []
    tags = {"t": "<type>scratch</type>", "l": "<location>top left</location>", "d": "<description>a thin line</description>"}
    sol_y, sol_n = "<think>g</think><location>upper left</location><type>scratch</type><answer>yes</answer>", "<think>g</think><answer>no</answer>"
    for combo in ("", "t", "l", "d", "tl", "td", "ld", "tld", "TLD"):
        body = "".join(tags[c.lower()].upper() if c.isupper() else tags[c] for c in combo)
        for ans in ("yes", "no", "Yes", " NO ", "maybe"):
            for sol in (sol_y, sol_n, "yes", "No", "<ANSWER>yes</ANSWER>"):
                comps.append(f"<think>x</think>{body}<answer>{ans}</answer>")
                sols.append(sol)
    comps += ["<think>a\nb</think><answer>no</answer>", "<think>a</think>\n<answer>no</answer>", "x<think>a</think><answer>no</answer>y", "<think>a</think><answer>no</answer>",
              "<type>multi\nline</type><answer>yes</answer>", "<answer>yes</answer>", ""]
    sols += [sol_n] * 4 + [sol_y] * 3
    return comps, sols


def main():
    reward = mg.import_reference()[0]
    c1, s1 = mg.build_reward_cases()
    c2, s2 = extra_cases()
    comps, sols = c1[::3] + c2, s1[::3] + s2
    wrapped = [[{"role": "assistant", "content": c}] for c in comps]
    out = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for n in NAMES:
            out[n] = getattr(reward, n)(wrapped, sols)
            assert len(out[n]) == len(comps)
        out["wo_format"] = reward.wo_format(wrapped, sols)
    path = os.path.join(mg.OUT, "reward_variants.json")
    json.dump({"meta": {"source": "train/stage_rl/reward.py:107-347 of Yanhui-Lee/IAD-R1, imported and executed", "generator": "tools/make_golden_reward_variants.py"},
               "completions": comps, "solutions": sols, "values": out}, open(path, "w"), indent=0)
    print(f"wrote {path}: {len(comps)} cases x {len(NAMES)} functions; wo_format -> {out['wo_format']!r}")


if __name__ == "__main__":
    main()
