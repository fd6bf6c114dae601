"""Host-side reward functions vs values captured from the reference (bit-exact floats)."""
import json
import os

import iadr1_amd  # noqa: F401
from iadr1_amd import rewards


def test_rewards_bit_exact(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "rewards.json")))
    comps = [[{"role": "assistant", "content": c}] for c in g["completions"]]
    acc = rewards.accuracy_reward(comps, g["solutions"], prompts=None, current_step=0)
    assert len(acc) == len(g["accuracy"]) >= 200
    bad = [(i, a, b) for i, (a, b) in enumerate(zip(acc, g["accuracy"])) if a != b]
    assert not bad, bad[:5]
    fmt = rewards.consistency_reward(comps, g["solutions"])
    assert fmt == g["format"]
    got = [rewards.type_score(a, b) for a, b in g["type_pairs"]]
    bad = [(p, x, y) for p, x, y in zip(g["type_pairs"], got, g["type_scores"]) if x != y]
    assert not bad, bad[:5]
    assert [rewards.location_score(a, b) for a, b in g["location_pairs"]] == g["location_scores"]


def test_format_quirk_neither_yes_no(golden_dir):
    g = json.load(open(os.path.join(golden_dir, "rewards.json")))
    out = rewards.consistency_reward([[{"role": "assistant", "content": "x"}]] * 2, ["maybe", "<answer>no</answer>"])
    assert len(out) == g["format_len_quirk"] == 1


def test_registry_keys():
    assert set(rewards.REWARD_FUNCS) == {"accuracy", "format"}


def test_ablation_reward_variants_bit_exact(golden_dir):
    """The reference's unregistered ablation rewards (REF reward.py:107-347) under the same names: value for value on 708 cases captured by importing and running the
    reference (tools/make_golden_reward_variants.py), including its failure modes (a missing tag zeroes the whole sample, `wo_format` returns the int 0)."""
    g = json.load(open(os.path.join(golden_dir, "reward_variants.json")))
    comps = [[{"role": "assistant", "content": c}] for c in g["completions"]]
    assert len(comps) >= 600
    for name, want in g["values"].items():
        got = getattr(rewards, name)(comps, g["solutions"])
        if name == "wo_format":
            assert got == want == 0 and isinstance(got, int)
            continue
        bad = [(i, a, b) for i, (a, b) in enumerate(zip(got, want)) if a != b]
        assert len(got) == len(want) and not bad, (name, bad[:5])
