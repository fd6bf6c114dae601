"""PA-SFT batch construction against vectors produced by the reference's own functions (tools/make_golden_sft_data.py)."""
import json
import os

import iadr1_amd  # noqa: F401
from iadr1_amd.sft_data import IGNORE_INDEX, supervised_labels, turn_budget

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sft_data.json")))


def test_turn_budget_matches_reference():
    assert len(G["infer_seqlen"]) >= 100
    for s, t, c, want in G["infer_seqlen"]:
        assert list(turn_budget(s, t, c)) == want, (s, t, c)


def test_supervised_labels_match_reference():
    assert len(G["encode"]) >= 90
    for c in G["encode"]:
        ids, labels = supervised_labels([tuple(p) for p in c["pairs"]], c["cutoff_len"], c["eos_token_id"], c["train_on_prompt"], c["mask_history"],
                                        c["efficient_eos"])
        assert ids == c["input_ids"], c
        assert labels == c["labels"], c


def test_supervised_labels_properties():
    turns = [(list(range(10, 40)), list(range(100, 125))), (list(range(40, 49)), list(range(200, 206)))]
    for cutoff in (1, 5, 16, 55, 70, 1000):
        ids, labels = supervised_labels(turns, cutoff)
        assert len(ids) == len(labels) <= cutoff
        assert all(l == IGNORE_INDEX or l == i for i, l in zip(ids, labels))
    ids, labels = supervised_labels(turns, 1000)
    assert ids == turns[0][0] + turns[0][1] + turns[1][0] + turns[1][1]
    assert [l for l in labels if l != IGNORE_INDEX] == turns[0][1] + turns[1][1]
    assert supervised_labels([], 10) == ([], [])
    assert supervised_labels(turns, 0) == ([], [])
