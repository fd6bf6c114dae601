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


# ------------------------------------------------------------------------------------------------ rows -> turns -> text (tests/golden/sft_text.json)
import pytest

from iadr1_amd.sft_data import (ShareGPTSchema, align_sharegpt, encode_turns, expand_image_placeholders, qwen2_vl_turn_texts, regular_image_size)

T = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sft_text.json")))


class _CharTok:
    def encode(self, text, add_special_tokens=False):
        return [ord(c) for c in text]


def test_align_sharegpt_matches_reference():
    kept = 0
    for c in T["rows"]:
        got = align_sharegpt(c["example"], ShareGPTSchema.from_dataset_info(c["entry"]), image_dir="/nonexistent")
        assert got == c["aligned"], c["example"]
        kept += bool(got["prompt"])
    assert kept >= 6 and kept < len(T["rows"])      # both well-formed rows and rows the reference drops are covered


def test_qwen2_vl_template_matches_reference():
    n = 0
    for c in T["rows"]:
        if "pairs" not in c:
            continue
        al = c["aligned"]
        pairs = encode_turns(_CharTok(), qwen2_vl_turn_texts(al["prompt"] + al["response"], al["system"]))
        assert [[s, t] for s, t in pairs] == c["pairs"], c["example"]
        n += 1
    assert n >= 6
    first = "".join(map(chr, T["rows"][0]["pairs"][0][0]))
    assert first.startswith("<|im_start|>system\n" + T["meta"]["default_system"] + "<|im_end|>\n<|im_start|>user\n")


def test_image_regularisation_matches_reference():
    for w, h, res, rw, rh, _mode in T["image_sizes"]:
        assert regular_image_size(w, h, res) == (rw, rh), (w, h, res)


def test_image_placeholder_expansion_matches_reference():
    for c in T["expand"]:
        assert expand_image_placeholders(c["messages"], c["grids"]) == c["expanded"]
    for c in T["expand_errors"]:
        with pytest.raises(ValueError) as e:
            expand_image_placeholders(c["messages"], c["grids"])
        assert str(e.value) == c["error"]


def test_regularize_image_pixels_match_reference():
    import hashlib

    import numpy as np
    from PIL import Image

    from iadr1_amd.sft_data import regularize_image
    for c in T["pixels"]:
        w, h = c["size"]
        arr = np.random.RandomState(c["seed"]).randint(0, 256, size=(h, w) + ((len(c["mode"]),) if len(c["mode"]) > 1 else ()), dtype=np.uint8)
        im = regularize_image(Image.fromarray(arr, mode=c["mode"]), c["max_pixels"])
        assert im.mode == "RGB" and [im.width, im.height] == c["out_size"]
        assert hashlib.sha1(im.tobytes()).hexdigest() == c["sha1"], c
