"""Evaluation-harness host logic vs golden vectors produced by the reference's own functions (tools/make_golden_eval.py)."""
import io
import json
import os

import numpy as np
import pytest

import iadr1_amd  # noqa: F401
from iadr1_amd import evaluate


def _g(golden_dir):
    return json.load(open(os.path.join(golden_dir, "eval.json")))


def test_get_ans_matches_reference(golden_dir):
    cases = _g(golden_dir)["get_ans"]
    assert len(cases) >= 50
    for c in cases:
        assert evaluate.get_ans(c["response"], c["options"]) == c["expected"], c


def test_parse_conversation_matches_reference(golden_dir):
    for c in _g(golden_dir)["parse_conversation"]:
        qs, ans = evaluate.parse_conversation(c["text_gt"])
        assert qs == c["questions"] and ans == c["answers"]


def test_accuracy_table_csv_is_bit_exact(golden_dir):
    acc = _g(golden_dir)["accuracy"]
    for flag, key in ((False, "csv"), (True, "csv_overkill_miss")):
        df, stats = evaluate.accuracy_table(acc["answers"], show_overkill_miss=flag)
        buf = io.StringIO()
        df.to_csv(buf)
        assert buf.getvalue() == acc[key]
        assert stats == acc["question_stats"]


def test_prompts_of_the_three_scripts_match_the_reference(golden_dir):
    """`build_messages` / `template_kwargs` per family against what the reference's own `build_prompt` methods hand to apply_chat_template
    (vLLM_Qwen_detect_format.py, vLLM_LLaVA_detect_format.py, vLLM_LLaVA_1_5_detect_format.py; captured by tools/make_golden_eval.py): 0-, 1- and 2-shot."""
    cases = _g(golden_dir)["prompts"]
    assert {(c["family"], c["n_few_shot"]) for c in cases} == {(f, n) for f in evaluate.FAMILIES for n in (0, 1, 2)}
    for c in cases:
        assert evaluate.build_messages(c["n_few_shot"], c["family"]) == c["messages"], c
        assert evaluate.template_kwargs(c["family"]) == c["kwargs"], c


def test_prompt_structure():
    m = evaluate.build_messages(2)
    kinds = [p["type"] for p in m[0]["content"]]
    assert kinds == ["text", "image", "image", "text", "image", "text"] and m[0]["content"][-1]["text"] == "Are there any defects in the test image?"
    assert [p["type"] for p in evaluate.build_messages(0)[0]["content"]] == ["image", "text"]


@pytest.mark.parametrize("family", ["llava_1_5", "llava"])
def test_llava_eval_prompts_are_tokenized_as_vllm_does(family):
    """The reference's LLaVA evaluation scripts hand the prompt TEXT to vLLM, whose tokenizer call keeps the default add_special_tokens=True: a Llama / Vicuna /
    Mistral tokenizer then prepends <s> (the port tokenized with add_special_tokens=False and its prompts were one token short).  Fixture: the offline
    LLaVA processor with the post-processor a Llama tokenizer carries (`<s> $A`).  `encode_prompts` must equal the tokenizer's default call on the same text."""
    import fixture_util as fx
    from tokenizers import processors
    cfg = fx.TINY_LLAVA15 if family == "llava_1_5" else fx.TINY_LLAVA_NEXT
    proc = fx.local_llava_processor(cfg)
    tok = proc.tokenizer
    tok._tokenizer.post_processor = processors.TemplateProcessing(single="<s> $A", special_tokens=[("<s>", tok.bos_token_id)])
    tok.padding_side = "left"
    prompts = [proc.apply_chat_template(evaluate.build_messages(n, family), **evaluate.template_kwargs(family)) for n in (0, 1)]
    images = [fx.synth_pil_image(40, 32, 1), fx.synth_pil_image(36, 36, 2), fx.synth_pil_image(32, 40, 3)]
    enc = evaluate.encode_prompts(proc, prompts, images, family)
    ids, mask = enc["input_ids"].numpy(), enc["attention_mask"].numpy()
    for r in range(2):
        row = ids[r][mask[r] == 1]
        assert row[0] == tok.bos_token_id and (row[1:] != tok.bos_token_id).all(), "exactly one BOS, in front"
    # the qwen family keeps add_special_tokens=False (Qwen2 tokenizers define no BOS: same ids either way)
    want = proc(text=prompts, images=images, return_tensors="pt", padding=True, padding_side="left")["input_ids"].numpy()
    assert np.array_equal(ids, want)
