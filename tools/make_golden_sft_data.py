#!/usr/bin/env python3
"""Golden vectors for the PA-SFT batch construction: the reference's own
`infer_seqlen` (llamafactory/data/processors/processor_utils.py:51-65) and `_encode_supervised_example`
(llamafactory/data/processors/supervised.py:33-87) driven with a fake template / tokenizer (the template only has to hand back the per-turn
(source_ids, target_ids) pairs -- rendering lives in the HF processor).  Build container only; `peft` / `trl` are absent and stubbed (import-time
only).  Writes tests/golden/sft_data.json."""
import importlib.machinery, json, os, random, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def stub(name, **attrs):
    m = types.ModuleType(name)
    m.__spec__ = importlib.machinery.ModuleSpec(name, None)
    m.__path__ = []
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m


stub("peft", __version__="0.0")
stub("peft.utils", SAFETENSORS_WEIGHTS_NAME="a", WEIGHTS_NAME="b")
stub("peft.tuners")
stub("peft.tuners.lora", LoraLayer=object)
stub("trl", __version__="0.0")
sys.path.insert(0, "/root/reference/train/stage_sft")
from llamafactory.data.processors.processor_utils import infer_seqlen  # noqa: E402
from llamafactory.data.processors.supervised import _encode_supervised_example  # noqa: E402

rs = random.Random(5)
seq_cases = [[s, t, c, list(infer_seqlen(s, t, c))] for s in (0, 1, 7, 50, 300, 4000) for t in (0, 1, 9, 120, 2500) for c in (1, 8, 64, 511, 4096)]


class FakePlugin:
    def process_messages(self, messages, images, videos, processor):
        return messages

    def process_token_ids(self, a, b, images, videos, tokenizer, processor):
        return [], []


class FakeTemplate:
    def __init__(self, pairs, efficient_eos):
        self.mm_plugin, self.pairs, self.efficient_eos = FakePlugin(), pairs, efficient_eos

    def encode_multiturn(self, tokenizer, messages, system, tools):
        return [(list(s), list(t)) for s, t in self.pairs]


tok = types.SimpleNamespace(eos_token_id=2)
enc_cases = []
for n_turns in (1, 2, 3):
    for cutoff in (16, 40, 100, 4096):
        for train_on_prompt in (False, True):
            for mask_history in (False, True):
                for efficient_eos in (False, True):
                    pairs = [([rs.randrange(10, 1000) for _ in range(rs.choice((3, 9, 30)))], [rs.randrange(10, 1000) for _ in range(rs.choice((1, 6, 25)))]) for _ in range(n_turns)]
                    ids, labels = _encode_supervised_example(prompt=[{"role": "user", "content": "x"}] * (2 * n_turns - 1), response=[{"role": "assistant", "content": "y"}],
                                                             system=None, tools=None, images=[], videos=[], template=FakeTemplate(pairs, efficient_eos), tokenizer=tok,
                                                             processor=None, cutoff_len=cutoff, train_on_prompt=train_on_prompt, mask_history=mask_history)
                    enc_cases.append({"pairs": pairs, "cutoff_len": cutoff, "train_on_prompt": train_on_prompt, "mask_history": mask_history,
                                      "efficient_eos": efficient_eos, "eos_token_id": 2, "input_ids": ids, "labels": labels})
json.dump({"meta": {"generator": "tools/make_golden_sft_data.py", "reference": "Yanhui-Lee/IAD-R1 train/stage_sft/llamafactory"}, "infer_seqlen": seq_cases, "encode": enc_cases},
          open(os.path.join(ROOT, "tests", "golden", "sft_data.json"), "w"))
print("sft_data.json:", len(seq_cases), "infer_seqlen cases,", len(enc_cases), "encode cases")
