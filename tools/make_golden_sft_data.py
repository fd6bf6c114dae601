#!/usr/bin/env python3
"""Golden vectors for the PA-SFT batch construction: the reference's own
`infer_seqlen` (llamafactory/data/processors/processor_utils.py:51-65) and `_encode_supervised_example`
(llamafactory/data/processors/supervised.py:33-87) driven with a fake template / tokenizer (the template only has to hand back the per-turn
(source_ids, target_ids) pairs -- rendering lives in the HF processor).  Build container only; `peft` / `trl` are absent and stubbed (import-time
only).  Writes tests/golden/sft_data.json.
Second file, tests/golden/sft_text.json: the path in front of the tokenizer -- convert_sharegpt (aligner.py:137-232), the "qwen2_vl" template's encode_multiturn with a
one-id-per-character tokenizer (template.py:85-160,1120-1133), Qwen2vlPlugin image regularisation and placeholder expansion (mm_plugin.py:108-123,
810-896) with a fake image processor (the grid it reports is part of the recorded input)."""
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


# ---------------------------------------------------------------------------------------------------------------- text path (sft_text.json)
import torch  # noqa: E402
from PIL import Image  # noqa: E402
from llamafactory.data.aligner import convert_sharegpt  # noqa: E402
from llamafactory.data.parser import DatasetAttr  # noqa: E402
from llamafactory.data.template import TEMPLATES  # noqa: E402

tpl = TEMPLATES["qwen2_vl"]


class CharTok:
    eos_token_id, bos_token_id = 2, None

    def encode(self, text, add_special_tokens=False):
        return [ord(c) for c in text]

    def convert_tokens_to_ids(self, t):
        return 7


def attr_from(entry):
    a = DatasetAttr("file", dataset_name=entry["file_name"])
    a.set_attr("formatting", entry, default="alpaca")
    if "columns" in entry:       # as get_dataset_list does (parser.py:134-142): a present `columns` / `tags` object resets every name it does not mention to None
        for c in ("system", "tools", "images", "videos", "messages"):
            a.set_attr(c, entry["columns"])
    if "tags" in entry:
        for t in ("role_tag", "content_tag", "user_tag", "assistant_tag", "observation_tag", "function_tag", "system_tag"):
            a.set_attr(t, entry["tags"])
    return a


readme_entry = {"file_name": "expert_ad.json", "formatting": "sharegpt", "columns": {"messages": "messages", "images": "images"},
                "tags": {"role_tag": "role", "content_tag": "content", "user_tag": "user", "assistant_tag": "assistant"}}
readme_sys_entry = {**readme_entry, "tags": {**readme_entry["tags"], "system_tag": "system"}}
default_entry = {"file_name": "d.json", "formatting": "sharegpt", "columns": {"messages": "conversations", "images": "images", "system": "sys"}}
bare_entry = {"file_name": "b.json", "formatting": "sharegpt"}
u, a_ = (lambda t: {"role": "user", "content": t}), (lambda t: {"role": "assistant", "content": t})
h, g = (lambda t: {"from": "human", "value": t}), (lambda t: {"from": "gpt", "value": t})
rows = [
    (readme_entry, {"images": "mvtec/bottle/000.png", "messages": [u("<image>\nAre there any defects in the query image?"), a_("<think>smooth rim</think><answer>No</answer>")]}),
    (readme_entry, {"images": ["a.png", "b.png"], "messages": [u("<image><image>\nCompare."), a_("<answer>Yes</answer>"), u("Where?"), a_("<location>top left</location>")]}),
    (readme_entry, {"images": [], "messages": [u("hello"), a_("hi")]}),
    (readme_sys_entry, {"images": "x.png", "messages": [{"role": "system", "content": "You are an inspector."}, u("<image>ok?"), a_("yes")]}),
    (readme_entry, {"images": "x.png", "messages": [a_("backwards"), u("order")]}),
    (readme_entry, {"images": "x.png", "messages": [u("one"), a_("two"), u("three")]}),
    (readme_entry, {"images": "x.png", "messages": []}),
    (default_entry, {"images": ["p.png"], "sys": "Be brief.", "conversations": [h("<image>what"), g("a screw"), {"from": "observation", "value": "{\"ok\": 1}"}, g("fine")]}),
    (default_entry, {"images": ["p.png"], "sys": "", "conversations": [{"from": "system", "value": "S!"}, h("q"), g("r")]}),
    (default_entry, {"images": ["p.png"], "sys": "col", "conversations": [h("q"), h("q2")]}),
    (bare_entry, {"conversations": [h("plain"), g("text"), h("more"), g("words")]}),
]
text_cases = []
data_args = types.SimpleNamespace(image_dir="/nonexistent")
for entry, ex in rows:
    out = convert_sharegpt(ex, attr_from(entry), data_args)
    case = {"entry": entry, "example": ex, "aligned": {"prompt": out["_prompt"], "response": out["_response"], "system": out["_system"], "images": out["_images"]}}
    if out["_prompt"]:
        pairs = tpl.encode_multiturn(CharTok(), out["_prompt"] + out["_response"], out["_system"], out["_tools"])
        case["pairs"] = [[list(s), list(t)] for s, t in pairs]
    text_cases.append(case)


class FakeImageProcessor:
    merge_size = 2

    def __call__(self, images=None, videos=None, return_tensors="pt"):
        return {"image_grid_thw": torch.tensor([[1, max(2, im.height // 28 * 2), max(2, im.width // 28 * 2)] for im in images])}


size_cases, expand_cases = [], []
for res in (512 * 512, 448 * 448, 480000, 10 ** 9):
    for wh in ((448, 448), (3000, 2000), (10, 300), (6000, 20), (20, 6000), (27, 27), (1024, 100), (517, 613), (5, 1200), (28, 5601)):
        im = tpl.mm_plugin._regularize_images([Image.new("L", wh)], image_resolution=res)[0]
        size_cases.append([wh[0], wh[1], res, im.width, im.height, im.mode])
import hashlib  # noqa: E402
import numpy as np  # noqa: E402
pixel_cases = []
for i, (mode, wh, res) in enumerate((("RGB", (300, 200), 100 * 100), ("L", (10, 300), 512 * 512), ("RGB", (6000, 20), 10 ** 9), ("P", (640, 480), 448 * 448), ("RGBA", (33, 47), 512 * 512))):
    arr = np.random.RandomState(i).randint(0, 256, size=(wh[1], wh[0]) + ((len(mode),) if len(mode) > 1 else ()), dtype=np.uint8)
    im = tpl.mm_plugin._regularize_images([Image.fromarray(arr if mode != "P" else arr, mode=mode)], image_resolution=res)[0]
    pixel_cases.append({"mode": mode, "size": list(wh), "seed": i, "max_pixels": res, "out_size": [im.width, im.height], "sha1": hashlib.sha1(im.tobytes()).hexdigest()})
for sizes, msgs in (([(56, 84)], [u("<image>\nAre there any defects?"), a_("no")]),
                    ([(56, 56), (112, 56)], [u("ref <image> query <image>"), a_("x"), u("again"), a_("y")]),
                    ([(56, 56), (84, 84)], [u("first <image>"), a_("<image> echoed")])):
    proc = types.SimpleNamespace(image_processor=FakeImageProcessor(), image_resolution=512 * 512)
    imgs = [Image.new("RGB", wh) for wh in sizes]
    grids = FakeImageProcessor()(images=imgs)["image_grid_thw"].tolist()
    expand_cases.append({"messages": msgs, "grids": grids, "expanded": tpl.mm_plugin.process_messages(msgs, imgs, [], proc)})
errors = []
for sizes, msgs in (([(56, 56)], [u("no placeholder"), a_("x")]), ([(56, 56)], [u("<image><image>"), a_("x")])):
    proc = types.SimpleNamespace(image_processor=FakeImageProcessor(), image_resolution=512 * 512)
    imgs = [Image.new("RGB", wh) for wh in sizes]
    try:
        tpl.mm_plugin.process_messages(msgs, imgs, [], proc)
        raise SystemExit("expected an error")
    except ValueError as e:
        errors.append({"messages": msgs, "grids": FakeImageProcessor()(images=imgs)["image_grid_thw"].tolist(), "error": str(e)})
json.dump({"meta": {"generator": "tools/make_golden_sft_data.py", "template": "qwen2_vl", "default_system": tpl.default_system}, "rows": text_cases, "image_sizes": size_cases,
           "pixels": pixel_cases, "expand": expand_cases, "expand_errors": errors}, open(os.path.join(ROOT, "tests", "golden", "sft_text.json"), "w"))
print("sft_text.json:", len(text_cases), "rows,", len(size_cases), "image sizes,", len(expand_cases), "+", len(errors), "expansions")
