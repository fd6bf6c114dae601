#!/usr/bin/env python3
"""Golden vectors for the PA-SFT text path of the LLaVA-OneVision launch scripts (scripts/train/PA_SFT/PA_SFT_LLaVA_OneVision_SI_*.sh: --template llava_next_qwen):
the reference's own template (llamafactory/data/template.py:899-913: ChatML turns, default system prompt) and LlavaNextPlugin (mm_plugin.py:327-379: image
regularisation of the BASE plugin :108-123, image processor call, `<image>` -> processor._get_number_of_features(...) copies of the image token) driven with the
offline transformers LlavaOnevisionProcessor of tests/fixture_util.py and a one-id-per-character tokenizer.  Build container only (imports /root/reference).
Writes tests/golden/sft_llava.json: images are recorded as (width, height, seed) of fixture_util.synth_pil_image, expanded contents with runs of the image token
collapsed to counts."""
import importlib.machinery, json, os, re, sys, types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixture_util as fx  # noqa: E402


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
from llamafactory.data.template import TEMPLATES  # noqa: E402

tpl = TEMPLATES["llava_next_qwen"]


class CharTok:
    eos_token_id, bos_token_id = 2, None

    def encode(self, text, add_special_tokens=False):
        return [ord(c) for c in text]

    def convert_tokens_to_ids(self, t):
        return 7


proc = fx.local_llava_ov_processor()
# the reference hands tensor scalars to the processor (fine with its pinned transformers 4.51.3; 5.x calls round() on them): same values as python ints
_gnf = proc._get_number_of_features
proc._get_number_of_features = lambda oh, ow, h, w: _gnf(int(oh), int(ow), int(h), int(w))
u, a_ = (lambda t: {"role": "user", "content": t}), (lambda t: {"role": "assistant", "content": t})
collapse = lambda s: re.sub(r"(?:<image>)+", lambda m: "<image*%d>" % (len(m.group(0)) // 7), s)
cases = []
for res, imgs, system, msgs in (
        (512 * 512, [(100, 80, 1)], None, [u("<image>\nAre there any defects in the query image?"), a_("<think>clean</think><answer>No</answer>")]),
        (512 * 512, [(100, 120, 2), (400, 150, 3)], "You are an inspector.", [u("ref <image> query <image>"), a_("x"), u("again"), a_("y")]),
        (60 * 60, [(300, 200, 4)], None, [u("<image>small budget"), a_("ok")]),                    # the area cap of the base plugin resizes before the crops are cut
        (512 * 512, [], None, [u("no picture"), a_("fine")])):
    proc.image_resolution = res
    pil = [fx.synth_pil_image(w, h, seed) for w, h, seed in imgs]
    reg = tpl.mm_plugin._regularize_images(pil, image_resolution=res)
    expanded = tpl.mm_plugin.process_messages(msgs, pil, [], proc)
    mm = tpl.mm_plugin.get_mm_inputs(pil, [], [len(pil)], [0], [[0]], proc) if pil else {}
    pairs = tpl.encode_multiturn(CharTok(), [{**m, "content": collapse(m["content"])} for m in expanded], system, None)
    cases.append({"image_resolution": res, "images": [list(i) for i in imgs], "system": system, "messages": msgs,
                  "regularized_sizes": [[im.width, im.height] for im in reg],
                  "expanded": [{**m, "content": collapse(m["content"])} for m in expanded],
                  "image_sizes": [list(map(int, s)) for s in mm["image_sizes"].tolist()] if pil else [],
                  "pixel_shape": list(mm["pixel_values"].shape) if pil else [],
                  "pixel_abs_sum": float(mm["pixel_values"].double().abs().sum()) if pil else 0.0,
                  "pairs_collapsed": [[list(s), list(t)] for s, t in pairs]})
# ---- the LLaVA-1.5 / LLaVA-1.6 scripts: templates "llava" (vicuna format, LlavaPlugin: a fixed image_seqlen per image) and "llava_next_mistral"
# (Llama2Template: BOS prefix, system folded into the first [INST], LlavaNextPlugin) -----------------------------------------------------------------
class CharTokSpecial(CharTok):
    eos_token_id, bos_token_id = 2, 1


other = {}
for tname, cfg_d in (("llava", fx.TINY_LLAVA15), ("llava_next_mistral", fx.TINY_LLAVA_NEXT)):
    t2 = TEMPLATES[tname]
    proc2 = fx.local_llava_processor(cfg_d)
    side = cfg_d["vision"]["image_size"] // cfg_d["vision"]["patch_size"]
    proc2.image_seqlen = side * side                      # llamafactory model/patcher.py:81 <- model_utils/visual.py:177-191 (strategy "default")
    if hasattr(proc2, "_get_number_of_features"):
        _g2 = proc2._get_number_of_features
        proc2._get_number_of_features = lambda oh, ow, h, w, _g=_g2: _g(int(oh), int(ow), int(h), int(w))
    rows = []
    for res, imgs, system, msgs in (
            (512 * 512, [(100, 80, 1)], None, [u("<image>\nAre there any defects in the query image?"), a_("<think>clean</think><answer>No</answer>")]),
            (512 * 512, [(100, 120, 2), (150, 60, 3)], "You are an inspector.", [u("ref <image> query <image>"), a_("x"), u("again"), a_("y")]),
            (512 * 512, [], None, [u("no picture"), a_("fine")])):
        proc2.image_resolution = res
        pil = [fx.synth_pil_image(w, h, seed) for w, h, seed in imgs]
        expanded = t2.mm_plugin.process_messages(msgs, pil, [], proc2)
        mm = t2.mm_plugin.get_mm_inputs(pil, [], [len(pil)], [0], [[0]], proc2) if pil else {}
        pairs = t2.encode_multiturn(CharTokSpecial(), [{**m, "content": collapse(m["content"])} for m in expanded], system, None)
        rows.append({"image_resolution": res, "images": [list(i) for i in imgs], "system": system, "messages": msgs,
                     "expanded": [{**m, "content": collapse(m["content"])} for m in expanded],
                     "image_sizes": [list(map(int, s_)) for s_ in mm["image_sizes"].tolist()] if "image_sizes" in mm else [],
                     "pixel_shape": list(mm["pixel_values"].shape) if pil else [], "pixel_abs_sum": float(mm["pixel_values"].double().abs().sum()) if pil else 0.0,
                     "pairs_collapsed": [[list(s_), list(t_)] for s_, t_ in pairs]})
    other[tname] = {"default_system": t2.default_system, "cases": rows}
    print(tname, [r["expanded"][0]["content"][:50] for r in rows])

json.dump({"other_templates": other, "meta": {"generator": "tools/make_golden_sft_llava.py", "template": "llava_next_qwen", "default_system": tpl.default_system, "image_token": tpl.mm_plugin.image_token,
                    "processor": "tests/fixture_util.py::local_llava_ov_processor (transformers LlavaOnevisionProcessor, 56-pixel crops)"}, "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "sft_llava.json"), "w"))
print("sft_llava.json:", len(cases), "cases;", [c["expanded"][0]["content"][:60] for c in cases])
