#!/usr/bin/env python3
"""Golden vectors for the evaluation harness, produced by the reference's own host functions:
`get_ans` (scripts/Inference/IAD-R1-Inference/vLLM_Qwen_detect_format.py:140-165), `GPT4Query.parse_conversation`
(GPT4/gpt4v.py:123-169) and `caculate_accuracy_mmad` (helper/summary.py:8-124).  Build container only (imports /root/reference);
cv2 / seaborn / vllm are absent here and stubbed -- none of them is touched by the three functions.  Writes tests/golden/eval.json."""
import importlib.util, json, os, sys, tempfile, types

os.environ["MPLBACKEND"] = "Agg"
REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for name in ("cv2", "seaborn", "vllm", "vllm.multimodal", "vllm.multimodal.utils"):
    m = types.ModuleType(name)
    m.__path__ = []
    sys.modules[name] = m
sys.modules["seaborn"].heatmap = lambda *a, **k: None
sys.modules["vllm"].LLM = object
sys.modules["vllm"].SamplingParams = object
sys.modules["vllm.multimodal.utils"].fetch_image = None
sys.path.insert(0, REF)
spec = importlib.util.spec_from_file_location("ref_detect", os.path.join(REF, "scripts/Inference/IAD-R1-Inference/vLLM_Qwen_detect_format.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)
from helper.summary import caculate_accuracy_mmad  # noqa: E402
from GPT4.gpt4v import GPT4Query  # noqa: E402

OPT = {"A": "Yes.", "B": "No."}
OPT4 = {"A": "Scratch on the surface.", "B": "Missing part", "C": "Color stain.", "D": "No defect"}
responses = [
    "<think>x</think><answer>yes</answer>", "<answer>Yes</answer>", "<answer> no </answer>", "<answer>NO.</answer>", "<answer>maybe</answer>",
    "no tags", "", "<answer></answer>", "<answer>yes</answer><answer>no</answer>", "<think>a</think>\n<answer>No</answer>", "<answer>scratch</answer>",
    "<answer>missing part</answer>", "<answer>there is a color stain. here</answer>", "<answer>no defect</answer>", "<answer>a</answer>", "<answer>Yes, there is</answer>",
    "<answer>s</answer>", "<answer>.</answer>", None,
]
get_ans_cases = []
for r in responses:
    for o in (OPT, OPT4, None):
        get_ans_cases.append({"response": r, "options": o, "expected": mod.get_ans(r, o)})

q = GPT4Query.__new__(GPT4Query)
text_gts = [
    {"conversation": [{"Question": "Is there any defect in the object?", "Answer": "A", "Options": {"A": "Yes.", "B": "No."}, "type": "Anomaly Detection"},
                      {"Question": "What is the type?", "Answer": "C", "Options": {"A": "x", "B": "y", "C": "z", "D": "w"}, "type": "Defect Classification"}], "random_templates": ["a.png"]},
    {"meta": 1, "conversation_v2": [{"Question": "Q?", "Answer": "B", "Options": {"B": "No.", "A": "Yes."}, "type": "Anomaly Detection"}]},
    {"nothing": []},
]
parse_cases = []
for t in text_gts:
    qs, ans = q.parse_conversation(t)
    parse_cases.append({"text_gt": t, "questions": qs, "answers": ans})

import random
rs = random.Random(7)
answers = []
types_ = ["Anomaly Detection", "Defect Classification", "Object Structure", "Object Details", "Defect Localization"]
for ds in ("MVTec-AD", "VisA", "DS-MVTec"):
    for i in range(40):
        good = rs.random() < 0.4
        img = f"{ds}/obj{i % 3}/test/{'good' if good else 'broken'}/{i:03d}.png"
        for qt in types_[: (1 if good else 5)]:
            ca = rs.choice("AB" if qt == "Anomaly Detection" else "ABCD")
            ga = ca if rs.random() < 0.7 else rs.choice("ABCDE")
            if rs.random() < 0.04:
                ga = "some free text"          # triggers the remove-while-iterating path of the reference
            answers.append({"image": img, "question": {"text": "q"}, "question_type": qt, "correct_answer": ca, "gpt_answer": ga})
with tempfile.TemporaryDirectory() as d:
    pth = os.path.join(d, "answers.json")
    json.dump(answers, open(pth, "w"))
    stats = caculate_accuracy_mmad(pth)
    csv = open(pth.replace(".json", "_accuracy.csv")).read()
    stats2 = caculate_accuracy_mmad(pth, show_overkill_miss=True)
    csv2 = open(pth.replace(".json", "_accuracy.csv")).read()
# ---- prompt construction of the three evaluation scripts: their own `build_prompt` methods, with a recording stand-in for the tokenizer / processor ----
class _Recorder:
    def apply_chat_template(self, messages, **kw):
        self.last = {"messages": messages, "kwargs": kw}
        return "PROMPT"


prompt_cases = []
for fam, fname, cls in (("qwen", "vLLM_Qwen_detect_format.py", "QwenVLLMQuery"), ("llava", "vLLM_LLaVA_detect_format.py", "LLaVAVLLMQuery"),
                        ("llava_1_5", "vLLM_LLaVA_1_5_detect_format.py", "LLaVAVLLMQuery")):
    sp = importlib.util.spec_from_file_location("ref_" + fam, os.path.join(REF, "scripts/Inference/IAD-R1-Inference", fname))
    m = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(m)
    for n_shot in (0, 1, 2):
        qy = getattr(m, cls).__new__(getattr(m, cls))
        rec = _Recorder()
        qy.few_shot = [f"t{i}.png" for i in range(n_shot)]
        qy.tokenizer = qy.processor = rec
        out = qy.build_prompt([{"type": "text", "text": "Question 1: ...", "options": OPT}])
        assert out == "PROMPT"
        prompt_cases.append({"family": fam, "script": fname, "n_few_shot": n_shot, **rec.last})
    # get_ans of the LLaVA scripts is the same function restated: run the same cases through it
    for c in get_ans_cases:
        assert m.get_ans(c["response"], c["options"]) == c["expected"], (fname, c)

json.dump({"meta": {"generator": "tools/make_golden_eval.py", "reference": "Yanhui-Lee/IAD-R1 @ /root/reference"}, "get_ans": get_ans_cases, "parse_conversation": parse_cases,
           "prompts": prompt_cases,
           "accuracy": {"answers": answers, "csv": csv, "csv_overkill_miss": csv2, "question_stats": stats}}, open(os.path.join(ROOT, "tests", "golden", "eval.json"), "w"))
print("eval.json:", len(get_ans_cases), "get_ans cases,", len(parse_cases), "parse cases,", len(answers), "answers")
print(csv)
