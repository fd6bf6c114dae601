#!/usr/bin/env python3
"""Golden for the SC-GRPO prompt constants and `make_conversation` (/root/reference/train/stage_rl/grpo_ad.py:72-118 the four
templates, :135-181 the row -> chat-prompt function).  Both live INSIDE the reference's `main()` (the templates in an if / elif on `script_args.single_img`, the
function nested under the dataset branch), so nothing can be imported: the file is parsed with `ast`, the four string constants are read from the assignment
nodes of each branch, and the nested function definition is compiled on its own and executed with those constants in its globals -- the reference's code runs,
on a table of synthetic rows.  The fixture is data: the template strings, and for every (row, single_img, use_system_prompt) the returned dict or the name of the
exception the reference raises.  Run here (build container) only:  python tools/make_golden_prompts.py"""
import ast
import copy
import json
import os

SRC = "/root/reference/train/stage_rl/grpo_ad.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "prompts.json")

ROWS = [
    {"problem": "Is there any defect in the object?", "image": "MVTec/bottle/test/broken_large/000.png", "solution": "<answer>yes</answer>"},
    {"problem": "Are there any anomalies? {curly} braces stay", "image": ["ref/good/001.png", "test/crack/004.png"], "solution": "<answer>no</answer>"},
    {"problem": "Q3", "image": [{"path": "a/b.png", "bytes": None}], "solution": "s"},
    {"problem": "Q4", "image": [{"path": "a/b.png"}, "c/d.png"], "solution": "s", "messages": [{"role": "user", "content": "x"}]},
    {"problem": "Q5", "image": {"path": "single/dict.png"}, "solution": "s"},
    {"problem": "Q6", "image": [3], "solution": "s"},
    {"problem": "Q7", "image": "", "solution": "s"},
    {"problem": "Q8", "solution": "s"},
    {"problem": "Q9", "image": 7, "solution": "s"},
]


def main():
    tree = ast.parse(open(SRC).read())
    fn_main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    # ---- the templates: `if script_args.single_img == 1: ... elif script_args.single_img == 0: ...`
    templates = {}
    for node in ast.walk(fn_main):
        if isinstance(node, ast.If) and isinstance(node.test, ast.Compare) and isinstance(node.test.left, ast.Attribute) and node.test.left.attr == "single_img":
            key = ast.literal_eval(node.test.comparators[0])
            got = {}
            for st in node.body:
                if isinstance(st, ast.Assign) and isinstance(st.targets[0], ast.Name) and isinstance(st.value, ast.Constant):
                    got[st.targets[0].id] = st.value.value
            if "GENERAL_SYSTEM_PROMPT" in got and "GENERAL_QUESTION_PROMPT" in got:
                templates[int(key)] = {"system": got["GENERAL_SYSTEM_PROMPT"], "question": got["GENERAL_QUESTION_PROMPT"]}
    assert sorted(templates) == [0, 1], sorted(templates)
    # ---- make_conversation: the nested def, compiled alone
    fn = next(n for n in ast.walk(fn_main) if isinstance(n, ast.FunctionDef) and n.name == "make_conversation")
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    code = compile(mod, SRC, "exec")
    cases = []
    for single_img in (1, 0):
        ns = {"os": os, "QUESTION_PROMPT": templates[single_img]["question"], "SYSTEM_PROMPT": templates[single_img]["system"]}
        exec(code, ns)
        for ri, row in enumerate(ROWS):
            for use_system in (False, True):
                try:
                    out = ns["make_conversation"](copy.deepcopy(row), image_path="/data/Expert-AD", use_system_prompt=use_system)
                    res = {"returns": out}
                except Exception as e:      # noqa: BLE001 -- the reference's own failure modes are part of the record
                    res = {"raises": type(e).__name__}
                cases.append({"row": ri, "single_img": single_img, "use_system_prompt": use_system, **res})
    json.dump({"meta": {"source": "train/stage_rl/grpo_ad.py:72-118,135-181 of Yanhui-Lee/IAD-R1 (constants read from the AST, the nested make_conversation compiled and executed)",
                        "generator": "tools/make_golden_prompts.py"}, "templates": {str(k): v for k, v in templates.items()}, "rows": ROWS, "cases": cases},
              open(OUT, "w"), indent=1)
    print(f"wrote {OUT}: {len(cases)} cases; returns None for {sum(1 for c in cases if c.get('returns', 0) is None)}; raises: {sorted({c['raises'] for c in cases if 'raises' in c})}")


if __name__ == "__main__":
    main()
