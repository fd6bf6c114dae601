"""SC-GRPO reward functions (host side, CPU, exact): the plugin API the reference trainer calls as
``f(prompts=, completions=, current_step=, **dataset_columns) -> list[float]``
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:773-781) with the registry keys of
/root/reference/train/stage_rl/grpo_ad.py:126-129.

Semantics follow /root/reference/train/stage_rl/reward.py:13-101 (format / accuracy),
reward_process/type_reward.py:5-232 (lexicon cascade) and reward_process/location_reward.py:1-49
(3x3 cell).  Parity is checked value-for-value against tests/golden/rewards.json, captured from
the reference.  Differences by design: the matcher is built once (it is stateless) instead of per
sample, and the per-sample debug printing is opt-in (IADR1_REWARD_VERBOSE=1).
"""
from __future__ import annotations

import os
import re
from difflib import SequenceMatcher
from functools import lru_cache

_VERBOSE = os.environ.get("IADR1_REWARD_VERBOSE", "0") == "1"

# group -> category -> synonyms.  Order matters: ties in the containment / fuzzy scans go to the
# earliest entry (category name first, then its synonyms, categories in this order).
LEXICON = (
    ("Surface Anomalies", (
        ("Contamination", ("surface contamination", "stain", "dirt", "impurity", "color anomaly")),
        ("Presence of foreign objects", ("foreign object", "foreign body", "debris", "contaminant object", "extraneous material", "foreign element", "foreign matter", "unwanted object")),
        ("Scratch", ("surface scratch", "scratch mark", "linear scratch", "score mark", "linear anomaly")),
        ("Missing parts", ("missing part", "surface notch", "notch", "gap", "chip", "surface discontinuity")),
    )),
    ("Structural Anomalies", (
        ("Deformation", ("shape distortion", "warping", "bending", "twisting", "shape deviation", "geometric distortion", "irregularity", "bent component")),
        ("Hole", ("opening", "perforation", "puncture", "cavity", "void", "aperture", "penetration defect", "through-hole")),
        ("Damage", ("structural damage", "breakage", "fracture", "rupture", "deterioration", "material damage", "surface damage")),
        ("Abrasion", ("wear", "grinding damage", "surface erosion", "wear mark", "surface wear")),
    )),
)
GROUP_ALIASES = {"Surface Anomalies": ("surface anomalies", "surface anomaly"), "Structural Anomalies": ("structural anomalies", "structural anomaly")}

S_EXACT, S_SEMANTIC, S_CATEGORY, S_FUZZY, S_GROUP, S_NONE = 1.0, 0.85, 0.6, 0.4, 0.3, 0.0
FUZZY_MIN = 0.7

_WS = re.compile(r"\s+")
_PUNCT = re.compile(r"[^\w\s-]")
_TAG = {t: re.compile(rf"<{t}>(.*?)</{t}>") for t in ("answer", "type", "location")}
# tag-order templates, matched with re.fullmatch + DOTALL (reward.py:14-15)
_FMT_NORMAL = re.compile(r"^(?!.*<location>)(?!.*<type>).*<think>.*?</think><answer>.*?</answer>.*$", re.DOTALL)
_FMT_ANOMALY = re.compile(r".*<think>.*?</think><location>.*?</location><type>.*?</type><answer>.*?</answer>.*", re.DOTALL)


def normalize(text: str) -> str:
    if not text:
        return ""
    return _PUNCT.sub("", _WS.sub(" ", text.lower().strip()))


class TypeMatcher:
    """Lexicon lookups for the anomaly-type score."""

    def __init__(self):
        self.term_category: dict[str, str] = {}
        self.category_group: dict[str, str] = {}
        self.group_terms: dict[str, str] = {}
        for group, cats in LEXICON:
            for cat, syns in cats:
                self.category_group[cat] = group
        # insertion order = categories in the reference's vocabulary order (surface x4, structural x4)
        for group, cats in LEXICON:
            for cat, syns in cats:
                self.term_category[normalize(cat)] = cat
                for s in syns:
                    self.term_category[normalize(s)] = cat
        for group, aliases in GROUP_ALIASES.items():
            self.group_terms[normalize(group)] = group
            for a in aliases:
                self.group_terms[normalize(a)] = group

    def best_category(self, text: str):
        """(category, confidence) of a free-text defect type.  A pure function of the string; the fuzzy pass is a difflib ratio against every vocabulary term
        (0.3 ms per call), and training batches repeat the same few type strings, so results are memoised per matcher."""
        memo = self.__dict__.setdefault("_best_memo", {})
        hit = memo.get(text)
        if hit is None:
            if len(memo) > 65536:
                memo.clear()
            hit = memo[text] = self._best_category(text)
        return hit

    def _best_category(self, text: str):
        n = normalize(text)
        hit = self.term_category.get(n)
        if hit is not None:
            return hit, 1.0
        best, conf = None, 0.0
        for term, cat in self.term_category.items():
            if n in term or term in n:
                a, b = len(n), len(term)
                c = min(a, b) / max(a, b)
                if c > conf:
                    best, conf = cat, c
        if best:
            return best, conf
        for term, cat in self.term_category.items():
            r = SequenceMatcher(None, n, term).ratio()
            if r >= FUZZY_MIN and r > conf:
                best, conf = cat, r
        return best, conf

    def group_of_text(self, text: str):
        return self.group_terms.get(normalize(text)) if text else None

    def score(self, predicted: str, actual: str) -> float:
        memo = self.__dict__.setdefault("_score_memo", {})
        key = (predicted, actual)
        hit = memo.get(key)
        if hit is None:
            if len(memo) > 65536:
                memo.clear()
            hit = memo[key] = self._score(predicted, actual)
        return hit

    def _score(self, predicted: str, actual: str) -> float:
        if not predicted or not actual:
            return S_NONE
        p, a = normalize(predicted), normalize(actual)
        pg_text, ag_text = self.group_of_text(predicted), self.group_of_text(actual)
        pc, pconf = self.best_category(predicted)
        ac, aconf = self.best_category(actual)
        pg = pg_text or self.category_group.get(pc)
        ag = ag_text or self.category_group.get(ac)
        if pg and ag and pg != ag:
            return S_NONE
        if pg_text and not ag_text and ag == pg_text:
            return S_GROUP
        if ag_text and not pg_text and pg == ag_text:
            return S_GROUP
        if p == a:
            return S_EXACT
        if p in a or a in p:
            return S_SEMANTIC
        if not pc or not ac:
            r = SequenceMatcher(None, p, a).ratio()
            return r * S_FUZZY if r >= FUZZY_MIN else S_NONE
        if pc == ac:
            return S_CATEGORY + (S_SEMANTIC - S_CATEGORY) * min(pconf, aconf)
        g1, g2 = self.category_group.get(pc), self.category_group.get(ac)
        if g1 and g2 and g1 == g2:
            return S_GROUP
        r = SequenceMatcher(None, p, a).ratio()
        return r * S_FUZZY if r >= FUZZY_MIN else S_NONE


@lru_cache(maxsize=1)
def _matcher() -> TypeMatcher:
    return TypeMatcher()


def type_score(predicted: str, actual: str) -> float:
    return _matcher().score(predicted, actual)


def _cell(text: str) -> int:
    t = text.lower().strip()
    c = 5
    if "left" in t:
        c -= 1
    elif "right" in t:
        c += 1
    if "top" in t or "upper" in t:
        c -= 3
    elif "bottom" in t or "lower" in t:
        c += 3
    return max(1, min(9, c))


def location_score(predicted: str, actual: str) -> int:
    return 1 if _cell(predicted) == _cell(actual) else 0


def _gt_answer(solution: str) -> str:
    m = _TAG["answer"].search(solution)
    return (m.group(1).strip() if m else solution.strip()).lower()


def consistency_reward(completions, solution, **kwargs):
    """'format' reward: 1.0 iff the completion matches the tag order implied by the ground truth.
    As in the reference, a ground truth that is neither yes nor no contributes NO entry."""
    out = []
    for comp, sol in zip(completions, solution):
        text = comp[0]["content"]
        gt = _gt_answer(sol)
        if gt == "yes":
            out.append(1.0 if _FMT_ANOMALY.fullmatch(text) else 0.0)
        elif gt == "no":
            out.append(1.0 if _FMT_NORMAL.fullmatch(text) else 0.0)
    return out


def _accuracy_one(text: str, sol: str) -> float:
    gt = _gt_answer(sol)
    if gt == "no":
        m = _TAG["answer"].search(text)
        return 1.0 if (m and m.group(1).strip().lower() == "no") else 0.0
    if gt != "yes":
        return 0.0
    total = 0.0
    pt, gtt = _TAG["type"].search(text), _TAG["type"].search(sol)
    if pt and gtt:
        total += type_score(pt.group(1).strip().lower(), gtt.group(1).strip().lower())
    pl, gl = _TAG["location"].search(text), _TAG["location"].search(sol)
    if pl and gl:
        total += location_score(pl.group(1).strip().lower(), gl.group(1).strip().lower())
    r = total / 2.0
    m = _TAG["answer"].search(text)
    if m and m.group(1).strip().lower() == "yes":
        r += 1.0
    return r


def accuracy_reward(completions, solution, **kwargs):
    """'accuracy' reward in [0, 2]: answer + (type + location)/2; any internal error -> 0."""
    out = []
    for comp, sol in zip(completions, solution):
        try:
            r = _accuracy_one(comp[0]["content"], sol)
        except Exception:
            r = 0.0
        if _VERBOSE:
            print(f"[accuracy_reward] gt={_gt_answer(sol)!r} reward={r} completion={comp[0]['content']!r}")
        out.append(r)
    return out


REWARD_FUNCS = {"accuracy": accuracy_reward, "format": consistency_reward}


# ------------------------------------------------------------------------------------------------------------------------------------
# The reference's ablation variants (REF train/stage_rl/reward.py:107-347).  The entry point registers none of them (grpo_ad.py:126-129), they
# are here so that the whole reward module has a counterpart under the same names; pinned value for value by tests/golden/reward_variants.json.
# ------------------------------------------------------------------------------------------------------------------------------------
_ANSWER_I = re.compile(r"<answer>(.*?)</answer>", re.IGNORECASE)
_COT_TAGS = tuple(re.compile(rf"<{t}>.*?</{t}>", re.IGNORECASE | re.DOTALL) for t in ("type", "location", "description"))
_COT_SCORE = {3: 1.0, 2: 0.7, 1: 0.4, 0: 0.0}
_FMT_BASE = re.compile(r".*<think>.*?</think><answer>.*?</answer>.*", re.DOTALL)


def _texts(completions):
    return [c[0]["content"] for c in completions]


def consistency_reward_cot(completions, solution, **kwargs):
    """REF reward.py:107-159: the answer (case-insensitive tag and value) must equal the ground truth; then "no" scores 1 only WITHOUT any of the <type> / <location> /
    <description> tags, "yes" scores 1.0 / 0.7 / 0.4 / 0 for three / two / one / none of them."""
    out = []
    for text, sol in zip(_texts(completions), solution):
        m = _ANSWER_I.search(sol)
        gt = (m.group(1) if m else sol).strip().lower()
        a = _ANSWER_I.search(text)
        ans = a.group(1).strip().lower() if a else None
        if ans is None or ans != gt:
            out.append(0.0)
            continue
        n = sum(1 for rx in _COT_TAGS if rx.search(text))
        out.append((1.0 if n == 0 else 0.0) if ans == "no" else (_COT_SCORE[n] if ans == "yes" else 0.0))
    return out


format_consistency_reward_cot = consistency_reward_cot        # REF reward.py:161-212: the same rule under a second name


def _accuracy_with_one_part(text: str, sol: str, part: str) -> float:
    """REF reward.py:215-301: the accuracy reward with ONE localisation term (weight 1) instead of the mean of two.  A missing tag on either side raises before the
    answer bonus is added (the reference catches it and keeps the value it had: 0)."""
    gt = _gt_answer(sol)
    if gt == "no":
        m = _TAG["answer"].search(text)
        return 1.0 if (m and m.group(1).strip().lower() == "no") else 0.0
    if gt != "yes":
        return 0.0
    pred, want = _TAG[part].search(text), _TAG[part].search(sol)
    if pred is None or want is None:
        return 0.0
    a, b = pred.group(1).strip().lower(), want.group(1).strip().lower()
    r = float(location_score(a, b)) if part == "location" else float(type_score(a, b))
    m = _TAG["answer"].search(text)
    if m and m.group(1).strip().lower() == "yes":
        r += 1.0
    return r


def accuracy_reward_cot_wo_type(completions, solution, **kwargs):
    return [_accuracy_with_one_part(t, s, "location") for t, s in zip(_texts(completions), solution)]


def accuracy_reward_cot_wo_location(completions, solution, **kwargs):
    return [_accuracy_with_one_part(t, s, "type") for t, s in zip(_texts(completions), solution)]


def format_reward_cot_base(completions, solution, **kwargs):
    """REF reward.py:303-312: <think>..</think><answer>..</answer> somewhere in the text, whatever the ground truth."""
    return [1.0 if _FMT_BASE.fullmatch(t) else 0.0 for t in _texts(completions)]


def accuracy_reward_cot_base(completions, solution, **kwargs):
    """REF reward.py:314-343: 1 when the (case-sensitive tag, case-insensitive value) answer equals a yes / no ground truth."""
    out = []
    for text, sol in zip(_texts(completions), solution):
        gt = _gt_answer(sol)
        m = _TAG["answer"].search(text)
        out.append(1.0 if (gt in ("yes", "no") and m and m.group(1).strip().lower() == gt) else 0.0)
    return out


def wo_format(completions, solution, **kwargs):
    """REF reward.py:345-347 returns the INT 0, not a list (the trainer's float conversion of a per-sample list would fail on it); reproduced."""
    return 0
