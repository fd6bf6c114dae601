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
