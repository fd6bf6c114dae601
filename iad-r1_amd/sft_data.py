"""PA-SFT batch construction: how one multi-turn conversation becomes (input_ids, labels) under a token budget.

Host-side integer logic, no device work.  Follows the reference's LLaMA-Factory fork:
  * `turn_budget`       <- infer_seqlen, train/stage_sft/llamafactory/data/processors/processor_utils.py:51-65
  * `supervised_labels` <- _encode_supervised_example, train/stage_sft/llamafactory/data/processors/supervised.py:33-87
Pinned bit-exactly against those two functions by tests/golden/sft_data.json (tools/make_golden_sft_data.py)."""
import math
import os
from dataclasses import dataclass, fields
from typing import Any, Dict, List, Optional, Sequence, Tuple

IGNORE_INDEX = -100


def turn_budget(source_len: int, target_len: int, budget: int) -> Tuple[int, int]:
    """How many prompt / answer tokens of one turn survive when only `budget` tokens are left.

    A short answer (less than half the budget) is kept whole and the prompt gives way; otherwise a short prompt is kept whole and the answer gives way;
    when both are long they share the budget in proportion to their lengths (answer share rounded down, prompt gets the remainder)."""
    if 2 * target_len < budget:
        target_cap = budget
    elif 2 * source_len < budget:
        target_cap = budget - source_len
    else:
        target_cap = int(budget * (target_len / (source_len + target_len)))
    keep_target = target_len if target_len < target_cap else target_cap
    room = budget - keep_target
    keep_source = min(source_len, room if room > 0 else 0)
    return keep_source, keep_target


def supervised_labels(turns: Sequence[Tuple[Sequence[int], Sequence[int]]], cutoff_len: int, eos_token_id: int = None, train_on_prompt: bool = False,
                      mask_history: bool = False, efficient_eos: bool = False) -> Tuple[List[int], List[int]]:
    """turns = [(prompt_ids, answer_ids)] per conversation turn  ->  (input_ids, labels), both at most cutoff_len long.

    Turns are admitted in order (newest first under `mask_history`, so old turns are the ones dropped) until the budget is used up, each trimmed by
    `turn_budget` against what is left.  Prompt tokens are labelled IGNORE_INDEX unless `train_on_prompt`; with `mask_history` only the last turn's answer
    is supervised.  `efficient_eos` templates omit the per-turn eos from the rendered text: the first prompt position of every turn is labelled eos and
    one eos is appended at the end (a slot reserved for it up front)."""
    used = 1 if efficient_eos else 0
    pieces = []     # (ids, labels) per admitted turn, admission order
    order = list(turns)[::-1] if mask_history else list(turns)
    for n, (src, tgt) in enumerate(order):
        if used >= cutoff_len:
            break
        ks, kt = turn_budget(len(src), len(tgt), cutoff_len - used)
        src, tgt = list(src[:ks]), list(tgt[:kt])
        used += ks + kt
        if train_on_prompt:
            src_lab = list(src)
        else:
            src_lab = [IGNORE_INDEX] * ks
            if efficient_eos and ks:
                src_lab[0] = eos_token_id
        tgt_lab = [IGNORE_INDEX] * kt if (mask_history and n) else list(tgt)
        pieces.append((src + tgt, src_lab + tgt_lab))
    if mask_history:
        pieces.reverse()
    ids = [t for p in pieces for t in p[0]]
    labels = [t for p in pieces for t in p[1]]
    if efficient_eos:
        ids.append(eos_token_id)
        labels.append(eos_token_id)
    return ids, labels


# ---------------------------------------------------------------------------------------------------------------------------------------------
# Dataset rows -> conversation turns -> text (the part of a22 in front of the tokenizer).  Follows
#   * `ShareGPTSchema` / `align_sharegpt`    <- DatasetAttr + get_dataset_list + convert_sharegpt + _convert_images,
#                                               llamafactory/data/parser.py:27-162, aligner.py:33-54,137-232
#   * `regular_image_size`                   <- BasePlugin._preprocess_image + Qwen2vlPlugin._preprocess_image, mm_plugin.py:108-123,810-824
#   * `expand_image_placeholders`            <- Qwen2vlPlugin.process_messages, mm_plugin.py:850-896
#   * `qwen2_vl_turn_texts` / `encode_turns` <- template "qwen2_vl" (template.py:1120-1133) through Template._encode / encode_multiturn /
#                                               _convert_elements_to_ids (template.py:85-160)
# Pinned by tests/golden/sft_text.json (tools/make_golden_sft_data.py drives the reference's own functions).
# ---------------------------------------------------------------------------------------------------------------------------------------------
IMAGE_PLACEHOLDER = "<image>"
QWEN2_VL_DEFAULT_SYSTEM = "You are a helpful assistant."


@dataclass
class ShareGPTSchema:
    """Column and tag names of a sharegpt-format dataset: the `columns` / `tags` objects of a dataset_info.json entry, with LLaMA-Factory's defaults."""
    messages: str = "conversations"
    images: Optional[str] = None
    system: Optional[str] = None
    role_tag: str = "from"
    content_tag: str = "value"
    user_tag: str = "human"
    assistant_tag: str = "gpt"
    observation_tag: str = "observation"
    function_tag: str = "function_call"
    system_tag: Optional[str] = "system"

    @classmethod
    def from_dataset_info(cls, entry: Dict[str, Any]) -> "ShareGPTSchema":
        if entry.get("formatting", "alpaca") != "sharegpt":
            raise ValueError("only sharegpt-format datasets are part of the PA-SFT path (dataset_info.json: \"formatting\": \"sharegpt\")")
        s = cls()
        for f in fields(cls):       # a present `columns` / `tags` object resets every name it does not mention to None (parser.py:134-155)
            key = "tags" if f.name.endswith("_tag") else "columns"
            if key in entry:
                setattr(s, f.name, entry[key].get(f.name))
        return s


def align_sharegpt(example: Dict[str, Any], schema: ShareGPTSchema, image_dir: Optional[str] = None) -> Dict[str, Any]:
    """One dataset row -> {"prompt": [...], "response": [...], "system": str, "images": list | None} with roles user / assistant / observation /
    function.  A leading system-tagged message becomes `system`.  Rows whose roles do not alternate (user|observation, assistant|function, ...) or
    whose turn count is odd are dropped the way the reference drops them: empty prompt and response."""
    role_of = {schema.user_tag: "user", schema.assistant_tag: "assistant", schema.observation_tag: "observation", schema.function_tag: "function",
               schema.system_tag: "system"}
    allowed = ((schema.user_tag, schema.observation_tag), (schema.assistant_tag, schema.function_tag))
    msgs = example[schema.messages]
    if schema.system_tag and len(msgs) and msgs[0][schema.role_tag] == schema.system_tag:
        system, msgs = msgs[0][schema.content_tag], msgs[1:]
    else:
        system = example[schema.system] if schema.system else ""
    broken = len(msgs) % 2 != 0
    turns = []
    for i, m in enumerate(msgs):
        if m[schema.role_tag] not in allowed[i % 2]:
            broken = True
        turns.append({"role": role_of[m[schema.role_tag]], "content": m[schema.content_tag]})
    images = None
    if schema.images:
        images = example[schema.images]
        if not isinstance(images, list):
            images = [images]
        elif not images:
            images = None
        else:
            images = list(images)
        if images is not None and image_dir is not None:
            images = [os.path.join(image_dir, im) if isinstance(im, str) and os.path.isfile(os.path.join(image_dir, im)) else im for im in images]
    return {"prompt": [] if broken else turns[:-1], "response": [] if broken else turns[-1:], "system": system, "images": images}


def _regular_size_steps(width: int, height: int, max_pixels: int):
    """The successive (width, height) targets of the reference's resizes, in order; empty when the image is left alone."""
    if width * height > max_pixels:
        f = math.sqrt(max_pixels / (width * height))
        width, height = int(width * f), int(height * f)
        yield width, height
    if min(width, height) < 28:
        width, height = max(width, 28), max(height, 28)
        yield width, height
    if width / height > 200:
        width = height * 180
        yield width, height
    if height / width > 200:
        height = width * 180
        yield width, height


def regular_image_size(width: int, height: int, max_pixels: int = 512 * 512):
    """(width, height) an image has when the HF image processor sees it: area capped at `max_pixels` (both sides scaled by the same factor,
    truncated), sides at least 28, aspect ratio cut back to 180 when it exceeds 200."""
    for width, height in _regular_size_steps(width, height, max_pixels):
        pass
    return width, height


def regularize_image(image, max_pixels: int = 512 * 512):
    """PIL image -> RGB PIL image of `regular_image_size`, through the same sequence of nearest-neighbour resizes as the reference."""
    from PIL import Image
    for size in _regular_size_steps(image.width, image.height, max_pixels):
        image = image.resize(size, resample=Image.Resampling.NEAREST)
    return image if image.mode == "RGB" else image.convert("RGB")


def regularize_image_base(image, max_pixels: int = 512 * 512):
    """The BASE plugin's image pre-processing (llamafactory mm_plugin.py:108-123), which the llava* templates use: area capped at `max_pixels` (both sides scaled
    by the same factor, truncated; nearest-neighbour), RGB -- without the minimum-side / aspect-ratio steps the Qwen2-VL plugin adds (`regularize_image`)."""
    from PIL import Image
    if image.width * image.height > max_pixels:
        f = math.sqrt(max_pixels / (image.width * image.height))
        image = image.resize((int(image.width * f), int(image.height * f)), resample=Image.Resampling.NEAREST)
    return image if image.mode == "RGB" else image.convert("RGB")


def expand_image_placeholders_llava(messages, image_sizes, tokens_of, image_token: str = "<image>"):
    """LlavaNextPlugin.process_messages (mm_plugin.py:327-366) for the LLaVA-OneVision processor: every "<image>" in the message contents, in order, becomes
    tokens_of((height, width)) copies of the image token (the packed any-resolution feature count, `iadr1_amd.llava_ov.num_image_tokens`; the
    `vision_feature_select_strategy` is "full": no class-token correction).  Counts must match exactly."""
    out, used = [], 0
    for m in messages:
        content = m["content"]
        while IMAGE_PLACEHOLDER in content:
            if used >= len(image_sizes):
                raise ValueError("`len(images)` is less than the number of %s tokens." % IMAGE_PLACEHOLDER)
            content = content.replace(IMAGE_PLACEHOLDER, "{{image}}" * int(tokens_of(tuple(int(v) for v in image_sizes[used]))), 1)
            used += 1
        out.append({**m, "content": content.replace("{{image}}", image_token)})
    if used != len(image_sizes):
        raise ValueError("The number of images does not match the number of %s tokens." % IMAGE_PLACEHOLDER)
    return out


def expand_image_placeholders(messages, grids, merge_size: int = 2, image_token: str = "<|image_pad|>"):
    """Every "<image>" in the message contents, in order, becomes <|vision_start|> + t*h*w / merge_size^2 image tokens + <|vision_end|> for the
    corresponding entry of `grids` ([t, h, w] patch grids).  Counts must match exactly."""
    out, used = [], 0
    for m in messages:
        content = m["content"]
        while IMAGE_PLACEHOLDER in content:
            if used >= len(grids):
                raise ValueError("`len(images)` is less than the number of %s tokens." % IMAGE_PLACEHOLDER)
            t, h, w = (int(v) for v in grids[used])
            content = content.replace(IMAGE_PLACEHOLDER, "<|vision_start|>" + image_token * (t * h * w // (merge_size * merge_size)) + "<|vision_end|>", 1)
            used += 1
        out.append({**m, "content": content})
    if used != len(grids):
        raise ValueError("The number of images does not match the number of %s tokens." % IMAGE_PLACEHOLDER)
    return out


def qwen2_vl_turn_texts(messages, system: Optional[str] = None):
    """ChatML rendering of the "qwen2_vl" template as [(prompt_pieces, answer_pieces)] per turn; every piece is tokenised on its own (that is what the
    reference does, so the system block and the user block never merge across their boundary).  Turn 0 carries the system block; the assistant's
    closing "<|im_end|>\\n" belongs to the answer and is therefore supervised."""
    if len(messages) % 2:
        raise ValueError("a conversation is user / assistant pairs")
    system = system or QWEN2_VL_DEFAULT_SYSTEM
    rendered = []
    for i, m in enumerate(messages):
        pieces = ["<|im_start|>system\n" + system + "<|im_end|>\n"] if i == 0 and system else []
        if m["role"] == "user":
            pieces.append("<|im_start|>user\n" + m["content"] + "<|im_end|>\n<|im_start|>assistant\n")
        elif m["role"] == "assistant":
            pieces.append(m["content"] + "<|im_end|>\n")
        elif m["role"] == "observation":
            pieces.append("<|im_start|>user\n<tool_response>\n" + m["content"] + "\n</tool_response><|im_end|>\n<|im_start|>assistant\n")
        else:
            raise NotImplementedError("Unexpected role: %s (tool calls are not part of the IAD-R1 data)" % m["role"])
        rendered.append(pieces)
    return [(rendered[i], rendered[i + 1]) for i in range(0, len(rendered), 2)]


# "llava_next_qwen" (llamafactory template.py:899-913, "copied from chatml template"): the same ChatML turns and the same default system prompt as "qwen2_vl"
llava_next_qwen_turn_texts = qwen2_vl_turn_texts


class _Special:
    """A special-token element of a template ({"bos_token"} / {"eos_token"} in llamafactory's slot lists): resolved against the tokenizer, dropped when it has none."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return f"<{self.name}>"


BOS, EOS = _Special("bos_token"), _Special("eos_token")
LLAVA_DEFAULT_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. "
                        "The assistant gives helpful, detailed, and polite answers to the user's questions.")


def llava_turn_texts(messages, system: Optional[str] = None):
    """The "llava" template (llamafactory template.py:832-841, "copied from vicuna template", the LLaVA-1.5 script): turn 0 = the system text as its own piece,
    then "USER: {content} ASSISTANT:"; the answer is the content followed by the EOS token."""
    if len(messages) % 2:
        raise ValueError("a conversation is user / assistant pairs")
    system = system or LLAVA_DEFAULT_SYSTEM
    rendered = []
    for i, m in enumerate(messages):
        pieces = [system] if i == 0 and system else []
        if m["role"] in ("user", "observation"):
            pieces.append("USER: " + m["content"] + " ASSISTANT:")
        elif m["role"] == "assistant":
            pieces += [m["content"], EOS]
        else:
            raise NotImplementedError("Unexpected role: %s (tool calls are not part of the IAD-R1 data)" % m["role"])
        rendered.append(pieces)
    return [(rendered[i], rendered[i + 1]) for i in range(0, len(rendered), 2)]


def llava_next_mistral_turn_texts(messages, system: Optional[str] = None):
    """The "llava_next_mistral" template (llamafactory template.py:885-896 with Llama2Template._encode :165-203, the LLaVA-1.6 script): turn 0 = BOS, then
    "[INST] " + (system + two newlines, when the row has a system prompt; there is no default one) + content + "[/INST]"; the answer is " " + content + EOS."""
    if len(messages) % 2:
        raise ValueError("a conversation is user / assistant pairs")
    rendered = []
    for i, m in enumerate(messages):
        pieces = [BOS] if i == 0 else []
        system_text = (system + "\n\n") if (i == 0 and system) else ""
        if m["role"] == "user":
            pieces.append("[INST] " + system_text + m["content"] + "[/INST]")
        elif m["role"] == "assistant":
            pieces += [" " + m["content"], EOS]
        elif m["role"] == "observation":
            pieces.append("[TOOL_RESULTS] {\"content\": " + m["content"] + "}[/TOOL_RESULTS]")
        else:
            raise NotImplementedError("Unexpected role: %s (tool calls are not part of the IAD-R1 data)" % m["role"])
        rendered.append(pieces)
    return [(rendered[i], rendered[i + 1]) for i in range(0, len(rendered), 2)]


TURN_TEXTS = {"qwen2_vl": qwen2_vl_turn_texts, "llava_next_qwen": llava_next_qwen_turn_texts, "llava": llava_turn_texts, "llava_next_mistral": llava_next_mistral_turn_texts}


def encode_turns(tokenizer, turn_texts):
    """[(prompt_pieces, answer_pieces)] -> [(prompt_ids, answer_ids)]: every text piece through tokenizer.encode(piece, add_special_tokens=False), BOS / EOS elements
    as the tokenizer's ids (llamafactory template.py:141-160)."""
    def enc(pieces):
        out = []
        for p in pieces:
            if isinstance(p, _Special):
                tid = getattr(tokenizer, p.name + "_id", None)
                if tid is not None:
                    out.append(tid)
            elif p:
                out += tokenizer.encode(p, add_special_tokens=False)
        return out
    return [(enc(s), enc(t)) for s, t in turn_texts]
