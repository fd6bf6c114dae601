"""Host-side integer / index work of the LLaVA-OneVision branch of the reference's model switch
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:124-132 -> transformers LlavaOnevisionForConditionalGeneration, the
reference's pinned dependency; TF: = transformers/models/llava_onevision/modeling_llava_onevision.py as installed, 5.15.0):

  * any-resolution crop grid of an image (TF:153-218 -> image_processing_utils.select_best_resolution): which of the configured
    `image_grid_pinpoints` an image is fitted to, how many 384x384 crops (+ the base image) the processor hands over;
  * `pack_image_features` (TF:280-348): base-image features, then the crop features re-assembled on the (rows x cols) grid of the
    fitted resolution, un-padded to the image's aspect ratio (TF:221-262), bilinearly shrunk when they exceed `anyres_max_N` crops'
    worth of tokens, one `image_newline` vector appended to every feature row -- restated as ONE sparse linear map
        packed[t] = sum_k w[t,k] * source[idx[t,k]],   source = [projector output rows ; image_newline]
    (one entry of weight 1 per token, four for an interpolated token) in CSR form, together with its transpose for the backward pass;
    the device applies either with `iadr1_rows_gather_sum`.  No floating-point work happens here except the interpolation weights,
    which follow torch.nn.functional.interpolate(mode="bilinear", align_corners=False).
"""
from __future__ import annotations

import math

import numpy as np


def select_best_resolution(original_size, possible_resolutions):
    """image_processing_utils.select_best_resolution: (height, width) of the pinpoint with the largest effective and least wasted area."""
    oh, ow = int(original_size[0]), int(original_size[1])
    best, max_eff, min_waste = None, 0, float("inf")
    for h, w in possible_resolutions:
        scale = min(w / ow, h / oh)
        dw, dh = int(ow * scale), int(oh * scale)
        eff = min(dw * dh, ow * oh)
        waste = w * h - eff
        if eff > max_eff or (eff == max_eff and waste < min_waste):
            max_eff, min_waste, best = eff, waste, (h, w)
    return best


def anyres_grid_shape(image_size, pinpoints, crop):
    """TF:153-181 -> (crop rows, crop columns) of the fitted resolution."""
    h, w = select_best_resolution(image_size, pinpoints)
    return h // crop, w // crop


def num_crops(image_size, pinpoints, crop):
    """TF:184-218: crops of the fitted resolution + the base image."""
    gh, gw = anyres_grid_shape(image_size, pinpoints, crop)
    return gh * gw + 1


def _unpad_window(cur_h, cur_w, orig_h, orig_w):
    """TF:221-262 unpad_image on a (cur_h x cur_w) feature map: the kept row / column range."""
    if orig_w / orig_h > cur_w / cur_h:
        new_h = int(round(orig_h * (cur_w / orig_w), 7))
        pad = (cur_h - new_h) // 2
        return pad, cur_h - pad, 0, cur_w
    new_w = int(round(orig_w * (cur_h / orig_h), 7))
    pad = (cur_w - new_w) // 2
    return 0, cur_h, pad, cur_w - pad


def _bilinear_taps(n_in, n_out):
    """Source indices and weights of torch's bilinear resize (align_corners=False, no antialias) along one axis: for every output
    position two taps (i0, i1) with weights (1 - l, l)."""
    scale = n_in / n_out
    src = (np.arange(n_out, dtype=np.float64) + 0.5) * scale - 0.5
    src = np.maximum(src, 0.0)
    i0 = np.minimum(np.floor(src).astype(np.int64), n_in - 1)
    i1 = np.minimum(i0 + 1, n_in - 1)
    lam = (src - i0).astype(np.float64)
    return i0, i1, 1.0 - lam, lam


def pack_plan(image_sizes, pinpoints, crop, side, max_patches=9):
    """CSR of pack_image_features for a list of images.

    image_sizes: [(height, width)] per image; `crop` = vision image_size (384), `side` = feature-map side of one crop (27); max_patches: the N of
    `vision_aspect_ratio = anyres_max_N` (LLaVA-OneVision), None for LLaVA-NeXT (transformers/models/llava_next/modeling_llava_next.py:265-330: the same
    assembly without the bilinear shrink).
    The source rows are numbered crop-major: image i owns rows [first_i, first_i + num_crops_i * side^2) in processor order (base image first);
    the row after the last image's rows is the `image_newline` vector.
    Returns dict(ptr [T+1] int32, idx [nnz] int32, w [nnz] float32, lens [n_images], n_src (incl. the newline row)) -- T = total packed tokens."""
    per = side * side
    crops = [num_crops(s, pinpoints, crop) for s in image_sizes]
    first = np.concatenate([[0], np.cumsum([c * per for c in crops])]).astype(np.int64)
    newline = int(first[-1])
    ptr, idx, wts, lens = [0], [], [], []

    def emit(entries):
        for i_, w_ in entries:
            idx.append(int(i_))
            wts.append(float(w_))
        ptr.append(len(idx))

    for im, (size, nc) in enumerate(zip(image_sizes, crops)):
        f0 = int(first[im])
        n0 = len(ptr) - 1
        for t in range(per):                                   # base image features first (TF:331)
            emit([(f0 + t, 1.0)])
        gh, gw = anyres_grid_shape(size, pinpoints, crop)
        H, W = gh * side, gw * side
        # feature (y, x) of the assembled map lives in crop (y // side, x // side), row-major inside the crop (TF:312-314)
        src_of = lambda y, x: f0 + per * (1 + (y // side) * gw + (x // side)) + (y % side) * side + (x % side)
        y0, y1, x0, x1 = _unpad_window(H, W, int(size[0]), int(size[1]))
        ch, cw = y1 - y0, x1 - x0
        ratio = math.sqrt(ch * cw / (max_patches * side**2)) if max_patches else 0.0      # max_patches None: LLaVA-NeXT's packing (LN:265-330) has no shrink step
        if ratio > 1.1:                                        # TF:318-323: bilinear shrink to (ch // ratio, cw // ratio)
            oh, ow = int(ch // ratio), int(cw // ratio)
            ya, yb, wya, wyb = _bilinear_taps(ch, oh)
            xa, xb, wxa, wxb = _bilinear_taps(cw, ow)
            for y in range(oh):
                for x in range(ow):
                    taps = {}
                    for yy, wy in ((ya[y], wya[y]), (yb[y], wyb[y])):
                        for xx, wx in ((xa[x], wxa[x]), (xb[x], wxb[x])):
                            if wy * wx != 0.0:
                                k = src_of(y0 + int(yy), x0 + int(xx))
                                taps[k] = taps.get(k, 0.0) + wy * wx
                    emit(sorted(taps.items()))
                emit([(newline, 1.0)])                         # one newline vector closes every feature row (TF:324-333)
        else:
            for y in range(y0, y1):
                for x in range(x0, x1):
                    emit([(src_of(y, x), 1.0)])
                emit([(newline, 1.0)])
        lens.append(len(ptr) - 1 - n0)
    return {"ptr": np.asarray(ptr, dtype=np.int32), "idx": np.asarray(idx, dtype=np.int32), "w": np.asarray(wts, dtype=np.float32),
            "lens": lens, "n_src": newline + 1, "crops": crops}


def transpose_plan(plan):
    """CSR of the transposed map (gradient of the packed tokens -> gradient of [projector rows ; newline]); entries of a row in token order."""
    T = len(plan["ptr"]) - 1
    rows = np.repeat(np.arange(T, dtype=np.int64), np.diff(plan["ptr"]))
    order = np.argsort(plan["idx"], kind="stable")
    ptr = np.zeros(plan["n_src"] + 1, dtype=np.int32)
    np.cumsum(np.bincount(plan["idx"], minlength=plan["n_src"]), out=ptr[1:])
    return {"ptr": ptr, "idx": rows[order].astype(np.int32), "w": plan["w"][order].astype(np.float32)}


def num_image_tokens(image_size, pinpoints, crop, side, max_patches=9):
    """Tokens the processor must reserve for one image (= its packed feature count)."""
    return pack_plan([image_size], pinpoints, crop, side, max_patches)["lens"][0]
