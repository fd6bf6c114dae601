"""Host input pipeline of the SC-GRPO step.  The reference builds every micro-batch INSIDE `compute_loss`
(/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:600-625: chat template, PIL decode, one processor call), i.e. on the critical path of the
step: SURVEY.md section 8(a) a23 measured 0.09 s per 448 x 448 image for `Qwen2VLImageProcessor`, ~0.7 s of serial host time for 8 prompts against a
1.27 s GPU step.  Here the SAME function (`trainer.prepare_batch`, unchanged) runs for micro-batch k+1 on one worker thread while the GPU executes
micro-batch k, and the pixel tensor goes to HBM through pinned memory on a copy stream.  The sampler order is fixed before any batch is prepared, so
the run is bit-identical to the inline form (tests/test_hip_model.py::test_trainer_prefetch_is_bit_identical_to_inline_preparation)."""
from __future__ import annotations

from concurrent.futures import Future, ThreadPoolExecutor

import torch


class BatchPrefetcher:
    """submit(inputs) -> Future of the prepared batch dict; `ready(batch)` (consumer thread, current stream) orders the stream behind the upload."""

    def __init__(self, prepare_fn, device=None):
        self.prepare_fn = prepare_fn
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda"
        self.copy_stream = torch.cuda.Stream(self.device) if self.cuda else None
        self._pool = ThreadPoolExecutor(max_workers=1, thread_name_prefix="iadr1-prefetch")      # ONE worker: batches come back in submission order

    def submit(self, inputs) -> Future:
        return self._pool.submit(self._work, inputs)

    def _work(self, inputs):
        batch = self.prepare_fn(inputs)
        px = batch.get("pixel_values") if isinstance(batch, dict) else None
        if self.cuda and isinstance(px, torch.Tensor) and not px.is_cuda:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.copy_stream):
                dev = px.pin_memory().to(self.device, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self.copy_stream)
            batch = dict(batch, pixel_values=dev, _upload_event=ev)
        return batch

    @staticmethod
    def ready(batch):
        """Call on the thread / stream that will consume the batch: waits (on the stream, not the host) for the upload."""
        ev = batch.pop("_upload_event", None) if isinstance(batch, dict) else None
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            batch["pixel_values"].record_stream(cur)       # allocated on the copy stream's pool, read on this one
        return batch

    def shutdown(self):
        self._pool.shutdown(wait=True, cancel_futures=True)
