"""iadr1_amd: MI355X-native engine for the IAD-R1 post-training hot path (PA-SFT / SC-GRPO on
Qwen2.5-VL): Python host code on PyTorch-ROCm (device memory, streams, torch.distributed only)
over a C-ABI library of hand-written gfx950 HIP kernels (csrc/, include/iadr1_hip.h).

Sub-modules import lazily; the HIP library is loaded by ``iadr1_amd.hip`` and its absence is a
hard error on any compute call (there is no CPU fallback in this package)."""
__version__ = "0.1.0"
