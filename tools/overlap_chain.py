#!/usr/bin/env python3
"""Probe driver (NOT product): tools/probe/overlap_chain.hip -- a decode-step-shaped chain of dependent weight-streaming kernels under hardware ordering vs
software dependencies with the successor already resident (two streams / any-order launch)."""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "liboverlap_chain.so"))
dev = "cuda"
MB = 1 << 20
# one 3B decoder layer of a decode step: RMSNorm (slabs), q|k|v, attention (KV of 64 sequences), o, RMSNorm, gate|up, down
sizes = [1 * MB, 10 * MB, 36 * MB, 8 * MB, 1 * MB, 90 * MB, 45 * MB]
layers = 36
layer_bytes = sum(sizes)
W = torch.empty(layers * layer_bytes, dtype=torch.uint8, device=dev)
W.random_(0, 255)
xbuf = torch.zeros(512, dtype=torch.int32, device=dev)
flags = torch.zeros(layers * len(sizes), dtype=torch.int32, device=dev)
err = torch.zeros(4, dtype=torch.int32, device=dev)
sz = (ctypes.c_longlong * len(sizes))(*sizes)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
names = {0: "one stream, hardware order", 1: "two streams + software dependency + prefetch", 3: "two streams + software dependency, no prefetch", 2: "any-order launch + software dependency + prefetch"}
def run(mode, grid):
    rc = L.run_chain(mode, ctypes.c_void_p(W.data_ptr()), ctypes.c_longlong(layer_bytes), layers, sz, len(sizes), ctypes.c_void_p(xbuf.data_ptr()), ctypes.c_void_p(flags.data_ptr()),
                     ctypes.c_void_p(err.data_ptr()), grid, ctypes.c_void_p(s0.cuda_stream), ctypes.c_void_p(s1.cuda_stream))
    assert rc == 0, rc
for grid in (256, 512):
    for mode in (0, 1, 3, 2):
        xbuf.zero_(); err.zero_()
        torch.cuda.synchronize()
        run(mode, grid); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        t0 = time.perf_counter()
        with torch.cuda.stream(s0):
            e0.record(s0)
            for _ in range(reps): run(mode, grid)
            e1.record(s0)
        t_host = (time.perf_counter() - t0) / reps
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        nk = layers * len(sizes)
        print(f"grid {grid} {names[mode]:52s} {ms:7.3f} ms per chain of {nk} ({ms / nk * 1e3:5.2f} us / kernel, {layers * layer_bytes / ms / 1e9:5.2f} TB/s)  host {t_host*1e3:6.2f} ms  x[0]={int(xbuf[0])} err={err.tolist()}", flush=True)
