#!/usr/bin/env python3
"""Probe driver (NOT product): per-round cost of a grid-wide barrier across 256 resident blocks (tools/probe/grid_barrier.hip)."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libgrid_barrier.so"))
dev = "cuda"
nb = torch.cuda.get_device_properties(0).multi_processor_count
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
names = {0: "flat counter", 1: "per-XCD + generation", 2: "flat + 4 KB exchange", 3: "per-XCD + 4 KB exchange", 4: "flags, all poll all", 5: "flags + 4 KB exchange"}
R = 2000
for v, nm in names.items():
    ctrs = torch.zeros(1024, dtype=torch.int32, device=dev); err = torch.zeros(4, dtype=torch.int32, device=dev)
    data = torch.zeros(2 * nb * 1024, dtype=torch.int32, device=dev)
    base = 0
    def go():
        global base
        rc = L.run_barrier(v, ctypes.c_void_p(ctrs.data_ptr()), ctypes.c_void_p(err.data_ptr()), ctypes.c_void_p(data.data_ptr()), R, ctypes.c_uint32(base), nb, st())
        assert rc == 0, rc
        base += R * nb
    go(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    print(f"{nm:28s} blocks {nb}  {e0.elapsed_time(e1) / R * 1e3:6.2f} us / round   err {err.tolist()}", flush=True)
