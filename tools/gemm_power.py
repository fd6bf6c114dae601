#!/usr/bin/env python3
"""Shader clock and socket power while gemm_nt streams back to back (is the GEMM family's rate a power / clock limit?): samples rocm-smi every 0.5 s during ~6 s of
[20480 x 22016 x 2048] launches, then during ~4 s of idle."""
import os, subprocess, sys, threading, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import iadr1_amd
from iadr1_amd import ops
dev = "cuda"
M, N, K = 20480, 22016, 2048
a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
samples, stop = [], False
def sample():
    while not stop:
        try:
            o = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showclocks", "--showpower", "--showtemp", "--csv"], capture_output=True, text=True, timeout=5).stdout
            samples.append((time.perf_counter(), o.strip().replace("\n", " || ")))
        except Exception as e:
            samples.append((time.perf_counter(), repr(e)))
        time.sleep(0.4)
th = threading.Thread(target=sample); th.start()
t0 = time.perf_counter()
n = 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
while time.perf_counter() - t0 < 6.0:
    for _ in range(50): ops.gemm_nt(a, b, out=out)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
t_busy = time.perf_counter()
print(f"{n} launches, {e0.elapsed_time(e1) / n * 1e3:.1f} us each, {2.0 * M * N * K * n / e0.elapsed_time(e1) / 1e9:.0f} TF sustained over {e0.elapsed_time(e1) / 1e3:.1f} s")
time.sleep(4.0)
stop = True; th.join()
for t, s in samples:
    print(f"t={t - t0:5.1f}s {'BUSY' if t < t_busy else 'idle'} {s[:600]}")
