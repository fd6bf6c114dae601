#!/usr/bin/env python3
"""Per-iteration time of attn_bwd_dkdv on the shared-prefix training shape (stamps build: IADR1_HIP_LIB=.../libiadr1_hip_stamps.so)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import iadr1_amd  # noqa
from iadr1_amd import hip, ops
dev = "cuda"
L = hip.lib()
Hq, Hkv, D = 16, 2, 128
ng, P, G, C = 8, 512, 8, 256
n = ng * G
starts = [b * P for b in range(ng)] + [ng * P + r * C for r in range(n)]
ends = [b * P + P for b in range(ng)] + [ng * P + r * C + C for r in range(n)]
prefix = [[0, 0, ng + b * G, G] for b in range(ng)] + [[(r // G) * P, P, 0, 0] for r in range(n)]
seg = ops.Segments(starts, ends, dev, prefix=prefix)
T = ng * P + n * C
qkv = torch.randn(T, (Hq + 2 * Hkv) * D, device=dev).to(torch.bfloat16)
q, k, v = qkv[:, :Hq*D], qkv[:, Hq*D:(Hq+Hkv)*D], qkv[:, (Hq+Hkv)*D:]
o, lse = ops.attn_fwd(q, k, v, seg, Hq, Hkv, D, True, D ** -0.5)
do = torch.randn_like(o); dqkv = torch.zeros_like(qkv)
for _ in range(3):
    ops.attn_bwd(q, k, v, o, do, lse, seg, Hq, Hkv, D, True, D ** -0.5, dqkv[:, :Hq*D], dqkv[:, Hq*D:(Hq+Hkv)*D], dqkv[:, (Hq+Hkv)*D:])
buf = (ctypes.c_ulonglong * (8 * 4096))()
fn = L.iadr1_debug_stamps_attn; fn.argtypes, fn.restype = [ctypes.c_void_p], ctypes.c_int
assert fn(buf) == 0
s = np.frombuffer(buf, dtype=np.uint64).reshape(8, 4096).astype(np.int64)
nit, t0, t1 = s[6], s[4], s[5]
ok = (nit > 0) & (t1 > t0)
print("blocks with iterations (of the 4096-slot table):", int(ok.sum()))
for lo, hi in ((1, 24), (24, 400)):
    m = ok & (nit >= lo) & (nit < hi)
    if m.any():
        per = (t1[m] - t0[m]) / 100.0 / nit[m]
        print(f"  nit in [{lo},{hi}): {int(m.sum())} blocks, nit median {int(np.median(nit[m]))}, us per iteration median {np.median(per):.2f} p10 {np.percentile(per, 10):.2f} p90 {np.percentile(per, 90):.2f};"
              f" block duration median {np.median((t1[m] - t0[m]) / 100.0):.1f} us; starts spread {(t0[m].max() - t0[m].min()) / 100.0:.1f} us")
m = ok & (nit >= 24)
if m.any():
    for k, lab in enumerate(["barrier + register tile -> LDS + barrier (incl. waiting for the prefetched loads)", "S / dP: fragment reads + 16 MFMA", "softmax terms (exp, pack)", "dV / dK: transpose reads + 16 MFMA"]):
        print(f"  phase {k} {lab}: median {np.median(s[k][m] / 100.0 / nit[m]):.3f} us per iteration")
occ = (ctypes.c_int * 6)()
L.iadr1_debug_attn_occupancy.argtypes, L.iadr1_debug_attn_occupancy.restype = [ctypes.c_void_p], ctypes.c_int
L.iadr1_debug_attn_occupancy(occ)
print("blocks per CU admitted (dkdv<4 waves>, dkdv<8 waves>, dq<R=1>, dq<R=2>, fwd<R=1>, fwd<R=2>):", list(occ))
hv = ok & (nit >= 24)
lt = ok & (nit < 24)
base = t0[ok].min()
print("heavy blocks seen:", int(hv.sum()), "start [us] min/max", (t0[hv].min() - base) / 100.0, (t0[hv].max() - base) / 100.0, "end min/max", (t1[hv].min() - base) / 100.0, (t1[hv].max() - base) / 100.0)
print("light blocks seen:", int(lt.sum()), "start [us] min/median/max", (t0[lt].min() - base) / 100.0, float(np.median(t0[lt] - base)) / 100.0, (t0[lt].max() - base) / 100.0, "end max", (t1[lt].max() - base) / 100.0)
print("kernel span (first start -> last end):", (t1[ok].max() - t0[ok].min()) / 100.0, "us")
