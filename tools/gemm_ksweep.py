import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import iadr1_amd
from iadr1_amd import ops
dev="cuda"
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M,N=20480,22016
prev=None
for K in (1024, 2048, 3072, 4096, 6144, 8192, 16384):
    a = torch.randn(M, K, device=dev).to(torch.bfloat16); b = torch.randn(N, K, device=dev).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    t = timeit(lambda: ops.gemm_nt(a, b, out=out))
    print(f"K={K:6d} {t:9.1f} us  {2.0*M*N*K/t/1e6:7.1f} TF  per K-tile per wave {t/ (K/64) / 26.875:6.3f} us" + (f"  marginal {(t-prev[1])/((K-prev[0])/64)/26.875:6.3f} us/Ktile" if prev else ""), flush=True)
    prev=(K,t)
    del a,b,out
