import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
L = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libstream_probe.so"))
dev = "cuda"
names = {0: "NB8 W8 Wonly d1", 1: "NB8 W8 Wonly d2", 2: "NB8 W8 Wonly d4", 3: "NB8 W8 W+X d2", 4: "NB8 W8 full d2", 5: "NB4 W8 Wonly d4", 6: "NB4 W16 Wonly d2", 7: "NB4 W16 full d2", 8: "NB2 W16 Wonly d4", 9: "NB4 W8 full d4", 10: "NB8 W4 Wonly d4", 11: "NB8 W16 Wonly d2"}
for N, K in [(22016, 2048), (151936, 2048), (2048, 11008)]:
    W = torch.randn(N * K // 2, device=dev).view(torch.int32)  # N*K bf16 worth of bytes
    X = torch.randn(64, K, device=dev).to(torch.bfloat16)
    out = torch.zeros(20000, device=dev)
    for v in range(12):
        def run():
            L.run_stream(v, ctypes.c_void_p(W.data_ptr()), ctypes.c_void_p(X.data_ptr()), ctypes.c_void_p(out.data_ptr()), N, K, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"N={N:6d} K={K:5d} {names[v]:18s} {us:8.1f} us {N*K*2/us/1e6:6.2f} TB/s", flush=True)
