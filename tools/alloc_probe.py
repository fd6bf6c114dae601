import torch, time
dev="cuda"
def t(nbytes, keep=36):
    torch.cuda.synchronize()
    xs=[]
    t0=time.perf_counter()
    for i in range(keep): xs.append(torch.empty(nbytes, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize(); t1=time.perf_counter()
    del xs
    t2=time.perf_counter()
    xs=[torch.empty(nbytes, dtype=torch.uint8, device=dev) for i in range(keep)]
    torch.cuda.synchronize(); t3=time.perf_counter()
    # alloc/free cycling of one block
    del xs
    t4=time.perf_counter()
    for i in range(20):
        x=torch.empty(nbytes, dtype=torch.uint8, device=dev); del x
    torch.cuda.synchronize(); t5=time.perf_counter()
    print(f"{nbytes/2**30:.3f} GiB: first {keep} allocs {(t1-t0)*1e3:.1f} ms, re-alloc {(t3-t2)*1e3:.1f} ms, cycle20 {(t5-t4)*1e3:.2f} ms, reserved {torch.cuda.memory_reserved()/2**30:.1f} GiB", flush=True)
t(24576*22016*2)
t(12288*22016*2)
t(int(0.99*2**30))
t(int(1.01*2**30))
print(torch.cuda.memory_stats()["num_device_alloc"], torch.cuda.memory_stats()["num_device_free"])
