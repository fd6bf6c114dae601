// Probe (NOT product): a chain of dependent weight-streaming kernels shaped like one decode step (36 layers x 7 launches, each reading its weights from HBM and a
// small activation vector its predecessor wrote).  How much of the ~2.6 us launch boundary + HBM ramp per kernel comes back when the NEXT kernel is already
// resident and has its first weight loads in flight while it waits for its predecessor in software?
//   mode 0: one stream, hardware ordering (what a hipGraph replay of the decode step does today)
//   mode 1: two streams alternating (kernel i on stream i & 1), no events; kernel i spins on a counter its predecessor's blocks bump at their end
//   mode 2: one stream, hipExtAnyOrderLaunch (AQL barrier bit cleared) + the same software dependency
//   mode 3: as 1 without the prefetch-before-wait (separates "resident early" from "loads in flight early")
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

template <bool SW, bool PREFETCH>
__global__ __launch_bounds__(512) void chain_k(const u32x4_t* __restrict__ W, long long n16, const unsigned* xin, unsigned* xout, unsigned* flags, int idx, unsigned* err) {
    constexpr int U = 8;
    const long long stride = (long long)gridDim.x * 512;
    long long i = (long long)blockIdx.x * 512 + threadIdx.x;
    u32x4_t acc = {0, 0, 0, 0};
    u32x4_t v[U];
    bool have = false;
    if (PREFETCH && i + (U - 1) * stride < n16) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(W + i + u * stride);
        have = true;
    }
    if (SW && idx > 0) {
        if (threadIdx.x == 0) {
            const unsigned want = gridDim.x;
            int spins = 0;
            while (ld_sc1(flags + idx - 1) < want) { __builtin_amdgcn_s_sleep(2); if (++spins > (1 << 22)) { err[0] = idx; break; } }
        }
        __syncthreads();
    }
    // the activation the predecessor wrote (device-scope load: the line may sit stale in this XCD's L2 from the previous replay)
    const unsigned x = __hip_atomic_load(xin + (threadIdx.x & 255), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (have) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
        i += U * stride;
    }
    for (; i + (U - 1) * stride < n16; i += U * stride) {
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(W + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    for (; i < n16; i += stride) acc ^= W[i];
    const unsigned r = (acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u ? 1u : 0u;
    if (blockIdx.x == 0 && threadIdx.x < 256) __hip_atomic_store(xout + threadIdx.x, x + 1 + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (SW) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(flags + idx, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// sizes[k] bytes for the k-th launch of a layer (7 per layer); W: base of `layers` consecutive layer buffers of layer_bytes each
extern "C" int run_chain(int mode, const void* W, long long layer_bytes, int layers, const long long* sizes, int per_layer, unsigned* xbuf, unsigned* flags, unsigned* err,
                         int grid, hipStream_t s0, hipStream_t s1) {
    const int nk = layers * per_layer;
    (void)hipMemsetAsync(flags, 0, sizeof(unsigned) * nk, s0);
    if (mode == 1 || mode == 3) {   // s1 must not start before the memset
        hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventRecord(e, s0); (void)hipStreamWaitEvent(s1, e, 0); (void)hipEventDestroy(e);
    }
    int k = 0;
    for (int l = 0; l < layers; ++l) {
        long long off = 0;
        for (int j = 0; j < per_layer; ++j, ++k) {
            const u32x4_t* w = (const u32x4_t*)((const char*)W + (long long)l * layer_bytes + off);
            const long long n16 = sizes[j] / 16;
            off += sizes[j];
            const unsigned* xin = xbuf + (k & 1) * 256;
            unsigned* xout = xbuf + ((k + 1) & 1) * 256;
            if (mode == 0) hipLaunchKernelGGL((chain_k<false, false>), dim3(grid), dim3(512), 0, s0, w, n16, xin, xout, flags, k, err);
            else if (mode == 1) hipLaunchKernelGGL((chain_k<true, true>), dim3(grid), dim3(512), 0, (k & 1) ? s1 : s0, w, n16, xin, xout, flags, k, err);
            else if (mode == 3) hipLaunchKernelGGL((chain_k<true, false>), dim3(grid), dim3(512), 0, (k & 1) ? s1 : s0, w, n16, xin, xout, flags, k, err);
            else hipExtLaunchKernelGGL((chain_k<true, true>), dim3(grid), dim3(512), 0, s0, nullptr, nullptr, hipExtAnyOrderLaunch, w, n16, xin, xout, flags, k, err);
        }
    }
    if (mode == 1 || mode == 3) {   // join s1 into s0 so that an event on s0 closes the chain
        hipEvent_t e; (void)hipEventCreateWithFlags(&e, hipEventDisableTiming); (void)hipEventRecord(e, s1); (void)hipStreamWaitEvent(s0, e, 0); (void)hipEventDestroy(e);
    }
    return (int)hipGetLastError();
}
