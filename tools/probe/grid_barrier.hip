// Probe (NOT product): cost of a hand-rolled grid-wide barrier on gfx950 (256 resident blocks, one per CU), the building block of a
// single-launch decode layer.  Variants: 0 flat counter, 1 per-XCD counter + global generation, 2 flat + 4 KB of cross-block traffic per round
// (checks visibility), 3 = 2 with the hierarchical barrier.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int SPIN_LIMIT = 1 << 22;

__device__ __forceinline__ bool barrier_flat(uint32_t* ctr, uint32_t target, uint32_t* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// ctrs[0..7]: per-XCD arrival counters (blocks b with b % 8 == x live on XCD x), ctrs[16]: global generation
__device__ __forceinline__ bool barrier_xcd(uint32_t* ctrs, uint32_t round, uint32_t per_xcd, uint32_t* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        const uint32_t x = blockIdx.x & 7;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        const uint32_t old = __hip_atomic_fetch_add(ctrs + x * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == round * per_xcd) __hip_atomic_fetch_add(ctrs + 16 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int spins = 0;
        while (__hip_atomic_load(ctrs + 16 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round * 8) {
            if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    return ok;
}

// flags[b] = last round block b has arrived at; every block polls all flags with its first nb threads (plain stores, no read-modify-write)
__device__ __forceinline__ bool barrier_flags(uint32_t* flags, uint32_t round, int nb, uint32_t* err) {
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_store(flags + blockIdx.x, round, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)threadIdx.x < nb) {
        int spins = 0;
        while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < round) {
            if (++spins > SPIN_LIMIT) { *err = 1; ok = false; break; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    return ok;
}

template <int VARIANT>
__global__ __launch_bounds__(512) void barrier_loop(uint32_t* ctrs, uint32_t* err, uint32_t* data, int rounds, uint32_t base) {
    const int nb = gridDim.x;
    uint32_t acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (VARIANT >= 2 && VARIANT != 4) {   // every block writes 4 KB, after the barrier reads the 4 KB of block (b + 37) % nb and checks it
            uint32_t* mine = data + ((size_t)(r & 1) * nb + blockIdx.x) * 1024;
            mine[threadIdx.x] = (uint32_t)(r * 1000003u + blockIdx.x * 1024 + threadIdx.x);
            mine[threadIdx.x + 512] = (uint32_t)(r * 1000003u + blockIdx.x * 1024 + threadIdx.x + 512);
        }
        bool ok;
        if (VARIANT == 4 || VARIANT == 5) ok = barrier_flags(ctrs, base / nb + r, nb, err);
        else if (VARIANT == 0 || VARIANT == 2) ok = barrier_flat(ctrs + 20 * 32, base + (uint32_t)r * nb, err);
        else ok = barrier_xcd(ctrs, base / nb + r, nb / 8, err);
        if (!ok) return;
        if (VARIANT >= 2 && VARIANT != 4) {
            const int ob = (blockIdx.x + 37) % nb;
            const uint32_t* theirs = data + ((size_t)(r & 1) * nb + ob) * 1024;
            const uint32_t v0 = __builtin_nontemporal_load(theirs + threadIdx.x), v1 = __builtin_nontemporal_load(theirs + threadIdx.x + 512);
            acc += (v0 != (uint32_t)(r * 1000003u + ob * 1024 + threadIdx.x)) + (v1 != (uint32_t)(r * 1000003u + ob * 1024 + threadIdx.x + 512));
        }
    }
    if (acc) atomicAdd(err + 1, acc);
}
}  // namespace

extern "C" int run_barrier(int variant, uint32_t* ctrs, uint32_t* err, uint32_t* data, int rounds, uint32_t base, int blocks, hipStream_t st) {
    switch (variant) {
        case 0: hipLaunchKernelGGL(barrier_loop<0>, dim3(blocks), dim3(512), 0, st, ctrs, err, data, rounds, base); break;
        case 1: hipLaunchKernelGGL(barrier_loop<1>, dim3(blocks), dim3(512), 0, st, ctrs, err, data, rounds, base); break;
        case 2: hipLaunchKernelGGL(barrier_loop<2>, dim3(blocks), dim3(512), 0, st, ctrs, err, data, rounds, base); break;
        case 3: hipLaunchKernelGGL(barrier_loop<3>, dim3(blocks), dim3(512), 0, st, ctrs, err, data, rounds, base); break;
        case 4: hipLaunchKernelGGL(barrier_loop<4>, dim3(blocks), dim3(512), 0, st, ctrs, err, data, rounds, base); break;
        case 5: hipLaunchKernelGGL(barrier_loop<5>, dim3(blocks), dim3(512), 0, st, ctrs, err, data, rounds, base); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
