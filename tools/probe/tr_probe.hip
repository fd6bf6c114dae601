// Hardware probe (NOT product code): lane->element mapping of ds_read_b64_tr_b16 on gfx950, used to
// design the native transposed-operand paths.  Built and run only by tools/gpu_probe.py.
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ void tr_probe(uint16_t* out, int mode) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
    __syncthreads();
    const int l = threadIdx.x;
    uint32_t off;  // byte offset inside lds
    if (mode == 0) off = l * 8;                                  // every lane its own consecutive 8-byte chunk
    else if (mode == 1) off = (l & 15) * 32 + (l >> 4) * 8;      // 16 rows of 32 B, lane group picks the 8-B column
    else if (mode == 2) off = ((l & 15) + (l >> 4) * 64) * 2;    // formula quoted in the programming guide
    else off = (l & 15) * 128 + (l >> 4) * 8;                    // 16 rows of 128 B
    const uint32_t addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint16_t*)lds + off;
    uint64_t r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    for (int j = 0; j < 4; ++j) out[(mode * 64 + l) * 4 + j] = (uint16_t)(r >> (16 * j));
}
extern "C" int run_tr_probe(uint16_t* out_dev) {
    for (int mode = 0; mode < 4; ++mode) hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, 0, out_dev, mode);
    return (int)hipDeviceSynchronize();
}
