// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 operand layout (fp8 e4m3 x e4m3, unit scales).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
// layout candidate c: byte index of element (row r, k) inside lane (r + 16*g)'s 32 bytes
__device__ __host__ inline void where(int cand, int k, int* g, int* byte) {
    if (cand == 0) { *g = k / 32; *byte = k % 32; }                                  // 32 contiguous k per lane group
    else if (cand == 1) { *g = (k % 64) / 16; *byte = (k / 64) * 16 + (k % 16); }    // two 64-deep halves, 16 contiguous each
    else { *g = (k % 32) / 8; *byte = (k / 32) * 8 + (k % 8); }                      // four 32-deep quarters, 8 contiguous each
}
__global__ void probe(const uint8_t* A, const uint8_t* B, float* C, int cand, int scale_word) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    uint8_t a[32], b[32];
    for (int k = 0; k < 128; ++k) { int gg, by; where(cand, k, &gg, &by); if (gg == g) { a[by] = A[r * 128 + k]; b[by] = B[r * 128 + k]; } }
    v8i av, bv;
    for (int i = 0; i < 8; ++i) { av[i] = a[4*i] | (a[4*i+1] << 8) | (a[4*i+2] << 16) | ((uint32_t)a[4*i+3] << 24); bv[i] = b[4*i] | (b[4*i+1] << 8) | (b[4*i+2] << 16) | ((uint32_t)b[4*i+3] << 24); }
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, acc, 0, 0, 0, scale_word, 0, scale_word);
    // C/D: col = lane & 15, row = (lane >> 4) * 4 + e    (D[i][j] = sum_k A[i][k] B[j][k]: A row i, B row j)
    for (int e = 0; e < 4; ++e) C[(g * 4 + e) * 16 + r] = acc[e];
}
static float e4m3(uint8_t v) { int s = v >> 7, e = (v >> 3) & 15, m = v & 7; float x = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7); return s ? -x : x; }
int main() {
    uint8_t hA[16*128], hB[16*128];
    srand(1);
    for (int i = 0; i < 16*128; ++i) { hA[i] = (rand() % 0x78) | ((rand() & 1) << 7); hB[i] = (rand() % 0x78) | ((rand() & 1) << 7); }   // finite values only (0x7F = NaN)
    double ref[16][16];
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { double s = 0; for (int k = 0; k < 128; ++k) s += (double)e4m3(hA[i*128+k]) * e4m3(hB[j*128+k]); ref[i][j] = s; }
    uint8_t *dA, *dB; float* dC; hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hB); hipMalloc(&dC, 1024);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int cand = 0; cand < 3; ++cand) for (int sw : {0x7F7F7F7F, 0}) {
        float hC[256]; hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, cand, sw); hipMemcpy(hC, dC, 1024, hipMemcpyDeviceToHost);
        double e1 = 0, e2 = 0, mx = 0;   // e1: D[i][j] = hC[i*16+j]; e2: transposed
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { e1 = fmax(e1, fabs(hC[i*16+j] - ref[i][j])); e2 = fmax(e2, fabs(hC[j*16+i] - ref[i][j])); mx = fmax(mx, fabs(ref[i][j])); }
        printf("cand %d scale %08x: max|err| as [row=(l>>4)*4+e][col=l&15] = A-row x B-row: %.4g, transposed: %.4g (max |ref| %.4g, C[0] %.4g ref %.4g)\n", cand, sw, e1, e2, mx, hC[0], ref[0][0]);
    }
    return 0;
}
