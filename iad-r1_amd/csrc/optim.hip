// Fused flat AdamW + global grad-norm (gfx950, HBM-bound; SURVEY.md section 2.3 K19: HF Trainer's default
// torch.optim.AdamW, betas (0.9, 0.999), eps 1e-8, decoupled weight decay, max_grad_norm clipping).
// All parameters live in ONE flat buffer per dtype, so a step is two launches, not thousands:
//   sumsq:  norm2 += sum(g^2)                       (clip coefficient computed on device, no host sync)
//   adamw:  m,v,master (fp32) updated from fp32 grads; bf16 shadow parameter re-materialised; grad zeroed.
#include "common.h"

namespace {

// Deterministic two-stage reduction (no float atomics): data-parallel replicas must compute a BIT-IDENTICAL clip
// coefficient from their identical all-reduced gradients, or they drift apart.
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* g, long long n, float* partial) {
    __shared__ float scratch[16];
    float s = 0.f;
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const f32x4_t v = *(const f32x4_t*)(g + i * 4);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    for (long long i = n4 * 4 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
    s = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(const float* partial, int nblocks, float* out) {
    __shared__ float scratch[16];
    float s = 0.f;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += partial[i];
    s = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) out[0] = s;
}

struct AdamArgs {
    float* master;
    float* m;
    float* v;
    float* grad;
    bf16_t* param;
    long long n;
    float lr, b1, b2, eps, wd, bc1, bc2;  // bc = 1 - beta^t
    float grad_scale;                     // e.g. 1/world_size or 1/accum
    const float* norm2;                   // optional: sum of squares of the (unscaled) grads
    float max_norm;                       // <=0: no clipping
};

__global__ __launch_bounds__(256) void adamw_kernel(AdamArgs p) {
    float scale = p.grad_scale;
    if (p.norm2 && p.max_norm > 0.f) {
        const float nrm = sqrtf(*p.norm2) * p.grad_scale;
        const float coef = p.max_norm / (nrm + 1e-6f);  // torch.nn.utils.clip_grad_norm_
        if (coef < 1.f) scale *= coef;
    }
    const float step = p.lr / p.bc1;
    const float isq = rsqrtf(p.bc2);
    for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < p.n; i += (long long)gridDim.x * 256 * 4) {
        if (i + 4 <= p.n) {
            f32x4_t g = *(const f32x4_t*)(p.grad + i), m = *(const f32x4_t*)(p.m + i), v = *(const f32x4_t*)(p.v + i), w = *(const f32x4_t*)(p.master + i);
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float ge = g[e] * scale;
                w[e] *= (1.f - p.lr * p.wd);
                m[e] = p.b1 * m[e] + (1.f - p.b1) * ge;
                v[e] = p.b2 * v[e] + (1.f - p.b2) * ge * ge;
                w[e] -= step * m[e] / (sqrtf(v[e]) * isq + p.eps);
                o[e] = w[e];
            }
            *(f32x4_t*)(p.m + i) = m;
            *(f32x4_t*)(p.v + i) = v;
            *(f32x4_t*)(p.master + i) = w;
            *(f32x4_t*)(p.grad + i) = (f32x4_t){0.f, 0.f, 0.f, 0.f};
            *(u32x2_t*)(p.param + i) = (u32x2_t){pack2bf(o[0], o[1]), pack2bf(o[2], o[3])};
        } else {
            for (long long j = i; j < p.n; ++j) {
                const float ge = p.grad[j] * scale;
                float w = p.master[j] * (1.f - p.lr * p.wd);
                const float m = p.b1 * p.m[j] + (1.f - p.b1) * ge;
                const float v = p.b2 * p.v[j] + (1.f - p.b2) * ge * ge;
                w -= step * m / (sqrtf(v) * isq + p.eps);
                p.m[j] = m; p.v[j] = v; p.master[j] = w; p.grad[j] = 0.f; p.param[j] = f2bf(w);
            }
        }
    }
}

}  // namespace

extern "C" int iadr1_sumsq(const float* g, long long n, float* partials2048, float* out, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && partials2048 != nullptr, "sumsq: empty input or missing 2048-float scratch");
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3((int)blocks), dim3(256), 0, stream, g, n, partials2048);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, stream, (const float*)partials2048, (int)blocks, out);
    return iadr1_check_launch("sumsq");
}

extern "C" int iadr1_adamw_flat(float* master, float* m, float* v, float* grad, void* param_bf16, long long n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float grad_scale, const float* norm2, float max_norm,
                                hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && step >= 1, "adamw: n>0 and step>=1 required");
    IADR1_REQUIRE((((uintptr_t)master | (uintptr_t)m | (uintptr_t)v | (uintptr_t)grad) & 15) == 0 && (((uintptr_t)param_bf16) & 7) == 0, "adamw: buffers must be 16-byte aligned");
    AdamArgs p{master, m, v, grad, (bf16_t*)param_bf16, n, lr, beta1, beta2, eps, weight_decay, 1.f - powf(beta1, (float)step), 1.f - powf(beta2, (float)step), grad_scale, norm2, max_norm};
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3((int)blocks), dim3(256), 0, stream, p);
    return iadr1_check_launch("adamw_flat");
}
