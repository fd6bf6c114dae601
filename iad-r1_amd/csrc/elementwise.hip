// HBM-bound elementwise / gather kernels of the hot path (gfx950): rotary embeddings, SwiGLU, GELU,
// bias-gradient column sums, embedding gather + image-feature scatter, dtype casts, transposes.
// All use 16-byte accesses (8 bf16 per lane) and fp32 arithmetic.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Rotary embedding, in place on `nheads` consecutive heads of width D inside each token row.
//   out[j]      = x[j]*cos[j]      - sign*x[j+D/2]*sin[j]
//   out[j+D/2]  = x[j+D/2]*cos[j]  + sign*x[j]*sin[j]            (j < D/2)
// sign=+1: forward rotate-half form (TF:modeling_qwen2_5_vl.py:153-171 vision, :557-599 M-RoPE, whose
// per-section t/h/w selection is folded into the cos/sin table by the host); sign=-1: its transpose
// (backward).  cos/sin: fp32 [T, D/2].
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* x, long long ld, const float* cs, const float* sn, int T, int nheads,
                                                   int D, float sign) {
    const int half = D >> 1, cpr = half >> 2;  // 4-element chunks per half head
    const long long total = (long long)T * nheads * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % cpr);
        const long long th = i / cpr;
        const int h = (int)(th % nheads);
        const long long t = th / nheads;
        bf16_t* p = x + t * ld + (long long)h * D + c * 4;
        const u32x2_t lo = *(const u32x2_t*)p, hi = *(const u32x2_t*)(p + half);
        const f32x4_t cc = *(const f32x4_t*)(cs + t * half + c * 4), ss = *(const f32x4_t*)(sn + t * half + c * 4);
        const float a[4] = {lo_bf(lo[0]), hi_bf(lo[0]), lo_bf(lo[1]), hi_bf(lo[1])};
        const float b[4] = {lo_bf(hi[0]), hi_bf(hi[0]), lo_bf(hi[1]), hi_bf(hi[1])};
        float oa[4], ob[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            oa[e] = a[e] * cc[e] - sign * b[e] * ss[e];
            ob[e] = b[e] * cc[e] + sign * a[e] * ss[e];
        }
        *(u32x2_t*)p = (u32x2_t){pack2bf(oa[0], oa[1]), pack2bf(oa[2], oa[3])};
        *(u32x2_t*)(p + half) = (u32x2_t){pack2bf(ob[0], ob[1]), pack2bf(ob[2], ob[3])};
    }
}

// ------------------------------------------------------------------------------------------------
// SwiGLU (TF:modeling_qwen2_5_vl.py:95-96, 552-554): gu = [gate | up] halves of one row, width 2*I.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }

__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* gu, long long ldg, bf16_t* a, long long lda, int T, int I) {
    const int cpr = I >> 3;
    const long long total = (long long)T * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / cpr;
        const int c = (int)(i % cpr);
        const u32x4_t g = *(const u32x4_t*)(gu + t * ldg + c * 8), u = *(const u32x4_t*)(gu + t * ldg + I + c * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // the reference rounds act(gate) to bf16 before the multiply
            const float s0 = bf2f(f2bf(silu_f(lo_bf(g[e])))), s1 = bf2f(f2bf(silu_f(hi_bf(g[e]))));
            o[e] = pack2bf(s0 * lo_bf(u[e]), s1 * hi_bf(u[e]));
        }
        *(u32x4_t*)(a + t * lda + c * 8) = o;
    }
}

__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* da, long long lda, const bf16_t* gu, long long ldg, bf16_t* dgu,
                                                         long long ldd, int T, int I) {
    const int cpr = I >> 3;
    const long long total = (long long)T * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / cpr;
        const int c = (int)(i % cpr);
        const u32x4_t g = *(const u32x4_t*)(gu + t * ldg + c * 8), u = *(const u32x4_t*)(gu + t * ldg + I + c * 8);
        const u32x4_t d = *(const u32x4_t*)(da + t * lda + c * 8);
        u32x4_t og, ou;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r[2][2];
#pragma unroll
            for (int hsel = 0; hsel < 2; ++hsel) {
                const float gv = hsel ? hi_bf(g[e]) : lo_bf(g[e]), uv = hsel ? hi_bf(u[e]) : lo_bf(u[e]), dv = hsel ? hi_bf(d[e]) : lo_bf(d[e]);
                const float sg = 1.f / (1.f + __expf(-gv));
                const float sl = gv * sg;
                r[hsel][0] = dv * uv * (sg * (1.f + gv * (1.f - sg)));  // d gate
                r[hsel][1] = dv * sl;                                   // d up
            }
            og[e] = pack2bf(r[0][0], r[1][0]);
            ou[e] = pack2bf(r[0][1], r[1][1]);
        }
        *(u32x4_t*)(dgu + t * ldd + c * 8) = og;
        *(u32x4_t*)(dgu + t * ldd + I + c * 8) = ou;
    }
}

// exact GELU (nn.GELU() in the patch merger, TF:modeling_qwen2_5_vl.py:143)
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* z, bf16_t* a, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *(const u32x4_t*)(z + i * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = lo_bf(v[e]), a1 = hi_bf(v[e]);
            o[e] = pack2bf(0.5f * a0 * (1.f + erff(a0 * 0.70710678f)), 0.5f * a1 * (1.f + erff(a1 * 0.70710678f)));
        }
        *(u32x4_t*)(a + i * 8) = o;
    }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const bf16_t* da, const bf16_t* z, bf16_t* dz, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *(const u32x4_t*)(z + i * 8), d = *(const u32x4_t*)(da + i * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hsel = 0; hsel < 2; ++hsel) {
                const float x = hsel ? hi_bf(v[e]) : lo_bf(v[e]), dv = hsel ? hi_bf(d[e]) : lo_bf(d[e]);
                const float cdf = 0.5f * (1.f + erff(x * 0.70710678f));
                const float pdf = 0.3989422804f * __expf(-0.5f * x * x);
                r[hsel] = dv * (cdf + x * pdf);
            }
            o[e] = pack2bf(r[0], r[1]);
        }
        *(u32x4_t*)(dz + i * 8) = o;
    }
}

// QuickGELU x*sigmoid(1.702x) of the Qwen2-VL vision MLP (hidden_act "quick_gelu", TF:models/qwen2_vl/modeling_qwen2_vl.py:293-301)
__global__ __launch_bounds__(256) void quick_gelu_fwd_kernel(const bf16_t* z, bf16_t* a, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *(const u32x4_t*)(z + i * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a0 = lo_bf(v[e]), a1 = hi_bf(v[e]);
            o[e] = pack2bf(a0 / (1.f + __expf(-1.702f * a0)), a1 / (1.f + __expf(-1.702f * a1)));
        }
        *(u32x4_t*)(a + i * 8) = o;
    }
}
__global__ __launch_bounds__(256) void quick_gelu_bwd_kernel(const bf16_t* da, const bf16_t* z, bf16_t* dz, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *(const u32x4_t*)(z + i * 8), d = *(const u32x4_t*)(da + i * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float r[2];
#pragma unroll
            for (int hsel = 0; hsel < 2; ++hsel) {
                const float x = hsel ? hi_bf(v[e]) : lo_bf(v[e]), dv = hsel ? hi_bf(d[e]) : lo_bf(d[e]);
                const float sg = 1.f / (1.f + __expf(-1.702f * x));
                r[hsel] = dv * (sg + 1.702f * x * sg * (1.f - sg));
            }
            o[e] = pack2bf(r[0], r[1]);
        }
        *(u32x4_t*)(dz + i * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// Column sums of a bf16 matrix into fp32 (bias gradients): out[n] += sum_t dY[t][n].
// Block = 64 columns x 4 row-lanes... each thread owns 8 columns (16 B) and strides over rows.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* dy, long long ld, float* ws, int T, int N) {
    __shared__ float red[8][32 * 8 + 1];
    const int cchunk = threadIdx.x & 31, rlane = threadIdx.x >> 5;  // 32 chunks (256 cols) x 8 row lanes
    const int col = blockIdx.x * 256 + cchunk * 8;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (col < N) {
        for (int t = blockIdx.y * 8 + rlane; t < T; t += gridDim.y * 8) {
            const u32x4_t v = *(const u32x4_t*)(dy + (long long)t * ld + col);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += lo_bf(v[e]); acc[2 * e + 1] += hi_bf(v[e]); }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) red[rlane][cchunk * 8 + e] = acc[e];
    __syncthreads();
    const int c = threadIdx.x;
    if (blockIdx.x * 256 + c < N) {
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 8; ++r) s += red[r][c];
        ws[(long long)blockIdx.y * N + blockIdx.x * 256 + c] = s;       // this row group's partial; summed in group order by colsum_reduce_kernel (no atomics)
    }
}
__global__ __launch_bounds__(256) void colsum_reduce_kernel(const float* ws, int nparts, float* out, int N) {
    // 32 columns x 8 partial groups per block, eight loads in flight per thread, group sums added in group order (norm.hip partial_reduce_acc_kernel)
    __shared__ float red[8][33];
    const int c = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + c;
    float s = 0.f;
    if (i < N) {
        int q = g;
        for (; q + 56 < nparts; q += 64) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ws[(long long)(q + 8 * u) * N + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; q < nparts; q += 8) s += ws[(long long)q * N + i];
    }
    red[g][c] = s;
    __syncthreads();
    if (g == 0 && i < N) {
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) t += red[u][c];
        out[i] += t;
    }
}

// ------------------------------------------------------------------------------------------------
// Token embedding gather with image-feature scatter (TF:modeling_qwen2_5_vl.py:1204-1215): row t of the
// output is img[img_index[t]] when img_index[t] >= 0 (an <|image_pad|> slot), else E[ids[t]].
// Backward: fp32 atomics into dE (text rows) / dimg (image rows; several sequences may share one image).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void embed_fwd_kernel(const long long* ids, const int* img_index, const bf16_t* E, const bf16_t* img,
                                                        bf16_t* out, int T, int H) {
    const int cpr = H >> 3;
    const long long total = (long long)T * cpr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long t = i / cpr;
        const int c = (int)(i % cpr);
        const int ii = img_index ? img_index[t] : -1;
        const bf16_t* src = ii >= 0 ? img + (long long)ii * H : E + ids[t] * (long long)H;
        *(u32x4_t*)(out + t * H + c * 8) = *(const u32x4_t*)(src + c * 8);
    }
}
// Backward of the gather (and of every "several rows read one row" map of the path): dst[rows[u]] += sum_{k in [ptr[u], ptr[u+1])} src[idx[k]], fp32, the terms of a
// row added in the order the CSR lists them.  One block per destination row: a single writer per row and a fixed order -- no atomics, bit-reproducible (round 5; the
// scatter this replaces added the token rows of one vocabulary entry with float atomics in arrival order).
__global__ __launch_bounds__(256) void rows_scatter_acc_kernel(const bf16_t* src, const long long* rows, const int* ptr, const int* idx, float* dst, int H) {
    const int u = blockIdx.x;
    const int k0 = ptr[u], k1 = ptr[u + 1];
    float* out = dst + rows[u] * (long long)H;
    for (int ch = threadIdx.x; ch < (H >> 3); ch += 256) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int k = k0; k < k1; ++k) {
            const u32x4_t a = *(const u32x4_t*)(src + (long long)idx[k] * H + ch * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += lo_bf(a[e]); v[2 * e + 1] += hi_bf(a[e]); }
        }
        f32x4_t o0 = *(const f32x4_t*)(out + ch * 8), o1 = *(const f32x4_t*)(out + ch * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { o0[e] += v[e]; o1[e] += v[4 + e]; }
        *(f32x4_t*)(out + ch * 8) = o0;
        *(f32x4_t*)(out + ch * 8 + 4) = o1;
    }
}

// GELU, tanh approximation (SigLIP MLP: ACT2FN["gelu_pytorch_tanh"], transformers/models/siglip/modeling_siglip.py:310-322):
// 0.5 x (1 + tanh(c (x + 0.044715 x^3))), c = sqrt(2/pi); fp32 math on bf16 values
__device__ __forceinline__ float gelu_tanh_f(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x);
    return 0.5f * x * (1.f + tanhf(u));
}
__device__ __forceinline__ float gelu_tanh_df(float x) {
    const float u = 0.7978845608028654f * (x + 0.044715f * x * x * x), th = tanhf(u);
    return 0.5f * (1.f + th) + 0.5f * x * (1.f - th * th) * 0.7978845608028654f * (1.f + 3.f * 0.044715f * x * x);
}
__global__ __launch_bounds__(256) void gelu_tanh_fwd_kernel(const bf16_t* z, bf16_t* a, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *(const u32x4_t*)(z + i * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(gelu_tanh_f(lo_bf(v[e])), gelu_tanh_f(hi_bf(v[e])));
        *(u32x4_t*)(a + i * 8) = o;
    }
}
__global__ __launch_bounds__(256) void gelu_tanh_bwd_kernel(const bf16_t* da, const bf16_t* z, bf16_t* dz, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = *(const u32x4_t*)(z + i * 8), d = *(const u32x4_t*)(da + i * 8);
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(lo_bf(d[e]) * gelu_tanh_df(lo_bf(v[e])), hi_bf(d[e]) * gelu_tanh_df(hi_bf(v[e])));
        *(u32x4_t*)(dz + i * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// out[t] = sum over k in [ptr[t], ptr[t+1]) of src[idx[k]]  (fp32 sum, rounded once; empty list -> zeros).  The scatter of the lm_head's
// input gradient back onto the token rows: a hidden row can feed several selected rows (the prompt's last token predicts the first token of
// every completion of its group), most feed none.  One block per token row, 16-byte accesses; deterministic (list order).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void rows_gather_sum_kernel(const bf16_t* src, const int* ptr, const int* idx, const float* wts, bf16_t* out, int T, int H) {
    const int t = blockIdx.x;
    const int k0 = ptr[t], k1 = ptr[t + 1];
    for (int ch = threadIdx.x; ch < (H >> 3); ch += 256) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
        for (int k = k0; k < k1; ++k) {
            const u32x4_t a = *(const u32x4_t*)(src + (long long)idx[k] * H + ch * 8);
            const float wk = wts ? wts[k] : 1.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) { v[2 * e] += wk * lo_bf(a[e]); v[2 * e + 1] += wk * hi_bf(a[e]); }
        }
        u32x4_t o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = pack2bf(v[2 * e], v[2 * e + 1]);
        *(u32x4_t*)(out + (long long)t * H + ch * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------------
// casts / strided copies
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* in, long long ldi, bf16_t* out, long long ldo, int R, int C, int Cpad) {
    // out[r][0:C] = bf16(in[r][0:C]); out[r][C:Cpad] = 0  (K padding for the patch-embed GEMM)
    const long long total = (long long)R * Cpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / Cpad;
        const int c = (int)(i % Cpad);
        out[r * ldo + c] = c < C ? f2bf(in[r * ldi + c]) : (bf16_t)0;
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_t* in, float* out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) out[i] = bf2f(in[i]);
}
// contiguous forms, 8 elements per thread and step: one 16-byte access on the bf16 side, two on the fp32 side (the gradient exchange casts
// GB-sized buckets: HBM-bound, 6 B per element)
__global__ __launch_bounds__(256) void cast_f32_bf16_vec_kernel(const float* in, bf16_t* out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const f32x4_t a = ((const f32x4_t*)in)[2 * i], b = ((const f32x4_t*)in)[2 * i + 1];
        u32x4_t o;
        o.x = pack2bf(a.x, a.y); o.y = pack2bf(a.z, a.w); o.z = pack2bf(b.x, b.y); o.w = pack2bf(b.z, b.w);
        ((u32x4_t*)out)[i] = o;
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_vec_kernel(const bf16_t* in, float* out, long long n8) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const u32x4_t v = ((const u32x4_t*)in)[i];
        const f32x4_t a = {lo_bf(v.x), hi_bf(v.x), lo_bf(v.y), hi_bf(v.y)}, b = {lo_bf(v.z), hi_bf(v.z), lo_bf(v.w), hi_bf(v.w)};
        ((f32x4_t*)out)[2 * i] = a;
        ((f32x4_t*)out)[2 * i + 1] = b;
    }
}
__global__ __launch_bounds__(256) void add_f32_to_bf16_kernel(const float* in, const bf16_t* bias, bf16_t* out, long long R, int C) {
    // out = bf16(in + bias)  ([R,C] fp32 split-K sums -> bf16 activation); `in` is re-zeroed
    const long long total = R * C;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float v = in[i];
        ((float*)in)[i] = 0.f;
        if (bias) v += bf2f(bias[i % C]);
        out[i] = f2bf(v);
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 transpose: out[c][r] = in[r][c], 64x64 tiles through LDS (padded), 16-byte global accesses.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* in, long long ldi, bf16_t* out, long long ldo, int R, int C) {
    __shared__ bf16_t tile[64][64 + 2];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int t = threadIdx.x;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + t, r = idx >> 3, ch = idx & 7;
        const int gr = r0 + r, gc = c0 + ch * 8;
        bf16_t v[8];
        if (gr < R && gc + 8 <= C) {
            *(u32x4_t*)v = *(const u32x4_t*)(in + (long long)gr * ldi + gc);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (gr < R && gc + e < C) ? in[(long long)gr * ldi + gc + e] : (bf16_t)0;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) tile[r][ch * 8 + e] = v[e];
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int idx = it * 256 + t, c = idx >> 3, ch = idx & 7;
        const int gc = c0 + c, gr = r0 + ch * 8;
        if (gc >= C) continue;
        bf16_t v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = tile[ch * 8 + e][c];
        bf16_t* dst = out + (long long)gc * ldo + gr;
        if (gr + 8 <= R && (ldo & 7) == 0) {
            *(u32x4_t*)dst = *(const u32x4_t*)v;
        } else {
            for (int e = 0; e < 8 && gr + e < R; ++e) dst[e] = v[e];
        }
    }
}

inline int grid_for(long long work, int block = 256, int cap = 256 * 8) {
    long long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

}  // namespace

extern "C" int iadr1_rope_inplace(void* x, long long ld, const float* cos_t, const float* sin_t, int T, int nheads, int D, int backward,
                                  hipStream_t stream) {
    IADR1_REQUIRE(T > 0 && nheads > 0 && (D % 8) == 0, "rope: D=%d must be a multiple of 8", D);
    IADR1_REQUIRE((ld % 4) == 0, "rope: ld must be a multiple of 4");
    const long long total = (long long)T * nheads * (D / 8);
    hipLaunchKernelGGL(rope_kernel, dim3(grid_for(total)), dim3(256), 0, stream, (bf16_t*)x, ld, cos_t, sin_t, T, nheads, D, backward ? -1.f : 1.f);
    return iadr1_check_launch("rope_inplace");
}

extern "C" int iadr1_swiglu_fwd(const void* gu, long long ldg, void* a, long long lda, int T, int I, hipStream_t stream) {
    IADR1_REQUIRE(T > 0 && I > 0 && (I % 8) == 0 && (ldg % 8) == 0 && (lda % 8) == 0, "swiglu_fwd: I, ld must be multiples of 8");
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3(grid_for((long long)T * (I / 8))), dim3(256), 0, stream, (const bf16_t*)gu, ldg, (bf16_t*)a, lda, T, I);
    return iadr1_check_launch("swiglu_fwd");
}
extern "C" int iadr1_swiglu_bwd(const void* da, long long lda, const void* gu, long long ldg, void* dgu, long long ldd, int T, int I,
                                hipStream_t stream) {
    IADR1_REQUIRE(T > 0 && I > 0 && (I % 8) == 0 && (ldg % 8) == 0 && (lda % 8) == 0 && (ldd % 8) == 0, "swiglu_bwd: I, ld must be multiples of 8");
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3(grid_for((long long)T * (I / 8))), dim3(256), 0, stream, (const bf16_t*)da, lda, (const bf16_t*)gu, ldg, (bf16_t*)dgu, ldd, T, I);
    return iadr1_check_launch("swiglu_bwd");
}
extern "C" int iadr1_gelu_fwd(const void* z, void* a, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && (n % 8) == 0, "gelu_fwd: n must be a multiple of 8");
    hipLaunchKernelGGL(gelu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)z, (bf16_t*)a, n / 8);
    return iadr1_check_launch("gelu_fwd");
}
extern "C" int iadr1_gelu_bwd(const void* da, const void* z, void* dz, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && (n % 8) == 0, "gelu_bwd: n must be a multiple of 8");
    hipLaunchKernelGGL(gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)da, (const bf16_t*)z, (bf16_t*)dz, n / 8);
    return iadr1_check_launch("gelu_bwd");
}
extern "C" int iadr1_quick_gelu_fwd(const void* z, void* a, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && (n % 8) == 0, "quick_gelu_fwd: n must be a multiple of 8");
    hipLaunchKernelGGL(quick_gelu_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)z, (bf16_t*)a, n / 8);
    return iadr1_check_launch("quick_gelu_fwd");
}
extern "C" int iadr1_quick_gelu_bwd(const void* da, const void* z, void* dz, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && (n % 8) == 0, "quick_gelu_bwd: n must be a multiple of 8");
    hipLaunchKernelGGL(quick_gelu_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)da, (const bf16_t*)z, (bf16_t*)dz, n / 8);
    return iadr1_check_launch("quick_gelu_bwd");
}
static int colsum_groups(int T) {
    const int gy = (T + 63) / 64;
    return gy > 64 ? 64 : gy;
}
extern "C" long long iadr1_colsum_workspace_bytes(int T, int N) { return (T <= 0 || N <= 0) ? 0 : (long long)colsum_groups(T) * N * 4; }
extern "C" int iadr1_colsum_acc(const void* dy, long long ld, float* out, float* workspace, int T, int N, hipStream_t stream) {
    IADR1_REQUIRE(T > 0 && N > 0 && (N % 8) == 0 && (ld % 8) == 0, "colsum: N, ld must be multiples of 8");
    IADR1_REQUIRE(workspace != nullptr, "colsum: the partial-sum workspace is required (iadr1_colsum_workspace_bytes)");
    const int gy = colsum_groups(T);
    hipLaunchKernelGGL(colsum_kernel, dim3((N + 255) / 256, gy), dim3(256), 0, stream, (const bf16_t*)dy, ld, workspace, T, N);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((N + 31) / 32), dim3(256), 0, stream, (const float*)workspace, gy, out, N);
    return iadr1_check_launch("colsum_acc");
}
extern "C" int iadr1_embed_fwd(const long long* ids, const int* img_index, const void* E, const void* img, void* out, int T, int H,
                               hipStream_t stream) {
    IADR1_REQUIRE(T > 0 && (H % 8) == 0, "embed_fwd: H must be a multiple of 8");
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(grid_for((long long)T * (H / 8))), dim3(256), 0, stream, ids, img_index, (const bf16_t*)E, (const bf16_t*)img, (bf16_t*)out, T, H);
    return iadr1_check_launch("embed_fwd");
}
extern "C" int iadr1_rows_scatter_acc(const void* src, const long long* rows, const int* ptr, const int* idx, float* dst, int U, int H, hipStream_t stream) {
    IADR1_REQUIRE(U > 0 && H > 0 && (H % 8) == 0 && ((((uintptr_t)dst) & 15) == 0), "rows_scatter_acc: H must be a multiple of 8, dst 16-byte aligned");
    hipLaunchKernelGGL(rows_scatter_acc_kernel, dim3(U), dim3(256), 0, stream, (const bf16_t*)src, rows, ptr, idx, dst, H);
    return iadr1_check_launch("rows_scatter_acc");
}
extern "C" int iadr1_gelu_tanh_fwd(const void* z, void* a, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && (n % 8) == 0, "gelu_tanh_fwd: n must be a multiple of 8");
    hipLaunchKernelGGL(gelu_tanh_fwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)z, (bf16_t*)a, n / 8);
    return iadr1_check_launch("gelu_tanh_fwd");
}
extern "C" int iadr1_gelu_tanh_bwd(const void* da, const void* z, void* dz, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0 && (n % 8) == 0, "gelu_tanh_bwd: n must be a multiple of 8");
    hipLaunchKernelGGL(gelu_tanh_bwd_kernel, dim3(grid_for(n / 8)), dim3(256), 0, stream, (const bf16_t*)da, (const bf16_t*)z, (bf16_t*)dz, n / 8);
    return iadr1_check_launch("gelu_tanh_bwd");
}
extern "C" int iadr1_rows_gather_sum(const void* src, const int* ptr, const int* idx, const float* weights, void* out, int T, int H, hipStream_t stream) {
    IADR1_REQUIRE(T > 0 && H > 0 && (H % 8) == 0, "rows_gather_sum: H must be a multiple of 8");
    hipLaunchKernelGGL(rows_gather_sum_kernel, dim3(T), dim3(256), 0, stream, (const bf16_t*)src, ptr, idx, weights, (bf16_t*)out, T, H);
    return iadr1_check_launch("rows_gather_sum");
}
extern "C" int iadr1_cast_f32_to_bf16(const float* in, long long ldi, void* out, long long ldo, int R, int C, int Cpad, hipStream_t stream) {
    IADR1_REQUIRE(R > 0 && C > 0 && Cpad >= C, "cast_f32_to_bf16: bad shape");
    const long long tot = (long long)R * Cpad;
    if (C == Cpad && (R == 1 || (ldi == Cpad && ldo == Cpad)) && (((uintptr_t)in | (uintptr_t)out) & 15) == 0 && tot >= 8) {   // contiguous: vector form + scalar tail
        const long long n8 = tot / 8;
        hipLaunchKernelGGL(cast_f32_bf16_vec_kernel, dim3(grid_for(n8)), dim3(256), 0, stream, in, (bf16_t*)out, n8);
        if (tot % 8) hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(1), dim3(64), 0, stream, in + n8 * 8, (long long)(tot % 8), (bf16_t*)out + n8 * 8, (long long)(tot % 8), 1, (int)(tot % 8), (int)(tot % 8));
        return iadr1_check_launch("cast_f32_to_bf16");
    }
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(grid_for((long long)R * Cpad)), dim3(256), 0, stream, in, ldi, (bf16_t*)out, ldo, R, C, Cpad);
    return iadr1_check_launch("cast_f32_to_bf16");
}
extern "C" int iadr1_cast_bf16_to_f32(const void* in, float* out, long long n, hipStream_t stream) {
    IADR1_REQUIRE(n > 0, "cast_bf16_to_f32: empty");
    if ((((uintptr_t)in | (uintptr_t)out) & 15) == 0 && n >= 8) {
        const long long n8 = n / 8;
        hipLaunchKernelGGL(cast_bf16_f32_vec_kernel, dim3(grid_for(n8)), dim3(256), 0, stream, (const bf16_t*)in, out, n8);
        if (n % 8) hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(1), dim3(64), 0, stream, (const bf16_t*)in + n8 * 8, out + n8 * 8, n % 8);
        return iadr1_check_launch("cast_bf16_to_f32");
    }
    hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(grid_for(n)), dim3(256), 0, stream, (const bf16_t*)in, out, n);
    return iadr1_check_launch("cast_bf16_to_f32");
}
extern "C" int iadr1_f32_bias_to_bf16(float* in, const void* bias, void* out, long long R, int C, hipStream_t stream) {
    IADR1_REQUIRE(R > 0 && C > 0, "f32_bias_to_bf16: empty");
    hipLaunchKernelGGL(add_f32_to_bf16_kernel, dim3(grid_for(R * C)), dim3(256), 0, stream, in, (const bf16_t*)bias, (bf16_t*)out, R, C);
    return iadr1_check_launch("f32_bias_to_bf16");
}
extern "C" int iadr1_transpose_bf16(const void* in, long long ldi, void* out, long long ldo, int R, int C, hipStream_t stream) {
    IADR1_REQUIRE(R > 0 && C > 0 && (ldi % 8) == 0, "transpose: ldi must be a multiple of 8");
    hipLaunchKernelGGL(transpose_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, stream, (const bf16_t*)in, ldi, (bf16_t*)out, ldo, R, C);
    return iadr1_check_launch("transpose_bf16");
}

// ------------------------------------------------------------------------------------------------
// Rollout bookkeeping kept ON DEVICE so that one decode step is a fixed launch sequence (hipGraph replay):
// rope_table: cos/sin rows for the current text positions (all three M-RoPE components equal for text,
//   TF:modeling_qwen2_5_vl.py:1165-1176: pos = kv_len + rope_delta).
// decode_advance: append the sampled token, EOS/pad handling as the reference's padded completions
//   (REF:train/stage_rl/trainer/sc_grpo_trainer.py:680-683,722-726), bump positions / cache slots / step.
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void rope_table_kernel(const int* pos, const float* inv_freq, float* cs, float* sn, int B, int half) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B * half) return;
    const int b = i / half, j = i - b * half;
    const float a = (float)pos[b] * inv_freq[j];
    cs[i] = cosf(a);
    sn[i] = sinf(a);
}

__global__ __launch_bounds__(256) void decode_advance_kernel(const long long* sampled, long long* cur_tok, long long* out_tokens, int C,
                                                             int* pos, int* ctx_len, long long* slot, const int* block_table,
                                                             int max_pages, int* finished, unsigned* step, int eos, int pad, int B,
                                                             int* all_done, const float* inv_freq, float* cs, float* sn, int half) {
    const int b = threadIdx.x;
    const unsigned st = *step;
    int fin = 1;
    if (b < B) {
        long long tok = sampled[b];
        fin = finished[b];
        if (fin) tok = pad;
        else if (eos >= 0 && tok == eos) finished[b] = fin = 1;
        if ((int)st < C) out_tokens[(long long)b * C + st] = tok;
        cur_tok[b] = tok;
        pos[b] += 1;
        const int n = ctx_len[b];  // keys already in the cache (includes the token just processed)
        ctx_len[b] = n + 1;
        slot[b] = (long long)block_table[(long long)b * max_pages + n / 32] * 32 + (n & 31);
    }
    const int every = __syncthreads_and(fin);          // also orders the pos[] writes before the table below
    if (inv_freq != nullptr) {     // the rotary table of the NEXT decode step's positions (what a separate rope_table launch computed at the start of that step)
        for (int i = threadIdx.x; i < B * half; i += 256) {
            const int r = i / half, j = i - r * half;
            const float a = (float)pos[r] * inv_freq[j];
            cs[i] = cosf(a);
            sn[i] = sinf(a);
        }
    }
    if (threadIdx.x == 0) {
        if (all_done != nullptr) *all_done = every;
        __threadfence();
        *step = st + 1;
    }
}
}  // namespace

// wait_counter: a stream-ordered wait on a DEVICE counter another stream's kernels bump (the decode step counter above).  One wave polls the counter
// with agent-scope loads (sleeping ~4 us between polls) and returns once it has reached `target` -- or after `timeout_ms`, so that a counter that never
// arrives cannot hang the queue.  Used instead of hipEventRecord / hipStreamWaitEvent pairs between the decode replays and the shadow pass: both release a
// chunk on time, but the event records between the graph launches cost the decode stream 0.06 ms per step (3.22 vs 3.16 ms, profiles/EXPERIMENTS.md round 5).
namespace {
__global__ __launch_bounds__(64) void wait_counter_kernel(const unsigned* counter, unsigned target, unsigned long long timeout_ticks, int* timed_out) {
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(127);
        if (wall_clock64() - t0 > timeout_ticks) {
            if (timed_out) *timed_out = 1;
            break;
        }
    }
}
}  // namespace

extern "C" int iadr1_wait_counter(const unsigned* counter, unsigned target, int timeout_ms, int* timed_out, hipStream_t stream) {
    IADR1_REQUIRE(counter != nullptr && timeout_ms > 0 && timeout_ms <= 60000, "wait_counter: a device counter and a timeout in (0, 60000] ms are required");
    hipLaunchKernelGGL(wait_counter_kernel, dim3(1), dim3(64), 0, stream, counter, target, (unsigned long long)timeout_ms * 100000ull, timed_out);     // wall_clock64: 100 MHz
    return iadr1_check_launch("wait_counter");
}

extern "C" int iadr1_rope_table(const int* pos, const float* inv_freq, float* cos_t, float* sin_t, int B, int half, hipStream_t stream) {
    IADR1_REQUIRE(B > 0 && half > 0, "rope_table: empty");
    hipLaunchKernelGGL(rope_table_kernel, dim3((B * half + 255) / 256), dim3(256), 0, stream, pos, inv_freq, cos_t, sin_t, B, half);
    return iadr1_check_launch("rope_table");
}
extern "C" int iadr1_decode_advance(const long long* sampled, long long* cur_tok, long long* out_tokens, int C, int* pos, int* ctx_len,
                                    long long* slot, const int* block_table, int max_pages, int* finished, unsigned* step, int eos,
                                    int pad, int B, int* all_done, const float* inv_freq, float* cos_t, float* sin_t, int half, hipStream_t stream) {
    IADR1_REQUIRE(B > 0 && B <= 256, "decode_advance: B must be in [1,256] (single block so the step bump is ordered)");
    IADR1_REQUIRE(inv_freq == nullptr || (cos_t != nullptr && sin_t != nullptr && half > 0), "decode_advance: inv_freq needs cos_t / sin_t / half");
    hipLaunchKernelGGL(decode_advance_kernel, dim3(1), dim3(256), 0, stream, sampled, cur_tok, out_tokens, C, pos, ctx_len, slot, block_table, max_pages, finished, step, eos, pad, B,
                       all_done, inv_freq, cos_t, sin_t, half);
    return iadr1_check_launch("decode_advance");
}
