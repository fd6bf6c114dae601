// Log-prob / cross-entropy / GRPO-loss kernels (gfx950).
//
// logprob_rows: per row of an fp32 logit chunk [R,V] (written by gemm_nt with out_mode=1, never by
// the reference's bf16 [B,S,V] tensor): lse = logsumexp(row), logp = row[target] - lse
//   == `logits_row.log_softmax(-1)` + gather, /root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:510-513
//   (no temperature division), and == -CE for PA-SFT (TF:loss/loss_utils.py:32-71).
// dlogits_rows: dlogit = g[row] * (onehot(target) - softmax(row)) in bf16, in a form ready for the two
// backward GEMMs (dH = dlogits . W, dW = dlogits^T . H).
// grpo_loss: sc_grpo_trainer.py:746,796-798,816 for one rank's [N,C] completion block.
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void logprob_rows_kernel(const float* logits, long long ld, const long long* targets, float* logp,
                                                           float* lse_out, int R, int V) {
    __shared__ float scratch[16];
    const int row = blockIdx.x;
    const float* x = logits + (long long)row * ld;
    float m = -INFINITY, s = 0.f;
    const int n4 = V >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
        const f32x4_t v = *(const f32x4_t*)(x + i * 4);
        const float mx = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (mx > m) { s *= __expf(m - mx); m = mx; }
        s += __expf(v[0] - m) + __expf(v[1] - m) + __expf(v[2] - m) + __expf(v[3] - m);
    }
    for (int i = n4 * 4 + threadIdx.x; i < V; i += 256) {
        const float v = x[i];
        if (v > m) { s *= __expf(m - v); m = v; }
        s += __expf(v - m);
    }
    const float gm = block_max<256>(m, scratch);
    s = (m == -INFINITY) ? 0.f : s * __expf(m - gm);
    const float gs = block_sum<256>(s, scratch);
    if (threadIdx.x == 0) {
        const float lse = gm + logf(gs);
        if (lse_out) lse_out[row] = lse;
        const long long tgt = targets[row];
        logp[row] = (tgt >= 0 && tgt < V) ? x[tgt] - lse : 0.f;  // ignored rows (target < 0, e.g. -100) contribute 0
    }
}

__global__ __launch_bounds__(256) void dlogits_rows_kernel(const float* logits, long long ld, const long long* targets, const float* lse,
                                                           const float* g, bf16_t* dl, long long ldd, int R, int V) {
    const int row = blockIdx.y;
    const float* x = logits + (long long)row * ld;
    const float l = lse[row], gr = g[row];
    const long long tgt = targets[row];
    const int n8 = V >> 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n8; i += gridDim.x * 256) {
        const f32x4_t a = *(const f32x4_t*)(x + i * 8), b = *(const f32x4_t*)(x + i * 8 + 4);
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = -gr * __expf(a[e] - l); o[4 + e] = -gr * __expf(b[e] - l); }
        if (tgt >= (long long)i * 8 && tgt < (long long)i * 8 + 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) if (tgt == (long long)i * 8 + e) o[e] += gr;
        }
        *(u32x4_t*)(dl + (long long)row * ldd + i * 8) = (u32x4_t){pack2bf(o[0], o[1]), pack2bf(o[2], o[3]), pack2bf(o[4], o[5]), pack2bf(o[6], o[7])};
    }
}

// One block per sequence row.  Outputs: dlogp [N,C] (gradient of the batch loss w.r.t. the policy
// log-probs), per-token kl [N,C], and per-row partials row_loss[N], row_kl[N] (means over the mask).
__global__ __launch_bounds__(256) void grpo_loss_kernel(const float* logp, const float* ref_logp, const float* adv, const int* mask,
                                                        float beta, float inv_nrows, float* dlogp, float* kl_out, float* row_loss,
                                                        float* row_kl, int N, int C) {
    __shared__ float scratch[16];
    const int n = blockIdx.x;
    const float A = adv[n];
    float cnt = 0.f, sl = 0.f, sk = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const int i = n * C + c;
        const float d = ref_logp[i] - logp[i];
        const float e = __expf(d);
        const float kl = e - d - 1.f;
        const float m = (float)mask[i];
        if (kl_out) kl_out[i] = kl;
        cnt += m;
        sl += m * (-(A - beta * kl));  // exp(p - p.detach()) == 1 in value
        sk += m * kl;
    }
    cnt = block_sum<256>(cnt, scratch);
    sl = block_sum<256>(sl, scratch);
    sk = block_sum<256>(sk, scratch);
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    if (threadIdx.x == 0) {
        row_loss[n] = sl * inv;
        row_kl[n] = sk * inv;
    }
    for (int c = threadIdx.x; c < C; c += 256) {
        const int i = n * C + c;
        const float d = ref_logp[i] - logp[i];
        // d/dp [ -(exp(p - sg p) A - beta (exp(r-p) - (r-p) - 1)) ] = -A + beta (1 - exp(r-p))
        dlogp[i] = (float)mask[i] * inv * inv_nrows * (-A + beta * (1.f - __expf(d)));
    }
}

}  // namespace

extern "C" int iadr1_logprob_rows(const float* logits, long long ld, const long long* targets, float* logp, float* lse, int R, int V,
                                  hipStream_t stream) {
    IADR1_REQUIRE(R > 0 && V > 0 && (ld % 4) == 0, "logprob_rows: ld must be a multiple of 4");
    hipLaunchKernelGGL(logprob_rows_kernel, dim3(R), dim3(256), 0, stream, logits, ld, targets, logp, lse, R, V);
    return iadr1_check_launch("logprob_rows");
}
extern "C" int iadr1_dlogits_rows(const float* logits, long long ld, const long long* targets, const float* lse, const float* g, void* dl,
                                  long long ldd, int R, int V, hipStream_t stream) {
    IADR1_REQUIRE(R > 0 && V > 0 && (V % 8) == 0 && (ld % 4) == 0 && (ldd % 8) == 0, "dlogits_rows: V, ldd multiples of 8, ld multiple of 4");
    int gx = (V / 8 + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(dlogits_rows_kernel, dim3(gx, R), dim3(256), 0, stream, logits, ld, targets, lse, g, (bf16_t*)dl, ldd, R, V);
    return iadr1_check_launch("dlogits_rows");
}
extern "C" int iadr1_grpo_loss(const float* logp, const float* ref_logp, const float* adv, const int* mask, float beta, int n_total_rows,
                               float* dlogp, float* kl, float* row_loss, float* row_kl, int N, int C, hipStream_t stream) {
    IADR1_REQUIRE(N > 0 && C > 0 && n_total_rows >= N, "grpo_loss: need 0 < N <= n_total_rows");
    hipLaunchKernelGGL(grpo_loss_kernel, dim3(N), dim3(256), 0, stream, logp, ref_logp, adv, mask, beta, 1.f / (float)n_total_rows, dlogp, kl, row_loss, row_kl, N, C);
    return iadr1_check_launch("grpo_loss");
}
