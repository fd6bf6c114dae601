// Fused sampler for the group rollout (gfx950): temperature -> top-k -> top-p -> multinomial, one block
// per sequence row over fp32 logits [B, V].  Same filter semantics as the sampler the reference drives
// (vLLM SamplingParams(temperature, top_p=0.9, top_k=50), /root/reference/train/stage_rl/trainer/
// sc_grpo_trainer.py:353-358): top-p is evaluated on the top-k-renormalised distribution and always keeps the
// most likely token.  Randomness is a counter-based Philox4x32-10 stream keyed by (seed; row, step), so
// token ids are reproducible run-to-run and checkable on the CPU.  temperature == 0 selects greedy argmax
// (lowest index wins ties), the mode used for bit-exact parity against the oracle.
#include "common.h"

namespace {

constexpr int NT = 256, MAXK = 64;

__device__ __forceinline__ uint32_t okey(float x) {
    const uint32_t u = __float_as_uint(x);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // order-preserving map float -> uint
}

__device__ __forceinline__ float philox_uniform(uint32_t k0, uint32_t k1, uint32_t c0, uint32_t c1) {
    uint32_t c[4] = {c0, c1, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return (float)(c[0] >> 8) * (1.0f / 16777216.0f);
}

struct SampleArgs {
    const float* logits;
    long long ld;
    long long* out;     // [B] sampled token ids
    int B, V;
    float temperature, top_p;
    int top_k;
    int suppress;       // token id forced to -inf (e.g. EOS for fixed-length benchmarking), -1 = none
    uint32_t seed_lo, seed_hi, step;
    const unsigned* step_ptr;  // optional device-resident step counter (hipGraph replays freeze kernel arguments)
};

__device__ __forceinline__ float fetch(const SampleArgs& p, const float* x, int i) { return i == p.suppress ? -INFINITY : x[i]; }

// exclusive block scan of one int per thread (NT=256); returns offset, total in *total
__device__ __forceinline__ int block_exscan(int v, int* sh, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(inc, o, WAVE);
        if (lane >= o) inc += t;
    }
    __syncthreads();
    if (lane == 63) sh[wv] = inc;
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wv; ++i) base += sh[i];
    *total = sh[0] + sh[1] + sh[2] + sh[3];
    return base + inc - v;
}

__global__ __launch_bounds__(NT) void sample_kernel(SampleArgs p) {
    __shared__ int hist[2048];
    __shared__ int sh_i[8];
    __shared__ float sh_f[16];
    __shared__ float cval[MAXK];
    __shared__ int cidx[MAXK];
    __shared__ float sval[MAXK];
    __shared__ int sidx[MAXK];
    const int row = blockIdx.x, t = threadIdx.x;
    const float* x = p.logits + (long long)row * p.ld;
    const int V = p.V;

    if (p.temperature <= 0.f) {  // greedy
        float bv = -INFINITY;
        int bi = 0x7fffffff;
        for (int i = t; i < V; i += NT) {
            const float v = fetch(p, x, i);
            if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(bv, o, WAVE);
            const int oi = __shfl_xor(bi, o, WAVE);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((t & 63) == 0) { sh_f[t >> 6] = bv; sh_i[t >> 6] = bi; }
        __syncthreads();
        if (t == 0) {
            for (int w = 1; w < 4; ++w)
                if (sh_f[w] > bv || (sh_f[w] == bv && sh_i[w] < bi)) { bv = sh_f[w]; bi = sh_i[w]; }
            p.out[row] = bi;
        }
        return;
    }

    const int K = min(min(p.top_k > 0 ? p.top_k : MAXK, MAXK), V);
    // ---- radix select of the K-th largest key: 11 + 11 + 10 bits ------------------------------------
    uint32_t prefix = 0, pmask = 0;
    int need = K;  // how many still to take from the current candidate set
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
    for (int pass = 0; pass < 3; ++pass) {
        const int nb = 1 << widths[pass];
        for (int i = t; i < nb; i += NT) hist[i] = 0;
        __syncthreads();
        for (int i = t; i < V; i += NT) {
            const uint32_t k = okey(fetch(p, x, i));
            if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (nb - 1)], 1);
        }
        __syncthreads();
        if (t == 0) {
            int acc = 0, b = nb - 1;
            for (; b > 0; --b) {
                if (acc + hist[b] >= need) break;
                acc += hist[b];
            }
            sh_i[0] = b;
            sh_i[1] = need - acc;
        }
        __syncthreads();
        prefix |= ((uint32_t)sh_i[0]) << shifts[pass];
        pmask |= ((uint32_t)(nb - 1)) << shifts[pass];
        need = sh_i[1];
        __syncthreads();
    }
    const uint32_t tau = prefix;  // K-th largest key; `need` of the elements equal to tau are taken (lowest index first)

    // ---- ordered compaction (index order => deterministic) ------------------------------------------------
    const int per = (V + NT - 1) / NT, lo = t * per, hi = min(V, lo + per);
    int ngt = 0, neq = 0;
    for (int i = lo; i < hi; ++i) {
        const uint32_t k = okey(fetch(p, x, i));
        ngt += k > tau;
        neq += k == tau;
    }
    int tot_gt, tot_eq;
    int off_gt = block_exscan(ngt, sh_i, &tot_gt);
    int off_eq = block_exscan(neq, sh_i, &tot_eq);
    for (int i = lo; i < hi; ++i) {
        const float v = fetch(p, x, i);
        const uint32_t k = okey(v);
        if (k > tau) { cval[off_gt] = v; cidx[off_gt] = i; ++off_gt; }
        else if (k == tau) {
            if (off_eq < need) { cval[tot_gt + off_eq] = v; cidx[tot_gt + off_eq] = i; }
            ++off_eq;
        }
    }
    __syncthreads();
    const int n = tot_gt + min(need, tot_eq);  // == K unless the row holds fewer finite entries
    // ---- rank sort by (value desc, index asc) ------------------------------------------------------------
    if (t < n) {
        const float v = cval[t];
        const int id = cidx[t];
        int rank = 0;
        for (int j = 0; j < n; ++j) rank += (cval[j] > v) || (cval[j] == v && cidx[j] < id);
        sval[rank] = v;
        sidx[rank] = id;
    }
    __syncthreads();
    if (t == 0) {
        const float invT = 1.f / p.temperature;
        const float mx = sval[0] * invT;
        float tot = 0.f;
        for (int j = 0; j < n; ++j) { const float e = __expf(sval[j] * invT - mx); cval[j] = e; tot += e; }
        // top-p on the top-k-renormalised distribution: keep j while the mass ranked before it is < top_p
        float before = 0.f, kept = 0.f;
        int nk = 0;
        for (int j = 0; j < n; ++j) {
            if (j > 0 && before >= p.top_p * tot) break;
            kept += cval[j];
            before += cval[j];
            ++nk;
        }
        const uint32_t step = p.step_ptr ? *p.step_ptr : p.step;
        const float u = philox_uniform(p.seed_lo, p.seed_hi, (uint32_t)row, step) * kept;
        float cum = 0.f;
        int pick = nk - 1;
        for (int j = 0; j < nk; ++j) {
            cum += cval[j];
            if (cum > u) { pick = j; break; }
        }
        p.out[row] = sidx[pick];
    }
}

}  // namespace

extern "C" int iadr1_sample_topk_topp(const float* logits, long long ld, long long* out, int B, int V, float temperature, int top_k,
                                      float top_p, int suppress_token, unsigned long long seed, unsigned step, const unsigned* step_ptr,
                                      hipStream_t stream) {
    IADR1_REQUIRE(B > 0 && V > 0, "sample: empty");
    IADR1_REQUIRE(top_k <= MAXK, "sample: top_k=%d exceeds the built maximum %d", top_k, MAXK);
    IADR1_REQUIRE(top_p > 0.f && top_p <= 1.f, "sample: top_p must be in (0,1]");
    SampleArgs p{logits, ld, out, B, V, temperature, top_p, top_k, suppress_token, (uint32_t)seed, (uint32_t)(seed >> 32), step, step_ptr};
    hipLaunchKernelGGL(sample_kernel, dim3(B), dim3(NT), 0, stream, p);
    return iadr1_check_launch("sample_topk_topp");
}
