// Fused sampler for the group rollout (gfx950): temperature -> top-k -> top-p -> multinomial over fp32
// logits [B, V].  Same filter semantics as the sampler the reference drives
// (vLLM SamplingParams(temperature, top_p=0.9, top_k=50), /root/reference/train/stage_rl/trainer/
// sc_grpo_trainer.py:353-358): top-p is evaluated on the top-k-renormalised distribution and always keeps the
// most likely token.  Randomness is a counter-based Philox4x32-10 stream keyed by (seed; row, step), so
// token ids are reproducible run-to-run and checkable on the CPU.  temperature == 0 selects greedy argmax
// (lowest index wins ties), the mode used for bit-exact parity against the oracle.
//
// HBM/L2-bound selection problem, two launches so that a 64-row batch still fills the chip:
//   stage 1: grid (NCHUNK, B): each block reads its slice of the row ONCE into registers and extracts its local
//            top-K with an exact radix select (11+11+10 bits, LDS histograms) + index-ordered compaction;
//   stage 2: grid (B): the NCHUNK*K survivors -> exact global top-K (same routine) -> rank sort -> softmax,
//            top-p cut, inverse-CDF draw.
#include "common.h"

namespace {

constexpr int NT = 256, MAXK = 64, NCHUNK = 32, ITEMS = 20;  // V <= NCHUNK*NT*ITEMS = 163840

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

// Exact top-K of the block's register-resident items (thread t owns the CONTIGUOUS items [t*NI, (t+1)*NI), so
// that index order == (thread, item) order).  Survivors are written in index order to oval/oidx[0..n).
// Ties at the K-th value are resolved towards the lowest index.  Returns n (<= K).
template <int NI>
__device__ __forceinline__ int block_topk(const float (&val)[NI], const int (&idx)[NI], int K, int* hist, int* sh_i, float* oval, int* oidx) {
    const int t = threadIdx.x;
    uint32_t prefix = 0, pmask = 0;
    int need = K;
    const int shifts[3] = {21, 10, 0}, widths[3] = {11, 11, 10};
#pragma unroll
    for (int pass = 0; pass < 3; ++pass) {
        const int nb = 1 << widths[pass];
        for (int i = t; i < nb; i += NT) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            if (idx[i] >= 0) {
                const uint32_t k = okey(val[i]);
                if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shifts[pass]) & (nb - 1)], 1);
            }
        }
        __syncthreads();
        if (t < 64) {
            // one wave scans the histogram from the top: lane owns nb/64 consecutive bins (descending order)
            const int per = nb / 64;
            const int hi_bin = nb - 1 - t * per;
            int s = 0;
            for (int b = 0; b < per; ++b) s += hist[hi_bin - b];
            int inc = s;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int u = __shfl_up(inc, o, WAVE);
                if (t >= o) inc += u;
            }
            const int before = inc - s;  // elements in bins above this lane's range
            const unsigned long long hit = __ballot(inc >= need);
            const int first = hit ? (int)__builtin_ctzll(hit) : 63;
            if (t == first) {
                int acc = before, b = hi_bin;
                for (int j = 0; j < per; ++j, --b) {
                    if (acc + hist[b] >= need || (b == 0)) break;
                    acc += hist[b];
                }
                if (b < 0) b = 0;
                sh_i[4] = b;
                sh_i[5] = need - acc;
            }
        }
        __syncthreads();
        prefix |= ((uint32_t)sh_i[4]) << shifts[pass];
        pmask |= ((uint32_t)(nb - 1)) << shifts[pass];
        need = sh_i[5];
        __syncthreads();
    }
    const uint32_t tau = prefix;
    int ngt = 0, neq = 0;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (idx[i] >= 0) {
            const uint32_t k = okey(val[i]);
            ngt += k > tau;
            neq += k == tau;
        }
    }
    int tot_gt, tot_eq;
    int off_gt = block_exscan(ngt, sh_i, &tot_gt);
    int off_eq = block_exscan(neq, sh_i, &tot_eq);
    const int take_eq = min(need, tot_eq);
    // merge ">" and the first `need` "==" in index order: an element's output slot is
    //   (# of '>' before it) + (# of taken '==' before it)
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        if (idx[i] >= 0) {
            const uint32_t k = okey(val[i]);
            if (k > tau) {
                const int slot = off_gt + min(off_eq, take_eq);
                oval[slot] = val[i]; oidx[slot] = idx[i];
                ++off_gt;
            } else if (k == tau) {
                if (off_eq < take_eq) {
                    const int slot = off_gt + off_eq;
                    oval[slot] = val[i]; oidx[slot] = idx[i];
                }
                ++off_eq;
            }
        }
    }
    __syncthreads();
    return tot_gt + take_eq;
}

struct SampleArgs {
    const float* logits;
    long long ld;
    long long* out;     // [B] sampled token ids
    float* cand_val;    // scratch [B][NCHUNK][MAXK]
    int* cand_idx;      // scratch [B][NCHUNK][MAXK]
    int B, V;
    float temperature, top_p;
    int top_k;
    int suppress;       // token id forced to -inf (e.g. EOS for fixed-length benchmarking), -1 = none
    uint32_t seed_lo, seed_hi, step;
    const unsigned* step_ptr;  // optional device-resident step counter (hipGraph replays freeze kernel arguments)
    const unsigned long long* seed_ptr;  // optional device-resident seed, same reason: a new seed per rollout must not force a re-capture
};

__device__ __forceinline__ int eff_k(const SampleArgs& p) {
    if (p.temperature <= 0.f) return 1;  // greedy == top-1 with lowest-index tie break
    return min(min(p.top_k > 0 ? p.top_k : MAXK, MAXK), p.V);
}

__global__ __launch_bounds__(NT) void sample_stage1(SampleArgs p) {
    __shared__ int hist[2048];
    __shared__ int sh_i[8];
    __shared__ float oval[MAXK];
    __shared__ int oidx[MAXK];
    const int row = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    const float* x = p.logits + (long long)row * p.ld;
    const int csize = (p.V + NCHUNK - 1) / NCHUNK;
    const int per = (csize + NT - 1) / NT;  // <= ITEMS
    const int c0 = chunk * csize, c1 = min(p.V, c0 + csize);
    float val[ITEMS];
    int idx[ITEMS];
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int g = c0 + t * per + i;
        const bool ok = i < per && g < c1;
        idx[i] = ok ? g : -1;
        val[i] = ok ? (g == p.suppress ? -INFINITY : x[g]) : -INFINITY;
    }
    const int K = eff_k(p);
    const int n = block_topk<ITEMS>(val, idx, K, hist, sh_i, oval, oidx);
    float* cv = p.cand_val + ((long long)row * NCHUNK + chunk) * MAXK;
    int* ci = p.cand_idx + ((long long)row * NCHUNK + chunk) * MAXK;
    if (t < MAXK) {
        cv[t] = t < n ? oval[t] : -INFINITY;
        ci[t] = t < n ? oidx[t] : -1;
    }
}

__global__ __launch_bounds__(NT) void sample_stage2(SampleArgs p) {
    __shared__ int hist[2048];
    __shared__ int sh_i[8];
    __shared__ float cval[MAXK];
    __shared__ int cidx[MAXK];
    __shared__ float sval[MAXK];
    __shared__ int sidx[MAXK];
    constexpr int NI = NCHUNK * MAXK / NT;  // 8 candidates per thread, contiguous => global index order is preserved
    const int row = blockIdx.x, t = threadIdx.x;
    const float* cv = p.cand_val + (long long)row * NCHUNK * MAXK;
    const int* ci = p.cand_idx + (long long)row * NCHUNK * MAXK;
    float val[NI];
    int idx[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        idx[i] = ci[t * NI + i];
        val[i] = cv[t * NI + i];
    }
    const int K = eff_k(p);
    const int n = block_topk<NI>(val, idx, K, hist, sh_i, cval, cidx);
    if (p.temperature <= 0.f) {
        if (t == 0) p.out[row] = n > 0 ? cidx[0] : 0;
        return;
    }
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
        const unsigned long long sd = p.seed_ptr ? *p.seed_ptr : (((unsigned long long)p.seed_hi << 32) | p.seed_lo);
        const float u = philox_uniform((uint32_t)sd, (uint32_t)(sd >> 32), (uint32_t)row, step) * kept;
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

extern "C" long long iadr1_sample_workspace_bytes(int B) { return (long long)B * NCHUNK * MAXK * 8; }

extern "C" int iadr1_sample_topk_topp(const float* logits, long long ld, long long* out, void* workspace, int B, int V, float temperature,
                                      int top_k, float top_p, int suppress_token, unsigned long long seed, unsigned step,
                                      const unsigned* step_ptr, const unsigned long long* seed_ptr, hipStream_t stream) {
    IADR1_REQUIRE(B > 0 && V > 0 && workspace != nullptr, "sample: empty problem or missing workspace (iadr1_sample_workspace_bytes)");
    IADR1_REQUIRE(V <= NCHUNK * NT * ITEMS, "sample: V=%d exceeds the built maximum %d", V, NCHUNK * NT * ITEMS);
    IADR1_REQUIRE(top_k <= MAXK, "sample: top_k=%d exceeds the built maximum %d", top_k, MAXK);
    IADR1_REQUIRE(top_p > 0.f && top_p <= 1.f, "sample: top_p must be in (0,1]");
    float* cv = (float*)workspace;
    int* ci = (int*)(cv + (long long)B * NCHUNK * MAXK);
    SampleArgs p{logits, ld, out, cv, ci, B, V, temperature, top_p, top_k, suppress_token, (uint32_t)seed, (uint32_t)(seed >> 32), step, step_ptr, seed_ptr};
    hipLaunchKernelGGL(sample_stage1, dim3(NCHUNK, B), dim3(NT), 0, stream, p);
    hipLaunchKernelGGL(sample_stage2, dim3(B), dim3(NT), 0, stream, p);
    return iadr1_check_launch("sample_topk_topp");
}
