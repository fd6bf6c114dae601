// Weight prefetcher of the group rollout (REF:/root/reference/train/stage_rl/trainer/sc_grpo_trainer.py:637-683 -- the decode loop the reference hands to vLLM).
//
// ONE persistent launch for the whole rollout, on a CU-masked stream of its own (iadr1_stream_create_cu_mask) whose hardware queue does not share a dispatch
// pipe with the decode queue (overlap.pick_concurrent_stream): nothing is forked or joined inside the decode graph, nothing is pending on the decode pipe.  The
// decode step publishes a PROGRESS MARK -- the first kernel of every decoder layer (the few-row RMSNorm, iadr1_side_out_t.mark) stores
// step * n_layers + layer into one device word -- and this kernel, paced by that word, pulls the weights of the layer `lead` marks ahead through L2 into the 256 MB
// memory-side cache (plain reads whose values are dropped), so that the weight-streaming launches of the decode step (gate|up, down: 135 of a 3B layer's 171 MB)
// find them there instead of in HBM.  What it is for and what it measured: profiles/EXPERIMENTS.md round 6.
//
// Start / stop without stream dependencies: the launch is enqueued with NO wait on any other stream (a dependency packet pending on its queue for a whole backward pass
// costs the neighbours of its dispatch pipe, profiles/EXPERIMENTS.md round 5 #11) -- the progress word reads 0 between rollouts (the decode stream zeroes it behind its
// last replay, and the host joins that stream before it goes on), and the launch returns when the `stop` word reaches its `epoch` (stored behind the last replay, also when
// EOS ended the rollout early).
//
// Pacing rules: a block never starts unit n + 1 before mark n + 1 - lead has been published (at most lead + 1 layers ahead: nothing is evicted before use);
// a block that finds the decode step already PAST its unit drops the rest of the unit (stale bytes only compete with the consumer); every wait is bounded
// (timeout_ms) so a decode stream that never arrives cannot hang the queue.
#include "common.h"

namespace {
struct PrefetchArgs {
    const long long* segs;      // device [n_units][n_seg][2]: (address, bytes) of the segments of unit u, in the order their consumers read them; bytes % 16 == 0
    int n_units, n_seg;
    const unsigned* mark;       // the decode step's progress word
    unsigned first_mark, last_mark;
    const unsigned* stop;       // optional: the launch returns once *stop >= epoch (the rollout stores its sequence number there behind its last replay)
    unsigned epoch;
    int lead;
    int nt;                     // 1: non-temporal loads (streaming hint), 0: plain
    unsigned long long timeout_ticks;
    int* status;                // optional device int[4]: timed-out flag, units read (block 0), units dropped as stale (block 0), MiB read (block 0)
};

// One 16-byte read whose value is dropped: the destination register is the same for every load of a wave (the hardware orders the write-backs; nothing reads it), so a
// trip keeps U requests per lane in flight with FOUR VGPRs -- the block must fit next to a resident 256 x 256 GEMM block of the side stream (432 of a SIMD's 512 VGPRs).
// The sink is a READ-WRITE operand ("+v"): the compiler does not know that the write-back arrives later, so the four registers must stay live from the first load to
// the final s_waitcnt -- as an output-only operand they were handed to address arithmetic between two loads and a late write-back turned an address into garbage
// (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION, round 6).
template <bool NT>
__device__ __forceinline__ void pf_touch(const u32x4_t* p, u32x4_t& sink) {
    if constexpr (NT) asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(sink) : "v"(p) : "memory");
    else asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(p) : "memory");
}

template <bool NT>
__global__ __launch_bounds__(256) void weight_prefetch_kernel(PrefetchArgs a) {
    constexpr int U = 16;                     // 16-byte loads in flight per lane: 64 KiB per block and trip (256 threads: one wave per SIMD, a few VGPRs each)
    __shared__ unsigned s_cur;
    __shared__ int s_quit;
    const int t = threadIdx.x, nb = gridDim.x, b = blockIdx.x;
    const unsigned long long t0 = wall_clock64();
    u32x4_t sink = {0, 0, 0, 0};
    unsigned next = a.first_mark;
    int done_units = 0, stale_units = 0;
    long long read_bytes = 0;
    while (next <= a.last_mark) {
        if (t == 0) {
            s_quit = 0;
            unsigned cur;
            while ((cur = __hip_atomic_load(a.mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) < next) {
                __builtin_amdgcn_s_sleep(16);       // ~0.5 us between polls: one lane of one wave per CU
                if (a.stop && __hip_atomic_load(a.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= a.epoch) { s_quit = 1; break; }      // the rollout is over
                if (wall_clock64() - t0 > a.timeout_ticks) {
                    if (a.status) a.status[0] = 1;
                    s_quit = 1;
                    break;
                }
            }
            s_cur = cur;
        }
        __syncthreads();
        if (s_quit) break;
        const unsigned cur = s_cur;
        __syncthreads();
        if (cur > a.last_mark) break;               // the rollout has ended (the host stores ~0 into the word behind the last replay)
        if (cur > next) next = cur;                 // late: the marks in between belong to launches that have already streamed their weights
        const unsigned unit_mark = next + (unsigned)a.lead;
        ++next;
        if (unit_mark > a.last_mark) continue;
        const long long* sg = a.segs + (long long)(unit_mark % (unsigned)a.n_units) * a.n_seg * 2;
        bool stale = false;
        for (int s = 0; s < a.n_seg && !stale; ++s) {
            const u32x4_t* p = (const u32x4_t*)sg[2 * s];
            const long long n16 = sg[2 * s + 1] >> 4;
            const long long stride = (long long)nb * 256;
            long long i = (long long)b * 256 + t;
            for (; i + (U - 1) * stride < n16; i += U * stride) {
#pragma unroll
                for (int u = 0; u < U; ++u) pf_touch<NT>(p + i + u * stride, sink);
                // has the decode step moved past this unit's consumers?  (one more request per trip, the same word for every lane)
                const unsigned now = __hip_atomic_load(a.mark, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                read_bytes += U * 16;
                if (now > unit_mark) { stale = true; break; }
            }
            if (!stale)
                for (; i < n16; i += stride) pf_touch<NT>(p + i, sink);
        }
        done_units += stale ? 0 : 1;
        stale_units += stale ? 1 : 0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((sink[0] ^ sink[1] ^ sink[2] ^ sink[3]) == 0x9e3779b9u && a.status) a.status[0] |= 2;      // (the sink is read once, after every load has returned)
    if (a.status && b == 0 && t == 0) {
        a.status[1] = done_units;
        a.status[2] = stale_units;
        a.status[3] = (int)((read_bytes * 256 * nb) >> 20);
    }
}
}  // namespace

extern "C" int iadr1_weight_prefetch(const long long* segs, int n_units, int n_seg, const unsigned* mark, unsigned first_mark, unsigned last_mark, const unsigned* stop,
                                     unsigned epoch, int lead, int nt, int n_blocks, int timeout_ms, int* status, hipStream_t stream) {
    IADR1_REQUIRE(segs != nullptr && mark != nullptr && n_units > 0 && n_seg > 0, "weight_prefetch: a segment table and a progress word are required");
    IADR1_REQUIRE(first_mark <= last_mark && lead >= 0 && lead < n_units, "weight_prefetch: marks [%u, %u], lead %d of %d units", first_mark, last_mark, lead, n_units);
    IADR1_REQUIRE(n_blocks > 0 && n_blocks <= 256 && timeout_ms > 0 && timeout_ms <= 60000, "weight_prefetch: 1..256 blocks (one per CU of the stream's mask) and a timeout in (0, 60000] ms");
    PrefetchArgs a{segs, n_units, n_seg, mark, first_mark, last_mark, stop, epoch, lead, nt, (unsigned long long)timeout_ms * 100000ull, status};
    if (nt) hipLaunchKernelGGL(weight_prefetch_kernel<true>, dim3(n_blocks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(weight_prefetch_kernel<false>, dim3(n_blocks), dim3(256), 0, stream, a);
    return iadr1_check_launch("weight_prefetch");
}
