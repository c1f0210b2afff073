// vsr_topk.h — workgroup-level running top-k over 64-bit keys held in LDS (gfx950, wave64).
//
// key = (monotone(fp32 ranking value) << 32) | internal_row.  Keys are unique, so the top-k of a key
// set is a well-defined total order: (distance asc, NaN last, row asc).  Internal rows are sorted by
// (document_id, block_id) at corpus load, which makes this the tie rule of the CPU oracle
// (oracle/vsr_oracle.c cmp_cand) and a deterministic stand-in for PostgreSQL's unspecified tie order.
//
// Protocol: waves append keys below the running threshold tau with one LDS atomic per wave; every
// <= APPEND_SLACK appended keys the workgroup checks for overflow and, if needed, bitonic-sorts the
// buffer, keeps the k smallest and lowers tau.  A stale (larger) tau only admits extra candidates,
// never loses one.
#pragma once
#include "vsr_device.h"

namespace vsr {

struct __align__(16) TopKCtrl {
    uint64_t tau;
    uint32_t count;
    uint32_t pad;
};

__device__ __forceinline__ uint32_t mono_bits(float v)
{
    v = v + 0.0f;                                  // -0 -> +0 (PostgreSQL compares them equal)
    uint32_t u = __float_as_uint(v);
    if (v != v) u = 0x7FC00000u;                   // one canonical NaN; sorts after +Inf like float8
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float mono_to_float(uint32_t m)
{
    uint32_t u = (m & 0x80000000u) ? (m & 0x7FFFFFFFu) : ~m;
    return __uint_as_float(u);
}

__device__ __forceinline__ uint64_t make_key(float v, uint32_t row)
{
    return ((uint64_t) mono_bits(v) << 32) | row;
}

// Wave-aggregated append.  Caller guarantees capacity (overflow protocol above).
__device__ __forceinline__ void topk_append(uint64_t* keys, TopKCtrl* ctrl, bool pass, uint64_t key)
{
    const uint64_t m = __ballot(pass);
    if (m) {
        const int lane = __lane_id();
        const int leader = __ffsll((unsigned long long) m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&ctrl->count, (uint32_t) __popcll(m));
        base = __shfl(base, leader);
        const uint32_t rank = __popcll(m & ((1ull << lane) - 1ull));
        if (pass) keys[base + rank] = key;
    }
}

// In-place ascending bitonic sort of n (power of two) keys by NT threads.
template <int NT>
__device__ __forceinline__ void bitonic_sort_lds(uint64_t* keys, uint32_t n, int tid)
{
    for (uint32_t size = 2; size <= n; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (n >> 1); t += NT) {
                const uint32_t i = 2 * t - (t & (stride - 1));
                const uint32_t j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = keys[i], b = keys[j];
                if ((a > b) == up) {
                    keys[i] = b;
                    keys[j] = a;
                }
            }
            __syncthreads();
        }
    }
}

// The same sort by ONE wave over keys only it touches: no workgroup barrier (a wave's LDS operations execute in
// order; the wavefront fence keeps the compiler from moving them across the stage boundary).
__device__ __forceinline__ void bitonic_sort_wave(uint64_t* keys, uint32_t n, int lane)
{
    constexpr int B = 4;                           // pairs per lane per batch: all reads of a batch issue before its writes
    const uint32_t half = n >> 1;
    for (uint32_t size = 2; size <= n; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t0 = 0; t0 < half; t0 += 64 * B) {
                uint64_t a[B], b[B];
                uint32_t ii[B];
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const uint32_t t = t0 + (uint32_t) (u * 64 + lane);
                    ii[u] = 2 * t - (t & (stride - 1));
                    if (t < half) {
                        a[u] = keys[ii[u]];
                        b[u] = keys[ii[u] + stride];
                    }
                }
#pragma unroll
                for (int u = 0; u < B; ++u) {
                    const uint32_t t = t0 + (uint32_t) (u * 64 + lane);
                    const bool up = (ii[u] & size) == 0;
                    if (t < half && (a[u] > b[u]) == up) {
                        keys[ii[u]] = b[u];
                        keys[ii[u] + stride] = a[u];
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
}

__device__ __forceinline__ uint32_t next_pow2(uint32_t v)
{
    return v <= 2 ? 2u : 1u << (32 - __clz(v - 1));
}

// Keep the k smallest keys (sorted) and lower tau.  Must be called by all threads, after a barrier
// that orders every append before it.  `always_sort` forces sorted output even when count <= k.
template <int NT>
__device__ __forceinline__ void topk_compact(uint64_t* keys, TopKCtrl* ctrl, uint32_t k, int tid, bool always_sort)
{
    const uint32_t n = ctrl->count;                // same value in every thread: no writer until below
    if (n > k || (always_sort && n > 1)) {
        const uint32_t np2 = next_pow2(n);
        for (uint32_t i = n + tid; i < np2; i += NT) keys[i] = KEY_EMPTY;
        __syncthreads();
        bitonic_sort_lds<NT>(keys, np2, tid);      // ends with a barrier
        if (tid == 0 && n >= k) {
            ctrl->count = k;
            ctrl->tau = keys[k - 1];
        }
    }
    __syncthreads();
}

}  // namespace vsr
