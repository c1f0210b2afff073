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

// Wave-level selection without sorting: the wave holds R keys per lane in registers (KEY_EMPTY = no key, n_real real
// ones in total).  tau = a threshold such that exactly min(k, n_real) real keys are <= tau; kth = the largest of those
// (the exact k-th smallest when n_real >= k).  MSB-first radix select over 8-bit digits; `hist` = 256 wave-private
// LDS words.  Keys are unique, so the selection is exact.
template <int R>
__device__ __forceinline__ void wave_radix_select(const uint64_t (&reg)[R], uint32_t n_real, uint32_t k, uint32_t* hist,
                                                  int lane, uint64_t& tau, uint64_t& kth)
{
    tau = KEY_EMPTY - 1;                                     // n_real <= k: every real key
    if (n_real > k) {
        // digits every key agrees on carry no information (and would serialise all 64 lanes on one histogram bin):
        // start at the highest byte in which two keys differ
        uint64_t all_and = ~0ull, all_or = 0;
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (reg[r] != KEY_EMPTY) { all_and &= reg[r]; all_or |= reg[r]; }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint32_t al = (uint32_t) __shfl_xor((int) (uint32_t) all_and, d), ah = (uint32_t) __shfl_xor((int) (uint32_t) (all_and >> 32), d);
            const uint32_t ol = (uint32_t) __shfl_xor((int) (uint32_t) all_or, d), oh = (uint32_t) __shfl_xor((int) (uint32_t) (all_or >> 32), d);
            all_and &= ((uint64_t) ah << 32) | al;
            all_or |= ((uint64_t) oh << 32) | ol;
        }
        const uint64_t diff = all_and ^ all_or;              // non-zero: n_real > k >= 1 distinct keys
        const int top = (63 - __clzll((long long) diff)) >> 3 << 3;               // shift of the first useful digit
        uint64_t mask = top >= 56 ? 0ull : ~0ull << (top + 8);
        uint64_t prefix = all_or & mask;
        uint32_t need = k;
        for (int shift = top; shift >= 0; shift -= 8) {
#pragma unroll
            for (int t = 0; t < 4; ++t) hist[t * 64 + lane] = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (reg[r] != KEY_EMPTY && (reg[r] & mask) == prefix) atomicAdd(&hist[(uint32_t) (reg[r] >> shift) & 255u], 1u);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const uint4 h = *reinterpret_cast<const uint4*>(&hist[4 * lane]);       // bins 4*lane .. 4*lane+3
            const uint32_t s4 = h.x + h.y + h.z + h.w;
            uint32_t incl = s4;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = (uint32_t) __shfl_up((int) incl, d);
                if (lane >= d) incl += o;
            }
            const uint64_t reach = __ballot(incl >= need);                          // non-empty: the total is >= need
            const int L = __ffsll((unsigned long long) reach) - 1;
            uint32_t before = incl - s4, digit = 4u * (uint32_t) lane, cnt = h.x;
            if (before + h.x < need) { before += h.x; digit += 1; cnt = h.y;
                if (before + h.y < need) { before += h.y; digit += 1; cnt = h.z;
                    if (before + h.z < need) { before += h.z; digit += 1; cnt = h.w; } } }
            before = (uint32_t) __shfl((int) before, L);
            digit = (uint32_t) __shfl((int) digit, L);
            cnt = (uint32_t) __shfl((int) cnt, L);
            need -= before;
            prefix |= (uint64_t) digit << shift;
            mask |= 0xFFull << shift;
            tau = prefix;                                    // shift == 0 ends here (keys are unique: cnt == need == 1)
            if (cnt == need) {                               // the whole bin is wanted: no need to look at lower digits
                tau = prefix | ((1ull << shift) - 1ull);
                break;
            }
        }
    }
    kth = 0;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (reg[r] <= tau && reg[r] != KEY_EMPTY && reg[r] > kth) kth = reg[r];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) kth, d), hi = (uint32_t) __shfl_xor((int) (uint32_t) (kth >> 32), d);
        const uint64_t o = ((uint64_t) hi << 32) | lo;
        kth = o > kth ? o : kth;
    }
}

// Writes the selected keys (see wave_radix_select) to dst[0 .. want), unordered except that the largest one goes last
// when the selection is full (n_real >= k).  Returns want = min(k, n_real).
template <int R>
__device__ __forceinline__ uint32_t wave_emit_selected(const uint64_t (&reg)[R], uint32_t n_real, uint32_t k, uint64_t tau,
                                                       uint64_t kth, uint64_t* dst, int lane)
{
    const uint32_t want = n_real < k ? n_real : k;
    const bool full = n_real >= k && want > 0;
    const uint32_t room = full ? want - 1 : want;            // never write past the list, whatever the keys are
    uint32_t at = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool sel = reg[r] != KEY_EMPTY && reg[r] <= tau && !(full && reg[r] == kth);
        const uint64_t m = __ballot(sel);
        const uint32_t pos = at + (uint32_t) __popcll(m & ((1ull << lane) - 1ull));
        if (sel && pos < room) dst[pos] = reg[r];
        at += (uint32_t) __popcll(m);
    }
    if (full && lane == 0) dst[want - 1] = kth;
    return want;
}

// The `keep` smallest of a long key stream src[0 .. n) by ONE wave holding only 64 * R keys at a time: a chunk of the stream
// joins the survivors so far, the radix select keeps `keep` of them, and so on (the keep smallest of a union are among
// the keep smallest of its parts).  Survivors live in `out` (LDS or global, >= keep keys), unordered except that the
// largest goes last when n >= keep.  Returns min(keep, n); kth = the largest survivor.  Needs keep < 64 * R.
template <int R>
__device__ __forceinline__ uint32_t wave_select_stream(const uint64_t* src, uint32_t n, uint32_t keep, uint64_t* out,
                                                       uint32_t* hist, int lane, uint64_t& kth)
{
    uint32_t have = 0, pos = 0;
    kth = KEY_EMPTY;
    do {
        const uint32_t room = (uint32_t) (64 * R) - have;
        const uint32_t take = n - pos < room ? n - pos : room;
        uint64_t reg[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const uint32_t i = (uint32_t) (r * 64 + lane);
            uint64_t v = KEY_EMPTY;
            if (i < have) v = out[i];
            else if (i - have < take) v = src[pos + (i - have)];
            reg[r] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const uint32_t n_real = have + take;
        uint64_t tsel;
        wave_radix_select<R>(reg, n_real, keep, hist, lane, tsel, kth);
        have = wave_emit_selected<R>(reg, n_real, keep, tsel, kth, out, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        pos += take;
    } while (pos < n);
    return have;
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
