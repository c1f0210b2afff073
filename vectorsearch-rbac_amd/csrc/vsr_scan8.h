// vsr_scan8.h -- the one-query call on the int8 planes (included by vsr_scan_l2.hip only: it defines a kernel and its launcher)
#pragma once
#include "vsr_scan.h"

namespace vsr {

#ifndef SCAN8_DEPTH
#define SCAN8_DEPTH 4          // tiles of a wave whose rows are in flight
#endif

// ---- K1 on the int8 planes: ONE query per call over a SIFT-like corpus (u8-exact rows, d <= 128, L2) ----
// The one-query call (the reference harness's shape) is a scan of the query's role partition plus the in-kernel merge; on
// the fp32 rows the scan is 512 bytes per row.  The int8 planes (x - 128, 128 bytes per row, |x - 128|^2 beside them) hold
// the same information for such corpora: |x - q|^2 = |x'|^2 + |q'|^2 - 2 x'.q' with x' = x - 128, q' = q - 128, every term an
// integer below 2^24, so the fp32 result is the value vector.c's loop produces, bit for bit.  Eight lanes share a row (one
// 16-byte chunk each, four v_dot4_i32_i8), a wave instruction reads eight rows = 1 KB, a tile of 16 rows is two instructions;
// the next tile's loads are issued before the current one is evaluated.  Top-k, publication and merge are K1's (fused_tail).
// The query is read as fp32 where the caller put it and converted by every workgroup; a query that is not integer-valued in
// 0..255 (device callers promise it with vsr_set_query_hint) makes the result FLAGGED.
__global__ __launch_bounds__(SCAN_THREADS, 2) void scan8_fused_kernel(const ScanParams p, uint32_t dim, uint32_t* q8_bad_host)
{
    constexpr int RW = 16, XCHK = scan_slack(RW) / (SCAN_WAVES * RW);
    static_assert(XCHK >= 1, "append slack");
    const uint64_t t_start = p.fused.dbg ? wall_clock64() : 0;
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = lane & 7;                                  // chunk of the row
    const int g = lane >> 3;                                 // row of the load instruction
    const ScanGroup grp = p.fused.group;
    const auto g_tiles = as_global(grp.tiles);
    const auto g_bitmap = as_global(grp.bitmap);
    const uint32_t local_block = blockIdx.x;
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t cap = p.cap, k = p.k;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                       // [cap]
    TopKCtrl* ctrl = reinterpret_cast<TopKCtrl*>(keys + cap);                 // [1]
    uint32_t* q8 = reinterpret_cast<uint32_t*>(ctrl + 1);                     // [32] the query as int8 (x - 128), 128 bytes
    uint32_t* flags = q8 + 32;                                                // [4] overflow votes, [4] = bad query, [5] = |q'|^2
    const uint32_t trigger = cap - (uint32_t) scan_slack(RW);
    const uint32_t iters = ((t1 - t0) + SCAN_WAVES - 1) / SCAN_WAVES;
    const uint32_t last_row = p.n_rows - 1u;

    // A wave's tiles are t0 + wave, t0 + wave + 8, ...  The scan of a role partition is a few dozen tiles per wave, i.e. nothing
    // but memory latency unless many loads are in flight: lane j of the wave fetches the descriptor (and permission window) of
    // the wave's j-th tile -- 64 tiles per load instruction --, and the rows of DEPTH tiles are in flight while one is evaluated.
    // Every load of the loop is issued unconditionally (masked lanes, clamped addresses), so the compiler keeps counting vmcnt.
    constexpr int DEPTH = SCAN8_DEPTH;
    struct Tile { uint4 x[2]; float rn[2]; };
    uint32_t round = 0;
    uint32_t d_start = 0, d_mask = 0;                        // lane j: tile cb + j of this wave
    uint32_t nj = 0;
    auto load_desc = [&](uint32_t cb) {
        const uint32_t t = t0 + (cb + (uint32_t) lane) * SCAN_WAVES + (uint32_t) wave;
        const bool have = cb + (uint32_t) lane < iters && t < t1;
        const uint32_t tc = have ? t : t0;                   // (t0 < t1 whenever there is a tile at all; else nothing is loaded)
        uint2 tl = make_uint2(0u, 0u);
        if (g_tiles) {
            if (t0 < t1) tl = load_tile(g_tiles, tc);
        } else {                                             // no filter: the identity tiling of the corpus
            tl.x = tc * RW;
            tl.y = p.n_rows - tl.x < (uint32_t) RW ? p.n_rows - tl.x : (uint32_t) RW;
        }
        uint32_t mask = tl.y >= 16u ? 0xFFFFu : (1u << tl.y) - 1u;
        if (g_bitmap) mask &= (uint32_t) bitmap_window(g_bitmap, tl.x);
        d_start = tl.x;
        d_mask = have ? mask : 0u;
        nj = iters - cb < 64u ? iters - cb : 64u;
    };
    auto fetch = [&](uint32_t j, Tile& tr, uint32_t& start, uint32_t& mask) {
        const uint32_t jc = j < nj ? j : 0u;
        start = (uint32_t) __builtin_amdgcn_readlane((int) d_start, (int) jc);
        mask = j < nj ? (uint32_t) __builtin_amdgcn_readlane((int) d_mask, (int) jc) : 0u;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            // a row the mask excludes reads the tile's first row instead (a line the wave fetches anyway): no branch around
            // the load, nothing extra from HBM; its value is dropped where the mask is applied
            const uint32_t row = start + (uint32_t) (r * 8 + g);
            const bool ok = (mask >> (r * 8 + g)) & 1u;
            const uint32_t rc = ok && row <= last_row ? row : (start <= last_row ? start : last_row);
            tr.x[r] = p.scr[(size_t) rc * 8 + l];         // (kernel-argument pointers: global_load)
            tr.rn[r] = p.norm2[rc];
        }
    };
    Tile ring[DEPTH];
    uint32_t r_start[DEPTH], r_mask[DEPTH];
    // the first descriptors and the first DEPTH tiles of rows are requested BEFORE the query is converted: the conversion (a
    // dependent global load, two barriers) then runs under their latency instead of in front of it
    float xs[4];                                             // the query first (every thread, clamped index: no branch), so that
#pragma unroll                                                // its wait does not cover the row loads behind it
    for (int e = 0; e < 4; ++e) {
        const uint32_t j = (uint32_t) (tid & 31) * 4 + (uint32_t) e;
        xs[e] = p.queries[j < dim ? j : 0u];
    }
    load_desc(0);
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) fetch((uint32_t) d, ring[d], r_start[d], r_mask[d]);

    if (tid == 0) {
        ctrl[0].tau = KEY_EMPTY;
        ctrl[0].count = 0;
    }
    if (tid < 8) flags[tid] = 0;
    __syncthreads();
    if (tid < 32) {                                          // four elements -> one word
        uint32_t word = 0, n2 = 0;
        bool bad = false;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t j = (uint32_t) tid * 4 + (uint32_t) e;
            int b = 0;
            if (j < dim) {
                const float x = xs[e];
                bad |= !(x >= 0.0f && x <= 255.0f && x == floorf(x));
                b = (int) fminf(fmaxf(x, 0.0f), 255.0f) - 128;
                n2 += (uint32_t) (b * b);
            }
            word |= ((uint32_t) b & 0xFFu) << (8 * e);
        }
        q8[tid] = word;
        atomicAdd(&flags[5], n2);
        if (bad) flags[4] = 1u;
    }
    __syncthreads();
    const bool bad_query = lds_peek(&flags[4]) != 0;
    const float qn = (float) lds_peek(&flags[5]);
    const int4 qv = *reinterpret_cast<const int4*>(q8 + l * 4);               // this lane's 16 query elements
    for (uint32_t cb = 0; cb < iters; cb += 64) {
        if (cb) {
            load_desc(cb);
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) fetch((uint32_t) d, ring[d], r_start[d], r_mask[d]);
        }
        for (uint32_t j0 = 0; j0 < nj; j0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const uint32_t j = j0 + (uint32_t) d, it = cb + j;
                const Tile cur = ring[d];
                const uint32_t c_start = r_start[d], c_mask = r_mask[d];
                fetch(j + DEPTH, ring[d], r_start[d], r_mask[d]);
                if (c_mask) {                                // wave-uniform
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        int dot = 0;
                        dot = __builtin_amdgcn_sdot4((int) cur.x[r].x, qv.x, dot, false);
                        dot = __builtin_amdgcn_sdot4((int) cur.x[r].y, qv.y, dot, false);
                        dot = __builtin_amdgcn_sdot4((int) cur.x[r].z, qv.z, dot, false);
                        dot = __builtin_amdgcn_sdot4((int) cur.x[r].w, qv.w, dot, false);
                        dot += __shfl_xor(dot, 1);
                        dot += __shfl_xor(dot, 2);
                        dot += __shfl_xor(dot, 4);
                        const uint32_t row = c_start + (uint32_t) (r * 8 + g);
                        const bool ok = l == 0 && ((c_mask >> (r * 8 + g)) & 1u);
                        const float v = fmaf(-2.0f, (float) dot, cur.rn[r] + qn);     // exact: integers below 2^24
                        const uint64_t key = make_key(v, row);                       // (never a list-ordered view: the launcher checks)
                        const bool pass = ok && key < lds_peek(&ctrl[0].tau);
                        topk_append(keys, &ctrl[0], pass, key);
                    }
                }
                if (j < nj && (it % XCHK) == XCHK - 1 && it + 1 < iters) {    // workgroup-uniform overflow vote (scan_kernel's)
                    const bool need = lds_peek(&ctrl[0].count) > trigger;
                    const uint32_t slot = round % 3;
                    if (need && lane == 0) atomicOr(&flags[slot], 1u);
                    __syncthreads();
                    const bool any = lds_peek(&flags[slot]) != 0;
                    if (tid == 0) flags[(round + 2) % 3] = 0;
                    ++round;
                    if (any && ctrl[0].count > trigger) topk_compact<SCAN_THREADS>(keys, &ctrl[0], k, tid, false);
                }
            }
        }
    }
    __syncthreads();
    const uint64_t t_scan = p.fused.dbg ? wall_clock64() : 0;
    // the k smallest, unordered, by a radix select on registers (vsr_scan.h: the merge does not need them sorted)
    uint32_t n;
    if (ctrl[0].count <= (uint32_t) (SCAN_THREADS * FUSED_KPT)) {
        const uint32_t have = ctrl[0].count;
        n = have > k ? fused_block_select<true>(keys, have, k, keys, tid) : have;
    } else {
        topk_compact<SCAN_THREADS>(keys, &ctrl[0], k, tid, false);
        n = ctrl[0].count < k ? ctrl[0].count : k;
    }
    uint64_t* dst = p.partial + (size_t) local_block * p.kp;
    for (uint32_t i = tid; i < p.kp; i += SCAN_THREADS)
        __hip_atomic_store(dst + i, i < n ? keys[i] : KEY_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (bad_query && blockIdx.x == 0 && tid == 0 && q8_bad_host) *q8_bad_host = 1u;
    fused_tail(p, grp, local_block, keys, tid, bad_query, t_start, t_scan, p.fused.dbg ? wall_clock64() : 0);
}

hipError_t launch_scan8_fused(const ScanParams& p, uint32_t dim, uint32_t* q8_bad_host, uint32_t n_blocks, hipStream_t s)
{
    if (p.rw != 16 || p.kp != p.k || !p.fused.enable || dim > 128 || p.rank) return hipErrorInvalidValue;
    const size_t lds = (size_t) p.cap * 8 + sizeof(TopKCtrl) + 32 * 4 + 8 * 4 + 16;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(scan8_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(scan8_fused_kernel, dim3(n_blocks), dim3(SCAN_THREADS), lds, s, p, dim, q8_bad_host);
    return hipGetLastError();
}

}  // namespace vsr
