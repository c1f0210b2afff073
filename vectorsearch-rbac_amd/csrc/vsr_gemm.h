// vsr_gemm.h — K2g: the batched-query x corpus GEMM path for LONG rows (d > 192: BASELINE configs 3 / 5's 768-d rows).
//
// K2w (vsr_mfmaw.h) keeps the B fragments of <= 128 queries in registers and stages 64 rows per workgroup.  For long rows
// that shape is bound by the path from L2 into the CU, not by the matrix pipe: per 64-float stage a workgroup pulls 16 KB
// of rows and 32 KB of query fragments for 48 MFMAs per wave -- ~60 bytes per CU cycle against the ~30 the CU can take in
// (MI355X_MICROARCH.md, "Indexed rows: gather into LDS": 66-73 GB/s per CU from L2); the round-2 counters showed exactly
// that: matrix pipe 26 % busy, waves waiting.  K2g is shaped like a GEMM instead:
//
//   * tile = 256 rows x 256 queries per workgroup of 8 waves (2 x 4 waves, 128 x 64 outputs each = 8 x 4 blocks of
//     v_mfma_f32_16x16x32_bf16, 128 accumulator registers per lane); BOTH operands go through LDS, so a K-step of 64
//     elements costs (256 + 256) x 128 bytes for 64 MFMAs per wave: 32 bytes per CU cycle at full matrix rate;
//   * operands are written into LDS by the load itself (global_load_lds_dwordx4: no staging registers, no ds_write), the
//     XOR swizzle of the 128-byte LDS rows is applied on the SOURCE address (cdna_hip_programming.md, rule 21);
//   * two stage buffers; the loads of K-step s + 1 are in flight under the MFMAs of step s, one barrier per K-step;
//   * the coarse planes are stored K-STEP-MAJOR in blocks of 256 rows (coarse_row_offset, vsr_device.h): the 64 bytes that
//     256 consecutive rows contribute to one K-step are one contiguous 16 KB slab, so an unfiltered tile streams whole
//     slabs (row-major planes made every K-step a 64-byte-per-1536-byte column walk over HBM: measured 2.5 TB/s);
//   * ONE product per element: the COARSE screening planes hold only hi = bf16(x) (vsr_corpus::d_scr_c), so a dot
//     product is x.q ~ xh.qh with |error| <= g |x||q|, g = 2^-8 (1 + 2^-9) + (d + 64) 2^-24 (bf16 rounds to nearest:
//     2^-9 relative per operand; fp32 accumulation).  That is 1/3 of K2w's products and 1/2 of its bytes.  The
//     screening only decides which kp candidates per query survive; select_rerank_kernel recomputes vector.c's exact
//     arithmetic for them and FLAGS a query unless the kept / dropped gap exceeds the bound, exactly as for K2w -- a
//     coarser screen needs a larger kp (the planner takes 4k) and flags sooner; a flagged query is re-run on the fine
//     planes (K2w) and, if still unproven, on the exact path (vsr_search, vsr_search_device_exact);
//   * thresholds are folded into the accumulators: for L2 the chain starts at (tau - |q|^2) / 2, so that "candidate" is
//     ONE compare per (row, query) pair, acc >= |x|^2 / 2 (IP: start at tau, compare with 0; cosine: compare with
//     (1 - tau) |q| |x|); survivors (~1 in 5000 pairs) are appended to their query's buffer with one atomic each.
//
// The same kernel is its own sample pass (SAMPLE: every ss-th tile of a workgroup, open thresholds, one minimum per
// query column and wave-tile), feeding seed_select_kernel like K2w's.
//
// Row mapping: a 256-row tile is 256 / rw list tiles of the pass (ScanGroup::tiles), resolved by threads 0..255 one
// tile ahead in three steps spread over three K-steps (descriptor -> row -> permission bit and |row|^2), each step's
// loads being complete at the next K-step's barrier anyway, so the mapping never stalls anybody.
#pragma once
#include <type_traits>
#include "vsr_device.h"
#include "vsr_topk.h"
#include "vsr_mfma.h"

namespace vsr {

#ifndef GM_READS_UPFRONT
#define GM_READS_UPFRONT 1
#endif
#ifndef GM_SETPRIO
#define GM_SETPRIO 0
#endif
#ifndef GM_SCHED
#define GM_SCHED 2
#endif
constexpr int GM_THREADS = 512;
constexpr int GM_BM = 256;                 // rows per workgroup tile
constexpr int GM_BN = 256;                 // query slots per pass
constexpr int GM_KC = 4;                   // 16-byte chunks (8 bf16) per row and K-step: 32 elements, 64 bytes
constexpr int GM_SLOTS = 4;                // stage ring: one slot being multiplied, three in flight
constexpr size_t GM_STAGE_U4 = (size_t) (GM_BM + GM_BN) * GM_KC;                  // uint4 per stage slot (32 KB)
// [stage ring | row ring: index, value | column constants, thresholds | mapping scratch: descriptor, bitmap word, norm | flag]
constexpr int GM_PARK = 64;               // candidates a wave parks per tile before they go to their queries' buffers
inline size_t gemm_lds_bytes() { return GM_SLOTS * GM_STAGE_U4 * 16 + 2 * GM_BM * 8 + GM_BN * 20 + GM_BM * 20 + 16 + 8 * (GM_PARK * 16 + 64 * 4); }

using lds_u4 = __attribute__((address_space(3))) uint4;
using lds_u32 = __attribute__((address_space(3))) uint32_t;

// Every vector-memory operation of the tile loop is an LDS-DMA load (global_load_lds): the waves count them with
// s_waitcnt vmcnt(N) and meet at a bare s_barrier, so that three stages stay in flight across barriers
// (cdna_hip_programming.md, "Pipelining across barriers").  Per K-step a wave issues its four operand pieces; the row
// mapping of the next tile (descriptor -> row -> permission word, |row|^2) rides in the same queue as 4-byte LDS-DMA
// loads into a scratch area and is read back two K-steps later, when the counted wait has covered it.
__device__ __forceinline__ void wave_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void gm_wait_barrier()
{
    asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int METRIC, bool SAMPLE>
__global__ __launch_bounds__(GM_THREADS, 2) void gemm_screen_kernel(const ScanParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2;                                                  // 128-row half of the tile
    const int wn = wave & 3;                                                   // 64-query quarter of the pass

    uint32_t lo = 0, mapped_block = 0;
    if (p.block_map) {
        const uint2 m = p.block_map[blockIdx.x];
        if (m.x == 0xFFFFFFFFu) return;                                        // padding workgroup of a short XCD lane
        lo = m.x;
        mapped_block = m.y;
    } else {
        uint32_t hi = p.n_groups;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (p.groups[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid;
        }
    }
    const ScanGroup grp = p.groups[lo];
    const uint32_t local_block = p.block_map ? mapped_block : blockIdx.x - grp.block_begin;
    const auto g_tiles = as_global(grp.tiles);
    const auto g_bitmap = as_global(grp.bitmap ? grp.bitmap : p.ones);          // no bitmap: one all-ones word
    const bool has_bitmap = grp.bitmap != nullptr;
    const auto g_norm2 = as_global(p.norm2);
    const auto g_rank = as_global(p.rank);

    uint4*   stage = reinterpret_cast<uint4*>(smem);
    int32_t* rowidx = reinterpret_cast<int32_t*>(smem + GM_SLOTS * GM_STAGE_U4 * 16);  // [2][256]
    float*   rowval = reinterpret_cast<float*>(rowidx + 2 * GM_BM);            // [2][256] what a candidate's acc must reach (NaN: no row)
    float4*  colc = reinterpret_cast<float4*>(rowval + 2 * GM_BM);             // [256] {acc start, compare scale, |q|^2, slot bits}
    float*   collim = reinterpret_cast<float*>(colc + GM_BN);                   // [256] the threshold itself (generic epilogue)
    uint32_t* mapd = reinterpret_cast<uint32_t*>(collim + GM_BN);               // [2][256] tile descriptor words (x | y) of every row slot
    uint32_t* mapb = mapd + 2 * GM_BM;                                          // [2][256] permission word (lo | hi)
    float*    mapn = reinterpret_cast<float*>(mapb + 2 * GM_BM);                // [256] |row|^2
    uint32_t* s_open = reinterpret_cast<uint32_t*>(mapn + GM_BM);               // any query column without a threshold
    // per wave: parked candidates {value bits, row, wave-local column, rank within the column} and the columns' counts
    uint32_t* pk_v = s_open + 4 + (size_t) wave * (GM_PARK * 4 + 64);
    uint32_t* pk_r = pk_v + GM_PARK;
    uint32_t* pk_c = pk_r + GM_PARK;
    uint32_t* pk_k = pk_c + GM_PARK;
    uint32_t* pk_n = pk_k + GM_PARK;                                            // [64] candidates per column of this wave
    // (all LDS lives in the one dynamic array: a second __shared__ object beside LDS-DMA staging can cost a vmcnt(0) per
    // fragment read, cdna_hip_programming.md "Three .s-level traps")

    const bool sample_fine = SAMPLE && (grp.partial_begin & 1u);               // sample pass: one entry per lane, not per column
    const uint32_t cstride4 = p.cstride4;                                      // chunks per coarse plane row
    const uint32_t nks = cstride4 / GM_KC;                                     // K-steps per tile (>= 8: d > 192)
    const uint32_t q_count = grp.q_count;

    // ---- query columns: thresholds folded into what the accumulators start from ----
    if (tid == 0) *s_open = 0u;
    __syncthreads();
    if (tid < GM_BN) {
        const bool qok = (uint32_t) tid < q_count;
        const uint32_t slot = p.q_slots[grp.q_begin + (qok ? (uint32_t) tid : 0u)];
        const uint64_t tau = (!SAMPLE && p.tau_init) ? p.tau_init[slot] : KEY_EMPTY;
        const bool open = tau == KEY_EMPTY;
        const float lim = mono_to_float((uint32_t) (tau >> 32));
        const float qn = p.q_norm2[slot];
        float start = 0.0f, scale = 0.0f;
        if (!SAMPLE && !open) {
            if constexpr (METRIC == M_L2) start = 0.5f * (lim - qn);
            else if constexpr (METRIC == M_IP) start = lim;
            else scale = (1.0f - lim) * sqrtf(qn);
        }
        if (!qok) {                                                            // pad column: nothing is ever a candidate
            if constexpr (METRIC == M_COSINE) scale = __builtin_inff();
            else start = -__builtin_inff();
        }
        if (qok && open) *s_open = 1u;
        colc[tid] = make_float4(start, scale, qn, __uint_as_float(qok ? slot : 0xFFFFFFFFu));
        collim[tid] = !qok ? -__builtin_inff() : open ? __builtin_inff() : lim;
    }
    // open thresholds (a query whose sample was too thin; the sample pass itself): the generic epilogue
    __syncthreads();
    const bool generic = SAMPLE || lds_peek(s_open) != 0u;

    // ---- this workgroup's tiles ----
    const uint32_t rw = p.rw, tps = GM_BM / rw;                                // list tiles per 256-row tile
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t n_super = (t1 - t0 + tps - 1) / tps;
    const uint32_t ss = p.sample_stride;
    const uint32_t n_it = (n_super + ss - 1) / ss;
    if (n_it == 0) return;
    const uint32_t tile_last = grp.n_tiles - 1u;
    const uint32_t last_row = p.n_rows - 1u;
    const uint32_t mslot = (uint32_t) tid & (GM_BM - 1);                        // row slot this thread maps (threads 0..255)

    // what row slot `mslot` of tile it_ is, from its list-tile descriptor d: (row or -1, clamped row)
    auto list_tile_of = [&](uint32_t it_) -> uint32_t { return t0 + it_ * ss * tps + mslot / rw; };
    auto slot_row = [&](uint32_t it_, uint2 d, uint32_t& rc) -> int32_t {
        const uint32_t t = list_tile_of(it_);
        const bool tile_ok = it_ < n_it && t < t1;
        const uint32_t r = mslot % rw;
        const uint32_t rr = d.x + r;
        const bool ok = tile_ok && r < d.y && rr <= last_row;
        rc = ok ? rr : 0u;
        return ok ? (int32_t) rr : -1;
    };
    auto commit = [&](uint32_t it_, int32_t row, uint64_t bw, float nrm) {
        if (row >= 0 && !((bw >> ((uint32_t) row & 63u)) & 1ull)) row = -1;
        float v;
        if (generic) v = nrm;                                                  // generic epilogue: |row|^2 itself
        else if constexpr (METRIC == M_L2) v = 0.5f * nrm;
        else if constexpr (METRIC == M_IP) v = 0.0f;
        else v = sqrtf(nrm);
        rowidx[(it_ & 1u) * GM_BM + mslot] = row;
        rowval[(it_ & 1u) * GM_BM + mslot] = row >= 0 ? v : __builtin_nanf("");
    };
    // tile 0 is mapped with plain loads (nothing is in flight yet); later tiles through the LDS-DMA chain in the loop
    if (tid < GM_BM) {
        const uint32_t t = list_tile_of(0);
        const uint2 d = load_tile(g_tiles, t < tile_last ? t : tile_last);
        uint32_t rc;
        const int32_t row = slot_row(0, d, rc);
        const uint64_t bw = g_bitmap[has_bitmap ? rc >> 6 : 0u];
        commit(0, row, bw, g_norm2[rc]);
    }
    __syncthreads();

    // ---- operand staging: wave w issues the 1-KB pieces 2w, 2w+1 of the row half and of the query half of a stage ----
    // piece = 16 slots x 64 bytes; lane -> (slot = 16 piece + (lane >> 2), LDS chunk = lane & 3) fetches source chunk
    // (lane & 3) ^ sw(slot) of that slot's plane row, sw(slot) = (-(slot >> 2)) & 3: with it the 16-byte fragment reads of
    // a 16-lane group (4 banks each) cover all 64 banks exactly once
    const uint32_t l_slot = (uint32_t) lane >> 2;
    const uint32_t l_src = ((uint32_t) lane & 3u) ^ ((0u - (l_slot >> 2)) & 3u);   // (slot >> 2) & 3 == (l_slot >> 2) & 3: pieces start at multiples of 16
    const uint4* a_cur[2];
    const uint4* a_nxt[2];
    const uint4* b_src[2];
#pragma unroll
    for (int pc = 0; pc < 2; ++pc) {
        const uint32_t qs = (uint32_t) (wave * 2 + pc) * 16u + l_slot;
        const uint32_t slot = p.q_slots[grp.q_begin + (qs < q_count ? qs : 0u)];
        b_src[pc] = p.q_scr_c + coarse_row_offset(slot, nks) + l_src;
    }
    auto rows_of = [&](uint32_t it_, const uint4* (&dst)[2]) {
        const int32_t* ridx = rowidx + (it_ & 1u) * GM_BM;
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            const int32_t r = ridx[(wave * 2 + pc) * 16 + (int) l_slot];
            dst[pc] = p.scr_c + coarse_row_offset((uint32_t) (r < 0 ? 0 : r), nks) + l_src;
        }
    };
    auto issue = [&](const uint4* (&a)[2], uint32_t ks, uint32_t slot_) {
        lds_u4* dst = (lds_u4*) (stage + (size_t) slot_ * GM_STAGE_U4);
#pragma unroll
        for (int pc = 0; pc < 2; ++pc) {
            __builtin_amdgcn_global_load_lds(as_global(a[pc] + (size_t) ks * COARSE_SLAB_U4), dst + (wave * 2 + pc) * 64, 16, 0, 0);
            __builtin_amdgcn_global_load_lds(as_global(b_src[pc] + (size_t) ks * COARSE_SLAB_U4), dst + GM_BM * GM_KC + (wave * 2 + pc) * 64, 16, 0, 0);
        }
    };

    // MFMA lane roles (16x16x32): A lane = (row li, k-octet kq); B / result lane = (k-octet kq | row quad kq, query jq)
    const int li = lane & 15;
    const int kq = lane >> 4;
    const uint32_t frag_off = (uint32_t) li * GM_KC + ((uint32_t) kq ^ ((0u - ((uint32_t) li >> 2)) & 3u));   // + 64 per 16-slot block
    f32x4 acc[8][4];

    // parked candidates of the previous tile -> their queries' buffers (see the epilogue)
    uint32_t n_park = 0, n_flush = 0;                                           // wave-uniform
    uint32_t fl_base = 0;                                                       // lane c: first position of column c's candidates
    const uint32_t my_col_slot = __float_as_uint(colc[wn * 64 + lane].w);      // lane c: query slot of the wave's column c
    auto flush_reserve = [&]() {
        n_flush = n_park;
        if (n_flush == 0) return;
        pk_n[lane] = 0u;
        wave_fence();
        if ((uint32_t) lane < n_flush) pk_k[lane] = atomicAdd(&pk_n[pk_c[lane]], 1u);      // LDS: rank within its column
        wave_fence();
        const uint32_t mine = pk_n[lane];
        if (mine) fl_base = atomicAdd(p.qcnt + my_col_slot, mine);             // one wave instruction; consumed a tile later
    };
    auto flush_store = [&]() {
        if (n_flush == 0) return;
        const uint32_t c = (uint32_t) lane < n_flush ? pk_c[lane] : 0u;
        const uint32_t base = (uint32_t) __shfl((int) fl_base, (int) c);
        const uint32_t slot = (uint32_t) __shfl((int) my_col_slot, (int) c);
        if ((uint32_t) lane < n_flush) {
            const uint32_t at = base + pk_k[lane];
            if (at < p.capq) p.qcand[(size_t) slot * p.capq + at] = ((uint64_t) pk_v[lane] << 32) | pk_r[lane];
        }
        n_flush = 0;
        wave_fence();
    };

    rows_of(0, a_cur);
    a_nxt[0] = a_cur[0];
    a_nxt[1] = a_cur[1];
    // stages are numbered through all tiles: g = it * nks + ks lives in ring slot g & 3; three are in flight
    issue(a_cur, 0, 0);
    issue(a_cur, 1, 1);
    issue(a_cur, 2, 2);
    uint32_t g = 0;
    for (uint32_t it = 0; it < n_it; ++it) {
        // accumulators start from the folded thresholds of their query columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float s = generic ? 0.0f : colc[wn * 64 + j * 16 + li].x;
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i][j] = f32x4{s, s, s, s};
        }
        for (uint32_t ks = 0; ks < nks; ++ks, ++g) {
            gm_wait_barrier();                               // stage g has landed in every wave's share; slot (g + 3) & 3 is free
            // ---- row mapping of tile it + 1 (threads 0..255), spread over K-steps 0 / 2 / 4 ----
            if (tid < GM_BM) {
                if (ks == 0) {
                    const uint32_t t = list_tile_of(it + 1);
                    const gptr<uint32_t> src = (gptr<uint32_t>) (g_tiles + (t < tile_last ? t : tile_last));
                    __builtin_amdgcn_global_load_lds(src, (lds_u32*) mapd + wave * 64, 4, 0, 0);
                    __builtin_amdgcn_global_load_lds(src + 1, (lds_u32*) mapd + GM_BM + wave * 64, 4, 0, 0);
                } else if (ks == 2) {
                    uint32_t rc;
                    (void) slot_row(it + 1, make_uint2(mapd[mslot], mapd[GM_BM + mslot]), rc);
                    const gptr<uint32_t> bsrc = (gptr<uint32_t>) (g_bitmap + (has_bitmap ? rc >> 6 : 0u));
                    __builtin_amdgcn_global_load_lds(bsrc, (lds_u32*) mapb + wave * 64, 4, 0, 0);
                    __builtin_amdgcn_global_load_lds(bsrc + 1, (lds_u32*) mapb + GM_BM + wave * 64, 4, 0, 0);
                    __builtin_amdgcn_global_load_lds((gptr<uint32_t>) (g_norm2 + rc), (lds_u32*) mapn + wave * 64, 4, 0, 0);
                } else if (ks == 4) {
                    uint32_t rc;
                    const int32_t row = slot_row(it + 1, make_uint2(mapd[mslot], mapd[GM_BM + mslot]), rc);
                    commit(it + 1, row, ((uint64_t) mapb[GM_BM + mslot] << 32) | mapb[mslot], mapn[mslot]);
                }
            }
            // ---- three stages ahead: stage g + 3 (of the next tile from K-step nks - 3 on) ----
            const uint32_t ks3 = ks + 3 < nks ? ks + 3 : ks + 3 - nks;
            if (ks + 3 == nks) rows_of(it + 1, a_nxt);       // (committed at K-step 4 <= nks - 4, a barrier ago at least)
            const bool nxt = ks + 3 >= nks;                   // past the last tile: row 0, never used
#if GM_SCHED < 2
            if (!nxt) issue(a_cur, ks3, (g + 3) & 3u);
            else issue(a_nxt, ks3, (g + 3) & 3u);
#endif
            const uint4* sa = stage + (size_t) (g & 3u) * GM_STAGE_U4 + (size_t) (wm * 128) * GM_KC + frag_off;
            const uint4* sb = stage + (size_t) (g & 3u) * GM_STAGE_U4 + (size_t) (GM_BM + wn * 64) * GM_KC + frag_off;
            bf16x8 bfr[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[j] = __builtin_bit_cast(bf16x8, sb[j * 64]);
#if GM_SCHED >= 2
            // The four LDS-DMA pieces of stage g + 3 go out one by one BETWEEN the MFMA groups: a piece costs the issuing wave
            // ~100 cycles, and issued in a burst right after the barrier both waves of a SIMD pay that at the same time, with
            // the matrix pipe idle.  All twelve fragment reads first (an LDS-DMA write may not overtake an LDS read).
            {
                bf16x8 afr[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) afr[i] = __builtin_bit_cast(bf16x8, sa[i * 64]);
                lds_u4* dst = (lds_u4*) (stage + (size_t) ((g + 3) & 3u) * GM_STAGE_U4);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
#pragma unroll
                    for (int i = 2 * qd; i < 2 * qd + 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[j], acc[i][j], 0, 0, 0);
                    const int pc = qd >> 1;
                    if ((qd & 1) == 0) {
                        const uint4* src = (nxt ? a_nxt[pc] : a_cur[pc]) + (size_t) ks3 * COARSE_SLAB_U4;
                        __builtin_amdgcn_global_load_lds(as_global(src), dst + (wave * 2 + pc) * 64, 16, 0, 0);
                    } else
                        __builtin_amdgcn_global_load_lds(as_global(b_src[pc] + (size_t) ks3 * COARSE_SLAB_U4),
                                                         dst + GM_BM * GM_KC + (wave * 2 + pc) * 64, 16, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#elif GM_READS_UPFRONT
            // all twelve fragment reads of the K-step are issued before its first MFMA: the LDS latency is paid once per
            // K-step, the later fragments arrive under the earlier blocks' MFMAs (counted lgkmcnt waits by the compiler)
            bf16x8 afr[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) afr[i] = __builtin_bit_cast(bf16x8, sa[i * 64]);
#if GM_SETPRIO
            __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[j], acc[i][j], 0, 0, 0);
#if GM_SETPRIO
            __builtin_amdgcn_s_setprio(0);
#endif
#if GM_SCHED == 1
            // emitted order: the four B fragments and two A fragments, then per row block its four MFMAs and the read of
            // the A fragment two blocks ahead (the compiler's own order waits for LDS four times per K-step)
            __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
            for (int n = 0; n < 6; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            }
            __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
#endif
#else
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                bf16x8 afr[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) afr[i] = __builtin_bit_cast(bf16x8, sa[(h * 4 + i) * 64]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[h * 4 + i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[i], bfr[j], acc[h * 4 + i][j], 0, 0, 0);
            }
#endif
        }
        a_cur[0] = a_nxt[0];
        a_cur[1] = a_nxt[1];

        // ---- epilogue: acc[i][j][r] belongs to row slot wm * 128 + i * 16 + kq * 4 + r and query column wn * 64 + j * 16 + li ----
        const int32_t* ridx = rowidx + (it & 1u) * GM_BM + wm * 128;
        const float* rval = rowval + (it & 1u) * GM_BM + wm * 128;
        auto append = [&](uint32_t slot, float v, int32_t row) {
            const uint32_t at = atomicAdd(p.qcnt + slot, 1u);
            if (at < p.capq) p.qcand[(size_t) slot * p.capq + at] = make_key(v, g_rank ? g_rank[row] : (uint32_t) row);
        };
        if (!generic) {
            // Candidates are PARKED in LDS while the tile's results are looked through (no atomic in there: a returning
            // global atomic per candidate would stall the wave a microsecond or two, eight times per tile).  Afterwards ONE
            // wave instruction reserves room for all of them (lane c: the candidates of column c), and the keys are stored
            // at the start of the NEXT tile's epilogue, when that atomic has long returned.
            flush_store();
            n_park = 0;
            float tr[8][4];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4 t4 = *reinterpret_cast<const float4*>(&rval[i * 16 + kq * 4]);
                tr[i][0] = t4.x; tr[i][1] = t4.y; tr[i][2] = t4.z; tr[i][3] = t4.w;
            }
            float cs[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) cs[j] = METRIC == M_COSINE ? colc[wn * 64 + j * 16 + li].y : 0.0f;
            // one pass over the 32 result blocks: the block's largest acc - (what it must reach) in 8 VALU operations (a NaN
            // row threshold drops out of fmaxf), one ballot, and only a block that holds a candidate (~8 of a wave-tile's
            // 8192 pairs are) looks at its four values again
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float sl[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if constexpr (METRIC == M_COSINE) sl[r] = fmaf(-cs[j], tr[i][r], acc[i][j][r]);
                        else sl[r] = acc[i][j][r] - tr[i][r];
                    }
                    const float best = fmaxf(fmaxf(sl[0], sl[1]), fmaxf(sl[2], sl[3]));
                    if (__ballot(best >= 0.0f)) {
                        const float4 cc = colc[wn * 64 + j * 16 + li];
                        const uint32_t slot = __float_as_uint(cc.w);
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const bool c = sl[r] >= 0.0f;
                            const uint64_t cm = __ballot(c);
                            if (!cm) continue;
                            const int32_t row = ridx[i * 16 + kq * 4 + r];
                            float v;
                            if constexpr (METRIC == M_L2) v = fmaf(-2.0f, acc[i][j][r] - cc.x, 2.0f * tr[i][r] + cc.z);
                            else if constexpr (METRIC == M_IP) v = -(acc[i][j][r] - cc.x);
                            else v = 1.0f - acc[i][j][r] * rsqrtf(tr[i][r] * tr[i][r] * cc.z);
                            const uint32_t at = n_park + __builtin_amdgcn_mbcnt_hi((uint32_t) (cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) cm, 0u));
                            if (c) {
                                if (at < (uint32_t) GM_PARK) {
                                    pk_v[at] = mono_bits(v);
                                    pk_r[at] = g_rank ? g_rank[row] : (uint32_t) row;
                                    pk_c[at] = (uint32_t) (j * 16 + li);
                                } else
                                    append(slot, v, row);                      // a burst beyond the parking area: directly
                            }
                            n_park += (uint32_t) __popcll(cm);
                        }
                    }
                }
            if (n_park > (uint32_t) GM_PARK) n_park = GM_PARK;
            flush_reserve();
        } else {
            // generic: the screening value of every valid pair.  SAMPLE keeps one minimum per query column and wave-tile
            // (128 rows); an open threshold admits every valid pair (only planned for filters that fit the buffer)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 cc = colc[wn * 64 + j * 16 + li];
                const uint32_t slot = __float_as_uint(cc.w);
                const float lim = collim[wn * 64 + j * 16 + li];                // (generic main pass: columns with a threshold keep it)
                uint64_t best = KEY_EMPTY;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int32_t row = ridx[i * 16 + kq * 4 + r];
                        const float nx = rval[i * 16 + kq * 4 + r];
                        const float v = screen_value<METRIC>(acc[i][j][r], nx, cc.z);
                        if (row < 0 || slot == 0xFFFFFFFFu) continue;
                        if constexpr (SAMPLE) {
                            const uint64_t key = make_key(v, g_rank ? g_rank[row] : (uint32_t) row);
                            best = key < best ? key : best;
                        } else {
                            if (!(v > lim)) append(slot, v, row);
                        }
                    }
                if constexpr (SAMPLE) {
                    if (!sample_fine) {                                        // one minimum per column and wave-tile (128 rows)
                        uint64_t o = __shfl_xor(best, 16);
                        best = o < best ? o : best;
                        o = __shfl_xor(best, 32);
                        best = o < best ? o : best;
                        if (kq != 0) best = KEY_EMPTY;
                    }
                    if (best != KEY_EMPTY) {
                        const uint32_t at = atomicAdd(p.qcnt + slot, 1u);
                        if (at < p.capq) p.qcand[(size_t) slot * p.capq + at] = best;
                    }
                }
            }
        }
    }
    if (!generic) flush_store();
    // the three stages still in flight write LDS: they must have landed before the workgroup gives its LDS back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <int METRIC>
hipError_t launch_gemm_metric(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    const size_t lds = gemm_lds_bytes();
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(GM_THREADS), lds, s, p);
        return hipGetLastError();
    };
    if (p.cstride4 < 8 * GM_KC || p.cstride4 % GM_KC != 0 || GM_BM % p.rw != 0 || !p.ones) return hipErrorInvalidValue;
    return p.sample_stride > 1 ? launch(gemm_screen_kernel<METRIC, true>) : launch(gemm_screen_kernel<METRIC, false>);
}

}  // namespace vsr
