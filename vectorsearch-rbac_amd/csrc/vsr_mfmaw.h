// vsr_mfmaw.h — K2w: GEMM-shaped shared-pass scan on the matrix cores for rows of 61 .. 192 floats.
//
// A permission class (or any filter part) that many queries see is a GEMM: dot[row][query], rows = the class's rows,
// queries = everyone whose role sees the class, K = d.  K2 (vsr_mfma.h) gave every WAVE its own row tile and 16 (32)
// queries, so a class seen by 330 queries was streamed 21 times.  K2w stages a 64-row tile ONCE per WORKGROUP in LDS
// and multiplies it against up to 64 queries whose B fragments live in the registers of the four waves:
//
//   >= 3 query groups (16 queries each): wave w owns group w and reads the whole shared tile
//   2 groups: two waves per group, two 16-row sub-tiles each; 1 group: four waves, one sub-tile each (row split),
//
// so narrow passes (a leaf class seen by 10 queries) still use all four waves.
//
// Arithmetic: the tile is not the fp32 rows but their SCREENING PLANES (vsr_device.h, plane_stride4): every element
// x = hi + mid (+ e, |e| <= 2^-18 |x|) as two bf16 values, so the dot products run on v_mfma_f32_16x16x32_bf16 at 16x
// the fp32 MFMA rate:  x q ~ xh qh + xh qm + xm qh  (fp32 accumulation).  The result only SCREENS: it decides which
// kp = 2k candidates per query survive; K5r recomputes the exact vector.c arithmetic for them from the fp32 rows and
// flags a query whose kept / dropped gap is inside the screening's error bound (plane_err_g).  A corpus whose elements
// are all exactly bf16 values (SIFT's 0..255 integers) has no mid plane at all (HO = "hi only"): half the bytes per
// row, and the screening is then exact for such queries.  With the dot products this cheap every pass is bound by the
// row stream, so what matters is bytes in flight:
//
// Pipeline (register staging, MI355X guide T14 / G15): a thread owns 4 chunks (16 bytes each) of every 64-row stage;
// the loads of tile i + DEPTH are issued as soon as the registers of tile i have been written to LDS, so DEPTH whole
// tiles per workgroup are in flight under the work of tile i; two LDS stage buffers, one barrier per stage.  The row
// mapping of a tile (tile descriptor -> row -> permission bit, |row|^2) is resolved by wave 0 up to 3 * DEPTH tiles
// ahead, one dependent load per DEPTH tiles, so its latency is covered like that of the row data.
//
// Candidates: every query of the call owns ONE buffer of `capq` keys in global memory (ScanParams::qcand / qcnt); there
// are no per-workgroup lists, no in-kernel compaction and no publish phase.  Room in a buffer is reserved with a
// RETURNING global atomic, and waiting for its result means s_waitcnt vmcnt(0): the row loads prefetched for the next
// tiles would be drained with it at every tile that has a survivor -- nearly all of them (64 x 64 pairs per tile).  So
// the main pass parks a wave's survivors {value, row, query slot} in a wave-private LDS ring (the position is the wave's
// running count, no atomic at all) and flushes them in two phases that ride on the row pipeline: the atomics (one per
// parked entry) are issued just before a tile's row loads, and the keys are stored after the next tile's rows have been
// waited for -- by then the positions are there too.  The sample pass keeps only minima (see its epilogue) and reserves
// directly.  The thresholds come from a sample pass of
// this same kernel (SAMPLE: every ss-th tile, threshold open, the query's sample buffer) through seed_select_kernel;
// select_rerank_kernel (vsr_kernels.hip) then picks the kp best of a query's buffer and re-ranks them exactly.  A buffer
// that overflows (count > capq) only loses candidates and is flagged there: the caller re-runs that query exactly.
#pragma once
#include <type_traits>
#include "vsr_device.h"
#include "vsr_scan.h"
#include "vsr_topk.h"
#include "vsr_mfma.h"
#include "vsr_i8s.h"

namespace vsr {

constexpr int MW_THREADS = 256;
constexpr int MW_WAVES = 4;
constexpr int MW_S = 16;                   // 16-byte chunks per row and stage
constexpr int MW_ROWS = 64;                // rows per workgroup tile
constexpr int MW_NQ = MF_NQ * MW_WAVES;    // query slots of a pass
constexpr int MW_RING = 8;                 // row-mapping ring (tiles): >= DEPTH + 2

// Tiles in flight per workgroup and workgroups per CU the register allocation aims at, by stages per row: the staging
// registers of DEPTH tiles (16 * NCH * DEPTH VGPRs) and the B fragments (16 or 32 * NCH) share the budget.
#ifndef VSR_MW_DEPTH1
#define VSR_MW_DEPTH1 1
#endif
#ifndef VSR_MW_OCC1
#define VSR_MW_OCC1 3
#endif
// Development ablations (-DVSR_ABLATE=bits, variant libraries only; results are wrong by construction): 1 = survivors are
// counted but never appended, 2 = no epilogue at all, 4 = no MFMA / A-fragment reads, 8 = row loads only for the first tile, 16 = no tiles at all (prologue only).
#ifndef VSR_ABLATE
#define VSR_ABLATE 0
#endif
constexpr int mw_depth(int nch) { return nch == 1 ? VSR_MW_DEPTH1 : nch == 2 ? 2 : 1; }   // (nch 0 = long rows: 1)
// (a deeper register ring for the sample pass -- all of a workgroup's few tiles in flight at once -- was measured: the
// occupancy it costs outweighs it: 72 -> 115 us)
#ifndef VSR_LONG_MAP_EVERY_STAGE
#define VSR_LONG_MAP_EVERY_STAGE 1     // (0: mapping loads at stage 0 only -- measured: no faster, 8.2 vs 8.0 ms at 1M x 768)
#endif
#ifndef VSR_MW_DEPTH8
#define VSR_MW_DEPTH8 1          // int8 planes: tiles are 8 KB, so a deeper ring is cheap in registers (8 VGPRs per tile)
#endif
constexpr int mw_sample_depth(int nch, int pl) { (void) pl; return mw_depth(nch); }
#ifndef VSR_MW_OCC8
#define VSR_MW_OCC8 4
#endif
#ifndef VSR_MW_OCCL
#define VSR_MW_OCCL 2            // long rows: two query groups per wave
#endif
constexpr int mw_occ(int nch, int pl = 0, bool sample = false)
{
    (void) sample;
    return pl == 2 ? VSR_MW_OCC8 : nch == 1 ? VSR_MW_OCC1 : nch == 0 ? VSR_MW_OCCL : 2;
}   // int8: 32 KB of LDS, ~120 VGPRs

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also releases global memory, and since gfx950 counts
// stores and loads in the one vmcnt queue that costs an s_waitcnt vmcnt(0): every prefetched row load would be drained at
// every stage.  Inside the tile loop only the LDS image and the LDS rings are handed between waves, so the waves wait for
// their own LDS operations (lgkmcnt) and meet at a bare s_barrier; the candidate stores to global memory are ordered by a
// full __syncthreads() before anybody reads them back (compaction, publish).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// NCH: stages per row (16 chunks = 256 bytes each): 64 floats per stage with hi + mid planes, 128 floats hi-only.
// PL: plane kind of the corpus (ScanParams::plane_ho): 0 = bf16 hi + mid, 1 = bf16 hi only, 2 = int8 (u8-exact corpus and
// queries, elements stored as x - 128; L2 only: |x' - q'|^2 = |x - q|^2, exact in int32 / fp32; 128 bytes per row, the
// tile image has 8 chunks per row and v_mfma_i32_16x16x64_i8 covers d = 128 in two instructions).
template <int METRIC, int NCH, bool SAMPLE, int PL, int DEPTH, int EPI = 0, int NGT = 1>
__global__ __launch_bounds__(MW_THREADS, mw_occ(NCH, PL, SAMPLE)) void mfma_wide_kernel(const ScanParams p)
{
    constexpr bool HO = PL == 1, I8 = PL == 2;
    // NCH == 0: LONG rows (more than 3 stages, d > 192): the stage count is a runtime value, ONE stage of rows and of B
    // fragments is in registers at a time, and the next stage's rows and fragments (the query planes sit in L2: 3 KB
    // per query at d = 768) are in flight under the current stage's MFMAs
    constexpr bool LONG = NCH == 0;
    constexpr int NCHR = LONG ? 1 : NCH;                                       // stages held in registers
    constexpr int SR = I8 ? 8 : MW_S;                                          // 16-byte chunks per row and stage
    constexpr int NU = I8 ? 2 : 4;                                             // chunks a thread stages per tile and stage
    static_assert(!I8 || (NCH == 1 && METRIC == M_L2), "int8 planes: one stage (d <= 128), L2 only");
    static_assert(!LONG || DEPTH == 1, "long rows: one stage ahead");
    static_assert(DEPTH >= 1 && DEPTH <= 4 && DEPTH + 2 <= MW_RING, "row-mapping ring too short");
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

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
    const auto g_tiles = as_global(grp.tiles);                                 // global_load, not flat (vsr_device.h)
    const auto g_bitmap = as_global(grp.bitmap);
    const auto g_norm2 = as_global(p.norm2);
    const auto g_ones = as_global(p.ones);
    const auto g_rank = as_global(p.rank);

    const uint32_t pstride4 = p.pstride4;
    const uint32_t q_count = grp.q_count;
    const bool sample_fine = SAMPLE && (grp.partial_begin & 1u);               // sample pass: one entry per lane, not per column

    // LDS: [stage buffers | row index ring | |row|^2 ring]
    uint4*    stage = reinterpret_cast<uint4*>(smem);                           // [2][64 * MW_S]
    unsigned char* after = smem + 2 * MW_ROWS * SR * 16;
    int32_t*  rowidx = reinterpret_cast<int32_t*>(after);                       // [MW_RING][64]
    float*    rownorm = reinterpret_cast<float*>(after + MW_RING * 64 * 4);     // [MW_RING][64]
    unsigned char* pend = after + MW_RING * 64 * 8;                             // parked survivors, per wave
    uint32_t* pend_v = reinterpret_cast<uint32_t*>(pend) + (size_t) wave * MW_PEND;                       // key high word
    uint32_t* pend_r = reinterpret_cast<uint32_t*>(pend + 4 * MW_PEND * 4) + (size_t) wave * MW_PEND;     // row
    uint32_t* pend_c = reinterpret_cast<uint32_t*>(pend + 8 * MW_PEND * 4) + (size_t) wave * MW_PEND;     // query column 0..15
    // int8 main launch with the mask epilogue: the candidate test runs on the integer accumulators.  |x - q|^2 <= tau  <=>
    // 2 dot + (tau - |q|^2) >= |x|^2; the chain of query column j starts at c0 = ceil((tau - |q|^2) / 2) and a pair is a
    // candidate when acc >= floor(|x|^2 / 2): ONE v_cmp per pair on the MFMA result as it stands (a superset of the exact
    // test by at most one distance unit, which only costs a few more parked candidates; their keys are exact)
    constexpr bool ITEST = I8 && EPI == 1 && !SAMPLE && METRIC == M_L2;
    int32_t* rowthr = reinterpret_cast<int32_t*>(pend + 12 * MW_PEND * 4);     // [MW_RING][64] floor(|row|^2 / 2); no row: INT_MAX

    // ---- wave roles ----
    const uint32_t ngt = (q_count + MF_NQ - 1) / MF_NQ;                         // 16-query groups of this pass (1..4)
    const uint32_t rsplit = ngt == 1 ? 4u : ngt == 2 ? 2u : 1u;                 // waves sharing one group's rows
    const uint32_t g0 = (uint32_t) wave / rsplit;                               // this wave's query group
    const uint32_t sub0 = ((uint32_t) wave % rsplit) * (4u / rsplit);           // its first 16-row sub-tile
    const bool gact = g0 < ngt;                                                 // wave-uniform

    // MFMA lane roles (16x16x32): A lane = (row li, k-octet kq); B / result lane = (k-octet kq | row quad kq, query jq)
    const int li = lane & 15;
    const int kq = lane >> 4;
    const int jq = li;
    constexpr int NBLK = I8 ? 2 : HO ? 4 : 2;                                  // K-blocks per stage (32 bf16 or 64 int8 each)
    bf16x8 bh[NCHR][NBLK], bm[NCHR][NBLK];                                      // B fragments: hi / mid planes of the query
    const uint4* my_qsrc = nullptr;                                            // LONG: where this lane's fragments come from
    i32x4 b8[NBLK];                                                            // ... or its int8 plane
    const uint32_t my_qi = g0 * MF_NQ + (uint32_t) jq < (uint32_t) MW_NQ ? g0 * MF_NQ + (uint32_t) jq : 0u;
    float my_qn;
    uint32_t my_slot;
    uint64_t my_tau;                                                            // this lane's query: threshold (KEY_EMPTY = open)
    {
        const uint32_t slot = p.q_slots[grp.q_begin + (my_qi < q_count ? my_qi : 0)];   // pad columns repeat query 0
        my_slot = slot;
        my_tau = p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        my_qn = p.q_norm2[slot];
        // query planes: per stage [hi chunks | mid chunks] (8 + 8 of a 64-float stage; 16 + 16 of a 128-float one)
        const uint4* qsrc = p.q_scr + (size_t) slot * (HO ? 2 * pstride4 : pstride4);
        my_qsrc = qsrc;
        if constexpr (I8) {
#pragma unroll
            for (int blk = 0; blk < NBLK; ++blk) b8[blk] = __builtin_bit_cast(i32x4, qsrc[blk * 4 + kq]);
        } else {
#pragma unroll
            for (int s = 0; s < NCHR; ++s)
#pragma unroll
                for (int blk = 0; blk < NBLK; ++blk) {
                    bh[s][blk] = __builtin_bit_cast(bf16x8, qsrc[s * (HO ? 32 : 16) + blk * 4 + kq]);
                    bm[s][blk] = __builtin_bit_cast(bf16x8, qsrc[s * (HO ? 32 : 16) + (HO ? 16 : 8) + blk * 4 + kq]);
                }
        }
    }

    // ---- this workgroup's tiles ----
    const uint32_t rw = p.rw, tps = MW_ROWS / rw;                               // list tiles per 64-row workgroup tile
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t n_super = (t1 - t0 + tps - 1) / tps;
    const uint32_t ss = p.sample_stride;                                        // sample pass: every ss-th tile
    const uint32_t n_it = (VSR_ABLATE & 16) ? 0u : (n_super + ss - 1) / ss;
    const bool qok = my_qi < q_count;
    const bool open = my_tau == KEY_EMPTY;
    // screening limit: a value passes unless it is greater (NaN values pass; an open threshold admits everything)
    const float tau_lim = open ? __builtin_inff() : mono_to_float((uint32_t) (my_tau >> 32));
    // (ITEST) what this lane's accumulators start from: ceil((tau - |q|^2) / 2); open threshold: everything passes, pad
    // column: nothing does.  Distances, norms and thresholds of the int8 path are integers below 2^24.
    const int32_t c0 = !qok ? -0x40000000 : open ? 0x3FFFFFFF : (((int32_t) tau_lim - (int32_t) my_qn) + 1) >> 1;
    // LONG rows: a pass holds up to 128 queries and wave w also owns query group w + 4 (the same A fragments feed both
    // groups' MFMAs: half the passes over the rows).  Its lanes carry a second set of per-query state.
    constexpr int NG = NGT;                                                    // query groups per wave (2: LONG only)
    static_assert(NG == 1 || LONG, "two query groups per wave: long rows only");
    const uint32_t g1 = (uint32_t) wave + 4u;
    const bool gact2 = NG == 2 && rsplit == 1u && g1 < ngt;                    // wave-uniform
    const uint32_t my_qi2 = g1 * MF_NQ + (uint32_t) jq;
    const bool qok2 = gact2 && my_qi2 < q_count;
    uint32_t my_slot2 = my_slot;
    float my_qn2 = 0.0f, tau_lim2 = -__builtin_inff();
    const uint4* my_qsrc2 = nullptr;
    bf16x8 bh2[NBLK], bm2[NBLK];                                               // group B's fragments of the current stage
    if constexpr (NG == 2) {
        my_slot2 = p.q_slots[grp.q_begin + (qok2 ? my_qi2 : 0u)];
        const uint64_t t2 = p.tau_init ? p.tau_init[my_slot2] : KEY_EMPTY;
        tau_lim2 = t2 == KEY_EMPTY ? __builtin_inff() : mono_to_float((uint32_t) (t2 >> 32));
        my_qn2 = p.q_norm2[my_slot2];
        my_qsrc2 = p.q_scr + (size_t) my_slot2 * (HO ? 2 * pstride4 : pstride4);
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            bh2[blk] = __builtin_bit_cast(bf16x8, my_qsrc2[blk * 4 + kq]);
            bm2[blk] = __builtin_bit_cast(bf16x8, my_qsrc2[(HO ? 16 : 8) + blk * 4 + kq]);
        }
    }
    const uint32_t col_slot = lane < MF_NQ ? my_slot : my_slot2;               // lane c < 16 * NG owns candidate column c

    // ---- row mapping pipeline (wave 0, lane = row slot): at tile `it` the descriptor of tile it+3*DEPTH is fetched, the
    // row / bitmap word / norm of tile it+2*DEPTH are started and those of tile it+DEPTH are written to the LDS ring, where
    // the loads issued during tile `it` (for tile it+DEPTH) and later that tile's epilogue find them: every dependent load
    // has the time of DEPTH tiles to arrive, like the row data itself ----
    // Branch-free on purpose (addresses are clamped, results selected): a load inside a conditional block leaves the
    // compiler's wait-count bookkeeping with several histories and it falls back to s_waitcnt vmcnt(0), which would
    // drain the row loads prefetched for the next tiles at every tile.  The planner always hands K2w an explicit tile
    // list (unfiltered queries use the corpus's identity list) and groups with n_tiles == 0 are never launched.
    const uint32_t tile_last = grp.n_tiles - 1u;
    auto fetch_desc = [&](uint32_t it_) -> uint2 {                 // (start, nrows) of this lane's list tile, or (0, 0)
        const uint32_t t = t0 + it_ * ss * tps + (uint32_t) lane / rw;
        const bool ok = it_ < n_it && t < t1;
        const uint2 d = load_tile(g_tiles, t < tile_last ? t : tile_last);
        return make_uint2(ok ? d.x : 0u, ok ? d.y : 0u);
    };
    // tile `it` uses ring slot D = it % DEPTH of every register ring below, so each is indexed at compile time
    int32_t  pend_row[DEPTH];                                      // tile it+2*DEPTH: row (before the permission bit)
    uint64_t pend_bw[DEPTH];                                       //                  its bitmap word (in flight)
    float    pend_nrm[DEPTH];                                      //                  its |row|^2 (in flight)
    uint2    dsc_a[DEPTH];                                         // tile it+3*DEPTH: descriptor (in flight)
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) {
        pend_row[j] = -1;
        pend_bw[j] = ~0ull;
        pend_nrm[j] = 0.0f;
        dsc_a[j] = make_uint2(0u, 0u);
    }
    bool bad_row = false;                                          // cannot happen; reported once at the end
    auto start_rows = [&](uint2 d, int32_t& row, uint64_t& bw, float& nrm) {
        const uint32_t r = (uint32_t) lane % rw;
        const uint32_t rr = d.x + r;
        const bool ok = r < d.y && rr < p.n_rows;
        bad_row |= r < d.y && rr >= p.n_rows;
        const uint32_t rc = ok ? rr : 0u;
        row = ok ? (int32_t) rr : -1;
        bw = *(g_bitmap ? g_bitmap + (rc >> 6) : g_ones);          // no bitmap: one all-ones word
        nrm = g_norm2[rc];
    };
    // Every wave runs the (identical) mapping loads and wave 0 writes the ring: a wave-0-only branch around the loads
    // would leave the compiler's wait-count bookkeeping with two histories to merge at every tile, and it then waits for
    // ALL outstanding loads -- the prefetched row data included -- instead of counting.
    auto finish_rows = [&](uint32_t it_, int32_t row, uint64_t bw, float nrm) {
        if (row >= 0 && !((bw >> ((uint32_t) row & 63u)) & 1ull)) row = -1;
        if (wave == 0) {
            rowidx[(it_ % MW_RING) * 64 + lane] = row;
            rownorm[(it_ % MW_RING) * 64 + lane] = row >= 0 ? nrm : __builtin_nanf("");   // NaN: an invalid slot's L2 value is NaN
                                                                                          // and fails every `<=` (fast path below)
            if constexpr (ITEST) rowthr[(it_ % MW_RING) * 64 + lane] = row >= 0 ? ((int32_t) nrm) >> 1 : 0x7FFFFFFF;
        }
    };
    {
        uint2 d[2 * DEPTH];
#pragma unroll
        for (int j = 0; j < 2 * DEPTH; ++j) d[j] = fetch_desc((uint32_t) j);
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) dsc_a[j] = fetch_desc((uint32_t) (2 * DEPTH + j));
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            int32_t r0;
            uint64_t b0;
            float n0;
            start_rows(d[j], r0, b0, n0);
            finish_rows((uint32_t) j, r0, b0, n0);
            start_rows(d[DEPTH + j], pend_row[j], pend_bw[j], pend_nrm[j]);
        }
    }
    __syncthreads();

    // ---- staging: thread -> (row slot u * 16 + lrow, chunk lchunk) of every stage ----
    const int lrow = I8 ? tid >> 3 : tid >> 4, lchunk = I8 ? tid & 7 : tid & 15;
    constexpr int LROWS = I8 ? 32 : 16;                            // row slots one staging step of the workgroup covers
    uint4 X[DEPTH][NCHR][NU];
    const uint32_t last_row = p.n_rows - 1u;
    auto issue = [&](auto dc, auto sc, uint32_t it_) {             // loads of tile it_, stage S into ring slot D (no waits)
        // An invalid slot (masked row, ragged tile) loads row 0 and its products are discarded by the epilogue's row
        // test (the runtime keeps corpora with NaN / Inf elements off the screening kernels).  Plane rows are zero
        // padded to whole stages, so every chunk index is in range.
        constexpr int D = decltype(dc)::value;
        constexpr int S = decltype(sc)::value;
        const uint32_t chunk = (uint32_t) (S * SR + lchunk);
        const int32_t* ridx = rowidx + (it_ % MW_RING) * 64;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int32_t r = ridx[u * LROWS + lrow];
            const uint32_t rc = (uint32_t) (r < 0 ? 0 : r);
            if ((VSR_ABLATE & 8) && it_ > 0) continue;
            const u32x4 v = *reinterpret_cast<const u32x4*>(p.scr + (size_t) (rc < last_row ? rc : last_row) * pstride4 + chunk);
            X[D][S][u] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    const uint32_t nch_rt = LONG ? pstride4 / (uint32_t) SR : (uint32_t) NCH;     // stages per row
    auto issue_rt = [&](uint32_t it_, uint32_t s_) {               // LONG: the rows of (tile it_, stage s_) into X[0][0]
        const uint32_t chunk = s_ * (uint32_t) SR + (uint32_t) lchunk;
        const int32_t* ridx = rowidx + (it_ % MW_RING) * 64;
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int32_t r = ridx[u * LROWS + lrow];
            const uint32_t rc = (uint32_t) (r < 0 ? 0 : r);
            const u32x4 v = *reinterpret_cast<const u32x4*>(p.scr + (size_t) (rc < last_row ? rc : last_row) * pstride4 + chunk);
            X[0][0][u] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto load_b = [&](uint32_t s_, bf16x8 (&h)[NBLK], bf16x8 (&m)[NBLK]) {    // LONG: B fragments of stage s_
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) {
            h[blk] = __builtin_bit_cast(bf16x8, my_qsrc[s_ * (HO ? 32u : 16u) + (uint32_t) (blk * 4 + kq)]);
            m[blk] = __builtin_bit_cast(bf16x8, my_qsrc[s_ * (HO ? 32u : 16u) + (HO ? 16u : 8u) + (uint32_t) (blk * 4 + kq)]);
        }
    };
    auto issue_tile = [&](auto dc, uint32_t it_) {
        if constexpr (LONG) {
            issue_rt(it_, 0u);
            return;
        }
        issue(dc, std::integral_constant<int, 0>{}, it_);
        if constexpr (NCH > 1) issue(dc, std::integral_constant<int, 1>{}, it_);
        if constexpr (NCH > 2) issue(dc, std::integral_constant<int, 2>{}, it_);
    };
    const uint32_t n_pad = (n_it + DEPTH - 1) / DEPTH * DEPTH;
    issue_tile(std::integral_constant<int, 0>{}, 0);
    if constexpr (DEPTH > 1) issue_tile(std::integral_constant<int, 1>{}, 1);
    if constexpr (DEPTH > 2) issue_tile(std::integral_constant<int, 2>{}, 2);
    if constexpr (DEPTH > 3) issue_tile(std::integral_constant<int, 3>{}, 3);

    // parked survivors -> their queries' buffers (all lanes of the wave; the wave's LDS operations complete in order)
    // Parking ring of this wave (wave-uniform bookkeeping, scalar registers): entries [p_head, p_tail) are parked, each
    // {key high word, row, query column of the wave's group}.  A flush takes the first f_n of them in two phases:
    //   A (right BEFORE a tile's row loads are issued): the entries are counted per column with ballots and lane c < 16
    //     reserves room for ALL entries of column c with ONE returning atomic on its query's counter -- atomics on one
    //     address serialise at ~0.2 us each on this chip, so per-entry atomics (~600 per query) would cost more than the
    //     whole scan;
    //   B (after the next tile's rows have been waited for; the atomics were issued before those loads, so their results
    //     are there too and nothing waits): entry -> position = its column's base + its rank, key stored.
    uint32_t p_head = 0, p_tail = 0, f_n = 0;
    constexpr int FL_R = 2;                                        // flush rounds: up to 64 * FL_R entries at a time
    uint32_t f_rank[FL_R];                                         // per lane and round: rank of its entry within the column
    uint32_t f_base = 0;                                           // lanes 0..15: first position of column `lane`
#pragma unroll
    for (int r = 0; r < FL_R; ++r) f_rank[r] = 0;
    auto flush_issue = [&]() __attribute__((always_inline)) {      // phase A
        const uint32_t have = p_tail - p_head;
        f_n = have < 64u * FL_R ? have : 64u * FL_R;
        uint32_t col[FL_R];
#pragma unroll
        for (int r = 0; r < FL_R; ++r) {
            const uint32_t e = (uint32_t) (r * 64 + lane);
            col[r] = e < f_n ? pend_c[(p_head + e) % MW_PEND] : 0xFFu;
        }
        uint32_t mine = 0;
#pragma unroll
        for (int c = 0; c < MF_NQ * NG; ++c) {
            uint32_t before = 0;
#pragma unroll
            for (int r = 0; r < FL_R; ++r) {
                const uint64_t m = __ballot(col[r] == (uint32_t) c);
                if (col[r] == (uint32_t) c)
                    f_rank[r] = before + __builtin_amdgcn_mbcnt_hi((uint32_t) (m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) m, 0u));
                before += (uint32_t) __popcll(m);
            }
            if (lane == c) mine = before;
        }
        if (lane < MF_NQ * NG && mine) f_base = atomicAdd(p.qcnt + col_slot, mine);   // lane c owns column c: (kq 0, group A's
                                                                                      // column c) or (kq 1, group B's c - 16)
    };
    auto flush_store = [&]() __attribute__((always_inline)) {      // phase B
#pragma unroll
        for (int r = 0; r < FL_R; ++r) {
            const uint32_t e = (uint32_t) (r * 64 + lane);
            const uint32_t idx = (p_head + e) % MW_PEND;
            const uint32_t c = e < f_n ? pend_c[idx] : 0u;
            const uint32_t base = (uint32_t) __shfl((int) f_base, (int) c);
            const uint32_t slot = (uint32_t) __shfl((int) col_slot, (int) c);
            if (e < f_n) {
                const uint32_t row = pend_r[idx];
                const uint32_t at = base + f_rank[r];
                if (at < p.capq)
                    p.qcand[(size_t) slot * p.capq + at] = ((uint64_t) pend_v[idx] << 32) | (g_rank ? g_rank[row] : row);
            }
        }
        p_head += f_n;
        f_n = 0;
    };
    // one entry per lane with `has` (wave-uniform call): the position is the wave's running count + the lane's rank
    auto park_mask = [&](uint64_t act, bool has, uint32_t key_hi, uint32_t row, uint32_t col = 0xFFFFFFFFu, uint32_t slot_ = 0u)
                         __attribute__((always_inline)) {
        if (col == 0xFFFFFFFFu) {                                  // group A: the lane's own column and query
            col = (uint32_t) jq;
            slot_ = my_slot;
        }
        const uint32_t room = MW_PEND - (p_tail - p_head);
        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t) (act >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) act, 0u));
        if (has) {
            if (rank < room) {
                const uint32_t at = (p_tail + rank) % MW_PEND;
                pend_v[at] = key_hi;
                pend_r[at] = row;
                pend_c[at] = col;
            } else {                                               // the ring is full (a burst): reserve directly
                const uint32_t ga = atomicAdd(p.qcnt + slot_, 1u);
                if (ga < p.capq) p.qcand[(size_t) slot_ * p.capq + ga] = ((uint64_t) key_hi << 32) | (g_rank ? g_rank[row] : row);
            }
        }
        const uint32_t n_act = (uint32_t) __popcll(act);
        p_tail += n_act < room ? n_act : room;
    };
    auto park = [&](bool has, uint32_t key_hi, uint32_t row, uint32_t col = 0xFFFFFFFFu, uint32_t slot_ = 0u)
                    __attribute__((always_inline)) {
        park_mask(__ballot(has), has, key_hi, row, col, slot_);
    };

    auto run = [&](auto nsc) {
        constexpr int NS = decltype(nsc)::value;                   // 16-row sub-tiles of this wave (1, 2 or 4)
        constexpr int NH = NS > 2 ? 2 : NS;                        // sub-tiles whose A fragments are live at a time
        int buf = 0;
        auto tile = [&](auto dc, uint32_t it) {
            constexpr int D = decltype(dc)::value;
            f32x4 acc[NS];
            f32x4 acc2[NS];                                        // LONG: the wave's second query group
            i32x4 acc8[NS];                                        // int8 planes: exact integer dot products
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc2[i] = f32x4{0.f, 0.f, 0.f, 0.f};
                acc8[i] = ITEST ? i32x4{c0, c0, c0, c0} : i32x4{0, 0, 0, 0};
            }

            auto do_stage = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                uint4* img = stage + (size_t) buf * (MW_ROWS * SR);
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int slot = u * LROWS + lrow;
                    img[slot * SR + (lchunk ^ (slot & (SR - 1)))] = X[D][S][u];  // XOR-swizzled image
                }
                if constexpr (S == 0) {                            // row mapping, one step per tile (see above)
                    if (f_n) flush_store();                        // the image write above waited for this tile's rows, which
                                                                   // were issued after the atomics: their results are here
                    finish_rows(it + DEPTH, pend_row[D], pend_bw[D], pend_nrm[D]);
                    start_rows(dsc_a[D], pend_row[D], pend_bw[D], pend_nrm[D]);
                    dsc_a[D] = fetch_desc(it + 3 * DEPTH);
                }
                lds_barrier();
                if constexpr (S == 0) {
                    if (!f_n && p_tail - p_head >= 64u * FL_R) flush_issue();
                }
                issue(dc, sc, it + DEPTH);                         // in flight under the work of DEPTH whole tiles (past the
                                                                   // last tile: all slots invalid, row 0 from the cache)
                if constexpr (I8) {
                    if (gact && !(VSR_ABLATE & 4)) {
#pragma unroll
                        for (int blk = 0; blk < NBLK; ++blk) {
                            i32x4 a8[NS];
#pragma unroll
                            for (int i = 0; i < NS; ++i) {
                                const int row = ((int) sub0 + i) * 16 + li;
                                a8[i] = __builtin_bit_cast(i32x4, img[row * SR + ((blk * 4 + kq) ^ (row & (SR - 1)))]);
                            }
#pragma unroll
                            for (int i = 0; i < NS; ++i) acc8[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a8[i], b8[blk], acc8[i], 0, 0, 0);
                        }
                    }
                } else if (gact && !(VSR_ABLATE & 4)) {
#pragma unroll
                    for (int h0 = 0; h0 < NS; h0 += NH)
#pragma unroll
                        for (int blk = 0; blk < NBLK; ++blk) {
                            bf16x8 ah[NH], am[NH];
#pragma unroll
                            for (int i = 0; i < NH; ++i) {
                                const int row = ((int) sub0 + h0 + i) * 16 + li;
                                ah[i] = __builtin_bit_cast(bf16x8, img[row * MW_S + ((blk * 4 + kq) ^ li)]);
                                if constexpr (!HO) am[i] = __builtin_bit_cast(bf16x8, img[row * MW_S + ((8 + blk * 4 + kq) ^ li)]);
                            }
                            // x q ~ xh qh + xh qm + xm qh  (the dropped xm qm and the split residues: plane_err_g)
#pragma unroll
                            for (int i = 0; i < NH; ++i) acc[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[S][blk], acc[h0 + i], 0, 0, 0);
#pragma unroll
                            for (int i = 0; i < NH; ++i) acc[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[S][blk], acc[h0 + i], 0, 0, 0);
                            if constexpr (!HO) {
#pragma unroll
                                for (int i = 0; i < NH; ++i) acc[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[S][blk], acc[h0 + i], 0, 0, 0);
                            }
                        }
                }
                buf ^= 1;
            };
            if constexpr (LONG) {
                for (uint32_t s = 0; s < nch_rt; ++s) {
                    uint4* img = stage + (size_t) buf * (MW_ROWS * SR);
#pragma unroll
                    for (int u = 0; u < NU; ++u) {
                        const int slot = u * LROWS + lrow;
                        img[slot * SR + (lchunk ^ (slot & (SR - 1)))] = X[0][0][u];
                    }
                    {
                        // row mapping of the tiles ahead: the loads run at EVERY stage (a load inside a conditional block
                        // would cost the counted waits; these hit the caches), their results are committed at stage 0
#if VSR_LONG_MAP_EVERY_STAGE
                        int32_t r2;
                        uint64_t b2;
                        float n2;
                        start_rows(dsc_a[0], r2, b2, n2);
                        const uint2 d2 = fetch_desc(it + 3);
                        if (s == 0) {
                            if (f_n) flush_store();
                            finish_rows(it + 1, pend_row[0], pend_bw[0], pend_nrm[0]);
                            pend_row[0] = r2;
                            pend_bw[0] = b2;
                            pend_nrm[0] = n2;
                            dsc_a[0] = d2;
                        }
#else
                        if (s == 0) {
                            if (f_n) flush_store();
                            finish_rows(it + 1, pend_row[0], pend_bw[0], pend_nrm[0]);
                            start_rows(dsc_a[0], pend_row[0], pend_bw[0], pend_nrm[0]);
                            dsc_a[0] = fetch_desc(it + 3);
                        }
#endif
                    }
                    lds_barrier();
                    if (s == 0 && !f_n && p_tail - p_head >= 64u * FL_R) flush_issue();
                    const bool last = s + 1 == nch_rt;
                    issue_rt(last ? it + 1 : it, last ? 0u : s + 1);       // next stage's rows ...
                    bf16x8 nh[NBLK], nm[NBLK], nh2[NBLK], nm2[NBLK];
                    load_b(last ? 0u : s + 1, nh, nm);                     // ... and B fragments, under this stage's MFMAs
                    if constexpr (NG == 2) {
                        const uint32_t sn = last ? 0u : s + 1;
#pragma unroll
                        for (int blk = 0; blk < NBLK; ++blk) {
                            nh2[blk] = __builtin_bit_cast(bf16x8, my_qsrc2[sn * (HO ? 32u : 16u) + (uint32_t) (blk * 4 + kq)]);
                            nm2[blk] = __builtin_bit_cast(bf16x8, my_qsrc2[sn * (HO ? 32u : 16u) + (HO ? 16u : 8u) + (uint32_t) (blk * 4 + kq)]);
                        }
                    }
                    if (gact) {
#pragma unroll
                        for (int h0 = 0; h0 < NS; h0 += NH)
#pragma unroll
                            for (int blk = 0; blk < NBLK; ++blk) {
                                bf16x8 ah[NH], am[NH];
#pragma unroll
                                for (int i = 0; i < NH; ++i) {
                                    const int row = ((int) sub0 + h0 + i) * 16 + li;
                                    ah[i] = __builtin_bit_cast(bf16x8, img[row * MW_S + ((blk * 4 + kq) ^ li)]);
                                    if constexpr (!HO) am[i] = __builtin_bit_cast(bf16x8, img[row * MW_S + ((8 + blk * 4 + kq) ^ li)]);
                                }
#pragma unroll
                                for (int i = 0; i < NH; ++i) acc[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[0][blk], acc[h0 + i], 0, 0, 0);
#pragma unroll
                                for (int i = 0; i < NH; ++i) acc[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[0][blk], acc[h0 + i], 0, 0, 0);
                                if constexpr (!HO) {
#pragma unroll
                                    for (int i = 0; i < NH; ++i) acc[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[0][blk], acc[h0 + i], 0, 0, 0);
                                }
                                if constexpr (NG == 2) if (gact2) {   // the same A fragments against the second group
#pragma unroll
                                    for (int i = 0; i < NH; ++i) acc2[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh2[blk], acc2[h0 + i], 0, 0, 0);
#pragma unroll
                                    for (int i = 0; i < NH; ++i) acc2[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm2[blk], acc2[h0 + i], 0, 0, 0);
                                    if constexpr (!HO) {
#pragma unroll
                                        for (int i = 0; i < NH; ++i) acc2[h0 + i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh2[blk], acc2[h0 + i], 0, 0, 0);
                                    }
                                }
                            }
                    }
#pragma unroll
                    for (int blk = 0; blk < NBLK; ++blk) {
                        bh[0][blk] = nh[blk];
                        bm[0][blk] = nm[blk];
                        if constexpr (NG == 2) {
                            bh2[blk] = nh2[blk];
                            bm2[blk] = nm2[blk];
                        }
                    }
                    buf ^= 1;
                }
            } else {
                do_stage(std::integral_constant<int, 0>{});
                if constexpr (NCH > 1) do_stage(std::integral_constant<int, 1>{});
                if constexpr (NCH > 2) do_stage(std::integral_constant<int, 2>{});
            }
            if constexpr (I8 && !ITEST) {
#pragma unroll
                for (int i = 0; i < NS; ++i)
                    acc[i] = f32x4{(float) acc8[i][0], (float) acc8[i][1], (float) acc8[i][2], (float) acc8[i][3]};
            }

            // results: acc[i][r] = dot(row slot (sub0 + i) * 16 + kq * 4 + r, query column jq of the wave's group).  Every
            // lane screens its pairs, reserves room in its query's buffer for all of its survivors with ONE returning
            // atomic and stores them.  Screening test in float: a value is a candidate unless it is greater than the
            // threshold's distance (NaN values and an open / NaN threshold pass): a superset of `key <= tau`.
            if (VSR_ABLATE & 2) {
                float sink = 0.f;
#pragma unroll
                for (int i = 0; i < NS; ++i) sink += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
                if (sink == 12345.678f) atomicOr(p.err, 2u);
            } else if (EPI == 1 && METRIC == M_L2 && !SAMPLE && !(VSR_ABLATE & 1)) {
                // L2 main pass, FEW survivors per wave-tile (EPI = 1; the planner picks it when it expects <= 4, else the
                // rounds below win): one compare per pair whose result is a lane mask in scalar registers -- an invalid row slot
                // has |row|^2 = NaN in the ring, a pad column has the limit -inf, so `value <= limit` is the whole test
                // (K2w corpora hold no NaN / Inf: vsr_corpus::k2_safe).  Survivors are then parked register by register:
                // the value of register j is a compile-time operand, no select tree, and the loop is over the ~3 - 10
                // survivors of the wave, not over lanes.
                if (gact) {
                    const int32_t* ridx = rowidx + (it % MW_RING) * 64;
                    const float* rnrm = rownorm + (it % MW_RING) * 64;
                    const float lim = qok ? tau_lim : -__builtin_inff();
                    uint64_t m[NS * 4];
                    uint64_t any = 0;
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const int base = ((int) sub0 + i) * 16 + kq * 4;
                        if constexpr (ITEST) {
                            const int4 th = *reinterpret_cast<const int4*>(&rowthr[(it % MW_RING) * 64 + base]);
                            const int32_t th4[4] = {th.x, th.y, th.z, th.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                m[i * 4 + r] = __ballot(acc8[i][r] >= th4[r]);
                                any |= m[i * 4 + r];
                            }
                        } else {
                            const float4 rn = *reinterpret_cast<const float4*>(&rnrm[base]);
                            const float nx4[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                m[i * 4 + r] = __ballot(screen_value<METRIC>(acc[i][r], nx4[r], my_qn) <= lim);
                                any |= m[i * 4 + r];
                            }
                        }
                    }
                    if (any) {
#pragma unroll
                        for (int j = 0; j < NS * 4; ++j)
                            if (m[j]) {                            // scalar test
                                const int slot = ((int) sub0 + (j >> 2)) * 16 + kq * 4 + (j & 3);
                                float dotf;
                                if constexpr (ITEST) dotf = (float) (acc8[j >> 2][j & 3] - c0);
                                else dotf = acc[j >> 2][j & 3];
                                const float v = screen_value<METRIC>(dotf, rnrm[slot], my_qn);
                                park_mask(m[j], (m[j] >> lane) & 1ull, mono_bits(v), (uint32_t) ridx[slot]);
                            }
                    }
                }
            } else {
              auto epi = [&](const f32x4 (&acc)[NS], float my_qn, float tau_lim, bool qok, uint32_t col, uint32_t eslot)
                             __attribute__((always_inline)) {
                const int32_t* ridx = rowidx + (it % MW_RING) * 64;
                const float* rnrm = rownorm + (it % MW_RING) * 64;
                uint32_t pmask = 0;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int base = ((int) sub0 + i) * 16 + kq * 4;
                    const int4 ri = *reinterpret_cast<const int4*>(&ridx[base]);
                    const float4 rn = *reinterpret_cast<const float4*>(&rnrm[base]);
                    const int32_t rows4[4] = {ri.x, ri.y, ri.z, ri.w};
                    const float nx4[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = screen_value<METRIC>(acc[i][r], nx4[r], my_qn);
                        const uint32_t ok = (uint32_t) !(v > tau_lim) & (uint32_t) (rows4[r] >= 0) & (uint32_t) qok;   // no branches
                        pmask |= ok << (i * 4 + r);
                    }
                }
                if (VSR_ABLATE & 1) {
                    if (pmask == 0xFFFFFFFFu) atomicOr(p.err, 2u);
                    pmask = 0;
                }
                if (__ballot(pmask != 0) != 0) {                   // wave-uniform
                    if constexpr (SAMPLE) {
                        // The seed only needs the SMALLEST sampled values of a query, so a lane contributes the minimum of
                        // its valid pairs (1 entry per 4 * NS rows; `fine` passes) or the four row-quad lanes of a query
                        // column reduce theirs to one (1 entry per 16 * NS rows).  The m-th smallest of any subset of
                        // the sample is >= the m-th smallest of the whole sample: the threshold only gets looser.
                        uint64_t best = KEY_EMPTY;
#pragma unroll
                        for (int j = 0; j < NS * 4; ++j) {
                            const int slot = ((int) sub0 + (j >> 2)) * 16 + kq * 4 + (j & 3);
                            const float v = screen_value<METRIC>(acc[j >> 2][j & 3], rnrm[slot], my_qn);
                            const uint64_t key = make_key(v, (uint32_t) ridx[slot]);
                            best = ((pmask >> j) & 1u) && key < best ? key : best;
                        }
                        if (!sample_fine) {
                            uint64_t o = __shfl_xor(best, 16);
                            best = o < best ? o : best;
                            o = __shfl_xor(best, 32);
                            best = o < best ? o : best;
                            if (kq != 0) best = KEY_EMPTY;
                        }
                        park(best != KEY_EMPTY, (uint32_t) (best >> 32), (uint32_t) best, col, eslot);
                    } else {
                        // one round per survivor of the busiest lane (1 - 2 at the usual ~1 % admission) instead of a
                        // predicated body per result register: the parking position is the wave's running count (wave
                        // uniform, in a scalar register) plus the lane's rank among the lanes of this round
                        while (__ballot(pmask != 0)) {
                            const bool has = pmask != 0;
                            const int j = has ? __builtin_ctz(pmask) : 0;
                            pmask &= pmask - 1u;                   // (0 stays 0)
                            const int slot = ((int) sub0 + (j >> 2)) * 16 + kq * 4 + (j & 3);
                            float a;                               // acc[j >> 2][j & 3] by a select tree on the bits of j
                            if constexpr (NS == 1) {
                                const float b0 = j & 1 ? acc[0][1] : acc[0][0], b1 = j & 1 ? acc[0][3] : acc[0][2];
                                a = j & 2 ? b1 : b0;
                            } else {
                                float c[NS];
#pragma unroll
                                for (int i = 0; i < NS; ++i) {
                                    const float b0 = j & 1 ? acc[i][1] : acc[i][0], b1 = j & 1 ? acc[i][3] : acc[i][2];
                                    c[i] = j & 2 ? b1 : b0;
                                }
                                if constexpr (NS == 2) a = j & 4 ? c[1] : c[0];
                                else {
                                    const float d0 = j & 4 ? c[1] : c[0], d1 = j & 4 ? c[3] : c[2];
                                    a = j & 8 ? d1 : d0;
                                }
                            }
                            const float v = screen_value<METRIC>(a, rnrm[slot], my_qn);
                            park(has, mono_bits(v), (uint32_t) ridx[slot], col, eslot);
                        }
                    }
                }
              };
              if (gact) epi(acc, my_qn, tau_lim, qok, (uint32_t) jq, my_slot);
              if constexpr (NG == 2) {
                  if (gact2) epi(acc2, my_qn2, tau_lim2, qok2, (uint32_t) (MF_NQ + jq), my_slot2);
              }
            }
        };
        // the loop body is branch-free around its loads and exactly periodic (the tile count is padded to a multiple of
        // DEPTH with all-invalid tiles) so that the compiler can COUNT the outstanding loads across the back edge
        for (uint32_t it = 0; it < n_pad; it += DEPTH) {
            tile(std::integral_constant<int, 0>{}, it);
            if constexpr (DEPTH > 1) tile(std::integral_constant<int, 1>{}, it + 1);
            if constexpr (DEPTH > 2) tile(std::integral_constant<int, 2>{}, it + 2);
            if constexpr (DEPTH > 3) tile(std::integral_constant<int, 3>{}, it + 3);
        }
    };
    if (rsplit == 4) run(std::integral_constant<int, 1>{});
    else if (rsplit == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 4>{});
    if (f_n) flush_store();                                                    // drain the parking ring
    while (p_tail != p_head) {
        flush_issue();
        flush_store();
    }

    if (bad_row) atomicOr(p.err, 1u);                                          // a tile reached past the corpus: results invalid
}

template <int METRIC>
hipError_t launch_mfmaw_metric(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    const uint32_t nch = p.plane_ho == 2 ? 1u : p.pstride4 / MW_S;
    const size_t lds = mfmaw_lds_bytes(p.plane_ho == 2);
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(MW_THREADS), lds, s, p);
        return hipGetLastError();
    };
    const bool sample = p.sample_stride > 1;
    const bool few = !sample && METRIC == M_L2 && p.epi == 1;                  // survivor handling of the main launch (EPI)
    auto pick = [&](auto nchc) -> hipError_t {
        constexpr int N = decltype(nchc)::value;
        constexpr int D = mw_depth(N);
        if (p.plane_ho == 2) {
            if constexpr (N == 1 && METRIC == M_L2) {                          // int8 planes: d <= 128, L2
                if ((few && (p.k2i & 1u)) || (sample && (p.k2i & 2u))) return launch_i8_stream(p, n_blocks, s);   // per-wave streams (vsr_i8s.h)
                return sample ? launch(mfma_wide_kernel<METRIC, 1, true, 2, mw_sample_depth(1, 2)>)
                              : few ? launch(mfma_wide_kernel<METRIC, 1, false, 2, VSR_MW_DEPTH8, 1>)
                                    : launch(mfma_wide_kernel<METRIC, 1, false, 2, VSR_MW_DEPTH8>);
            }
            return hipErrorInvalidValue;
        }
        if (p.plane_ho) {
            if constexpr (N <= 2)                                              // hi-only: 128 floats per stage, d <= 192 -> <= 2 stages
                return sample ? launch(mfma_wide_kernel<METRIC, N, true, 1, mw_sample_depth(N, 1)>)
                              : few ? launch(mfma_wide_kernel<METRIC, N, false, 1, D, METRIC == M_L2 ? 1 : 0>) : launch(mfma_wide_kernel<METRIC, N, false, 1, D>);
            return hipErrorInvalidValue;
        }
        return sample ? launch(mfma_wide_kernel<METRIC, N, true, 0, mw_sample_depth(N, 0)>)
                      : few ? launch(mfma_wide_kernel<METRIC, N, false, 0, D, METRIC == M_L2 ? 1 : 0>) : launch(mfma_wide_kernel<METRIC, N, false, 0, D>);
    };
    if (nch > 3 || (p.plane_ho == 1 && nch > 2)) {                            // long rows: runtime stage count (NCH = 0)
        if (p.plane_ho == 2) return hipErrorInvalidValue;
        if (p.qmax > 64) {                                                     // passes of up to 128 queries: two groups per wave
            if (p.plane_ho) return sample ? launch(mfma_wide_kernel<METRIC, 0, true, 1, 1, 0, 2>) : launch(mfma_wide_kernel<METRIC, 0, false, 1, 1, 0, 2>);
            return sample ? launch(mfma_wide_kernel<METRIC, 0, true, 0, 1, 0, 2>) : launch(mfma_wide_kernel<METRIC, 0, false, 0, 1, 0, 2>);
        }
        if (p.plane_ho) return sample ? launch(mfma_wide_kernel<METRIC, 0, true, 1, 1>) : launch(mfma_wide_kernel<METRIC, 0, false, 1, 1>);
        return sample ? launch(mfma_wide_kernel<METRIC, 0, true, 0, 1>) : launch(mfma_wide_kernel<METRIC, 0, false, 0, 1>);
    }
    switch (nch) {
    case 0: return hipErrorInvalidValue;
    case 1: return pick(std::integral_constant<int, 1>{});
    case 2: return pick(std::integral_constant<int, 2>{});
    case 3: return pick(std::integral_constant<int, 3>{});
    default: return hipErrorInvalidValue;
    }
}

}  // namespace vsr
