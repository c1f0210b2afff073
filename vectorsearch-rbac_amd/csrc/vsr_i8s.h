// vsr_i8s.h — K2i: the int8 main launch as independent per-wave streams (SIFT-like corpora: d = 65 .. 128, u8-exact rows).
//
// K2w's int8 instantiation (vsr_mfmaw.h) stages a 64-row tile per WORKGROUP through registers: load -> ds_write -> barrier
// -> fragment reads, one tile (8 KB) per workgroup in flight, four workgroups per CU = 32 KB per CU.  HBM needs about
// 8 TB/s x 2 us = 16 MB in flight chip-wide (64 KB per CU) before it streams at full rate; the counters of round 2 showed
// the waves waiting 59 % of their cycles and the launch at 4.8 TB/s of pass bytes.  A deeper register ring costs VGPRs the
// kernel does not have (measured: spills).  K2i keeps the arithmetic and the candidate protocol and changes the data path:
//
//   * every WAVE is its own stream: it owns a ring of four 32-row stages in LDS (4 KB each), fills it with LDS-DMA
//     (global_load_lds_dwordx4: no staging registers, no ds_write), reads its A fragments from its own ring and multiplies
//     them against ALL the pass's queries, whose B fragments (up to 64 queries x 128 bytes) sit in its registers.  Nothing is
//     shared between the waves of a workgroup inside the loop, so there is NO barrier: a wave waits for its own loads with
//     a counted s_waitcnt vmcnt and three stages (12 KB) per wave stay in flight: 8 waves per CU = 96 KB per CU;
//   * the row mapping (list tile -> first row, row count, permission bits) rides in the same queue as 4-byte LDS-DMA loads,
//     32 list tiles (16 stages) at a time and two such chunks ahead; |row|^2 of a stage's 32 rows is one more LDS-DMA load
//     next to its four row pieces.  No load of the loop's common path returns into a register, so no wait drains the queue;
//   * the candidate test is K2w's integer test: the chain of query column j starts at c0 = ceil((tau - |q|^2) / 2), a
//     pair is a candidate when acc >= floor(|x|^2 / 2) (one v_cmp per pair).  Candidates (~1.5 per stage) are parked in
//     LDS; when half the parking area is used the wave reserves room for all of them with one returning atomic per query
//     column and stores the keys -- synchronously: that wait drains the stream, once per ~20 stages.
//
// Two things the compiler must not do to this loop (hipcc, ROCm 7.2; checked in the .s): (1) wait for the B fragments'
// loads inside the loop -- they are first used in blocks it branches around, so an empty asm "uses" them before the loop;
// (2) put s_waitcnt vmcnt(0) in front of an LDS read because LDS-DMA writes are pending: reads written as
// __builtin_bit_cast(vector, ((const uint4*) p)[i]) are left alone, reads through float4 / int4 / scalar lvalues of a region
// that is also stored to are not (observed, not explained).
//
// LDS image of a stage: 32 rows x 8 chunks of 16 bytes; chunk c of row r sits at position c ^ ((r >> 1) & 7), applied on the
// SOURCE address of the DMA (the DMA writes lane i at base + 16 i): the sixteen lanes of an A-fragment read (rows 0..15 of
// a block, one k-chunk) then cover sixteen different 16-byte bank groups.
//
// Same inputs and outputs as K2w's main launch (ScanParams: groups, tiles, bitmaps, thresholds, per-query candidate
// buffers); the sample pass stays K2w's.  The launcher falls back to K2w when the shape does not fit (rw != 16, an
// explicit sample stride, the rounds epilogue).
#pragma once
#include "vsr_device.h"
#include "vsr_topk.h"
#include "vsr_mfma.h"
#include "vsr_gemm.h"

namespace vsr {

constexpr int KI_THREADS = 256;
constexpr int KI_WAVES = 4;
constexpr int KI_ROWS = 32;                // rows per stage: two list tiles of 16 rows
constexpr int KI_SLOTS = 4;                // stage ring of a wave: one being multiplied, three in flight
constexpr int KI_STAGE_U4 = KI_ROWS * 8;   // uint4 per stage (4 KB)
constexpr int KI_CHUNK = 8;                // stages per mapping chunk (16 list tiles)
constexpr int KI_PARK = 64;                // candidates a wave parks before it reserves room for them
constexpr int KI_FLUSH_AT = 32;            // ... and the fill level that triggers the reservation
constexpr int KI_NQ = 128;                 // query columns of a pass, at most (NQG = 8 groups of 16)
// per wave: [stage ring | |row|^2 ring | thresholds of the current stage | descriptor ring (3 chunks) | permission words ring
// (2 chunks) | parked candidates {value, row, column} | per-column counts]; per workgroup: [column constants]
constexpr size_t KI_WAVE_BYTES = (size_t) KI_SLOTS * KI_STAGE_U4 * 16 + KI_SLOTS * KI_ROWS * 4 + KI_ROWS * 4 + 3 * 2 * (2 * KI_CHUNK) * 4 +
                                 2 * 4 * (2 * KI_CHUNK) * 4 + KI_PARK * 12 + KI_NQ * 4;
inline size_t i8s_lds_bytes() { return KI_WAVES * KI_WAVE_BYTES + KI_NQ * 16 + 16; }
static_assert(2 * (KI_WAVES * KI_WAVE_BYTES + KI_NQ * 16 + 16) <= 160 * 1024, "two workgroups per CU");

// SAMPLE: the threshold-seeding pass (ScanParams::sample_stride > 1): every ss-th stage of the workgroup's range, no
// thresholds; every lane keeps the smallest (value, row) of each of its query columns over the wave's whole stream and
// appends them to the queries' sample buffers when the stream ends (4 entries per column and wave: a subset of the
// sampled values, which can only loosen the seed -- vsr_mfmaw.h).  The values are compared as integers (|x'|^2 - 2 x'.q').
template <int NQG, bool SAMPLE = false>
__global__ __launch_bounds__(KI_THREADS, 2) void i8_stream_kernel(const ScanParams p)
{
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
    const auto g_tiles = as_global(grp.tiles);
    const auto g_bitmap = as_global(grp.bitmap ? grp.bitmap : p.ones);          // no bitmap: one all-ones word
    const bool has_bitmap = grp.bitmap != nullptr;
    const auto g_norm2 = as_global(p.norm2);
    const auto g_rank = as_global(p.rank);
    const uint32_t q_count = grp.q_count;

    unsigned char* wbase = smem + (size_t) wave * KI_WAVE_BYTES;
    uint4*    ring = reinterpret_cast<uint4*>(wbase);                           // [4][32 rows][8 chunks]
    float*    nring = reinterpret_cast<float*>(wbase + (size_t) KI_SLOTS * KI_STAGE_U4 * 16);   // [4][32] |row|^2 (DMA)
    int32_t*  thr = reinterpret_cast<int32_t*>(nring + KI_SLOTS * KI_ROWS);     // [32] floor(|row|^2 / 2); no row: INT_MAX
    constexpr uint32_t CT = 2 * KI_CHUNK;                                       // list tiles per mapping chunk
    uint32_t* dring = reinterpret_cast<uint32_t*>(thr + KI_ROWS);               // [3][2][CT] descriptor words x | y of a chunk's list tiles
    uint32_t* wring = dring + 3 * 2 * CT;                                       // [2][4][CT] permission words (w0 lo, hi, w1 lo, hi)
    uint32_t* pk_v = wring + 2 * 4 * CT;                                        // parked candidates: key high word
    uint32_t* pk_r = pk_v + KI_PARK;                                            //                    row
    uint32_t* pk_c = pk_r + KI_PARK;                                            //                    query column
    uint32_t* pk_n = pk_c + KI_PARK;                                            // [NC] candidates per column, then their first positions
    float4*   colc = reinterpret_cast<float4*>(smem + KI_WAVES * KI_WAVE_BYTES);   // [NC] {c0 bits, |q|^2, slot bits, -}
    uint32_t* s_done = reinterpret_cast<uint32_t*>(colc + KI_NQ);                // SAMPLE: waves of the workgroup whose stream is over
    constexpr int NC = NQG * 16;                                                // query columns of this instantiation

    // ---- this wave's stages: stage i of the wave = list tiles t0 + 2 (wave + 4 i) + {0, 1} of the workgroup's range ----
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t ss = SAMPLE ? p.sample_stride : 1u;                          // sample pass: every ss-th stage
    const uint32_t n_st = (((t1 - t0 + 1u) >> 1) + ss - 1u) / ss;
    const uint32_t n_w = n_st > (uint32_t) wave ? (n_st - (uint32_t) wave + 3u) >> 2 : 0u;
    const uint32_t tile_last = grp.n_tiles - 1u;
    const uint32_t last_row = p.n_rows - 1u;

    // list tile J (0 .. ) of mapping chunk c, as this wave numbers them: stage KI_CHUNK c + (J >> 1), half J & 1
    auto tile_of = [&](uint32_t c, uint32_t J) -> uint32_t { return t0 + 2u * ss * ((uint32_t) wave + 4u * (c * KI_CHUNK + (J >> 1))) + (J & 1u); };
    auto fetch_desc = [&](uint32_t c) {                                        // chunk c's descriptors -> LDS (lane L < CT: tile L)
        if ((uint32_t) lane >= CT) return;
        const uint32_t t = tile_of(c, (uint32_t) lane);
        const gptr<uint32_t> src = (gptr<uint32_t>) (g_tiles + (t < tile_last ? t : tile_last));
        uint32_t* dst = dring + (c % 3u) * (2 * CT);
        __builtin_amdgcn_global_load_lds(src, (lds_u32*) dst, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 1, (lds_u32*) dst + CT, 4, 0, 0);
    };
    // The first two chunks' descriptors are requested before anything else: their latency runs under the query columns' and
    // the B fragments' loads (a wave lives ~30 stages; every serialised round trip of the prologue is ~5 % of it).
    if (n_w) {
        fetch_desc(0);
        fetch_desc(1);
    }

    // ---- query columns: the thresholds folded into what the integer accumulators start from (vsr_mfmaw.h, ITEST) ----
    if (tid == 0) *s_done = 0u;
    if (tid < NC) {
        const bool qok = (uint32_t) tid < q_count;
        const uint32_t slot = p.q_slots[grp.q_begin + (qok ? (uint32_t) tid : 0u)];
        const uint64_t tau = !SAMPLE && p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        const bool open = tau == KEY_EMPTY;
        const float lim = open ? __builtin_inff() : mono_to_float((uint32_t) (tau >> 32));
        const float qn = p.q_norm2[slot];
        const int32_t c0 = SAMPLE ? 0 : !qok ? -0x40000000 : open ? 0x3FFFFFFF : (((int32_t) lim - (int32_t) qn) + 1) >> 1;
        colc[tid] = make_float4(__int_as_float(c0), qn, __uint_as_float(qok ? slot : 0xFFFFFFFFu), 0.0f);
    }
    __syncthreads();
    if (n_w == 0) return;                                                       // (after the only workgroup barrier)

    // MFMA lane roles (16x16x64 int8): A lane = (row li, 16-byte k-chunk kq); B / result lane = (k-chunk kq | row quad kq, query li)
    const int li = lane & 15;
    const int kq = lane >> 4;
    const uint32_t ngt = (q_count + 15u) >> 4;                                  // 16-query groups in use (wave-uniform)
    i32x4 b8[NQG][2];
    int32_t c0j[NQG];
#pragma unroll
    for (int j = 0; j < NQG; ++j) {
        const uint32_t qi = (uint32_t) (j * 16 + li);
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0u)];
        const uint4* qsrc = p.q_scr + (size_t) slot * p.pstride4;
        b8[j][0] = __builtin_bit_cast(i32x4, qsrc[kq]);
        b8[j][1] = __builtin_bit_cast(i32x4, qsrc[4 + kq]);
        c0j[j] = __float_as_int(colc[qi].x);
    }
    // the fragments are first used inside the loop, in blocks the compiler branches around: without a use here it would wait
    // for "their" loads (vmcnt(0): the whole queue) at every one of those blocks of every stage
#pragma unroll
    for (int j = 0; j < NQG; ++j) asm volatile("" : "+v"(b8[j][0]), "+v"(b8[j][1]), "+v"(c0j[j]));

    // first row of tile J of chunk c (0 for a tile past the workgroup's range: its loads read rows 0 .. 15, nobody uses them)
    auto desc_x = [&](uint32_t c, uint32_t J) -> uint32_t {
        if (J >= CT) { c += 1u; J -= CT; }
        const uint32_t x = dring[(c % 3u) * (2 * CT) + J];
        return tile_of(c, J) < t1 ? x : 0u;
    };
    auto fetch_words = [&](uint32_t c) {                                       // needs chunk c's descriptors in LDS
        if (!has_bitmap || (uint32_t) lane >= CT) return;
        const uint32_t x = desc_x(c, (uint32_t) lane);
        const uint32_t r0 = x <= last_row ? x : last_row;
        const gptr<uint32_t> src = (gptr<uint32_t>) (g_bitmap + (r0 >> 6));    // (bitmaps carry two pad words: the window never leaves them)
        uint32_t* dst = wring + (c & 1u) * (4 * CT);
        __builtin_amdgcn_global_load_lds(src, (lds_u32*) dst, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 1, (lds_u32*) dst + CT, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 2, (lds_u32*) dst + 2 * CT, 4, 0, 0);
        __builtin_amdgcn_global_load_lds(src + 3, (lds_u32*) dst + 3 * CT, 4, 0, 0);
    };

    // The loads of a stage: four 1-KB pieces (8 rows x 128 bytes: list tile 0 rows 0-7, 8-15, list tile 1 rows 0-7, 8-15) and one
    // 128-byte piece of |row|^2.  A list tile's 16 rows are consecutive, so a piece is (wave-uniform base of the tile) + (a
    // lane constant): the lane's row within the tile and its swizzled chunk.  Whole tiles are loaded whatever their row count
    // (rows past it are masked by their threshold; the planes and norms are padded so that a tile may start at the last row).
    uint32_t voff[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const uint32_t rit = (uint32_t) (h * 8) + ((uint32_t) lane >> 3);
        voff[h] = (rit * 8u + (((uint32_t) lane & 7u) ^ ((rit >> 1) & 7u))) * 16u;
    }
    const uint32_t noff = (uint32_t) lane & 15u;
    auto issue = [&](uint32_t i, uint32_t x0, uint32_t x1) {                   // x0, x1: wave-uniform first rows of the two tiles
        const uint32_t slot_ = i & (KI_SLOTS - 1);
        lds_u4* dst = (lds_u4*) (ring + (size_t) slot_ * KI_STAGE_U4);
        const char* b0 = reinterpret_cast<const char*>(p.scr) + (size_t) x0 * 128u;
        const char* b1 = reinterpret_cast<const char*>(p.scr) + (size_t) x1 * 128u;
        __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const uint4*>(b0 + voff[0])), dst, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const uint4*>(b0 + voff[1])), dst + 64, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const uint4*>(b1 + voff[0])), dst + 128, 16, 0, 0);
        __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const uint4*>(b1 + voff[1])), dst + 192, 16, 0, 0);
        if (lane < KI_ROWS) {
            const uint32_t row = (lane < 16 ? x0 : x1) + noff;
            __builtin_amdgcn_global_load_lds((gptr<uint32_t>) (g_norm2 + row), (lds_u32*) (nring + slot_ * KI_ROWS), 4, 0, 0);
        }
    };

    // Parked candidates -> their queries' buffers: one LDS atomic per candidate ranks it within its column, ONE returning
    // global atomic per column reserves room, then the keys are stored.  The returning atomic sits in the same in-order queue
    // as the stage loads, so waiting for it drains the three stages in flight: the flush is synchronous but RARE -- a wave
    // parks ~1.5 candidates per stage and flushes when half the parking area is used (every ~20 stages).
    uint32_t n_park = 0;                                                        // wave-uniform
    auto flush = [&]() {
#pragma unroll
        for (int c = 0; c < NC; c += 64) pk_n[c + lane] = 0u;
        wave_fence();
        uint32_t rank_in_col = 0;
        if ((uint32_t) lane < n_park) rank_in_col = atomicAdd(&pk_n[pk_c[lane]], 1u);      // LDS
        wave_fence();
#pragma unroll
        for (int c = 0; c < NC; c += 64) {                                      // lane: columns lane, lane + 64
            const uint32_t mine = pk_n[c + lane];
            if (mine) pk_n[c + lane] = atomicAdd(p.qcnt + __float_as_uint(colc[c + lane].z), mine);
        }
        wave_fence();
        if ((uint32_t) lane < n_park) {
            const uint32_t c = pk_c[lane];
            const uint32_t at = pk_n[c] + rank_in_col;
            const uint32_t slot = __float_as_uint(colc[c].z);
            const uint32_t row = pk_r[lane];
            if (at < p.capq) p.qcand[(size_t) slot * p.capq + at] = ((uint64_t) pk_v[lane] << 32) | (g_rank ? g_rank[row] : row);
        }
        n_park = 0;
        wave_fence();
    };

    // ---- mapping state of the current chunk, one list tile per lane (L < CT) ----
    uint32_t cx = 0;       // compute side: first row of tile L
    uint32_t cm = 0;       //               its validity mask: bit o = row o exists and is permitted
    uint32_t ix = 0;       // issue side: first row of tile L + 6 (the stage three ahead of tile L's)
    bool bad_row = false;
    auto enter_chunk = [&](uint32_t c) {                                       // chunk c's and c + 1's descriptors, chunk c's words are in LDS
        const uint32_t L = (uint32_t) lane & (CT - 1u);
        const uint32_t* d = dring + (c % 3u) * (2 * CT);
        const bool tile_ok = tile_of(c, L) < t1;
        const uint32_t x = tile_ok ? d[L] : 0u;
        const uint32_t y = tile_ok ? d[CT + L] : 0u;
        uint32_t m = y >= 16u ? 0xFFFFu : (1u << y) - 1u;
        if (y && x + y - 1u > last_row) {                                      // cannot happen: reported once at the end
            bad_row = true;
            m = 0;
        }
        if (has_bitmap) {
            const uint32_t* w = wring + (c & 1u) * (4 * CT) + L;
            const uint32_t r0 = x <= last_row ? x : last_row;
            const uint32_t b0 = r0 & 63u;                                       // the window starts at the word that holds row x
            const uint32_t lo = w[(b0 >> 5) * CT], hi = w[((b0 >> 5) + 1u) * CT];
            m &= (uint32_t) ((((uint64_t) hi << 32) | lo) >> (b0 & 31u));
        }
        cx = x;
        cm = m;
        ix = desc_x(c, L + 6u);
    };

    // ---- prologue: descriptors of chunks 0 and 1, permission words of chunk 0, three stages in flight ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    fetch_words(0);
    {
        const uint32_t px = desc_x(0, (uint32_t) lane & (CT - 1u));
#pragma unroll
        for (int s3 = 0; s3 < 3; ++s3)
            issue((uint32_t) s3, (uint32_t) __builtin_amdgcn_readlane((int) px, 2 * s3), (uint32_t) __builtin_amdgcn_readlane((int) px, 2 * s3 + 1));
    }

    int64_t smin[NQG];                                                          // SAMPLE: per column of this lane, (value, row)
#pragma unroll
    for (int j = 0; j < NQG; ++j) smin[j] = ((int64_t) 0x3FFFFFFF << 32);
    for (uint32_t i = 0; i < n_w; ++i) {
        const uint32_t ph = i % KI_CHUNK;
        // stage i has landed when at most the ten operations of stages i + 1 and i + 2 are outstanding (anything else in the
        // queue -- a flush's stores -- only makes this wait for a little more than it needs); right after a chunk boundary
        // that chunk's mapping loads sit in between and are counted too
        if (ph == 1 || ph == 2) {
            if (has_bitmap) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        if (ph == 0) {
            const uint32_t c = i / KI_CHUNK;
            enter_chunk(c);
            fetch_words(c + 1);
            fetch_desc(c + 2);
        }
        const int e = (int) (ph * 2u);
        issue(i + 3, (uint32_t) __builtin_amdgcn_readlane((int) ix, e), (uint32_t) __builtin_amdgcn_readlane((int) ix, e + 1));

        // ---- thresholds of this stage's rows: floor(|row|^2 / 2), INT_MAX for a row that does not exist or is not permitted ----
        const uint32_t slot_ = i & (KI_SLOTS - 1);
        const float* nrm = nring + slot_ * KI_ROWS;
        const uint32_t m32 = (uint32_t) __builtin_amdgcn_readlane((int) cm, e) | ((uint32_t) __builtin_amdgcn_readlane((int) cm, e + 1) << 16);
        if (lane < KI_ROWS) {
            if constexpr (SAMPLE) thr[lane] = (m32 >> lane) & 1u ? (int32_t) nrm[lane] : 0x3FFFFFFF;     // |row|^2 itself; no row: never the minimum
            else thr[lane] = (m32 >> lane) & 1u ? ((int32_t) nrm[lane]) >> 1 : 0x7FFFFFFF;
        }

        // ---- 2 row blocks x NQG query groups x 2 k-steps ----
        i32x4 acc[2][NQG];
        const uint4* img = ring + (size_t) slot_ * KI_STAGE_U4;
        i32x4 a8[2][2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = rb * 16 + li;
                a8[rb][ks] = __builtin_bit_cast(i32x4, img[row * 8 + ((ks * 4 + kq) ^ ((row >> 1) & 7))]);
            }
#pragma unroll
        for (int j = 0; j < NQG; ++j) {
            if ((uint32_t) j < ngt) {
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    acc[rb][j] = i32x4{c0j[j], c0j[j], c0j[j], c0j[j]};
                    acc[rb][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a8[rb][0], b8[j][0], acc[rb][j], 0, 0, 0);
                    acc[rb][j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a8[rb][1], b8[j][1], acc[rb][j], 0, 0, 0);
                }
            }
        }

        // ---- epilogue: acc[rb][j][r] = row rb * 16 + kq * 4 + r of the stage, query column j * 16 + li ----
        if constexpr (SAMPLE) {
            // smallest (|x'|^2 - 2 x'.q', row) of this lane's pairs of every column, as one signed 64-bit compare
#pragma unroll
            for (int rb = 0; rb < 2; ++rb) {
                const i32x4 nx = __builtin_bit_cast(i32x4, reinterpret_cast<const uint4*>(thr)[rb * 4 + kq]);
                const uint32_t x_rb = (uint32_t) __builtin_amdgcn_readlane((int) cx, e + rb);
#pragma unroll
                for (int j = 0; j < NQG; ++j)
                    if ((uint32_t) j < ngt) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int32_t w = nx[r] - 2 * acc[rb][j][r];          // (no row: 0x3FFFFFFF - 2 dot, above every real value)
                            const int64_t cand = ((int64_t) w << 32) | (int64_t) (x_rb + (uint32_t) (kq * 4 + r));
                            smin[j] = cand < smin[j] ? cand : smin[j];
                        }
                    }
            }
            continue;
        }
#pragma unroll
        for (int rb = 0; rb < 2; ++rb) {
            const i32x4 th = __builtin_bit_cast(i32x4, reinterpret_cast<const uint4*>(thr)[rb * 4 + kq]);
            uint64_t m[NQG][4];
            uint64_t any = 0;
#pragma unroll
            for (int j = 0; j < NQG; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) m[j][r] = 0;
#pragma unroll
            for (int j = 0; j < NQG; ++j)
                if ((uint32_t) j < ngt) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        m[j][r] = __ballot(acc[rb][j][r] >= th[r]);
                        any |= m[j][r];
                    }
                }
            if (any) {
                const uint32_t x_rb = (uint32_t) __builtin_amdgcn_readlane((int) cx, e + rb);
#pragma unroll
                for (int j = 0; j < NQG; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const uint64_t cmk = m[j][r];
                        if (!cmk) continue;                                     // scalar test
                        const bool has = (cmk >> lane) & 1ull;
                        const uint32_t o = (uint32_t) (kq * 4 + r);
                        const uint32_t row = x_rb + o;
                        const f32x4 cc = __builtin_bit_cast(f32x4, reinterpret_cast<const uint4*>(colc)[j * 16 + li]);
                        const f32x4 nx4 = __builtin_bit_cast(f32x4, reinterpret_cast<const uint4*>(nrm)[rb * 4 + kq]);
                        const float v = screen_value<M_L2>((float) (acc[rb][j][r] - c0j[j]), nx4[r], cc[1]);
                        const uint32_t at = n_park + __builtin_amdgcn_mbcnt_hi((uint32_t) (cmk >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t) cmk, 0u));
                        if (has) {
                            if (at < (uint32_t) KI_PARK) {
                                pk_v[at] = mono_bits(v);
                                pk_r[at] = row;
                                pk_c[at] = (uint32_t) (j * 16 + li);
                            } else {                                            // a burst beyond the parking area: directly
                                const uint32_t slot = __float_as_uint(cc[2]);
                                const uint32_t ga = atomicAdd(p.qcnt + slot, 1u);
                                if (ga < p.capq)
                                    p.qcand[(size_t) slot * p.capq + ga] = ((uint64_t) mono_bits(v) << 32) | (g_rank ? g_rank[row] : row);
                            }
                        }
                        n_park += (uint32_t) __popcll(cmk);
                        if (n_park > (uint32_t) KI_PARK) n_park = KI_PARK;
                    }
            }
        }
        if (n_park >= (uint32_t) KI_FLUSH_AT) flush();
    }
    if constexpr (SAMPLE) {
        // The stream is over.  Every lane's minima go to its wave's (now idle) stage ring; the LAST wave of the workgroup to
        // get here appends all of them: one returning atomic per query column and workgroup, like K2w's sample pass
        // (atomics on one address serialise at ~0.2 us each: per-lane appends made this launch 8 x longer than K2w's).
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the ring is about to be reused
        uint64_t* mine = reinterpret_cast<uint64_t*>(ring);                     // [NC][4]: column, row quad
#pragma unroll
        for (int j = 0; j < NQG; ++j) {
            const int32_t w = (int32_t) (smin[j] >> 32);
            uint64_t key = KEY_EMPTY;
            if ((uint32_t) j < ngt && w < 0x20000000) {
                const f32x4 cc = __builtin_bit_cast(f32x4, reinterpret_cast<const uint4*>(colc)[j * 16 + li]);
                const uint32_t row = (uint32_t) smin[j];
                if (__float_as_uint(cc[2]) != 0xFFFFFFFFu) key = make_key((float) w + cc[1], g_rank ? g_rank[row] : row);   // integers below 2^24: the fp32 distance
            }
            mine[(j * 16 + li) * 4 + kq] = key;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        uint32_t arrived = 0;
        if (lane == 0) arrived = atomicAdd(s_done, 1u);
        arrived = (uint32_t) __shfl((int) arrived, 0);
        const uint32_t active = n_st < (uint32_t) KI_WAVES ? n_st : (uint32_t) KI_WAVES;
        if (arrived + 1u == active) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#pragma unroll
            for (int c0 = 0; c0 < NC; c0 += 64) {
                const int c = c0 + lane;
                const uint32_t slot = __float_as_uint(colc[c].z);
                uint64_t keys16[16];
                uint32_t n = 0;
#pragma unroll
                for (int w2 = 0; w2 < KI_WAVES; ++w2) {
                    const uint64_t* other = reinterpret_cast<const uint64_t*>(smem + (size_t) w2 * KI_WAVE_BYTES);
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const uint64_t kk = (uint32_t) w2 < active ? other[c * 4 + q4] : KEY_EMPTY;
                        keys16[w2 * 4 + q4] = kk;
                        n += kk != KEY_EMPTY;
                    }
                }
                if (n && slot != 0xFFFFFFFFu) {
                    uint32_t at = atomicAdd(p.qcnt + slot, n);
#pragma unroll
                    for (int t = 0; t < 16; ++t)
                        if (keys16[t] != KEY_EMPTY) {
                            if (at < p.capq) p.qcand[(size_t) slot * p.capq + at] = keys16[t];
                            ++at;
                        }
                }
            }
        }
    } else if (n_park)
        flush();
    if (bad_row) atomicOr(p.err, 1u);                                           // a tile reached past the corpus: results invalid
    // the stages still in flight write LDS: they must have landed before the workgroup gives its LDS back
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

inline hipError_t launch_i8_stream(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    if (p.plane_ho != 2 || p.pstride4 != 8 || p.rw != 16 || p.qmax > (uint32_t) KI_NQ || !p.ones) return hipErrorInvalidValue;
    const size_t lds = i8s_lds_bytes();
    auto launch = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(KI_THREADS), lds, s, p);
        return hipGetLastError();
    };
    if (p.sample_stride > 1) return launch(i8_stream_kernel<4, true>);         // (sample groups never exceed 64 columns)
    return p.qmax > 64 ? launch(i8_stream_kernel<8>) : launch(i8_stream_kernel<4>);
}

}  // namespace vsr
