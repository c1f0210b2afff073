// vsr_mfmaw.h — K2w: GEMM-shaped shared-pass scan on the matrix cores for rows of 61 .. 192 floats.
//
// A permission class (or any filter part) that many queries see is a GEMM: dot[row][query], rows = the class's rows,
// queries = everyone whose role sees the class, K = d.  K2 (vsr_mfma.h) gave every WAVE its own row tile and 16 (32)
// queries, so a class seen by 330 queries was streamed 21 times.  K2w stages a 64-row tile ONCE per WORKGROUP in LDS
// and multiplies it against up to 64 * NGW queries whose B fragments live in the registers of the four waves:
//
//   wave w owns query group(s) {w, w + 4} (16 queries each) and reads the whole shared tile        (>= 3 groups)
//   2 groups: two waves per group, two 16-row sub-tiles each; 1 group: four waves, one sub-tile each (row split),
//
// so narrow passes (a leaf class seen by 10 queries) still use all four waves and stay HBM-bound, while fat passes run
// at the matrix pipe's pace with the rows read once per <= 128 queries.
//
// Pipeline (register staging, MI355X guide T14 / G15): a thread owns 4 float4 of every 64-row x 64-float stage; the
// loads of tile i+1 are issued as soon as the registers of tile i have been written to LDS (a whole tile = NSTR stages
// = 32 KB per workgroup at d = 128 stays in flight under the MFMAs of tile i), two LDS stage buffers, one barrier per
// stage.  The row mapping of a tile (tile descriptor -> row -> permission bit, |row|^2) is resolved by wave 0 three
// tiles ahead, one dependent load per iteration, so no wave ever waits for it.
//
// Arithmetic, screening keys, candidate buffers, compaction votes and the final per-query radix selection are those of
// K2 (vsr_mfma.h); K5r re-ranks the survivors exactly.
#pragma once
#include <type_traits>
#include "vsr_device.h"
#include "vsr_scan.h"
#include "vsr_topk.h"
#include "vsr_mfma.h"

namespace vsr {

constexpr int MW_THREADS = 256;
constexpr int MW_WAVES = 4;
constexpr int MW_S = 16;                   // float4 chunks per stage (64 floats)
constexpr int MW_ROWS = 64;                // rows per workgroup tile

#ifndef VSR_MW_OCC
#define VSR_MW_OCC 2                       // workgroups per CU the register allocation aims at (one query group per wave)
#endif

template <int METRIC, int NSTR, bool SAMPLE, int NGW>
__global__ __launch_bounds__(MW_THREADS, NGW == 1 && NSTR <= 2 ? VSR_MW_OCC : 2) void mfma_wide_kernel(const ScanParams p)
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

    const uint32_t pstride4 = p.pstride4, cap = p.cap, keep = p.k;
    const uint32_t q_count = grp.q_count;
    constexpr int NQ = MF_NQ * MW_WAVES * NGW;                                 // query slots of a pass (64 or 128)

    // LDS: [stage buffers | row index ring | |row|^2 ring | top-k control | |q|^2 | vote flags]
    const uint32_t stage_bytes = mfmaw_stage_bytes(cap);
    uint4*    stage = reinterpret_cast<uint4*>(smem);                           // [2][64 * MW_S]
    unsigned char* after = smem + stage_bytes;
    int32_t*  rowidx = reinterpret_cast<int32_t*>(after);                       // [4][64]
    float*    rownorm = reinterpret_cast<float*>(after + 4 * 64 * 4);           // [4][64]
    TopKCtrl* ctrl = reinterpret_cast<TopKCtrl*>(after + 4 * 64 * 8);
    float*    qnl = reinterpret_cast<float*>(ctrl + NQ);
    uint32_t* flags = reinterpret_cast<uint32_t*>(qnl + NQ);
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem);

    for (uint32_t qi = tid; qi < (uint32_t) NQ; qi += MW_THREADS) {
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        ctrl[qi].tau = p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        ctrl[qi].count = 0;
        qnl[qi] = p.q_norm2[slot];
    }
    if (tid < 4) flags[tid] = 0;

    // ---- wave roles ----
    const uint32_t ngt = (q_count + MF_NQ - 1) / MF_NQ;                         // 16-query groups of this pass
    const uint32_t rsplit = ngt == 1 ? 4u : ngt == 2 ? 2u : 1u;                 // waves sharing one group's rows
    const uint32_t g0 = (uint32_t) wave / rsplit;                               // this wave's (first) query group
    const uint32_t sub0 = ((uint32_t) wave % rsplit) * (4u / rsplit);           // its first 16-row sub-tile
    bool gact[NGW];
#pragma unroll
    for (int g = 0; g < NGW; ++g) gact[g] = g0 + (uint32_t) g * MW_WAVES < ngt;

    // MFMA lane roles: A operand lane = (row i, k-quad kq); B operand / result lane = (k-quad kq, query jq)
    const int li = lane & 15;
    const int kq = lane >> 4;
    const int jq = li;
    bf16x8 bh[NGW][NSTR][2], bm[NGW][NSTR][2];                                  // B fragments: hi / mid, 2 K-blocks of 32 per stage
    float my_qn[NGW];
    uint32_t my_qi[NGW];
#pragma unroll
    for (int g = 0; g < NGW; ++g) {
        const uint32_t qi = (g0 + (uint32_t) g * MW_WAVES) * MF_NQ + (uint32_t) jq;
        my_qi[g] = qi < (uint32_t) NQ ? qi : 0u;
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];   // pad columns repeat query 0
        my_qn[g] = p.q_norm2[slot];
        const uint4* qsrc = p.q_scr + (size_t) slot * pstride4;                    // planes are zero padded to whole stages
#pragma unroll
        for (int s = 0; s < NSTR; ++s)
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                bh[g][s][blk] = __builtin_bit_cast(bf16x8, qsrc[s * MW_S + blk * 4 + kq]);
                bm[g][s][blk] = __builtin_bit_cast(bf16x8, qsrc[s * MW_S + 8 + blk * 4 + kq]);
            }
    }

    // ---- this workgroup's tiles ----
    const uint32_t rw = p.rw, tps = MW_ROWS / rw;                               // list tiles per 64-row workgroup tile
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t n_super = (t1 - t0 + tps - 1) / tps;
    const uint32_t ss = p.sample_stride;                                        // sample pass: every ss-th tile
    const uint32_t n_it = (n_super + ss - 1) / ss;
    const uint32_t slack = mfmaw_slack();
    const uint32_t trigger = cap - slack;
    uint64_t* cand = p.cand + (size_t) (grp.partial_begin + local_block) * cand_pitch(cap);
    const size_t cand_qstride = (size_t) grp.n_blocks * cand_pitch(cap);

    // ---- row mapping pipeline (wave 0, lane = row slot): descriptor at it+3, row + bitmap word + norm at it+2,
    // written to the LDS ring at it+1, used by the loads issued during tile it and by the epilogue of tile it+1 ----
    auto fetch_desc = [&](uint32_t it_) -> uint2 {                 // (start, nrows) of this lane's list tile, or (0, 0)
        if (it_ >= n_it) return make_uint2(0u, 0u);
        const uint32_t t = t0 + it_ * ss * tps + (uint32_t) lane / rw;
        if (t >= t1) return make_uint2(0u, 0u);
        if (t >= grp.n_tiles) {                                    // cannot happen; never read past the tile list
            atomicOr(p.err, 4u);
            return make_uint2(0u, 0u);
        }
        if (grp.tiles) return grp.tiles[t];
        const uint32_t start = t * rw;
        return make_uint2(start, p.n_rows - start < rw ? p.n_rows - start : rw);
    };
    int32_t  pend_row = -1;                                        // tile it+1: row (before the permission bit)
    uint64_t pend_bw = ~0ull;                                      //            its bitmap word (in flight)
    float    pend_nrm = 0.0f;                                      //            its |row|^2 (in flight)
    uint2    dsc_a = make_uint2(0u, 0u);                           // tile it+2: descriptor (in flight)
    auto start_rows = [&](uint2 d, int32_t& row, uint64_t& bw, float& nrm) {
        const uint32_t r = (uint32_t) lane % rw;
        row = -1;
        bw = ~0ull;
        nrm = 0.0f;
        if (r < d.y) {
            const uint32_t rr = d.x + r;
            if (rr >= p.n_rows) {                                  // cannot happen; never read past the corpus
                atomicOr(p.err, 1u);
            } else {
                row = (int32_t) rr;
                if (grp.bitmap) bw = grp.bitmap[rr >> 6];
                nrm = p.norm2[rr];
            }
        }
    };
    auto finish_rows = [&](uint32_t it_, int32_t row, uint64_t bw, float nrm) {
        if (row >= 0 && !((bw >> ((uint32_t) row & 63u)) & 1ull)) row = -1;
        rowidx[(it_ & 3u) * 64 + lane] = row;
        rownorm[(it_ & 3u) * 64 + lane] = row >= 0 ? nrm : 0.0f;
    };
    if (wave == 0) {
        const uint2 d0 = fetch_desc(0), d1 = fetch_desc(1);
        dsc_a = fetch_desc(2);
        int32_t r0;
        uint64_t b0;
        float n0;
        start_rows(d0, r0, b0, n0);
        start_rows(d1, pend_row, pend_bw, pend_nrm);
        finish_rows(0, r0, b0, n0);
    }
    __syncthreads();

    // ---- staging: thread -> (row slot u * 16 + lrow, chunk lchunk) of every stage ----
    const int lrow = tid >> 4, lchunk = tid & 15;
    uint4 X[NSTR][4];
    const uint32_t last_row = p.n_rows - 1u;
    auto issue = [&](auto sc, uint32_t it_) {                      // loads of tile it_, stage S into X[S] (no waits)
        // An invalid slot (masked row, ragged tile) loads row 0 and its products are discarded by the epilogue's row
        // test (the runtime keeps corpora with NaN / Inf elements off the screening kernels).  Plane rows are zero
        // padded to whole stages, so every chunk index is in range.
        constexpr int S = decltype(sc)::value;
        const uint32_t chunk = (uint32_t) (S * MW_S + lchunk);
        const int32_t* ridx = rowidx + (it_ & 3u) * 64;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int32_t r = ridx[u * 16 + lrow];
            const uint32_t rc = (uint32_t) (r < 0 ? 0 : r);
            const u32x4 v = *reinterpret_cast<const u32x4*>(p.scr + (size_t) (rc < last_row ? rc : last_row) * pstride4 + chunk);
            X[S][u] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    auto issue_all = [&](uint32_t it_) {
        issue(std::integral_constant<int, 0>{}, it_);
        if constexpr (NSTR > 1) issue(std::integral_constant<int, 1>{}, it_);
        if constexpr (NSTR > 2) issue(std::integral_constant<int, 2>{}, it_);
    };
    if (n_it > 0) issue_all(0);

    uint32_t round = 0;
    auto run = [&](auto nsc) {
        constexpr int NS = decltype(nsc)::value;                   // 16-row sub-tiles of this wave (1, 2 or 4)
        int buf = 0;
        for (uint32_t it = 0; it < n_it; ++it) {
            f32x4 acc[NGW][NS];
#pragma unroll
            for (int g = 0; g < NGW; ++g)
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[g][i] = f32x4{0.f, 0.f, 0.f, 0.f};

            auto do_stage = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                uint4* img = stage + (size_t) buf * (MW_ROWS * MW_S);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int slot = u * 16 + lrow;
                    img[slot * MW_S + (lchunk ^ (slot & 15))] = X[S][u];
                }
                if (S == 0 && wave == 0) {                         // row mapping, one step per tile (see above)
                    finish_rows(it + 1, pend_row, pend_bw, pend_nrm);
                    start_rows(dsc_a, pend_row, pend_bw, pend_nrm);
                    dsc_a = fetch_desc(it + 3);
                }
                __syncthreads();
                if (it + 1 < n_it) issue(sc, it + 1);              // in flight under the MFMAs of a whole tile
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {                // two K-blocks of 32 per stage
                    bf16x8 ah[NS], am[NS];
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const int row = (int) sub0 * 16 + i * 16 + li;
                        ah[i] = __builtin_bit_cast(bf16x8, img[row * MW_S + ((blk * 4 + kq) ^ li)]);
                        am[i] = __builtin_bit_cast(bf16x8, img[row * MW_S + ((8 + blk * 4 + kq) ^ li)]);
                    }
#pragma unroll
                    for (int g = 0; g < NGW; ++g) {
                        if (!gact[g]) continue;                    // wave-uniform
                        // x q ~ xh qh + xh qm + xm qh  (the dropped xm qm and the split residues are inside plane_err_g)
#pragma unroll
                        for (int i = 0; i < NS; ++i) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bh[g][S][blk], acc[g][i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NS; ++i) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah[i], bm[g][S][blk], acc[g][i], 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NS; ++i) acc[g][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am[i], bh[g][S][blk], acc[g][i], 0, 0, 0);
                    }
                }
                buf ^= 1;
            };
            do_stage(std::integral_constant<int, 0>{});
            if constexpr (NSTR > 1) do_stage(std::integral_constant<int, 1>{});
            if constexpr (NSTR > 2) do_stage(std::integral_constant<int, 2>{});

            // results: acc[g][i][r] = dot(row slot (sub0 + i) * 16 + kq * 4 + r, query group g's column jq).  Every lane
            // screens its pairs, reserves room for all of its survivors with ONE LDS atomic and stores them.
            const int32_t* ridx = rowidx + (it & 3u) * 64;
            const float* rnrm = rownorm + (it & 3u) * 64;
#pragma unroll
            for (int g = 0; g < NGW; ++g) {
                if (!gact[g]) continue;                            // wave-uniform
                const uint32_t qi = my_qi[g];
                const uint64_t tau = *reinterpret_cast<volatile uint64_t*>(&ctrl[qi].tau);
                const bool qok = qi < q_count;
                // screening test in float: a value is a candidate unless it is greater than the threshold's distance
                // (NaN values and an open / NaN threshold pass): a superset of `key < tau`, never a missing candidate
                const bool open = tau == KEY_EMPTY;
                const float tau_f = mono_to_float((uint32_t) (tau >> 32));
                float vv[NS * 4];
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
                        const float v = screen_value<METRIC>(acc[g][i][r], nx4[r], my_qn[g]);
                        vv[i * 4 + r] = v;
                        if (qok && rows4[r] >= 0 && (open || !(v > tau_f))) pmask |= 1u << (i * 4 + r);
                    }
                }
                if (__ballot(pmask != 0) != 0) {                   // wave-uniform
                    uint32_t base = 0;
                    if (pmask) base = atomicAdd(&ctrl[qi].count, (uint32_t) __popc(pmask));
                    if (pmask && base + (uint32_t) __popc(pmask) > cap) {      // cannot happen (append slack protocol)
                        atomicOr(p.err, 2u);
                        pmask = 0;
                    }
                    uint64_t* dst = cand + (size_t) qi * cand_qstride + base;
#pragma unroll
                    for (int j = 0; j < NS * 4; ++j)
                        if (pmask & (1u << j)) {                   // the key is built for survivors only
                            const int32_t row = ridx[((int) sub0 + (j >> 2)) * 16 + kq * 4 + (j & 3)];
                            dst[__popc(pmask & ((1u << j) - 1u))] = make_key(vv[j], (uint32_t) row);
                        }
                }
            }

            if (it + 1 < n_it && (it + 1) % MW_VOTE == 0) {        // compaction vote (between two votes a query gains
                bool need = false;                                 // at most 64 * MW_VOTE keys: the buffers' slack)
                for (uint32_t q = (uint32_t) tid; q < q_count; q += MW_THREADS)
                    need |= *reinterpret_cast<volatile uint32_t*>(&ctrl[q].count) > trigger;
                const uint32_t fslot = round % 3;
                if (need) atomicOr(&flags[fslot], 1u);
                __syncthreads();                                   // also: every wave is done with the stage buffers
                const bool any = *reinterpret_cast<volatile uint32_t*>(&flags[fslot]) != 0;
                if (tid == 0) flags[(round + 2) % 3] = 0;
                ++round;
                if (any) {
                    for (uint32_t q = 0; q < q_count; ++q) {
                        const uint32_t n = ctrl[q].count < cap ? ctrl[q].count : cap;
                        if (n > trigger) {                         // only the buffers that are filling up
                            uint64_t* cq = cand + (size_t) q * cand_qstride;
                            for (uint32_t i = tid; i < n; i += MW_THREADS) sortbuf[i] = cq[i];
                            __syncthreads();
                            topk_compact<MW_THREADS>(sortbuf, &ctrl[q], keep, tid, false);
                            for (uint32_t i = tid; i < keep; i += MW_THREADS) cq[i] = sortbuf[i];
                            __syncthreads();
                        }
                    }
                }
            }
        }
    };
    if (rsplit == 4) run(std::integral_constant<int, 1>{});
    else if (rsplit == 2) run(std::integral_constant<int, 2>{});
    else run(std::integral_constant<int, 4>{});

    __syncthreads();
    constexpr int PR = 32;                                                     // candidate keys per lane at publish
    if (cap <= (uint32_t) (64 * PR)) {
        // publish, one wave per query: the candidates of a (workgroup, query) buffer go to registers and the `keep`
        // smallest are picked by a radix select (vsr_topk.h).  The partial list is unordered; K5 selects again.
        uint32_t* hist = reinterpret_cast<uint32_t*>(smem) + wave * 256;       // wave-private, the images are dead by now
        for (uint32_t q = (uint32_t) wave; q < q_count; q += MW_WAVES) {
            const uint32_t n = ctrl[q].count < cap ? ctrl[q].count : cap;
            const uint64_t* cq = cand + (size_t) q * cand_qstride;
            uint64_t* dst = p.partial + (size_t) (grp.partial_begin + q * grp.n_blocks + local_block) * p.kp;
            if (n <= keep) {                                                   // nothing to drop
                for (uint32_t i = (uint32_t) lane; i < p.kp; i += 64) dst[i] = i < n ? cq[i] : KEY_EMPTY;
                continue;
            }
            auto pick = [&](auto rc) {                                         // RR keys per lane cover the n candidates
                constexpr int RR = decltype(rc)::value;
                uint64_t reg[RR];
#pragma unroll
                for (int r = 0; r < RR; ++r) {
                    const uint32_t i = (uint32_t) (r * 64 + lane);
                    reg[r] = cq[i < n ? i : 0u];
                }
#pragma unroll
                for (int r = 0; r < RR; ++r)
                    if ((uint32_t) (r * 64 + lane) >= n) reg[r] = KEY_EMPTY;
                uint64_t tsel, kth;
                wave_radix_select<RR>(reg, n, keep, hist, lane, tsel, kth);
                const uint32_t want = wave_emit_selected<RR>(reg, n, keep, tsel, kth, dst, lane);
                for (uint32_t i = want + (uint32_t) lane; i < p.kp; i += 64) dst[i] = KEY_EMPTY;
            };
            if (n <= 256) pick(std::integral_constant<int, 4>{});
            else if (n <= 512) pick(std::integral_constant<int, 8>{});
            else if (n <= 1024) pick(std::integral_constant<int, 16>{});
            else pick(std::integral_constant<int, PR>{});
        }
        return;
    }
    for (uint32_t q = 0; q < q_count; ++q) {
        const uint32_t n = ctrl[q].count < cap ? ctrl[q].count : cap;
        const uint64_t* cq = cand + (size_t) q * cand_qstride;
        for (uint32_t i = tid; i < n; i += MW_THREADS) sortbuf[i] = cq[i];
        __syncthreads();
        topk_compact<MW_THREADS>(sortbuf, &ctrl[q], keep, tid, false);
        const uint32_t m = ctrl[q].count < keep ? ctrl[q].count : keep;
        uint64_t* dst = p.partial + (size_t) (grp.partial_begin + q * grp.n_blocks + local_block) * p.kp;
        for (uint32_t i = tid; i < p.kp; i += MW_THREADS) dst[i] = i < m ? sortbuf[i] : KEY_EMPTY;
        __syncthreads();
    }
}

template <int METRIC>
hipError_t launch_mfmaw_metric(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    const uint32_t nstage = p.pstride4 / MW_S;
    const int ngw = p.qmax > (uint32_t) (MF_NQ * MW_WAVES) ? 2 : 1;
    const size_t lds = mfmaw_lds_bytes(p.cap, ngw);
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
    auto pick = [&](auto nstr) -> hipError_t {
        constexpr int N = decltype(nstr)::value;
        if constexpr (N <= 2)                                                  // two query groups per wave: B fragments fit for d <= 128
            if (ngw == 2) return sample ? launch(mfma_wide_kernel<METRIC, N, true, 2>) : launch(mfma_wide_kernel<METRIC, N, false, 2>);
        if (ngw == 2) return hipErrorInvalidValue;
        return sample ? launch(mfma_wide_kernel<METRIC, N, true, 1>) : launch(mfma_wide_kernel<METRIC, N, false, 1>);
    };
    switch (nstage) {
    case 1: return pick(std::integral_constant<int, 1>{});
    case 2: return pick(std::integral_constant<int, 2>{});
    case 3: return pick(std::integral_constant<int, 3>{});
    default: return hipErrorInvalidValue;                                     // longer rows: K2 (vsr_mfma.h)
    }
}

}  // namespace vsr
