// vsr_mfma.h — K2: shared-pass scan on the matrix cores (fp32 MFMA screening) + exact re-rank (K5r).
//
// For passes that many queries share, the distance work is GEMM-shaped: dot[row][query] = sum_k x[row][k] * q[query][k].
// K2 computes the dot products with v_mfma_f32_16x16x4_f32 (exact fp32 FMA chain, one VGPR per operand; 16 rows x
// 16 queries per instruction group) and turns them into SCREENING keys
//     L2:  |x|^2 + |q|^2 - 2 dot        IP: -dot        cosine: 1 - dot * rsqrt(|x|^2 |q|^2)
// whose only job is to keep, per query, the kp (= 2k) best candidates.  The expansion form of L2 cancels for
// near-duplicates, so screening keys are never reported: K5r (rerank_kernel, vsr_kernels.hip) recomputes the
// exact operator arithmetic of vector.c:549-563 / :596-606 / :638-685 for the kp survivors, selects the final k and
// flags a query when the gap between the kept set and the rest is inside the fp32 error bound of the screening
// (then the caller re-runs that query on the exact K1 path).  On integer-valued data the expansion is exact.
//
// Data path per wave and 64-row tile: coalesced global loads (256-byte row segments) -> wave-private LDS image,
// XOR-swizzled so that the A-fragment reads (lane = (row i, k-quad kq) reads float4 [i][4t+kq]) are conflict-free
// -> 4 x 16-row sub-tiles x 4 k-steps of MFMA per float4.  The 16 (or 2 x 16) query vectors sit in registers as B
// fragments for the whole kernel when d <= 256 (32 VGPRs at d = 128); for longer rows the fragments of one 64-float
// stage at a time are streamed from global memory (L1 / L2 resident) one stage ahead of their use.
#pragma once
#include <type_traits>
#include "vsr_device.h"
#include "vsr_scan.h"
#include "vsr_topk.h"

namespace vsr {


constexpr int MF_THREADS = 256;
constexpr int MF_WAVES = 4;
constexpr int MF_S = 16;                  // float4 chunks per stage
constexpr int MF_NQ = 16;                 // query columns of one MFMA tile
constexpr uint32_t MF_SLACK = K2_SLACK;    // keys per query a workgroup can append between two votes
static_assert(K2_SLACK == MF_WAVES * 64 * K2_VOTE_EVERY, "slack covers one vote interval of every wave");

template <int METRIC>
__device__ __forceinline__ float screen_value(float dot, float nx, float nq)
{
    if constexpr (METRIC == M_L2) return fmaf(-2.0f, dot, nx + nq);
    else if constexpr (METRIC == M_IP) return -dot;
    else return 1.0f - dot * rsqrtf(nx * nq);
}

// NSTR > 0: B fragments of NSTR stages live in registers (d <= 64 * NSTR); NSTR == 0: the B fragments of one stage at a
// time are streamed from global memory (L1 / L2 resident: 3 KB per query at d = 768) one stage ahead of their use.
// SAMPLE only gives the seeding pass (p.sample_stride > 1) its own kernel symbol in profiles.
// NG: 16-query groups served by one pass (1 or 2): the staged tile is multiplied against NG sets of B fragments.
template <int METRIC, int NSTR, bool SAMPLE, int NG>
__global__ __launch_bounds__(MF_THREADS, 2) void mfma_scan_kernel(const ScanParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // workgroup -> (pass, block of the pass): through the planner's XCD-aware map (passes that read the same rows get
    // the same workgroup id modulo 8, i.e. one XCD and one L2, and neighbouring dispatch slots), else by block range
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
    const auto g_rank = as_global(p.rank);

    const uint32_t stride4 = p.stride4, cap = p.cap, keep = p.k;
    const uint32_t nstage = (stride4 + MF_S - 1) / MF_S;
    const uint32_t q_count = grp.q_count;

    float4*   stage = reinterpret_cast<float4*>(smem) + (size_t) wave * 64 * MF_S;
    unsigned char* after = smem + (size_t) MF_WAVES * 64 * MF_S * 16;
    int32_t*  rowidx = reinterpret_cast<int32_t*>(after) + wave * 128;             // [wave][2][64]
    float*    rownorm = reinterpret_cast<float*>(after + MF_WAVES * 128 * 4) + wave * 128;
    constexpr int NQ = MF_NQ * NG;
    TopKCtrl* ctrl = reinterpret_cast<TopKCtrl*>(after + MF_WAVES * 128 * 8);
    float*    qnl = reinterpret_cast<float*>(ctrl + NQ);
    uint32_t* flags = reinterpret_cast<uint32_t*>(qnl + NQ);
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem);

    for (uint32_t qi = tid; qi < (uint32_t) NQ; qi += MF_THREADS) {
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        ctrl[qi].tau = p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        ctrl[qi].count = 0;
        qnl[qi] = p.q_norm2[slot];
    }
    if (tid < 4) flags[tid] = 0;
    __syncthreads();

    // MFMA lane roles: A operand lane = (row i, k-quad kq); B operand / result lane = (k-quad kq, query jq)
    const int li = lane & 15;
    const int kq = lane >> 4;
    const int jq = li;
    constexpr int NB = NSTR > 0 ? NSTR : 1;
    float4 bq[NG][NB][4];
    float my_qn[NG];
    const float4* qsrc_g[NG];                                                  // this lane's query column per group
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        const uint32_t qi = (uint32_t) (g * MF_NQ + jq);
        my_qn[g] = qnl[qi];
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];   // pad columns repeat query 0
        const float4* qsrc = reinterpret_cast<const float4*>(p.queries) + (size_t) slot * stride4;
        qsrc_g[g] = qsrc;
        if constexpr (NSTR > 0) {                                              // B fragments straight from global
#pragma unroll
            for (int s = 0; s < NSTR; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint32_t idx = (uint32_t) (s * MF_S + 4 * t + kq);
                    bq[g][s][t] = idx < stride4 ? qsrc[idx] : make_float4(0.f, 0.f, 0.f, 0.f);
                }
        }
    }
    // NSTR == 0: B fragments of the stage in use (bcur) and of the next one (bnxt, raw loads: zeroed for chunks past
    // the row end only when they become bcur, so that nothing touches them while they are in flight)
    f32x4 bcur[NG][4], bnxt[NG][4];
    auto issue_b = [&](uint32_t s_) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint32_t idx = s_ * MF_S + (uint32_t) (4 * t + kq);
                bnxt[g][t] = *reinterpret_cast<const f32x4*>(qsrc_g[g] + (idx < stride4 ? idx : 0u));
            }
    };
    auto take_b = [&](uint32_t s_) {
#pragma unroll
        for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = s_ * MF_S + (uint32_t) (4 * t + kq) < stride4;
                bcur[g][t] = ok ? bnxt[g][t] : f32x4{0.f, 0.f, 0.f, 0.f};
            }
    };

    const uint32_t rw = p.rw, tps = 64 / rw;
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t n_super = (t1 - t0 + tps - 1) / tps;
    const uint32_t ss = p.sample_stride;                                       // sample pass: every ss-th super-tile
    const uint32_t iters = ((n_super + ss - 1) / ss + MF_WAVES - 1) / MF_WAVES;
    const uint32_t trigger = cap - MF_SLACK;
    uint64_t* cand = p.cand + (size_t) (grp.partial_begin + local_block) * cand_pitch(cap);
    const size_t cand_qstride = (size_t) grp.n_blocks * cand_pitch(cap);

    const int lps_row = lane / MF_S, lps_chunk = lane % MF_S;
    constexpr int RPI = 64 / MF_S;

    // the next tile's descriptor is fetched one tile ahead of its use
    auto fetch_desc = [&](uint32_t it_) -> uint2 {               // (start, nrows) of this lane's list tile, or (0, 0)
        const uint32_t sup = (it_ * MF_WAVES + wave) * ss;
        if (it_ >= iters || sup >= n_super) return make_uint2(0u, 0u);
        const uint32_t t = t0 + sup * tps + (uint32_t) lane / rw;
        if (t >= t1) return make_uint2(0u, 0u);
        if (t >= grp.n_tiles) {                                  // cannot happen; never read past the tile list
            atomicOr(p.err, 4u);
            return make_uint2(0u, 0u);
        }
        if (g_tiles) return load_tile(g_tiles, t);
        const uint32_t start = t * rw;
        return make_uint2(start, p.n_rows - start < rw ? p.n_rows - start : rw);
    };
    auto resolve = [&](uint2 d) -> int32_t {                     // this lane's corpus row of the tile, or -1
        const uint32_t r = (uint32_t) lane % rw;
        if (r >= d.y) return -1;
        const uint32_t row = d.x + r;
        if (row >= p.n_rows) {                                   // cannot happen; never read past the corpus
            atomicOr(p.err, 1u);
            return -1;
        }
        if (g_bitmap && !((g_bitmap[row >> 6] >> (row & 63)) & 1ull)) return -1;
        return (int32_t) row;
    };
    float4 x[MF_S];
    auto issue = [&](uint32_t s, const int32_t* ridx) {           // global loads of stage s into x (no waits)
        // Nothing may consume the loaded registers here: any VALU touch (a select that zeroes an invalid slot) makes
        // hipcc wait for the loads BEFORE the MFMAs of the previous stage, which serialises streaming and matrix work.
        // An invalid slot (masked row, ragged tile) loads row 0 instead and its products are discarded by the
        // `rows4[r] >= 0` test of the epilogue; the padding chunks of a ragged last stage load chunk 0 and meet the
        // zero padding of the B fragments (finite x 0: the runtime keeps corpora with NaN / Inf elements off K2).
        const uint32_t chunk = s * MF_S + lps_chunk;
        const uint32_t cchunk = chunk < stride4 ? chunk : 0u;
        int32_t worst = -1;
        const uint32_t last = p.n_rows - 1u;
#pragma unroll
        for (int u = 0; u < MF_S; ++u) {
            const int32_t r = ridx[u * RPI + lps_row];
            worst = r > worst ? r : worst;
            const uint32_t rc = (uint32_t) (r < 0 ? 0 : r);
            // (loaded as a native vector: a float4 struct copy from global memory keeps x[] out of registers)
            const f32x4 v = *reinterpret_cast<const f32x4*>(p.rows + (size_t) (rc < last ? rc : last) * stride4 + cchunk);
            x[u] = make_float4(v[0], v[1], v[2], v[3]);
        }
        if (worst >= 0 && (uint32_t) worst >= p.n_rows) atomicOr(p.err, 1u);   // cannot happen; never read past the corpus
    };

    int32_t myrow = -1;
    float myrn = 0.0f;
    bool have = false;
    uint2 desc0 = fetch_desc(0);
    uint32_t round = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        int32_t* ridx = rowidx;
        float* rnrm = rownorm;
        myrow = resolve(desc0);
        myrn = myrow >= 0 ? p.norm2[myrow] : 0.0f;
        have = __ballot(myrow >= 0) != 0;
        if (have) {
            ridx[lane] = myrow;
            issue(0, ridx);
        }
        desc0 = fetch_desc(it + 1);                              // next tile's descriptor rides under this tile's work
        if (have) {                                                            // wave-uniform
            rnrm[lane] = myrn;

            f32x4 acc[NG][4];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) acc[g][sub] = f32x4{0.f, 0.f, 0.f, 0.f};

            if constexpr (NSTR == 0) {
                issue_b(0);
                take_b(0);
            }
            for (uint32_t s = 0; s < nstage; ++s) {
#pragma unroll
                for (int u = 0; u < MF_S; ++u) {
                    const int row = u * RPI + lps_row;
                    stage[row * MF_S + (lps_chunk ^ (row & 15))] = x[u];       // swizzled image
                }
                if (s + 1 < nstage) issue(s + 1, ridx);                        // in flight during the MFMAs
                if constexpr (NSTR == 0)
                    if (s + 1 < nstage) issue_b(s + 1);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    float4 b[NG];
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        if constexpr (NSTR > 0) {
                            b[g] = bq[g][0][t];
#pragma unroll
                            for (int s2 = 1; s2 < NSTR; ++s2)
                                if (s == (uint32_t) s2) b[g] = bq[g][s2][t];
                        } else {
                            b[g] = make_float4(bcur[g][t][0], bcur[g][t][1], bcur[g][t][2], bcur[g][t][3]);
                        }
                    }
                    // A fragments of the four 16-row sub-tiles first, then the MFMAs component by component so that
                    // consecutive instructions hit different accumulators (a dependent 16x16x4 chain has 40 cycles of
                    // latency against a 32-cycle issue interval)
                    float4 a[4];
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub) a[sub] = stage[(sub * 16 + li) * MF_S + ((4 * t + kq) ^ li)];
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sub].x, b[g].x, acc[g][sub], 0, 0, 0);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sub].y, b[g].y, acc[g][sub], 0, 0, 0);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sub].z, b[g].z, acc[g][sub], 0, 0, 0);
#pragma unroll
                    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                        for (int g = 0; g < NG; ++g)
                            acc[g][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[sub].w, b[g].w, acc[g][sub], 0, 0, 0);
                }
                if constexpr (NSTR == 0)
                    if (s + 1 < nstage) take_b(s + 1);                         // landed under this stage's MFMAs
            }

            // results: acc[g][sub][r] = dot(row sub*16 + kq*4 + r, query g*16 + jq).  Every lane screens its 16 pairs per
            // group, then reserves room for all of its survivors with ONE LDS atomic and stores them: no atomic
            // round trip per pair.
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint32_t qi = (uint32_t) (g * MF_NQ + jq);
                const uint64_t tau = lds_peek(&ctrl[qi].tau);
                const bool qok = qi < q_count;
                // screening test in float: a value is a candidate unless it is greater than the threshold's distance
                // (NaN values and an open / NaN threshold pass).  That admits a superset of `key < tau` (ties of the
                // threshold distance): extra candidates are harmless, a missing one is not.  The 64-bit key is only
                // built for the few survivors.
                const bool open = tau == KEY_EMPTY;
                const float tau_f = mono_to_float((uint32_t) (tau >> 32));
                float vv[16];
                uint32_t pmask = 0;
#pragma unroll
                for (int sub = 0; sub < 4; ++sub) {
                    const int4 ri = *reinterpret_cast<const int4*>(&ridx[sub * 16 + kq * 4]);
                    const float4 rn = *reinterpret_cast<const float4*>(&rnrm[sub * 16 + kq * 4]);
                    const int32_t rows4[4] = {ri.x, ri.y, ri.z, ri.w};
                    const float nx4[4] = {rn.x, rn.y, rn.z, rn.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float v = screen_value<METRIC>(acc[g][sub][r], nx4[r], my_qn[g]);
                        vv[sub * 4 + r] = v;
                        if (qok && rows4[r] >= 0 && (open || !(v > tau_f))) pmask |= 1u << (sub * 4 + r);
                    }
                }
                if (__ballot(pmask != 0) != 0) {                               // wave-uniform
                    uint32_t base = 0;
                    if (pmask) base = atomicAdd(&ctrl[qi].count, (uint32_t) __popc(pmask));
                    if (pmask && base + (uint32_t) __popc(pmask) > cap) {      // cannot happen (append slack protocol)
                        atomicOr(p.err, 2u);
                        pmask = 0;
                    }
                    uint64_t* dst = cand + (size_t) qi * cand_qstride + base;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        if (pmask & (1u << i)) {                               // the key is built for survivors only
                            const int32_t row = ridx[(i >> 2) * 16 + kq * 4 + (i & 3)];
                            dst[__popc(pmask & ((1u << i) - 1u))] = make_key(vv[i], g_rank ? g_rank[row] : (uint32_t) row);
                        }
                }
            }
        }

        if (it + 1 < iters && (it + 1) % K2_VOTE_EVERY == 0) {
            bool need = false;
            for (uint32_t q = 0; q < q_count; ++q)
                need |= lds_peek(&ctrl[q].count) > trigger;
            const uint32_t slot = round % 3;
            if (need && lane == 0) atomicOr(&flags[slot], 1u);
            __syncthreads();
            const bool any = lds_peek(&flags[slot]) != 0;
            if (tid == 0) flags[(round + 2) % 3] = 0;
            ++round;
            if (any) {
                for (uint32_t q = 0; q < q_count; ++q) {
                    const uint32_t n = ctrl[q].count < cap ? ctrl[q].count : cap;
                    if (n > trigger) {                                         // only the buffers that are filling up
                        uint64_t* cq = cand + (size_t) q * cand_qstride;
                        for (uint32_t i = tid; i < n; i += MF_THREADS) sortbuf[i] = cq[i];
                        __syncthreads();
                        topk_compact<MF_THREADS>(sortbuf, &ctrl[q], keep, tid, false);
                        for (uint32_t i = tid; i < keep; i += MF_THREADS) cq[i] = sortbuf[i];
                        __syncthreads();
                    }
                }
            }
        }
    }

    __syncthreads();
    constexpr int PR = 32;                                                     // candidate keys per lane at publish
    if (cap <= (uint32_t) (64 * PR)) {
        // publish, one wave per query: the candidates of a (workgroup, query) buffer go to registers and the `keep`
        // smallest are picked by a radix select (vsr_topk.h) -- no sort, no workgroup barrier.  The partial list is
        // unordered; K5 selects again and only the final answer is sorted.
        uint32_t* hist = reinterpret_cast<uint32_t*>(stage);                   // wave-private, the image is dead by now
        for (uint32_t q = (uint32_t) wave; q < q_count; q += MF_WAVES) {
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
            else pick(std::integral_constant<int, PR>{});
        }
        return;
    }
    for (uint32_t q = 0; q < q_count; ++q) {
        const uint32_t n = ctrl[q].count < cap ? ctrl[q].count : cap;
        const uint64_t* cq = cand + (size_t) q * cand_qstride;
        for (uint32_t i = tid; i < n; i += MF_THREADS) sortbuf[i] = cq[i];
        __syncthreads();
        topk_compact<MF_THREADS>(sortbuf, &ctrl[q], keep, tid, false);
        const uint32_t m = ctrl[q].count < keep ? ctrl[q].count : keep;
        uint64_t* dst = p.partial + (size_t) (grp.partial_begin + q * grp.n_blocks + local_block) * p.kp;
        for (uint32_t i = tid; i < p.kp; i += MF_THREADS) dst[i] = i < m ? sortbuf[i] : KEY_EMPTY;
        __syncthreads();
    }
}

template <int METRIC>
hipError_t launch_mfma_metric(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    const uint32_t nstage = (p.stride4 + MF_S - 1) / MF_S;
    const int ng = p.qmax > (uint32_t) MF_NQ ? 2 : 1;
    const size_t lds = mfma_lds_bytes(p.stride4, ng * MF_NQ);
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(MF_THREADS), lds, s, p);
        return hipGetLastError();
    };
    const bool sample = p.sample_stride > 1;
    if (nstage > 4) {
        if (ng == 2) return sample ? launch(mfma_scan_kernel<METRIC, 0, true, 2>) : launch(mfma_scan_kernel<METRIC, 0, false, 2>);
        return sample ? launch(mfma_scan_kernel<METRIC, 0, true, 1>) : launch(mfma_scan_kernel<METRIC, 0, false, 1>);
    }
    // rows of <= 192 floats run on K2w (vsr_mfmaw.h); what reaches this kernel with <= 4 stages are rows of 193 .. 256 floats
    if (ng == 2) return sample ? launch(mfma_scan_kernel<METRIC, 4, true, 2>) : launch(mfma_scan_kernel<METRIC, 4, false, 2>);
    return sample ? launch(mfma_scan_kernel<METRIC, 4, true, 1>) : launch(mfma_scan_kernel<METRIC, 4, false, 1>);
}

}  // namespace vsr
