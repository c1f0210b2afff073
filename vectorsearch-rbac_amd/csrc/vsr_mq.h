// vsr_mq.h — K1m: the shared-pass (multi-query) form of the fused distance + permission + top-k scan.
//
// Same contract as vsr_scan.h's K1 (it replaces the same pgvector loops, vector.c:549-563 / :596-606 /
// :638-655 / :714-724, for whole ORDER BY ... LIMIT k scans), but built for passes that 2..16 queries share:
//
//   * row per lane.  A wave stages 64 corpus rows x S float4 chunks through its private LDS region:
//     coalesced global loads (S lanes per row: S*16 contiguous bytes per row, 64/S rows per instruction),
//     ds_write_b128 into a [row][S+1] padded image, then every lane reads ITS row back chunk by chunk
//     (conflict-free: pitch (S+1)*16 B).  No cross-lane reduction, no shuffles: a lane finishes whole rows.
//   * queries broadcast from LDS.  For each chunk the 4 queries of a sub-batch are read with wave-uniform
//     ds_read_b128 (broadcast) and applied to the lane's row chunk: 8 VALU per query per chunk for L2
//     (4 sub + 4 fma), so the kernel runs at the fp32 VALU rate until ~25 queries share a pass and is
//     HBM-bound below that.  Exact (x-q)^2 arithmetic: no GEMM expansion, hence no MFMA here on purpose.
//   * the next stage's global loads are issued before the current stage is computed (registers are the
//     staging buffer; the wave's LDS image is single-buffered and wave-private: no workgroup barriers in
//     the steady state).
//   * candidates (key < tau) go to a per-(workgroup, query) buffer in global memory through one LDS atomic
//     per wave; the running threshold tau and the counts live in LDS.  Overflow votes and compactions
//     (sort in the LDS staging area) are the only workgroup-wide synchronisation.
#pragma once
#include "vsr_device.h"
#include "vsr_scan.h"
#include "vsr_topk.h"

namespace vsr {

constexpr int MQ_THREADS = 256;          // 4 waves per workgroup, two workgroups per CU
constexpr int MQ_WAVES = MQ_THREADS / 64;
constexpr int MQ_S = 16;                 // float4 chunks per stage (256 contiguous bytes per row and load)
constexpr int MQ_PITCH = MQ_S + 1;       // LDS row pitch in float4 (pad one chunk: conflict-free b128 reads)
constexpr int MQ_SLACK = MQ_WAVES * 64;  // keys a workgroup can append per query between two votes

inline size_t mq_lds_bytes(uint32_t qmax, uint32_t stride4)
{
    const uint32_t nstage = (stride4 + MQ_S - 1) / MQ_S;
    return (size_t) MQ_WAVES * 64 * MQ_PITCH * 16          // staging images
         + (size_t) MQ_WAVES * 64 * 4                      // row index per slot
         + (size_t) qmax * nstage * MQ_S * 16              // queries, zero padded to whole stages
         + (size_t) qmax * (sizeof(TopKCtrl) + 4) + 32;    // tau/count, |q|^2, vote flags
}

template <int METRIC, int NSUB>
__device__ __forceinline__ void mq_body(const ScanParams& p, const ScanGroup& grp, uint32_t local_block,
                                        unsigned char* smem)
{
    constexpr int NQ = NSUB * 4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t stride4 = p.stride4, cap = p.cap, k = p.k, qmax = p.qmax;
    const uint32_t nstage = (stride4 + MQ_S - 1) / MQ_S;
    const uint32_t qpitch = nstage * MQ_S;                                   // float4 per query in LDS
    const uint32_t q_count = grp.q_count;
    const auto g_tiles = as_global(grp.tiles);                                 // global_load, not flat (vsr_device.h)
    const auto g_bitmap = as_global(grp.bitmap);
    const auto g_rank = as_global(p.rank);

    float4*   stage = reinterpret_cast<float4*>(smem) + (size_t) wave * 64 * MQ_PITCH;
    int32_t*  rowidx = reinterpret_cast<int32_t*>(smem + (size_t) MQ_WAVES * 64 * MQ_PITCH * 16) + wave * 64;
    float4*   qlds = reinterpret_cast<float4*>(smem + (size_t) MQ_WAVES * 64 * MQ_PITCH * 16 + (size_t) MQ_WAVES * 64 * 4);
    TopKCtrl* ctrl = reinterpret_cast<TopKCtrl*>(qlds + (size_t) qmax * qpitch);
    float*    qnl = reinterpret_cast<float*>(ctrl + qmax);
    uint32_t* flags = reinterpret_cast<uint32_t*>(qnl + qmax);
    uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem);                   // staging area, reused by compactions

    // ---- queries -> LDS (pad slots repeat query 0 so that every sub-batch is full) ----
    for (uint32_t qi = tid; qi < qmax; qi += MQ_THREADS) {
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        ctrl[qi].tau = p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        ctrl[qi].count = 0;
        qnl[qi] = (METRIC == M_COSINE) ? p.q_norm2[slot] : 0.0f;
    }
    if (tid < 4) flags[tid] = 0;
    for (uint32_t qi = 0; qi < (uint32_t) NQ; ++qi) {
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        const float4* qsrc = reinterpret_cast<const float4*>(p.queries) + (size_t) slot * stride4;
        for (uint32_t i = tid; i < qpitch; i += MQ_THREADS)
            qlds[(size_t) qi * qpitch + i] = i < stride4 ? qsrc[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();

    // ---- this workgroup's tiles, 64 row slots (= tps list tiles) per wave iteration ----
    const uint32_t rw = p.rw;                                                 // rows per list tile (divides 64)
    const uint32_t tps = 64 / rw;
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);
    const uint32_t n_super = (t1 - t0 + tps - 1) / tps;
    const uint32_t ss = p.sample_stride;                                       // sample pass: every ss-th super-tile
    const uint32_t iters = ((n_super + ss - 1) / ss + MQ_WAVES - 1) / MQ_WAVES;
    const uint32_t trigger = cap - MQ_SLACK;
    uint64_t* cand = p.cand + (size_t) (grp.partial_begin + local_block) * cand_pitch(cap);   // + qs * n_blocks * cap
    const size_t cand_qstride = (size_t) grp.n_blocks * cand_pitch(cap);

    const int lps_row = lane / MQ_S;          // which of the 64/S rows of a load instruction
    const int lps_chunk = lane % MQ_S;        // which chunk of the stage
    constexpr int RPI = 64 / MQ_S;            // rows per load instruction

    uint32_t round = 0;
    for (uint32_t it = 0; it < iters; ++it) {
        const uint32_t sup = (it * MQ_WAVES + wave) * ss;                      // wave-uniform
        const bool active = sup < n_super;
        int32_t myrow = -1;
        if (active) {
            const uint32_t t = t0 + sup * tps + (uint32_t) lane / rw;
            const uint32_t r = (uint32_t) lane % rw;
            if (t < t1) {
                uint32_t start, nrows;
                if (g_tiles) {
                    const uint2 tl = load_tile(g_tiles, t);
                    start = tl.x;
                    nrows = tl.y;
                } else {
                    start = t * rw;
                    nrows = p.n_rows - start < rw ? p.n_rows - start : rw;
                }
                if (r < nrows) {
                    const uint32_t row = start + r;
                    bool ok = true;
                    if (g_bitmap) ok = (g_bitmap[row >> 6] >> (row & 63)) & 1ull;
                    if (ok) myrow = (int32_t) row;
                }
            }
        }
        const bool any_row = __ballot(myrow >= 0) != 0;                        // wave-uniform
        if (any_row) {
            rowidx[lane] = myrow;                                              // wave-private LDS: in-order
            float rn = 0.0f;
            if constexpr (METRIC == M_COSINE) {
                if (myrow >= 0) rn = p.norm2[myrow];
            }
            float acc[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) acc[q] = 0.0f;

            float4 x[MQ_S];
            auto issue = [&](uint32_t s) {                                     // global loads of stage s
                const uint32_t chunk = s * MQ_S + lps_chunk;
#pragma unroll
                for (int u = 0; u < MQ_S; ++u) {
                    const int32_t r = rowidx[u * RPI + lps_row];
                    x[u] = (r >= 0 && chunk < stride4) ? p.rows[(size_t) r * stride4 + chunk]
                                                        : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            };
            issue(0);
            for (uint32_t s = 0; s < nstage; ++s) {
#pragma unroll
                for (int u = 0; u < MQ_S; ++u) stage[(u * RPI + lps_row) * MQ_PITCH + lps_chunk] = x[u];
                if (s + 1 < nstage) issue(s + 1);                              // in flight during the compute below
                const float4* myx = stage + lane * MQ_PITCH;
                const float4* qs_base = qlds + (size_t) s * MQ_S;
#pragma unroll 2
                for (int c = 0; c < MQ_S; ++c) {
                    // all reads of the chunk first (1 row chunk + NQ broadcast query chunks), then the VALU work:
                    // in-order LDS returns let the compiler wait with counted lgkmcnt while later reads fly
                    const float4 xv = myx[c];
                    float4 qv[NQ];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) qv[q] = qs_base[(size_t) q * qpitch + c];
#pragma unroll
                    for (int q = 0; q < NQ; ++q) accum4<METRIC>(acc[q], xv, qv[q]);
                }
            }

            const bool valid = myrow >= 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                if ((uint32_t) q < q_count) {                                  // workgroup-uniform
                    const float v = rank_value<METRIC>(acc[q], rn, qnl[q]);
                    const uint64_t key = make_key(v, g_rank && valid ? g_rank[myrow] : (uint32_t) myrow);
                    const uint64_t tau = lds_peek(&ctrl[q].tau);
                    topk_append(cand + (size_t) q * cand_qstride, &ctrl[q], valid && key < tau, key);
                }
            }
        }

        if (it + 1 < iters) {                                                  // workgroup-uniform vote
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
                    const uint32_t n = ctrl[q].count;                          // same in every thread
                    if (n > trigger) {                                         // only the buffers that are filling up
                        uint64_t* cq = cand + (size_t) q * cand_qstride;
                        for (uint32_t i = tid; i < n; i += MQ_THREADS) sortbuf[i] = cq[i];
                        __syncthreads();
                        topk_compact<MQ_THREADS>(sortbuf, &ctrl[q], k, tid, false);
                        for (uint32_t i = tid; i < k; i += MQ_THREADS) cq[i] = sortbuf[i];
                        __syncthreads();
                    }
                }
            }
        }
    }

    // ---- publish this workgroup's k best per query ----
    __syncthreads();
    for (uint32_t q = 0; q < q_count; ++q) {
        const uint32_t n = ctrl[q].count;
        const uint64_t* cq = cand + (size_t) q * cand_qstride;
        for (uint32_t i = tid; i < n; i += MQ_THREADS) sortbuf[i] = cq[i];
        __syncthreads();
        topk_compact<MQ_THREADS>(sortbuf, &ctrl[q], k, tid, false);
        const uint32_t m = ctrl[q].count < k ? ctrl[q].count : k;
        uint64_t* dst = p.partial + (size_t) (grp.partial_begin + q * grp.n_blocks + local_block) * p.kp;
        for (uint32_t i = tid; i < p.kp; i += MQ_THREADS) dst[i] = i < m ? sortbuf[i] : KEY_EMPTY;
        __syncthreads();
    }
}

// SAMPLE only gives the seeding pass (p.sample_stride > 1) its own kernel symbol in profiles.
template <int METRIC, bool SAMPLE>
__global__ __launch_bounds__(MQ_THREADS, 2) void mq_scan_kernel(const ScanParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
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
    switch ((grp.q_count + 3) / 4) {                                           // workgroup-uniform
    case 1:  mq_body<METRIC, 1>(p, grp, local_block, smem); break;
    case 2:  mq_body<METRIC, 2>(p, grp, local_block, smem); break;
    case 3:  mq_body<METRIC, 3>(p, grp, local_block, smem); break;
    default: mq_body<METRIC, 4>(p, grp, local_block, smem); break;
    }
}

template <int METRIC>
hipError_t launch_mq_metric(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    const size_t lds = mq_lds_bytes(p.qmax, p.stride4);
    auto launch = [&](auto kern) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(MQ_THREADS), lds, s, p);
        return hipGetLastError();
    };
    return p.sample_stride > 1 ? launch(mq_scan_kernel<METRIC, true>) : launch(mq_scan_kernel<METRIC, false>);
}

}  // namespace vsr
