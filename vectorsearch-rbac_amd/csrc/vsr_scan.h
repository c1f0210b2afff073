// vsr_scan.h — K1: fused distance + RBAC permission test + running top-k over corpus tiles.
//
// Replaces, for a whole ORDER BY <distance> LIMIT k scan, the per-row calls of
//   pgvector/src/vector.c:549-563 VectorL2SquaredDistance / :596-606 VectorInnerProduct /
//   :638-655 VectorCosineSimilarity / :714-724 VectorL1Distance
// plus the top-N sort above them and the row-level-security predicate
//   controller/baseline/pg_row_security/row_level_security.py:54-65 (as a per-row permission bit).
//
// Mapping (wave64): LPR lanes share one corpus row, each lane owning C float4 chunks of it, so one
// wave load instruction reads 64/LPR rows as contiguous LPR*16-byte segments (1 KiB per instruction,
// fully coalesced).  A wave iteration covers a tile of RW = R * 64/LPR rows with R*C independent
// 16-byte loads in flight per lane, and the NEXT tile's loads are issued before the current tile is
// computed.  Up to `qmax` queries share the pass: their vectors sit in LDS and are applied to the
// register-resident tile in sub-batches of QI, so one HBM read serves all of them.  Per-row partial
// sums are combined with a halving butterfly (R registers over LPR lanes), after which lane
// (l % D == 0) owns the finished value of row slot l / D.
// HBM-bound: 3 flop per 4 bytes per query; no MFMA on purpose.
#pragma once
#include "vsr_device.h"
#include "vsr_topk.h"

namespace vsr {

template <int V> struct Log2 { static constexpr int value = 1 + Log2<V / 2>::value; };
template <> struct Log2<1> { static constexpr int value = 0; };

template <int METRIC>
__device__ __forceinline__ void accum4(float& p, const float4& x, const float4& q)
{
    if constexpr (METRIC == M_L2) {
        const float d0 = x.x - q.x, d1 = x.y - q.y, d2 = x.z - q.z, d3 = x.w - q.w;
        p = fmaf(d0, d0, p); p = fmaf(d1, d1, p); p = fmaf(d2, d2, p); p = fmaf(d3, d3, p);
    } else if constexpr (METRIC == M_L1) {
        p += fabsf(x.x - q.x); p += fabsf(x.y - q.y); p += fabsf(x.z - q.z); p += fabsf(x.w - q.w);
    } else {
        p = fmaf(x.x, q.x, p); p = fmaf(x.y, q.y, p); p = fmaf(x.z, q.z, p); p = fmaf(x.w, q.w, p);
    }
}

// fp32 ranking value, monotone in the SQL-level operator result:
//   L2: the fp32 sum of squares (vector.c:584-594 ranks by it too); IP: -dot (vector.c:626-636);
//   cosine: 1 - clamp(dot / sqrt(na*nb)) evaluated in double like vector.c:638-685; L1: the sum.
template <int METRIC>
__device__ __forceinline__ float rank_value(float p, float row_norm2, float q_norm2)
{
    if constexpr (METRIC == M_IP) {
        return -p;
    } else if constexpr (METRIC == M_COSINE) {
        double sim = (double) p / sqrt((double) row_norm2 * (double) q_norm2);
        if (sim > 1.0) sim = 1.0;
        else if (sim < -1.0) sim = -1.0;
        return (float) (1.0 - sim);
    } else {
        return p;
    }
}

// Halving butterfly: N live registers over lanes differing in bit M (and below).
template <int M, int N>
__device__ __forceinline__ void reduce_slots(float* p, int lane)
{
    if constexpr (M >= 1) {
        if constexpr (N > 1) {
            const bool hi = (lane & M) != 0;
#pragma unroll
            for (int i = 0; i < N / 2; ++i) {
                const float keep = hi ? p[i + N / 2] : p[i];
                const float send = hi ? p[i] : p[i + N / 2];
                p[i] = keep + __shfl_xor(send, M);
            }
            reduce_slots<M / 2, N / 2>(p, lane);
        } else {
            p[0] += __shfl_xor(p[0], M);
            reduce_slots<M / 2, 1>(p, lane);
        }
    }
}

__device__ __forceinline__ uint64_t bitmap_window(gptr<uint64_t> bm, uint32_t start)
{
    // the bitmap carries zero pad words, so w + 1 is always readable
    const uint32_t w = start >> 6, sh = start & 63;
    uint64_t v = bm[w] >> sh;
    if (sh) v |= bm[w + 1] << (64 - sh);
    return v;
}

template <int LPR, int R>
struct ScanShape {
    static constexpr int G = 64 / LPR;                 // rows per load instruction
    static constexpr int RW = R * G;                   // rows per wave iteration (tile)
    static constexpr int H = Log2<R>::value < Log2<LPR>::value ? Log2<R>::value : Log2<LPR>::value;
    static constexpr int D = LPR >> H;                 // lanes holding the same finished value
    static constexpr int SLACK = scan_slack(RW);       // keys a workgroup may append between two checks
    static constexpr int XCHK = SLACK / (SCAN_WAVES * RW);
    static_assert(R <= LPR || LPR == 1, "one finished value per lane");
    static_assert(XCHK >= 1 && SCAN_WAVES * RW * XCHK <= SLACK, "append slack");
    static_assert(RW <= 64, "tile rows fit one bitmap window");
};

// One tile's worth of state: descriptor + the lane's share of its rows.
template <int R, int CC>
struct TileRegs {
    uint32_t start;
    uint64_t mask;       // rows of the tile to evaluate (row validity & permission bits); 0 = nothing to do
    float    rn;         // |row|^2 of the row this lane finishes (cosine)
    float4   x[R][CC];
};

// C > 0: compile-time chunk count.  C == 0: runtime chunk loop (any dimension; LPR = 64, qmax <= QI).
// QI: queries evaluated per sub-batch; up to p.qmax queries (a multiple of QI) share one pass.
#ifndef VSR_PREFETCH
#define VSR_PREFETCH 0          // issue the next tile's loads before computing the current one
#endif
#ifndef VSR_MINWAVES
#define VSR_MINWAVES 2          // launch-bounds waves per SIMD the register allocator must allow
#endif

template <int METRIC, int LPR, int C, int R, int QI>
__global__ __launch_bounds__(SCAN_THREADS, VSR_MINWAVES) void scan_kernel(const ScanParams p)
{
    using S = ScanShape<LPR, R>;
    constexpr int G = S::G, RW = S::RW, D = S::D, XCHK = S::XCHK;
    constexpr int CC = C > 0 ? C : 1;

    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l = lane % LPR;
    const int g = lane / LPR;

    // ---- which (filter, query chunk) does this workgroup serve ----
    uint32_t lo = 0, hi = p.n_groups;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p.groups[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid;
    }
    const ScanGroup grp = p.groups[lo];
    const auto g_tiles = as_global(grp.tiles);                                 // global_load, not flat (vsr_device.h)
    const auto g_bitmap = as_global(grp.bitmap);
    const auto g_rank = as_global(p.rank);                                     // list-ordered views: keys carry the rank
    const uint32_t local_block = blockIdx.x - grp.block_begin;
    const uint32_t t0 = (uint32_t) (((uint64_t) grp.n_tiles * local_block) / grp.n_blocks);
    const uint32_t t1 = (uint32_t) (((uint64_t) grp.n_tiles * (local_block + 1)) / grp.n_blocks);

    const uint32_t cap = p.cap, k = p.k, qmax = p.qmax, stride4 = p.stride4;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                       // [qmax][cap]
    TopKCtrl* ctrl = reinterpret_cast<TopKCtrl*>(keys + (size_t) qmax * cap); // [qmax]
    float4*   qlds = reinterpret_cast<float4*>(ctrl + qmax);                  // [qmax][stride4]
    float*    qnl = reinterpret_cast<float*>(qlds + (size_t) qmax * stride4); // [qmax] |q|^2
    uint32_t* flags = reinterpret_cast<uint32_t*>(qnl + qmax);                // [4] overflow votes

    const uint32_t q_count = grp.q_count;
    const uint32_t n_sub = (q_count + QI - 1) / QI;                           // wave-uniform
    for (uint32_t qi = tid; qi < qmax; qi += SCAN_THREADS) {
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        ctrl[qi].tau = p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        ctrl[qi].count = 0;
        qnl[qi] = (METRIC == M_COSINE) ? p.q_norm2[slot] : 0.0f;
    }
    if (tid < 4) flags[tid] = 0;
    for (uint32_t qi = 0; qi < n_sub * QI; ++qi) {                            // pad slots repeat query 0
        const uint32_t slot = p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        const float4* qsrc = reinterpret_cast<const float4*>(p.queries) + (size_t) slot * stride4;
        for (uint32_t i = tid; i < stride4; i += SCAN_THREADS) qlds[(size_t) qi * stride4 + i] = qsrc[i];
    }
    __syncthreads();

    // the row this lane finishes after the butterfly
    const bool own = (l % D) == 0;
    const int row_own = (l / D) * G + g;
    const uint32_t trigger = cap - S::SLACK;
    const uint32_t ss = p.sample_stride;                                       // sample pass: every ss-th tile
    const uint32_t iters = ((t1 - t0 + ss - 1) / ss + SCAN_WAVES - 1) / SCAN_WAVES;

    // issue the loads of tile t (no waits): descriptor, permission bits, row chunks
    auto fetch = [&](uint32_t t, TileRegs<R, CC>& tr) {
        tr.mask = 0;
        tr.start = 0;
        tr.rn = 0.0f;
        if (t >= t1) return;
        uint32_t start, nrows;
        if (g_tiles) {
            const uint2 tl = load_tile(g_tiles, t);
            start = tl.x;
            nrows = tl.y;
        } else {
            start = t * RW;
            nrows = p.n_rows - start < (uint32_t) RW ? p.n_rows - start : (uint32_t) RW;
        }
        uint64_t mask = nrows >= 64 ? ~0ull : ((1ull << nrows) - 1ull);
        if (g_bitmap) mask &= bitmap_window(g_bitmap, start);
        tr.start = start;
        tr.mask = mask;
        if (!mask) return;
        if constexpr (METRIC == M_COSINE) {
            if (own && ((mask >> row_own) & 1ull)) tr.rn = p.norm2[start + row_own];
        }
        if constexpr (C > 0) {
            const float4* base = p.rows + (size_t) start * stride4;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = r * G + g;
                const bool ok = (mask >> row) & 1ull;
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const uint32_t chunk = c * LPR + l;
                    tr.x[r][c] = (ok && chunk < stride4) ? base[(size_t) row * stride4 + chunk]
                                                         : make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };

    TileRegs<R, CC> cur;
#if VSR_PREFETCH
    TileRegs<R, CC> nxt;
    fetch(t0 + wave * ss, cur);
#endif
    uint32_t round = 0;
    for (uint32_t it = 0; it < iters; ++it) {
#if VSR_PREFETCH
        if (it + 1 < iters) fetch(t0 + ((it + 1) * SCAN_WAVES + wave) * ss, nxt);   // prefetch the next tile
        else nxt.mask = 0;
#else
        fetch(t0 + (it * SCAN_WAVES + wave) * ss, cur);
#endif

        if (cur.mask) {                                                        // wave-uniform
            const uint32_t start = cur.start;
            const bool ok_own = (cur.mask >> row_own) & 1ull;
            if constexpr (C > 0) {
                for (uint32_t sb = 0; sb < n_sub; ++sb) {
                    float acc[QI][R];
#pragma unroll
                    for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                        for (int r = 0; r < R; ++r) acc[qi][r] = 0.0f;
#pragma unroll
                    for (int c = 0; c < C; ++c) {
                        const uint32_t chunk = c * LPR + l;
#pragma unroll
                        for (int qi = 0; qi < QI; ++qi) {
                            const float4 qv = chunk < stride4 ? qlds[(size_t) (sb * QI + qi) * stride4 + chunk]
                                                              : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                            for (int r = 0; r < R; ++r) accum4<METRIC>(acc[qi][r], cur.x[r][c], qv);
                        }
                    }
#pragma unroll
                    for (int qi = 0; qi < QI; ++qi) {
                        const uint32_t qs = sb * QI + qi;
                        reduce_slots<LPR / 2, R>(acc[qi], lane);
                        const float v = rank_value<METRIC>(acc[qi][0], cur.rn, qnl[qs]);
                        const uint64_t key = make_key(v, g_rank ? g_rank[start + row_own] : start + row_own);
                        const uint64_t tau = lds_peek(&ctrl[qs].tau);
                        const bool pass = own && ok_own && qs < q_count && key < tau;
                        topk_append(keys + (size_t) qs * cap, &ctrl[qs], pass, key);
                    }
                }
            } else {
                // any dimension: stream the rows chunk by chunk, all (<= QI) queries at once
                const float4* base = p.rows + (size_t) start * stride4;
                float acc[QI][R];
#pragma unroll
                for (int qi = 0; qi < QI; ++qi)
#pragma unroll
                    for (int r = 0; r < R; ++r) acc[qi][r] = 0.0f;
#pragma unroll 4
                for (uint32_t chunk = l; chunk < stride4; chunk += 64) {
                    float4 x[R];
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const bool ok = (cur.mask >> r) & 1ull;
                        x[r] = ok ? base[(size_t) r * stride4 + chunk] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int qi = 0; qi < QI; ++qi) {
                        const float4 qv = qlds[(size_t) qi * stride4 + chunk];
#pragma unroll
                        for (int r = 0; r < R; ++r) accum4<METRIC>(acc[qi][r], x[r], qv);
                    }
                }
#pragma unroll
                for (int qi = 0; qi < QI; ++qi) {
                    reduce_slots<LPR / 2, R>(acc[qi], lane);
                    const float v = rank_value<METRIC>(acc[qi][0], cur.rn, qnl[qi]);
                    const uint64_t key = make_key(v, g_rank ? g_rank[start + row_own] : start + row_own);
                    const uint64_t tau = lds_peek(&ctrl[qi].tau);
                    const bool pass = own && ok_own && (uint32_t) qi < q_count && key < tau;
                    topk_append(keys + (size_t) qi * cap, &ctrl[qi], pass, key);
                }
            }
        }

        if ((it % XCHK) == XCHK - 1 && it + 1 < iters) {                       // workgroup-uniform
            // overflow vote: one barrier; flag slot `round % 3`, recycled two rounds later
            bool need = false;
            for (uint32_t qs = 0; qs < q_count; ++qs)
                need |= lds_peek(&ctrl[qs].count) > trigger;
            const uint32_t slot = round % 3;
            if (need && lane == 0) atomicOr(&flags[slot], 1u);
            __syncthreads();
            const bool any = lds_peek(&flags[slot]) != 0;
            if (tid == 0) flags[(round + 2) % 3] = 0;
            ++round;
            if (any) {
                for (uint32_t qs = 0; qs < q_count; ++qs)
                    if (ctrl[qs].count > trigger)                              // same value in every thread
                        topk_compact<SCAN_THREADS>(keys + (size_t) qs * cap, &ctrl[qs], k, tid, false);
            }
        }
#if VSR_PREFETCH
        cur = nxt;
#endif
    }

    // ---- publish this workgroup's k best per query ----
    __syncthreads();
    for (uint32_t qs = 0; qs < q_count; ++qs) {
        topk_compact<SCAN_THREADS>(keys + (size_t) qs * cap, &ctrl[qs], k, tid, false);
        const uint32_t n = ctrl[qs].count < k ? ctrl[qs].count : k;
        uint64_t* dst = p.partial + (size_t) (grp.partial_begin + qs * grp.n_blocks + local_block) * p.kp;
        for (uint32_t i = tid; i < p.kp; i += SCAN_THREADS) dst[i] = i < n ? keys[(size_t) qs * cap + i] : KEY_EMPTY;
    }
}

template <int METRIC, int LPR, int C, int R, int QI>
hipError_t launch_scan_inst(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    const size_t lds = scan_lds_bytes(p.qmax, p.cap, p.stride4);
    auto kern = scan_kernel<METRIC, LPR, C, R, QI>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(SCAN_THREADS), lds, s, p);
    return hipGetLastError();
}

// shape dispatch for one metric; instantiated once per metric in its own translation unit
template <int METRIC>
hipError_t launch_scan_metric(const ScanParams& p, int dim, int qi, uint32_t n_blocks, hipStream_t s)
{
    const KernelShape sh = scan_shape_for_dim(dim);
#define VSR_CASE(LPR_, C_, R_)                                                             \
    if (sh.lpr == LPR_ && sh.c == C_) {                                                    \
        if (qi == 1) return launch_scan_inst<METRIC, LPR_, C_, R_, 1>(p, n_blocks, s);     \
        if (qi == 4) return launch_scan_inst<METRIC, LPR_, C_, R_, 4>(p, n_blocks, s);     \
        return hipErrorInvalidValue;                                                       \
    }
    VSR_CASE(1, 1, 1)
    VSR_CASE(4, 1, 4)
    VSR_CASE(16, 1, 8)
    VSR_CASE(32, 1, 8)
    VSR_CASE(64, 1, 8)
    VSR_CASE(64, 2, 4)
    VSR_CASE(64, 3, 4)
    VSR_CASE(64, 4, 2)
    VSR_CASE(64, 0, 2)
#undef VSR_CASE
    return hipErrorInvalidValue;
}

}  // namespace vsr
