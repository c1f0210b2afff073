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
// 16-byte loads in flight per lane (several waves per SIMD overlap one tile's loads with another's
// arithmetic; issuing the next tile's loads early -- VSR_PREFETCH -- is compiled out: it costs the
// registers that occupancy needs).  Up to `qmax` queries share the pass: their vectors sit in LDS and are applied to the
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

// ---- nq == 1: in-kernel merge tree (FusedTail, vsr_device.h) ----
constexpr int FUSED_KPT = 16;                               // keys per thread of one merge: <= 8192 keys
constexpr uint32_t FUSED_CLS = 64;                          // a tie class this small is ranked directly (ranking is quadratic: 1024 cost 10-25 us)
// LDS of a merge (in units of keys, inside the workgroup's top-k buffer of >= 4096 keys): the selected keys, the final
// tie class, one digit histogram per wave, scalars
constexpr uint32_t FUSED_OUT = 0, FUSED_CLS_AT = 512, FUSED_HIST_AT = 1536, FUSED_SC_AT = 2560;
// The k smallest of up to SCAN_THREADS * 16 keys at `src` (global memory written by other workgroups of this launch:
// device-scope loads) -> out[0 .. count) in LDS, unordered.  The keys sit in registers; an MSB-first radix select over
// 8-bit digits (one histogram per wave in LDS, digits every key shares are skipped) narrows the class holding the k-th
// key; as soon as that class has <= 1024 keys (SIFT's integer distances tie a lot, and ties are broken by the row id in
// the low word: without this the select would walk all 8 digits) its keys are ranked directly.  No sorting network.
// Returns count (same in every thread).
template <bool LOCAL = false>                               // LOCAL: `src` is the workgroup's own LDS buffer (may alias `lds`)
__device__ __forceinline__ uint32_t fused_block_select(const uint64_t* src, uint32_t n, uint32_t k, uint64_t* lds, int tid)
{
    uint64_t* out = lds + FUSED_OUT;
    uint64_t* cls = lds + FUSED_CLS_AT;
    uint32_t* hist = reinterpret_cast<uint32_t*>(lds + FUSED_HIST_AT);         // [SCAN_WAVES][256]
    uint64_t* sc = lds + FUSED_SC_AT;                                          // [0] min, [1] max
    uint32_t* cnt = reinterpret_cast<uint32_t*>(sc + 2);    // [0] real keys, [1] selected so far, [2] digit, [3] keys before it,
                                                            // [4] population of the chosen bin, [5] class fill
    const int wave = tid >> 6;
    uint64_t reg[FUSED_KPT];
#pragma unroll
    for (int r = 0; r < FUSED_KPT; ++r) {
        const uint32_t i = (uint32_t) (r * SCAN_THREADS + tid);
        if constexpr (LOCAL) reg[r] = i < n ? src[i] : KEY_EMPTY;
        else reg[r] = i < n ? __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : KEY_EMPTY;
    }
    __syncthreads();
    if (tid == 0) {
        sc[0] = KEY_EMPTY;
        sc[1] = 0;
        cnt[0] = cnt[1] = cnt[5] = 0;
    }
    __syncthreads();
    {
        uint64_t mn = KEY_EMPTY, mx = 0;
        uint32_t real = 0;
#pragma unroll
        for (int r = 0; r < FUSED_KPT; ++r)
            if (reg[r] != KEY_EMPTY) {
                mn = reg[r] < mn ? reg[r] : mn;
                mx = reg[r] > mx ? reg[r] : mx;
                ++real;
            }
        // wave-level reduction first: one LDS atomic per wave, not per thread
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const uint64_t omn = __shfl_xor(mn, d), omx = __shfl_xor(mx, d);
            mn = omn < mn ? omn : mn;
            mx = omx > mx ? omx : mx;
            real += (uint32_t) __shfl_xor((int) real, d);
        }
        if ((tid & 63) == 0 && real) {
            atomicMin(reinterpret_cast<unsigned long long*>(sc), (unsigned long long) mn);
            atomicMax(reinterpret_cast<unsigned long long*>(sc + 1), (unsigned long long) mx);
            atomicAdd(cnt, real);
        }
    }
    __syncthreads();
    const uint32_t n_real = cnt[0];
    uint64_t prefix = 0, mask = 0;                          // class of the k-th key: (key & mask) == prefix
    uint32_t want = k, pop = n_real;                        // keys to take from the class / keys in it
    if (n_real > k) {
        const uint64_t diff = sc[0] ^ sc[1];
        int shift = diff ? (63 - __builtin_clzll(diff)) / 8 * 8 : 0;           // first digit in which the keys differ
        mask = shift >= 56 ? 0ull : ~0ull << (shift + 8);
        prefix = sc[0] & mask;
        while (pop > FUSED_CLS && pop != want) {
            for (int i = tid; i < SCAN_WAVES * 256; i += SCAN_THREADS) hist[i] = 0;
            __syncthreads();
#pragma unroll
            for (int r = 0; r < FUSED_KPT; ++r)
                if (reg[r] != KEY_EMPTY && (reg[r] & mask) == prefix)
                    atomicAdd(&hist[wave * 256 + ((uint32_t) (reg[r] >> shift) & 255u)], 1u);
            __syncthreads();
            if (tid < 64) {                                 // wave 0: the bin holding the want-th key of the class
                uint32_t h[4], sum = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    h[j] = 0;
#pragma unroll
                    for (int w = 0; w < SCAN_WAVES; ++w) h[j] += hist[w * 256 + tid * 4 + j];
                    sum += h[j];
                }
                uint32_t incl = sum;                        // inclusive prefix sum over the lanes
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = (uint32_t) __shfl_up((int) incl, d);
                    if (tid >= d) incl += o;
                }
                uint32_t before = incl - sum;
                if (before < want && want <= incl) {        // exactly one lane
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (before < want && want <= before + h[j]) {
                            cnt[2] = (uint32_t) (tid * 4 + j);
                            cnt[3] = before;
                            cnt[4] = h[j];
                        }
                        before += h[j];
                    }
                }
            }
            __syncthreads();
            prefix |= (uint64_t) cnt[2] << shift;
            mask |= 0xFFull << shift;
            want -= cnt[3];
            pop = cnt[4];
            shift -= 8;                                     // (keys are unique: pop reaches 1 at the last digit at the latest)
            __syncthreads();                                // everybody has read cnt before the next round rewrites it
        }
    }
    // everything below the class is selected; the class itself entirely, or its `want` smallest by direct ranking
    const bool rank_class = n_real > k && pop != want;
#pragma unroll
    for (int r = 0; r < FUSED_KPT; ++r) {
        if (reg[r] == KEY_EMPTY) continue;
        const uint64_t hi = reg[r] & mask;
        if (n_real <= k || hi < prefix || (hi == prefix && !rank_class)) out[atomicAdd(cnt + 1, 1u)] = reg[r];
        else if (hi == prefix) cls[atomicAdd(cnt + 5, 1u)] = reg[r];
    }
    __syncthreads();
    if (rank_class) {
        const uint32_t fill = cnt[5];                       // == pop <= FUSED_CLS
        for (uint32_t j = (uint32_t) tid; j < fill; j += SCAN_THREADS) {
            const uint64_t mine = cls[j];
            uint32_t rank = 0;
            for (uint32_t i = 0; i < fill; ++i) rank += cls[i] < mine;          // LDS broadcast reads
            if (rank < want) out[atomicAdd(cnt + 1, 1u)] = mine;
        }
        __syncthreads();
    }
    return cnt[1];
}

// bad: the query broke a promise the kernel relied on (scan8: not integer-valued in 0..255): the result is written but FLAGGED
// (negative count, flag word set), like a query the screening could not prove; the caller re-runs it on the exact path
__device__ __forceinline__ void fused_tail(const ScanParams& p, const ScanGroup& grp, uint32_t local_block, uint64_t* keys,
                                        int tid, bool bad = false, uint64_t t_start = 0, uint64_t t_scan = 0, uint64_t t_pub = 0)
{
    __shared__ uint32_t s_last;
    const FusedTail& f = p.fused;
    uint64_t t_m1 = 0, t_m1s = 0;
    const uint32_t B = grp.n_blocks, fan = f.fan, n_g = (B + fan - 1) / fan, k = p.k, kp = p.kp;
    const uint32_t my_g = local_block / fan;
    const uint32_t g_size = (my_g + 1) * fan <= B ? fan : B - my_g * fan;
    uint64_t* lists = p.partial + (size_t) grp.partial_begin * kp;            // [B] workgroup lists, then [n_g] merged lists
    // Arrival.  The lists are exchanged with DEVICE-SCOPE stores and loads (write-through / cache-bypassing), not with
    // __threadfence(): a device-scope release on this chip writes the whole L2 back, and a thousand workgroups doing
    // that cost 0.3 ms.  __syncthreads() waits for the workgroup's stores to complete, so they are visible device-wide
    // before the counter moves; the workgroup that sees the last arrival owns the merge.  Nobody ever waits: every
    // workgroup but one per counter simply ends.
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(f.done + 1 + my_g, 1u) == g_size - 1u;
    __syncthreads();
    if (!s_last) return;
    const uint64_t* src = lists;
    uint32_t n_src = B * kp;
    if (n_g > 1) {
        if (f.dbg) t_m1s = wall_clock64();
        const uint32_t n = fused_block_select(lists + (size_t) my_g * fan * kp, g_size * kp, k, keys, tid);   // (the top-k buffer is free now)
        if (f.dbg) t_m1 = wall_clock64();
        uint64_t* dst = lists + (size_t) (B + my_g) * kp;
        for (uint32_t i = tid; i < kp; i += SCAN_THREADS)
            __hip_atomic_store(dst + i, i < n ? keys[i] : KEY_EMPTY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid == 0) s_last = atomicAdd(f.done, 1u) == n_g - 1u;
        __syncthreads();
        if (!s_last) return;
        src = lists + (size_t) B * kp;
        n_src = n_g * kp;
    }
    const uint64_t t_m2s = f.dbg ? wall_clock64() : 0;
    const uint32_t m = fused_block_select(src, n_src, k, keys, tid);
    const uint64_t t_m2 = f.dbg ? wall_clock64() : 0;
    // the caller's order: every thread that holds a key counts the keys below it (keys are unique; m <= 512 broadcast reads)
    // and writes its row at that position -- no sorting network, no further barrier
    for (uint32_t i = tid; i < k; i += SCAN_THREADS) {
        if (i < m) {
            const uint64_t key = keys[i];
            uint32_t at = 0;
            for (uint32_t j = 0; j < m; ++j) at += keys[j] < key;
            const uint32_t row = (uint32_t) key;
            const float v = mono_to_float((uint32_t) (key >> 32));
            f.out_block[at] = f.block_ids[row];
            f.out_doc[at] = f.doc_ids[row];
            if (f.out_row) f.out_row[at] = f.orig_rows[row];
            f.out_dist[at] = f.metric == M_L2 ? (float) sqrt((double) v) : v;  // vector.c:577
            if (f.out_keys) f.out_keys[at] = (key & 0xFFFFFFFF00000000ull) | (uint64_t) (row + f.row_offset);
        } else {
            f.out_block[i] = -1;
            f.out_doc[i] = -1;
            if (f.out_row) f.out_row[i] = -1;
            f.out_dist[i] = __builtin_inff();
            if (f.out_keys) f.out_keys[i] = KEY_EMPTY;
        }
    }
    if (tid == 0) {
        f.out_count[0] = bad ? -1 - (int32_t) m : (int32_t) m;
        f.out_flag[0] = bad ? 1 : 0;
        if (bad && f.flag_total) atomicAdd(f.flag_total, 1);
    }
    for (uint32_t i = tid; i < 1 + n_g; i += SCAN_THREADS) f.done[i] = 0;     // every other workgroup is past its counter
    if (f.dbg && tid == 0) {
        f.dbg[0] = t_start; f.dbg[1] = t_scan; f.dbg[2] = t_pub; f.dbg[3] = t_m1s; f.dbg[4] = t_m1; f.dbg[5] = t_m2s; f.dbg[6] = t_m2;
        f.dbg[7] = wall_clock64();
    }
}

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
    const bool fused = QI == 1 && p.fused.enable;                              // one query, one pass, one launch
    uint32_t lo = 0, hi = fused ? 1u : p.n_groups;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (p.groups[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid;
    }
    const ScanGroup grp = fused ? p.fused.group : p.groups[lo];
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
        const uint32_t slot = fused ? 0u : p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
        ctrl[qi].tau = p.tau_init ? p.tau_init[slot] : KEY_EMPTY;
        ctrl[qi].count = 0;
        qnl[qi] = (METRIC == M_COSINE) ? p.q_norm2[slot] : 0.0f;
    }
    if (tid < 4) flags[tid] = 0;
    for (uint32_t qi = 0; qi < n_sub * QI; ++qi) {                            // pad slots repeat query 0
        const uint32_t slot = fused ? 0u : p.q_slots[grp.q_begin + (qi < q_count ? qi : 0)];
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
        uint32_t n;
        if (fused && ctrl[qs].count <= (uint32_t) (SCAN_THREADS * FUSED_KPT)) {
            // one query per call: the k smallest UNORDERED by a radix select on registers (the merge does not need them sorted;
            // the bitonic sort of ~1000 keys, 55 barriers, was a quarter of the call)
            const uint32_t have = ctrl[qs].count;           // (everybody reads it before the select reuses the buffer)
            n = have > k ? fused_block_select<true>(keys, have, k, keys, tid) : have;
        } else {
            topk_compact<SCAN_THREADS>(keys + (size_t) qs * cap, &ctrl[qs], k, tid, false);
            n = ctrl[qs].count < k ? ctrl[qs].count : k;
        }
        uint64_t* dst = p.partial + (size_t) (grp.partial_begin + qs * grp.n_blocks + local_block) * p.kp;
        if (fused) {                                        // device-scope stores: read by another workgroup of this launch
            for (uint32_t i = tid; i < p.kp; i += SCAN_THREADS)
                __hip_atomic_store(dst + i, i < n ? keys[(size_t) qs * cap + i] : KEY_EMPTY, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (uint32_t i = tid; i < p.kp; i += SCAN_THREADS) dst[i] = i < n ? keys[(size_t) qs * cap + i] : KEY_EMPTY;
        }
    }
    if constexpr (QI == 1) {
        if (fused) fused_tail(p, grp, local_block, keys, tid);
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
