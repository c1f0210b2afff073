// vsr_hnsw.h — K4: HNSW layer search on the GPU, one wave per query.
//
// Replaces HnswSearchLayer (pgvector/src/hnswutils.c:813-976) as hnswgettuple's first call drives it (GetScanItems,
// hnswscan.c:15-45): greedy descent with ef = 1 through the upper layers, then the ef_search beam on layer 0, then the
// heap TIDs of the result elements nearest first (newest TID of an element first, hnswscan.c:278-311), the permission bit
// of every row (the executor's RLS filter above the index scan) and the first k.
//
// Candidates are totally ordered by key = (monotone fp32 index distance << 32) | element id: a fixed choice where
// pgvector's pairing heaps leave candidates of equal distance to insertion history (the tests' CPU checker makes the
// same choice).  Both of
// Algorithm 2's sets live in ONE sorted array S in LDS: every element ever pushed, with an "expanded" flag.  W (the ef
// best found) is S's first min(pushed, ef) entries, C (still to expand) its unexpanded entries.  An expansion reads the
// neighbour list (2m ids on layer 0), marks them in the query's visited bitmap (global memory, one atomicOr each), computes
// the distances of the unvisited ones with 8 row gathers in flight per wave (half a wave per row), and then applies the
// admission rule  d < furthest(W) || |W| < ef  to them one by one in list order, exactly like the sequential loop.
#pragma once
#include "vsr_device.h"
#include "vsr_topk.h"

namespace vsr {

struct HnswParams {
    const float4*   rows;          // base corpus rows (internal order)
    uint32_t        stride4;
    int             metric;        // M_L2 / M_IP / M_COSINE (unit rows: ranked by negative inner product)
    const float*    queries;       // [nq][stride4 * 4] zero padded
    uint32_t        n_elem;
    int32_t         entry, entry_level;
    uint32_t        m, max_level;
    const int32_t*  elem_row;      // element -> internal row holding its vector
    const int32_t*  nbr0;          // [n_elem][2m], -1 padded
    const int32_t*  up_slot;       // element -> slot in up_nbr or -1
    const int32_t*  up_nbr;        // [n_upper][max_level][m], -1 padded
    const int32_t*  level;         // element -> top level
    const int32_t*  tid_count;     // element -> heap TIDs (<= 10)
    const int32_t*  tids;          // [n_elem][10] internal rows
    const uint64_t* const* bitmaps;  // per query: permission bitmap over internal rows, or nullptr
    uint32_t        ef, k, caps;   // caps: capacity of S (>= ef + 2m)
    uint32_t*       visited;       // [nq][visited_words]
    uint32_t        visited_words;
    const int64_t*  block_ids; const int32_t* doc_ids; const int64_t* orig_rows;
    int64_t* out_block; int32_t* out_doc; int64_t* out_row; float* out_dist; int32_t* out_count;
    int64_t*        out_visited;   // optional [nq]: elements entered into the visited set on layer 0
    uint32_t*       err;
};

constexpr int HN_UPPER_VISITED = 1024;     // visited list of an upper-layer (ef = 1) search, in LDS

__device__ __forceinline__ float hnsw_rank_value(int metric, float s) { return metric == M_L2 ? s : -s; }

__global__ __launch_bounds__(64) void hnsw_search_kernel(const HnswParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* S = reinterpret_cast<uint64_t*>(smem);                         // [caps] sorted keys
    uint8_t*  X = reinterpret_cast<uint8_t*>(S + p.caps);                    // [caps] expanded flags
    int32_t*  nb = reinterpret_cast<int32_t*>(X + ((p.caps + 15) & ~15u));   // [64] neighbour ids of the expansion
    float*    nd = reinterpret_cast<float*>(nb + 64);                        // [64] their distances
    int32_t*  uv = reinterpret_cast<int32_t*>(nd + 64);                      // [HN_UPPER_VISITED] upper-layer visited list
    const int lane = threadIdx.x;
    const uint32_t qi = blockIdx.x;
    const float4* q = reinterpret_cast<const float4*>(p.queries) + (size_t) qi * p.stride4;
    uint32_t* vis = p.visited + (size_t) qi * p.visited_words;
    const int half = lane >> 5, hl = lane & 31;

    // distance of the query to `cnt` elements listed in nb[] -> nd[] (fp32 sum; exact on integer-valued data in any order)
    auto distances = [&](int cnt) {
        constexpr int U = 4;
        for (int c0 = 0; c0 < cnt; c0 += 2 * U) {
            float s[U];
            int32_t row[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + 2 * u + half;
                s[u] = 0.0f;
                row[u] = c < cnt ? p.elem_row[nb[c]] : 0;
            }
            for (uint32_t ch = hl; ch < p.stride4; ch += 32) {
                const float4 b = q[ch];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(p.rows + (size_t) row[u] * p.stride4 + ch);
                    if (p.metric == M_L2) {
                        const float d0 = av[0] - b.x, d1 = av[1] - b.y, d2 = av[2] - b.z, d3 = av[3] - b.w;
                        s[u] = fmaf(d0, d0, s[u]); s[u] = fmaf(d1, d1, s[u]); s[u] = fmaf(d2, d2, s[u]); s[u] = fmaf(d3, d3, s[u]);
                    } else {
                        s[u] = fmaf(av[0], b.x, s[u]); s[u] = fmaf(av[1], b.y, s[u]); s[u] = fmaf(av[2], b.z, s[u]); s[u] = fmaf(av[3], b.w, s[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                for (int mm = 16; mm >= 1; mm >>= 1) s[u] += __shfl_xor(s[u], mm);
                const int c = c0 + 2 * u + half;
                if (hl == 0 && c < cnt) nd[c] = hnsw_rank_value(p.metric, s[u]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    uint32_t count = 0;                      // entries of S
    uint32_t pushed = 0;                     // wlen of the reference: pushes so far (never decremented)
    // insert (key, unexpanded) into S keeping it sorted; beyond caps the largest entry falls off
    auto insert = [&](uint64_t key) {
        uint32_t pos = 0;
        for (uint32_t i = (uint32_t) lane; i < ((count + 63) & ~63u); i += 64)
            pos += (uint32_t) __popcll(__ballot(i < count && S[i] < key));
        if (pos >= p.caps) return;
        const uint32_t last = count < p.caps ? count : p.caps - 1;      // index the shifted tail ends at
        for (int64_t base = (int64_t) ((last - 1) & ~63u); last > pos && base >= (int64_t) (pos & ~63u); base -= 64) {   // from the end
            const uint32_t i = (uint32_t) base + (uint32_t) lane;
            const bool mv = i >= pos && i < last;
            const uint64_t kk = mv ? S[i] : 0;
            const uint8_t xx = mv ? X[i] : 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (mv) { S[i + 1] = kk; X[i + 1] = xx; }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (lane == 0) { S[pos] = key; X[pos] = 0; }
        if (count < p.caps) ++count;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };

    int64_t visited_l0 = 0;
    // Algorithm 2 on layer lc with beam ef_; S holds the entry points (unexpanded) on entry and W (sorted) on exit
    auto search_layer = [&](int lc, uint32_t ef_) {
        const uint32_t lm = lc == 0 ? 2 * p.m : p.m;
        uint32_t n_uv = 0;
        // entry points count as visited
        for (uint32_t i = (uint32_t) lane; i < count; i += 64) {
            const uint32_t e = (uint32_t) S[i];
            if (lc == 0) atomicOr(&vis[e >> 5], 1u << (e & 31));
        }
        if (lc != 0) {                       // (one entry point per upper layer)
            if (lane == 0) uv[0] = (int32_t) (uint32_t) S[0];
            n_uv = 1;
        } else
            visited_l0 += count;
        pushed = count;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (;;) {
            // c = nearest unexpanded entry
            uint32_t cpos = 0xFFFFFFFFu;
            for (uint32_t base = 0; base < count && cpos == 0xFFFFFFFFu; base += 64) {
                const uint32_t i = base + (uint32_t) lane;
                const uint64_t mk = __ballot(i < count && X[i] == 0);
                if (mk) cpos = base + (uint32_t) __ffsll((unsigned long long) mk) - 1;
            }
            if (cpos == 0xFFFFFFFFu) break;                                  // C is empty
            const uint64_t ckey = S[cpos];
            const uint32_t wl = pushed < ef_ ? pushed : ef_;                   // |W|
            const uint64_t fkey = S[(wl < count ? wl : count) - 1];
            if (mono_to_float((uint32_t) (ckey >> 32)) > mono_to_float((uint32_t) (fkey >> 32))) break;
            if (lane == 0) X[cpos] = 1;
            const uint32_t ce = (uint32_t) ckey;
            // neighbour list of c on this layer
            int32_t my = -1;
            if ((uint32_t) lane < lm) {
                if (lc == 0) my = p.nbr0[(size_t) ce * 2 * p.m + lane];
                else {
                    const int32_t slot = p.up_slot[ce];
                    my = slot >= 0 ? p.up_nbr[((size_t) slot * p.max_level + (uint32_t) (lc - 1)) * p.m + lane] : -1;
                }
            }
            // unvisited ones, in list order
            bool fresh = false;
            if (my >= 0) {
                if (lc == 0) {
                    const uint32_t old = atomicOr(&vis[(uint32_t) my >> 5], 1u << ((uint32_t) my & 31));
                    fresh = !((old >> ((uint32_t) my & 31)) & 1u);
                } else {
                    fresh = true;
                    for (uint32_t j = 0; j < n_uv; ++j) fresh &= uv[j] != my;
                }
            }
            const uint64_t fm = __ballot(fresh);
            const int cnt = __popcll(fm);
            if (lc == 0) visited_l0 += cnt;
            if (fresh) {
                const int at = __popcll(fm & ((1ull << lane) - 1ull));
                nb[at] = my;
                if (lc != 0) {
                    if (n_uv + (uint32_t) at < (uint32_t) HN_UPPER_VISITED) uv[n_uv + (uint32_t) at] = my;
                    else atomicOr(p.err, 8u);                               // cannot happen at ef = 1
                }
            }
            n_uv += (uint32_t) cnt;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (cnt == 0) continue;
            distances(cnt);
            for (int i = 0; i < cnt; ++i) {                                  // the sequential admission of Algorithm 2
                const uint32_t e = (uint32_t) nb[i];
                const float ed = nd[i];
                const bool always = pushed < ef_;
                const uint32_t wl2 = pushed < ef_ ? pushed : ef_;
                const float fd = mono_to_float((uint32_t) (S[(wl2 < count ? wl2 : count) - 1] >> 32));
                if (!(ed < fd || always)) continue;
                if (p.level[e] < lc) continue;
                insert(make_key(ed, e));
                ++pushed;
            }
        }
        const uint32_t wl = pushed < ef_ ? pushed : ef_;
        count = wl < count ? wl : count;                                     // S = W, nearest first
    };

    if (p.entry < 0) {
        if (lane == 0) p.out_count[qi] = 0;
        for (uint32_t i = (uint32_t) lane; i < p.k; i += 64) {
            p.out_block[(size_t) qi * p.k + i] = -1; p.out_doc[(size_t) qi * p.k + i] = -1;
            if (p.out_row) p.out_row[(size_t) qi * p.k + i] = -1;
            p.out_dist[(size_t) qi * p.k + i] = __builtin_inff();
        }
        return;
    }
    // HnswEntryCandidate
    if (lane == 0) nb[0] = p.entry;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    distances(1);
    insert(make_key(nd[0], (uint32_t) p.entry));
    for (int lc = p.entry_level; lc >= 1; --lc) {
        search_layer(lc, 1);
        if (lane == 0)
            for (uint32_t i = 0; i < count; ++i) X[i] = 0;                    // W becomes the next layer's entry points
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    search_layer(0, p.ef);
    if (p.out_visited && lane == 0) p.out_visited[qi] = visited_l0;

    // hnswgettuple: elements nearest first, their heap TIDs newest first, the permission bit, the first k
    const uint64_t* bm = p.bitmaps ? p.bitmaps[qi] : nullptr;
    uint32_t out = 0;
    for (uint32_t base = 0; base < count && out < p.k; base += 64) {
        const uint32_t i = base + (uint32_t) lane;
        uint32_t e = 0, nt = 0, okmask = 0;
        float d = 0.0f;
        if (i < count) {
            e = (uint32_t) S[i];
            d = mono_to_float((uint32_t) (S[i] >> 32));
            nt = (uint32_t) p.tid_count[e];
            for (uint32_t t = 0; t < nt; ++t) {                              // bit t: TID nt-1-t (newest first) is permitted
                const uint32_t row = (uint32_t) p.tids[(size_t) e * 10 + (nt - 1 - t)];
                if (!bm || ((bm[row >> 6] >> (row & 63)) & 1ull)) okmask |= 1u << t;
            }
        }
        uint32_t mine = (uint32_t) __popc(okmask), incl = mine;
        for (int dd = 1; dd < 64; dd <<= 1) {
            const uint32_t o = (uint32_t) __shfl_up((int) incl, dd);
            if (lane >= dd) incl += o;
        }
        uint32_t at = out + incl - mine;
        for (uint32_t t = 0; t < nt; ++t)
            if ((okmask >> t) & 1u) {
                if (at < p.k) {
                    const uint32_t row = (uint32_t) p.tids[(size_t) e * 10 + (nt - 1 - t)];
                    const size_t o = (size_t) qi * p.k + at;
                    p.out_block[o] = p.block_ids[row];
                    p.out_doc[o] = p.doc_ids[row];
                    if (p.out_row) p.out_row[o] = p.orig_rows[row];
                    float v;
                    if (p.metric == M_L2) v = (float) sqrt((double) d);       // l2_distance, vector.c:577
                    else if (p.metric == M_IP) v = d;                        // <#> = negative inner product
                    else {                                                   // unit rows: cosine distance = 1 - dot
                        double sim = -(double) d;
                        if (sim > 1.0) sim = 1.0; else if (sim < -1.0) sim = -1.0;
                        v = (float) (1.0 - sim);
                    }
                    p.out_dist[o] = v;
                }
                ++at;
            }
        out += (uint32_t) __shfl((int) incl, 63);
    }
    if (out > p.k) out = p.k;
    for (uint32_t i = out + (uint32_t) lane; i < p.k; i += 64) {
        const size_t o = (size_t) qi * p.k + i;
        p.out_block[o] = -1; p.out_doc[o] = -1;
        if (p.out_row) p.out_row[o] = -1;
        p.out_dist[o] = __builtin_inff();
    }
    if (lane == 0) p.out_count[qi] = (int32_t) out;
}

inline size_t hnsw_lds_bytes(uint32_t caps)
{
    return (size_t) caps * 8 + ((caps + 15) & ~15u) + 64 * 4 + 64 * 4 + (size_t) HN_UPPER_VISITED * 4;
}

inline hipError_t launch_hnsw_search(const HnswParams& p, uint32_t nq, hipStream_t s)
{
    if (nq == 0) return hipSuccess;
    const size_t lds = hnsw_lds_bytes(p.caps);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hnsw_search_kernel, dim3(nq), dim3(64), lds, s, p);
    return hipGetLastError();
}

}  // namespace vsr
