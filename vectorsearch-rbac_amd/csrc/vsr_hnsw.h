// vsr_hnsw.h — K4: HNSW layer search on the GPU, one wave per query, up to four queries per workgroup.
//
// Replaces HnswSearchLayer (pgvector/src/hnswutils.c:813-976) as hnswgettuple's first call drives it (GetScanItems,
// hnswscan.c:15-45): greedy descent with ef = 1 through the upper layers, then the ef_search beam on layer 0, then the
// heap TIDs of the result elements nearest first (newest TID of an element first, hnswscan.c:278-311), the permission bit
// of every row (the executor's RLS filter above the index scan) and the first k.
//
// Candidates are totally ordered by key = (monotone fp32 index distance << 32) | element id: a fixed choice where
// pgvector's pairing heaps leave candidates of equal distance to insertion history (the tests' CPU checker makes the
// same choice).  Both of Algorithm 2's sets live in ONE sorted array S in LDS: every element ever pushed, with an
// "expanded" flag.  W (the ef best found) is S's first min(pushed, ef) entries, C (still to expand) its unexpanded
// entries.  An expansion reads the neighbour list (2m ids on layer 0, up to 200: HNSW_MAX_M = 100, hnsw.h:40), marks them
// in the query's visited set, computes the distances of the unvisited ones with 8 row gathers in flight per wave (half a
// wave per row) and then applies the admission rule  d < furthest(W) || |W| < ef  to them one by one in list order,
// exactly like the sequential loop (the order matters for ties).
//
// Visited set of layer 0 (the reference: a hash table of TIDs, hnswutils.c:662-699), by size of the graph:
//   VIS_LDS_BITMAP  n_elem / 8 bytes of LDS per query: exact, no global traffic at all (graphs up to ~1M elements)
//   VIS_LDS_HASH    open-addressing table in LDS sized by ef (not by the graph): a query that fills it beyond 3/4 reports
//                   status 1 and is re-run by the host entry point with the global bitmap
//   VIS_GLOBAL      one bitmap per query in global memory (round 2's only form; n_elem / 8 bytes per query, cleared per call)
// Upper layers (ef = 1) keep a short visited list in LDS.
//
// predicate_aware (vsr_hnsw_set_predicate_aware; off by default = pgvector's behaviour, where the executor filters what the
// index returns): the layer-0 walk itself applies the query's permission bitmap, in the manner of ACORN-1 (Patel et al., the
// library acorn_benchmark/src/acorn_search.cpp:144-181 calls; its source is not in the reference tree, so this variant is
// pinned by the tests' CPU restatement of THIS walk and by recall against the exact filtered scan, not by ACORN's code):
// W and C hold permitted elements only; expanding c takes c's unvisited neighbours that are permitted, and for every
// unvisited neighbour that is NOT permitted its permitted unvisited neighbours (two hops), in list order, up to HN_NBR
// candidates per expansion.  The descent through the upper layers and the entry point are unfiltered.
#pragma once
#include "vsr_device.h"
#include "vsr_topk.h"

namespace vsr {

enum HnswVisited : int { VIS_LDS_BITMAP = 0, VIS_LDS_HASH = 1, VIS_GLOBAL = 2 };

struct HnswParams {
    const float4*   rows = nullptr;          // base corpus rows (internal order)
    uint32_t        stride4 = 0;
    int             metric = 0;              // M_L2 / M_IP / M_COSINE (unit rows: ranked by negative inner product)
    const float*    queries = nullptr;       // [nq][q_stride] floats (q_stride >= dim; elements past dim are not read)
    uint32_t        q_stride = 0, dim = 0, nq = 0;
    uint32_t        n_elem = 0;
    int32_t         entry = -1, entry_level = -1;
    uint32_t        m = 0, max_level = 0;
    const int32_t*  elem_row = nullptr;      // element -> internal row holding its vector
    const int32_t*  nbr0 = nullptr;          // [n_elem][2m], -1 padded
    const int32_t*  up_slot = nullptr;       // element -> slot in up_nbr or -1
    const int32_t*  up_nbr = nullptr;        // [n_upper][max_level][m], -1 padded
    const int32_t*  level = nullptr;         // element -> top level
    const int32_t*  tid_count = nullptr;     // element -> heap TIDs (<= 10)
    const int32_t*  tids = nullptr;          // [n_elem][10] internal rows
    const uint64_t* const* bitmaps = nullptr;  // per query: permission bitmap over internal rows, or nullptr
    int             predicate_aware = 0;     // 1: the layer-0 walk applies the bitmap (see above); 0: only the results are filtered
    uint32_t        ef = 0, k = 0, caps = 0;   // caps: capacity of S (>= ef + 2m)
    int             vis_mode = VIS_GLOBAL;
    uint32_t        vis_words = 0;           // LDS bitmap / global bitmap: 32-bit words per query; LDS hash: slots (power of two)
    uint32_t*       visited = nullptr;       // VIS_GLOBAL: [nq][vis_words], zero on entry
    uint32_t        qpb = 1;                 // queries (waves) per workgroup
    uint32_t        lds_per_query = 0;       // bytes
    const int64_t*  block_ids = nullptr; const int32_t* doc_ids = nullptr; const int64_t* orig_rows = nullptr;
    int64_t* out_block = nullptr; int32_t* out_doc = nullptr; int64_t* out_row = nullptr; float* out_dist = nullptr; int32_t* out_count = nullptr;
    int64_t*        out_visited = nullptr;   // optional [nq]: elements entered into the visited set on layer 0
    int32_t*        out_status = nullptr;    // optional [nq]: 0 = ok, 1 = the LDS hash overflowed (result not valid: re-run with VIS_GLOBAL)
    uint32_t*       err = nullptr;
};

constexpr int HN_UPPER_VISITED = 1024;     // visited list of an upper-layer (ef = 1) search, in LDS
constexpr int HN_NBR = 256;                // neighbour ids of one expansion (2m <= 200)

__device__ __forceinline__ float hnsw_rank_value(int metric, float s) { return metric == M_L2 ? s : -s; }

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(256) void hnsw_search_kernel(const HnswParams p)
{
    extern __shared__ __align__(16) unsigned char smem_all[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t qi = blockIdx.x * p.qpb + wave;
    if (wave >= p.qpb || qi >= p.nq) return;                                 // (waves are independent: no workgroup barrier below)
    unsigned char* smem = smem_all + (size_t) wave * p.lds_per_query;
    uint64_t* S = reinterpret_cast<uint64_t*>(smem);                         // [caps] sorted keys
    uint8_t*  X = reinterpret_cast<uint8_t*>(S + p.caps);                    // [caps] expanded flags
    int32_t*  nb = reinterpret_cast<int32_t*>(X + ((p.caps + 15) & ~15u));   // [HN_NBR] neighbour ids of the expansion
    float*    nd = reinterpret_cast<float*>(nb + HN_NBR);                    // [HN_NBR] their distances
    int32_t*  uv = reinterpret_cast<int32_t*>(nd + HN_NBR);                  // [HN_UPPER_VISITED] upper-layer visited list
    uint32_t* lv = reinterpret_cast<uint32_t*>(uv + HN_UPPER_VISITED);       // layer-0 visited set: bitmap words or hash slots
    const float* q = p.queries + (size_t) qi * p.q_stride;
    uint32_t* gvis = p.vis_mode == VIS_GLOBAL ? p.visited + (size_t) qi * p.vis_words : nullptr;
    const int half = lane >> 5, hl = lane & 31;
    const uint32_t d4 = (p.dim + 3) / 4;
    bool overflow = false;
    uint32_t hash_used = 0;

    if (p.vis_mode != VIS_GLOBAL) {
        for (uint32_t i = (uint32_t) lane; i < p.vis_words; i += 64) lv[i] = 0u;
        wave_sync();
    }
    // mark element e visited; true when it was not yet (lanes of the wave call this concurrently)
    auto visit = [&](uint32_t e) -> bool {
        if (p.vis_mode == VIS_GLOBAL) {
            const uint32_t old = atomicOr(&gvis[e >> 5], 1u << (e & 31));
            return !((old >> (e & 31)) & 1u);
        }
        if (p.vis_mode == VIS_LDS_BITMAP) {
            const uint32_t old = atomicOr(&lv[e >> 5], 1u << (e & 31));
            return !((old >> (e & 31)) & 1u);
        }
        const uint32_t mask = p.vis_words - 1u;
        uint32_t h = (e * 2654435761u) >> 7;
        for (uint32_t probe = 0; probe <= mask; ++probe, ++h) {
            const uint32_t old = atomicCAS(&lv[h & mask], 0u, e + 1u);
            if (old == 0u) return true;
            if (old == e + 1u) return false;
        }
        return false;                                                        // table full: reported through `overflow` below
    };

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
            for (uint32_t ch = hl; ch < d4; ch += 32) {
                float4 b;
                if (4 * ch + 3 < p.dim && (p.q_stride & 3u) == 0) b = *reinterpret_cast<const float4*>(q + 4 * ch);
                else {
                    b.x = 4 * ch < p.dim ? q[4 * ch] : 0.0f;
                    b.y = 4 * ch + 1 < p.dim ? q[4 * ch + 1] : 0.0f;
                    b.z = 4 * ch + 2 < p.dim ? q[4 * ch + 2] : 0.0f;
                    b.w = 4 * ch + 3 < p.dim ? q[4 * ch + 3] : 0.0f;
                }
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
        wave_sync();
    };

    const uint64_t* bm = p.bitmaps ? p.bitmaps[qi] : nullptr;
    const bool pa = p.predicate_aware && bm != nullptr;
    // element e is permitted when one of its heap rows is
    auto permitted = [&](uint32_t e) -> bool {
        const uint32_t nt = (uint32_t) p.tid_count[e];
        for (uint32_t t = 0; t < nt; ++t) {
            const uint32_t row = (uint32_t) p.tids[(size_t) e * 10 + t];
            if ((bm[row >> 6] >> (row & 63)) & 1ull) return true;
        }
        return false;
    };
    uint32_t count = 0;                      // entries of S
    uint32_t pushed = 0;                     // wlen of the reference: pushes so far (never decremented)
    uint32_t first_open = 0;                 // every entry before this position is expanded
    // insert (key, unexpanded) into S keeping it sorted; beyond caps the largest entry falls off
    auto insert = [&](uint64_t key) {
        // position = number of keys < key.  A binary search is a chain of log2(count) dependent LDS reads (eleven at ef_search
        // in the thousands: over a microsecond per insert); the wave does a 64-ary search instead: lane l compares the LAST key of
        // segment l (count / 64 keys, rounded up), the segments entirely below the key are a prefix, and the one segment that is
        // left is compared key by key -- two or three LDS round trips whatever the count.
        uint32_t pos = 0;
        if (count) {
            const uint32_t seg_len = (count + 63u) / 64u;
            const uint32_t s0 = (uint32_t) lane * seg_len;
            const uint32_t s1 = s0 + seg_len < count ? s0 + seg_len : count;
            const bool has = s0 < count;
            const uint64_t pivot = has ? S[s1 - 1] : KEY_EMPTY;
            const uint32_t seg = (uint32_t) __popcll(__ballot(has && pivot < key));      // segments whose every key is < key
            const uint32_t b0 = seg * seg_len;
            const uint32_t b1 = b0 + seg_len < count ? b0 + seg_len : count;
            pos = b0 < count ? b0 : count;
            for (uint32_t o = b0; o < b1; o += 64) {
                const uint32_t i = o + (uint32_t) lane;
                pos += (uint32_t) __popcll(__ballot(i < b1 && S[i] < key));
            }
        }
        if (pos >= p.caps) return;
        const uint32_t last = count < p.caps ? count : p.caps - 1;      // index the shifted tail ends at
        // the tail [pos, last) moves up by one, 256 entries per step from the end (four per lane: a step is two wave barriers
        // whatever it moves, and at ef_search in the thousands the tail is a thousand entries long)
        for (int64_t base = (int64_t) ((last - 1) & ~255u); last > pos && base >= (int64_t) (pos & ~255u); base -= 256) {
            const uint32_t i0 = (uint32_t) base + 4u * (uint32_t) lane;
            uint64_t kk[4];
            uint8_t xx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + (uint32_t) u;
                const bool mv = i >= pos && i < last;
                kk[u] = mv ? S[i] : 0;
                xx[u] = mv ? X[i] : 0;
            }
            wave_sync();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i = i0 + (uint32_t) u;
                if (i >= pos && i < last) { S[i + 1] = kk[u]; X[i + 1] = xx[u]; }
            }
            wave_sync();
        }
        if (lane == 0) { S[pos] = key; X[pos] = 0; }
        if (count < p.caps) ++count;
        if (pos < first_open) first_open = pos;
        wave_sync();
    };

    int64_t visited_l0 = 0;
    // Algorithm 2 on layer lc with beam ef_; S holds the entry points (unexpanded) on entry and W (sorted) on exit
    auto search_layer = [&](int lc, uint32_t ef_) {
        const uint32_t lm = lc == 0 ? 2 * p.m : p.m;
        uint32_t n_uv = 0;
        // entry points count as visited
        if (lc == 0) {
            for (uint32_t i = (uint32_t) lane; i < count; i += 64) (void) visit((uint32_t) S[i]);
            visited_l0 += count;
            hash_used += count;
        } else {                             // (one entry point per upper layer)
            if (lane == 0) uv[0] = (int32_t) (uint32_t) S[0];
            n_uv = 1;
        }
        pushed = count;
        first_open = 0;
        wave_sync();
        for (;;) {
            // c = nearest unexpanded entry
            uint32_t cpos = 0xFFFFFFFFu;
            for (uint32_t base = first_open & ~63u; base < count && cpos == 0xFFFFFFFFu; base += 64) {
                const uint32_t i = base + (uint32_t) lane;
                const uint64_t mk = __ballot(i < count && i >= first_open && X[i] == 0);
                if (mk) cpos = base + (uint32_t) __ffsll((unsigned long long) mk) - 1;
            }
            if (cpos == 0xFFFFFFFFu) break;                                  // C is empty
            first_open = cpos + 1;
            const uint64_t ckey = S[cpos];
            const uint32_t wl = pushed < ef_ ? pushed : ef_;                   // |W|
            const uint64_t fkey = S[(wl < count ? wl : count) - 1];
            if (mono_to_float((uint32_t) (ckey >> 32)) > mono_to_float((uint32_t) (fkey >> 32))) break;
            if (lane == 0) X[cpos] = 1;
            const uint32_t ce = (uint32_t) ckey;
            // neighbour list of c on this layer, 64 ids at a time, unvisited ones kept in list order
            int cnt = 0;
            const int32_t* nl;
            if (lc == 0) nl = p.nbr0 + (size_t) ce * 2 * p.m;
            else {
                const int32_t slot = p.up_slot[ce];
                nl = slot >= 0 ? p.up_nbr + ((size_t) slot * p.max_level + (uint32_t) (lc - 1)) * p.m : nullptr;
            }
            if (pa && lc == 0) {
                // ---- predicate-aware expansion: permitted neighbours, then the permitted neighbours of the others ----
                int32_t* hop = reinterpret_cast<int32_t*>(nd);               // (nd is free until distances(): <= 2m <= 200 ids)
                int n_hop = 0;
                uint32_t marked = 0;
                for (uint32_t j0 = 0; j0 < lm; j0 += 64) {
                    const uint32_t j = j0 + (uint32_t) lane;
                    const int32_t my = j < lm ? nl[j] : -1;
                    const bool fresh = my >= 0 && visit((uint32_t) my);
                    const bool okp = fresh && permitted((uint32_t) my);
                    const uint64_t fm = __ballot(okp), hm = __ballot(fresh && !okp);
                    if (okp) nb[cnt + __popcll(fm & ((1ull << lane) - 1ull))] = my;
                    if (fresh && !okp) hop[n_hop + __popcll(hm & ((1ull << lane) - 1ull))] = my;
                    cnt += __popcll(fm);
                    n_hop += __popcll(hm);
                    marked += (uint32_t) __popcll(fm | hm);
                    wave_sync();
                }
                for (int h = 0; h < n_hop && cnt < HN_NBR; ++h) {
                    const int32_t* nl2 = p.nbr0 + (size_t) (uint32_t) hop[h] * 2 * p.m;
                    for (uint32_t j0 = 0; j0 < lm && cnt < HN_NBR; j0 += 64) {
                        const uint32_t j = j0 + (uint32_t) lane;
                        const int32_t my = j < lm ? nl2[j] : -1;
                        const bool ok2 = my >= 0 && permitted((uint32_t) my) && visit((uint32_t) my);
                        const uint64_t fm = __ballot(ok2);
                        const int at = cnt + __popcll(fm & ((1ull << lane) - 1ull));
                        if (ok2 && at < HN_NBR) nb[at] = my;                 // (beyond the cap: marked visited, dropped)
                        marked += (uint32_t) __popcll(fm);
                        cnt += __popcll(fm);
                        if (cnt > HN_NBR) cnt = HN_NBR;
                        wave_sync();
                    }
                }
                visited_l0 += marked;
                hash_used += marked;
                if (p.vis_mode == VIS_LDS_HASH && hash_used * 4u > p.vis_words * 3u) { overflow = true; break; }
            } else {
            for (uint32_t j0 = 0; j0 < lm && nl; j0 += 64) {
                const uint32_t j = j0 + (uint32_t) lane;
                const int32_t my = j < lm ? nl[j] : -1;
                bool fresh = false;
                if (my >= 0) {
                    if (lc == 0) fresh = visit((uint32_t) my);
                    else {
                        fresh = true;
                        for (uint32_t u = 0; u < n_uv; ++u) fresh &= uv[u] != my;
                    }
                }
                const uint64_t fm = __ballot(fresh);
                const int add = __popcll(fm);
                if (fresh) {
                    const int at = cnt + __popcll(fm & ((1ull << lane) - 1ull));
                    nb[at] = my;
                    if (lc != 0) {
                        const uint32_t ua = n_uv + (uint32_t) (at - cnt);
                        if (ua < (uint32_t) HN_UPPER_VISITED) uv[ua] = my;
                        else atomicOr(p.err, 8u);                           // cannot happen at ef = 1
                    }
                }
                if (lc != 0) n_uv += (uint32_t) add;
                cnt += add;
                wave_sync();
            }
            if (lc == 0) {
                visited_l0 += cnt;
                hash_used += (uint32_t) cnt;
                if (p.vis_mode == VIS_LDS_HASH && hash_used * 4u > p.vis_words * 3u) { overflow = true; break; }
            }
            }
            if (cnt == 0) continue;
            distances(cnt);
            for (int i = 0; i < cnt; ++i) {                                  // the sequential admission of Algorithm 2
                const uint32_t e = (uint32_t) nb[i];
                const float ed = nd[i];
                const bool always = pushed < ef_;
                const uint32_t wl2 = pushed < ef_ ? pushed : ef_;
                const float fd = mono_to_float((uint32_t) (S[(wl2 < count ? wl2 : count) - 1] >> 32));
                if (!(ed < fd || always)) continue;
                if (lc != 0 && p.level[e] < lc) continue;                    // (every element lives on layer 0)
                insert(make_key(ed, e));
                ++pushed;
            }
        }
        const uint32_t wl = pushed < ef_ ? pushed : ef_;
        count = wl < count ? wl : count;                                     // S = W, nearest first
    };

    auto write_empty = [&](uint32_t from) {
        for (uint32_t i = from + (uint32_t) lane; i < p.k; i += 64) {
            const size_t o = (size_t) qi * p.k + i;
            p.out_block[o] = -1; p.out_doc[o] = -1;
            if (p.out_row) p.out_row[o] = -1;
            p.out_dist[o] = __builtin_inff();
        }
    };
    if (p.out_status && lane == 0) p.out_status[qi] = 0;
    if (p.entry < 0) {
        if (lane == 0) p.out_count[qi] = 0;
        write_empty(0);
        return;
    }
    // HnswEntryCandidate
    if (lane == 0) nb[0] = p.entry;
    wave_sync();
    distances(1);
    insert(make_key(nd[0], (uint32_t) p.entry));
    for (int lc = p.entry_level; lc >= 1; --lc) {
        search_layer(lc, 1);
        if (lane == 0)
            for (uint32_t i = 0; i < count; ++i) X[i] = 0;                    // W becomes the next layer's entry points
        wave_sync();
    }
    search_layer(0, p.ef);
    if (p.out_visited && lane == 0) p.out_visited[qi] = visited_l0;
    if (overflow) {                                                          // the visited table filled up: no result
        if (lane == 0) {
            p.out_count[qi] = -1;
            if (p.out_status) p.out_status[qi] = 1;
        }
        write_empty(0);
        return;
    }

    // hnswgettuple: elements nearest first, their heap TIDs newest first, the permission bit, the first k
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
    write_empty(out);
    if (lane == 0) p.out_count[qi] = (int32_t) out;
}

constexpr size_t HN_LDS_BUDGET = 156 * 1024;   // of the CU's 160 KB (dynamic LDS of one workgroup)

inline size_t hnsw_lds_fixed(uint32_t caps)
{
    return (((size_t) caps * 8 + ((caps + 15) & ~15u) + (size_t) HN_NBR * 8 + (size_t) HN_UPPER_VISITED * 4) + 15) & ~(size_t) 15;
}

// visited form, table size, queries per workgroup for a graph of n_elem elements searched with ef (see the header comment);
// force_global: the re-run of queries whose LDS hash overflowed.  false: ef too large for the LDS
inline bool hnsw_plan(HnswParams& p, bool force_global)
{
    const size_t fixed = hnsw_lds_fixed(p.caps);
    if (fixed > HN_LDS_BUDGET) return false;
    const size_t bitmap_bytes = (((size_t) p.n_elem + 31) / 32) * 4;
    const size_t room = HN_LDS_BUDGET - fixed;
    size_t vis_bytes = 0;
    if (force_global) {
        p.vis_mode = VIS_GLOBAL;
        p.vis_words = (uint32_t) (bitmap_bytes / 4);
    } else if (bitmap_bytes <= room && bitmap_bytes <= 128 * 1024) {
        p.vis_mode = VIS_LDS_BITMAP;
        p.vis_words = (uint32_t) (bitmap_bytes / 4);
        vis_bytes = bitmap_bytes;
    } else {
        size_t want = 2048;                                                  // slots: ~4/3 x (48 visits per unit of ef + slack), power of two
        while (want < (size_t) p.ef * 64 + 4096) want <<= 1;
        while (want * 4 > room && want > 1024) want >>= 1;
        if (want * 4 > room) {
            p.vis_mode = VIS_GLOBAL;
            p.vis_words = (uint32_t) (bitmap_bytes / 4);
        } else {
            p.vis_mode = VIS_LDS_HASH;
            p.vis_words = (uint32_t) want;
            vis_bytes = want * 4;
        }
    }
    p.lds_per_query = (uint32_t) ((fixed + vis_bytes + 15) & ~(size_t) 15);
    const size_t fit = HN_LDS_BUDGET / p.lds_per_query;
    // several waves per workgroup only while that does not cost residency: ~40 KB per workgroup keeps 4 workgroups per CU
    p.qpb = (uint32_t) (fit >= 4 && (size_t) p.lds_per_query * 4 <= 40 * 1024 ? 4 : fit >= 2 && (size_t) p.lds_per_query * 2 <= 40 * 1024 ? 2 : 1);
    return true;
}

inline hipError_t launch_hnsw_search(const HnswParams& p, hipStream_t s)
{
    if (p.nq == 0) return hipSuccess;
    const size_t lds = (size_t) p.lds_per_query * p.qpb;
    if (lds > HN_LDS_BUDGET || p.qpb < 1 || p.qpb > 4) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(hnsw_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(hnsw_search_kernel, dim3((p.nq + p.qpb - 1) / p.qpb), dim3(64 * p.qpb), lds, s, p);
    return hipGetLastError();
}

}  // namespace vsr
