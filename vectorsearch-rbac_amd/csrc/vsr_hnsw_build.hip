// vsr_hnsw_build.hip — CREATE INDEX ... USING hnsw on the GPU: batched insertion (pgvector/src/hnswbuild.c:357-470,
// hnswutils.c:813-1346).
//
// pgvector inserts one element at a time: HnswFindElementNeighbors searches the graph as it stands (greedy descent, then
// HnswSearchLayer with ef_construction on every layer the element lives on), SelectNeighbors picks up to m (2m on layer 0)
// of the candidates with the "closer to the query than to any neighbour already selected" heuristic, and
// HnswUpdateConnection adds the reverse edge to every selected neighbour, pruning that neighbour's list with the same
// heuristic when it is full.  Its own parallel build runs several such insertions at once against a graph the others are
// changing under it, so the graph is not reproducible even in the reference; what it guarantees (and tests:
// test/t/012_hnsw_vector_build_recall.pl) is recall.
//
// The GPU build inserts BATCHES: every element of a batch searches the graph as it stood when the batch began (one wave per
// element, vsr_hnsw.h's search machinery), selects its neighbours and writes its own lists; the reverse edges of the whole
// batch are then sorted by (layer, target) and one wave per target applies its run of HnswUpdateConnection calls one after
// another.  Elements of one batch do not see each other, so a batch is never larger than 1/8 of the graph it is inserted
// into (the first elements go in one by one).  Levels come from the same seeded xorshift64* stream as the test suite's serial CPU restatement
// (level = floor(-ln(u) / ln(m)), hnswutils.c:243), so both builds place the same elements on the same layers.
// Not done: the merging of identical vectors into one element's heap TIDs (hnswbuild.c:329-351) -- every row is its own
// element, which changes nothing a search returns.
//
// Parity definition (tests/test_gpu_index.py): recall@20 at ef_search = 40 against the exact scan >= pgvector's TAP
// thresholds on its own case (10 000 x 3-d: 0.99; inner product 0.97), and within 0.01 of that serial build's recall on
// the same rows.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "vsr_device.h"
#include "vsr_topk.h"
#include "vsr_hnsw_build.h"

namespace vsr {

__device__ __forceinline__ float hnsw_rank_value(int metric, float s) { return metric == M_L2 ? s : -s; }
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ int32_t* hb_list(const HnswBuildParams& p, uint32_t e, int lc, float** dist)
{
    if (lc == 0) {
        *dist = p.dist0 + (size_t) e * 2 * p.m;
        return p.nbr0 + (size_t) e * 2 * p.m;
    }
    const size_t at = ((size_t) p.up_slot[e] * p.max_level + (uint32_t) (lc - 1)) * p.m;
    *dist = p.up_dist + at;
    return p.up_nbr + at;
}

// distance (the opclass's index distance: squared L2 or negative inner product) of row `a` to `cnt` elements listed in ids[]
// -> out[]; two elements per wave instruction (32 lanes each), fp32 sums
__device__ __forceinline__ void hb_distances(const HnswBuildParams& p, const float4* a, const int32_t* ids, int cnt, float* out, int lane)
{
    const int half = lane >> 5, hl = lane & 31;
    for (int c0 = 0; c0 < cnt; c0 += 8) {
        float s[4];
        const float4* b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 2 * u + half;
            s[u] = 0.0f;
            b[u] = p.rows + (size_t) (c < cnt ? ids[c] : ids[0]) * p.stride4;
        }
        for (uint32_t ch = (uint32_t) hl; ch < p.stride4; ch += 32) {
            const float4 av = a[ch];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 bv = b[u][ch];
                if (p.metric == M_L2) {
                    const float d0 = av.x - bv.x, d1 = av.y - bv.y, d2 = av.z - bv.z, d3 = av.w - bv.w;
                    s[u] = fmaf(d0, d0, s[u]); s[u] = fmaf(d1, d1, s[u]); s[u] = fmaf(d2, d2, s[u]); s[u] = fmaf(d3, d3, s[u]);
                } else {
                    s[u] = fmaf(av.x, bv.x, s[u]); s[u] = fmaf(av.y, bv.y, s[u]); s[u] = fmaf(av.z, bv.z, s[u]); s[u] = fmaf(av.w, bv.w, s[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            for (int mm = 16; mm >= 1; mm >>= 1) s[u] += __shfl_xor(s[u], mm);
            const int c = c0 + 2 * u + half;
            if (hl == 0 && c < cnt) out[c] = hnsw_rank_value(p.metric, s[u]);
        }
    }
    wave_sync();
}

// SelectNeighbors (hnswutils.c:1053-1154): cand[0 .. n) = (distance to the owner, element) nearest first; keeps up to lm of them
// in sel[] (indices into cand, selection order).  A candidate is taken when it is closer to the owner than to every
// neighbour already taken (CheckElementCloser); when fewer than lm pass, the passed-over ones fill up in order ("keep
// pruned connections").  *pruned = index of the candidate pgvector would call pruned (only meaningful for n > lm).
// scratch: ids[lm], nd[lm] in LDS.  Returns the number selected.
__device__ __forceinline__ int hb_select(const HnswBuildParams& p, const uint64_t* cand, int n, int lm, int32_t* sel, int32_t* wd,
                                         int32_t* ids, float* nd, int* pruned, int lane)
{
    if (n <= lm) {
        if (lane < n) sel[lane] = lane;
        for (int i = 64 + lane; i < n; i += 64) sel[i] = i;
        *pruned = -1;
        wave_sync();
        return n;
    }
    int nsel = 0, nwd = 0, i = 0;
    for (; i < n && nsel < lm; ++i) {
        const uint32_t e = (uint32_t) cand[i];
        const float de = mono_to_float((uint32_t) (cand[i] >> 32));
        bool closer = true;
        if (nsel > 0) {
            hb_distances(p, p.rows + (size_t) e * p.stride4, ids, nsel, nd, lane);
            bool bad = false;
            for (int j = lane; j < nsel; j += 64) bad |= nd[j] <= de;
            closer = __ballot(bad) == 0;
        }
        if (lane == 0) {
            if (closer) { sel[nsel] = i; ids[nsel] = (int32_t) e; }
            else wd[nwd] = i;
        }
        if (closer) ++nsel; else ++nwd;
        wave_sync();
    }
    int wdoff = 0;
    while (wdoff < nwd && nsel < lm) {
        if (lane == 0) sel[nsel] = wd[wdoff];
        ++nsel;
        ++wdoff;
    }
    wave_sync();
    // pruned: the next passed-over candidate, or (none left) the furthest candidate still in the working list
    *pruned = wdoff < nwd ? wd[wdoff] : n - 1;
    return nsel;
}

// ---- phase 1: one wave per new element: search the frozen graph, select, write the element's own lists, emit reverse edges ----
__global__ __launch_bounds__(256) void hnsw_build_search_kernel(const HnswBuildParams p)
{
    extern __shared__ __align__(16) unsigned char smem_all[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t bi = blockIdx.x * p.wpb + wave;
    if (wave >= p.wpb || bi >= p.count) return;                               // (waves are independent: no workgroup barrier)
    const uint32_t me = p.first + bi;
    unsigned char* smem = smem_all + (size_t) wave * p.lds_per_wave;
    uint64_t* S = reinterpret_cast<uint64_t*>(smem);                          // [caps] candidates, sorted, nearest first
    uint8_t*  X = reinterpret_cast<uint8_t*>(S + p.caps);                     // [caps] expanded flags
    int32_t*  nb = reinterpret_cast<int32_t*>(X + ((p.caps + 15) & ~15u));    // [HB_NBR] neighbour ids of an expansion / selected ids
    float*    nd = reinterpret_cast<float*>(nb + HB_NBR);                     // [HB_NBR] distances
    int32_t*  sel = reinterpret_cast<int32_t*>(nd + HB_NBR);                  // [HB_NBR] selected candidate indices
    int32_t*  wd = sel + HB_NBR;                                              // [caps] passed-over candidate indices
    uint32_t* lv = reinterpret_cast<uint32_t*>(wd + p.caps);                  // [hash_slots] visited set
    const float4* q = p.rows + (size_t) me * p.stride4;
    const int my_level = p.level[me];

    // the element's own lists start empty
    for (int lc = 0; lc <= my_level; ++lc) {
        float* dl;
        int32_t* nl = hb_list(p, me, lc, &dl);
        const uint32_t lm = lc == 0 ? 2 * p.m : p.m;
        for (uint32_t j = (uint32_t) lane; j < lm; j += 64) { nl[j] = -1; dl[j] = 0.0f; }
    }
    if (p.entry < 0) return;                                                  // the very first element

    bool overflow = false;
    uint32_t used = 0;
    auto clear_visited = [&]() {
        for (uint32_t i = (uint32_t) lane; i < p.hash_slots; i += 64) lv[i] = 0u;
        used = 0;
        wave_sync();
    };
    auto visit = [&](uint32_t e) -> bool {
        const uint32_t mask = p.hash_slots - 1u;
        uint32_t h = (e * 2654435761u) >> 7;
        for (uint32_t probe = 0; probe <= mask; ++probe, ++h) {
            const uint32_t old = atomicCAS(&lv[h & mask], 0u, e + 1u);
            if (old == 0u) return true;
            if (old == e + 1u) return false;
        }
        return false;
    };
    uint32_t count = 0, pushed = 0, first_open = 0;
    auto insert = [&](uint64_t key) {
        uint32_t lo = 0, hi = count;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (S[mid] < key) lo = mid + 1; else hi = mid;
        }
        const uint32_t pos = lo;
        if (pos >= p.caps) return;
        const uint32_t last = count < p.caps ? count : p.caps - 1;
        for (int64_t base = (int64_t) ((last - 1) & ~63u); last > pos && base >= (int64_t) (pos & ~63u); base -= 64) {
            const uint32_t i = (uint32_t) base + (uint32_t) lane;
            const bool mv = i >= pos && i < last;
            const uint64_t kk = mv ? S[i] : 0;
            const uint8_t xx = mv ? X[i] : 0;
            wave_sync();
            if (mv) { S[i + 1] = kk; X[i + 1] = xx; }
            wave_sync();
        }
        if (lane == 0) { S[pos] = key; X[pos] = 0; }
        if (count < p.caps) ++count;
        if (pos < first_open) first_open = pos;
        wave_sync();
    };
    // HnswSearchLayer (hnswutils.c:813-976) on layer lc with beam ef_: S holds the entry points on entry, W on exit
    auto search_layer = [&](int lc, uint32_t ef_) {
        const uint32_t lm = lc == 0 ? 2 * p.m : p.m;
        clear_visited();
        for (uint32_t i = (uint32_t) lane; i < count; i += 64) (void) visit((uint32_t) S[i]);
        used = count;
        pushed = count;
        first_open = 0;
        wave_sync();
        for (;;) {
            uint32_t cpos = 0xFFFFFFFFu;
            for (uint32_t base = first_open & ~63u; base < count && cpos == 0xFFFFFFFFu; base += 64) {
                const uint32_t i = base + (uint32_t) lane;
                const uint64_t mk = __ballot(i < count && i >= first_open && X[i] == 0);
                if (mk) cpos = base + (uint32_t) __ffsll((unsigned long long) mk) - 1;
            }
            if (cpos == 0xFFFFFFFFu) break;
            first_open = cpos + 1;
            const uint64_t ckey = S[cpos];
            const uint32_t wl = pushed < ef_ ? pushed : ef_;
            const uint64_t fkey = S[(wl < count ? wl : count) - 1];
            if (mono_to_float((uint32_t) (ckey >> 32)) > mono_to_float((uint32_t) (fkey >> 32))) break;
            if (lane == 0) X[cpos] = 1;
            const uint32_t ce = (uint32_t) ckey;
            float* dl;
            const int32_t* nl = hb_list(p, ce, lc, &dl);
            int cnt = 0;
            for (uint32_t j0 = 0; j0 < lm; j0 += 64) {
                const uint32_t j = j0 + (uint32_t) lane;
                const int32_t my = j < lm ? nl[j] : -1;
                const bool fresh = my >= 0 && visit((uint32_t) my);
                const uint64_t fm = __ballot(fresh);
                if (fresh) nb[cnt + __popcll(fm & ((1ull << lane) - 1ull))] = my;
                cnt += __popcll(fm);
                wave_sync();
            }
            used += (uint32_t) cnt;
            if (used * 4u > p.hash_slots * 3u) { overflow = true; break; }
            if (cnt == 0) continue;
            hb_distances(p, q, nb, cnt, nd, lane);
            for (int i = 0; i < cnt; ++i) {
                const uint32_t e = (uint32_t) nb[i];
                const float ed = nd[i];
                const bool always = pushed < ef_;
                const uint32_t wl2 = pushed < ef_ ? pushed : ef_;
                const float fd = mono_to_float((uint32_t) (S[(wl2 < count ? wl2 : count) - 1] >> 32));
                if (!(ed < fd || always)) continue;
                insert(make_key(ed, e));
                ++pushed;
            }
        }
        const uint32_t wl = pushed < ef_ ? pushed : ef_;
        count = wl < count ? wl : count;
    };

    // HnswFindElementNeighbors (hnswutils.c:1270-1346)
    if (lane == 0) nb[0] = p.entry;
    wave_sync();
    hb_distances(p, q, nb, 1, nd, lane);
    insert(make_key(nd[0], (uint32_t) p.entry));
    for (int lc = p.entry_level; lc >= my_level + 1; --lc) {
        search_layer(lc, 1);
        if (lane == 0)
            for (uint32_t i = 0; i < count; ++i) X[i] = 0;
        wave_sync();
    }
    for (int lc = my_level < p.entry_level ? my_level : p.entry_level; lc >= 0; --lc) {
        search_layer(lc, p.efc);
        const int lm = (int) (lc == 0 ? 2 * p.m : p.m);
        int pruned;
        const int ns = hb_select(p, S, (int) count, lm, sel, wd, nb, nd, &pruned, lane);
        // AddConnections + the reverse edges HnswUpdateNeighborsInMemory will apply
        float* dl;
        int32_t* nl = hb_list(p, me, lc, &dl);
        uint32_t base = 0;
        if (lane == 0 && ns > 0) base = atomicAdd(p.rec_count, (uint32_t) ns);
        base = (uint32_t) __shfl((int) base, 0);
        for (int j = lane; j < ns; j += 64) {
            const uint64_t key = S[sel[j]];
            nl[j] = (int32_t) (uint32_t) key;
            dl[j] = mono_to_float((uint32_t) (key >> 32));
            if (base + (uint32_t) j < p.rec_cap) {
                p.rec_key[base + j] = ((uint64_t) (uint32_t) lc << 32) | (uint32_t) key;
                p.rec_val[base + j] = ((uint64_t) me << 32) | __float_as_uint(mono_to_float((uint32_t) (key >> 32)));
            }
        }
        if (lane == 0)
            for (uint32_t i = 0; i < count; ++i) X[i] = 0;                    // W is the next layer's entry set
        wave_sync();
    }
    if (overflow && lane == 0) atomicOr(p.err, 16u);
}

// ---- phase 2: reverse edges, sorted by (layer, target): the wave of a run's first record applies the whole run ----
__global__ __launch_bounds__(256) void hnsw_build_link_kernel(const HnswBuildParams p)
{
    extern __shared__ __align__(16) unsigned char smem_all[];
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (uint32_t) __builtin_amdgcn_readfirstlane((int) (threadIdx.x >> 6));
    const uint32_t ri = blockIdx.x * 4 + wave;
    const uint32_t n_rec = *p.rec_count < p.rec_cap ? *p.rec_count : p.rec_cap;
    if (ri >= n_rec) return;
    const uint64_t key = p.rec_key[ri];
    if (ri > 0 && p.rec_key[ri - 1] == key) return;                           // not the first of its run
    const int lc = (int) (key >> 32);
    const uint32_t owner = (uint32_t) key;
    const int lm = (int) (lc == 0 ? 2 * p.m : p.m);
    unsigned char* smem = smem_all + (size_t) wave * (HB_NBR * 32);
    uint64_t* cand = reinterpret_cast<uint64_t*>(smem);                       // [lm + 1] (distance, element), sorted
    int32_t*  sel = reinterpret_cast<int32_t*>(cand + HB_NBR);
    int32_t*  wd = sel + HB_NBR;
    int32_t*  ids = wd + HB_NBR;
    float*    nd = reinterpret_cast<float*>(ids + HB_NBR);
    float* dl;
    int32_t* nl = hb_list(p, owner, lc, &dl);
    for (uint32_t r = ri; r < n_rec && p.rec_key[r] == key; ++r) {            // HnswUpdateConnection, one after another
        const uint32_t e = (uint32_t) (p.rec_val[r] >> 32);
        const float de = __uint_as_float((uint32_t) p.rec_val[r]);
        // current length of the list (-1 padded)
        int len = 0;
        for (int j0 = 0; j0 < lm; j0 += 64) {
            const int j = j0 + lane;
            len += __popcll(__ballot(j < lm && nl[j] >= 0));
        }
        if (len < lm) {
            if (lane == 0) { nl[len] = (int32_t) e; dl[len] = de; }
            wave_sync();
            continue;
        }
        // full: the list and the new element, nearest first (ties: the lower element id first, the order list_sort leaves
        // with CompareCandidateDistances read from the tail)
        for (int j = lane; j < lm; j += 64) cand[j] = make_key(dl[j], (uint32_t) nl[j]);
        if (lane == 0) cand[lm] = make_key(de, e);
        wave_sync();
        const int n = lm + 1;
        for (int j = lane; j < n; j += 64) {                                  // rank sort: n <= 201
            const uint64_t kk = cand[j];
            int rank = 0;
            for (int t = 0; t < n; ++t) rank += cand[t] < kk || (cand[t] == kk && t < j);
            sel[j] = rank;
        }
        wave_sync();
        uint64_t mine[4];
        int nm = 0;
        for (int j = lane; j < n; j += 64) mine[nm++] = cand[j];
        wave_sync();
        nm = 0;
        for (int j = lane; j < n; j += 64) cand[sel[j]] = mine[nm++];
        wave_sync();
        int pruned;
        (void) hb_select(p, cand, n, lm, sel, wd, ids, nd, &pruned, lane);
        const uint32_t pe = (uint32_t) cand[pruned];
        if (pe != e) {                                                        // the new element replaces the pruned one
            for (int j = lane; j < lm; j += 64)
                if ((uint32_t) nl[j] == pe) { nl[j] = (int32_t) e; dl[j] = de; }
        }
        wave_sync();
    }
}

}  // namespace vsr

using namespace vsr;

// host side: see vsr_runtime.hip (vsr_hnsw_build) for the batch loop; these are the two launches and the sort of a batch
hipError_t vsr_hnsw_build_batch(HnswBuildParams& p, void* d_sort_tmp, size_t sort_tmp_bytes, uint64_t* d_key_alt, uint64_t* d_val_alt,
                                hipStream_t s)
{
    hipError_t e = hipMemsetAsync(p.rec_count, 0, sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    // unused record slots sort to the end
    e = hipMemsetAsync(p.rec_key, 0xFF, (size_t) p.rec_cap * sizeof(uint64_t), s);
    if (e != hipSuccess) return e;
    const size_t lds1 = (size_t) p.lds_per_wave * p.wpb;
    if (lds1 > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(hnsw_build_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds1);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(hnsw_build_search_kernel, dim3((p.count + p.wpb - 1) / p.wpb), dim3(64 * p.wpb), lds1, s, p);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (p.entry < 0) return hipSuccess;                                       // the first element has nobody to link to
    hipcub::DoubleBuffer<uint64_t> keys(p.rec_key, d_key_alt), vals(p.rec_val, d_val_alt);
    e = hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, sort_tmp_bytes, keys, vals, (int) p.rec_cap, 0, 40, s);
    if (e != hipSuccess) return e;
    HnswBuildParams q = p;
    q.rec_key = keys.Current();
    q.rec_val = vals.Current();
    hipLaunchKernelGGL(hnsw_build_link_kernel, dim3((p.rec_cap + 3) / 4), dim3(256), (size_t) 4 * HB_NBR * 32, s, q);
    return hipGetLastError();
}

size_t vsr_hnsw_build_sort_bytes(uint32_t rec_cap)
{
    size_t bytes = 0;
    hipcub::DoubleBuffer<uint64_t> keys(nullptr, nullptr), vals(nullptr, nullptr);
    (void) hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, keys, vals, (int) rec_cap, 0, 40, (hipStream_t) 0);
    return bytes;
}
