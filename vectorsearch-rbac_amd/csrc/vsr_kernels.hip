// vsr_kernels.hip — K5 (top-k select / merge) and the small support kernels of the filtered k-NN path.
#include <algorithm>
#include "vsr_device.h"
#include "vsr_topk.h"

namespace vsr {

// per-metric scan launchers live in their own translation units (vsr_scan_*.hip)
hipError_t launch_scan_l2(const ScanParams&, int, int, uint32_t, hipStream_t);
hipError_t launch_scan_ip(const ScanParams&, int, int, uint32_t, hipStream_t);
hipError_t launch_scan_cosine(const ScanParams&, int, int, uint32_t, hipStream_t);
hipError_t launch_scan_l1(const ScanParams&, int, int, uint32_t, hipStream_t);

hipError_t launch_mq_l2(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mq_ip(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mq_cosine(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mq_l1(const ScanParams&, uint32_t, hipStream_t);

hipError_t launch_mfma_l2(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mfma_ip(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mfma_cosine(const ScanParams&, uint32_t, hipStream_t);

hipError_t launch_mfmaw_l2(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mfmaw_ip(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_mfmaw_cosine(const ScanParams&, uint32_t, hipStream_t);

hipError_t launch_gemm_l2(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_gemm_ip(const ScanParams&, uint32_t, hipStream_t);
hipError_t launch_gemm_cosine(const ScanParams&, uint32_t, hipStream_t);

hipError_t launch_gemm(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s)
{
    switch (metric) {
    case M_L2:     return launch_gemm_l2(p, n_blocks, s);
    case M_IP:     return launch_gemm_ip(p, n_blocks, s);
    case M_COSINE: return launch_gemm_cosine(p, n_blocks, s);
    default:       return hipErrorInvalidValue;
    }
}

hipError_t launch_mfmaw(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s)
{
    switch (metric) {
    case M_L2:     return launch_mfmaw_l2(p, n_blocks, s);
    case M_IP:     return launch_mfmaw_ip(p, n_blocks, s);
    case M_COSINE: return launch_mfmaw_cosine(p, n_blocks, s);
    default:       return hipErrorInvalidValue;
    }
}

hipError_t launch_mfma(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s)
{
    switch (metric) {
    case M_L2:     return launch_mfma_l2(p, n_blocks, s);
    case M_IP:     return launch_mfma_ip(p, n_blocks, s);
    case M_COSINE: return launch_mfma_cosine(p, n_blocks, s);
    default:       return hipErrorInvalidValue;
    }
}

bool mq_supported(int dim) { return (dim + 3) / 4 >= 16; }

int mq_qmax(int dim)
{
    // LDS per workgroup = 4 wave staging images (~70 KB) + the queries; prefer two workgroups per CU
    const size_t stride4 = (size_t) (dim + 3) / 4;
    const size_t per_query = (stride4 + 15) / 16 * 16 * 16 + 32;
    const size_t fixed = (size_t) 4 * 64 * 17 * 16 + 4 * 64 * 4 + 64;
    auto fit = [&](size_t budget) { return budget > fixed ? (int) ((budget - fixed) / per_query) : 0; };
    int q = fit(80 * 1024);
    if (q < 8) q = fit(160 * 1024);
    q = q < SCAN_QMAX ? q : SCAN_QMAX;
    return q >= 4 ? q / 4 * 4 : 0;
}

hipError_t launch_mq(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s)
{
    switch (metric) {
    case M_L2:     return launch_mq_l2(p, n_blocks, s);
    case M_IP:     return launch_mq_ip(p, n_blocks, s);
    case M_COSINE: return launch_mq_cosine(p, n_blocks, s);
    case M_L1:     return launch_mq_l1(p, n_blocks, s);
    default:       return hipErrorInvalidValue;
    }
}

KernelShape scan_shape_for_dim(int dim)
{
    const int d4 = (dim + 3) / 4;
    if (d4 <= 1)   return {1, 1, 1, 64};
    if (d4 <= 4)   return {4, 1, 4, 64};
    if (d4 <= 16)  return {16, 1, 8, 32};
    if (d4 <= 32)  return {32, 1, 8, 16};
    if (d4 <= 64)  return {64, 1, 8, 8};
    if (d4 <= 128) return {64, 2, 4, 4};
    if (d4 <= 192) return {64, 3, 4, 4};
    if (d4 <= 256) return {64, 4, 2, 2};
    return {64, 0, 2, 2};
}

uint32_t scan_cap_for_k(int k, int dim)
{
    // room for k kept keys plus one check interval of new ones, trigger level >= 2k so compactions amortise
    const int slack = scan_slack(scan_shape_for_dim(dim).rw);
    uint32_t cap = 512;
    while (cap < (uint32_t) (2 * k + slack)) cap <<= 1;
    return cap;
}

int scan_qmax(int dim, int k)
{
    const KernelShape sh = scan_shape_for_dim(dim);
    const uint32_t stride4 = (uint32_t) ((dim + 3) / 4);
    const size_t per_query = scan_lds_bytes(1, scan_cap_for_k(k, dim), stride4) - 16;
    int q = (int) (SCAN_LDS_BUDGET / per_query);
    if (sh.c == 0) q = q < 4 ? q : 4;            // runtime-chunk kernel: one sub-batch only
    if (q >= 4) q = (q < SCAN_QMAX ? q : SCAN_QMAX) / 4 * 4;
    else q = 1;
    return q;
}

hipError_t launch_scan(const ScanParams& p, int metric, int dim, int qi, uint32_t n_blocks, hipStream_t s)
{
    switch (metric) {
    case M_L2:     return launch_scan_l2(p, dim, qi, n_blocks, s);
    case M_IP:     return launch_scan_ip(p, dim, qi, n_blocks, s);
    case M_COSINE: return launch_scan_cosine(p, dim, qi, n_blocks, s);
    case M_L1:     return launch_scan_l1(p, dim, qi, n_blocks, s);
    default:       return hipErrorInvalidValue;
    }
}

// -------------------------------------------------------------------------------------------------
// K5: per query, the k smallest keys among its partial lists -> sorted output rows.
// Replaces the executor's top-N sort / the client-side merge of
// controller/dynamic_partition/search.py:347-364 (no dedup needed: a row is scanned once).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float output_distance(int metric, float v)
{
    // L2 ranks by the fp32 sum; the operator value is sqrt((double) sum), vector.c:577
    return metric == M_L2 ? (float) sqrt((double) v) : v;
}

// What a finished selection (count keys, ascending, in `keys`) turns into: a seed threshold, a partial list for the
// next stage (level-2 K5 / K5r), or the caller's output rows.  Run by NT cooperating threads (a workgroup or a wave).
template <int NT>
__device__ __forceinline__ void select_emit(const SelectParams& p, const SelectQuery& sq, const uint64_t* keys,
                                            uint32_t count, int tid)
{
    const uint32_t k = p.k;
    if (sq.dst_list == SEL_SEED) {
        // seed threshold from the sample pass: every row ranking at or before the k-th sampled candidate stays
        // eligible in the main pass (low word all ones: ties of that distance included); too few samples: no seed
        if (tid == 0) p.tau_out[sq.out_slot] = count >= k ? (keys[k - 1] | 0xFFFFFFFFull) : KEY_EMPTY;
        return;
    }
    if (sq.dst_list != SEL_FINAL) {                          // level 1 of a two-level merge / K2 survivors for K5r
        const uint32_t n = count < k ? count : k;
        uint64_t* dst = p.partial + (size_t) sq.dst_list * p.kp;
        for (uint32_t i = tid; i < p.kp; i += NT) dst[i] = i < n ? keys[i] : KEY_EMPTY;
        return;
    }

    const uint32_t m = count < k ? count : k;
    const size_t out = (size_t) sq.out_slot * k;
    for (uint32_t i = tid; i < k; i += NT) {
        if (i < m) {
            const uint64_t key = keys[i];
            const uint32_t row = (uint32_t) key;
            const float v = mono_to_float((uint32_t) (key >> 32));
            p.out_block[out + i] = p.block_ids[row];
            p.out_doc[out + i] = p.doc_ids[row];
            if (p.out_row) p.out_row[out + i] = p.orig_rows[row];
            p.out_dist[out + i] = output_distance(p.metric, v);
            if (p.out_keys) p.out_keys[out + i] = (key & 0xFFFFFFFF00000000ull) | (uint64_t) (row + p.row_offset);
        } else {
            p.out_block[out + i] = -1;
            p.out_doc[out + i] = -1;
            if (p.out_row) p.out_row[out + i] = -1;
            p.out_dist[out + i] = __builtin_inff();
            if (p.out_keys) p.out_keys[out + i] = KEY_EMPTY;
        }
    }
    if (tid == 0) {
        const bool flag = p.seeded && m < k && m < sq.allowed;   // a seeded threshold cut below the k-th result
        // a flagged query reports its count as -1 - count: a caller that never looks at the flags still cannot take
        // the unproven rows for a result (include/vsrbac.h, "flagged queries")
        p.out_count[sq.out_slot] = flag ? -1 - (int32_t) m : (int32_t) m;
        if (flag) {
            p.out_flags[sq.out_slot] = 1;
            atomicAdd(p.flagged_total, 1);
        }
    }
}

// Short candidate sets (<= 64 lists, <= 64 * R keys): ONE WAVE per query, four queries per workgroup, every key of
// the query in registers.  The k smallest are found by an MSB-first radix select (8-bit digits, histogram in LDS,
// wave-wide prefix sum): no sorting network, no workgroup barrier.  Only the caller-visible output (SEL_FINAL) is
// sorted afterwards, and then only its k keys.
constexpr uint32_t SEL_WAVE_MAX_K = 512;
template <int R>
__global__ __launch_bounds__(256) void select_radix_kernel(const SelectParams p, uint32_t n_items)
{
    __shared__ uint32_t sm_hist[4][256];
    __shared__ __align__(16) uint64_t sm_out[4][SEL_WAVE_MAX_K];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t item = blockIdx.x * 4 + (uint32_t) wave;
    if (item >= n_items) return;                             // wave-uniform; no workgroup barrier below
    const SelectQuery sq = p.queries[item];
    uint32_t* hist = sm_hist[wave];
    const uint32_t kp = p.kp, k = p.k;
    const uint32_t total = sq.n_lists * kp;                  // <= 64 * R (launcher contract)
    const uint32_t my_list = (uint32_t) lane < sq.n_lists ? p.list_ids[sq.ids_begin + (uint32_t) lane] : 0u;

    // key i of the concatenated lists = partial[list j = i / kp][i % kp]; list ids come from lane j's register, so a
    // chunk of C keys per lane costs one memory round trip
    uint64_t reg[R];
    constexpr int C = R < 32 ? R : 32;
#pragma unroll
    for (int c0 = 0; c0 < R; c0 += C) {
        uint32_t src[C];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const uint32_t i = (uint32_t) ((c0 + c) * 64 + lane);
            const uint32_t ic = i < total ? i : 0u;
            const uint32_t j = ic / kp;
            src[c] = (uint32_t) __shfl((int) my_list, (int) j) * kp + (ic - j * kp);
        }
#pragma unroll
        for (int c = 0; c < C; ++c) reg[c0 + c] = p.partial[src[c]];
    }
    uint32_t n_real = 0;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if ((uint32_t) (r * 64 + lane) >= total) reg[r] = KEY_EMPTY;
        n_real += (uint32_t) __popcll(__ballot(reg[r] != KEY_EMPTY));
    }

    const uint32_t want = n_real < k ? n_real : k;
    uint64_t tau, kth;
    wave_radix_select<R>(reg, n_real, k, hist, lane, tau, kth);

    if (sq.dst_list == SEL_SEED) {
        // seed threshold from the sample pass: every row ranking at or before the k-th sampled candidate stays
        // eligible in the main pass (low word all ones: ties of that distance included); too few samples: no seed
        if (lane == 0) p.tau_out[sq.out_slot] = n_real >= k ? (kth | 0xFFFFFFFFull) : KEY_EMPTY;
        return;
    }

    // compaction of the selected keys; the largest goes last (K5r reads it as the worst kept screening value)
    const bool to_list = sq.dst_list != SEL_FINAL;
    uint64_t* dst = to_list ? p.partial + (size_t) sq.dst_list * kp : sm_out[wave];
    wave_emit_selected<R>(reg, n_real, k, tau, kth, dst, lane);
    if (to_list) {
        for (uint32_t i = want + (uint32_t) lane; i < kp; i += 64) dst[i] = KEY_EMPTY;
        return;
    }
    // caller-visible rows: sort the k selected keys, then the shared output stage
    const uint32_t np2 = next_pow2(want);
    for (uint32_t i = want + (uint32_t) lane; i < np2; i += 64) dst[i] = KEY_EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (want > 1) bitonic_sort_wave(dst, np2, lane);
    select_emit<64>(p, sq, dst, want, lane);
}

// NT = 1024 for long candidate streams, 256 for short ones (cheaper barriers, 8 workgroups per CU)
template <int NT>
__global__ __launch_bounds__(NT) void select_kernel(const SelectParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    const int tid = threadIdx.x;
    const SelectQuery sq = p.queries[blockIdx.x];
    const uint32_t cap = p.cap, k = p.k;
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);
    TopKCtrl* ctrl = reinterpret_cast<TopKCtrl*>(keys + cap);
    if (tid == 0) {
        ctrl->tau = KEY_EMPTY;
        ctrl->count = 0;
    }
    __syncthreads();

    const uint32_t* ids = p.list_ids + sq.ids_begin;
    const uint32_t kp = p.kp;
    const uint64_t total = (uint64_t) sq.n_lists * kp;
    const uint32_t trigger = cap - NT;
    auto fetch = [&](uint64_t i) -> uint64_t {              // key i of the query's concatenated lists
        if (i >= total) return KEY_EMPTY;
        const uint32_t j = (uint32_t) (i / kp);
        return p.partial[(size_t) ids[j] * kp + (uint32_t) (i - (uint64_t) j * kp)];
    };
    uint64_t next = fetch(tid);                              // one key ahead of the loop
    for (uint64_t base = 0; base < total; base += NT) {
        const uint64_t key = next;
        next = fetch(base + NT + tid);
        const uint64_t tau = lds_peek(&ctrl->tau);
        topk_append(keys, ctrl, key < tau, key);            // KEY_EMPTY never passes (tau <= KEY_EMPTY)
        if (base + NT < total) {
            __syncthreads();                                 // every append of this round is counted
            if (ctrl->count > trigger) topk_compact<NT>(keys, ctrl, k, tid, false);
            else __syncthreads();                            // nobody appends before all have read count
        }
    }
    __syncthreads();
    topk_compact<NT>(keys, ctrl, k, tid, true);

    select_emit<NT>(p, sq, keys, ctrl->count, tid);
}

hipError_t launch_select(const SelectParams& p, uint32_t n_queries, int threads, hipStream_t s)
{
    const size_t lds = (size_t) p.cap * sizeof(uint64_t) + sizeof(TopKCtrl);
    auto launch = [&](auto kern, int nt) -> hipError_t {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kern, dim3(n_queries), dim3(nt), lds, s, p);
        return hipGetLastError();
    };
    if (threads == 64) {                                     // one wave per query; p.cap carries the key capacity 64 * R
        const dim3 grid((n_queries + 3) / 4), block(256);
        if (p.cap <= 1024) hipLaunchKernelGGL(select_radix_kernel<16>, grid, block, 0, s, p, n_queries);
        else if (p.cap <= 2048) hipLaunchKernelGGL(select_radix_kernel<32>, grid, block, 0, s, p, n_queries);
        else if (p.cap <= 4096) hipLaunchKernelGGL(select_radix_kernel<64>, grid, block, 0, s, p, n_queries);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    return threads == 256 ? launch(select_kernel<256>, 256) : launch(select_kernel<1024>, 1024);
}

// Fan-in (lists per K5 item) of the one-wave-per-query selection for lists of kp keys, 0 if it does not apply.
uint32_t select_wave_fanin(uint32_t kp)
{
    if (kp > SEL_WAVE_MAX_K) return 0;
    const uint32_t f = 4096u / kp;
    return f >= 8 ? (f < 64 ? f : 64u) : 0u;
}

uint32_t select_cap(uint32_t k, int threads)
{
    uint32_t cap = 512;
    while (cap < 2 * k + (uint32_t) threads) cap <<= 1;
    return cap;
}

// -------------------------------------------------------------------------------------------------
// Multi-GPU merge: n_parts per-shard sorted result lists per query -> global top-k (K5 on the
// all-gathered candidates).  keys carry (monotone distance, global row), so the order across shards
// is the same total order as on one GPU.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void merge_lists_kernel(const uint64_t* keys, const int64_t* blocks,
                                                          const int32_t* docs, const float* dist,
                                                          uint32_t n_parts, uint32_t n_queries, uint32_t k,
                                                          uint32_t np2, size_t part_stride, int64_t* out_block,
                                                          int32_t* out_doc, float* out_dist, uint64_t* out_keys,
                                                          int32_t* out_count)
{
    // part_stride == 0: four arrays of layout [n_parts][nq][k]; else every part is one packed record
    // {keys[nq][k], blocks[nq][k], docs[nq][k], dist[nq][k]} and part p starts part_stride bytes after part p-1
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* skey = reinterpret_cast<uint64_t*>(smem);          // [np2]
    uint32_t* spos = reinterpret_cast<uint32_t*>(skey + np2);    // [np2]
    const int tid = threadIdx.x;
    const uint32_t q = blockIdx.x;
    const uint32_t total = n_parts * k;
    for (uint32_t i = tid; i < np2; i += 256) {
        uint64_t key = KEY_EMPTY;
        uint32_t pos = 0;
        if (i < total) {
            const uint32_t part = i / k, j = i % k;
            if (part_stride) {
                pos = i;                                     // position = (part, j); resolved again when reading
                key = reinterpret_cast<const uint64_t*>(reinterpret_cast<const char*>(keys) + part * part_stride)[q * k + j];
            } else {
                pos = (part * n_queries + q) * k + j;
                key = keys[pos];
            }
        }
        skey[i] = key;
        spos[i] = pos;
    }
    __syncthreads();
    for (uint32_t size = 2; size <= np2; size <<= 1) {
        for (uint32_t stride = size >> 1; stride > 0; stride >>= 1) {
            for (uint32_t t = tid; t < (np2 >> 1); t += 256) {
                const uint32_t i = 2 * t - (t & (stride - 1)), j = i + stride;
                const bool up = (i & size) == 0;
                const uint64_t a = skey[i], b = skey[j];
                if ((a > b) == up) {
                    skey[i] = b; skey[j] = a;
                    const uint32_t pa = spos[i]; spos[i] = spos[j]; spos[j] = pa;
                }
            }
            __syncthreads();
        }
    }
    uint32_t m = 0;
    for (uint32_t i = tid; i < k; i += 256) {
        const size_t o = (size_t) q * k + i;
        const uint64_t key = i < np2 ? skey[i] : KEY_EMPTY;
        if (key != KEY_EMPTY) {
            const uint32_t pos = spos[i];
            if (part_stride) {
                const size_t off = (size_t) (pos / k) * part_stride;
                const uint32_t e = q * k + pos % k;
                out_block[o] = reinterpret_cast<const int64_t*>(reinterpret_cast<const char*>(blocks) + off)[e];
                out_doc[o] = reinterpret_cast<const int32_t*>(reinterpret_cast<const char*>(docs) + off)[e];
                out_dist[o] = reinterpret_cast<const float*>(reinterpret_cast<const char*>(dist) + off)[e];
            } else {
                out_block[o] = blocks[pos];
                out_doc[o] = docs[pos];
                out_dist[o] = dist[pos];
            }
            if (out_keys) out_keys[o] = key;
        } else {
            out_block[o] = -1;
            out_doc[o] = -1;
            out_dist[o] = __builtin_inff();
            if (out_keys) out_keys[o] = KEY_EMPTY;
        }
    }
    // count = number of real keys among the first k (keys are sorted, EMPTY last)
    if (tid == 0) {
        const uint32_t lim = k < np2 ? k : np2;
        uint32_t lo = 0, hi = lim;                   // first EMPTY position
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (skey[mid] != KEY_EMPTY) lo = mid + 1; else hi = mid;
        }
        m = lo;
        out_count[q] = (int32_t) m;
    }
}

hipError_t launch_merge_lists(const uint64_t* keys, const int64_t* blocks, const int32_t* docs, const float* dist,
                              uint32_t n_parts, uint32_t n_queries, uint32_t k, size_t part_stride, int64_t* out_block,
                              int32_t* out_doc, float* out_dist, uint64_t* out_keys, int32_t* out_count,
                              hipStream_t s)
{
    uint32_t np2 = 2;
    while (np2 < n_parts * k) np2 <<= 1;
    const size_t lds = (size_t) np2 * (sizeof(uint64_t) + sizeof(uint32_t));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(merge_lists_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(merge_lists_kernel, dim3(n_queries), dim3(256), lds, s, keys, blocks, docs, dist, n_parts,
                       n_queries, k, np2, part_stride, out_block, out_doc, out_dist, out_keys, out_count);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Row norms (sum x^2 in fp32, the normb accumulator of vector.c:638-655), one wave per row.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_norms_kernel(const float4* rows, uint32_t n_rows, uint32_t stride4,
                                                        float* norm2)
{
    const int lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * 256) >> 6;
    for (uint32_t r = wave; r < n_rows; r += n_waves) {
        float s = 0.0f;
        for (uint32_t c = lane; c < stride4; c += 64) {
            const float4 x = rows[(size_t) r * stride4 + c];
            s = fmaf(x.x, x.x, s); s = fmaf(x.y, x.y, s); s = fmaf(x.z, x.z, s); s = fmaf(x.w, x.w, s);
        }
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) norm2[r] = s;
    }
}

hipError_t launch_row_norms(const float4* rows, uint32_t n_rows, uint32_t stride4, float* norm2, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    uint32_t blocks = (n_rows + 3) / 4;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(row_norms_kernel, dim3(blocks), dim3(256), 0, s, rows, n_rows, stride4, norm2);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Screening planes (K2w): 8 consecutive floats -> one chunk of 8 bf16 "hi" values and one of 8 "mid" values
// (hi = bf16(x), mid = bf16(x - hi); round to nearest even; x - hi is exact in fp32).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split8(const float (&x)[8], uint4& hi, uint4& mid)
{
    unsigned short h[8], m[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const __bf16 bh = (__bf16) x[j];
        const float r = x[j] - (float) bh;
        const __bf16 bm = (__bf16) r;
        h[j] = __builtin_bit_cast(unsigned short, bh);
        m[j] = __builtin_bit_cast(unsigned short, bm);
    }
    hi = make_uint4(h[0] | (uint32_t) h[1] << 16, h[2] | (uint32_t) h[3] << 16, h[4] | (uint32_t) h[5] << 16, h[6] | (uint32_t) h[7] << 16);
    mid = make_uint4(m[0] | (uint32_t) m[1] << 16, m[2] | (uint32_t) m[3] << 16, m[4] | (uint32_t) m[5] << 16, m[6] | (uint32_t) m[7] << 16);
}

// planes of one padded fp32 row (stride4 float4, zeros past dim).  Two layouts (vsr_device.h, plane_stride4):
//   hi + mid:  item = (64-float stage s, chunk c < 8)   -> hi at prow[16 s + c], mid at prow[16 s + 8 + c]
//   hi only :  item = (128-float stage s, chunk c < 16) -> hi at prow[hs s + c] and, for query rows (mid_off = 16,
//              hs = 32), mid at prow[hs s + 16 + c]; corpus rows (mid_off = 0, hs = 16) store no mid at all
__device__ __forceinline__ void split_row_item(const float4* row, uint32_t stride4, uint4* prow, uint32_t item, bool ho,
                                               uint32_t ho_stage_chunks, uint32_t ho_mid_off)
{
    const uint32_t s = ho ? item >> 4 : item >> 3, c = ho ? item & 15u : item & 7u;
    const uint32_t f4 = (ho ? s * 32 : s * 16) + c * 2;      // first of the two float4 holding the 8 floats
    const float4 a = f4 < stride4 ? row[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = f4 + 1 < stride4 ? row[f4 + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint4 hi, mid;
    split8(x, hi, mid);
    if (ho) {
        prow[s * ho_stage_chunks + c] = hi;
        if (ho_mid_off) prow[s * ho_stage_chunks + ho_mid_off + c] = mid;
    } else {
        prow[s * 16 + c] = hi;
        prow[s * 16 + 8 + c] = mid;
    }
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float4* rows, uint32_t n_rows, uint32_t stride4, uint4* scr,
                                                           uint32_t pstride4, int ho)
{
    const uint32_t items_per_row = ho ? pstride4 : pstride4 / 2;
    const uint64_t total = (uint64_t) n_rows * items_per_row;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t) gridDim.x * 256) {
        const uint32_t r = (uint32_t) (i / items_per_row), item = (uint32_t) (i % items_per_row);
        split_row_item(rows + (size_t) r * stride4, stride4, scr + (size_t) r * pstride4, item, ho != 0, 16u, 0u);
    }
}

hipError_t launch_split_planes(const float4* rows, uint32_t n_rows, uint32_t stride4, uint4* scr, uint32_t pstride4, bool ho,
                               hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t total = (uint64_t) n_rows * (ho ? pstride4 : pstride4 / 2);
    uint32_t blocks = (uint32_t) std::min<uint64_t>((total + 255) / 256, 8192);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, s, rows, n_rows, stride4, scr, pstride4, ho ? 1 : 0);
    return hipGetLastError();
}

// does any element differ from its bf16 rounding?  (0 / -0 count as exact)
__global__ __launch_bounds__(256) void check_bf16_exact_kernel(const float4* rows, uint64_t n4, uint32_t* any_inexact)
{
    bool bad = false;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t) gridDim.x * 256) {
        const float4 v = rows[i];
        bad |= v.x != (float) (__bf16) v.x || v.y != (float) (__bf16) v.y || v.z != (float) (__bf16) v.z || v.w != (float) (__bf16) v.w;
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(any_inexact, 1u);
}

hipError_t launch_check_bf16_exact(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t* any_inexact, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t n4 = (uint64_t) n_rows * stride4;
    uint32_t blocks = (uint32_t) std::min<uint64_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(check_bf16_exact_kernel, dim3(blocks), dim3(256), 0, s, rows, n4, any_inexact);
    return hipGetLastError();
}

// is every element an integer in 0..255?
__global__ __launch_bounds__(256) void check_u8_exact_kernel(const float4* rows, uint64_t n4, uint32_t* any_inexact)
{
    bool bad = false;
    auto ok = [](float v) { return v >= 0.0f && v <= 255.0f && v == floorf(v); };
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < n4; i += (uint64_t) gridDim.x * 256) {
        const float4 v = rows[i];
        bad |= !(ok(v.x) && ok(v.y) && ok(v.z) && ok(v.w));
    }
    if (__ballot(bad) != 0 && (threadIdx.x & 63) == 0) atomicOr(any_inexact, 1u);
}

hipError_t launch_check_u8_exact(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t* any_inexact, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t n4 = (uint64_t) n_rows * stride4;
    uint32_t blocks = (uint32_t) std::min<uint64_t>((n4 + 255) / 256, 8192);
    hipLaunchKernelGGL(check_u8_exact_kernel, dim3(blocks), dim3(256), 0, s, rows, n4, any_inexact);
    return hipGetLastError();
}

// 16 consecutive elements of a padded fp32 row -> one chunk of int8 (x - 128; elements past dim: 0); returns sum (x-128)^2
__device__ __forceinline__ float chunk8(const float4* row, uint32_t stride4, uint32_t dim, uint32_t c, uint4& out, bool& bad)
{
    uint32_t w[4];
    float n2 = 0.0f;
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const uint32_t f4 = c * 4 + (uint32_t) f;
        const float4 v = f4 < stride4 ? row[f4] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float x[4] = {v.x, v.y, v.z, v.w};
        uint32_t word = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t j = f4 * 4 + (uint32_t) e;
            int b = 0;
            if (j < dim) {
                bad |= !(x[e] >= 0.0f && x[e] <= 255.0f && x[e] == floorf(x[e]));
                const float cl = fminf(fmaxf(x[e], 0.0f), 255.0f);
                b = (int) cl - 128;
                n2 += (float) (b * b);                      // integers below 2^24: exact in any order
            }
            word |= ((uint32_t) b & 0xFFu) << (8 * e);
        }
        w[f] = word;
    }
    out = make_uint4(w[0], w[1], w[2], w[3]);
    return n2;
}

__global__ __launch_bounds__(256) void split_planes8_kernel(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t dim,
                                                            uint4* scr8, float* norm8)
{
    // 8 threads per row, one chunk each; the row's norm by a shuffle over the 8
    const uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x;
    const uint32_t r = (uint32_t) (i >> 3), c = (uint32_t) (i & 7);
    const bool live = r < n_rows;
    uint4 out = make_uint4(0u, 0u, 0u, 0u);
    bool bad = false;
    float n2 = live ? chunk8(rows + (size_t) r * stride4, stride4, dim, c, out, bad) : 0.0f;
    if (live) scr8[(size_t) r * 8 + c] = out;
    n2 += __shfl_xor(n2, 1);
    n2 += __shfl_xor(n2, 2);
    n2 += __shfl_xor(n2, 4);
    if (live && c == 0) norm8[r] = n2;
}

hipError_t launch_split_planes8(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t dim, uint4* scr8, float* norm8,
                                hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t threads = (uint64_t) n_rows * 8;
    hipLaunchKernelGGL(split_planes8_kernel, dim3((uint32_t) ((threads + 255) / 256)), dim3(256), 0, s, rows, n_rows, stride4, dim,
                       scr8, norm8);
    return hipGetLastError();
}

// Coarse planes (K2g): 8 consecutive floats -> one chunk of 8 bf16 values hi = bf16(x), round to nearest even; chunks past
// the row's float4s are zero (rows are padded to whole 64-element K-steps)
__device__ __forceinline__ uint4 coarse8(const float4* row, uint32_t stride4, uint32_t c)
{
    const float4 a = 2 * c < stride4 ? row[2 * c] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 b = 2 * c + 1 < stride4 ? row[2 * c + 1] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned short h[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) h[j] = __builtin_bit_cast(unsigned short, (__bf16) x[j]);
    return make_uint4(h[0] | (uint32_t) h[1] << 16, h[2] | (uint32_t) h[3] << 16, h[4] | (uint32_t) h[5] << 16, h[6] | (uint32_t) h[7] << 16);
}

__global__ __launch_bounds__(256) void split_coarse_kernel(const float4* rows, uint32_t n_rows, uint32_t stride4, uint4* scr_c,
                                                           uint32_t cstride4)
{
    const uint64_t total = (uint64_t) n_rows * cstride4;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t) gridDim.x * 256) {
        const uint32_t r = (uint32_t) (i / cstride4), c = (uint32_t) (i % cstride4);
        scr_c[coarse_row_offset(r, cstride4 / 4) + (size_t) (c >> 2) * COARSE_SLAB_U4 + (c & 3u)] = coarse8(rows + (size_t) r * stride4, stride4, c);
    }
}

hipError_t launch_split_coarse(const float4* rows, uint32_t n_rows, uint32_t stride4, uint4* scr_c, uint32_t cstride4, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t total = (uint64_t) n_rows * cstride4;
    hipLaunchKernelGGL(split_coarse_kernel, dim3((uint32_t) std::min<uint64_t>((total + 255) / 256, 16384)), dim3(256), 0, s, rows,
                       n_rows, stride4, scr_c, cstride4);
    return hipGetLastError();
}

// Per-batch staging, see StageParams.  Workgroups [0, nq): one query each; the rest copy the descriptor block.
__global__ __launch_bounds__(256) void stage_kernel(const StageParams p)
{
    const int tid = threadIdx.x;
    if (blockIdx.x >= p.nq) {
        const uint32_t nb = gridDim.x - p.nq;
        for (uint32_t i = (blockIdx.x - p.nq) * 256 + (uint32_t) tid; i < p.n16; i += nb * 256) p.dst16[i] = p.src16[i];
        return;
    }
    const uint32_t s = blockIdx.x;
    const float* src = p.q_src + (size_t) s * p.q_stride;
    float* dst = p.q_dst + (size_t) s * p.qfloats;
    for (uint32_t j = (uint32_t) tid; j < p.qfloats; j += 256) dst[j] = j < p.dim ? src[j] : 0.0f;
    if (tid == 0) {
        p.flags[s] = 0;
        p.tau[s] = KEY_EMPTY;
        if (p.qcnt) p.qcnt[s] = 0;
        if (p.scnt) p.scnt[s] = 0;
    }
    __syncthreads();
    if (p.q_scr) {                                           // K2w: the query's bf16 hi / mid planes
        const uint32_t qstride = p.plane_ho ? 2 * p.pstride4 : p.pstride4;
        const uint32_t items = p.plane_ho ? p.pstride4 : p.pstride4 / 2;
        for (uint32_t item = (uint32_t) tid; item < items; item += 256)
            split_row_item(reinterpret_cast<const float4*>(dst), p.qfloats / 4, p.q_scr + (size_t) s * qstride, item,
                           p.plane_ho != 0, 32u, 16u);
    }
    if (p.q_scr_c)                                           // K2g: the query's coarse plane
        for (uint32_t c = (uint32_t) tid; c < p.cstride4; c += 256)
            p.q_scr_c[coarse_row_offset(s, p.cstride4 / 4) + (size_t) (c >> 2) * COARSE_SLAB_U4 + (c & 3u)] =
                coarse8(reinterpret_cast<const float4*>(dst), p.qfloats / 4, c);
    if (p.q_scr8 && tid >= 64 && tid < 72) {                 // int8 path: the query as q - 128, validated
        const uint32_t c = (uint32_t) tid - 64;
        uint4 out;
        bool bad = false;
        float n2 = chunk8(reinterpret_cast<const float4*>(dst), p.qfloats / 4, p.dim, c, out, bad);
        p.q_scr8[(size_t) s * 8 + c] = out;
        n2 += __shfl_xor(n2, 1);
        n2 += __shfl_xor(n2, 2);
        n2 += __shfl_xor(n2, 4);
        const bool any_bad = __ballot(bad) != 0;
        if (c == 0) {
            p.q_norm2_8[s] = n2;
            p.q8_bad[s] = any_bad ? 1u : 0u;
            if (any_bad) *p.q8_bad_host = 1u;
        }
    }
    if (tid < 64) {                                          // the arithmetic of row_norms_kernel, one wave per row
        const float4* row = reinterpret_cast<const float4*>(dst);
        float acc = 0.0f;
        for (uint32_t c = (uint32_t) tid; c < p.qfloats / 4; c += 64) {
            const float4 x = row[c];
            acc = fmaf(x.x, x.x, acc); acc = fmaf(x.y, x.y, acc); acc = fmaf(x.z, x.z, acc); acc = fmaf(x.w, x.w, acc);
        }
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m);
        if (tid == 0) p.q_norm2[s] = acc;
    }
}

hipError_t launch_stage(const StageParams& p, hipStream_t s)
{
    const uint32_t copy_blocks = p.n16 ? (p.n16 + 1023) / 1024 < 64 ? (p.n16 + 1023) / 1024 : 64 : 0;
    if (p.nq + copy_blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(stage_kernel, dim3(p.nq + copy_blocks), dim3(256), 0, s, p);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Permission bitmaps.  allowed(user,row) <=> docmask[doc(row)] & usermask != 0
// (row_level_security.py:54-65 with PermissionAssignment folded into per-document role bitsets).
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void build_bitmap_kernel(const uint32_t* row_doc_idx, uint32_t n_rows,
                                                           const uint64_t* doc_mask, uint32_t words,
                                                           const uint64_t* user_mask, uint64_t* bitmap)
{
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    bool allowed = false;
    if (row < n_rows) {
        const uint64_t* dm = doc_mask + (size_t) row_doc_idx[row] * words;
        for (uint32_t w = 0; w < words; ++w) allowed |= (dm[w] & user_mask[w]) != 0;
    }
    const uint64_t bits = __ballot(allowed);
    if ((threadIdx.x & 63) == 0 && row < n_rows) bitmap[row >> 6] = bits;
}

// permission bitmap of ONE permission class: bit(row) = class_of_doc[doc(row)] == cls
__global__ __launch_bounds__(256) void build_class_bitmap_kernel(const uint32_t* row_doc_idx, uint32_t n_rows,
                                                                 const uint32_t* doc_class, uint32_t cls, uint64_t* bitmap)
{
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    const bool allowed = row < n_rows && doc_class[row_doc_idx[row]] == cls;
    const uint64_t bits = __ballot(allowed);
    if ((threadIdx.x & 63) == 0 && row < n_rows) bitmap[row >> 6] = bits;
}

hipError_t launch_build_class_bitmap(const uint32_t* row_doc_idx, uint32_t n_rows, const uint32_t* doc_class, uint32_t cls,
                                     uint64_t* bitmap, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(build_class_bitmap_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, row_doc_idx, n_rows, doc_class,
                       cls, bitmap);
    return hipGetLastError();
}

hipError_t launch_build_bitmap(const uint32_t* row_doc_idx, uint32_t n_rows, const uint64_t* doc_mask,
                               uint32_t words, const uint64_t* user_mask, uint64_t* bitmap, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(build_bitmap_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, row_doc_idx, n_rows,
                       doc_mask, words, user_mask, bitmap);
    return hipGetLastError();
}

// byte-per-row masks in the caller's row order (acorn_benchmark/src/benchmark_utils.cpp:366-391,
// test_postfilter.cpp:169-195) -> bit-per-row in internal order
__global__ __launch_bounds__(256) void pack_bytemask_kernel(const uint8_t* mask_by_orig, const int64_t* orig_rows,
                                                            uint32_t n_rows, uint64_t* bitmap)
{
    const uint32_t row = blockIdx.x * 256 + threadIdx.x;
    const bool allowed = row < n_rows && mask_by_orig[orig_rows[row]] != 0;
    const uint64_t bits = __ballot(allowed);
    if ((threadIdx.x & 63) == 0 && row < n_rows) bitmap[row >> 6] = bits;
}

hipError_t launch_pack_bytemask(const uint8_t* mask_by_orig, const int64_t* orig_rows, uint32_t n_rows,
                                uint64_t* bitmap, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(pack_bytemask_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, mask_by_orig, orig_rows,
                       n_rows, bitmap);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Batched operator values for explicit vector pairs: the fmgr functions l2_distance,
// vector_negative_inner_product, cosine_distance, l1_distance (vector.c:568-578, 626-636, 660-685,
// 729-739) for n pairs at once.  One wave per pair, fp32 accumulate, float8 post-processing.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pair_distance_kernel(const float* a, const float* b, int64_t n_pairs, int dim,
                                                            int b_broadcast, int metric, double* out)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t) blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t) gridDim.x * 256) >> 6;
    for (int64_t i = wave; i < n_pairs; i += n_waves) {
        const float* x = a + i * dim;
        const float* y = b_broadcast ? b : b + i * dim;
        float s = 0.0f, na = 0.0f, nb = 0.0f;
        for (int j = lane; j < dim; j += 64) {
            const float u = x[j], v = y[j];
            if (metric == M_L2) { const float d = u - v; s = fmaf(d, d, s); }
            else if (metric == M_L1) s += fabsf(u - v);
            else {
                s = fmaf(u, v, s);
                if (metric == M_COSINE) { na = fmaf(u, u, na); nb = fmaf(v, v, nb); }
            }
        }
        for (int m = 32; m >= 1; m >>= 1) {
            s += __shfl_xor(s, m);
            na += __shfl_xor(na, m);
            nb += __shfl_xor(nb, m);
        }
        if (lane == 0) {
            double r;
            if (metric == M_L2) r = sqrt((double) s);
            else if (metric == M_IP) r = (double) -s;
            else if (metric == M_L1) r = (double) s;
            else {
                double sim = (double) s / sqrt((double) na * (double) nb);
                if (sim > 1.0) sim = 1.0; else if (sim < -1.0) sim = -1.0;
                r = 1.0 - sim;
            }
            out[i] = r;
        }
    }
}

hipError_t launch_pair_distances(const float* a, const float* b, int64_t n_pairs, int dim, int b_broadcast,
                                 int metric, double* out, hipStream_t s)
{
    if (n_pairs == 0) return hipSuccess;
    int64_t blocks = (n_pairs + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pair_distance_kernel, dim3((uint32_t) blocks), dim3(256), 0, s, a, b, n_pairs, dim,
                       b_broadcast, metric, out);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// Support functions of the cosine / inner-product opclasses and of IVFFlat's spherical k-means, batched: vector_norm
// (vector.c:756-769: sum of squares in double), l2_normalize (vector.c:774-808: x / norm in double, rounded to float4,
// zero vector stays zero, overflow reported) and vector_spherical_distance (vector.c:692-711: acos(clamp(fp32 dot)) / pi).
// One wave per vector; mode 0 = norm, 1 = normalize, 2 = spherical distance.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vector_fn_kernel(int mode, const float* a, const float* b, int64_t n, int dim, int b_broadcast,
                                                        double* out_d, float* out_f, int* overflow)
{
    const int lane = threadIdx.x & 63;
    const int64_t wave = ((int64_t) blockIdx.x * 256 + threadIdx.x) >> 6;
    const int64_t n_waves = ((int64_t) gridDim.x * 256) >> 6;
    for (int64_t i = wave; i < n; i += n_waves) {
        const float* x = a + i * dim;
        if (mode == 2) {
            const float* y = b_broadcast ? b : b + i * dim;
            float s = 0.0f;
            for (int j = lane; j < dim; j += 64) s = fmaf(x[j], y[j], s);
            for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
            double d = (double) s;
            if (d > 1) d = 1; else if (d < -1) d = -1;
            if (lane == 0) out_d[i] = acos(d) / 3.14159265358979323846;
            continue;
        }
        double norm = 0.0;
        for (int j = lane; j < dim; j += 64) norm += (double) x[j] * (double) x[j];
        for (int m = 32; m >= 1; m >>= 1) norm += __shfl_xor(norm, m);
        norm = sqrt(norm);
        if (mode == 0) {
            if (lane == 0) out_d[i] = norm;
        } else {
            bool inf = false;
            for (int j = lane; j < dim; j += 64) {
                const float r = norm > 0 ? (float) ((double) x[j] / norm) : 0.0f;
                out_f[i * dim + j] = r;
                inf |= isinf(r);
            }
            if (__ballot(inf) != 0 && lane == 0) atomicOr(overflow, 1);
        }
    }
}

hipError_t launch_vector_fn(int mode, const float* a, const float* b, int64_t n, int dim, int b_broadcast, double* out_d,
                            float* out_f, int* overflow, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    int64_t blocks = (n + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vector_fn_kernel, dim3((uint32_t) blocks), dim3(256), 0, s, mode, a, b, n, dim, b_broadcast, out_d, out_f, overflow);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// K5r: exact re-rank after K2's screening.  One workgroup per query; one wave per candidate recomputes the
// operator arithmetic of vector.c (fp32 accumulation, float8 post-processing), then the kp exact keys are
// sorted and the first k reported.  A query is flagged when a row OUTSIDE the kept set could still beat the
// k-th result, i.e. when  (worst kept screening value) - err  <=  (k-th exact value),  err = fp32 error bound of
// the screening arithmetic for this corpus (|x|^2 <= norm2_max).
// -------------------------------------------------------------------------------------------------
// keys[0 .. np2): the query's screening survivors (KEY_EMPTY padded) already in LDS; worst_kept: the largest kept
// screening key when the survivor list is full, else KEY_EMPTY; force_flag: the caller already knows the result is unproven
__device__ __forceinline__ void rerank_body(const RerankParams& p, uint32_t slot, uint64_t* keys, uint32_t np2,
                                            uint64_t worst_kept, bool force_flag)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t out_slot = p.queries[slot].out_slot;
    const float4* q = reinterpret_cast<const float4*>(p.queries_f) + (size_t) slot * p.stride4;

    float qn_part = 0.0f;
    for (uint32_t c = lane; c < p.stride4; c += 64) {
        const float4 v = q[c];
        qn_part = fmaf(v.x, v.x, qn_part); qn_part = fmaf(v.y, v.y, qn_part);
        qn_part = fmaf(v.z, v.z, qn_part); qn_part = fmaf(v.w, v.w, qn_part);
    }
    for (int m = 32; m >= 1; m >>= 1) qn_part += __shfl_xor(qn_part, m);
    const float qn = qn_part;

    // half a wave per candidate (32 lanes x float4 = 128 floats per step), U candidates per half-wave in flight:
    // 2U independent row gathers per wave hide the HBM/L2 latency of these scattered 512-byte reads.  The exact keys
    // overwrite the screening keys slot by slot: every slot is read by the half-wave that later writes it
    const int half = lane >> 5, hl = lane & 31;
    constexpr int U = 4;                                                   // candidates in flight per half-wave
    // (int8 planes: the screening value IS the exact fp32 distance -- integer sums below 2^24 -- so there is nothing to
    // recompute and no gap to prove: the kp = k smallest screening keys are the answer)
    for (uint32_t c0 = p.exact_screen ? np2 : (uint32_t) wave * 2 * U; c0 < np2; c0 += 4 * 2 * U) {
        uint64_t sk[U];
        float s[U], nx[U];
#pragma unroll
        for (int u = 0; u < U; ++u) s[u] = nx[u] = 0.f;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const uint32_t c = c0 + 2 * u + half;
            sk[u] = c < np2 ? keys[c] : KEY_EMPTY;
        }
        for (uint32_t ch = hl; ch < p.stride4; ch += 32) {
            const float4 b = q[ch];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // no branch around the gather (an empty slot reads row 0 and is dropped below): the U loads of a half-
                // wave must all be in flight before the first FMA waits
                const f32x4 av = *reinterpret_cast<const f32x4*>(p.rows + (size_t) (sk[u] == KEY_EMPTY ? 0u : (uint32_t) sk[u]) * p.stride4 + ch);
                const float4 a = make_float4(av[0], av[1], av[2], av[3]);
                if (p.metric == M_L2) {
                    const float d0 = a.x - b.x, d1 = a.y - b.y, d2 = a.z - b.z, d3 = a.w - b.w;
                    s[u] = fmaf(d0, d0, s[u]); s[u] = fmaf(d1, d1, s[u]); s[u] = fmaf(d2, d2, s[u]); s[u] = fmaf(d3, d3, s[u]);
                } else {
                    s[u] = fmaf(a.x, b.x, s[u]); s[u] = fmaf(a.y, b.y, s[u]); s[u] = fmaf(a.z, b.z, s[u]); s[u] = fmaf(a.w, b.w, s[u]);
                    if (p.metric == M_COSINE) {
                        nx[u] = fmaf(a.x, a.x, nx[u]); nx[u] = fmaf(a.y, a.y, nx[u]);
                        nx[u] = fmaf(a.z, a.z, nx[u]); nx[u] = fmaf(a.w, a.w, nx[u]);
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            for (int m = 16; m >= 1; m >>= 1) {                            // within the half-wave
                s[u] += __shfl_xor(s[u], m);
                nx[u] += __shfl_xor(nx[u], m);
            }
            const uint32_t c = c0 + 2 * u + half;
            uint64_t out = KEY_EMPTY;
            if (sk[u] != KEY_EMPTY) {
                float v;
                if (p.metric == M_L2) v = s[u];
                else if (p.metric == M_IP) v = -s[u];
                else {
                    double sim = (double) s[u] / sqrt((double) nx[u] * (double) qn);
                    if (sim > 1.0) sim = 1.0; else if (sim < -1.0) sim = -1.0;
                    v = (float) (1.0 - sim);
                }
                out = make_key(v, (uint32_t) sk[u]);
            }
            if (hl == 0 && c < np2) keys[c] = out;
        }
    }
    __syncthreads();
    bitonic_sort_lds<256>(keys, np2, tid);

    // how many exact keys exist, and the flag
    __shared__ uint32_t s_count;
    __shared__ int s_flag;
    if (tid == 0) {
        uint32_t lo = 0, hi = p.kp < np2 ? p.kp : np2;
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (keys[mid] != KEY_EMPTY) lo = mid + 1; else hi = mid;
        }
        s_count = lo;
        int flag = force_flag ? 1 : 0;
        // every row outside the kept set ranks at or after `bound` in screening value: the worst kept candidate when
        // the list is full, else the seeded threshold (if any), else there is no outside row at all
        uint64_t bound = worst_kept;
        if (bound == KEY_EMPTY && p.seeded) bound = p.tau_init[slot];
        if (bound != KEY_EMPTY) {
            const uint32_t allowed = p.queries[slot].allowed;
            if (lo < p.k) {
                if (lo < allowed) flag = 1;                                // the threshold cut below the k-th result
            } else if (!p.exact_screen) {
                const float a_last = mono_to_float((uint32_t) (bound >> 32));
                const float d_k = mono_to_float((uint32_t) (keys[p.k - 1] >> 32));
                const float g = p.err_g;                                   // relative error of the screening dot product
                const float nxm = *p.norm2_max;
                float err;
                if (p.err_tight) {
                    // |dot_s - dot| <= g |x||q|:  L2 value = |x|^2 + |q|^2 - 2 dot -> 2 g |x||q| <= g (|x|^2 + |q|^2);
                    // IP -> g |x||q|;  cosine = 1 - dot / (|x||q|) -> g.  Plus the fp32 rounding of forming the value itself
                    // (a few ulp of its largest term) and of the folded-threshold accumulator start (vsr_gemm.h)
                    if (p.metric == M_L2) err = g * (nxm + qn) * 1.0001f + 4e-6f * (nxm + qn);
                    else if (p.metric == M_IP) err = g * sqrtf(nxm * qn) * 1.0001f + 4e-6f * sqrtf(nxm * qn);
                    else err = g * 1.0001f + 4e-6f;
                } else if (p.metric == M_L2) err = 2.0f * g * (nxm + qn) + g * fabsf(a_last);
                else if (p.metric == M_IP) err = 2.0f * g * sqrtf(nxm * qn);
                else err = 8.0f * g;
                if (!(a_last - err > d_k)) flag = 1;                       // also catches NaN
            }
        }
        p.out_flags[out_slot] = flag;
        s_flag = flag;
        if (flag) atomicAdd(p.flagged_total, 1);
    }
    __syncthreads();
    const uint32_t m = s_count < p.k ? s_count : p.k;
    const size_t o = (size_t) out_slot * p.k;
    for (uint32_t i = tid; i < p.k; i += 256) {
        if (i < m) {
            const uint64_t key = keys[i];
            const uint32_t row = (uint32_t) key;
            const float v = mono_to_float((uint32_t) (key >> 32));
            p.out_block[o + i] = p.block_ids[row];
            p.out_doc[o + i] = p.doc_ids[row];
            if (p.out_row) p.out_row[o + i] = p.orig_rows[row];
            p.out_dist[o + i] = output_distance(p.metric, v);
            if (p.out_keys) p.out_keys[o + i] = (key & 0xFFFFFFFF00000000ull) | (uint64_t) (row + p.row_offset);
        } else {
            p.out_block[o + i] = -1;
            p.out_doc[o + i] = -1;
            if (p.out_row) p.out_row[o + i] = -1;
            p.out_dist[o + i] = __builtin_inff();
            if (p.out_keys) p.out_keys[o + i] = KEY_EMPTY;
        }
    }
    if (tid == 0) p.out_count[out_slot] = s_flag ? -1 - (int32_t) m : (int32_t) m;    // flagged: -1 - count (see select_emit)
}

__global__ __launch_bounds__(256) void rerank_kernel(const RerankParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                   // [np2]
    const uint32_t slot = blockIdx.x;
    const uint64_t* list = p.lists + (size_t) slot * p.kp;
    uint32_t np2 = 2;
    while (np2 < p.kp) np2 <<= 1;
    // the survivors' screening keys first (one round trip for the whole list, not one per gather round)
    for (uint32_t c = threadIdx.x; c < np2; c += 256) keys[c] = c < p.kp ? list[c] : KEY_EMPTY;
    __syncthreads();
    rerank_body(p, slot, keys, np2, list[p.kp - 1], false);              // K5 leaves the largest kept key last
}

// K2w: per query, the kp best of its candidate buffer (one wave, every key in registers, radix select) and then the
// exact re-rank of those kp rows, in one launch.  A buffer that overflowed lost candidates: the query is flagged.
__global__ __launch_bounds__(256) void select_rerank_kernel(const RerankParams p)
{
    extern __shared__ __align__(16) unsigned char smem[];
    uint64_t* keys = reinterpret_cast<uint64_t*>(smem);                   // [np2]
    __shared__ uint32_t hist[256];
    __shared__ uint64_t s_worst;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t slot = blockIdx.x;
    uint32_t np2 = 2;
    while (np2 < p.kp) np2 <<= 1;
    const uint32_t cnt = p.qcnt[slot];
    const uint32_t n = cnt < p.capq ? cnt : p.capq;
    if (tid < 64) {
        // 1024 candidates at a time (16 keys per lane): the registers of this kernel stay few enough for every query of a
        // 1000-query batch to be resident at once, and typical buffers (a few hundred to ~1500 keys) take 1-2 rounds
        uint64_t kth;
        const uint32_t want = wave_select_stream<16>(p.qcand + (size_t) slot * p.capq, n, p.kp, keys, hist, lane, kth);
        for (uint32_t i = want + (uint32_t) lane; i < np2; i += 64) keys[i] = KEY_EMPTY;
        if (lane == 0) s_worst = n >= p.kp ? kth : KEY_EMPTY;
    }
    __syncthreads();
    rerank_body(p, slot, keys, np2, s_worst, cnt > p.capq || (p.qbad && p.qbad[slot]));
}

hipError_t launch_select_rerank(const RerankParams& p, uint32_t n_queries, hipStream_t s)
{
    if (p.capq != GQ_CAP || p.kp > GQ_MAX_KP) return hipErrorInvalidValue;
    uint32_t np2 = 2;
    while (np2 < p.kp) np2 <<= 1;
    hipLaunchKernelGGL(select_rerank_kernel, dim3(n_queries), dim3(256), (size_t) np2 * sizeof(uint64_t), s, p);
    return hipGetLastError();
}

// Threshold seeds of K2w: per query, the m-th smallest of its sampled keys; every row ranking at or before it stays
// eligible in the main pass (low word all ones: ties of that distance included); too few samples: no threshold.
__global__ __launch_bounds__(256) void seed_select_kernel(const uint64_t* samp, const uint32_t* samp_cnt, uint32_t cap,
                                                          float kp_frac, uint64_t* tau, uint32_t n_queries)
{
    __shared__ uint32_t sm_hist[4][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t q = blockIdx.x * 4 + (uint32_t) wave;
    if (q >= n_queries) return;                              // wave-uniform; no workgroup barrier below
    __shared__ uint64_t sm_keep[4][GQ_SAMPLE_CAP / 4];
    const uint32_t cnt = samp_cnt[q];
    const uint32_t n = cnt < cap ? cnt : cap;
    // a buffer that overflowed holds a subset of the sample: the effective sampling fraction shrinks with it
    const float lambda = kp_frac * (cnt > n ? (float) n / (float) cnt : 1.0f);
    const uint32_t m = (uint32_t) ceilf(lambda + 6.0f * sqrtf(lambda)) + 4u;
    uint64_t kth = KEY_EMPTY;
    if (n >= m && m < 1024) (void) wave_select_stream<16>(samp + (size_t) q * cap, n, m, sm_keep[wave], sm_hist[wave], lane, kth);
    if (lane == 0) tau[q] = (n >= m && m < 1024) ? (kth | 0xFFFFFFFFull) : KEY_EMPTY;
}

hipError_t launch_seed_select(const uint64_t* samp, const uint32_t* samp_cnt, uint32_t cap, float kp_frac, uint64_t* tau,
                              uint32_t n_queries, hipStream_t s)
{
    if (cap != GQ_SAMPLE_CAP || !(kp_frac > 0.0f)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(seed_select_kernel, dim3((n_queries + 3) / 4), dim3(256), 0, s, samp, samp_cnt, cap, kp_frac, tau, n_queries);
    return hipGetLastError();
}

// -------------------------------------------------------------------------------------------------
// K3 support: list-ordered view of a corpus (IVFFlat).  Rows of a list are contiguous in the view; `rank[p]` is the row
// of the (document_id, block_id) order that physical row p holds, which is what ordering keys carry.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4* src, const float* src_norm, const uint32_t* rank,
                                                          uint32_t n_rows, uint32_t stride4, float4* dst, float* dst_norm)
{
    const uint64_t total = (uint64_t) n_rows * stride4;
    for (uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x; i < total; i += (uint64_t) gridDim.x * 256) {
        const uint32_t p = (uint32_t) (i / stride4), c = (uint32_t) (i % stride4);
        const uint32_t r = rank[p];
        dst[i] = src[(size_t) r * stride4 + c];
        if (c == 0) dst_norm[p] = src_norm[r];
    }
}

hipError_t launch_gather_rows(const float4* src, const float* src_norm, const uint32_t* rank, uint32_t n_rows, uint32_t stride4,
                              float4* dst, float* dst_norm, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    const uint64_t total = (uint64_t) n_rows * stride4;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t) std::min<uint64_t>((total + 255) / 256, 16384)), dim3(256), 0, s, src,
                       src_norm, rank, n_rows, stride4, dst, dst_norm);
    return hipGetLastError();
}

// a filter of the base corpus as a per-row bitmap in view order: bit(p) = base row rank[p] lies in one of the filter's
// tiles (sorted by start) and, when the filter carries a bitmap, has its bit set
__global__ __launch_bounds__(256) void view_bitmap_kernel(const uint32_t* rank, uint32_t n_rows, const uint2* tiles,
                                                          uint32_t n_tiles, const uint64_t* bitmap, uint64_t* out)
{
    const uint32_t p = blockIdx.x * 256 + threadIdx.x;
    bool ok = p < n_rows;
    if (ok) {
        const uint32_t r = rank ? rank[p] : p;            // no rank map: the base corpus itself
        if (tiles) {
            uint32_t lo = 0, hi = n_tiles;                   // last tile with start <= r
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if (tiles[mid].x <= r) lo = mid; else hi = mid;
            }
            ok = n_tiles > 0 && tiles[lo].x <= r && r - tiles[lo].x < tiles[lo].y;
        }
        if (ok && bitmap) ok = (bitmap[r >> 6] >> (r & 63)) & 1ull;
    }
    const uint64_t bits = __ballot(ok);
    if ((threadIdx.x & 63) == 0 && p < n_rows) out[p >> 6] = bits;
}

hipError_t launch_view_bitmap(const uint32_t* rank, uint32_t n_rows, const uint2* tiles, uint32_t n_tiles, const uint64_t* bitmap,
                              uint64_t* out, hipStream_t s)
{
    if (n_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(view_bitmap_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, rank, n_rows, tiles, n_tiles, bitmap, out);
    return hipGetLastError();
}

// GetScanLists (ivfscan.c:36-107): per query the `probes` nearest of `lists` centres under the opclass distance (L2
// squared, or negative inner product), nearest first, the lower list id first among equals.  One workgroup per query.
// The sums run in the order and rounding of vector.c's loops compiled without contraction (one lane per centre).
__global__ __launch_bounds__(256) void ivf_probe_kernel(const float* queries, uint32_t q_stride, const float* centers_t, int dim,
                                                        int lists, int probes, int metric, int32_t* out)
{
    // one 32-bit monotone image of the distance per list (the list id is the slot): 32768 lists, the reloption's maximum
    // (ivfflat.h:42-44), take 128 KiB of the CU's 160 KiB
    extern __shared__ __align__(16) unsigned char smem[];
    uint32_t* keys = reinterpret_cast<uint32_t*>(smem);     // [lists]; 0xFFFFFFFF = taken (above the canonical NaN's image)
    __shared__ uint64_t s_best[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* q = queries + (size_t) blockIdx.x * q_stride;         // (corpus rows as queries: the index build's assignment)
    // centers_t[j][c]: the centres TRANSPOSED (element j of all lists contiguous), so that the 64 lanes of a wave -- one centre
    // each -- read 256 contiguous bytes per element instead of 64 lines 4 * dim bytes apart; every lane still adds its own
    // centre's terms in element order (the reference's loop), and q[j] is one broadcast load
    for (int c = tid; c < lists; c += 256) {
        const float* x = centers_t + c;
        float sum = 0.0f;
        if (metric == M_L2) {
            for (int j = 0; j < dim; ++j) {
                const float d = __fsub_rn(x[(size_t) j * lists], q[j]);
                sum = __fadd_rn(sum, __fmul_rn(d, d));
            }
        } else {
            for (int j = 0; j < dim; ++j) sum = __fadd_rn(sum, __fmul_rn(x[(size_t) j * lists], q[j]));
            sum = -sum;
        }
        keys[c] = mono_bits(sum);
    }
    __syncthreads();
    for (int pr = 0; pr < probes; ++pr) {
        uint64_t best = KEY_EMPTY;
        for (int c = tid; c < lists; c += 256) {
            const uint32_t kc = keys[c];
            const uint64_t cand = kc == 0xFFFFFFFFu ? KEY_EMPTY : (((uint64_t) kc << 32) | (uint32_t) c);
            best = cand < best ? cand : best;
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const uint32_t lo = (uint32_t) __shfl_xor((int) (uint32_t) best, m), hi = (uint32_t) __shfl_xor((int) (uint32_t) (best >> 32), m);
            const uint64_t o = ((uint64_t) hi << 32) | lo;
            best = o < best ? o : best;
        }
        if (lane == 0) s_best[wave] = best;
        __syncthreads();
        uint64_t b = s_best[0];
        for (int w = 1; w < 4; ++w) b = s_best[w] < b ? s_best[w] : b;
        if (tid == 0) {
            out[(size_t) blockIdx.x * probes + pr] = b == KEY_EMPTY ? -1 : (int32_t) (uint32_t) b;
            if (b != KEY_EMPTY) keys[(uint32_t) b] = 0xFFFFFFFFu;
        }
        __syncthreads();
    }
}

hipError_t launch_ivf_probe(const float* queries, uint32_t q_stride, uint32_t nq, const float* centers_t, int dim, int lists, int probes,
                            int metric, int32_t* out, hipStream_t s)
{
    if (nq == 0) return hipSuccess;
    const size_t lds = (size_t) lists * sizeof(uint32_t);
    if (lds > 128 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(ivf_probe_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int) lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(ivf_probe_kernel, dim3(nq), dim3(256), lds, s, queries, q_stride, centers_t, dim, lists, probes, metric, out);
    return hipGetLastError();
}

hipError_t launch_rerank(const RerankParams& p, uint32_t n_queries, hipStream_t s)
{
    uint32_t np2 = 2;
    while (np2 < p.kp) np2 <<= 1;
    hipLaunchKernelGGL(rerank_kernel, dim3(n_queries), dim3(256), (size_t) np2 * sizeof(uint64_t), s, p);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void norm_max_kernel(const float* norm2, uint32_t n, uint32_t* out_bits)
{
    float m = 0.0f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float v = norm2[i];
        m = fmaxf(m, v < INFINITY ? v : INFINITY);                           // NaN counts as +Inf (fmaxf would drop it)
    }
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_uint(m));   // non-negative floats order as uints
}

hipError_t launch_norm_max(const float* norm2, uint32_t n, float* out_max, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(out_max, 0, sizeof(float), s);
    if (e != hipSuccess || n == 0) return e;
    uint32_t blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(norm_max_kernel, dim3(blocks), dim3(256), 0, s, norm2, n, reinterpret_cast<uint32_t*>(out_max));
    return hipGetLastError();
}

}  // namespace vsr
