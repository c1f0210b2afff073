// vsr_device.h — shared host/device definitions of the gfx950 kernels (internal, not the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vsr {

using f32x4 = __attribute__((ext_vector_type(4))) float;   // native vector: loads as one dwordx4 and stays in registers
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using i32x4 = __attribute__((ext_vector_type(4))) int;       // A / B fragment and accumulator of v_mfma_i32_16x16x64_i8
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16; // A / B fragment of v_mfma_f32_16x16x32_bf16

// Explicit address spaces.  A pointer that comes out of a struct in memory (ScanGroup::tiles / bitmap) or an access the
// compiler must not cache (volatile) is "generic" to hipcc and becomes a FLAT instruction, which counts on both the
// vector-memory and the LDS counter and has to be waited for with s_waitcnt vmcnt(0) lgkmcnt(0): one such load in a tile
// loop drains every row load that was prefetched for the next tiles.  as_global / lds_peek keep those accesses on
// global_load (counted vmcnt) and ds_read (lgkmcnt only).
#ifdef __HIPCC__
template <class T> using gptr = const __attribute__((address_space(1))) T*;
template <class T> __device__ __forceinline__ gptr<T> as_global(const T* p) { return (gptr<T>) p; }
// LDS word that other waves update (running threshold, candidate count, vote flag): re-read it every time
template <class T> __device__ __forceinline__ T lds_peek(const T* p)
{
    return *(const volatile __attribute__((address_space(3))) T*) p;
}
// (start, rows) tile descriptor through a global-address-space pointer, as one 8-byte load
__device__ __forceinline__ uint2 load_tile(gptr<uint2> tiles, uint32_t t)
{
    const uint64_t v = ((gptr<uint64_t>) tiles)[t];
    return make_uint2((uint32_t) v, (uint32_t) (v >> 32));
}
#endif

constexpr uint64_t KEY_EMPTY = ~0ull;     // sorts after every real key (NaN keys included)
constexpr int      SCAN_THREADS = 512;    // 8 waves per workgroup
constexpr int      SCAN_WAVES = SCAN_THREADS / 64;
constexpr int      SELECT_THREADS = 1024;
constexpr int      MAX_K = 2048;
constexpr int      SCAN_QMAX = 16;        // most queries that can share one corpus pass
constexpr size_t   SCAN_LDS_BUDGET = 72 * 1024;   // per workgroup, so two workgroups fit a CU's 160 KiB

// keys one workgroup may append per query between two overflow checks (tile rows x waves, >= 256)
constexpr int scan_slack(int rw) { return SCAN_WAVES * rw > 256 ? SCAN_WAVES * rw : 256; }

enum Metric : int { M_L2 = 0, M_IP = 1, M_COSINE = 2, M_L1 = 3 };

// One unit of scan work: a (filter, <=QB queries) pair, spread over n_blocks workgroups.
struct ScanGroup {
    const uint2*    tiles;         // (row_start, n_rows<=RW) list; nullptr => implicit tiles over [0, n_rows)
    const uint64_t* bitmap;        // per-row permission bits over internal row order; nullptr => none
    uint32_t        n_tiles;
    uint32_t        q_begin;       // this pass's queries are slots q_slots[q_begin .. q_begin + q_count)
    uint32_t        q_count;       // 1..QB
    uint32_t        block_begin;   // first workgroup of this group in the launch
    uint32_t        n_blocks;
    uint32_t        partial_begin; // partial list index = partial_begin + qi * n_blocks + local_block
};

// nq == 1 fast path (the reference's own call shape: one query per call): the whole search in ONE launch.  K1 scans,
// every workgroup publishes its k best, and the LAST workgroup of every group of `fan` (a counter, no waiting) merges
// the group's lists, the last of those merges the merged lists and writes the caller's rows.  No staging kernel (the one
// pass travels in the kernel arguments, the query is read where the caller put it), no selection launch.
struct FusedTail {
    uint32_t        enable;
    uint32_t        fan;           // workgroups per first-level merge
    ScanGroup       group;         // the single pass
    uint32_t*       done;          // [1 + groups] arrival counters; zero between calls
    const int64_t*  block_ids;
    const int32_t*  doc_ids;
    const int64_t*  orig_rows;
    int64_t*        out_block;
    int32_t*        out_doc;
    int64_t*        out_row;       // may be nullptr
    float*          out_dist;
    uint64_t*       out_keys;      // may be nullptr
    int32_t*        out_count;
    int32_t*        out_flag;      // the call's flag word (the exact path never flags: cleared)
    int32_t*        flag_total;    // the session's count of flagged queries (scan8: a query that broke the u8 promise); may be null
    uint32_t        row_offset;
    int             metric;
    uint64_t*       dbg;           // development (VSR_FUSED_DBG=1): 100 MHz timestamps of the workgroup that finishes the call; nullptr otherwise
};

struct ScanParams {
    const float4*    rows;         // [n_rows][stride4] row-major, zero padded
    const float*     norm2;        // [n_rows] sum x^2 (cosine)
    uint32_t         n_rows;
    uint32_t         stride4;      // float4 per row
    const float*     queries;      // [n_slots][stride4*4], zero padded
    const float*     q_norm2;      // [n_slots]
    const ScanGroup* groups;
    uint32_t         n_groups;
    const uint32_t*  q_slots;      // query slots of all passes, concatenated
    uint64_t*        partial;      // [n_partial][kp] keys, KEY_EMPTY padded
    uint32_t         kp;           // partial list length (>= k)
    uint32_t         k;
    uint32_t         cap;          // LDS candidate capacity per query (power of two)
    uint32_t         qmax;         // query slots per workgroup (multiple of the kernel's QI)
    uint64_t*        cand;         // K1m only: [n_partial][cap] candidate buffers in global memory
    uint32_t         rw;           // rows per list tile (the corpus shape's RW)
    const uint64_t*  tau_init;     // [n_slots] seeded thresholds (sample pass), nullptr = none
    uint32_t         sample_stride;  // 1 = every tile; S > 1 = sample pass over every S-th tile of each workgroup
    const uint2*     block_map;    // shared-pass launches: workgroup -> (group, block of the group); x == ~0u: idle
                                   // workgroup.  nullptr: groups own contiguous workgroup ranges (block_begin)
    // K2w screening planes (bf16 hi / mid split of every element, vsr_planes.h layout): corpus rows and query slots
    const uint4*     scr;          // [n_rows][pstride4] 16-byte chunks
    const uint4*     q_scr;        // [n_slots][pstride4]
    uint32_t         pstride4;     // 16-byte chunks per corpus plane row (plane_stride4)
    // K2g coarse planes (vsr_gemm.h): hi = bf16(x) only, rows of cstride4 16-byte chunks (d padded to whole 64-element K-steps)
    const uint4*     scr_c;        // [n_rows][cstride4]
    const uint4*     q_scr_c;      // [n_slots][cstride4]
    uint32_t         cstride4;
    uint32_t         plane_ho;     // plane kind: 0 bf16 hi + mid; 1 bf16 hi only (every element exactly a bf16 value), 128
                                   // floats per stage; 2 int8 (u8-exact corpus and queries, x - 128; L2; 128 bytes per row)
    // K2w candidates: one buffer of capq keys per query slot, filled with returning atomics on qcnt (may exceed capq)
    uint64_t*        qcand;        // [n_slots][capq]
    uint32_t*        qcnt;         // [n_slots]
    uint32_t         capq;
    const uint32_t*  rank;         // physical row -> row of the (document_id, block_id) order that keys carry; nullptr:
                                   // identity.  Set for list-ordered views (IVF): see vsr_corpus::base
    const uint64_t*  ones;         // one all-ones 64-bit word (the "bitmap" of passes without a permission bitmap)
    uint32_t*        err;          // bounds-guard word: 1 = row index out of range, 2 = candidate buffer overflow, 4 = tile index
    uint32_t         epi;          // K2w main launch, L2: 1 = few survivors expected per wave-tile (mask epilogue), 0 = rounds
    uint32_t         k2i;          // int8 planes: bit 0 = the main launch (epi == 1), bit 1 = the sample launch runs as per-wave streams (K2i, vsr_i8s.h)
    FusedTail        fused;        // K1, nq == 1 only (enable = 0 otherwise)
};

// Per query: which partial lists to merge and where to put the result.
struct SelectQuery {
    uint32_t ids_begin;            // the query's partial lists are list_ids[ids_begin + j], j < n_lists
    uint32_t n_lists;
    uint32_t out_slot;             // output row in the result arrays (caller's query index)
    uint32_t dst_list;             // SEL_FINAL: write final results; SEL_SEED: write the seed threshold;
                                   // else: write the k best keys as partial list dst_list
    uint32_t allowed;              // rows the query's filter admits (saturated): completeness check of seeded runs
    uint32_t pad;
};

constexpr uint32_t SEL_FINAL = 0xFFFFFFFFu;
constexpr uint32_t SEL_SEED = 0xFFFFFFFEu;

struct SelectParams {
    uint64_t*          tau_out;        // SEL_SEED items: [n_slots] thresholds
    int32_t*           out_flags;      // final items of seeded runs: 1 = fewer results than the filter admits
    int32_t*           flagged_total;
    int                seeded;
    uint64_t*          partial;
    const uint32_t*    list_ids;       // partial list indices, per query a contiguous run
    const SelectQuery* queries;
    uint32_t           kp;
    uint32_t           k;
    uint32_t           cap;
    int                metric;
    uint32_t           row_offset;     // added to internal rows in out_keys (shard offset)
    // id maps of the corpus (internal row order)
    const int64_t*     block_ids;
    const int32_t*     doc_ids;
    const int64_t*     orig_rows;
    // outputs, [n_queries][k]
    int64_t*           out_block;
    int32_t*           out_doc;
    int64_t*           out_row;
    float*             out_dist;
    uint64_t*          out_keys;       // optional raw keys (multi-GPU merge)
    int32_t*           out_count;
};

// K5r: exact re-rank of the kp screening survivors of each query (after K2)
struct RerankParams {
    const uint64_t*    lists;          // [n_queries][kp] screening keys (ascending, KEY_EMPTY padded), by slot
    const SelectQuery* queries;        // out_slot per slot
    const float4*      rows;
    uint32_t           stride4;
    const float*       queries_f;      // [n_slots][stride4*4] padded query vectors (slot order)
    uint32_t           kp, k;
    int                metric;
    int                dim;
    const float*       norm2_max;      // max |row|^2 of the corpus (error bound of the screening)
    uint32_t           row_offset;
    const int64_t*     block_ids;
    const int32_t*     doc_ids;
    const int64_t*     orig_rows;
    int64_t*           out_block;
    int32_t*           out_doc;
    int64_t*           out_row;
    float*             out_dist;
    uint64_t*          out_keys;
    int32_t*           out_count;
    // select_rerank_kernel (K2w): the query's candidates come from its global buffer, not from a list
    const uint64_t*    qcand;          // [n_slots][capq]
    const uint32_t*    qcnt;           // [n_slots] appended keys (may exceed capq: overflow)
    uint32_t           capq;
    float              err_g;          // relative error bound of the screening dot product: |dot_s - dot| <= err_g |x| |q|
    uint32_t           err_tight;      // 1: the flag test uses the bound itself (K2g's coarse planes: g ~ 2^-8, where the roomy
                                       // multiples that cost nothing at g ~ 1e-5 would flag every query)
    const uint32_t*    qbad;           // [n_slots] != 0: the query's screening input was invalid (int8 path): flag it
    uint32_t           exact_screen;   // int8 planes: screening values are the exact distances, no re-rank, kp = k
    int                seeded;         // thresholds were seeded from a sample: also check completeness
    const uint64_t*    tau_init;       // [n_slots] the seeds (bound on every excluded row when the list is not full)
    int32_t*           out_flags;      // [n_queries] by out_slot: 1 = screening gap inside the error bound
    int32_t*           flagged_total;  // running count of flagged queries
};

struct KernelShape {
    int lpr;     // lanes per row
    int c;       // float4 chunks per lane per row (0 = runtime loop)
    int r;       // row slots per lane group per iteration
    int rw;      // rows per wave iteration = r * (64 / lpr)
};

// host-side launchers (vsr_kernels.hip)
KernelShape scan_shape_for_dim(int dim);
uint32_t scan_cap_for_k(int k, int dim);
int  scan_qmax(int dim, int k);    // queries per pass the LDS budget allows (1, or a multiple of 4 up to SCAN_QMAX)
inline size_t scan_lds_bytes(uint32_t qmax, uint32_t cap, uint32_t stride4)
{
    return (size_t) qmax * ((size_t) cap * 8 + 16 + (size_t) stride4 * 16 + 4) + 16;
}
hipError_t launch_scan(const ScanParams& p, int metric, int dim, int qi, uint32_t n_blocks, hipStream_t s);
// one query per call over the int8 planes (vsr_scan8.h): K1's top-k and in-kernel merge, a quarter of the bytes per row
hipError_t launch_scan8_fused(const ScanParams& p, uint32_t dim, uint32_t* q8_bad_host, uint32_t n_blocks, hipStream_t s);
// K1m (vsr_mq.h): shared-pass kernel for 2..16 queries per pass; needs dim >= 61 (>= 16 float4 per row)
bool mq_supported(int dim);
int  mq_qmax(int dim);
hipError_t launch_mq(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s);
// K2 (vsr_mfma.h): fp32-MFMA screening for shared passes (L2 / IP / cosine), followed by K5r
hipError_t launch_mfma(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s);
inline size_t mfma_lds_bytes(uint32_t stride4, int nq = 16)
{
    (void) stride4;                                        // the queries never sit in LDS (registers, or streamed per stage)
    return (size_t) 4 * 64 * 16 * 16                       // 4 wave staging images (swizzled, no padding)
         + (size_t) 4 * 128 * 8                            // row index + |row|^2 per slot, double-buffered
         + (size_t) nq * 20 + 32;
}
// K2 workgroups vote on compaction every K2_VOTE_EVERY tiles per wave (a workgroup barrier per tile costs more than
// the rare compaction it guards once thresholds are seeded); between two votes a query can gain K2_SLACK keys.
// Candidate buffers are cap keys long but CAND_SKEW keys apart more: a power-of-two pitch would put the (short) filled
// head of every buffer on the same few memory channels.
constexpr uint32_t CAND_SKEW = 32;
__host__ __device__ inline size_t cand_pitch(uint32_t cap) { return (size_t) cap + CAND_SKEW; }
constexpr uint32_t K2_VOTE_EVERY = 4;
constexpr uint32_t K2_SLACK = 4 * 64 * K2_VOTE_EVERY;
inline uint32_t mfma_cap_for_k(uint32_t kp)
{
    uint32_t cap = 512;
    while (cap < 2 * kp + K2_SLACK) cap <<= 1;
    return cap;                    // sorted in the staging LDS: must stay <= 8192 keys (planner gate)
}
inline int mfma_qmax(uint32_t stride4) { (void) stride4; return 32; }
// K2w (vsr_mfmaw.h): GEMM-shaped shared passes for rows of <= 256 floats -- one 64-row tile staged per WORKGROUP and
// multiplied against up to 64 (128 at d <= 128) queries whose B fragments live in the four waves' registers.
constexpr uint32_t MW_PEND = 256;                       // survivors a wave parks in LDS between two flushes (vsr_mfmaw.h)
inline size_t mfmaw_lds_bytes(bool int8 = false)
{
    // two 64-row stage buffers (16 chunks per row; int8 planes: 8) + the row-mapping ring + per wave {value, row, column}[MW_PEND]
    return (size_t) 2 * 64 * (int8 ? 8 : 16) * 16 + 8 * 64 * 8 + (size_t) 4 * MW_PEND * 12 + (int8 ? 8 * 64 * 4 : 0);   // (+ int8: threshold ring)
}
constexpr uint32_t GQ_CAP = 16384;                      // candidate keys per query (a filter this small needs no threshold at all)
constexpr uint32_t GQ_SAMPLE_CAP = 4096;                // sampled minima per query kept for the threshold seed (more: dropped)
constexpr uint32_t GQ_MAX_KP = 512;                     // screening survivors the fused select + re-rank handles
inline bool mfmaw_supported(uint32_t stride4) { return stride4 >= 16 && stride4 <= 256; }  // d = 61 .. 1024 (more than 3 stages:
                                                                                              // runtime stage loop); longer rows: K2
// queries per pass: 64 (one 16-query group per wave); long rows (hi + mid planes of more than 3 stages, hi-only of more than
// 2): 128, two groups per wave
inline int  mfmaw_qmax(uint32_t pstride4, bool ho) { return pstride4 / 16 > (ho ? 2u : 3u) ? 128 : 64; }
hipError_t launch_mfmaw(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s);
// K2g (vsr_gemm.h): long rows, passes of up to 256 queries, 256 x 256 tiles with both operands in LDS, ONE bf16 product
// per element on the coarse planes (hi = bf16(x) only; rows padded to whole 64-element K-steps)
constexpr uint32_t GM_QMAX = 256;
inline uint32_t coarse_stride4(int dim) { return 8u * (uint32_t) ((dim + 63) / 64); }
// Coarse planes are stored K-step-major in blocks of 256 rows: block b = rows 256 b .. 256 b + 255, inside it one 16 KB slab
// per K-step (32 elements = 4 chunks of 16 bytes per row), rows in order inside a slab.  Offsets in 16-byte units:
constexpr uint32_t COARSE_SLAB_U4 = 256 * 4;
__host__ __device__ inline size_t coarse_row_offset(uint32_t row, uint32_t nks)     // of the row's K-step 0 chunk 0; + ks * COARSE_SLAB_U4 + chunk
{
    return ((size_t) (row >> 8) * nks) * COARSE_SLAB_U4 + (size_t) (row & 255u) * 4u;
}
inline size_t coarse_plane_u4(uint64_t n_rows, uint32_t cstride4) { return ((n_rows + 255) / 256) * 256 * (size_t) cstride4; }
// relative error bound of the coarse product xh * qh accumulated in fp32: |dot_s - dot| <= g |x| |q|
inline float coarse_err_g(int dim) { return 3.9138794e-3f + (float) (dim + 64) * 5.9604645e-8f; }     // 2^-8 (1 + 2^-9) + ...
hipError_t launch_gemm(const ScanParams& p, int metric, uint32_t n_blocks, hipStream_t s);
hipError_t launch_split_coarse(const float4* rows, uint32_t n_rows, uint32_t stride4, uint4* scr_c, uint32_t cstride4, hipStream_t s);
// Screening planes: element x = hi + mid + e with hi = bf16(x), mid = bf16(x - hi) (|e| <= 2^-18 |x|).  A plane row holds,
// for every 64-float stage s, 8 chunks of 8 hi values followed by 8 chunks of 8 mid values (16 bytes each, zero padded):
// the same 256 bytes per row and stage as the fp32 image, but ready for v_mfma_f32_16x16x32_bf16 (16x the fp32 rate).
// Hi-only layout (ho, every element exactly a bf16 value): 16 hi chunks per 128-float stage, half the bytes.  Query
// planes always carry hi and mid: [8 hi | 8 mid] per 64-float stage, or (ho) [16 hi | 16 mid] per 128-float stage.
inline uint32_t plane_stride4(int dim, bool ho) { return 16u * (uint32_t) (ho ? (dim + 127) / 128 : (dim + 63) / 64); }
hipError_t launch_check_bf16_exact(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t* any_inexact, hipStream_t s);
hipError_t launch_split_planes(const float4* rows, uint32_t n_rows, uint32_t stride4, uint4* scr, uint32_t pstride4, bool ho,
                               hipStream_t s);
// relative error bound of the plane product  xh*qh + xh*qm + xm*qh  accumulated in fp32 over `dim` elements
inline float plane_err_g(int dim) { return 3.0f * 3.8146973e-6f + (float) (3 * dim + 8) * 5.9604645e-8f; }
hipError_t launch_rerank(const RerankParams& p, uint32_t n_queries, hipStream_t s);
hipError_t launch_select_rerank(const RerankParams& p, uint32_t n_queries, hipStream_t s);
// threshold seeds of K2w: per query the m-th smallest of its sampled keys (low word all ones), KEY_EMPTY if fewer;
// m = lambda + 6 sqrt(lambda) + 4 with lambda = kp * frac * (kept / sampled): frac = the densest pass's sampling fraction
hipError_t launch_seed_select(const uint64_t* samp, const uint32_t* samp_cnt, uint32_t cap, float kp_frac, uint64_t* tau,
                              uint32_t n_queries, hipStream_t s);
hipError_t launch_norm_max(const float* norm2, uint32_t n, float* out_max, hipStream_t s);
hipError_t launch_select(const SelectParams& p, uint32_t n_queries, int threads, hipStream_t s);   // threads: 64 (one wave per query) | 256 | 1024
uint32_t select_wave_fanin(uint32_t kp);
uint32_t select_cap(uint32_t k, int threads);
// Per-batch staging (one launch): descriptor block host -> device, queries padded to the row stride, |q|^2, flag / seed init.
// Always declared value-initialised (`StageParams st{};`): the kernel takes a null pointer for "not needed".  A field added
// here and set at one call site only, with the struct left uninitialised at the other, was the round-2 memory fault
// ("The round-2 GPU memory fault"); tests/test_abi_cpu.py keeps every *Params declaration value-initialised.
struct StageParams {
    const uint4* src16;            // pinned host memory as the device sees it
    uint4*       dst16;
    uint32_t     n16;              // 16-byte units
    const float* q_src;            // nq x q_stride floats (caller's device buffer: q_stride = dim; staged host copy: qfloats)
    uint32_t     q_stride;
    float*       q_dst;            // nq x qfloats, zero padded
    uint32_t     dim, qfloats, nq;
    float*       q_norm2;          // [nq]
    uint4*       q_scr;            // [nq][q plane stride] bf16 hi / mid planes of the padded queries (nullptr: not needed)
    uint32_t     pstride4;         // corpus plane stride
    uint32_t     plane_ho;         // corpus layout; query planes: pstride4 chunks (hi + mid per 64-float stage) or, hi-only
                                   // corpus, 2 * pstride4 chunks (16 hi + 16 mid chunks per 128-float stage)
    int32_t*     flags;            // [nq] <- 0
    uint64_t*    tau;              // [nq] <- KEY_EMPTY (no seed)
    uint32_t*    qcnt;             // [nq] <- 0: K2w candidate counts (nullptr: not used)
    uint32_t*    scnt;             // [nq] <- 0: K2w sample counts
    uint4*       q_scr_c;          // [nq][cstride4] coarse planes of the queries (hi = bf16(q) only), nullptr: not needed
    uint32_t     cstride4;
    uint4*       q_scr8;           // [nq][8] int8 planes of the queries (q - 128, 16 per chunk), nullptr: not needed
    float*       q_norm2_8;        // [nq] sum (q - 128)^2 over the padded row
    uint32_t*    q8_bad;           // [nq] <- 1 for a query that is not integer-valued in 0..255 (select_rerank flags it)
    uint32_t*    q8_bad_host;      // pinned host word <- 1 if any such query (the session then leaves the int8 path)
};
hipError_t launch_stage(const StageParams& p, hipStream_t s);
// int8 planes of a u8-exact corpus (d <= 128): scr8[row][8 chunks of 16 elements x - 128, zero padded], norm8 = sum (x-128)^2
hipError_t launch_check_u8_exact(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t* any_inexact, hipStream_t s);
hipError_t launch_split_planes8(const float4* rows, uint32_t n_rows, uint32_t stride4, uint32_t dim, uint4* scr8, float* norm8,
                                hipStream_t s);
hipError_t launch_row_norms(const float4* rows, uint32_t n_rows, uint32_t stride4, float* norm2, hipStream_t s);
hipError_t launch_build_bitmap(const uint32_t* row_doc_idx, uint32_t n_rows, const uint64_t* doc_mask,
                               uint32_t words, const uint64_t* user_mask, uint64_t* bitmap, hipStream_t s);
hipError_t launch_build_class_bitmap(const uint32_t* row_doc_idx, uint32_t n_rows, const uint32_t* doc_class, uint32_t cls,
                                     uint64_t* bitmap, hipStream_t s);
hipError_t launch_pack_bytemask(const uint8_t* mask_by_orig_row, const int64_t* orig_rows, uint32_t n_rows,
                                uint64_t* bitmap, hipStream_t s);
hipError_t launch_pair_distances(const float* a, const float* b, int64_t n_pairs, int dim, int b_broadcast,
                                 int metric, double* out, hipStream_t s);
hipError_t launch_gather_rows(const float4* src, const float* src_norm, const uint32_t* rank, uint32_t n_rows, uint32_t stride4,
                              float4* dst, float* dst_norm, hipStream_t s);
hipError_t launch_view_bitmap(const uint32_t* rank, uint32_t n_rows, const uint2* tiles, uint32_t n_tiles, const uint64_t* bitmap,
                              uint64_t* out, hipStream_t s);
hipError_t launch_ivf_probe(const float* queries, uint32_t q_stride, uint32_t nq, const float* centers_t /* [dim][lists] */, int dim, int lists, int probes,
                            int metric, int32_t* out, hipStream_t s);
hipError_t launch_vector_fn(int mode, const float* a, const float* b, int64_t n, int dim, int b_broadcast, double* out_d,
                            float* out_f, int* overflow, hipStream_t s);
hipError_t launch_merge_lists(const uint64_t* keys, const int64_t* blocks, const int32_t* docs, const float* dist,
                              uint32_t n_parts, uint32_t n_queries, uint32_t k, size_t part_stride, int64_t* out_block,
                              int32_t* out_doc, float* out_dist, uint64_t* out_keys, int32_t* out_count,
                              hipStream_t s);
}  // namespace vsr
