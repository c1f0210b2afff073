// vsr_kmeans.hip — IVFFlat index build on the GPU: k-means++ seeding and Elkan's k-means over the sampled rows
// (pgvector/src/ivfkmeans.c:21-93 InitCenters, :192-246 ComputeNewCenters, :259-498 ElkanKmeans; ivfbuild.c:404-445
// ComputeCenters calls it on max(lists * 50, 10000) sampled rows).  The pass over every row that follows (row -> nearest
// centre, ivfbuild.c:141-227) is vsr_ivf_assign (vsr_runtime.hip).
//
// Parity definition.  The reference's result depends on (a) its random stream (RandomDouble / RandomInt of a PostgreSQL
// backend: not reproducible outside one; a seeded xorshift64* stands in here, the same one the CPU checker uses), (b) the
// order of its floating-point operations.  This file keeps BOTH: every sample runs the reference's bound tests in the
// reference's centre order with the reference's float / double types, distances are summed left to right without
// contraction, new centres are summed in sample order.  What runs in parallel is what is independent in the reference too:
// samples within an assignment step, (centre, dimension) pairs of the mean, (centre, centre) pairs of the half-distance
// matrix.  The one serial piece -- the weighted draw of k-means++ walks the samples subtracting weights -- is a one-thread
// kernel (the distance sweep before it is parallel).  Same seed and samples => the same centres as the checker bit for bit
// for L2; the spherical variant (inner product / cosine opclasses) goes through acos(), whose last bit may differ between
// the device's and the host's libm: there the centres agree within 1e-6 and the assignments are checked instead.
#include "../../include/vsrbac.h"
#include "vsr_device.h"

#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <cstdio>

namespace vsr {

struct KmRng { uint64_t s; };
__host__ __device__ inline uint64_t km_next(KmRng* r)
{
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1DULL;
}
__host__ __device__ inline double km_double(KmRng* r) { return (double) (km_next(r) >> 11) * (1.0 / 9007199254740992.0); }
__host__ __device__ inline void km_seed(KmRng* r, uint64_t seed)
{
    r->s = seed * 0x9E3779B97F4A7C15ULL + 0x1234567ULL;
    if (!r->s) r->s = 1;
    (void) km_next(r);
}

// IVFFLAT_KMEANS_DISTANCE_PROC: l2_distance for vector_l2_ops, vector_spherical_distance for ip / cosine (vector.sql:292-333)
__device__ __forceinline__ double km_distance(int metric, int dim, const float* a, const float* b)
{
    float sum = 0.0f;
    if (metric == M_L2) {
        for (int i = 0; i < dim; ++i) {
            const float d = __fsub_rn(a[i], b[i]);
            sum = __fadd_rn(sum, __fmul_rn(d, d));
        }
        return sqrt((double) sum);                                            // vector.c:568-578
    }
    for (int i = 0; i < dim; ++i) sum = __fadd_rn(sum, __fmul_rn(a[i], b[i]));
    double d = (double) sum;                                                  // vector.c:692-711
    if (d > 1) d = 1; else if (d < -1) d = -1;
    return acos(d) / 3.14159265358979323846;
}

// l2_normalize of one centre in place (vector.c:774-808; a centre that overflows stays as it is, like NormCenters)
__device__ inline void km_norm_center(int dim, float* c)
{
    double norm = 0;
    for (int i = 0; i < dim; ++i) norm += (double) c[i] * (double) c[i];
    norm = sqrt(norm);
    if (!(norm > 0)) { for (int i = 0; i < dim; ++i) c[i] = 0.0f; return; }
    bool inf = false;
    for (int i = 0; i < dim; ++i) inf |= isinf((float) (c[i] / norm));
    if (inf) return;
    for (int i = 0; i < dim; ++i) c[i] = (float) (c[i] / norm);
}

struct KmState {
    int metric, dim, nc;
    int64_t ns;
    const float* samples;      // [ns][dim]
    float* centers;            // [nc][dim]
    float* newc;               // [nc][dim]
    float* lower;              // [nc][ns] (centre-major: the lanes of a wave are consecutive samples, so lower[c][j] is coalesced)
    float* upper;              // [ns]
    float* weight;             // [ns]
    float* s;                  // [nc]
    float* half;               // [nc][nc]
    float* newcdist;           // [nc]
    int* counts;               // [nc]
    int* closest;              // [ns]
    int* closest_sorted;       // [ns] radix sort output (keys)
    int* order_val_in;         // [ns] 0 .. ns - 1
    int* order_val;            // [ns] sample indices ordered by (closest, index)
    int* starts;               // [nc] first position of a centre's members in order_val
    int* changes;              // [1]
    KmRng* rng;                // [1]
};

__global__ __launch_bounds__(1) void km_first_center_kernel(KmState k)
{
    const int64_t j = (int64_t) (km_next(k.rng) % (uint64_t) k.ns);
    for (int t = 0; t < k.dim; ++t) k.centers[t] = k.samples[(size_t) j * k.dim + t];
}

// k-means++: distances of every sample to centre i, running minimum of their squares (ivfkmeans.c:45-63)
__global__ __launch_bounds__(256) void km_seed_sweep_kernel(KmState k, int i)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= k.ns) return;
    double d = km_distance(k.metric, k.dim, k.samples + (size_t) j * k.dim, k.centers + (size_t) i * k.dim);
    k.lower[(size_t) i * k.ns + j] = (float) d;
    d *= d;
    if (i == 0) k.weight[j] = FLT_MAX;
    if (d < k.weight[j]) k.weight[j] = (float) d;
}

// ... and the weighted draw of the next centre (ivfkmeans.c:65-90): choice = sum(weight) * random, then the first sample j at
// which choice - (weight[0] + ... + weight[j]) <= 0.  The reference walks the samples with one running double; here one
// workgroup does it as a prefix sum: thread t adds up its contiguous chunk, the chunk sums are scanned, the thread whose chunk
// brings the running value to <= 0 walks that chunk the reference's way.  Squared distances of integer-valued rows are
// integers whose partial sums stay below 2^53, so every double here is exact and j is the reference's; for real-valued rows
// the sums differ from the serial ones in their last bits and j can differ only when `choice` lies within ~1e-12 of a
// sample's boundary.  (One thread walking 50 000 weights took 3.5 ms per centre: 3.5 s of a 4 s build with 1000 lists.)
constexpr int KM_PICK_THREADS = 1024;
__global__ __launch_bounds__(KM_PICK_THREADS) void km_seed_pick_kernel(KmState k, int i)
{
    __shared__ double s_sum[KM_PICK_THREADS];
    __shared__ double s_choice;
    __shared__ long long s_pick;
    const int t = threadIdx.x;
    const int64_t chunk = (k.ns + KM_PICK_THREADS - 1) / KM_PICK_THREADS;
    const int64_t j0 = (int64_t) t * chunk, j1 = j0 + chunk < k.ns ? j0 + chunk : k.ns;
    double mine = 0.0;
    for (int64_t j = j0; j < j1; ++j) mine += k.weight[j];
    s_sum[t] = mine;
    if (t == 0) s_pick = -1;
    __syncthreads();
    for (int d = 1; d < KM_PICK_THREADS; d <<= 1) {          // inclusive scan of the chunk sums
        const double o = t >= d ? s_sum[t - d] : 0.0;
        __syncthreads();
        s_sum[t] += o;
        __syncthreads();
    }
    if (t == 0) s_choice = s_sum[KM_PICK_THREADS - 1] * km_double(k.rng);
    __syncthreads();
    const double choice = s_choice;
    const double before = t ? s_sum[t - 1] : 0.0;
    // the first chunk at whose end the running value is <= 0 (exactly one thread: the predecessor's end is still > 0)
    if (j0 < j1 && choice - s_sum[t] <= 0 && (t == 0 || choice - before > 0)) {
        double c = choice - before;
        int64_t j = j0;
        for (; j < j1; ++j) {
            c -= k.weight[j];
            if (c <= 0) break;
        }
        s_pick = (long long) (j < j1 ? j : j1 - 1);
    }
    __syncthreads();
    long long j = s_pick;
    if (j < 0 || j > k.ns - 1) j = k.ns - 1;                 // the reference's loop stops at ns - 1 whatever is left of choice
    for (int d = t; d < k.dim; d += KM_PICK_THREADS) k.centers[(size_t) (i + 1) * k.dim + d] = k.samples[(size_t) j * k.dim + d];
}

__global__ __launch_bounds__(256) void km_init_assign_kernel(KmState k)      // ivfkmeans.c:325-345
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= k.ns) return;
    float mind = FLT_MAX;
    int cc = 0;
    for (int c = 0; c < k.nc; ++c)
        if (k.lower[(size_t) c * k.ns + j] < mind) { mind = k.lower[(size_t) c * k.ns + j]; cc = c; }
    k.upper[j] = mind;
    k.closest[j] = cc;
}

__global__ __launch_bounds__(256) void km_half_kernel(KmState k)             // ivfkmeans.c:361-372
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t) k.nc * k.nc) return;
    const int a = (int) (i / k.nc), b = (int) (i % k.nc);
    if (a >= b) return;
    const float d = (float) (0.5 * km_distance(k.metric, k.dim, k.centers + (size_t) a * k.dim, k.centers + (size_t) b * k.dim));
    k.half[(size_t) a * k.nc + b] = d;
    k.half[(size_t) b * k.nc + a] = d;
}

__global__ __launch_bounds__(256) void km_s_kernel(KmState k)                // ivfkmeans.c:375-388
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= k.nc) return;
    float mind = FLT_MAX;
    for (int c = 0; c < k.nc; ++c)
        if (j != c && k.half[(size_t) j * k.nc + c] < mind) mind = k.half[(size_t) j * k.nc + c];
    k.s[j] = mind;
    if (j == 0) *k.changes = 0;
}

// the assignment step of Elkan's algorithm for one sample, bound tests in the reference's order (ivfkmeans.c:392-451)
__global__ __launch_bounds__(256) void km_assign_kernel(KmState k, int rjreset)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= k.ns) return;
    const float* x = k.samples + (size_t) j * k.dim;
    float* lo = k.lower + j;                                  // lo[c * ns]: this sample's bound on centre c
    const size_t ls = (size_t) k.ns;
    float up = k.upper[j];
    int cl = k.closest[j];
    int changed = 0;
    if (!(up <= k.s[cl])) {
        int rj = rjreset;
        for (int c = 0; c < k.nc; ++c) {
            float dxcx;
            if (c == cl) continue;
            if (up <= lo[c * ls]) continue;
            if (up <= k.half[(size_t) cl * k.nc + c]) continue;
            if (rj) {
                dxcx = (float) km_distance(k.metric, k.dim, x, k.centers + (size_t) cl * k.dim);
                lo[cl * ls] = dxcx;
                up = dxcx;
                rj = 0;
            } else
                dxcx = up;
            if (dxcx > lo[c * ls] || dxcx > k.half[(size_t) cl * k.nc + c]) {
                const float dxc = (float) km_distance(k.metric, k.dim, x, k.centers + (size_t) c * k.dim);
                lo[c * ls] = dxc;
                if (dxc < dxcx) {
                    cl = c;
                    up = dxc;
                    ++changed;
                }
            }
        }
    }
    k.upper[j] = up;
    k.closest[j] = cl;
    if (changed) atomicAdd(k.changes, changed);
}

// ComputeNewCenters (ivfkmeans.c:192-246).  The counts are integers (any order); the float sums must run in SAMPLE order to be
// the reference's: the samples are sorted by (centre, sample index) -- a stable radix sort of the closest[] column, per
// iteration -- and one thread per (centre, dimension) adds up its centre's members in that order.  (One thread per (centre,
// dimension) walking ALL samples was 3 ms per iteration, one thread per centre counting 1.8 ms.)
__global__ __launch_bounds__(256) void km_count_kernel(KmState k)
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= k.ns) return;
    atomicAdd(&k.counts[k.closest[j]], 1);
    k.order_val_in[j] = (int) j;
}

__global__ __launch_bounds__(1024) void km_offsets_kernel(KmState k)         // exclusive prefix of the counts: where a centre's members start
{
    __shared__ int s_part[1024];
    const int t = threadIdx.x;
    const int chunk = (k.nc + 1023) / 1024;
    const int c0 = t * chunk, c1 = c0 + chunk < k.nc ? c0 + chunk : k.nc;
    int mine = 0;
    for (int c = c0; c < c1; ++c) mine += k.counts[c];
    s_part[t] = mine;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {
        const int o = t >= d ? s_part[t - d] : 0;
        __syncthreads();
        s_part[t] += o;
        __syncthreads();
    }
    int run = t ? s_part[t - 1] : 0;
    for (int c = c0; c < c1; ++c) {
        k.starts[c] = run;
        run += k.counts[c];
    }
}

__global__ __launch_bounds__(256) void km_sum_kernel(KmState k)
{
    const int64_t i = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (i >= (int64_t) k.nc * k.dim) return;
    const int c = (int) (i / k.dim), t = (int) (i % k.dim);
    const int n = k.counts[c];
    const int* members = k.order_val + k.starts[c];         // the centre's samples, ascending
    float sum = 0.0f;
    for (int m = 0; m < n; ++m) sum = __fadd_rn(sum, k.samples[(size_t) members[m] * k.dim + t]);
    if (n > 0) {
        if (isinf(sum)) sum = sum > 0 ? FLT_MAX : -FLT_MAX;
        sum = __fdiv_rn(sum, (float) n);
    }
    k.newc[i] = sum;
}

// empty centres get random values, in centre order (the stream is serial); spherical variant: centres normalised
__global__ __launch_bounds__(1) void km_fix_centers_kernel(KmState k)
{
    for (int c = 0; c < k.nc; ++c) {
        float* x = k.newc + (size_t) c * k.dim;
        if (k.counts[c] == 0)
            for (int t = 0; t < k.dim; ++t) x[t] = (float) km_double(k.rng);
        if (k.metric != M_L2) km_norm_center(k.dim, x);
    }
}

__global__ __launch_bounds__(256) void km_newcdist_kernel(KmState k)         // ivfkmeans.c:459-462
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= k.nc) return;
    k.newcdist[c] = (float) km_distance(k.metric, k.dim, k.centers + (size_t) c * k.dim, k.newc + (size_t) c * k.dim);
}

__global__ __launch_bounds__(256) void km_bounds_kernel(KmState k)           // ivfkmeans.c:464-487
{
    const int64_t j = (int64_t) blockIdx.x * 256 + threadIdx.x;
    if (j >= k.ns) return;
    for (int c = 0; c < k.nc; ++c) {
        const float d = __fsub_rn(k.lower[(size_t) c * k.ns + j], k.newcdist[c]);
        k.lower[(size_t) c * k.ns + j] = d < 0 ? 0 : d;
    }
    k.upper[j] = __fadd_rn(k.upper[j], k.newcdist[k.closest[j]]);
}

__global__ __launch_bounds__(256) void km_random_centers_kernel(KmState k)   // RandomCenters, ivfkmeans.c:124-147 (no samples)
{
    if (blockIdx.x || threadIdx.x) return;
    for (int64_t i = 0; i < (int64_t) k.nc * k.dim; ++i) k.centers[i] = (float) km_double(k.rng);
    if (k.metric != M_L2)
        for (int c = 0; c < k.nc; ++c) km_norm_center(k.dim, k.centers + (size_t) c * k.dim);
}

}  // namespace vsr

using namespace vsr;

#define KM_HIP(expr)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) { rc = vsr_kmeans_fail(#expr, hipGetErrorString(e_), e_ == hipErrorOutOfMemory); goto done; } \
    } while (0)

int vsr_kmeans_fail(const char* what, const char* why, bool oom);     // vsr_runtime.hip: sets vsr_last_error
int vsr_ctx_device(const vsr_ctx* ctx, hipStream_t* stream);          // vsr_runtime.hip

extern "C" int vsr_ivf_kmeans(vsr_ctx* ctx, int metric, int dim, const float* samples, int64_t n_samples, int lists, uint64_t seed,
                              float* out_centers, int* out_iterations)
{
    int rc = VSR_OK;
    hipStream_t st = nullptr;
    if (!ctx || !out_centers || n_samples < 0 || (n_samples > 0 && !samples))
        return vsr_kmeans_fail("vsr_ivf_kmeans", "NULL argument", false), VSR_ERR_INVALID;
    if (dim < 1 || dim > 16000 || lists < 1 || lists > 32768)      /* IVFFLAT_MAX_LISTS, ivfflat.h:44 */
        return vsr_kmeans_fail("vsr_ivf_kmeans", "dim must be 1..16000 and lists 1..32768", false), VSR_ERR_INVALID;
    if (metric != VSR_METRIC_L2 && metric != VSR_METRIC_IP && metric != VSR_METRIC_COSINE)
        return vsr_kmeans_fail("vsr_ivf_kmeans", "metric has no ivfflat operator class", false), VSR_ERR_UNSUPPORTED;
    if (out_iterations) *out_iterations = 0;
    const int dev = vsr_ctx_device(ctx, &st);
    if (hipSetDevice(dev) != hipSuccess) return vsr_kmeans_fail("hipSetDevice", "failed", false), VSR_ERR_HIP;
    const int nc = lists;
    const int64_t ns = n_samples;
    KmState k{};
    k.metric = metric == VSR_METRIC_L2 ? M_L2 : M_IP;
    k.dim = dim;
    k.nc = nc;
    k.ns = ns;
    void* bufs[32] = {nullptr};
    int nb = 0;
    auto alloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        bufs[nb++] = p;
        return p;
    };
    KmRng h_rng;
    km_seed(&h_rng, seed);
    int iterations = 0;
    {
        float* d_samples = (float*) alloc((size_t) ns * dim * 4);
        k.centers = (float*) alloc((size_t) nc * dim * 4);
        k.newc = (float*) alloc((size_t) nc * dim * 4);
        k.lower = (float*) alloc((size_t) ns * nc * 4);
        k.upper = (float*) alloc((size_t) ns * 4);
        k.weight = (float*) alloc((size_t) ns * 4);
        k.s = (float*) alloc((size_t) nc * 4);
        k.half = (float*) alloc((size_t) nc * nc * 4);
        k.newcdist = (float*) alloc((size_t) nc * 4);
        k.counts = (int*) alloc((size_t) nc * 4);
        k.closest = (int*) alloc((size_t) ns * 4);
        k.closest_sorted = (int*) alloc((size_t) std::max<int64_t>(ns, 1) * 4);
        k.order_val_in = (int*) alloc((size_t) std::max<int64_t>(ns, 1) * 4);
        k.order_val = (int*) alloc((size_t) std::max<int64_t>(ns, 1) * 4);
        k.starts = (int*) alloc((size_t) nc * 4);
        size_t sort_bytes = 0;
        (void) hipcub::DeviceRadixSort::SortPairs(nullptr, sort_bytes, (const int*) nullptr, (int*) nullptr, (const int*) nullptr,
                                                  (int*) nullptr, (int) std::max<int64_t>(ns, 1), 0, 16, st);
        void* d_sort_tmp = alloc(std::max<size_t>(sort_bytes, 16));
        k.changes = (int*) alloc(16);
        k.rng = (KmRng*) alloc(16);
        k.samples = d_samples;
        if (!d_samples || !k.centers || !k.newc || !k.lower || !k.upper || !k.weight || !k.s || !k.half || !k.newcdist || !k.counts ||
            !k.closest || !k.changes || !k.rng || !k.closest_sorted || !k.order_val_in || !k.order_val || !k.starts || !d_sort_tmp) {
            rc = vsr_kmeans_fail("vsr_ivf_kmeans", "out of device memory", true);
            goto done;
        }
        KM_HIP(hipMemcpyAsync(k.rng, &h_rng, sizeof h_rng, hipMemcpyHostToDevice, st));
        if (ns == 0) {
            hipLaunchKernelGGL(km_random_centers_kernel, dim3(1), dim3(64), 0, st, k);
        } else {
            KM_HIP(hipMemcpyAsync(d_samples, samples, (size_t) ns * dim * 4, hipMemcpyHostToDevice, st));
            const unsigned gs = (unsigned) ((ns + 255) / 256), gc = (unsigned) ((nc + 255) / 256);
            hipLaunchKernelGGL(km_first_center_kernel, dim3(1), dim3(1), 0, st, k);
            for (int i = 0; i < nc; ++i) {
                hipLaunchKernelGGL(km_seed_sweep_kernel, dim3(gs), dim3(256), 0, st, k, i);
                if (i + 1 < nc) hipLaunchKernelGGL(km_seed_pick_kernel, dim3(1), dim3(KM_PICK_THREADS), 0, st, k, i);
            }
            hipLaunchKernelGGL(km_init_assign_kernel, dim3(gs), dim3(256), 0, st, k);
            for (int iteration = 0; iteration < 500; ++iteration) {
                hipLaunchKernelGGL(km_half_kernel, dim3((unsigned) (((int64_t) nc * nc + 255) / 256)), dim3(256), 0, st, k);
                hipLaunchKernelGGL(km_s_kernel, dim3(gc), dim3(256), 0, st, k);
                hipLaunchKernelGGL(km_assign_kernel, dim3(gs), dim3(256), 0, st, k, iteration != 0 ? 1 : 0);
                KM_HIP(hipMemsetAsync(k.counts, 0, (size_t) nc * 4, st));
                hipLaunchKernelGGL(km_count_kernel, dim3(gs), dim3(256), 0, st, k);
                hipLaunchKernelGGL(km_offsets_kernel, dim3(1), dim3(1024), 0, st, k);
                {   // stable: within a centre the sample indices stay ascending (lists <= 32768: 16 key bits)
                    size_t sb = sort_bytes;
                    KM_HIP(hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, sb, (const int*) k.closest, k.closest_sorted,
                                                              (const int*) k.order_val_in, k.order_val, (int) ns, 0, 16, st));
                }
                hipLaunchKernelGGL(km_sum_kernel, dim3((unsigned) (((int64_t) nc * dim + 255) / 256)), dim3(256), 0, st, k);
                hipLaunchKernelGGL(km_fix_centers_kernel, dim3(1), dim3(1), 0, st, k);
                hipLaunchKernelGGL(km_newcdist_kernel, dim3(gc), dim3(256), 0, st, k);
                hipLaunchKernelGGL(km_bounds_kernel, dim3(gs), dim3(256), 0, st, k);
                KM_HIP(hipMemcpyAsync(k.centers, k.newc, (size_t) nc * dim * 4, hipMemcpyDeviceToDevice, st));
                int changes = 0;
                KM_HIP(hipMemcpyAsync(&changes, k.changes, sizeof changes, hipMemcpyDeviceToHost, st));
                KM_HIP(hipStreamSynchronize(st));
                iterations = iteration + 1;
                if (changes == 0 && iteration != 0) break;
            }
        }
        KM_HIP(hipGetLastError());
        KM_HIP(hipMemcpyAsync(out_centers, k.centers, (size_t) nc * dim * 4, hipMemcpyDeviceToHost, st));
        KM_HIP(hipStreamSynchronize(st));
        if (out_iterations) *out_iterations = iterations;
    }
done:
    for (int i = 0; i < nb; ++i) (void) hipFree(bufs[i]);
    return rc;
}
