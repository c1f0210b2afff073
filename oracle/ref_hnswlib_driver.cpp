// ref_hnswlib_driver.cpp — TEST INFRASTRUCTURE ONLY (see vsr_oracle.h): a thin C entry point over the hnswlib copy the
// reference vendors under logical_partition_benchmark/benchmark/hnswlib.  The headers are compiled from where they lie
// in the reference tree (oracle/Makefile, target `ref`, -I$(HNSWLIB)); nothing of the reference is copied here.  The
// result, oracle/_ref/libref_hnswlib.so, is a second, reference-derived witness for the CPU oracle:
//   * BruteforceSearch (hnswlib/bruteforce.h:107-135) with a BaseFilterFunctor = exact filtered k-NN, the same contract
//     as orc_filtered_topk (L2Space returns the squared L2 sum, InnerProductSpace returns 1 - dot);
//   * HierarchicalNSW (hnswlib/hnswalg.h) with the parameters of the reference's own comparison test
//     (src/tests/test_hnsw_compare.cpp:71-79: M = 16, efConstruction = 60, efSearch = 32).
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "hnswlib/hnswlib.h"

namespace {

struct MaskFilter : hnswlib::BaseFilterFunctor {
    const uint8_t* mask;
    explicit MaskFilter(const uint8_t* m) : mask(m) {}
    bool operator()(hnswlib::labeltype id) override { return mask[id] != 0; }
};

std::unique_ptr<hnswlib::SpaceInterface<float>> make_space(int metric, int dim)
{
    if (metric == 0) return std::make_unique<hnswlib::L2Space>((size_t) dim);
    return std::make_unique<hnswlib::InnerProductSpace>((size_t) dim);
}

template <class Index>
int run_queries(Index& index, const float* queries, int nq, int dim, int k, const uint8_t* mask, int64_t* out_ids,
                float* out_dist, int32_t* out_counts)
{
    MaskFilter filter(mask);
    for (int q = 0; q < nq; ++q) {
        auto res = index.searchKnnCloserFirst(queries + (size_t) q * dim, (size_t) k, mask ? &filter : nullptr);
        out_counts[q] = (int32_t) res.size();
        for (int i = 0; i < k; ++i) {
            out_ids[(size_t) q * k + i] = i < (int) res.size() ? (int64_t) res[i].second : -1;
            out_dist[(size_t) q * k + i] = i < (int) res.size() ? res[i].first : 0.0f;
        }
    }
    return 0;
}

}  // namespace

extern "C" {

// metric: 0 = L2 (value = squared distance), 1 = inner product (value = 1 - dot).  mask: one byte per row or NULL.
int ref_bruteforce_topk(int metric, int dim, const float* rows, int64_t n, const float* queries, int nq, int k,
                        const uint8_t* mask, int64_t* out_ids, float* out_dist, int32_t* out_counts)
{
    if (n < 1 || k < 1 || k > n) return 1;
    try {
        auto space = make_space(metric, dim);
        hnswlib::BruteforceSearch<float> index(space.get(), (size_t) n);
        for (int64_t i = 0; i < n; ++i) index.addPoint(rows + (size_t) i * dim, (hnswlib::labeltype) i);
        return run_queries(index, queries, nq, dim, k, mask, out_ids, out_dist, out_counts);
    } catch (...) {
        return 2;
    }
}

int ref_hnsw_topk(int metric, int dim, const float* rows, int64_t n, int M, int ef_construction, int ef_search,
                  const float* queries, int nq, int k, const uint8_t* mask, int64_t* out_ids, float* out_dist,
                  int32_t* out_counts)
{
    if (n < 1 || k < 1) return 1;
    try {
        auto space = make_space(metric, dim);
        hnswlib::HierarchicalNSW<float> index(space.get(), (size_t) n, (size_t) M, (size_t) ef_construction, 100);
        for (int64_t i = 0; i < n; ++i) index.addPoint(rows + (size_t) i * dim, (hnswlib::labeltype) i);
        index.setEf((size_t) ef_search);
        return run_queries(index, queries, nq, dim, k, mask, out_ids, out_dist, out_counts);
    } catch (...) {
        return 2;
    }
}

// A persistent HNSW index for the reported CPU baseline (tools/cpu_hnsw_baseline.py): build once, query with several ef.
struct RefHnsw {
    std::unique_ptr<hnswlib::SpaceInterface<float>> space;
    std::unique_ptr<hnswlib::HierarchicalNSW<float>> index;
    int dim;
};

void* ref_hnsw_open(int metric, int dim, const float* rows, int64_t n, int M, int ef_construction)
{
    try {
        auto* h = new RefHnsw();
        h->dim = dim;
        h->space = make_space(metric, dim);
        h->index = std::make_unique<hnswlib::HierarchicalNSW<float>>(h->space.get(), (size_t) n, (size_t) M,
                                                                     (size_t) ef_construction, 100);
        for (int64_t i = 0; i < n; ++i) h->index->addPoint(rows + (size_t) i * dim, (hnswlib::labeltype) i);
        return h;
    } catch (...) {
        return nullptr;
    }
}

int ref_hnsw_query(void* handle, int ef_search, const float* queries, int nq, int k, int64_t* out_ids, float* out_dist,
                   int32_t* out_counts)
{
    auto* h = static_cast<RefHnsw*>(handle);
    if (!h) return 1;
    try {
        h->index->setEf((size_t) ef_search);
        return run_queries(*h->index, queries, nq, h->dim, k, nullptr, out_ids, out_dist, out_counts);
    } catch (...) {
        return 2;
    }
}

void ref_hnsw_close(void* handle) { delete static_cast<RefHnsw*>(handle); }

}  // extern "C"
