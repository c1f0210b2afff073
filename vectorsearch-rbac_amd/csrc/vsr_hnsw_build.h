// vsr_hnsw_build.h -- parameters of the batched HNSW build (vsr_hnsw_build.hip), shared with the host loop in vsr_runtime.hip
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace vsr {

constexpr int HB_NBR = 256;                // neighbour ids of one expansion / one list (2m <= 200)

struct HnswBuildParams {
    const float4*  rows = nullptr;           // corpus rows (internal order), zero padded to stride4 float4
    uint32_t       stride4 = 0;
    int            metric = 0;               // M_L2 / M_IP (cosine opclass: unit rows, negative inner product)
    uint32_t       m = 0, efc = 0, max_level = 1;
    int32_t*       nbr0 = nullptr;           // [n][2m] neighbour ids, -1 padded
    float*         dist0 = nullptr;          // [n][2m] their distances to the owner
    const int32_t* up_slot = nullptr;        // element -> slot of its upper lists, -1 for level 0
    int32_t*       up_nbr = nullptr;         // [n_upper][max_level][m]
    float*         up_dist = nullptr;
    const int32_t* level = nullptr;          // element -> top level
    int32_t        entry = -1, entry_level = -1;   // the graph's entry point when the batch begins
    uint32_t       first = 0, count = 0;     // the batch: elements first .. first + count - 1
    uint64_t*      rec_key = nullptr;        // reverse edges: (layer << 32) | target
    uint64_t*      rec_val = nullptr;        //                (new element << 32) | distance bits
    uint32_t*      rec_count = nullptr;
    uint32_t       rec_cap = 0;
    uint32_t       caps = 0;                 // entries of the sorted candidate array (ef_construction + 2m)
    uint32_t       hash_slots = 0;           // visited set of a layer search: open-addressing table in LDS (power of two)
    uint32_t       lds_per_wave = 0;
    uint32_t       wpb = 1;                  // waves (elements) per workgroup
    uint32_t*      err = nullptr;            // bit 16: a visited table filled up (the element keeps the candidates found so far)
};

}  // namespace vsr

// one batch: search + select + own lists, sort of the reverse edges, their application (all on stream s, no synchronisation)
hipError_t vsr_hnsw_build_batch(vsr::HnswBuildParams& p, void* d_sort_tmp, size_t sort_tmp_bytes, uint64_t* d_key_alt, uint64_t* d_val_alt,
                                hipStream_t s);
size_t vsr_hnsw_build_sort_bytes(uint32_t rec_cap);
