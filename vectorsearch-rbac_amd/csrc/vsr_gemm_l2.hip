// vsr_gemm_l2.hip — instantiates the long-row GEMM screening kernels (K2g) for one metric.
#include "vsr_gemm.h"

namespace vsr {

hipError_t launch_gemm_l2(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_gemm_metric<M_L2>(p, n_blocks, s);
}

}  // namespace vsr
