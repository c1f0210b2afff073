// vsr_mfmaw_ip.hip — instantiates the GEMM-shaped shared-pass kernels (K2w) for one metric.
#include "vsr_mfmaw.h"

namespace vsr {

hipError_t launch_mfmaw_ip(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mfmaw_metric<M_IP>(p, n_blocks, s);
}

}  // namespace vsr
