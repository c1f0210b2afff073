// vsr_scan_l2.hip — instantiates the K1 scan kernels for one metric (one TU per metric: parallel builds).
#include "vsr_scan.h"
#include "vsr_scan8.h"
#include "vsr_mq.h"
#include "vsr_mfma.h"

namespace vsr {

hipError_t launch_scan_l2(const ScanParams& p, int dim, int qb, uint32_t n_blocks, hipStream_t s)
{
    return launch_scan_metric<M_L2>(p, dim, qb, n_blocks, s);
}

hipError_t launch_mq_l2(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mq_metric<M_L2>(p, n_blocks, s);
}

hipError_t launch_mfma_l2(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mfma_metric<M_L2>(p, n_blocks, s);
}

}  // namespace vsr
