// vsr_scan_l1.hip — instantiates the K1 scan kernels for one metric (one TU per metric: parallel builds).
#include "vsr_scan.h"
#include "vsr_mq.h"

namespace vsr {

hipError_t launch_scan_l1(const ScanParams& p, int dim, int qb, uint32_t n_blocks, hipStream_t s)
{
    return launch_scan_metric<M_L1>(p, dim, qb, n_blocks, s);
}

hipError_t launch_mq_l1(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mq_metric<M_L1>(p, n_blocks, s);
}

}  // namespace vsr
