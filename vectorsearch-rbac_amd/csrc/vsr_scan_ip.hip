// vsr_scan_ip.hip — instantiates the K1 scan kernels for one metric (one TU per metric: parallel builds).
#include "vsr_scan.h"
#include "vsr_mq.h"
#include "vsr_mfma.h"

namespace vsr {

hipError_t launch_scan_ip(const ScanParams& p, int dim, int qb, uint32_t n_blocks, hipStream_t s)
{
    return launch_scan_metric<M_IP>(p, dim, qb, n_blocks, s);
}

hipError_t launch_mq_ip(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mq_metric<M_IP>(p, n_blocks, s);
}

hipError_t launch_mfma_ip(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mfma_metric<M_IP>(p, n_blocks, s);
}

}  // namespace vsr
