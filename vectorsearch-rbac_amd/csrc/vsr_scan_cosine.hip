// vsr_scan_cosine.hip — instantiates the K1 scan kernels for one metric (one TU per metric: parallel builds).
#include "vsr_scan.h"
#include "vsr_mq.h"
#include "vsr_mfma.h"

namespace vsr {

hipError_t launch_scan_cosine(const ScanParams& p, int dim, int qb, uint32_t n_blocks, hipStream_t s)
{
    return launch_scan_metric<M_COSINE>(p, dim, qb, n_blocks, s);
}

hipError_t launch_mq_cosine(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mq_metric<M_COSINE>(p, n_blocks, s);
}

hipError_t launch_mfma_cosine(const ScanParams& p, uint32_t n_blocks, hipStream_t s)
{
    return launch_mfma_metric<M_COSINE>(p, n_blocks, s);
}

}  // namespace vsr
