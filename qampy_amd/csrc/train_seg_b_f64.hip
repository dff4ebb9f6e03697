// Segment trainer kernels (train_seg.h), group B (cma, cma2, mcma, sbd, mddma, dd), double precision: a translation unit of its own for build time.
#define QH_SEG_KERNELS
#include "train_seg.h"

namespace qh {
template int launch_seg_group_b<double>(const SegArgs<double> &, int, int, int, int, dim3, size_t);
}
