// Segment trainer kernels (train_seg.h) of one error function in one precision: a translation unit of its own for build time.
#define QH_SEG_KERNELS
#include "train_seg.h"

namespace qh {
template int launch_seg_m<double, QH_M_MDDMA>(const SegArgs<double> &, int, int, int, dim3, size_t);
}
