// Segment trainer kernels (train_seg.h) with the adaptive step, one error function, single precision: a translation unit of its own.
#define QH_SEG_KERNELS
#include "train_seg.h"

namespace qh {
template int launch_seg_ad<float, QH_M_MCMA>(const SegArgs<float> &, int, int, dim3, size_t);
}
