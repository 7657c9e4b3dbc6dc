// Form 0 of the fused row-strip backward kernels (conv_rowbw.h): one translation unit per form, for build time and because
// each kernel carries exactly one version of its prologue / epilogue.
#include "conv_rowbw.h"
namespace hrp {
template int rowbw_launch_form<0>(const RowBwArgs&, int, int, hipStream_t);
#ifdef HRP_TIMELINE
template int rowbw_timeline_form<0>(void*, int, int);
#endif
}  // namespace hrp
