// The M <= 112 instantiations of the one-kernel update (fused_update_f64_kernel<7, 12, ...>) as a translation unit of their own:
// the kernel template and its launchers live in fused_update.hip, which this file includes with XIVO_FUSED_TU = 7 (that leaves
// out the M <= 64 entry points and the dispatch). Nothing here in a trace build: fused_update.hip keeps everything then.
#ifndef XIVO_FUSED_TRACE
#define XIVO_FUSED_TRACE 0
#endif
#if !XIVO_FUSED_TRACE
#define XIVO_FUSED_TU 7
#include "fused_update.hip"
#endif
