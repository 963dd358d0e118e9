// mgx_fused_inst.hip -- the fused kernel's instantiations for ONE view size (compiled once per V with -DMGX_INST_V=<V>, so
// the seven view sizes build in parallel; -DMGX_SINGLE_TU builds of mgx_kernels.hip include this file once per V instead).
#include "mgx_fused.h"

#ifndef MGX_INST_V
#error "compile with -DMGX_INST_V=<view size>"
#endif
#define MGX_CAT2(a, b) a##b
#define MGX_CAT(a, b) MGX_CAT2(a, b)

namespace mgx_fused {
int MGX_CAT(launch_v, MGX_INST_V)(int mode, const KernelArgs &ka, int threads, int lds_bytes, int64_t nwg, hipStream_t stream,
                                  int *hip_err, int *occupancy) {
    return launch_view<MGX_INST_V>(mode, ka, threads, lds_bytes, nwg, stream, hip_err, occupancy);
}
}  // namespace mgx_fused
#undef MGX_CAT
#undef MGX_CAT2
