// K8 fused data + weight-gradient kernel (mlp_backward_fused.h), the instances that recompute the tile's forward from its inputs.
#include <hip/hip_runtime.h>
#ifdef GNC_PHASE_PROBE
__device__ unsigned long long gnc_phase_probe_bwd[4096 * 12];
extern "C" int gnc_phase_probe_bwd_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnc_phase_probe_bwd), bytes); }
#define GNC_BWD_PROBE_SYMBOL gnc_phase_probe_bwd
#endif
#include "mlp_backward_fused.h"

int gnc_mlp::launch_fused_recompute(const gnc_mlp_desc_t& d, const BwdArgs& b, const FusedOut& fo, int nadd, hipStream_t stream) {
  return launch_fused_variant<false>(d, b, fo, nadd, stream);
}
