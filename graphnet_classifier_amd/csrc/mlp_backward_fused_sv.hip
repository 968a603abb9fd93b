// K8 fused data + weight-gradient kernel (mlp_backward_fused.h), the instances that read the forward's saved post-activations.
#include <hip/hip_runtime.h>
#ifdef GNC_PHASE_PROBE
__device__ unsigned long long gnc_phase_probe_bwd_sv[4096 * 12];
extern "C" int gnc_phase_probe_bwd_sv_read(void* dst, size_t bytes) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(gnc_phase_probe_bwd_sv), bytes); }
#define GNC_BWD_PROBE_SYMBOL gnc_phase_probe_bwd_sv
#endif
#include "mlp_backward_fused.h"

int gnc_mlp::launch_fused_saved(const gnc_mlp_desc_t& d, const BwdArgs& b, const FusedOut& fo, int nadd, hipStream_t stream) {
  return launch_fused_variant<true>(d, b, fo, nadd, stream);
}
