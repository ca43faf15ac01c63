// Stand-alone launches of riders (rider.h); the parking slot is per-stream state in reduce.hip.
#include "rider.h"

namespace emb {

__global__ __launch_bounds__(64) void rider_mlp_fwd_kernel(const MlpArgs<__bf16> a, const MmFwdLayout lay) {
  extern __shared__ __attribute__((aligned(16))) char rider_smem[];
  mlp_fwd_mfma_body(a, lay, (int)blockIdx.x, rider_smem);
}
__global__ __launch_bounds__(64) void rider_mlp_bwd_kernel(const MlpBwdArgs<__bf16> a, const MmBwdLayout lay) {
  extern __shared__ __attribute__((aligned(16))) char rider_smem[];
  mlp_bwd_mfma_body(a, lay, (int)blockIdx.x, rider_smem);
}

int rider_launch(const Rider& r) {
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rider_mlp_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&rider_mlp_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr = true;
  }
  if (r.kind == RIDER_MLP_FWD) rider_mlp_fwd_kernel<<<r.nwg, 64, r.lds, r.stream>>>(r.fa, r.fl);
  else if (r.kind == RIDER_MLP_BWD) rider_mlp_bwd_kernel<<<r.nwg, 64, r.lds, r.stream>>>(r.ba, r.bl);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb
