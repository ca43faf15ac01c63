// Parking slot and stand-alone launches of riders (rider.h).
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

static thread_local bool t_defer = false;
static thread_local Rider t_slot{};

bool rider_deferring() { return t_defer; }

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

int rider_flush() {
  if (t_slot.kind == RIDER_NONE) return EMB_OK;
  const Rider r = t_slot;
  t_slot.kind = RIDER_NONE;
  return rider_launch(r);
}

void rider_park(const Rider& r) {
  (void)rider_flush();
  t_slot = r;
}

bool rider_take(hipStream_t s, int kind, Rider* out) {
  if (t_slot.kind != kind || t_slot.stream != s) return false;
  *out = t_slot;
  t_slot.kind = RIDER_NONE;
  return true;
}

}  // namespace emb

using namespace emb;

extern "C" int emb_rider_defer(int on) {
  t_defer = on != 0;
  return EMB_OK;
}
extern "C" int emb_rider_flush(void) { return rider_flush(); }
