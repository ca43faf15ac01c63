// The fp32 ring GEMM kernel (gemm_jobs.h) and its host entry points (gemm_jobs_api.h).
#include "gemm_jobs_api.h"
#include "gemm_jobs.h"

namespace emb {

int gemm_jobs_bwd(const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0, const void* W1, void* dX0, void* dX1,
                  void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B, int d0, int d1, int c, int force_S,
                  hipStream_t s) {
  return gemm_jobs_bwd_impl(dD0, dD1, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, ws, ws_bytes, B, d0, d1, c, force_S, s);
}
int gemm_jobs_conv(bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows, int B, int L,
                   int cin, int KK, int N, int pad, hipStream_t s) {
  return gemm_jobs_conv_impl(fwd, x, w, bias, out, partial, partial_rows, B, L, cin, KK, N, pad, s);
}
int gemm_jobs_conv_wgrad(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad, int* S_io,
                         hipStream_t s) {
  return gemm_jobs_conv_wgrad_impl(dy, x, slab, B, L, cin, KK, Cout, pad, S_io, s);
}

}  // namespace emb
