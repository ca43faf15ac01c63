// Host entry points of the fp32 ring GEMM (gemm_jobs.h, built in gemm_jobs.hip).  Each returns EMB_OK, a negative error, or 1 when
// the shapes / alignments do not qualify (the caller keeps its other kernels).
#pragma once
#include "common.h"

namespace emb {

// EmbraceNet backward on pre-masked gradients (emb_embrace_bwd_masked, EMB_F32).  dD1 == nullptr: one modality only -- the
// backward of a single Linear layer y = x W^T + b on its pre-masked gradient (linear.hip): c = out features, d0 = in features
int gemm_jobs_bwd(const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0, const void* W1, void* dX0, void* dX1,
                  void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B, int d0, int d1, int c, int force_S,
                  hipStream_t s);
// Conv1d of a stored-activation block on channels-last fp32 rows (CNN_pre.py:37-38): forward (fwd: + bias, + BatchNorm partial sums
// [*partial_rows][2][N]) or input gradient (the same job on dy with the tap-flipped packed weights); cin % 32 == 0, N >= 64
int gemm_jobs_conv(bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows, int B, int L,
                   int cin, int KK, int N, int pad, hipStream_t s);
// its weight gradient: slab[*S_io][Cout][KK + 1] (column KK = bias gradient); *S_io in: slab capacity in slices, out: slices written
int gemm_jobs_conv_wgrad(const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad, int* S_io,
                         hipStream_t s);

}  // namespace emb
