// Argument blocks of the fused MLP kernels (mlp.hip, mlp_mfma.h), shared with the kernels that can carry an MLP launch as a
// rider (rider.h).
#pragma once
#include "common.h"

namespace emb {

constexpr int kMlpMaxL = 4, kMlpWBudget = 12288;   // LDS elements for all weights of the stack
constexpr int kMlpRB = 16, kMlpThreads = 1024;     // rows per workgroup, threads per workgroup

template <typename T> struct MlpArgs {
  using P = typename AccOf<T>::type;
  const T* x;            // [B][F]
  const T* W[kMlpMaxL];  // [N_l][K_l] in compute dtype
  const P* b[kMlpMaxL];
  T* h[kMlpMaxL];        // outputs of every layer [B][N_l] (the last one is the result)
  uint8_t* mask[kMlpMaxL];   // bit0 pre-activation > 0, bit1 kept by dropout (nullable when the layer has neither)
  int N[kMlpMaxL], relu[kMlpMaxL], layer_id[kMlpMaxL];
  float drop[kMlpMaxL];
  int B, F, L;
  uint64_t seed, step_val;
  const uint64_t* step_dev;
  int64_t row0;
};

template <typename T> struct MlpBwdArgs {
  using P = typename AccOf<T>::type;
  const T* x;
  const T* W[kMlpMaxL];
  const T* h[kMlpMaxL];
  const uint8_t* mask[kMlpMaxL];
  const T* dy;           // [B][N_{L-1}]
  T* dx;                 // [B][F] or nullptr
  P* part;               // [nblk][total] partial sums; layout per layer: dW [N][K] then db [N]
  int N[kMlpMaxL], relu[kMlpMaxL];
  float drop[kMlpMaxL];
  int B, F, L, total;
};

}  // namespace emb
