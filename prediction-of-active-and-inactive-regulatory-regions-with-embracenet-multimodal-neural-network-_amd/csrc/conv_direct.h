// Direct (im2col-free) 1-D convolution kernels on channels-last activations, gfx950.
//
// A workgroup stages the activation rows its 128 output rows need ONCE in LDS (halo included, zero padded at
// sequence ends) and then runs the whole k*Cin reduction out of LDS: the MFMA A-fragment of tap j is simply the
// LDS tile shifted by j rows, so nothing is re-read from HBM/L2 per tap (the generic GEMM view re-fetched every
// activation k times).  Weights stream through LDS in double-buffered chunks.  Short sequences (L < 128) are
// packed several per tile, each in its own halo-padded LDS slot.
//   forward / input-gradient:  conv_direct_kernel   (same kernel; dgrad = conv of dy with tap-flipped weights)
//   weight-gradient:           conv_wgrad_direct_kernel (reduction over rows; per-slice partial slabs, reduced in
//                              fixed order by conv_wgrad_reduce_kernel)
#pragma once
#include "gemm_core.h"

namespace emb {

constexpr int kConvBT = 128;   // output rows per workgroup tile

struct ConvTiling {
  int SB;        // sequences per tile (>= 1)
  int tiles_t;   // tiles along time per sequence (1 when L <= kConvBT)
  int tiles_m;   // row tiles in total
  int slot;      // LDS rows per sequence slot
};

inline ConvTiling conv_tiling(int B, int L, int pad) {
  ConvTiling t;
  if (L >= kConvBT) {
    t.SB = 1;
    t.tiles_t = (L + kConvBT - 1) / kConvBT;
    t.slot = kConvBT + 2 * pad;
  } else {
    t.SB = kConvBT / L;
    t.tiles_t = 1;
    t.slot = L + 2 * pad;
  }
  t.tiles_m = ((B + t.SB - 1) / t.SB) * t.tiles_t;
  return t;
}

// forward (FWD: + bias, + per-tile BatchNorm partial sums) or plain (dgrad); N = output channels
// *partial_rows (nullable) receives the number of [2][N] partial-sum rows the forward wrote
int launch_conv_direct(int dtype, bool fwd, const void* x, const void* w, const void* bias, void* out, void* partial, int* partial_rows,
                       int B, int L, int cin, int KK, int N, int pad, hipStream_t s);
// slab[S][Cout][KK+1] partial weight gradients (+ bias gradient in column KK); returns S through *S_out
int launch_conv_wgrad_direct(int dtype, const void* dy, const void* x, void* slab, int B, int L, int cin, int KK, int Cout, int pad,
                             int S, hipStream_t s);
int conv_wgrad_slices(int B, int L, int cin, int pad, int KK, int Cout, int dtype);
// bf16: weight gradient (slab[S][Cout][k*cin+1]) and input gradient (dx = conv of dy with the tap-flipped weights) in one launch;
// EMB_OK, a negative error, or 1 when the shapes do not qualify (S = conv_wgrad_slices(...))
int launch_conv_bwd_dual(const void* dy, const void* x, void* slab, const void* wflip, void* dx, int B, int L, int cin, int k, int Cout,
                         int pad, int S, hipStream_t s);

}  // namespace emb
