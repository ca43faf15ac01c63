// Slab reductions of the backward kernels (reduce.hip).  Every weight-gradient kernel of this library writes per-slice /
// per-workgroup partial results ("slabs") that are then summed in a fixed order -- deterministic, no float atomics.  All
// of those sums have one shape: S slices of `per` elements, element q of the sum scattered to a parameter-gradient layout.
// They run through ONE kernel that takes a table of such jobs, so a trainer can defer them (emb_reduce_defer) and pay a
// single launch per backward pass (emb_reduce_flush) instead of one per layer.
#pragma once
#include "common.h"

namespace emb {

enum { RJ_LINEAR = 0, RJ_CONV = 1, RJ_MLP = 2, RJ_HEAD_STATS = 3 };

struct ReduceJob {
  const void* in;    // [S][per] partial sums (P-typed)
  void* out[8];      // RJ_LINEAR: dW, db;  RJ_CONV: dW, dbias;  RJ_MLP: dW_0..3, db_0..3;  RJ_HEAD_STATS: loss (float), confusion (int64[4])
  long per;
  int S, kind;
  int iv[9];         // RJ_LINEAR: N, pitch (row = [N values | bias | pad], pitch 0 = N + 1);  RJ_CONV: Cin, cin_pad, k;  RJ_MLP: L, N_0..3, K_0..3
};

// immediate mode: launches on `s`; deferred mode: queued until emb_reduce_flush().  is_double selects P.
int reduce_submit(const ReduceJob& job, bool is_double, hipStream_t s);

}  // namespace emb
