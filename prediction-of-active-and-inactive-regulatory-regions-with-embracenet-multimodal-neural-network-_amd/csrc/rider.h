// Riders: a small, latency-bound launch that is independent of the kernels around it (the epigenomic MLP stack next to the
// sequence CNN: FFNN_pre.py vs CNN_pre.py, joined only by the EmbraceNet layer) is not launched on its own but carried by the
// next suitable launch of the same stream as its FIRST workgroups: one wave of each such workgroup runs the rider's body
// (mlp_mfma.h, no barriers), the other waves exit, and the carrying kernel's own workgroups follow with their indices shifted.
// The chains overlap on the CUs without a second stream (fork / join edges in a captured graph cost more than they hide here:
// measured 0.248 vs 0.239 ms per step).
//
// Protocol (host side, per STREAM; state in reduce.hip): emb_rider_defer(stream, 1) arms deferral; an eligible emb_mlp_fwd / emb_mlp_bwd then parks its
// launch instead of issuing it; the next carrier launch on the same stream (first-block statistics pass for the forward, the
// BatchNorm backward gather pass for the backward) takes it; emb_rider_flush(stream) launches whatever is still parked on its own.
// The caller keeps every tensor of a parked launch alive until the flush.
#pragma once
#include "mlp_args.h"
#include "mlp_mfma.h"

namespace emb {

enum { RIDER_NONE = 0, RIDER_MLP_FWD = 1, RIDER_MLP_BWD = 2 };

struct Rider {
  int kind;
  int nwg;                 // workgroups (one wave of each works on 16 rows)
  size_t lds;              // dynamic LDS bytes the body needs
  hipStream_t stream;
  MlpArgs<__bf16> fa;
  MmFwdLayout fl;
  MlpBwdArgs<__bf16> ba;
  MmBwdLayout bl;
};

bool rider_deferring(hipStream_t s);                      // deferral armed on this stream
void rider_park(const Rider& r);                          // into r.stream's slot (flushes a previously parked one first)
bool rider_take(hipStream_t s, int kind, Rider* out);     // a parked rider of that kind on that stream, removed from the slot
int rider_flush(hipStream_t s);                           // EMB_OK or a launch error
int rider_launch(const Rider& r);                         // stand-alone launch of a rider's body

}  // namespace emb
