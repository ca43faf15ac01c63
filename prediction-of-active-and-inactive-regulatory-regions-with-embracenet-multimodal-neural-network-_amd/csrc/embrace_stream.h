// Fused EmbraceNet forward for SMALL B*c with LONG K (the A549 shapes: B = 1024, c = 256..512, d1 = 1856).
//
// With a small output tile nothing is shared between waves except the output, so LDS staging of the operands
// only adds latency.  Here every wave streams its own K slices straight from global memory into MFMA fragment
// registers (16 bytes per lane per load, 24 loads in flight per lane), the four waves of a workgroup split K
// round-robin in 16-byte-per-lane blocks (neighbouring waves read neighbouring 64-byte segments of the same
// rows), and the partial tiles meet in LDS once, summed in fixed wave order.  No barrier inside the K loop.
// The tile is 32 rows x 16 columns so that a 1024 x 256 problem gives 512 workgroups (2 per CU, ~8 waves per
// CU in flight); the (tiny) modality-0 operands are requested before the modality-1 stream starts and consumed
// after it.  Epilogue shared with the tiled kernel (embrace_epilogue.h).
//
// f32 / f64: an MFMA step consumes ONE element per lane, so a 16-byte load feeds 4 (2) consecutive steps; the
// k values are visited in a permuted order (identical for A and B), which only changes the summation order.
#pragma once
#include "embrace_epilogue.h"

namespace emb {

template <typename T> struct StreamFrag;
template <> struct StreamFrag<__bf16> {
  static constexpr int STEPS = 1;
  __device__ static bf16x8 get(const bf16x8& v, int) { return v; }
};
template <> struct StreamFrag<float> {
  static constexpr int STEPS = 4;
  __device__ static float get(const f32x4& v, int j) { return v[j]; }
};
template <> struct StreamFrag<double> {
  static constexpr int STEPS = 2;
  __device__ static double get(const f64x2& v, int j) { return v[j]; }
};

constexpr int kSM = 32, kSN = 16;   // output tile of the streaming kernel
template <typename T> struct StreamCfg {   // epilogue geometry
  using type = TileCfg<T, kSM, kSN, Mma<T>::KSTEP * 4, 1, 1, 4, false, false>;
};

template <typename T> struct StreamLane {
  const T* ap[2];
  const T* bp;
  bool av[2], bv;
};

template <typename T> __device__ __forceinline__ StreamLane<T> stream_lane(const T* X, const T* W, int d, int B, int c, int row0, int col0) {
  constexpr int VEC = Elem<T>::VEC;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  StreamLane<T> s;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + i * 16 + r;
    s.av[i] = row < B;
    s.ap[i] = X + (long)min(row, B - 1) * d + g * VEC;
  }
  s.bv = col0 + r < c;
  s.bp = W + (long)min(col0 + r, c - 1) * d + g * VEC;
  return s;
}

// loads of U k-blocks (blocks j0, j0+4, ...: this wave's share), zero-filled outside the matrix
template <typename T, int U>
__device__ __forceinline__ void stream_load(const StreamLane<T>& s, int d, int j0, typename Vec16<T>::type (&a)[U][2],
                                            typename Vec16<T>::type (&b)[U]) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC, KV = VEC * 4;
  const int g = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int uu = 0; uu < U; ++uu) {
    const int k0 = (j0 + 4 * uu) * KV;
    const bool kin = k0 + g * VEC < d;          // d % VEC == 0 (checked by the launcher)
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
      a[uu][0][e] = (T)0.0f;
      a[uu][1][e] = (T)0.0f;
      b[uu][e] = (T)0.0f;
    }
    if (kin && s.av[0]) a[uu][0] = *reinterpret_cast<const V*>(s.ap[0] + k0);
    if (kin && s.av[1]) a[uu][1] = *reinterpret_cast<const V*>(s.ap[1] + k0);
    if (kin && s.bv) b[uu] = *reinterpret_cast<const V*>(s.bp + k0);
  }
}

template <typename T, int U>
__device__ __forceinline__ void stream_mma(int nk, int j0, const typename Vec16<T>::type (&a)[U][2],
                                           const typename Vec16<T>::type (&b)[U], typename Mma<T>::AccV (&acc)[2][1]) {
#pragma unroll
  for (int uu = 0; uu < U; ++uu) {
    if (j0 + 4 * uu < nk) {   // wave-uniform
#pragma unroll
      for (int st = 0; st < StreamFrag<T>::STEPS; ++st)
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          acc[mi][0] = Mma<T>::mma(StreamFrag<T>::get(a[uu][mi], st), StreamFrag<T>::get(b[uu], st), acc[mi][0]);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads, 2) void embrace_fwd_stream_kernel(
    const T* __restrict__ X0, const T* __restrict__ X1, const T* __restrict__ W0, const T* __restrict__ W1,
    const typename AccOf<T>::type* __restrict__ b0, const typename AccOf<T>::type* __restrict__ b1, const SelArgs sel,
    const double* __restrict__ u, uint64_t seed, uint64_t step_val, const uint64_t* __restrict__ step_dev, int64_t grow0,
    T* __restrict__ E, uint8_t* __restrict__ code, int B, int d0, int d1, int c, int tiles_n, int ntiles, bool vec_c) {
  using Cfg = typename StreamCfg<T>::type;
  using Mm = Mma<T>;
  using Acc = typename AccOf<T>::type;
  using V = typename Vec16<T>::type;
  constexpr int KV = Elem<T>::VEC * 4, U1 = 8, U0 = 2;
  __shared__ __attribute__((aligned(16))) Acc part[4][2][Cfg::SLAB];   // [wave][modality][32 x (16+4)]
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int row0 = (tile / tiles_n) * kSM, col0 = (tile % tiles_n) * kSN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  typename Mm::AccV acc0[2][1], acc1[2][1];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) { acc0[mi][0][q] = 0; acc1[mi][0][q] = 0; }

  const StreamLane<T> s0 = stream_lane<T>(X0, W0, d0, B, c, row0, col0), s1 = stream_lane<T>(X1, W1, d1, B, c, row0, col0);
  const int nk0 = (d0 + KV - 1) / KV, nk1 = (d1 + KV - 1) / KV;
  // modality 0 (K <= 256): requested up front, consumed after the modality-1 stream
  V a0[U0][2], b0v[U0];
  stream_load<T, U0>(s0, d0, wave, a0, b0v);
  for (int j0 = wave; j0 < nk1; j0 += 4 * U1) {
    V a[U1][2], b[U1];
    stream_load<T, U1>(s1, d1, j0, a, b);
    stream_mma<T, U1>(nk1, j0, a, b, acc1);
  }
  stream_mma<T, U0>(nk0, wave, a0, b0v, acc0);
  for (int j0 = wave + 4 * U0; j0 < nk0; j0 += 4 * U0) {   // only when d0 > 8 k-blocks
    stream_load<T, U0>(s0, d0, j0, a0, b0v);
    stream_mma<T, U0>(nk0, j0, a0, b0v, acc0);
  }

  // partial tiles -> LDS (one region per wave), one barrier, summed in wave order
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = (mi * 16 + Mm::acc_row(lane, q)) * Cfg::CS + (lane & 15);
      part[wave][0][idx] = acc0[mi][0][q];
      part[wave][1][idx] = acc1[mi][0][q];
    }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * kSM * kSN; i += kThreads) {
    const int m = i / (kSM * kSN), e = i % (kSM * kSN), idx = (e / kSN) * Cfg::CS + e % kSN;
    part[0][m][idx] = ((part[0][m][idx] + part[1][m][idx]) + part[2][m][idx]) + part[3][m][idx];
  }
  __syncthreads();
  embrace_epilogue<Cfg>(part[0][0], part[0][1], b0, b1, sel, u, seed, step_val, step_dev, grow0, E, code, B, c, row0, col0, vec_c);
}

// returns 1 when the shapes do not qualify (caller uses the LDS-tiled kernel)
template <typename T>
static int launch_embrace_fwd_stream(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1, const void* b1,
                                     const SelArgs& sel, const double* u, uint64_t seed, uint64_t step_val, const uint64_t* step_dev,
                                     int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  if (d0 % VEC || d1 % VEC || !aligned16(X0) || !aligned16(X1) || !aligned16(W0) || !aligned16(W1)) return 1;
  const int tiles_n = cdiv(c, kSN), ntiles = cdiv(B, kSM) * tiles_n;
  const bool vec_c = (c % 4 == 0) && aligned16(E) && aligned16(u) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  embrace_fwd_stream_kernel<T><<<ntiles, kThreads, 0, s>>>((const T*)X0, (const T*)X1, (const T*)W0, (const T*)W1, (const Acc*)b0,
                                                          (const Acc*)b1, sel, u, seed, step_val, step_dev, row0, (T*)E, code, B, d0,
                                                          d1, c, tiles_n, ntiles, vec_c);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb
