// Fused EmbraceNet forward for SMALL B*c with LONG K (the A549 shapes: B = 1024, c = 256..512, d1 = 1856).
//
// With a 32x32 output tile nothing is shared between waves except the output, so LDS staging of the
// operands only adds latency.  Here every wave streams its own K slices straight from global memory into
// MFMA fragment registers (16 bytes per lane per load, up to 32 loads in flight per lane), the four waves of
// a workgroup split K round-robin in 16-byte-per-lane blocks (neighbouring waves read neighbouring 64-byte
// segments of the same rows), and the partial tiles meet in LDS only once, in fixed wave order.  No barrier
// inside the K loop.  Epilogue shared with the tiled kernel (embrace_epilogue.h).
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

template <typename T> struct StreamCfg {   // reduction / epilogue geometry: 32x32 tile, 4 waves along K
  using type = TileCfg<T, 32, 32, Mma<T>::KSTEP * 4, 1, 1, 4, false, false>;
};

template <typename T, int U>
__device__ __forceinline__ void stream_gemm(const T* __restrict__ X, const T* __restrict__ W, int d, int B, int c, int row0, int col0,
                                            typename Mma<T>::AccV (&acc)[2][2]) {
  using Mm = Mma<T>;
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC, KV = VEC * 4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 15, g = lane >> 4;
  const T* ap[2];
  const T* bp[2];
  bool av[2], bv[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = row0 + i * 16 + r, col = col0 + i * 16 + r;
    av[i] = row < B;
    bv[i] = col < c;
    ap[i] = X + (long)min(row, B - 1) * d + g * VEC;
    bp[i] = W + (long)min(col, c - 1) * d + g * VEC;
  }
  const int nk = (d + KV - 1) / KV;
  for (int j0 = wave; j0 < nk; j0 += 4 * U) {
    V a[U][2], b[U][2];
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
      const int k0 = (j0 + 4 * uu) * KV;
      const bool kin = k0 + g * VEC < d;          // d % VEC == 0 (checked by the launcher)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          a[uu][i][e] = (T)0.0f;
          b[uu][i][e] = (T)0.0f;
        }
        if (kin && av[i]) a[uu][i] = *reinterpret_cast<const V*>(ap[i] + k0);
        if (kin && bv[i]) b[uu][i] = *reinterpret_cast<const V*>(bp[i] + k0);
      }
    }
#pragma unroll
    for (int uu = 0; uu < U; ++uu) {
      if (j0 + 4 * uu < nk) {   // wave-uniform
#pragma unroll
        for (int st = 0; st < StreamFrag<T>::STEPS; ++st)
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
              acc[mi][ni] = Mm::mma(StreamFrag<T>::get(a[uu][mi], st), StreamFrag<T>::get(b[uu][ni], st), acc[mi][ni]);
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void embrace_fwd_stream_kernel(
    const T* __restrict__ X0, const T* __restrict__ X1, const T* __restrict__ W0, const T* __restrict__ W1,
    const typename AccOf<T>::type* __restrict__ b0, const typename AccOf<T>::type* __restrict__ b1, const float* __restrict__ cdf0,
    const double* __restrict__ u, uint64_t seed, uint64_t step_val, const uint64_t* __restrict__ step_dev, int64_t grow0,
    T* __restrict__ E, uint8_t* __restrict__ code, int B, int d0, int d1, int c, int tiles_n, int ntiles, bool vec_c) {
  using Cfg = typename StreamCfg<T>::type;
  using Acc = typename AccOf<T>::type;
  __shared__ __attribute__((aligned(16))) Acc slabs[2 * Cfg::SLAB];
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int row0 = (tile / tiles_n) * 32, col0 = (tile % tiles_n) * 32;
  typename Mma<T>::AccV acc0[2][2], acc1[2][2];
  zero_acc<Cfg>(acc0);
  zero_acc<Cfg>(acc1);
  stream_gemm<T, 8>(X1, W1, d1, B, c, row0, col0, acc1);
  stream_gemm<T, 4>(X0, W0, d0, B, c, row0, col0, acc0);
  reduce_to_slab<Cfg>(acc0, slabs);
  reduce_to_slab<Cfg>(acc1, slabs + Cfg::SLAB);
  embrace_epilogue<Cfg>(slabs, slabs + Cfg::SLAB, b0, b1, cdf0, u, seed, step_val, step_dev, grow0, E, code, B, c, row0, col0, vec_c);
}

// returns 1 when the shapes do not qualify (caller uses the LDS-tiled kernel)
template <typename T>
static int launch_embrace_fwd_stream(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1, const void* b1,
                                     const float* cdf0, const double* u, uint64_t seed, uint64_t step_val, const uint64_t* step_dev,
                                     int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c, hipStream_t s) {
  using Acc = typename AccOf<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  if (d0 % VEC || d1 % VEC || !aligned16(X0) || !aligned16(X1) || !aligned16(W0) || !aligned16(W1)) return 1;
  const int tiles_n = cdiv(c, 32), ntiles = cdiv(B, 32) * tiles_n;
  const bool vec_c = (c % 4 == 0) && aligned16(E) && aligned16(u) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  embrace_fwd_stream_kernel<T><<<ntiles, kThreads, 0, s>>>((const T*)X0, (const T*)X1, (const T*)W0, (const T*)W1, (const Acc*)b0,
                                                          (const Acc*)b1, cdf0, u, seed, step_val, step_dev, row0, (T*)E, code, B, d0,
                                                          d1, c, tiles_n, ntiles, vec_c);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

}  // namespace emb
