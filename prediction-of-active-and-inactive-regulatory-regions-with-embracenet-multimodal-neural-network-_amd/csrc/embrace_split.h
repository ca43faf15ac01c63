// Fused EmbraceNet forward, "K split over waves" form (split_core.h) -- the kernel every shape with 16-byte-aligned
// rows takes.  Replaces EmbraceNetMultimodal.py:52-60, :80-88 like the tiled kernel in embrace_fwd.hip.
//
// One workgroup = one (16*MI) x (16*NI) tile of E.  Wave w accumulates the docking_1 product over the 128-byte chunks
// w, w+4, ... of the d1 range from its private LDS-DMA ring (NSTAGE chunks of X1 and W1 rows in flight per wave, whole
// 128-byte lines, no VGPR staging, no barrier) and its share of the short docking_0 product straight from fragment-shaped
// register loads requested before the ring is primed.  The eight partial tiles (4 waves x 2 modalities) meet in LDS, are
// summed in wave order (deterministic) and finished by the shared epilogue (embrace_epilogue.h: threshold, Philox / injected
// uniform, bias, ReLU, select, E and code stores).
#pragma once
#include "embrace_epilogue.h"
#include "split_core.h"

namespace emb {

template <typename T, int MI, int NI> struct SplitFwdCfg {   // epilogue geometry
  using type = TileCfg<T, 16 * MI, 16 * NI, Mma<T>::KSTEP * 4, 1, 1, 4, false, false>;
};

// fragment-shaped loads of one k-block (4 lane groups x 16 bytes) of the modality-0 operands; guarded element-wise when
// rows are not 16-byte multiples (d0 = 4 in bf16)
template <typename T, int NT>
__device__ __forceinline__ void frag_block_load(const T* __restrict__ M, int ld, int row0, int nrows, int kblock, bool vec_ok,
                                                typename Vec16<T>::type (&out)[NT]) {
  using V = typename Vec16<T>::type;
  constexpr int VEC = Elem<T>::VEC;
  const int lane = threadIdx.x & 63, r = lane & 15, g = lane >> 4;
  const int k = (kblock * 4 + g) * VEC;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int row = row0 + 16 * t + r;
    V v;
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = (T)0.0f;
    if (row < nrows) {
      const T* p = M + (long)row * ld + k;
      if (vec_ok) {
        if (k < ld) v = *reinterpret_cast<const V*>(p);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (k + e < ld) v[e] = p[e];
      }
    }
    out[t] = v;
  }
}

template <typename T, int MI, int NI, int NSTAGE>
__global__ __launch_bounds__(kThreads, 1) void embrace_fwd_split_kernel(
    const T* __restrict__ X0, const T* __restrict__ X1, const T* __restrict__ W0, const T* __restrict__ W1,
    const typename AccOf<T>::type* __restrict__ b0, const typename AccOf<T>::type* __restrict__ b1, const SelArgs sel,
    const double* __restrict__ u, uint64_t seed, uint64_t step_val, const uint64_t* __restrict__ step_dev, int64_t grow0,
    T* __restrict__ E, uint8_t* __restrict__ code, int B, int d0, int d1, int c, int tiles_n, int ntiles, bool vec0, bool vec_c,
    int thr_off) {
  using Cfg = typename SplitFwdCfg<T, MI, NI>::type;
  using Mm = Mma<T>;
  using Acc = typename AccOf<T>::type;
  using V = typename Vec16<T>::type;
  constexpr int TM = 16 * MI, TN = 16 * NI;
  constexpr int A_BYTES = TM * 128, STAGE = (TM + TN) * 128;
  constexpr int G = (TM + TN) / 8;               // LDS-DMA instructions per chunk
  constexpr int KC = 128 / (int)sizeof(T);       // elements per chunk
  constexpr int KV = Elem<T>::VEC * 4;           // elements per modality-0 k-block
  extern __shared__ __attribute__((aligned(16))) char smem[];

  EMB_STAMP(2);
  EMB_STAMP_KIND(0);
  const int tile = xcd_remap(blockIdx.x, ntiles);
  const int row0 = (tile / tiles_n) * TM, col0 = (tile % tiles_n) * TN;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t ring = (uint32_t)(uintptr_t)smem + (uint32_t)(wave * NSTAGE * STAGE);

  // ---- modality 1: prime this wave's ring (chunks wave, wave + 4, ...)
  const int n1 = (d1 + KC - 1) / KC;
  const int n_my = n1 > wave ? (n1 - wave + 3) / 4 : 0;
  const int rowbytes = d1 * (int)sizeof(T);
  DmaImage<TM> dx;
  DmaImage<TN> dw;
  dx.init((uint32_t)rowbytes, 0, 128, lane);
  dw.init((uint32_t)rowbytes, 0, 128, lane);
  const char* Xo = reinterpret_cast<const char*>(X1) + (long)row0 * rowbytes;    // rows >= B / >= c read zeros (range check)
  const char* Wo = reinterpret_cast<const char*>(W1) + (long)col0 * rowbytes;
  const long xbytes = (long)(B - row0) * rowbytes, wbytes = (long)(c - col0) * rowbytes;
  auto issue = [&](int i, uint32_t st) {         // i-th chunk of this wave -> stage st
    const int cb = (wave + 4 * i) * 128;
    if (cb + 128 <= rowbytes) {
      dx.issue(Xo + cb, dma_nrec(xbytes - cb), st);
      dw.issue(Wo + cb, dma_nrec(wbytes - cb), st + A_BYTES);
    } else {
      dx.issue_tail(Xo + cb, dma_nrec(xbytes - cb), rowbytes - cb, st);
      dw.issue_tail(Wo + cb, dma_nrec(wbytes - cb), rowbytes - cb, st + A_BYTES);
    }
  };
#pragma unroll
  for (int s = 0; s < NSTAGE; ++s)
    if (s < n_my) issue(s, ring + s * STAGE);

  // ---- row thresholds of the tile -> LDS (while the first chunks are in flight)
  const uint64_t step = step_val + (step_dev ? *step_dev : 0);
  float* thr = reinterpret_cast<float*>(smem + thr_off);
  if ((int)threadIdx.x < TM) {
    const int row = row0 + (int)threadIdx.x;
    float cdf = 0.0f;
    if (row < B) {
      if (sel.cdf0 != nullptr) {
        cdf = sel.cdf0[row];
      } else {
        bool ok;
        cdf = select_cdf(sel, row, seed, step, grow0, &ok);
        if (!ok && col0 == 0) atomicOr(sel.status, EMB_STATUS_INVALID_DISTRIBUTION);   // (one column tile per row reports)
      }
    }
    thr[threadIdx.x] = cdf;
  }

  typename Mm::AccV acc0[MI][NI], acc1[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) { acc0[mi][ni][q] = 0; acc1[mi][ni][q] = 0; }

  // ---- modality 0 (d0 <= 256): fragment-shaped register loads, k-blocks wave, wave + 4, ...
  const int nk0 = (d0 + KV - 1) / KV;
  for (int kb = wave; kb < nk0; kb += 4) {
    V a0[MI], w0[NI];
    frag_block_load<T, MI>(X0, d0, row0, B, kb, vec0, a0);
    frag_block_load<T, NI>(W0, d0, col0, c, kb, vec0, w0);
    mma_frags<T, MI, NI>(a0, w0, acc0);
  }

  EMB_STAMP(3);
  const RmLane rl = rm_lane(lane);
  for (int it = 0; it < n_my; ++it) {
    wait_chunks_in_flight<G>(min(n_my - it - 1, NSTAGE - 1));   // chunk `it` has landed
    if (it == 0) EMB_STAMP(4);
    const uint32_t st = ring + (uint32_t)((it % NSTAGE) * STAGE);
    V a[2][MI], b[2][NI];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a[h][mi] = lds_read16<T>(st + mi * 2048 + rl.off[h]);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b[h][ni] = lds_read16<T>(st + A_BYTES + ni * 2048 + rl.off[h]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // fragments are in registers: the stage may be refilled
    if (it + NSTAGE < n_my) issue(it + NSTAGE, st);
    mma_frags<T, MI, NI>(a[0], b[0], acc1);
    mma_frags<T, MI, NI>(a[1], b[1], acc1);
  }
  EMB_STAMP(5);
  __syncthreads();                               // every wave is done with its ring

  // ---- partial tiles -> LDS (one region per wave and modality)
  Acc* part = reinterpret_cast<Acc*>(smem);      // [wave][modality][TM][CS]
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int idx = (mi * 16 + Mm::acc_row(lane, q)) * Cfg::CS + ni * 16 + (lane & 15);
        part[(wave * 2 + 0) * Cfg::SLAB + idx] = acc0[mi][ni][q];
        part[(wave * 2 + 1) * Cfg::SLAB + idx] = acc1[mi][ni][q];
      }
  __syncthreads();
  EMB_STAMP(6);

  // ---- sum in wave order (deterministic), select, bias, ReLU, stores: 4 consecutive columns of one row per thread
  const uint64_t stream = rng_stream(step, EMB_RNG_SELECT);
  for (int gidx = threadIdx.x; gidx < TM * TN / 4; gidx += kThreads) {
    const int r = gidx / (TN / 4), cq = (gidx % (TN / 4)) * 4;
    const int row = row0 + r, col = col0 + cq;
    if (row >= B || col >= c) continue;
    Acc v0[4], v1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int idx = r * Cfg::CS + cq + j;
      v0[j] = ((part[idx] + part[2 * Cfg::SLAB + idx]) + part[4 * Cfg::SLAB + idx]) + part[6 * Cfg::SLAB + idx];
      v1[j] = ((part[Cfg::SLAB + idx] + part[3 * Cfg::SLAB + idx]) + part[5 * Cfg::SLAB + idx]) + part[7 * Cfg::SLAB + idx];
    }
    const float cdf = thr[r];
    const long base = (long)row * c + col;
    const int nval = min(4, c - col);
    bool s1[4];
    if (u != nullptr) {                          // parity mode: the host generator's fp64 uniforms, ATen's compare
      const double t = (double)cdf;
#pragma unroll
      for (int j = 0; j < 4; ++j) s1[j] = j < nval ? (t < u[base + j]) : false;
    } else {
      const uint64_t t32 = select_threshold32(cdf);
      uint32_t w4[4];
      select_words4(seed, stream, (uint64_t)(grow0 + row) * (uint64_t)c + (uint64_t)col, w4);
#pragma unroll
      for (int j = 0; j < 4; ++j) s1[j] = t32 < (uint64_t)w4[j];
    }
    T ev[4];
    uint8_t cv[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int cc = min(col + j, c - 1);
      const Acc pre = s1[j] ? v1[j] + b1[cc] : v0[j] + b0[cc];
      const bool act = pre > (Acc)0;
      ev[j] = (T)(act ? pre : (Acc)0);
      cv[j] = (uint8_t)((s1[j] ? EMB_CODE_IDX : 0) | (act ? (EMB_CODE_ACTIVE | (s1[j] ? EMB_CODE_KEEP1 : EMB_CODE_KEEP0)) : 0));
    }
    if (nval == 4 && vec_c) {
      typedef T TV4 __attribute__((ext_vector_type(4)));
      TV4 o = {ev[0], ev[1], ev[2], ev[3]};
      *reinterpret_cast<TV4*>(E + base) = o;
      *reinterpret_cast<uint32_t*>(code + base) = (uint32_t)cv[0] | ((uint32_t)cv[1] << 8) | ((uint32_t)cv[2] << 16) | ((uint32_t)cv[3] << 24);
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (j < nval) {
          E[base + j] = ev[j];
          code[base + j] = cv[j];
        }
    }
  }
  EMB_STAMP(8);
}

template <typename T, int MI, int NI, int NSTAGE>
static int launch_fwd_split(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1, const void* b1,
                            const SelArgs& sel, const double* u, uint64_t seed, uint64_t step_val, const uint64_t* step_dev,
                            int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c, hipStream_t s) {
  using Cfg = typename SplitFwdCfg<T, MI, NI>::type;
  using Acc = typename AccOf<T>::type;
  constexpr int TM = 16 * MI, TN = 16 * NI, VEC = Elem<T>::VEC;
  constexpr int ring_bytes = 4 * NSTAGE * (TM + TN) * 128, part_bytes = 8 * Cfg::SLAB * (int)sizeof(Acc);
  constexpr int thr_off = ring_bytes > part_bytes ? ring_bytes : part_bytes;   // row thresholds live behind ring / partials
  constexpr int lds = thr_off + TM * 4;
  static_assert(lds <= 160 * 1024, "LDS budget");
  static_assert((NSTAGE - 1) * ((TM + TN) / 8) <= 63, "vmcnt range");
  const int tiles_n = cdiv(c, TN), ntiles = cdiv(B, TM) * tiles_n;
  const bool vec0 = (d0 % VEC == 0) && aligned16(X0) && aligned16(W0);
  const bool vec_c = (c % 4 == 0) && aligned16(E) && aligned16(u) && ((reinterpret_cast<uintptr_t>(code) & 3u) == 0);
  auto kern = &embrace_fwd_split_kernel<T, MI, NI, NSTAGE>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    attr_set = true;
  }
  kern<<<ntiles, kThreads, lds, s>>>((const T*)X0, (const T*)X1, (const T*)W0, (const T*)W1, (const Acc*)b0, (const Acc*)b1, sel, u,
                                     seed, step_val, step_dev, row0, (T*)E, code, B, d0, d1, c, tiles_n, ntiles, vec0, vec_c, thr_off);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}

// returns 1 when the shapes do not qualify (caller falls back to the LDS-tiled kernel)
template <typename T>
static int fwd_split_dispatch(const void* X0, const void* X1, const void* W0, const void* b0, const void* W1, const void* b1,
                              const SelArgs& sel, const double* u, uint64_t seed, uint64_t step_val, const uint64_t* step_dev,
                              int64_t row0, void* E, uint8_t* code, int B, int d0, int d1, int c, hipStream_t s) {
  constexpr int VEC = Elem<T>::VEC;
  if (d1 % VEC || !aligned16(X1) || !aligned16(W1)) return 1;
  // largest tile that still gives every CU a workgroup (256 CUs)
  const long t64 = (long)cdiv(B, 64) * cdiv(c, 64), t6432 = (long)cdiv(B, 64) * cdiv(c, 32);
  if constexpr (sizeof(T) <= 4) {   // (fp64: the partial tiles of the larger shapes exceed LDS; parity path, 32x32 only)
    if (t64 >= 200)
      return launch_fwd_split<T, 4, 4, 2>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
    if (t6432 >= 200)
      return launch_fwd_split<T, 4, 2, 3>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
  }
  return launch_fwd_split<T, 2, 2, 4>(X0, X1, W0, b0, W1, b1, sel, u, seed, step_val, step_dev, row0, E, code, B, d0, d1, c, s);
}

}  // namespace emb
