// Persistent ring GEMM: the EmbraceNet backward on PRE-MASKED gradients (bf16).
//
//   dX_m[B,d_m] = dD_m   W_m      "dgrad"  A = dD_m row-major [M = B][K = c],  Bm = W_m K-major [c][d_m]
//   dW_m[c,d_m] = dD_m^T X_m      "wgrad"  A = dD_m K-major  [K = B][M = c],   Bm = X_m K-major [B][d_m]   (+ db_m = sum_b dD_m)
// with dD_m = dE * [idx == m] * [pre_m > 0] written ONCE by the kernel that produces dE (the classifier head, head.hip, or
// emb_embrace_premask) instead of being re-derived from dE and the code bytes by every wave for every fragment: the
// mask arithmetic of embrace_bwd_split.h (28 vector instructions per 8 MFMAs -- its main loop was bound by vector issue, not by
// the matrix pipe) and the code images (a fifth of the staged bytes) are gone, the four jobs are plain GEMMs.
// Replaces autograd through EmbraceNetMultimodal.py:52-60,80-88 (utils/training_models_multimodal.py:156).
//
// Structure.  One PERSISTENT workgroup per CU (8 waves as 4 x 2, two per SIMD) walks a static list of 128 x 128 output tiles of
// all four jobs (list position v = workgroup + i * grid, XCD-aware order).  All operand traffic of a workgroup is ONE stream of
// 32 KB stages (64 reduction indices of a tile: m-side image 16 KB + n-side image 16 KB) through a ring of five LDS slots,
// requested by LDS-DMA four stages ahead of the stage being multiplied -- across tile boundaries: while a tile's last stages
// are multiplied and its accumulators leave, the next tile's first stages are already landing.  One raw s_barrier per stage;
// a wave waits for its own four DMA instructions of a stage by a counted vmcnt.  Finished accumulators go to memory straight from
// registers (buffer stores: the range check drops rows past the matrix; 8- / 16-byte pieces per lane, four adjacent column
// tiles complete every 128-byte line) -- no staging tile, no extra barrier, and they overlap the next tile's multiplies.
// Per-workgroup set-up is a handful of scalar instructions per tile (job constants are kernel arguments; magic divisions).
#pragma once
#include "embrace_bwd_split.h"

namespace emb {

constexpr int kGjThreads = 512;
constexpr int kGjKC = 64;                         // reduction indices per stage
constexpr int kGjStage = 32 * 1024;               // m-side image (16 KB) + n-side image (16 KB)
constexpr int kGjSlots = 5;
constexpr int kGjLds = kGjSlots * kGjStage;       // 160 KB
constexpr int kGjDma = 4;                         // LDS-DMA instructions per wave and stage
constexpr int kGjAhead = kGjSlots - 1;            // stages in flight ahead of the one being multiplied

struct GJob {
  const char* A;        // m-side operand (pre-masked gradient)
  const char* Bm;       // n-side operand, K-major [K][N]
  char* C;              // dgrad: dX [M][ldc] bf16;  wgrad: dW [M][ldc] f32, or the slabs [S][M][ldc] when S > 1
  float* bias;          // wgrad, S == 1: db [M]
  long slice_stride;    // wgrad, S > 1: bytes between the slabs of two slices
  int M, N, K;          // C is M x N, reduction length K    (dgrad: B, d, c;  wgrad: c, d, B)
  int lda, ldb, ldc;    // row pitches in elements
  int tiles_m, tiles_n, tiles;   // 128-wide tiles; tiles = tiles_m * tiles_n
  int S, kper;          // wgrad: the reduction is cut into S slices of kper rows
  int first, count;     // this job's range of the launch's tile list; count = tiles * S
  int kind;             // 0 dgrad (A row-major), 1 wgrad (A K-major)
  int m_fast;           // tile order inside a slice: 1 = m tile fastest (tiles sharing an n-side panel are neighbours), 0 = n tile fastest
  uint32_t magic_tiles, magic_inner;   // ceil(2^32 / tiles), ceil(2^32 / (m_fast ? tiles_m : tiles_n))
};

__device__ __forceinline__ GJob gj_pick(int k, const GJob& a, const GJob& b, const GJob& c, const GJob& d) {
  GJob j;
#define EMB_PICK(f) j.f = k == 0 ? a.f : (k == 1 ? b.f : (k == 2 ? c.f : d.f))
  EMB_PICK(A); EMB_PICK(Bm); EMB_PICK(C); EMB_PICK(bias); EMB_PICK(slice_stride); EMB_PICK(M); EMB_PICK(N); EMB_PICK(K);
  EMB_PICK(lda); EMB_PICK(ldb); EMB_PICK(ldc); EMB_PICK(tiles_m); EMB_PICK(tiles_n); EMB_PICK(tiles); EMB_PICK(S); EMB_PICK(kper);
  EMB_PICK(first); EMB_PICK(count); EMB_PICK(kind); EMB_PICK(m_fast); EMB_PICK(magic_tiles); EMB_PICK(magic_inner);
#undef EMB_PICK
  return j;
}

// one tile of the list, decoded (all wave-uniform)
struct GTile {
  int valid, kind;
  int m0, n0, slice, k_begin, k_end, nstages;
  int M, N, lda, ldb, ldc, S;
  const char *A, *Bm;
  char* C;
  float* bias;
};

__device__ __forceinline__ GTile gj_decode(int v, int total, const GJob& j0, const GJob& j1, const GJob& j2, const GJob& j3) {
  GTile t;
  t.valid = v < total;
  const int id = t.valid ? xcd_remap(v, total) : 0;
  const int k = (id >= j0.first + j0.count) + (id >= j1.first + j1.count) + (id >= j2.first + j2.count);
  const GJob j = gj_pick(k, j0, j1, j2, j3);
  const int q = id - j.first;
  const int slice = j.S > 1 ? div_magic(q, j.magic_tiles) : 0;
  const int t2 = q - slice * j.tiles;
  const int inner = j.m_fast ? j.tiles_m : j.tiles_n;
  const int hi = div_magic(t2, j.magic_inner), lo = t2 - hi * inner;
  const int tm = j.m_fast ? lo : hi, tn = j.m_fast ? hi : lo;
  t.kind = j.kind;
  t.m0 = tm * 128;
  t.n0 = tn * 128;
  t.slice = slice;
  t.k_begin = slice * j.kper;
  t.k_end = min(j.K, t.k_begin + j.kper);
  t.nstages = (t.k_end - t.k_begin + kGjKC - 1) / kGjKC;
  t.M = j.M; t.N = j.N; t.lda = j.lda; t.ldb = j.ldb; t.ldc = j.ldc; t.S = j.S;
  t.A = j.A; t.Bm = j.Bm;
  t.C = j.C + (long)slice * j.slice_stride;
  t.bias = j.bias;
  return t;
}

// The request side of the ring: the tile whose stages are being requested, and where its next stage starts.
struct GFeed {
  GTile t;
  int v;                 // list position of t
  int stage;             // next stage of t to request
  int slot;              // ring slot of the next request
  // per-lane offsets of this wave's four instructions (recomputed per tile)
  DmaImage<128, 8, 8> arow;       // dgrad: A image, 128 m rows x 128 B of k
  DmaImage<64, 8, 8> akm[2];      // wgrad: A image halves, 64 k rows x 128 B (64 m)
  DmaImage<64, 8, 8> bkm[2];      // n-side image halves, 64 k rows x 128 B (64 n)
};

__device__ __forceinline__ void gj_feed_tile(GFeed& f, int lane, int wave) {
  const GTile& t = f.t;
  if (!t.valid) return;
  if (t.kind == 0) f.arow.init((uint32_t)t.lda * 2, 0, 128, lane, wave);
  else {
    f.akm[0].init((uint32_t)t.lda * 2, t.m0 * 2, t.M * 2, lane, wave);
    f.akm[1].init((uint32_t)t.lda * 2, t.m0 * 2 + 128, t.M * 2, lane, wave);
  }
  f.bkm[0].init((uint32_t)t.ldb * 2, t.n0 * 2, t.N * 2, lane, wave);
  f.bkm[1].init((uint32_t)t.ldb * 2, t.n0 * 2 + 128, t.N * 2, lane, wave);
}

// request the next stage of the feed's tile into its slot; returns false when the list is exhausted
__device__ __forceinline__ bool gj_request(GFeed& f, uint32_t lds0, int grid, int total, int lane, int wave, const GJob& j0,
                                           const GJob& j1, const GJob& j2, const GJob& j3) {
  if (!f.t.valid) return false;
  const GTile& t = f.t;
  const uint32_t buf = lds0 + (uint32_t)(f.slot * kGjStage);
  const int k0 = t.k_begin + f.stage * kGjKC;                      // first reduction index of the stage
  const long brem = ((long)t.k_end - k0) * t.ldb * 2;              // k rows at or beyond k_end read zeros
  const char* borg = t.Bm + (long)k0 * t.ldb * 2;
  if (t.kind == 0) {
    const char* aorg = t.A + ((long)t.m0 * t.lda + k0) * 2;         // rows >= M read zeros (range check)
    const long arem = ((long)t.M - t.m0) * t.lda * 2 - (long)k0 * 2;
    if (k0 + kGjKC <= t.k_end) f.arow.issue(aorg, dma_nrec(arem), buf);
    else f.arow.issue_tail(aorg, dma_nrec(arem), (t.k_end - k0) * 2, buf);   // k beyond K reads zeros (not the next row)
  } else {
    const char* aorg = t.A + (long)k0 * t.lda * 2;
    const long arem = ((long)t.k_end - k0) * t.lda * 2;
    f.akm[0].issue(aorg, dma_nrec(arem), buf);
    f.akm[1].issue(aorg, dma_nrec(arem), buf + 8192);
  }
  f.bkm[0].issue(borg, dma_nrec(brem), buf + 16384);
  f.bkm[1].issue(borg, dma_nrec(brem), buf + 16384 + 8192);
  f.slot = f.slot + 1 == kGjSlots ? 0 : f.slot + 1;
  if (++f.stage == t.nstages) {
    f.v += grid;
    f.stage = 0;
    f.t = gj_decode(f.v, total, j0, j1, j2, j3);
    gj_feed_tile(f, lane, wave);
  }
  return true;
}

// wait until this wave's DMA instructions of the stage about to be multiplied have landed: `younger` later stages (0 .. 4) may
// stay in flight.  Accumulator stores of a finished tile are younger still and only make the wait conservative.
__device__ __forceinline__ void gj_wait(int younger) {
  if (younger >= 4) EMB_WAIT_VMCNT(4 * kGjDma);
  else if (younger == 3) EMB_WAIT_VMCNT(3 * kGjDma);
  else if (younger == 2) EMB_WAIT_VMCNT(2 * kGjDma);
  else if (younger == 1) EMB_WAIT_VMCNT(kGjDma);
  else EMB_WAIT_VMCNT(0);
}

struct GStep {                                    // fragments of one k-step (32 reduction indices): 12 LDS reads
  bf16x8 a[4], b[2];
};

__global__ __launch_bounds__(kGjThreads, 2) void gemm_jobs_kernel(const GJob j0, const GJob j1, const GJob j2, const GJob j3,
                                                                   int total) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;           // this wave: m rows 32 wr .. +32, n columns 64 wc .. +64
  const int grid = gridDim.x;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;

  GFeed feed;
  feed.v = blockIdx.x;
  feed.stage = 0;
  feed.slot = 0;
  feed.t = gj_decode(feed.v, total, j0, j1, j2, j3);
  gj_feed_tile(feed, lane, wave);
  int requested = 0;                                 // stages requested so far / multiplied so far
#pragma unroll 1
  for (int i = 0; i < kGjAhead; ++i) requested += gj_request(feed, lds0, grid, total, lane, wave, j0, j1, j2, j3) ? 1 : 0;

  // per-lane parts of the fragment reads (tile-independent)
  const RmLane rl = rm_lane(lane);
  const KmLane kl = km_lane(lane);
  uint32_t boff[4];                                  // n-side column tiles of this wave inside its image half
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) boff[ni] = km_off(kl, ni);
  const int chalf = wr >> 1, ct0 = (wr & 1) * 2;     // wgrad: this wave's m rows live in image half chalf, column tiles ct0, ct0 + 1
  uint32_t aoff[2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) aoff[ci] = km_off(kl, ct0 + ci);
  [[maybe_unused]] const int r = lane & 15, g = lane >> 4;       // (used by the device pass only)
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  int done = 0, slot = 0;
  int v = blockIdx.x;
#pragma unroll 1
  for (;;) {
    const GTile t = gj_decode(v, total, j0, j1, j2, j3);
    if (!t.valid) break;
    f32x4 acc[4][2], accb[2];                        // [n tile][m tile]; accb: the bias gradient (wgrad, first n tile)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int q = 0; q < 4; ++q) accb[mi][q] = 0.0f;
#pragma unroll
      for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[ni][mi][q] = 0.0f;
    }
    const bool with_bias = t.kind == 1 && t.n0 == 0 && wc == 0;
#pragma unroll 1
    for (int s = 0; s < t.nstages; ++s) {
      gj_wait(requested - done - 1);
      __builtin_amdgcn_s_barrier();                  // raw barrier: a __syncthreads() would drain the younger stages
      asm volatile("" ::: "memory");
      // every wave has finished the stage multiplied before this one: its slot takes the next request
      requested += gj_request(feed, lds0, grid, total, lane, wave, j0, j1, j2, j3) ? 1 : 0;
      const uint32_t buf = lds0 + (uint32_t)(slot * kGjStage);
      const int nsteps = min(2, (t.k_end - t.k_begin - s * kGjKC + 31) / 32);   // k-steps of this stage that hold data
      const uint32_t bimg = buf + 16384 + (uint32_t)(wc * 8192);
      GStep f[2];
      auto load = [&](int h, GStep& x) {
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) x.a[ni] = km_frag_at(bimg + boff[ni] + (uint32_t)(h * 4096));
        if (t.kind == 0) {
          const uint32_t aimg = buf + (uint32_t)(wr * 2 * 2048) + rl.off[h];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) x.b[mi] = lds_read16<__bf16>(aimg + mi * 2048);
        } else {
          const uint32_t aimg = buf + (uint32_t)(chalf * 8192 + h * 4096);
#pragma unroll
          for (int ci = 0; ci < 2; ++ci) x.b[ci] = km_frag_at(aimg + aoff[ci]);
        }
      };
      load(0, f[0]);
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h < nsteps) {
          if (h + 1 < nsteps) load(h + 1, f[(h + 1) & 1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
            for (int ni = 0; ni < 4; ++ni)
              acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[h & 1].a[ni], f[h & 1].b[mi], acc[ni][mi], 0, 0, 0);
            if (with_bias) accb[mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, f[h & 1].b[mi], accb[mi], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      slot = slot + 1 == kGjSlots ? 0 : slot + 1;
      ++done;
    }
    // the tile leaves from registers: lane (r, g) holds row m0 + 32 wr + 16 mi + r, columns n0 + 64 wc + 16 ni + 4 g .. + 3
#if defined(__HIP_DEVICE_COMPILE__)
    {
      const int esz = t.kind == 0 ? 2 : 4;
      const long rem = ((long)t.M - t.m0) * t.ldc * esz - (long)t.n0 * esz;      // rows >= M are dropped by the range check
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(t.C + ((long)t.m0 * t.ldc + t.n0) * esz), 0, dma_nrec(rem), 0x00020000);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi) {
        const uint32_t rowoff = (uint32_t)(wr * 32 + mi * 16 + r) * (uint32_t)t.ldc;
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) {
          const int ncol = wc * 64 + ni * 16 + 4 * g;
          const bool inside = t.n0 + ncol < t.N;                                  // N % 8 == 0: four columns are inside together
          const uint32_t off = inside ? (rowoff + (uint32_t)ncol) * (uint32_t)esz : kDmaInvalid;
          if (t.kind == 0) {
            bf16x4 o;
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q] = (__bf16)acc[ni][mi][q];
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, o), rs, off, 0, 0);
          } else {
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, acc[ni][mi]), rs, off, 0, 0);
          }
        }
      }
      if (with_bias && g == 0) {                                                  // every register of accb holds the column sum
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const int crow = t.m0 + wr * 32 + mi * 16 + r;
          if (crow < t.M) {
            if (t.S > 1) reinterpret_cast<float*>(t.C)[(long)crow * t.ldc + t.N] = accb[mi][0];
            else t.bias[crow] = accb[mi][0];
          }
        }
      }
    }
#endif
    v += grid;
  }
}

// host: number of workgroups that can be resident (one per CU: the ring takes the whole LDS)
static int gj_grid_limit() {
  static int cus = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 256;
    return n;
  }();
  return cus;
}

// returns 1 when the shapes do not qualify (the caller keeps its other kernels)
static int gemm_jobs_bwd_dispatch(const void* dD0, const void* dD1, const void* X0, const void* X1, const void* W0, const void* W1,
                                  void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B,
                                  int d0, int d1, int c, int force_S, hipStream_t s) {
  if (c % 16 || d0 % 8 || d1 % 8) return 1;
  const void* ptrs[] = {dD0, dD1, X0, X1, W0, W1, dX0, dX1, dW0, dW1, ws};
  for (const void* p : ptrs)
    if (p != nullptr && !aligned16(p)) return 1;
  const int dmax = d1 > d0 ? d1 : d0;
  if ((long)B * (dmax > c ? dmax : c) * 4 >= (1l << 31) || (long)c * (dmax + 4) * 4 >= (1l << 31)) return 1;   // 32-bit buffer offsets
  if ((long)cdiv(B, 128) * cdiv(dmax, 128) >= 65536 || (long)cdiv(c, 128) * cdiv(dmax, 128) * 16 >= 65536) return 1;   // div_magic range
  int n = 0;
  int64_t ws_used = 0;
  struct SlabInfo { float* slab; int pitch; int S; } slabs[2] = {{nullptr, 0, 1}, {nullptr, 0, 1}};
  auto wgrad = [&](const void* dD, const void* X, void* dW, void* db, int d, int m) {
    GJob j{};
    j.A = (const char*)dD; j.Bm = (const char*)X; j.C = (char*)dW; j.bias = (float*)db;
    j.M = c; j.N = d; j.K = B; j.lda = c; j.ldb = d; j.ldc = d;
    j.tiles_m = cdiv(c, 128); j.tiles_n = cdiv(d, 128); j.tiles = j.tiles_m * j.tiles_n;
    j.S = 1; j.kper = B; j.kind = 1; j.m_fast = 1;
    // slices of 256 batch rows (four stages) unless that leaves the CUs short of tiles or the scratch is too small
    int S = force_S > 0 ? force_S : cdiv(B, 256);
    if (S > 16) S = 16;
    const int pitch = cdiv(d + 1, 4) * 4;
    const int64_t per = (int64_t)c * pitch * 4;
    if (ws == nullptr) S = 1;
    else if ((int64_t)S * per > ws_bytes - ws_used) S = (int)((ws_bytes - ws_used) / per);
    if (S > 1) {
      j.kper = cdiv(cdiv(B, S), kGjKC) * kGjKC;
      j.S = cdiv(B, j.kper);
      if (j.S > 1) {
        j.C = (char*)ws + ws_used;
        j.ldc = pitch;
        j.slice_stride = per;
        ws_used += (int64_t)j.S * per;
        slabs[m] = SlabInfo{(float*)j.C, pitch, j.S};
      } else {
        j.kper = B;
      }
    }
    j.first = n;
    j.count = j.tiles * j.S;
    n += j.count;
    j.magic_tiles = make_magic(j.tiles);
    j.magic_inner = make_magic(j.tiles_m);
    return j;
  };
  auto dgrad = [&](const void* dD, const void* W, void* dX, int d) {
    GJob j{};
    j.A = (const char*)dD; j.Bm = (const char*)W; j.C = (char*)dX;
    j.M = B; j.N = d; j.K = c; j.lda = c; j.ldb = d; j.ldc = d;
    j.tiles_m = cdiv(B, 128); j.tiles_n = cdiv(d, 128); j.tiles = j.tiles_m * j.tiles_n;
    j.S = 1; j.kper = c; j.kind = 0;
    j.m_fast = d > B ? 1 : 0;            // neighbours in the list share the LARGER operand (W panel when d > B, else the gradient rows)
    j.first = n;
    j.count = dX != nullptr ? j.tiles : 0;
    n += j.count;
    j.magic_tiles = make_magic(j.tiles);
    j.magic_inner = make_magic(j.m_fast ? j.tiles_m : j.tiles_n);
    return j;
  };
  // longest tiles first: a wgrad tile multiplies kper / 64 stages, a dgrad tile c / 64
  const GJob wg1 = wgrad(dD1, X1, dW1, db1, d1, 1);
  const GJob dg1 = dgrad(dD1, W1, dX1, d1);
  const GJob wg0 = wgrad(dD0, X0, dW0, db0, d0, 0);
  const GJob dg0 = dgrad(dD0, W0, dX0, d0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_jobs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kGjLds);
    attr_set = true;
  }
  const int grid = n < gj_grid_limit() ? n : gj_grid_limit();
  gemm_jobs_kernel<<<grid, kGjThreads, kGjLds, s>>>(wg1, dg1, wg0, dg0, n);
  EMB_CHECK_LAUNCH();
  for (int m = 1; m >= 0; --m) {
    if (slabs[m].S > 1) {
      const int d = m ? d1 : d0;
      ReduceJob j{};
      j.in = slabs[m].slab; j.out[0] = m ? dW1 : dW0; j.out[1] = m ? db1 : db0;
      j.per = (long)c * slabs[m].pitch; j.S = slabs[m].S; j.kind = RJ_LINEAR; j.iv[0] = d; j.iv[1] = slabs[m].pitch;
      const int rc = reduce_submit(j, false, s);
      if (rc != EMB_OK) return rc;
    }
  }
  return EMB_OK;
}

}  // namespace emb
