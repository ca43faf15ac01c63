// EmbraceNet backward for bf16, LDS-DMA form (split_core.h).  One launch, four kinds of 64 x 64 tile job:
//   dgrad_m  dX_m[B,d_m]  = dD_m   W_m      64 rows x 64 columns, reduction over c
//   wgrad_m  dW_m[c,d_m]  = dD_m^T X_m      64 (c) x 64 columns, reduction over the batch rows, optionally cut into S slices
//            db_m[c]      = sum_b dD_m      (per-slice slabs summed later: reduce.hip, or by the optimizer launch)
// with dD_m = dE * [idx == m] * [pre_m > 0] never materialised: dE and the forward's code bytes travel to LDS untouched
// (LDS-DMA) and the mask is applied to each MFMA fragment as it is read.  The code byte carries the two per-modality keep
// bits the forward kernel prepared (EMB_CODE_KEEP0/1 = bits 6 / 7), so a fragment mask is v_perm_b32 (byte -> high byte of
// a 16-bit lane), a packed arithmetic shift and an AND per dword.
//
// A tile is processed in ROUNDS of 256 reduction indices: the four waves request the whole round (80 KB: operand images
// plus code bytes, every request issued before anything is waited for), meet at one barrier and then each wave computes its
// share of the OUTPUT from the shared images -- no partial tiles, no reduction through LDS, results go from the accumulators
// straight to memory as 8 / 16-byte stores:
//   dgrad: wave w owns rows 16w .. 16w+15 (its dE / code fragments are masked once), all 64 columns;
//          MFMA operands A = W_m^T fragment (ds_read_b64_tr_b16 of the K-major image), B = dD fragment -> C[col][row]
//   wgrad: waves 2 x 2, each a 32 (c) x 32 (columns) quadrant; both operands K-major (image rows = batch rows), the code
//          bytes of a transposed fragment come from the matching 8-bit transposing read (ds_read_b64_tr_b8);
//          A = X fragment, B = dD^T fragment -> C[col][c]; bias gradient = a ones-fragment product, column tile 0 only.
// Two workgroups fit a CU (80 KB each), so one tile's request / wait phase overlaps the other's MFMA phase.
// Replaces autograd through EmbraceNetMultimodal.py:52-60,80-88 (utils/training_models_multimodal.py:156) like embrace_bwd.hip.
#pragma once
#include "reduce.h"
#include "split_core.h"

namespace emb {

typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

struct SplitJob {
  const __bf16* Bptr;   // dgrad: W_m [c][d];  wgrad: X_m [B][d]
  void* C;              // dgrad: dX_m [B][d] bf16;  wgrad: dW_m [c][d] f32, or the slabs [S][c][pitch] when S > 1
  float* bias;          // wgrad, S == 1: db_m [c]
  int d, tiles_n, tiles;   // columns; 64-wide column tiles; tiles per slice
  int S, kper;          // wgrad: the batch is cut into S slices of kper rows
  int pitch;            // wgrad: row pitch of C in floats (d when S == 1)
  int end;              // exclusive end of this job's block range
  int mod;              // modality
  uint32_t magic_a, magic_b;   // ceil(2^32 / x) for x = tiles (wgrad: slice of a block), tiles_n (dgrad) / tiles_m (wgrad)
};
// floor(x / n) for x, n < 2^16 with magic = ceil(2^32 / n); n == 1 (whose magic does not fit 32 bits) is encoded as 0
__device__ __forceinline__ int div_magic(int x, uint32_t magic) { return magic ? (int)__umulhi((uint32_t)x, magic) : x; }
static inline uint32_t make_magic(int n) { return n <= 1 ? 0u : (uint32_t)(((1ull << 32) + (uint64_t)n - 1) / (uint64_t)n); }

// keep-mask of one dword (two bf16) from two code bytes of `cw` (sel = v_perm selector placing them in the high bytes);
// sh = 0 for modality 1 (KEEP1 = bit 7 is the sign bit of the 16-bit lane), 1 for modality 0 (KEEP0 = bit 6)
__device__ __forceinline__ uint32_t mask_pair(uint32_t data, uint32_t cw, uint32_t sel, s16x2 sh) {
  const uint32_t m = __builtin_amdgcn_perm(0u, cw, sel);        // [code_hi, 0, code_lo, 0]
  s16x2 s = __builtin_bit_cast(s16x2, m);
  s = (s << sh) >> 15;                                          // 0xFFFF where the keep bit is set
  return data & __builtin_bit_cast(uint32_t, s);
}
__device__ __forceinline__ bf16x8 mask_frag(bf16x8 v, uint32_t c_lo, uint32_t c_hi, s16x2 sh) {
  u32x4 d = __builtin_bit_cast(u32x4, v);
  d[0] = mask_pair(d[0], c_lo, 0x010c000cu, sh);
  d[1] = mask_pair(d[1], c_lo, 0x030c020cu, sh);
  d[2] = mask_pair(d[2], c_hi, 0x010c000cu, sh);
  d[3] = mask_pair(d[3], c_hi, 0x030c020cu, sh);
  return __builtin_bit_cast(bf16x8, d);
}

// per-lane address parts of the transposing fragment reads of a K-major chunk image (rows = k, 128 bytes = 64 columns):
// column tile ni (16 columns), k-step h (32 rows), second half (rows +4): kb + off[ni] + h * 4096 + second * 512
struct KmLane {
  uint32_t kb;                                    // row and 8-byte half part
  int ps, sw;                                     // p >> 1 and the slot permutation of this lane's rows
};
__device__ __forceinline__ KmLane km_lane(int lane) {
  const int g = lane >> 4, w = lane & 15, q = w >> 2, p = w & 3;
  KmLane k;
  k.kb = (uint32_t)((8 * g + q) * 128 + (p & 1) * 8);
  k.ps = p >> 1;
  k.sw = (((q >> 1) & 1) << 1) | ((g & 1) << 2);  // swz16 of rows 8g + q (+4, +32)
  return k;
}
// byte offset of column tile ni inside the image for this lane (computed, not tabulated: ni may be a run-time value and a
// run-time-indexed register array would live in scratch)
__device__ __forceinline__ uint32_t km_off(const KmLane& k, int ni) { return k.kb + (uint32_t)(((2 * ni + k.ps) ^ k.sw) << 4); }
__device__ __forceinline__ bf16x8 km_frag_at(uint32_t a) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)a);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)(a + 512u));
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = lo;
  u.s.hi = hi;
  return u.v;
}

constexpr int kBwdLds = 80 * 1024;                // one round: operand images + code bytes of 256 reduction indices

// ------------------------------------------------------------------------------------------------ dgrad tile
// LDS of a round: [4 chunks][dE 64 rows x 128 B | code 64 rows x 64 B | W 64 k-rows x 128 B] = 4 x 20 KB; wave w requests
// chunk w (k range [256 round + 64 w, +64)).
struct DgradStep {                                // fragments of one k-step (32 k): 10 LDS reads
  bf16x8 a[4], braw;
  u32x2 cw;
};
__device__ __forceinline__ void dgrad_tile(const __bf16* __restrict__ dE, const uint8_t* __restrict__ code, int B, int c,
                                           const SplitJob& job, int tile, char* smem) {
  constexpr int A_BYTES = 64 * 128, C_BYTES = 64 * 64, STAGE = A_BYTES + C_BYTES + 64 * 128;
  EMB_STAMP(2);
  EMB_STAMP_KIND(2 + job.mod);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int tm = div_magic(tile, job.magic_b), tn = tile - tm * job.tiles_n;
  const int row0 = tm * 64, n0 = tn * 64;
  const int d = job.d;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const uint32_t mine = lds0 + (uint32_t)(wave * STAGE);
  const char* dEo = reinterpret_cast<const char*>(dE) + (long)row0 * c * 2;       // rows >= B read zeros (range check)
  const char* cdo = reinterpret_cast<const char*>(code) + (long)row0 * c;
  const char* Wo = reinterpret_cast<const char*>(job.Bptr);
  const long de_bytes = (long)(B - row0) * c * 2, cd_bytes = (long)(B - row0) * c, w_bytes = (long)c * d * 2;
  DmaImage<64, 8> de, dw;
  DmaImage<64, 16> dc;
  de.init((uint32_t)c * 2, 0, 128, lane);
  dc.init((uint32_t)c, 0, 64, lane);
  dw.init((uint32_t)d * 2, n0 * 2, d * 2, lane);                 // columns >= d read zeros
  auto request = [&](int rd) {                                   // this wave's chunk of round rd: k range [k0, k0 + 64)
    const int k0 = rd * 256 + wave * 64;
    if (k0 + 64 <= c) {
      de.issue(dEo + k0 * 2, dma_nrec(de_bytes - k0 * 2), mine);
      dc.issue(cdo + k0, dma_nrec(cd_bytes - k0), mine + A_BYTES);
    } else {                                                     // k beyond c reads zeros (not the next row)
      de.issue_tail(dEo + k0 * 2, dma_nrec(de_bytes - k0 * 2), (c - k0) * 2, mine);
      dc.issue_tail(cdo + k0, dma_nrec(cd_bytes - k0), c - k0, mine + A_BYTES);
    }
    dw.issue(Wo + (long)k0 * d * 2, dma_nrec(w_bytes - (long)k0 * d * 2), mine + A_BYTES + C_BYTES);
  };
  request(0);                                                    // in flight while the fragment addresses are formed
  EMB_STAMP(3);

  const RmLane rl = rm_lane(lane);
  const KmLane kl = km_lane(lane);
  uint32_t woff[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni) woff[ni] = km_off(kl, ni);
  const int r = lane & 15, g = lane >> 4;
  uint32_t coff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) coff[h] = (uint32_t)(r * 64 + (((2 * h + (g >> 1)) ^ ((r >> 2) & 3)) << 4) + (g & 1) * 8);
  typedef __attribute__((address_space(3))) u32x2 lds_u2;
  const short shv = job.mod ? 0 : 1;
  const s16x2 sh = {shv, shv};
  auto load = [&](int step, DgradStep& f) {                      // step = 2 * chunk + h
    const uint32_t st = lds0 + (uint32_t)((step >> 1) * STAGE);
    const int h = step & 1;
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) f.a[ni] = km_frag_at(st + A_BYTES + C_BYTES + woff[ni] + (uint32_t)(h * 4096));
    f.braw = lds_read16<__bf16>(st + wave * 2048 + rl.off[h]);
    f.cw = *(const lds_u2*)(uintptr_t)(st + A_BYTES + wave * 1024 + coff[h]);
  };

  f32x4 acc[4];
#pragma unroll
  for (int ni = 0; ni < 4; ++ni)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[ni][q] = 0.0f;

  const int rounds = (c + 255) / 256;
  for (int rd = 0; rd < rounds; ++rd) {
    if (rd > 0) {
      __syncthreads();                                           // every wave has read the previous round's images
      request(rd);
    }
    EMB_WAIT_VMCNT(0);
    __syncthreads();
    if (rd == 0) EMB_STAMP(4);
    const int nsteps = min(8, (c - rd * 256 + 31) / 32);         // k-steps of this round that hold data
    // software pipeline over the k-steps: the reads of step s + 1 are issued before step s is multiplied
    DgradStep f[2];
    load(0, f[0]);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nsteps) {
        if (s + 1 < nsteps) load(s + 1, f[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8 b = mask_frag(f[s & 1].braw, f[s & 1].cw[0], f[s & 1].cw[1], sh);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s & 1].a[ni], b, acc[ni], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  EMB_STAMP(5);
  // C[col][row]: lane & 15 = row of this wave's 16, registers = 4 consecutive columns
  __bf16* out = reinterpret_cast<__bf16*>(job.C);
  const int row = row0 + wave * 16 + r;
  if (row < B) {
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
      const int n = n0 + ni * 16 + 4 * g;
      if (n < d) {                                               // d % 8 == 0: four columns are inside together
        bf16x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (__bf16)acc[ni][q];
        *reinterpret_cast<bf16x4*>(out + (long)row * d + n) = o;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ wgrad tile
// LDS of a round: [8 chunks][dE 32 rows x 128 B | code 32 rows x 64 B | X 32 rows x 128 B] = 8 x 10 KB; wave w requests
// chunks 2w, 2w + 1 (batch rows [k_begin + 256 round + 32 chunk, +32)).
struct WgradStep {                                // fragments of one chunk (32 batch rows): 10 LDS reads
  bf16x8 a[2], braw[2];
  i32x2 cw[2];
};
__device__ __forceinline__ void wgrad_tile(const __bf16* __restrict__ dE, const uint8_t* __restrict__ code, int B, int c,
                                           const SplitJob& job, int tile, int slice, char* smem) {
  constexpr int KB = 32;
  constexpr int A_BYTES = KB * 128, C_BYTES = KB * 64, STAGE = A_BYTES + C_BYTES + KB * 128;
  EMB_STAMP(2);
  EMB_STAMP_KIND(job.mod);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_m = (c + 63) >> 6;
  const int tn = div_magic(tile, job.magic_b), tmc = tile - tn * tiles_m;   // c tile fastest: tiles sharing an X panel are neighbours
  const int c0 = tmc * 64, n0 = tn * 64;
  const int k_begin = slice * job.kper, k_end = min(B, k_begin + job.kper);
  const int d = job.d;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const char* dEo = reinterpret_cast<const char*>(dE);
  const char* cdo = reinterpret_cast<const char*>(code);
  const char* Xo = reinterpret_cast<const char*>(job.Bptr);
  DmaImage<KB, 8> de, dx;
  DmaImage<KB, 16> dc;
  de.init((uint32_t)c * 2, c0 * 2, c * 2, lane);
  dc.init((uint32_t)c, c0, c, lane);
  dx.init((uint32_t)d * 2, n0 * 2, d * 2, lane);
  auto request = [&](int rd) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int ch = 2 * wave + i;
      const long r0 = k_begin + rd * 256 + ch * KB;              // rows >= k_end read zeros (range check on the slice end)
      const uint32_t st = lds0 + (uint32_t)(ch * STAGE);
      de.issue(dEo + r0 * c * 2, dma_nrec(((long)k_end - r0) * c * 2), st);
      dc.issue(cdo + r0 * c, dma_nrec(((long)k_end - r0) * c), st + A_BYTES);
      dx.issue(Xo + r0 * d * 2, dma_nrec(((long)k_end - r0) * d * 2), st + A_BYTES + C_BYTES);
    }
  };
  request(0);
  EMB_STAMP(3);
  const bool with_bias = (n0 == 0) && (wn == 0);

  const KmLane kl = km_lane(lane);
  uint32_t doff[2], xoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    doff[i] = km_off(kl, 2 * wm + i);
    xoff[i] = km_off(kl, 2 * wn + i);
  }
  // 8-bit transposing read: within a 16-lane group lane 2q'+p' supplies the address of row q', columns 8p' .. 8p'+7 of an
  // 8-row x 16-column byte block; lane i of the group receives column i, rows 0..7 (= the k order of the bf16 fragment)
  const int g = lane >> 4, w = lane & 15;
  uint32_t ctr[2];
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int row = 8 * g + (w >> 1);
    ctr[ci] = (uint32_t)(row * 64 + (((2 * wm + ci) ^ ((row >> 2) & 3)) << 4) + (w & 1) * 8);
  }
  typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;
  const short shv = job.mod ? 0 : 1;
  const s16x2 sh = {shv, shv};
  auto load = [&](int ch, WgradStep& f) {
    const uint32_t st = lds0 + (uint32_t)(ch * STAGE);
#pragma unroll
    for (int ci = 0; ci < 2; ++ci) {
      f.braw[ci] = km_frag_at(st + doff[ci]);
      f.cw[ci] = __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(uintptr_t)(st + A_BYTES + ctr[ci]));
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) f.a[ni] = km_frag_at(st + A_BYTES + C_BYTES + xoff[ni]);
  };

  f32x4 acc[2][2], accb[2];                                      // [c tile][column tile]
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
#pragma unroll
    for (int q = 0; q < 4; ++q) accb[ci][q] = 0.0f;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[ci][ni][q] = 0.0f;
  }

  const int rounds = (k_end - k_begin + 255) / 256;
  for (int rd = 0; rd < rounds; ++rd) {
    if (rd > 0) {
      __syncthreads();
      request(rd);
    }
    EMB_WAIT_VMCNT(0);
    __syncthreads();
    if (rd == 0) EMB_STAMP(4);
    const int nsteps = min(8, (k_end - k_begin - rd * 256 + KB - 1) / KB);   // chunks of this round that hold rows
    WgradStep f[2];
    load(0, f[0]);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      if (s < nsteps) {
        if (s + 1 < nsteps) load(s + 1, f[(s + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 b[2];
#pragma unroll
        for (int ci = 0; ci < 2; ++ci) b[ci] = mask_frag(f[s & 1].braw[ci], (uint32_t)f[s & 1].cw[ci][0], (uint32_t)f[s & 1].cw[ci][1], sh);
#pragma unroll
        for (int ci = 0; ci < 2; ++ci)
#pragma unroll
          for (int ni = 0; ni < 2; ++ni) acc[ci][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f[s & 1].a[ni], b[ci], acc[ci][ni], 0, 0, 0);
        if (with_bias) {
#pragma unroll
          for (int ci = 0; ci < 2; ++ci) accb[ci] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, b[ci], accb[ci], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  EMB_STAMP(5);
  // C[col][c]: lane & 15 = c row inside the 16-row tile, registers = 4 consecutive columns
  float* out = reinterpret_cast<float*>(job.C) + (job.S > 1 ? (long)slice * c * job.pitch : 0);
  const int pitch = job.pitch;
#pragma unroll
  for (int ci = 0; ci < 2; ++ci) {
    const int crow = c0 + (2 * wm + ci) * 16 + w;
    if (crow < c) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        const int n = n0 + (2 * wn + ni) * 16 + 4 * g;
        if (n < d) *reinterpret_cast<f32x4*>(out + (long)crow * pitch + n) = acc[ci][ni];
      }
      if (with_bias && g == 0) {                                 // every register of accb holds the column sum
        if (job.S > 1) out[(long)crow * pitch + d] = accb[ci][0];
        else job.bias[crow] = accb[ci][0];
      }
    }
  }
}

// field-wise selection of one of the four by-value job descriptors (all kernel arguments are fetched in one go; a run-time
// INDEX into a by-value argument is what hipcc 7.2 miscompiles, see embrace_bwd.hip)
__device__ __forceinline__ SplitJob pick_job(int k, const SplitJob& a, const SplitJob& b, const SplitJob& c, const SplitJob& d) {
  SplitJob j;
#define EMB_PICK(f) j.f = k == 0 ? a.f : (k == 1 ? b.f : (k == 2 ? c.f : d.f))
  EMB_PICK(Bptr); EMB_PICK(C); EMB_PICK(bias); EMB_PICK(d); EMB_PICK(tiles_n); EMB_PICK(tiles); EMB_PICK(S); EMB_PICK(kper);
  EMB_PICK(pitch); EMB_PICK(end); EMB_PICK(mod); EMB_PICK(magic_a); EMB_PICK(magic_b);
#undef EMB_PICK
  return j;
}

__global__ __launch_bounds__(kThreads, 2) void embrace_bwd_split_kernel(const __bf16* __restrict__ dE, const uint8_t* __restrict__ code,
                                                                      int B, int c, const SplitJob wg1, const SplitJob dg1,
                                                                      const SplitJob wg0, const SplitJob dg0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int bid = blockIdx.x;
  const int kind = (bid >= wg1.end) + (bid >= dg1.end) + (bid >= wg0.end);     // 0 wgrad1, 1 dgrad1, 2 wgrad0, 3 dgrad0
  const SplitJob job = pick_job(kind, wg1, dg1, wg0, dg0);
  const int first = kind == 0 ? 0 : (kind == 1 ? wg1.end : (kind == 2 ? dg1.end : wg0.end));
  const int q = bid - first;
  if (kind & 1) {
    dgrad_tile(dE, code, B, c, job, kind == 1 ? xcd_remap(q, job.tiles) : q, smem);
  } else {
    const int slice = div_magic(q, job.magic_a), t = q - slice * job.tiles;
    wgrad_tile(dE, code, B, c, job, kind == 0 ? xcd_remap(t, job.tiles) : t, slice, smem);
  }
  EMB_STAMP(8);
}

// returns 1 when the shapes do not qualify (caller uses the tiled kernel of embrace_bwd.hip)
static int bwd_split_dispatch(const void* dE, const uint8_t* code, const void* X0, const void* X1, const void* W0, const void* W1,
                              void* dX0, void* dX1, void* dW0, void* db0, void* dW1, void* db1, void* ws, int64_t ws_bytes, int B,
                              int d0, int d1, int c, int force_S, hipStream_t s) {
  if (c % 16 || d0 % 8 || d1 % 8) return 1;
  const void* ptrs[] = {dE, X0, X1, W0, W1, dX0, dX1, dW0, dW1};
  for (const void* p : ptrs)
    if (p != nullptr && !aligned16(p)) return 1;
  if ((reinterpret_cast<uintptr_t>(code) & 15u) != 0) return 1;
  const long big = (long)B * (d1 > c ? d1 : c) * 2;
  if (big >= (1l << 31) || (long)c * d1 * 2 >= (1l << 31)) return 1;     // 32-bit buffer offsets (split_core.h)
  if ((long)cdiv(B, 64) * cdiv(d1, 64) >= 65536 || (long)cdiv(c, 64) * cdiv(d1, 64) * 16 >= 65536) return 1;   // div_magic range
  int n = 0;
  int64_t ws_used = 0;
  struct SlabInfo { float* slab; int pitch; int S; } slabs[2] = {{nullptr, 0, 1}, {nullptr, 0, 1}};
  auto wgrad = [&](const void* X, void* dW, void* db, int d, int m) {
    SplitJob j{};
    j.Bptr = (const __bf16*)X; j.C = dW; j.bias = (float*)db; j.d = d;
    j.tiles_n = cdiv(d, 64);
    j.tiles = cdiv(c, 64) * j.tiles_n;
    j.S = 1; j.kper = B; j.pitch = d;
    // a workgroup takes 256 batch rows per round; slice the batch so that a tile needs at most two rounds
    int S = force_S > 0 ? force_S : cdiv(B, 512);
    if (S > 16) S = 16;
    const int pitch = cdiv(d + 1, 4) * 4;
    const int64_t per = (int64_t)c * pitch * 4;
    if (ws == nullptr) S = 1;
    else if ((int64_t)S * per > ws_bytes - ws_used) S = (int)((ws_bytes - ws_used) / per);
    if (S > 1) {
      j.kper = cdiv(cdiv(B, S), 32) * 32;
      j.S = cdiv(B, j.kper);
      if (j.S > 1) {
        j.C = (char*)ws + ws_used;
        j.pitch = pitch;
        ws_used += (int64_t)j.S * per;
        slabs[m] = SlabInfo{(float*)j.C, pitch, j.S};
      } else {
        j.kper = B;
      }
    }
    n += j.tiles * j.S;
    j.end = n;
    j.mod = m;
    j.magic_a = make_magic(j.tiles);
    j.magic_b = make_magic(cdiv(c, 64));
    return j;
  };
  auto dgrad = [&](const void* W, void* dX, int d, int m) {
    SplitJob j{};
    j.Bptr = (const __bf16*)W; j.C = dX; j.d = d;
    j.tiles_n = cdiv(d, 64);
    j.tiles = cdiv(B, 64) * j.tiles_n;
    if (dX != nullptr) n += j.tiles;
    j.end = n;
    j.mod = m;
    j.magic_a = make_magic(j.tiles);
    j.magic_b = make_magic(j.tiles_n);
    return j;
  };
  const SplitJob wg1 = wgrad(X1, dW1, db1, d1, 1);
  const SplitJob dg1 = dgrad(W1, dX1, d1, 1);
  const SplitJob wg0 = wgrad(X0, dW0, db0, d0, 0);
  const SplitJob dg0 = dgrad(W0, dX0, d0, 0);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embrace_bwd_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kBwdLds);
    attr_set = true;
  }
  embrace_bwd_split_kernel<<<n, kThreads, kBwdLds, s>>>((const __bf16*)dE, code, B, c, wg1, dg1, wg0, dg0);
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
