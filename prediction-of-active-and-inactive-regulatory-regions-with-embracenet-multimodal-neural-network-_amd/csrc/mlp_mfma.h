// The small MLP stacks (FFNN_pre.py:18-49, EmbraceNetMultimodal.py:134-154) on the matrix cores, bf16, widths that are
// multiples of 16 (input width: of 8).  Included by mlp.hip; same arguments, stored activations, mask bytes, RNG contract and
// partial-sum layout as the scalar kernels there, which remain the path for every other shape / precision.
//
// The scalar kernels spend ~2 us per layer on LDS dot products (two LDS reads per multiply-add) behind a 2.6-4 us staging
// phase that converts every operand to fp32 in LDS.  Sixteen rows of such a layer are ONE 16x16x32 MFMA tile per 16 outputs, so
// here a single WAVE carries 16 rows through the whole stack:
//   forward   h = relu(x W^T + b): A = the rows (row-major fragments: straight from global memory for the first layer, from
//             a 16-row LDS tile afterwards), B = W as fragment-ready 1 KiB LDS blocks fetched by LDS-DMA (the lane permutation is
//             in the source offsets); the epilogue writes the bf16 tile the next layer reads and the stored copies.
//   backward  per layer dW = dz^T hin and dprev = dz W: the contraction index of dW is the 16 rows, of dprev the layer's outputs;
//             both K-major operands are read with the transposing LDS read (ds_read_b64_tr_b16) from row-major images that
//             arrive by LDS-DMA as plain linear copies (the rows of a workgroup are contiguous in every tensor); rows 16..31 of
//             the images are zeros so that a full 32-deep MFMA step can be issued.
// A workgroup is one wave: no barriers anywhere; 64 workgroups at B = 1024.
#pragma once
#include "common.h"
#include "mlp_args.h"
#include "philox.h"
#include "split_core.h"

namespace emb {

constexpr int kMmMaxKS = 8;                 // k-steps of 32: widths <= 256

struct MmFwdLayout {                        // LDS byte offsets (host computed)
  int wblk[kMlpMaxL];                       // fragment-ready weight blocks of layer l: block (nt, ks) at (nt * nks + ks) * 1024
  int bias[kMlpMaxL];                       // fp32 bias vector of layer l
  int act[2];                               // ping-pong [16][AP] bf16 activation tiles
  int msk;                                  // [16][maxN] mask bytes of the current layer
  int AP;                                   // tile pitch in elements (multiple of 8)
};

struct MmBwdLayout {
  int wimg[kMlpMaxL];                       // W_l as stored, [N_l rounded up to 32][K_l] bf16 (pad rows zero)
  int hin[kMlpMaxL];                        // layer input rows [32][K_l] bf16 (rows 16..31 zero)
  int msk[kMlpMaxL];                        // [16][N_l] mask bytes
  int dz[2];                                // ping-pong [32][NP] bf16 gradient tiles (rows 16..31 zero)
  int dy;                                   // [16][N_last] bf16 incoming gradient
  int NP;                                   // gradient tile pitch in elements
  int zero_begin, zero_end;                 // byte range cleared at start (covers every pad region)
};

// NOTE on the layer loops below: they are written with the constant trip count kMlpMaxL, fully unrolled, and skip layers >= L.
// Indexing a by-value kernel argument array with a run-time layer index would make the compiler keep the whole argument block in
// scratch memory (select chains are folded back into an indexed load); after full unrolling every subscript is a constant.

__device__ __forceinline__ bf16x8 mm_zero8() {
  bf16x8 z;
#pragma unroll
  for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
  return z;
}
__device__ __forceinline__ bf16x8 mm_tr16(uint32_t lo_addr, uint32_t hi_addr) {
  typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
  union { struct { s16x4 lo, hi; } s; bf16x8 v; } u;
  u.s.lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)lo_addr);
  u.s.hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(uintptr_t)hi_addr);
  return u.v;
}
// linear copy of `bytes` (multiple of 16, < 2 GiB of source left) global -> LDS by LDS-DMA; bytes past `valid` read zeros
__device__ __forceinline__ void mm_dma_copy(const void* src, long valid, int bytes, uint32_t lds, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(valid < 0 ? 0 : (valid < 0x7fffffffL ? valid : 0x7fffffffL)), 0x00020000);
  for (int o = 0; o < bytes; o += 1024) {
    const int off = o + lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(uintptr_t)(lds + (uint32_t)o), 16, off < bytes ? (uint32_t)off : kDmaInvalid, 0, 0, 0);
  }
#endif
}

// one wave, 16 rows starting at 16 * blk; `smraw` = the wave's LDS (layout `lay`).  No barriers: callable from any kernel whose
// other waves do something else (rider.h).
__device__ __forceinline__ void mlp_fwd_mfma_body(const MlpArgs<__bf16>& a, const MmFwdLayout& lay, int blk, char* smraw) {
  using T = __bf16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smraw;
  const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15;
  const int rb = blk * 16, L = a.L, AP = lay.AP;

#if defined(__HIP_DEVICE_COMPILE__)
  {   // every layer's weights as fragment-ready blocks: lane (n = r16, k group g) of block (nt, ks) holds W[16 nt + n][32 ks + 8 g ..]
    int K = a.F;
#pragma unroll
    for (int l = 0; l < kMlpMaxL; ++l) {
      if (l >= L) continue;
      const int N = a.N[l], nks = (K + 31) >> 5, NT = N >> 4;
      const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc((void*)a.W[l], 0, N * K * 2, 0x00020000);
      for (int nt = 0; nt < NT; ++nt)
        for (int ks = 0; ks < nks; ++ks) {
          const int k0 = 32 * ks + 8 * g;
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (lds_void_t*)(uintptr_t)(lds0 + lay.wblk[l] + (nt * nks + ks) * 1024), 16,
                                                   k0 < K ? (uint32_t)(((nt * 16 + r16) * K + k0) * 2) : kDmaInvalid, 0, 0, 0);
        }
      K = N;
    }
  }
#endif
#pragma unroll
  for (int l = 0; l < kMlpMaxL; ++l) {
    if (l >= L) continue;
    float* bl = reinterpret_cast<float*>(smraw + lay.bias[l]);
    for (int n = lane; n < a.N[l]; n += 64) bl[n] = a.b[l][n];
  }
  bf16x8 A[kMmMaxKS];
  {
    const int row = rb + r16;
#pragma unroll
    for (int ks = 0; ks < kMmMaxKS; ++ks) {
      const int k0 = 32 * ks + 8 * g;
      A[ks] = (row < a.B && k0 < a.F) ? *reinterpret_cast<const bf16x8*>(a.x + (long)row * a.F + k0) : mm_zero8();
    }
  }
  const uint64_t step = a.step_val + (a.step_dev ? *a.step_dev : 0);
  EMB_WAIT_VMCNT(0);   // the LDS-DMA blocks have landed (the compiler does not order LDS reads behind LDS-DMA by itself)

  int K = a.F;
#pragma unroll
  for (int l = 0; l < kMlpMaxL; ++l) {
    if (l >= L) continue;
    const int N = a.N[l], nks = (K + 31) >> 5, NT = N >> 4;
    const uint32_t wb = lds0 + lay.wblk[l] + (uint32_t)lane * 16u;
    const float* bl = reinterpret_cast<const float*>(smraw + lay.bias[l]);
    T* out = reinterpret_cast<T*>(smraw + lay.act[(l) & 1]);
    uint8_t* mk = reinterpret_cast<uint8_t*>(smraw + lay.msk);
    const float p = a.drop[l], keep_scale = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
    const uint64_t stream = rng_stream(step, EMB_RNG_DROPOUT0 + a.layer_id[l]);
    const bool relu = a.relu[l] != 0;
    for (int nt = 0; nt < NT; ++nt) {
      f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int ks = 0; ks < kMmMaxKS; ++ks)
        if (ks < nks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks], lds_read16<T>(wb + (uint32_t)((nt * nks + ks) * 1024)), acc, 0, 0, 0);
      const int n = nt * 16 + r16;
      const float bias = bl[n];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tr = 4 * g + r, row = rb + tr;
        float v = acc[r] + bias;
        bool act = true, keep = true;
        if (relu) {
          act = v > 0.0f;
          v = act ? v : 0.0f;
        }
        if (p > 0.f && row < a.B) {
          keep = uniform24(philox4x32_10(a.seed, stream, (uint64_t)(a.row0 + row) * (uint64_t)N + n).x) >= p;
          v = keep ? v * keep_scale : 0.0f;
        }
        out[tr * AP + n] = (T)v;
        mk[tr * N + n] = (uint8_t)((act ? 1 : 0) | (keep ? 2 : 0));
      }
    }
    // stored copies (whole 16-byte vectors, rows of the batch only) and the next layer's A fragments: what the next layer
    // sees is what the stored activation holds
    const int vpr = N >> 3;                                     // vectors per row
    for (int i = lane; i < 16 * vpr; i += 64) {
      const int tr = i / vpr, cv = i - tr * vpr;
      if (rb + tr < a.B) *reinterpret_cast<bf16x8*>(a.h[l] + (long)(rb + tr) * N + cv * 8) = *reinterpret_cast<const bf16x8*>(out + tr * AP + cv * 8);
    }
    if (a.mask[l] != nullptr) {
      const int mpr = N >> 4;
      for (int i = lane; i < 16 * mpr; i += 64) {
        const int tr = i / mpr, cv = i - tr * mpr;
        if (rb + tr < a.B) *reinterpret_cast<uint4*>(a.mask[l] + (long)(rb + tr) * N + cv * 16) = *reinterpret_cast<const uint4*>(mk + tr * N + cv * 16);
      }
    }
#pragma unroll
    for (int ks = 0; ks < kMmMaxKS; ++ks) {
      const int k0 = 32 * ks + 8 * g;
      A[ks] = k0 < N ? *reinterpret_cast<const bf16x8*>(out + r16 * AP + k0) : mm_zero8();
    }
    K = N;
  }
}


__device__ __forceinline__ void mlp_bwd_mfma_body(const MlpBwdArgs<__bf16>& a, const MmBwdLayout& lay, int blk, char* smraw) {
  using T = __bf16;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smraw;
  const int lane = threadIdx.x & 63, g = lane >> 4, r16 = lane & 15, q = r16 >> 2, p4 = r16 & 3;
  const int rb = blk * 16, L = a.L, NP = lay.NP;
  const long rows_left = (long)a.B - rb;                        // > 0

  // pad regions (rows 16..31 of the K-major images, weight rows past N) must hold zeros: clear the whole range first
  for (int o = lay.zero_begin + lane * 16; o < lay.zero_end; o += 1024) *reinterpret_cast<uint4*>(smraw + o) = make_uint4(0u, 0u, 0u, 0u);
  __builtin_amdgcn_s_waitcnt(0xc07f);                           // lgkmcnt(0): the clears precede the LDS-DMA writes below
  {
    int K = a.F;
#pragma unroll
    for (int l = 0; l < kMlpMaxL; ++l) {
      if (l >= L) continue;
      const int N = a.N[l];
      if (l > 0 || a.dx != nullptr) mm_dma_copy(a.W[l], (long)N * K * 2, N * K * 2, lds0 + lay.wimg[l], lane);
      const T* hsrc = l == 0 ? a.x : a.h[l > 0 ? l - 1 : 0];
      mm_dma_copy(hsrc + (long)rb * K, rows_left * K * 2, 16 * K * 2, lds0 + lay.hin[l], lane);
      if (a.mask[l] != nullptr) mm_dma_copy(a.mask[l] + (long)rb * N, rows_left * N, 16 * N, lds0 + lay.msk[l], lane);
      K = N;
    }
    int NL = a.N[0];
#pragma unroll
    for (int l = 1; l < kMlpMaxL; ++l) NL = (l == L - 1) ? a.N[l] : NL;
    mm_dma_copy(a.dy + (long)rb * NL, rows_left * NL * 2, 16 * NL * 2, lds0 + lay.dy, lane);
  }
  EMB_WAIT_VMCNT(0);   // the LDS-DMA copies have landed (the compiler does not order LDS reads behind LDS-DMA by itself)
#pragma unroll
  for (int li = 0; li < kMlpMaxL; ++li) {
    const int l = kMlpMaxL - 1 - li;
    if (l >= L) continue;
    const int N = a.N[l], K = l == 0 ? a.F : a.N[l > 0 ? l - 1 : 0];
    if (l == L - 1) {   // dz of the last layer = dy * mask factor
      const uint8_t need = (uint8_t)((a.relu[l] ? 1 : 0) | (a.drop[l] > 0.f ? 2 : 0));
      const float scale = a.drop[l] > 0.f ? 1.0f / (1.0f - a.drop[l]) : 1.0f;
      const T* dyt = reinterpret_cast<const T*>(smraw + lay.dy);
      const uint8_t* mk = reinterpret_cast<const uint8_t*>(smraw + lay.msk[l]);
      T* dz = reinterpret_cast<T*>(smraw + lay.dz[l & 1]);
      for (int i = lane; i < 16 * N; i += 64) {
        const int tr = i / N, n = i - tr * N;
        const uint8_t m = a.mask[l] != nullptr ? mk[i] : (uint8_t)3;
        const float v = (float)dyt[i];
        dz[tr * NP + n] = (T)(((m & need) == need && rb + tr < a.B) ? v * scale : 0.0f);
      }
    }
    int off = 0;                                                  // this layer's block in the partial vector: dW [N][K], db [N]
#pragma unroll
    for (int j = 0; j < kMlpMaxL; ++j)
      if (j < l) off += a.N[j] * ((j == 0 ? a.F : a.N[j > 0 ? j - 1 : 0]) + 1);
    float* part = a.part + (long)blk * a.total + off;
    const uint32_t dzb = lds0 + lay.dz[(l) & 1];
    const T* dz = reinterpret_cast<const T*>(smraw + lay.dz[(l) & 1]);
    // db[n] = sum of dz over the 16 rows
    for (int n = lane; n < N; n += 64) {
      float s = 0.0f;
#pragma unroll
      for (int tr = 0; tr < 16; ++tr) s += (float)dz[tr * NP + n];
      part[N * K + n] = s;
    }
    // dW[n][k] = sum over rows dz[row][n] * hin[row][k]: both operands K-major (rows 16..31 are zeros)
    const uint32_t hb = lds0 + lay.hin[l];
    const uint32_t a_lo = dzb + (uint32_t)(((8 * g + q) * NP + 4 * p4) * 2), a_hi = a_lo + (uint32_t)(4 * NP * 2);
    const uint32_t b_lo = hb + (uint32_t)(((8 * g + q) * K + 4 * p4) * 2), b_hi = b_lo + (uint32_t)(4 * K * 2);
    const int MT = N >> 4, CT = (K + 15) >> 4;
    for (int mt = 0; mt < MT; ++mt) {
      const bf16x8 af = mm_tr16(a_lo + mt * 32, a_hi + mt * 32);
      for (int ct = 0; ct < CT; ++ct) {
        const bf16x8 bf = mm_tr16(b_lo + ct * 32, b_hi + ct * 32);
        const f32x4 acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf, f32x4{0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
        const int k = ct * 16 + r16;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < 4; ++r) part[(mt * 16 + 4 * g + r) * K + k] = acc[r];
        }
      }
    }
    // dprev[row][k] = sum over n dz[row][n] * W[n][k]
    if (l > 0 || a.dx != nullptr) {
      bf16x8 A[kMmMaxKS];
      const int nks = (N + 31) >> 5;
#pragma unroll
      for (int ks = 0; ks < kMmMaxKS; ++ks) {
        const int n0 = 32 * ks + 8 * g;
        A[ks] = (ks < nks && n0 < N) ? *reinterpret_cast<const bf16x8*>(dz + r16 * NP + n0) : mm_zero8();
      }
      const uint32_t wb = lds0 + lay.wimg[l];
      const uint32_t w_lo = wb + (uint32_t)(((8 * g + q) * K + 4 * p4) * 2), w_hi = w_lo + (uint32_t)(4 * K * 2);
      uint8_t need = 0;
      float scale = 1.0f;
      const uint8_t* mk = nullptr;
      T* dzn = nullptr;
      if (l > 0) {
        need = (uint8_t)((a.relu[l > 0 ? l - 1 : 0] ? 1 : 0) | (a.drop[l > 0 ? l - 1 : 0] > 0.f ? 2 : 0));
        scale = a.drop[l > 0 ? l - 1 : 0] > 0.f ? 1.0f / (1.0f - a.drop[l > 0 ? l - 1 : 0]) : 1.0f;
        mk = a.mask[l > 0 ? l - 1 : 0] != nullptr ? reinterpret_cast<const uint8_t*>(smraw + lay.msk[l > 0 ? l - 1 : 0]) : nullptr;
        dzn = reinterpret_cast<T*>(smraw + lay.dz[(l - 1) & 1]);
      }
      for (int ct = 0; ct < CT; ++ct) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < kMmMaxKS; ++ks)
          if (ks < nks) acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[ks], mm_tr16(w_lo + (uint32_t)(ks * 32 * K * 2) + ct * 32, w_hi + (uint32_t)(ks * 32 * K * 2) + ct * 32), acc, 0, 0, 0);
        const int k = ct * 16 + r16;
        if (k < K) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int tr = 4 * g + r;
            const T dv = (T)acc[r];                              // what the lower layer receives is stored in T
            if (l > 0) {
              const uint8_t m = mk != nullptr ? mk[tr * K + k] : (uint8_t)3;
              dzn[tr * NP + k] = (T)(((m & need) == need) ? (float)dv * scale : 0.0f);
            } else if (rb + tr < a.B) {
              a.dx[(long)(rb + tr) * K + k] = dv;
            }
          }
        }
      }
    }
  }
}


// ---- host side
static bool mlp_mfma_ok(int F, const int* N, int L) {
  if (L < 1 || L > kMlpMaxL || F < 8 || F > 256 || F % 8 != 0) return false;
  for (int l = 0; l < L; ++l)
    if (N[l] < 16 || N[l] > 256 || N[l] % 16 != 0) return false;
  return true;
}

static size_t mlp_mfma_fwd_layout(int F, const int* N, int L, MmFwdLayout* lay) {
  size_t o = 0;
  int K = F, maxN = 0;
  for (int l = 0; l < L; ++l) {
    lay->wblk[l] = (int)o;
    o += (size_t)(N[l] / 16) * ((K + 31) / 32) * 1024;
    K = N[l];
    maxN = N[l] > maxN ? N[l] : maxN;
  }
  for (int l = 0; l < L; ++l) { lay->bias[l] = (int)o; o += (size_t)N[l] * 4; }
  o = (o + 15) & ~(size_t)15;
  lay->AP = maxN + 8;
  for (int i = 0; i < 2; ++i) { lay->act[i] = (int)o; o += (size_t)16 * lay->AP * 2; }
  lay->msk = (int)o;
  o += (size_t)16 * maxN;
  return (o + 15) & ~(size_t)15;
}

static size_t mlp_mfma_bwd_layout(int F, const int* N, int L, MmBwdLayout* lay) {
  size_t o = 0;
  int K = F, maxN = 0;
  // every region that an LDS-DMA copy fills is a multiple of 1 KiB: the lanes past the end of a copy write zeros there
  auto kib = [](size_t b) { return (b + 1023) & ~(size_t)1023; };
  for (int l = 0; l < L; ++l) {
    lay->msk[l] = (int)o;
    o += kib((size_t)16 * N[l]);
    maxN = N[l] > maxN ? N[l] : maxN;
  }
  lay->dy = (int)o;
  o += kib((size_t)16 * N[L - 1] * 2);
  lay->zero_begin = (int)o;
  for (int l = 0; l < L; ++l) {
    lay->wimg[l] = (int)o;
    o += kib((size_t)((N[l] + 31) / 32 * 32 + 8) * K * 2);       // (+8 rows: the transposing reads of a partial last column tile stay inside)
    lay->hin[l] = (int)o;
    o += kib((size_t)(32 + 8) * K * 2);
    K = N[l];
  }
  lay->NP = maxN + 8;
  for (int i = 0; i < 2; ++i) { lay->dz[i] = (int)o; o += kib((size_t)(32 + 8) * lay->NP * 2); }
  lay->zero_end = (int)o;
  return o;
}

}  // namespace emb
