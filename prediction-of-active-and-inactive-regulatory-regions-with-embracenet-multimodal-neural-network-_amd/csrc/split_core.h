// Building blocks of the "K split over waves" kernels (embrace_split.h: fused forward; embrace_bwd2.hip: backward).
//
// Shape of these kernels: a workgroup owns one output tile; its 4 waves split the REDUCTION range in 128-byte-wide
// chunks (wave w takes chunks w, w+4, ...), every wave streams its own chunks global -> LDS by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, whole 128-byte lines, up to NSTAGE chunks in flight per wave) into a private
// ring, reads MFMA fragments back and accumulates a full partial tile.  There is NO workgroup barrier in the main loop
// (a wave's LDS-DMA data is ordered for its own reads by its own counted vmcnt); the four partial tiles meet in LDS once.
//
// Chunk image: ROWS rows x 128 bytes, the eight 16-byte slots of a row XOR-permuted by swz16(row).  The permutation is
// applied on the per-lane SOURCE address (the LDS-DMA destination is lane-linear) and again in the fragment reads.
// With swz16 both kinds of fragment read are bank-conflict free: ds_read_b128 of a row-major operand (rows = tile rows,
// bytes = k) and ds_read_b64_tr_b16 of a K-major operand (rows = k, bytes = tile columns).
#pragma once
#include "gemm_core.h"

namespace emb {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_cvoid_t;

__device__ __forceinline__ int swz16(int row) { return (((row >> 1) & 1) << 1) | (((row >> 3) & 1) << 2); }

// 16 zero bytes: what an LDS-DMA lane outside the matrix reads
__device__ __attribute__((aligned(16))) const unsigned int g_zero16[4] = {0u, 0u, 0u, 0u};

// Issue the LDS-DMA of one chunk image: rows row0 .. row0+ROWS-1 (rows >= nrows read zeros) of a row-major matrix with
// `ld_bytes` per row, bytes [colb0, colb0+128) of each row (bytes >= rowbytes read zeros; rowbytes % 16 == 0).
// ROWS/8 wave-instructions of 1 KiB.  `lds` = wave-uniform LDS byte address of the image.
template <int ROWS>
__device__ __forceinline__ void dma_chunk(const char* __restrict__ g, long ld_bytes, int row0, int nrows, int colb0,
                                          int rowbytes, uint32_t lds, int lane) {
  const int rl = lane >> 3, ps = lane & 7;
#pragma unroll
  for (int j = 0; j < ROWS / 8; ++j) {
    const int row = 8 * j + rl;
    const int cb = colb0 + 16 * (ps ^ swz16(row));
    const bool ok = (row0 + row < nrows) && (cb < rowbytes);
    const char* src = ok ? g + (long)(row0 + row) * ld_bytes + cb : reinterpret_cast<const char*>(g_zero16);
    __builtin_amdgcn_global_load_lds((gbl_cvoid_t*)src, (lds_void_t*)(uintptr_t)(lds + j * 1024), 16, 0, 0);
  }
}

// per-lane byte offsets of the two fragment reads (k bytes [0,64) and [64,128) of the chunk) of a ROW-MAJOR operand
// inside a 16-row tile: lane (r = lane & 15, g = lane >> 4) reads slot 4h + g of row r
struct RmLane {
  uint32_t off[2];
};
__device__ __forceinline__ RmLane rm_lane(int lane) {
  const int r = lane & 15, g = lane >> 4, s = swz16(r);
  return RmLane{{(uint32_t)(r * 128 + ((g ^ s) << 4)), (uint32_t)(r * 128 + (((4 + g) ^ s) << 4))}};
}
template <typename T> __device__ __forceinline__ typename Vec16<T>::type lds_read16(uint32_t addr) {
  using V = typename Vec16<T>::type;
  typedef __attribute__((address_space(3))) V lds_V;
  return *(const lds_V*)(uintptr_t)addr;
}

// one 16-byte fragment = STEPS MFMA steps (bf16: one 16x16x32; f32: four 16x16x4; f64: two 16x16x4); for f32 / f64 the
// k values are visited in a permuted order that is the same for both operands
template <typename T> struct FragSteps;
template <> struct FragSteps<__bf16> {
  static constexpr int STEPS = 1;
  __device__ static bf16x8 get(const bf16x8& v, int) { return v; }
};
template <> struct FragSteps<float> {
  static constexpr int STEPS = 4;
  __device__ static float get(const f32x4& v, int j) { return v[j]; }
};
template <> struct FragSteps<double> {
  static constexpr int STEPS = 2;
  __device__ static double get(const f64x2& v, int j) { return v[j]; }
};

template <typename T, int MI, int NI>
__device__ __forceinline__ void mma_frags(const typename Vec16<T>::type (&a)[MI], const typename Vec16<T>::type (&b)[NI],
                                          typename Mma<T>::AccV (&acc)[MI][NI]) {
#pragma unroll
  for (int st = 0; st < FragSteps<T>::STEPS; ++st)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        acc[mi][ni] = Mma<T>::mma(FragSteps<T>::get(a[mi], st), FragSteps<T>::get(b[ni], st), acc[mi][ni]);
}


// The same images through BUFFER loads (buffer_load_dwordx4 ... offen lds): the matrix is described by a buffer resource in
// SGPRs, every lane keeps ONE 32-bit byte offset per instruction, formed once per tile, and a chunk is addressed by moving
// the resource's base (scalar arithmetic) -- no vector instruction per LDS-DMA instruction, and the hardware range check
// (offset >= num_records reads zeros) covers the rows past the end of the matrix or of a batch slice.  Regions must be
// smaller than 2 GiB (kDmaInvalid marks lanes that always read zeros).
// RPI = rows per LDS-DMA instruction: 8 for 128-byte image rows (slot permutation swz16), 16 for the 64-byte rows of the
// code images (slot permutation (row >> 2) & 3).
constexpr uint32_t kDmaInvalid = 0x80000000u;
template <int RPI> __device__ __forceinline__ int dma_slot(int j, int lane) {
  if (RPI == 8) return (lane & 7) ^ swz16(8 * j + (lane >> 3));
  return (lane & 3) ^ (((lane >> 2) >> 2) & 3);
}
__device__ __forceinline__ uint32_t dma_nrec(long bytes) { return bytes > 0 ? (uint32_t)bytes : 0u; }

template <int ROWS, int RPI = 8, int JS = 1> struct DmaImage {
  static constexpr int NI = ROWS / RPI;           // LDS-DMA instructions of the whole image
  static constexpr int NJ = NI / JS;              // ... of which this wave issues j = j0, j0 + JS, ... (JS = 1: all)
  static_assert(NI % JS == 0 && (JS == 1 || JS % 2 == 0), "instruction split");
  uint32_t voff[NJ];                              // byte offset of this lane's 16 bytes of its i-th instruction from the chunk origin
  int slot16;                                     // byte offset of the lane's slot inside the window (JS > 1: one parity only)
  int slot16_odd;                                 // JS == 1: the slot of odd instructions
  int j0;
  // the window starts at byte `win0` of a row; slots starting at or beyond byte `rowbytes` of the row read zeros
  __device__ __forceinline__ void init(uint32_t ld_bytes, int win0, int rowbytes, int lane, int j0_ = 0) {
    const int rl = lane / (64 / RPI);
    j0 = j0_;
    slot16 = 16 * dma_slot<RPI>(j0, lane);
    slot16_odd = 16 * dma_slot<RPI>(j0 + 1, lane);
#pragma unroll
    for (int i = 0; i < NJ; ++i) {
      const int j = j0 + i * JS;
      const int cb = win0 + ((JS == 1 && (i & 1)) ? slot16_odd : slot16);
      voff[i] = cb < rowbytes ? (uint32_t)(RPI * j + rl) * ld_bytes + (uint32_t)cb : kDmaInvalid;
    }
  }
  // origin: wave-uniform address of (first image row, byte 0 of the window base); nrec: valid bytes from there on
  __device__ __forceinline__ void issue(const char* origin, uint32_t nrec, uint32_t lds) const {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of a kernel TEMPLATE must not see the buffer builtins: it would drop the stub)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, nrec, 0x00020000);
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(uintptr_t)(lds + (j0 + i * JS) * 1024), 16, voff[i], 0, 0, 0);
#endif
  }
  // the same for a window whose last `128 - rem` (64 - rem) bytes lie beyond the end of the row: those slots read zeros
  // (reading on would enter the next row)
  __device__ __forceinline__ void issue_tail(const char* origin, uint32_t nrec, int rem, uint32_t lds) const {
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)origin, 0, nrec, 0x00020000);
#pragma unroll
    for (int i = 0; i < NJ; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(uintptr_t)(lds + (j0 + i * JS) * 1024), 16,
                                               ((JS == 1 && (i & 1)) ? slot16_odd : slot16) < rem ? voff[i] : kDmaInvalid, 0, 0, 0);
#endif
  }
};

// ---- diagnostic builds only (-DEMB_SPLIT_PROF, tools/kbench): per-workgroup phase stamps of wave 0 into a debug buffer that
// nothing else reads.  In the library build EMB_STAMP() is empty and no stamp executes.
#ifdef EMB_SPLIT_PROF
__device__ unsigned long long* g_split_prof = nullptr;     // [gridDim][16]: slot 0 job kind, 1 wall clock at entry, 2.. shader clock
#define EMB_STAMP(k)                                                                                   \
  do {                                                                                                 \
    if (g_split_prof != nullptr && threadIdx.x == 0) {                                                 \
      if ((k) == 2) g_split_prof[(size_t)blockIdx.x * 16 + 1] = __builtin_amdgcn_s_memrealtime();      \
      if ((k) == 8) g_split_prof[(size_t)blockIdx.x * 16 + 9] = __builtin_amdgcn_s_memrealtime();      \
      g_split_prof[(size_t)blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memtime();                      \
    }                                                                                                  \
  } while (0)
#define EMB_STAMP_KIND(v)                                                                              \
  do {                                                                                                 \
    if (g_split_prof != nullptr && threadIdx.x == 0) g_split_prof[(size_t)blockIdx.x * 16] = (v);      \
  } while (0)
#else
#define EMB_STAMP(k) do { } while (0)
#define EMB_STAMP_KIND(v) do { } while (0)
#endif

#define EMB_WAIT_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

// wait until at most `chunks` (wave-uniform, 0 .. 3) groups of G vector-memory operations of this wave are outstanding
template <int G> __device__ __forceinline__ void wait_chunks_in_flight(int chunks) {
  static_assert(3 * G <= 63, "vmcnt range");
  if (chunks >= 3) EMB_WAIT_VMCNT(3 * G);
  else if (chunks == 2) EMB_WAIT_VMCNT(2 * G);
  else if (chunks == 1) EMB_WAIT_VMCNT(G);
  else EMB_WAIT_VMCNT(0);
}

}  // namespace emb
