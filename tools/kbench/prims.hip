// Micro-benchmarks of the primitives the fusion kernels are built from (MI355X): dispatch floor of a launch, burst
// streaming rate of one workgroup per CU from L2/MALL-resident data into registers or LDS (LDS-DMA).
//   hipcc --offload-arch=gfx950 -O3 prims.hip -o prims && ./prims
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <functional>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

__global__ void empty_kernel(int* out) { if (out == nullptr && threadIdx.x == 9999) *out = 1; }

// every workgroup reads `bytes` (multiple of 256*16*U) starting at region (blockIdx % regions) * bytes; U loads in flight
template <int U> __global__ __launch_bounds__(256) void read_regs(const u32x4* __restrict__ src, unsigned* out, int bytes, int regions) {
  const u32x4* p = src + (size_t)(blockIdx.x % regions) * (bytes / 16) + threadIdx.x;
  const int iters = bytes / (256 * 16 * U);
  u32x4 acc = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = p[(it * U + u) * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345679u) out[0] = 1;
}

// LDS-DMA: 1 KiB per wave-instruction into a ring of LDS (64 KiB), waits only when the ring wraps
template <int THREADS> __global__ __launch_bounds__(THREADS) void read_lds(const char* __restrict__ src, unsigned* out, int bytes, int regions) {
  extern __shared__ __attribute__((aligned(16))) char sm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = THREADS / 64;
  const char* base = src + (size_t)(blockIdx.x % regions) * bytes;
  const int pieces = bytes / 1024;            // 1 KiB pieces
  constexpr int RING = 64;                    // pieces in the ring (64 KiB)
  for (int p0 = 0; p0 < pieces; p0 += RING) {
    for (int p = p0 + wave; p < p0 + RING && p < pieces; p += nw)
      __builtin_amdgcn_global_load_lds((gbl_void*)(base + (size_t)p * 1024 + lane * 16), (lds_void*)(sm + (p - p0) * 1024), 16, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (sm[threadIdx.x * 4] == 77 && sm[threadIdx.x] == 99) out[0] = 1;
}

static float time_graph(hipStream_t s, int n, const std::function<void()>& launch) {
  for (int i = 0; i < 3; ++i) launch();
  CK(hipStreamSynchronize(s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < n; ++i) launch();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / n;   // us per launch
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const size_t total = 64u << 20;
  char* buf; CK(hipMalloc(&buf, total)); CK(hipMemset(buf, 1, total));
  unsigned* out; CK(hipMalloc(&out, 64));
  const int N = 200;
  for (int g : {256, 512, 1024, 2048}) {
    float us = time_graph(s, N, [&] { empty_kernel<<<g, 256, 0, s>>>((int*)out); });
    printf("empty grid=%d: %.2f us/launch\n", g, us);
  }
  CK(hipFuncSetAttribute((const void*)read_lds<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute((const void*)read_lds<512>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  for (int regions : {8, 64}) {
    for (int grid : {256, 512}) {
      for (int kb : {32, 64, 128, 256, 512}) {
        const int bytes = kb * 1024;
        float t8 = time_graph(s, N, [&] { read_regs<8><<<grid, 256, 0, s>>>((const u32x4*)buf, out, bytes, regions); });
        float t16 = bytes >= 65536 ? time_graph(s, N, [&] { read_regs<16><<<grid, 256, 0, s>>>((const u32x4*)buf, out, bytes, regions); }) : 0.f;
        float tl = time_graph(s, N, [&] { read_lds<256><<<grid, 256, 65536, s>>>(buf, out, bytes, regions); });
        float tl5 = time_graph(s, N, [&] { read_lds<512><<<grid, 512, 65536, s>>>(buf, out, bytes, regions); });
        printf("regions=%d grid=%d KB/WG=%d: regs U8 %.2f us  U16 %.2f us  glds256 %.2f us  glds512 %.2f us   (total %.1f MB; glds256 -> %.1f TB/s)\n",
               regions, grid, kb, t8, t16, tl, tl5, grid * (double)bytes / 1e6, grid * (double)bytes / tl / 1e6);
      }
    }
  }
  return 0;
}
