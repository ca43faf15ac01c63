// Developer harness for the split kernels (diagnostic build with phase stamps; correctness lives in tests/).
//   make -C tools/kbench   ->   ./tools/kbench/kbench bwd B d0 d1 c [S]   |   ./tools/kbench/kbench fwd B d0 d1 c
#define EMB_SPLIT_PROF
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>
#include "embrace_bwd_split.h"
#include "gemm_jobs.h"
#include "embrace_split.h"

namespace emb {
void set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fputc('\n', stderr); }
int reduce_submit(const ReduceJob&, bool, hipStream_t) { return 0; }   // slab sums are timed separately (library)
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

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
    best = std::min(best, ms);
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / n;
}

template <typename T> static T* dev_random(size_t n, unsigned seed, float scale) {
  std::vector<T> h(n);
  unsigned x = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (T)(((x >> 8) & 0xFFFF) / 65536.0f * 2.0f * scale - scale); }
  T* d; CK(hipMalloc(&d, n * sizeof(T))); CK(hipMemcpy(d, h.data(), n * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

static void report_stamps(unsigned long long* dprof, int nblk, const char* const* kind_names, int nkinds) {
  std::vector<unsigned long long> h((size_t)nblk * 16);
  CK(hipMemcpy(h.data(), dprof, h.size() * 8, hipMemcpyDeviceToHost));
  unsigned long long w0 = ~0ull;
  for (int b = 0; b < nblk; ++b) if (h[(size_t)b * 16 + 1]) w0 = std::min(w0, h[(size_t)b * 16 + 1]);
  for (int k = 0; k < nkinds; ++k) {
    double sum[16] = {0}, start_sum = 0, start_max = 0, end_max = 0; int cnt = 0;
    for (int b = 0; b < nblk; ++b) {
      const unsigned long long* r = &h[(size_t)b * 16];
      if ((int)r[0] != k || r[2] == 0) continue;
      ++cnt;
      const double st = (double)(r[1] - w0) * 0.01;          // us (100 MHz wall clock)
      start_sum += st; start_max = std::max(start_max, st);
      sum[3] += (double)(r[3] - r[2]); sum[4] += (double)(r[4] - r[3]); sum[5] += (double)(r[5] - r[4]);
      if (r[6]) { sum[6] += (double)(r[6] - r[5]); sum[8] += (double)(r[8] - r[6]); } else sum[8] += (double)(r[8] - r[5]);
      // total duration in shader clocks -> us is unknown (clock varies); report cycles
      end_max = std::max(end_max, st);
    }
    if (!cnt) continue;
    {   // in-kernel clock: shader-clock ticks per 100 MHz wall tick between entry and exit (MI355X_MICROARCH.md, DVFS item 6)
      double ticks = 0, wall = 0;
      for (int b = 0; b < nblk; ++b) {
        const unsigned long long* r = &h[(size_t)b * 16];
        if ((int)r[0] != k || r[2] == 0 || r[9] == 0) continue;
        ticks += (double)(r[8] - r[2]); wall += (double)(r[9] - r[1]);
      }
      if (wall > 0) printf("  %-8s in-kernel clock %.2f GHz\n", kind_names[k], ticks / wall * 0.1);
    }
    printf("  %-8s n=%4d start avg %.2f max %.2f us | cycles: setup+request %.0f, first wait %.0f, main loop %.0f, park %.0f, tail (store / epilogue) %.0f | total %.0f\n",
           kind_names[k], cnt, start_sum / cnt, start_max, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt, sum[8] / cnt,
           (sum[3] + sum[4] + sum[5] + sum[6] + sum[8]) / cnt);
  }
}

int main(int argc, char** argv) {
  if (argc < 6) { printf("usage: kbench bwd|gj|fwd B d0 d1 c [S]\n"); return 1; }
  const bool gj = strcmp(argv[1], "gj") == 0;             // the persistent ring GEMM on pre-masked gradients (gemm_jobs.h)
  const bool pre = strcmp(argv[1], "bwdpre") == 0;        // the split kernel on pre-masked gradients (no mask stage)
  const bool bwd = strcmp(argv[1], "bwd") == 0 || gj || pre;
  const int B = atoi(argv[2]), d0 = atoi(argv[3]), d1 = atoi(argv[4]), c = atoi(argv[5]), S = argc > 6 ? atoi(argv[6]) : 0;
  hipStream_t s; CK(hipStreamCreate(&s));
  __bf16* X0 = dev_random<__bf16>((size_t)B * d0, 1, 1.0f);
  __bf16* X1 = dev_random<__bf16>((size_t)B * d1, 2, 1.0f);
  __bf16* W0 = dev_random<__bf16>((size_t)c * d0, 3, 0.1f);
  __bf16* W1 = dev_random<__bf16>((size_t)c * d1, 4, 0.02f);
  __bf16* dE = dev_random<__bf16>((size_t)B * c, 5, 1.0f);
  std::vector<uint8_t> hc((size_t)B * c);
  unsigned x = 99;
  for (auto& v : hc) { x = x * 1664525u + 1013904223u; const int idx = (x >> 10) & 1, act = (x >> 11) & 1; v = (uint8_t)(idx | (act << 1) | (act ? (idx ? 128 : 64) : 0)); }
  uint8_t* code; CK(hipMalloc(&code, hc.size())); CK(hipMemcpy(code, hc.data(), hc.size(), hipMemcpyHostToDevice));
  __bf16 *dX0, *dX1, *E; float *dW0, *dW1, *db0, *db1, *b0, *b1, *cdf0; void* ws;
  CK(hipMalloc(&dX0, (size_t)B * d0 * 2)); CK(hipMalloc(&dX1, (size_t)B * d1 * 2)); CK(hipMalloc(&E, (size_t)B * c * 2));
  CK(hipMalloc(&dW0, (size_t)c * d0 * 4)); CK(hipMalloc(&dW1, (size_t)c * d1 * 4)); CK(hipMalloc(&db0, c * 4)); CK(hipMalloc(&db1, c * 4));
  CK(hipMalloc(&b0, c * 4)); CK(hipMalloc(&b1, c * 4)); CK(hipMemset(b0, 0, c * 4)); CK(hipMemset(b1, 0, c * 4));
  CK(hipMalloc(&cdf0, B * 4));
  { std::vector<float> h(B, 0.578f); CK(hipMemcpy(cdf0, h.data(), B * 4, hipMemcpyHostToDevice)); }
  const int64_t ws_bytes = 64ll << 20; CK(hipMalloc(&ws, ws_bytes));
  unsigned long long* dprof; const int max_blk = 1 << 16; CK(hipMalloc(&dprof, (size_t)max_blk * 16 * 8));
  unsigned long long* null_prof = nullptr;

  std::function<void()> launch;
  emb::SelArgs sel{cdf0, nullptr, nullptr, nullptr, 0, 0};
  float *fX0 = nullptr, *fX1 = nullptr, *fW0 = nullptr, *fW1 = nullptr, *fdE = nullptr, *fdX0 = nullptr, *fdX1 = nullptr;
  if (gj) {   // fp32 operands
    fX0 = dev_random<float>((size_t)B * d0, 1, 1.0f); fX1 = dev_random<float>((size_t)B * d1, 2, 1.0f);
    fW0 = dev_random<float>((size_t)c * d0, 3, 0.1f); fW1 = dev_random<float>((size_t)c * d1, 4, 0.02f);
    fdE = dev_random<float>((size_t)B * c, 5, 1.0f);
    CK(hipMalloc(&fdX0, (size_t)B * d0 * 4)); CK(hipMalloc(&fdX1, (size_t)B * d1 * 4));
    launch = [&] { if (emb::gemm_jobs_bwd_impl(fdE, fdE, fX0, fX1, fW0, fW1, fdX0, fdX1, dW0, db0, dW1, db1, ws, ws_bytes, B, d0, d1, c, S, s) != 0) { printf("dispatch refused\n"); exit(1); } };
  } else if (bwd) {
    launch = [&] { if (emb::bwd_split_dispatch(dE, code, pre ? dE : nullptr, pre ? dE : nullptr, X0, X1, W0, W1, dX0, dX1, dW0, db0, dW1, db1, ws, ws_bytes, B, d0, d1, c, S, s) != 0) { printf("dispatch refused\n"); exit(1); } };
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, emb::embrace_bwd_split_kernel<true, 2>, emb::kBwdThreads, emb::kBwdLds<true, 2>));
    printf("occupancy query (bwd): %d blocks/CU\n", occ);
  } else {
    launch = [&] { if (emb::fwd_split_dispatch<__bf16>(X0, X1, W0, b0, W1, b1, sel, nullptr, 7, 1, nullptr, 0, E, code, B, d0, d1, c, s) != 0) { printf("dispatch refused\n"); exit(1); } };
  }
  CK(hipMemcpyToSymbol(HIP_SYMBOL(emb::g_split_prof), &null_prof, sizeof(null_prof)));
  const float us = time_graph(s, 100, launch);
  printf("%s B=%d d0=%d d1=%d c=%d S=%d: %.2f us/launch (stamps off)\n", argv[1], B, d0, d1, c, S, us);
  {
    CK(hipMemset(dprof, 0, (size_t)max_blk * 16 * 8));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(emb::g_split_prof), &dprof, sizeof(dprof)));
    launch(); CK(hipStreamSynchronize(s));
    CK(hipMemset(dprof, 0, (size_t)max_blk * 16 * 8));
    launch(); CK(hipStreamSynchronize(s));
    static const char* const names[] = {"wgrad0", "wgrad1", "dgrad0", "dgrad1"};
    static const char* const gnames[] = {"dgrad", "wgrad"};   // first tile of every persistent workgroup
    static const char* const fnames[] = {"fwd"};
    report_stamps(dprof, max_blk, gj ? gnames : (bwd ? names : fnames), gj ? 2 : (bwd ? 4 : 1));
  }
  return 0;
}
