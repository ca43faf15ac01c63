// Input staging: a batch is a row gather out of the split resident in HBM.
// Replaces the per-sample fetch / convert / .to(device) of Dataset_Wrap.__getitem__ (data_pipe/dataprepare.py:399-412) and
// the DataLoader collate: dst_t[i][:] = src_t[idx[i]][:] for up to four row tables in ONE launch (features, sequence codes,
// labels ... share the index list).  Pure byte movement, HBM/latency bound: one 16-byte chunk per thread where the row
// size and both base pointers allow it, narrower units otherwise.  An index outside [0, n_rows) writes a zero row (the
// host validates the index lists once per epoch; the kernel never reads out of bounds).
#include "common.h"

namespace emb {

constexpr int kGatherTables = 4;
struct GatherTable {
  const char* src;
  char* dst;
  long row_bytes;
  int unit;          // bytes per thread: 16, 8, 4 or 1
  long chunks;       // chunks per row = row_bytes / unit
  long end;          // exclusive prefix end of this table's chunk range: sum over tables of n * chunks
};
struct GatherArgs {
  GatherTable t[kGatherTables];
  int n_tables;
};

template <typename V> __device__ __forceinline__ void copy_chunk(const char* src, char* dst, bool ok) {
  V v = {};
  if (ok) v = *reinterpret_cast<const V*>(src);
  *reinterpret_cast<V*>(dst) = v;
}

__global__ __launch_bounds__(256) void gather_rows_kernel(const GatherArgs a, const int64_t* __restrict__ idx, long n_rows) {
  const long q = (long)blockIdx.x * 256 + threadIdx.x;
  // (tables are read at constant offsets -- no run-time index into the by-value argument, see DESIGN "toolchain traps")
  long begin = 0;
  GatherTable t = a.t[0];
  if (a.n_tables > 1 && q >= a.t[0].end) { begin = a.t[0].end; t = a.t[1]; }
  if (a.n_tables > 2 && q >= a.t[1].end) { begin = a.t[1].end; t = a.t[2]; }
  if (a.n_tables > 3 && q >= a.t[2].end) { begin = a.t[2].end; t = a.t[3]; }
  if (q >= t.end) return;
  const long w = q - begin, row = w / t.chunks, ch = w - row * t.chunks;
  const int64_t r = idx[row];
  const bool ok = r >= 0 && r < n_rows;
  const char* s = t.src + (ok ? r : 0) * t.row_bytes + ch * t.unit;
  char* d = t.dst + row * t.row_bytes + ch * t.unit;
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  switch (t.unit) {
    case 16: copy_chunk<u32x4>(s, d, ok); break;
    case 8: copy_chunk<uint64_t>(s, d, ok); break;
    case 4: copy_chunk<uint32_t>(s, d, ok); break;
    default: copy_chunk<uint8_t>(s, d, ok); break;
  }
}

}  // namespace emb

using namespace emb;

// ---- host side: the shuffles behind the balanced batch lists -----------------------------------------------------------
// BalancePos_BatchSampler (data_pipe/dataprepare.py:431-447) shuffles with Python's `random` module.  To yield the SAME
// lists without spending ~50 ms of interpreter time per epoch on a 130 k-row split, the two building blocks of
// random.shuffle are restated here: the MT19937 generator (Matsumoto & Nishimura; CPython's _randommodule.c keeps the
// 624-word state + position this function receives from random.Random(seed).getstate()), getrandbits(k) = top k bits of
// one 32-bit output, _randbelow(n) = draw bit_length(n) bits until the value is below n, and the shuffle itself:
// for i = n-1 .. 1: swap(x[i], x[randbelow(i+1)]).  Host code only; no device work.
static inline uint32_t mt_next(uint32_t* mt, int* pos) {
  if (*pos >= 624) {
    const uint32_t UP = 0x80000000u, LO = 0x7fffffffu, A = 0x9908b0dfu;
    int k = 0;
    for (; k < 624 - 397; ++k) {
      const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
      mt[k] = mt[k + 397] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    for (; k < 623; ++k) {
      const uint32_t y = (mt[k] & UP) | (mt[k + 1] & LO);
      mt[k] = mt[k + (397 - 624)] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    }
    const uint32_t y = (mt[623] & UP) | (mt[0] & LO);
    mt[623] = mt[396] ^ (y >> 1) ^ ((y & 1u) ? A : 0u);
    *pos = 0;
  }
  uint32_t y = mt[(*pos)++];
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

extern "C" int emb_mt19937_shuffle(uint32_t* state, int* pos, int64_t* items, int64_t n) {
  EMB_CHECK_ARG(state && pos && (items || n == 0), "emb_mt19937_shuffle: null pointer");
  EMB_CHECK_ARG(n >= 0 && n < (1LL << 31) && *pos >= 0 && *pos <= 624, "emb_mt19937_shuffle: bad length or state position");
  for (int64_t i = n - 1; i >= 1; --i) {
    const uint32_t bound = (uint32_t)(i + 1);
    const int bits = 32 - __builtin_clz(bound);
    uint32_t r;
    do r = mt_next(state, pos) >> (32 - bits); while (r >= bound);
    const int64_t t = items[i];
    items[i] = items[r];
    items[r] = t;
  }
  return EMB_OK;
}

extern "C" int emb_gather_rows(const void* const* src, void* const* dst, const int64_t* row_bytes, int n_tables,
                               const int64_t* idx, int64_t n, int64_t n_rows, emb_stream_t stream) {
  EMB_CHECK_ARG(src && dst && row_bytes && idx, "emb_gather_rows: null pointer");
  EMB_CHECK_ARG(n_tables >= 1 && n_tables <= kGatherTables, "emb_gather_rows: 1..%d tables per call", kGatherTables);
  EMB_CHECK_ARG(n >= 0 && n_rows > 0, "emb_gather_rows: bad row counts n=%lld n_rows=%lld", (long long)n, (long long)n_rows);
  if (n == 0) return EMB_OK;
  GatherArgs a{};
  a.n_tables = n_tables;
  long total = 0;
  for (int i = 0; i < n_tables; ++i) {
    EMB_CHECK_ARG(src[i] && dst[i] && row_bytes[i] > 0, "emb_gather_rows: table %d: null pointer or empty rows", i);
    GatherTable& t = a.t[i];
    t.src = (const char*)src[i];
    t.dst = (char*)dst[i];
    t.row_bytes = row_bytes[i];
    const uintptr_t mix = reinterpret_cast<uintptr_t>(src[i]) | reinterpret_cast<uintptr_t>(dst[i]) | (uintptr_t)row_bytes[i];
    t.unit = (mix % 16 == 0) ? 16 : (mix % 8 == 0) ? 8 : (mix % 4 == 0) ? 4 : 1;
    t.chunks = t.row_bytes / t.unit;
    total += n * t.chunks;
    t.end = total;
  }
  EMB_CHECK_ARG(total < (1L << 31) * 256, "emb_gather_rows: batch too large for one launch");
  const unsigned grid = (unsigned)((total + 255) / 256);
  gather_rows_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(a, idx, n_rows);
  EMB_CHECK_LAUNCH();
  return EMB_OK;
}
