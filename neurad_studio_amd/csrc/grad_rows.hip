// the level-sparse form of the gradient exchange (neurad_studio_amd/parallel/data_parallel.py, opt-in).
// A hash table's coarse levels receive few distinct rows per step, so those levels travel between the ranks as
// (row, values) lists instead of dense slabs.  These kernels are the device side of that: count the non-zero rows of every
// level, compact chosen levels into ordered lists, and apply lists to a gradient.  Everything is ordered and atomic-free on
// the data (the only atomics are integer counters), so every rank that applies the same lists in the same order ends with
// bit-identical gradients -- the property DDP's all-reduce gives the reference (pipelines/base_pipeline.py:304-307).
//
//   count   : grad [L*T, F] -> block_counts [L, nblk] (non-zero rows per block of 256 rows), level_counts [L]
//   scan    : block_counts -> exclusive prefix per level, in place
//   compact : for the levels of a list: rows [cap_l] int32 (ascending, level-local; the caller pre-fills -1 = padding) and
//             vals [cap_l, F] = grad * scale at those rows
//   apply   : grad[level, row] = 0 (mode 0) or += vals (mode 1) for the entries with row >= 0; rows of ONE list are distinct
#include "common.h"
#include "wave_scan.h"

namespace nrhip {
namespace {

constexpr int kRowsPerBlock = NRHIP_GRAD_ROWS_PER_BLOCK;
constexpr int kMaxListLevels = 32;

struct ListLevels {  // the levels of one list, in list order: entries [start[i], start[i + 1]) belong to level[i]
  int32_t n;
  int32_t level[kMaxListLevels];
  int64_t start[kMaxListLevels + 1];
};

template <int F>
__device__ __forceinline__ bool row_nonzero(const float* __restrict__ g, int64_t row) {
  const float* p = g + row * F;
  if constexpr (F % 4 == 0) {
    bool nz = false;
#pragma unroll
    for (int q = 0; q < F / 4; ++q) {
      const float4 v = reinterpret_cast<const float4*>(p)[q];
      nz = nz || v.x != 0.f || v.y != 0.f || v.z != 0.f || v.w != 0.f;
    }
    return nz;
  } else if constexpr (F == 2) {
    const float2 v = *reinterpret_cast<const float2*>(p);
    return v.x != 0.f || v.y != 0.f;
  } else {
    bool nz = false;
#pragma unroll
    for (int q = 0; q < F; ++q) nz = nz || p[q] != 0.f;
    return nz;
  }
}

// (NaN != 0 is true: a poisoned row is sent, like any other value)
template <int F>
__global__ __launch_bounds__(kRowsPerBlock) void rows_count_kernel(const float* __restrict__ grad, int64_t T, int nblk,
                                                                    uint32_t* __restrict__ block_counts,
                                                                    unsigned long long* __restrict__ level_counts) {
  __shared__ uint32_t wsum[kRowsPerBlock / 64];
  const int l = blockIdx.y;
  const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x;
  const bool nz = r < T && row_nonzero<F>(grad, (int64_t)l * T + r);
  const unsigned long long m = __ballot(nz);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = (uint32_t)__popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t c = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    block_counts[(size_t)l * nblk + blockIdx.x] = c;
    if (c) atomicAdd(&level_counts[l], (unsigned long long)c);
  }
}

__global__ __launch_bounds__(1024) void rows_scan_kernel(uint32_t* __restrict__ block_counts, int nblk) {
  __shared__ uint32_t part[16];
  uint32_t* row = block_counts + (size_t)blockIdx.x * nblk;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t carry = 0;
  for (int b0 = 0; b0 < nblk; b0 += 1024) {
    const int b = b0 + threadIdx.x;
    const uint32_t v = b < nblk ? row[b] : 0u;
    const uint32_t incl = wscan::incl<wscan::Add>(v, lane);
    if (lane == 63) part[wave] = incl;
    __syncthreads();
    uint32_t before = carry, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      before += w < wave ? part[w] : 0u;
      total += part[w];
    }
    if (b < nblk) row[b] = before + incl - v;
    carry += total;
    __syncthreads();
  }
}

template <int F>
__global__ __launch_bounds__(kRowsPerBlock) void rows_compact_kernel(const float* __restrict__ grad, int64_t T, int nblk,
                                                                      const uint32_t* __restrict__ block_offsets,
                                                                      ListLevels ll, float scale, int32_t* __restrict__ rows,
                                                                      float* __restrict__ vals) {
  __shared__ uint32_t wsum[kRowsPerBlock / 64];
  const int l = ll.level[blockIdx.y];
  const int64_t r = (int64_t)blockIdx.x * kRowsPerBlock + threadIdx.x;
  const bool nz = r < T && row_nonzero<F>(grad, (int64_t)l * T + r);
  const unsigned long long m = __ballot(nz);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) wsum[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  if (!nz) return;
  uint32_t rank = block_offsets[(size_t)l * nblk + blockIdx.x] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
  for (int w = 0; w < wave; ++w) rank += wsum[w];
  const int64_t o = ll.start[blockIdx.y] + rank;
  if (o >= ll.start[blockIdx.y + 1]) return;  // (cannot happen: the capacity is the agreed maximum over the ranks)
  rows[o] = (int32_t)r;
  const float* p = grad + ((int64_t)l * T + r) * F;
#pragma unroll
  for (int q = 0; q < F; ++q) vals[o * F + q] = p[q] * scale;
}

template <int F>
__global__ __launch_bounds__(256) void rows_apply_kernel(float* __restrict__ grad, int64_t T, ListLevels ll,
                                                          const int32_t* __restrict__ rows, const float* __restrict__ vals,
                                                          int mode) {
  const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (e >= ll.start[ll.n]) return;
  const int32_t r = rows[e];
  if (r < 0) return;  // padding
  int i = 0;
  while (i + 1 < ll.n && e >= ll.start[i + 1]) ++i;
  float* p = grad + ((int64_t)ll.level[i] * T + r) * F;
#pragma unroll
  for (int q = 0; q < F; ++q) p[q] = mode == 0 ? 0.f : p[q] + vals[e * F + q];
}

int fill_levels(const int32_t* levels, const int64_t* caps, int32_t n, int32_t n_levels, ListLevels* ll, const char* what) {
  NR_REQUIRE(n >= 1 && n <= kMaxListLevels && levels && caps, NRHIP_ERR_INVALID_ARG, "%s: 1..%d list levels", what,
             kMaxListLevels);
  ll->n = n;
  ll->start[0] = 0;
  for (int i = 0; i < n; ++i) {
    NR_REQUIRE(levels[i] >= 0 && levels[i] < n_levels && caps[i] >= 0, NRHIP_ERR_INVALID_ARG, "%s: bad level / capacity", what);
    ll->level[i] = levels[i];
    ll->start[i + 1] = ll->start[i] + caps[i];
  }
  return NRHIP_OK;
}

}  // namespace
}  // namespace nrhip

using namespace nrhip;

#define NR_ROWS_F(F, CALL)                                                                                  \
  switch (F) {                                                                                              \
    case 1: CALL(1); break;                                                                                 \
    case 2: CALL(2); break;                                                                                 \
    case 4: CALL(4); break;                                                                                 \
    case 8: CALL(8); break;                                                                                 \
    default: NR_REQUIRE(false, NRHIP_ERR_UNSUPPORTED, "grad_rows: features_per_level %d not in {1,2,4,8}", (int)F); \
  }

static int64_t rows_blocks(int64_t rows_per_level) { return (rows_per_level + kRowsPerBlock - 1) / kRowsPerBlock; }

extern "C" int nrhip_grad_rows_count(const float* grad, int32_t n_levels, int64_t rows_per_level, int32_t f,
                                     uint32_t* block_counts, int64_t* level_counts, void* stream) {
  NR_REQUIRE(grad && block_counts && level_counts && n_levels >= 1 && n_levels <= 65535 && rows_per_level >= 1,
             NRHIP_ERR_INVALID_ARG, "grad_rows_count: bad argument");
  NR_REQUIRE((reinterpret_cast<uintptr_t>(grad) & 15) == 0, NRHIP_ERR_INVALID_ARG, "grad_rows_count: grad must be 16-byte aligned");
  const int nblk = (int)rows_blocks(rows_per_level);
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(level_counts, 0, (size_t)n_levels * sizeof(int64_t), st) != hipSuccess) return check_launch("grad_rows_count");
#define CALL(F)                                                                                                       \
  rows_count_kernel<F><<<dim3((unsigned)nblk, (unsigned)n_levels), kRowsPerBlock, 0, st>>>(                           \
      grad, rows_per_level, nblk, block_counts, reinterpret_cast<unsigned long long*>(level_counts))
  NR_ROWS_F(f, CALL);
#undef CALL
  rows_scan_kernel<<<n_levels, 1024, 0, st>>>(block_counts, nblk);
  return check_launch("grad_rows_count");
}

extern "C" int nrhip_grad_rows_compact(const float* grad, int32_t n_levels, int64_t rows_per_level, int32_t f,
                                       const uint32_t* block_offsets, const int32_t* levels, const int64_t* caps,
                                       int32_t n_list_levels, float scale, int32_t* rows, float* vals, void* stream) {
  NR_REQUIRE(grad && block_offsets && rows && vals && rows_per_level >= 1, NRHIP_ERR_INVALID_ARG,
             "grad_rows_compact: bad argument");
  ListLevels ll;
  if (int e = fill_levels(levels, caps, n_list_levels, n_levels, &ll, "grad_rows_compact")) return e;
  if (ll.start[ll.n] == 0) return NRHIP_OK;
  const int nblk = (int)rows_blocks(rows_per_level);
#define CALL(F)                                                                                                           \
  rows_compact_kernel<F><<<dim3((unsigned)nblk, (unsigned)ll.n), kRowsPerBlock, 0, (hipStream_t)stream>>>(                \
      grad, rows_per_level, nblk, block_offsets, ll, scale, rows, vals)
  NR_ROWS_F(f, CALL);
#undef CALL
  return check_launch("grad_rows_compact");
}

extern "C" int nrhip_grad_rows_apply(float* grad, int32_t n_levels, int64_t rows_per_level, int32_t f, const int32_t* levels,
                                     const int64_t* caps, int32_t n_list_levels, const int32_t* rows, const float* vals,
                                     int32_t mode, void* stream) {
  NR_REQUIRE(grad && rows && (mode == 0 || vals) && (mode == 0 || mode == 1) && rows_per_level >= 1, NRHIP_ERR_INVALID_ARG,
             "grad_rows_apply: bad argument");
  ListLevels ll;
  if (int e = fill_levels(levels, caps, n_list_levels, n_levels, &ll, "grad_rows_apply")) return e;
  const int64_t total = ll.start[ll.n];
  if (total == 0) return NRHIP_OK;
#define CALL(F)                                                                                                      \
  rows_apply_kernel<F><<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(grad, rows_per_level, ll, rows, \
                                                                                          vals, mode)
  NR_ROWS_F(f, CALL);
#undef CALL
  return check_launch("grad_rows_apply");
}
