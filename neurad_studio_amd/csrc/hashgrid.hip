// H1 / H2-H4 / F3 / S2 standalone kernels: hash-grid lookup (fwd + scatter-add bwd), the fused
// "encode" (gaussian -> contraction -> lookup -> rescale), SH deg-4 and the proposal density head.
//
// Thread mapping for the lookups: one thread per (sample, level), LEVEL FASTEST.  Adjacent lanes are
// adjacent levels of the same sample, so the [N, L*F] output row is written fully coalesced
// (L*F*4 contiguous bytes per sample) while every lane issues its 8 independent corner gathers.
#include <cstdlib>

#include "common.h"

namespace nrhip {

// --------------------------------------------------------------------------------------------
template <int F, bool HALF>
__global__ __launch_bounds__(256) void hashgrid_fwd_kernel(GridDev g, const void* __restrict__ table,
                                                            const float* __restrict__ x, int64_t n,
                                                            float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * g.L) return;
  const int64_t i = t / g.L;
  const int l = (int)(t - i * g.L);
  const uint32_t mask = (1u << g.log2T) - 1u;
  float v[F];
  hash_level<F, HALF, true>(table, (uint32_t)l << g.log2T, x[3 * i], x[3 * i + 1], x[3 * i + 2], g.scal[l], mask, v);
  float* o = out + t * F;
#pragma unroll
  for (int k = 0; k < F; ++k) o[k] = v[k];
}

template <int F>
__global__ __launch_bounds__(256) void hashgrid_bwd_kernel(GridDev g, const float* __restrict__ x,
                                                            const float* __restrict__ go, int64_t n,
                                                            float* __restrict__ gt) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * g.L) return;
  const int64_t i = t / g.L;
  const int l = (int)(t - i * g.L);
  const uint32_t mask = (1u << g.log2T) - 1u;
  const Corners c = hash_corners(x[3 * i], x[3 * i + 1], x[3 * i + 2], g.scal[l], mask);
  float w[8];
  corner_weights(c, w);
  float gv[F];
#pragma unroll
  for (int k = 0; k < F; ++k) gv[k] = go[t * F + k];
  float* base = gt + ((size_t)l << g.log2T) * F;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float* p = base + (size_t)c.idx[k] * F;
#pragma unroll
    for (int j = 0; j < F; ++j) unsafeAtomicAdd(p + j, w[k] * gv[j]);
  }
}

// several grids of one shape (the per-actor grids, neurad_encoding.py:270-295): sample i looks into
// tables[grid_id[i]] -- one launch instead of one lookup per actor id.
template <int F, bool HALF>
__global__ __launch_bounds__(256) void hashgrid_multi_fwd_kernel(GridDev g, const void* const* __restrict__ tables,
                                                                  const int32_t* __restrict__ grid_id,
                                                                  const float* __restrict__ x, int64_t n,
                                                                  float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * g.L) return;
  const int64_t i = t / g.L;
  const int l = (int)(t - i * g.L);
  const uint32_t mask = (1u << g.log2T) - 1u;
  float v[F];
  hash_level<F, HALF>(tables[grid_id[i]], (uint32_t)l << g.log2T, x[3 * i], x[3 * i + 1], x[3 * i + 2], g.scal[l], mask,
                      v);
  float* o = out + t * F;
#pragma unroll
  for (int k = 0; k < F; ++k) o[k] = v[k];
}

template <int F>
__global__ __launch_bounds__(256) void hashgrid_multi_bwd_kernel(GridDev g, const int32_t* __restrict__ grid_id,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ go, int64_t n,
                                                                  float* const* __restrict__ gts, int n_grids) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (t >= n * g.L) return;
  const int64_t i = t / g.L;
  const int l = (int)(t - i * g.L);
  const uint32_t gid = (uint32_t)grid_id[i];
  if (gid >= (uint32_t)n_grids || gts[gid] == nullptr) return;  // no such grid / no gradient wanted: the row sends nothing
  const uint32_t mask = (1u << g.log2T) - 1u;
  const Corners c = hash_corners(x[3 * i], x[3 * i + 1], x[3 * i + 2], g.scal[l], mask);
  float w[8];
  corner_weights(c, w);
  float* base = gts[gid] + ((size_t)l << g.log2T) * F;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int j = 0; j < F; ++j) unsafeAtomicAdd(base + (size_t)c.idx[k] * F + j, w[k] * go[t * F + j]);
}

// The same scatter-add with run combining (round 5).  The rows of the actor branch are (sample, actor) pairs in sample
// order: consecutive rows are consecutive samples of one ray inside one box, which share a cell of the coarse actor levels
// (62 / 25 cm cells against samples a few cm apart).  One lane = one row, looping over the levels; equal (grid, entry) in
// neighbouring lanes of a 16-lane row form a run that is summed onto its first lane on DPP row shifts, and only run heads
// issue the memory-side atomics -- the whole cost of this pass (config[4] training: 12.8 M atomics, 0.72 + 2 x 0.31 ms).
template <int F>
__global__ __launch_bounds__(256) void hashgrid_multi_bwd_runs_kernel(GridDev g, const int32_t* __restrict__ grid_id,
                                                                       const float* __restrict__ x,
                                                                       const float* __restrict__ go, int64_t n,
                                                                       float* const* __restrict__ gts, int n_grids) {
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t i = i0 < n ? i0 : n - 1;
  const int lane = threadIdx.x & 63;
  const uint32_t gid0 = (uint32_t)grid_id[i];
  // a row whose grid does not exist (id outside [0, n_grids)) or wants no gradient (NULL entry) sends nothing
  float* const table = (i0 < n && gid0 < (uint32_t)n_grids) ? gts[gid0] : nullptr;
  const bool live = table != nullptr;
  const uint32_t gid = live ? gid0 : 0xffffffffu;
  const float px = x[3 * i], py = x[3 * i + 1], pz = x[3 * i + 2];
  const uint32_t mask = (1u << g.log2T) - 1u;
  for (int l = 0; l < g.L; ++l) {
    const Corners c = hash_corners(px, py, pz, g.scal[l], mask);
    float w[8];
    corner_weights(c, w);
    float gv[F];
#pragma unroll
    for (int j = 0; j < F; ++j) gv[j] = live ? go[(i * g.L + l) * F + j] : 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t key = live ? c.idx[k] : 0xffffffffu;
      const bool head = live && (dpp_row_shr<1>(key, ~key) != key || dpp_row_shr<1>(gid, ~gid) != gid);
      const unsigned long long hm = __ballot(head);
      float v[F];
#pragma unroll
      for (int j = 0; j < F; ++j) v[j] = w[k] * gv[j];
      if (hm != __ballot(live)) {  // some run is longer than 1: segmented suffix sum onto the run heads, inside each row
        const uint32_t run = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull));
#define NR_SEG_STEP(OFF)                                                   \
  {                                                                        \
    const bool same = dpp_row_shl<OFF>(run, 0xffffffffu) == run;           \
    _Pragma("unroll") for (int j = 0; j < F; ++j) {                        \
      const float t = dpp_row_shl<OFF>(v[j], 0.f);                         \
      if (same) v[j] += t;                                                 \
    }                                                                      \
  }
        NR_SEG_STEP(1)
        NR_SEG_STEP(2)
        NR_SEG_STEP(4)
        NR_SEG_STEP(8)
#undef NR_SEG_STEP
      }
      if (head) {
        float* e = table + (((size_t)l << g.log2T) + c.idx[k]) * F;
#pragma unroll
        for (int j = 0; j < F; ++j) unsafeAtomicAdd(e + j, v[j]);
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
// encode: H2 -> H3 -> H1 -> H4, thread per (sample, level)
template <int F, bool HALF>
__global__ __launch_bounds__(256) void encode_fwd_kernel(GridDev g, const void* __restrict__ table, float scale,
                                                          RaysDev r, float* __restrict__ out) {
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t n = r.R * r.S;
  if (t >= n * g.L) return;
  const int64_t i = t / g.L;
  const int l = (int)(t - i * g.L);
  const int64_t ray = i / r.S;
  const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                      r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + (i - ray * r.S)], r.ends[ray * r.stride + (i - ray * r.S)], scale);
  const uint32_t mask = (1u << g.log2T) - 1u;
  float v[F];
  hash_level<F, HALF, true>(table, (uint32_t)l << g.log2T, p.x, p.y, p.z, g.scal[l], mask, v);
  const float w = rescale_weight(g.scal[l], p.std);
  float* o = out + t * F;
#pragma unroll
  for (int k = 0; k < F; ++k) o[k] = v[k] * w;
}

// Scatter-add with wave-level run combining.  One lane = one sample, a wave = 64 CONSECUTIVE samples of a
// ray, looping over the levels.  Neighbouring samples of a ray fall into the same cell at the coarse levels, so
// equal corner indices come in runs of adjacent lanes: a segmented suffix sum (6 shuffle steps) collapses each
// run onto its first lane and only run heads issue the fp32 atomics.  Cuts the device-scope atomics -- the
// whole cost of this kernel -- by the mean run length (10-60x on the coarse half of the levels).
template <int F>
__global__ __launch_bounds__(256) void encode_bwd_runs_kernel(GridDev g, float scale, RaysDev r,
                                                               const float* __restrict__ go,
                                                               float* __restrict__ gt) {
  const int64_t n = r.R * r.S;
  const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i0 < n;
  const int64_t i = live ? i0 : n - 1;
  const int lane = threadIdx.x & 63;
  const int64_t ray = i / r.S;
  const int s = (int)(i - ray * r.S);
  const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                      r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + s],
                                      r.ends[ray * r.stride + s], scale);
  const uint32_t mask = (1u << g.log2T) - 1u;
  for (int l = 0; l < g.L; ++l) {
    const float sc = g.scal[l];
    const Corners c = hash_corners(p.x, p.y, p.z, sc, mask);
    float w[8];
    corner_weights(c, w);
    const float rw = live ? rescale_weight(sc, p.std) : 0.f;
    float gv[F];
#pragma unroll
    for (int k = 0; k < F; ++k) gv[k] = go[(i * g.L + l) * F + k] * rw;
    float* base = gt + ((size_t)l << g.log2T) * F;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t key = c.idx[k];
      const uint32_t prev = __shfl_up(key, 1, 64);
      const bool head = lane == 0 || prev != key;
      const unsigned long long hm = __ballot(head);
      float v[F];
#pragma unroll
      for (int j = 0; j < F; ++j) v[j] = w[k] * gv[j];
      if (hm != ~0ull) {  // at least one run longer than 1 in this wave
        const int run = __popcll(hm & ((2ull << lane) - 1ull));
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int orun = __shfl_down(run, off, 64);
          const bool same = (lane + off < 64) && orun == run;
#pragma unroll
          for (int j = 0; j < F; ++j) {
            const float t = __shfl_down(v[j], off, 64);
            if (same) v[j] += t;
          }
        }
      }
      if (head) {
        float* q = base + (size_t)key * F;
#pragma unroll
        for (int j = 0; j < F; ++j) unsafeAtomicAdd(q + j, v[j]);
      }
    }
  }
}

// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sh4_kernel(const float* __restrict__ d, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float c[16];
  sh4(d[3 * i], d[3 * i + 1], d[3 * i + 2], c);
  float4* o = reinterpret_cast<float4*>(out + 16 * i);
#pragma unroll
  for (int k = 0; k < 4; ++k) o[k] = make_float4(c[4 * k], c[4 * k + 1], c[4 * k + 2], c[4 * k + 3]);
}

// --------------------------------------------------------------------------------------------
// S2: proposal density = exp( sum_l w_l * rescaled_feature_l )  (F == 1).  Thread per sample; the L
// levels x 8 gathers of a sample are all independent loads issued back to back.
template <bool HALF>
__global__ __launch_bounds__(256) void proposal_density_fwd_kernel(GridDev g, const void* __restrict__ table,
                                                                    float scale, const float* __restrict__ dec,
                                                                    RaysDev r, float* __restrict__ dens,
                                                                    float* __restrict__ lf) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= r.R * r.S) return;
  const int64_t ray = i / r.S;
  const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                      r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + (i - ray * r.S)], r.ends[ray * r.stride + (i - ray * r.S)], scale);
  const uint32_t mask = (1u << g.log2T) - 1u;
  float acc = 0.f;
  for (int l = 0; l < g.L; ++l) {
    float v[1];
    hash_level<1, HALF>(table, (uint32_t)l << g.log2T, p.x, p.y, p.z, g.scal[l], mask, v);
    const float f = v[0] * rescale_weight(g.scal[l], p.std);
    if (lf) lf[(size_t)l * (r.R * r.S) + i] = f;  // level-major [L][N], as the level-partitioned path writes them
    acc += f * dec[l];
  }
  dens[i] = expf(acc);
}

// Training forward of S2, level-partitioned.  A proposal level (2^20 entries x 4 B) is exactly one XCD L2; a kernel
// in which every thread walks all levels keeps 6 x 4 MB in flight per XCD and misses most of the time.  Here the
// work is cut into (level, sample-quarter) units dealt round-robin to the XCDs in the dispatcher's observed block ->
// XCD order (block b on XCD b % 8; a wrong guess costs speed, not correctness): each XCD works through its units one
// after the other, so its L2 holds one level at a time.  Output: the rescaled per-level features, LEVEL-MAJOR [L][N]
// (coalesced), which the backward wants anyway; proposal_density_from_levels_kernel turns them into densities.
// (Round 3 tried the lanes of a wave on 64 NEIGHBOURING RAYS of a camera patch at one sample index instead of 64 consecutive
// samples of one ray -- the L1 charges a gather per distinct line, scripts/probes/l1_coalesce_probe.hip -- with the tiles
// transposed through LDS: 691 / 704 us per call against 497 us for this kernel (profiles/r03_proposal_fwd_transposed.txt).
// Consecutive samples of a ray already share the coarse levels' lines, and the tile form serialises four passes per wave.)
template <bool HALF>
__global__ __launch_bounds__(256) void proposal_levels_lp_kernel(GridDev g, const void* __restrict__ table, float scale,
                                                                  RaysDev r, float* __restrict__ lf, int64_t n,
                                                                  int64_t per_quarter, int64_t blocks_per_unit) {
  constexpr int kQuarters = 4;
  const int xcd = blockIdx.x & 7;
  const int64_t q = blockIdx.x >> 3;
  const int unit = xcd + 8 * (int)(q / blocks_per_unit);
  if (unit >= g.L * kQuarters) return;
  const int l = unit / kQuarters, quarter = unit - l * kQuarters;
  const int64_t i = quarter * per_quarter + (q % blocks_per_unit) * 256 + threadIdx.x;
  const int64_t hi = (quarter + 1) * per_quarter < n ? (quarter + 1) * per_quarter : n;
  if (i >= hi) return;
  const int64_t ray = i / r.S;
  const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                      r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + (i - ray * r.S)], r.ends[ray * r.stride + (i - ray * r.S)], scale);
  const uint32_t mask = (1u << g.log2T) - 1u;
  float v[1];
  hash_level<1, HALF, true>(table, (uint32_t)l << g.log2T, p.x, p.y, p.z, g.scal[l], mask, v);
  // streaming store: the feature stream must not push this XCD's level (exactly one L2) out of the L2
  __builtin_nontemporal_store(v[0] * rescale_weight(g.scal[l], p.std), lf + (size_t)l * n + i);
}

// VEC = 4: four consecutive samples per thread on 16-byte accesses (n % 4 == 0, 16-byte aligned buffers): the pass streams
// L + 1 floats per sample and is worth what it keeps in flight (scalar form: 46 us per 57 344 x 128 call, ~3 TB/s)
template <int VEC>
__global__ __launch_bounds__(256) void proposal_density_from_levels_kernel(const float* __restrict__ lf,
                                                                            const float* __restrict__ dec, int64_t n,
                                                                            int L, float* __restrict__ dens) {
  const int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  if (i >= n) return;
  float acc[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
  for (int l = 0; l < L; ++l) {  // same order of operations as the fused kernel
    const float w = dec[l];
    if constexpr (VEC == 4) {
      const float4 f = *reinterpret_cast<const float4*>(lf + (size_t)l * n + i);
      acc[0] += f.x * w, acc[1] += f.y * w, acc[2] += f.z * w, acc[3] += f.w * w;
    } else {
      acc[0] += lf[(size_t)l * n + i] * w;
    }
  }
  if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(dens + i) = make_float4(expf(acc[0]), expf(acc[1]), expf(acc[2]), expf(acc[3]));
  } else {
    dens[i] = expf(acc[0]);
  }
}

// backward of S2 through trunc_exp (activations.py:37-41: g * exp(clamp(x,-15,15))) into the decoder
// weight and the table.  x = log(density) is recovered from the saved forward output.
// TABLE=false: decoder gradient only (the table gradient is then produced by the binned path).
template <bool TABLE>
__global__ __launch_bounds__(256) void proposal_density_bwd_kernel(GridDev g, const void* __restrict__ table,
                                                                    float scale, const float* __restrict__ dec,
                                                                    RaysDev r, const float* __restrict__ dens,
                                                                    const float* __restrict__ gd,
                                                                    float* __restrict__ gt,
                                                                    float* __restrict__ gdec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = i < r.R * r.S;
  float gl[NRHIP_MAX_LEVELS > 8 ? 8 : NRHIP_MAX_LEVELS];
#pragma unroll
  for (int l = 0; l < 8; ++l) gl[l] = 0.f;
  if (live) {
    const int64_t ray = i / r.S;
    const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                        r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + (i - ray * r.S)], r.ends[ray * r.stride + (i - ray * r.S)], scale);
    const uint32_t mask = (1u << g.log2T) - 1u;
    const float xlog = logf(dens[i]);
    const float gx = gd[i] * expf(fminf(fmaxf(xlog, -15.f), 15.f));
#pragma unroll
    for (int l = 0; l < 8; ++l) {
      if (l < g.L) {
        const Corners c = hash_corners(p.x, p.y, p.z, g.scal[l], mask);
        float w[8];
        corner_weights(c, w);
        const float rw = rescale_weight(g.scal[l], p.std);
        const uint32_t row0 = (uint32_t)l << g.log2T;
        float f = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          float v[1];
          Entry<1, false>::load(table, row0 + c.idx[k], v);
          f += w[k] * v[0];
          if constexpr (TABLE) unsafeAtomicAdd(gt + row0 + c.idx[k], w[k] * (gx * dec[l] * rw));
        }
        gl[l] = gx * f * rw;
      }
    }
  }
  // block-reduce the decoder gradient: wave shuffle -> LDS -> one atomic per level per block
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    float v = gl[l];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wid][l] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8 && threadIdx.x < g.L) {
    const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    unsafeAtomicAdd(gdec + threadIdx.x, s);
  }
}

// --------------------------------------------------------------------------------------------
#define DISPATCH_F_HALF(F_, HALF_, CALL)                                   \
  do {                                                                     \
    const int f__ = (F_);                                                  \
    const bool h__ = (HALF_);                                              \
    if (f__ == 1) { if (h__) { CALL(1, true); } else { CALL(1, false); } } \
    else if (f__ == 2) { if (h__) { CALL(2, true); } else { CALL(2, false); } } \
    else if (f__ == 4) { if (h__) { CALL(4, true); } else { CALL(4, false); } } \
    else { if (h__) { CALL(8, true); } else { CALL(8, false); } }          \
  } while (0)

#define DISPATCH_F(F_, CALL)         \
  do {                               \
    const int f__ = (F_);            \
    if (f__ == 1) { CALL(1); }       \
    else if (f__ == 2) { CALL(2); }  \
    else if (f__ == 4) { CALL(4); }  \
    else { CALL(8); }                \
  } while (0)

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_hashgrid_fwd(const nrhip_grid* g, const void* table, const float* x, int64_t n, float* out,
                                  void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(n >= 0, NRHIP_ERR_INVALID_ARG, "hashgrid_fwd: negative n");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(table && x && out, NRHIP_ERR_INVALID_ARG, "hashgrid_fwd: null pointer");
  const GridDev gd = to_dev(*g);
  const int blocks = grid_for(n * gd.L, 256);
#define CALL(F, H) hashgrid_fwd_kernel<F, H><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, table, x, n, out)
  DISPATCH_F_HALF(gd.F, gd.dtype == 1, CALL);
#undef CALL
  return check_launch("hashgrid_fwd");
}

extern "C" int nrhip_hashgrid_bwd(const nrhip_grid* g, const float* x, const float* grad_out, int64_t n,
                                  float* grad_table, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(x && grad_out && grad_table && n >= 0, NRHIP_ERR_INVALID_ARG, "hashgrid_bwd: null pointer");
  if (n == 0) return NRHIP_OK;
  const GridDev gd = to_dev(*g);
  const int blocks = grid_for(n * gd.L, 256);
#define CALL(F) hashgrid_bwd_kernel<F><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, x, grad_out, n, grad_table)
  DISPATCH_F(gd.F, CALL);
#undef CALL
  return check_launch("hashgrid_bwd");
}

extern "C" int nrhip_encode_fwd(const nrhip_grid* g, const void* table, float static_scale, const nrhip_rays* rays,
                                float* out, void* stream) {
  if (int e = validate_grid(g)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(table && out && static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "encode_fwd: bad argument");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  const GridDev gd = to_dev(*g);
  const RaysDev rd = to_dev(*rays);
  const int blocks = grid_for(n * gd.L, 256);
#define CALL(F, H) encode_fwd_kernel<F, H><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, table, static_scale, rd, out)
  DISPATCH_F_HALF(gd.F, gd.dtype == 1, CALL);
#undef CALL
  return check_launch("encode_fwd");
}

extern "C" int nrhip_encode_bwd(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                                const float* grad_out, float* grad_table, void* stream) {
  if (int e = validate_grid(g)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(grad_out && grad_table && static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "encode_bwd: bad argument");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  const GridDev gd = to_dev(*g);
  const RaysDev rd = to_dev(*rays);
  const int blocks = grid_for(n, 256);
#define CALL(F) encode_bwd_runs_kernel<F><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, static_scale, rd, grad_out, grad_table)
  DISPATCH_F(gd.F, CALL);
#undef CALL
  return check_launch("encode_bwd");
}

extern "C" int nrhip_sh4_fwd(const float* dirs, int64_t n, float* out, void* stream) {
  NR_REQUIRE(dirs && out && n >= 0, NRHIP_ERR_INVALID_ARG, "sh4_fwd: null pointer");
  if (n == 0) return NRHIP_OK;
  sh4_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(dirs, n, out);
  return check_launch("sh4_fwd");
}

extern "C" int nrhip_proposal_density_fwd(const nrhip_proposal* p, const nrhip_rays* rays, float* density, float* level_features,
                                          void* stream) {
  NR_REQUIRE(p, NRHIP_ERR_INVALID_ARG, "proposal_density_fwd: null descriptor");
  if (int e = validate_grid(&p->grid)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(p->grid.n_features == 1, NRHIP_ERR_UNSUPPORTED,
             "proposal_density: features_per_level must be 1 (neurad_field.py:166), got %d", p->grid.n_features);
  NR_REQUIRE(p->table && p->decoder_weight && density && p->static_scale > 0.f, NRHIP_ERR_INVALID_ARG,
             "proposal_density_fwd: bad argument");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  const GridDev gd = to_dev(p->grid);
  const RaysDev rd = to_dev(*rays);
  if (level_features) {  // training: level-partitioned lookups + a streaming pass for the densities
    constexpr int kQuarters = 4;
    const int64_t per_quarter = ((n + kQuarters - 1) / kQuarters + 255) / 256 * 256;
    const int64_t blocks_per_unit = per_quarter / 256;
    const int units = gd.L * kQuarters;
    const int64_t nblk = 8 * ((units + 7) / 8) * blocks_per_unit;
    if (gd.dtype == 1)
      proposal_levels_lp_kernel<true><<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(
          gd, p->table, p->static_scale, rd, level_features, n, per_quarter, blocks_per_unit);
    else
      proposal_levels_lp_kernel<false><<<(unsigned)nblk, 256, 0, (hipStream_t)stream>>>(
          gd, p->table, p->static_scale, rd, level_features, n, per_quarter, blocks_per_unit);
    if (int e = check_launch("proposal_density_fwd levels")) return e;
    if (n % 4 == 0 && ((reinterpret_cast<uintptr_t>(level_features) | reinterpret_cast<uintptr_t>(density)) & 15) == 0)
      proposal_density_from_levels_kernel<4><<<grid_for(n / 4, 256), 256, 0, (hipStream_t)stream>>>(
          level_features, p->decoder_weight, n, gd.L, density);
    else
      proposal_density_from_levels_kernel<1><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(
          level_features, p->decoder_weight, n, gd.L, density);
    return check_launch("proposal_density_fwd");
  }
  if (gd.dtype == 1)
    proposal_density_fwd_kernel<true><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(
        gd, p->table, p->static_scale, p->decoder_weight, rd, density, level_features);
  else
    proposal_density_fwd_kernel<false><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(
        gd, p->table, p->static_scale, p->decoder_weight, rd, density, level_features);
  return check_launch("proposal_density_fwd");
}

extern "C" int nrhip_proposal_density_bwd(const nrhip_proposal* p, const nrhip_rays* rays, const float* density,
                                          const float* grad_density, float* grad_table, float* grad_decoder,
                                          void* stream) {
  NR_REQUIRE(p, NRHIP_ERR_INVALID_ARG, "proposal_density_bwd: null descriptor");
  if (int e = validate_grid(&p->grid)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(p->grid.n_features == 1 && p->grid.num_levels <= 8 && p->grid.param_dtype == 0, NRHIP_ERR_UNSUPPORTED,
             "proposal_density_bwd: needs F=1, L<=8, fp32 table");
  NR_REQUIRE(p->table && p->decoder_weight && density && grad_density && grad_table && grad_decoder,
             NRHIP_ERR_INVALID_ARG, "proposal_density_bwd: null pointer");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  proposal_density_bwd_kernel<true><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(
      to_dev(p->grid), p->table, p->static_scale, p->decoder_weight, to_dev(*rays), density, grad_density, grad_table,
      grad_decoder);
  return check_launch("proposal_density_bwd");
}

namespace nrhip {
// decoder gradient from the per-level features the forward saved: d dec[l] = sum_i g_i * exp(clamp(x_i)) * f_il
template <int VEC>
__global__ __launch_bounds__(256) void proposal_decoder_grad_kernel(const float* __restrict__ lf,
                                                                     const float* __restrict__ dens,
                                                                     const float* __restrict__ gd, int64_t n, int L,
                                                                     float* __restrict__ gdec) {
  float gl[8];
#pragma unroll
  for (int l = 0; l < 8; ++l) gl[l] = 0.f;
  const int64_t nv = n / VEC;  // (VEC == 4: n % 4 == 0, 16-byte aligned buffers -- checked by the launcher)
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nv; i += (int64_t)gridDim.x * 256) {
    float gx[VEC];
    if constexpr (VEC == 4) {
      const float4 g4 = reinterpret_cast<const float4*>(gd)[i], d4 = reinterpret_cast<const float4*>(dens)[i];
      gx[0] = g4.x * expf(fminf(fmaxf(logf(d4.x), -15.f), 15.f));
      gx[1] = g4.y * expf(fminf(fmaxf(logf(d4.y), -15.f), 15.f));
      gx[2] = g4.z * expf(fminf(fmaxf(logf(d4.z), -15.f), 15.f));
      gx[3] = g4.w * expf(fminf(fmaxf(logf(d4.w), -15.f), 15.f));
    } else {
      gx[0] = gd[i] * expf(fminf(fmaxf(logf(dens[i]), -15.f), 15.f));
    }
#pragma unroll
    for (int l = 0; l < 8; ++l)
      if (l < L) {  // level-major [L][N]
        if constexpr (VEC == 4) {
          const float4 f = reinterpret_cast<const float4*>(lf + (size_t)l * n)[i];
          gl[l] = fmaf(gx[0], f.x, gl[l]), gl[l] = fmaf(gx[1], f.y, gl[l]);
          gl[l] = fmaf(gx[2], f.z, gl[l]), gl[l] = fmaf(gx[3], f.w, gl[l]);
        } else {
          gl[l] = fmaf(gx[0], lf[(size_t)l * n + i], gl[l]);
        }
      }
  }
  __shared__ float red[4][8];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int l = 0; l < 8; ++l) {
    float v = gl[l];
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wid][l] = v;
  }
  __syncthreads();
  if (threadIdx.x < L) unsafeAtomicAdd(gdec + threadIdx.x, red[0][threadIdx.x] + red[1][threadIdx.x] +
                                                               red[2][threadIdx.x] + red[3][threadIdx.x]);
}
}  // namespace nrhip

namespace nrhip {
int proposal_table_grad_binned(const nrhip_proposal* p, const nrhip_rays* rays, const float* density,
                               const float* grad_density, float* grad_table, bool overwrite, void* workspace,
                               int64_t workspace_bytes, void* stream);  // encode_bwd_binned.hip
}

extern "C" int nrhip_proposal_density_bwd_binned(const nrhip_proposal* p, const nrhip_rays* rays,
                                                 const float* density, const float* level_features,
                                                 const float* grad_density, float* grad_table, float* grad_decoder,
                                                 int32_t overwrite, void* workspace, int64_t workspace_bytes,
                                                 void* stream) {
  NR_REQUIRE(p, NRHIP_ERR_INVALID_ARG, "proposal_density_bwd_binned: null descriptor");
  if (int e = validate_grid(&p->grid)) return e;
  if (int e = validate_rays(rays)) return e;
  // (with the forward's per-level features at hand the table itself is never read: fp16-storage tables are fine then)
  NR_REQUIRE(p->grid.n_features == 1 && p->grid.num_levels <= 8 && (p->grid.param_dtype == 0 || level_features),
             NRHIP_ERR_UNSUPPORTED,
             "proposal_density_bwd_binned: needs F=1, L<=8, and an fp32 table unless the forward's level features are passed");
  NR_REQUIRE(p->table && p->decoder_weight && density && grad_density && grad_table && grad_decoder,
             NRHIP_ERR_INVALID_ARG, "proposal_density_bwd_binned: null pointer");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  if (level_features) {
    int blocks = grid_for(n, 256);
    if (blocks > 1024) blocks = 1024;
    const bool vec = n % 4 == 0 && ((reinterpret_cast<uintptr_t>(level_features) | reinterpret_cast<uintptr_t>(density) |
                                     reinterpret_cast<uintptr_t>(grad_density)) & 15) == 0;
    if (vec)
      proposal_decoder_grad_kernel<4><<<blocks, 256, 0, (hipStream_t)stream>>>(level_features, density, grad_density, n,
                                                                             p->grid.num_levels, grad_decoder);
    else
      proposal_decoder_grad_kernel<1><<<blocks, 256, 0, (hipStream_t)stream>>>(level_features, density, grad_density, n,
                                                                             p->grid.num_levels, grad_decoder);
  } else {  // forward did not save them: recompute the interpolated features
    proposal_density_bwd_kernel<false><<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(
        to_dev(p->grid), p->table, p->static_scale, p->decoder_weight, to_dev(*rays), density, grad_density,
        grad_table, grad_decoder);
  }
  if (int e = check_launch("proposal_density_bwd_binned decoder")) return e;
  return proposal_table_grad_binned(p, rays, density, grad_density, grad_table, overwrite != 0, workspace,
                                    workspace_bytes, stream);
}

extern "C" int nrhip_hashgrid_multi_fwd(const nrhip_grid* g, const void* const* tables, int32_t n_grids,
                                        const int32_t* grid_id, const float* x, int64_t n, float* out, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(n >= 0 && n_grids >= 1, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_fwd: bad argument");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(tables && grid_id && x && out, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_fwd: null pointer");
  const GridDev gd = to_dev(*g);
  const int blocks = grid_for(n * gd.L, 256);
  if (g->param_dtype == 1) {  // fp16-storage grids (all grids of one call share the dtype)
#define CALL(F) hashgrid_multi_fwd_kernel<F, true><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, tables, grid_id, x, n, out)
    DISPATCH_F(gd.F, CALL);
#undef CALL
  } else {
#define CALL(F) hashgrid_multi_fwd_kernel<F, false><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, tables, grid_id, x, n, out)
    DISPATCH_F(gd.F, CALL);
#undef CALL
  }
  return check_launch("hashgrid_multi_fwd");
}

extern "C" int nrhip_hashgrid_multi_bwd(const nrhip_grid* g, int32_t n_grids, const int32_t* grid_id, const float* x,
                                        const float* grad_out, int64_t n, float* const* grad_tables, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(n >= 0 && n_grids >= 1, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_bwd: bad argument");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(grid_id && x && grad_out && grad_tables, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_bwd: null pointer");
  const GridDev gd = to_dev(*g);
  if (!tuning().multi_bwd_runs) {  // NRHIP_MULTI_BWD_RUNS=0: one thread per (row, level), every corner term its own atomic (A/B)
    const int blocks = grid_for(n * gd.L, 256);
#define CALL(F) hashgrid_multi_bwd_kernel<F><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, grid_id, x, grad_out, n, grad_tables, n_grids)
    DISPATCH_F(gd.F, CALL);
#undef CALL
  } else {
    const int blocks = grid_for(n, 256);
#define CALL(F) hashgrid_multi_bwd_runs_kernel<F><<<blocks, 256, 0, (hipStream_t)stream>>>(gd, grid_id, x, grad_out, n, grad_tables, n_grids)
    DISPATCH_F(gd.F, CALL);
#undef CALL
  }
  return check_launch("hashgrid_multi_bwd");
}
