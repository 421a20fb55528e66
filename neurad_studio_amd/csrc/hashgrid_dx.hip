// dL/dx of the hash-grid lookup: needed only for samples inside dynamic actors, whose box-frame positions depend on
// the optimised actor trajectories ("dL/dx only for actor-hit samples ... flowing through
// offset = scaled - floor").  Autograd of encodings.py:425-464 w.r.t. in_tensor: floor/ceil carry no gradient, so
// d enc / d x_a = scalings_l * d(lerp tree)/d offset_a.
#include "common.h"

namespace nrhip {

template <int F, bool HALF>
__device__ __forceinline__ void dx_of_sample(const GridDev& g, const void* __restrict__ table,
                                             const float* __restrict__ x, const float* __restrict__ go, int64_t i,
                                             float* __restrict__ gx) {
  const uint32_t mask = (1u << g.log2T) - 1u;
  const float px = x[3 * i], py = x[3 * i + 1], pz = x[3 * i + 2];
  float ax = 0.f, ay = 0.f, az = 0.f;
  for (int l = 0; l < g.L; ++l) {
    const float sc = g.scal[l];
    const Corners c = hash_corners(px, py, pz, sc, mask);
    float f[8][F];
#pragma unroll
    for (int k = 0; k < 8; ++k) Entry<F, HALF>::load(table, ((uint32_t)l << g.log2T) + c.idx[k], f[k]);
    const float ox = c.ox, oy = c.oy, oz = c.oz, mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
    float dx = 0.f, dy = 0.f, dz = 0.f;
#pragma unroll
    for (int q = 0; q < F; ++q) {
      const float gq = go[(i * g.L + l) * F + q];
      // corner order: 0 ccc, 1 cfc, 2 ffc, 3 fcc, 4 ccf, 5 cff, 6 fff, 7 fcf ; weight(c) = o, weight(f) = 1 - o
      const float f03 = f[0][q] * ox + f[3][q] * mx, f12 = f[1][q] * ox + f[2][q] * mx;
      const float f56 = f[5][q] * ox + f[6][q] * mx, f47 = f[4][q] * ox + f[7][q] * mx;
      const float d03 = f[0][q] - f[3][q], d12 = f[1][q] - f[2][q], d56 = f[5][q] - f[6][q], d47 = f[4][q] - f[7][q];
      dx += gq * ((d03 * oy + d12 * my) * oz + (d47 * oy + d56 * my) * mz);
      dy += gq * ((f03 - f12) * oz + (f47 - f56) * mz);
      dz += gq * ((f03 * oy + f12 * my) - (f47 * oy + f56 * my));
    }
    ax += sc * dx, ay += sc * dy, az += sc * dz;
  }
  gx[3 * i] = ax, gx[3 * i + 1] = ay, gx[3 * i + 2] = az;
}

template <int F, bool HALF>
__global__ __launch_bounds__(256) void hashgrid_bwd_input_kernel(GridDev g, const void* __restrict__ table,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ go, int64_t n,
                                                                  float* __restrict__ gx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx_of_sample<F, HALF>(g, table, x, go, i, gx);
}

// several grids of one shape (the per-actor grids): sample i looks into tables[grid_id[i]]
template <int F, bool HALF>
__global__ __launch_bounds__(256) void hashgrid_multi_bwd_input_kernel(GridDev g, const void* const* __restrict__ tables,
                                                                        const int32_t* __restrict__ grid_id,
                                                                        const float* __restrict__ x,
                                                                        const float* __restrict__ go, int64_t n,
                                                                        float* __restrict__ gx) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dx_of_sample<F, HALF>(g, tables[grid_id[i]], x, go, i, gx);
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_hashgrid_bwd_input(const nrhip_grid* g, const void* table, const float* x, const float* grad_out,
                                        int64_t n, float* grad_x, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(n >= 0, NRHIP_ERR_INVALID_ARG, "hashgrid_bwd_input: negative n");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(table && x && grad_out && grad_x, NRHIP_ERR_INVALID_ARG, "hashgrid_bwd_input: null pointer");
  const GridDev gd = to_dev(*g);
  const int blocks = grid_for(n, 256);
  const hipStream_t st = (hipStream_t)stream;
#define CALL(F)                                                                        \
  do {                                                                                 \
    if (g->param_dtype == 1) hashgrid_bwd_input_kernel<F, true><<<blocks, 256, 0, st>>>(gd, table, x, grad_out, n, grad_x);   \
    else hashgrid_bwd_input_kernel<F, false><<<blocks, 256, 0, st>>>(gd, table, x, grad_out, n, grad_x);                      \
  } while (0)
  switch (gd.F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    default: CALL(8); break;
  }
#undef CALL
  return check_launch("hashgrid_bwd_input");
}

extern "C" int nrhip_hashgrid_multi_bwd_input(const nrhip_grid* g, const void* const* tables, int32_t n_grids,
                                              const int32_t* grid_id, const float* x, const float* grad_out, int64_t n,
                                              float* grad_x, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(n >= 0 && n_grids >= 1, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_bwd_input: bad argument");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(tables && grid_id && x && grad_out && grad_x, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_bwd_input: null pointer");
  const GridDev gd = to_dev(*g);
  const int blocks = grid_for(n, 256);
  const hipStream_t st = (hipStream_t)stream;
#define CALL(F)                                                                        \
  do {                                                                                 \
    if (g->param_dtype == 1) hashgrid_multi_bwd_input_kernel<F, true><<<blocks, 256, 0, st>>>(gd, tables, grid_id, x, grad_out, n, grad_x);   \
    else hashgrid_multi_bwd_input_kernel<F, false><<<blocks, 256, 0, st>>>(gd, tables, grid_id, x, grad_out, n, grad_x);                      \
  } while (0)
  switch (gd.F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    default: CALL(8); break;
  }
#undef CALL
  return check_launch("hashgrid_multi_bwd_input");
}
