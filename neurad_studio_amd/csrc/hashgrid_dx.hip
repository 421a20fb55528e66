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


// ---------------------------------------------------------------------------------------------------------------------
// dL/d(origins, directions) of the static NeuRADHashEncoding path (H2 -> H3 -> H1 -> H4): what autograd hands a camera
// optimizer that moves the rays (cameras/camera_optimizers.py:173-182; `neurad-scaleopt`, method_configs.py:438-447).
//   mean = o + d t (rays.py:119; t from detached bins, ray_samplers.py:363-364)
//   u = mean / scale;  m = |u|_inf >= 1: c = (2 - 1/m) u / m, std' = (std / scale) ((2m - 1)^(1/3) / m)^2
//                                                                                 (spatial_distortions.py:126-141)
//   x = (c + 2) / 4, s = std' / 4;  enc_l = w_l(s) lerp_l(x),  w_l = 1 / max(1, 2 scal_l s)   (neurad_encoding.py:297-304)
// Both paths into the mean are differentiated: the trilinear offsets (floor / ceil carry no gradient, encodings.py:425-464)
// and the contracted std's dependence on m through the rescale weight.  View directions carry no gradient in the parity
// target (SHEncoding.pytorch_fwd is @torch.no_grad, encodings.py:797).
// One group of G lanes per ray (G = 16 / 32 / 64 by samples per ray), lanes stride over the samples, xor-butterfly over the
// group, lane 0 WRITES the ray's two rows: no atomics, bit-reproducible.  Samples whose whole gradient row is zero (behind
// an opaque surface; rows overridden by an actor) skip their 8 L gathers.
template <int F, bool HALF>
__global__ __launch_bounds__(256) void encode_bwd_rays_kernel(GridDev g, const void* __restrict__ table, float scale,
                                                               RaysDev r, const float* __restrict__ go, int G,
                                                               float* __restrict__ g_o, float* __restrict__ g_d) {
  const int lane = threadIdx.x & 63;
  const int sub = lane & (G - 1);
  const int64_t ray = ((int64_t)blockIdx.x * 256 + threadIdx.x) / G;
  const bool live_ray = ray < r.R;
  const uint32_t mask = (1u << g.log2T) - 1u;
  float ao[3] = {0.f, 0.f, 0.f}, ad[3] = {0.f, 0.f, 0.f};
  if (live_ray) {
    const float ox_ = r.o[3 * ray], oy_ = r.o[3 * ray + 1], oz_ = r.o[3 * ray + 2];
    const float dx_ = r.d[3 * ray], dy_ = r.d[3 * ray + 1], dz_ = r.d[3 * ray + 2];
    const float area = r.area[ray];
    for (int s = sub; s < r.S; s += G) {
      const int64_t i = ray * r.S + s;
      const float* gi = go + i * (int64_t)(g.L * F);
      bool any = false;
      for (int k = 0; k < g.L * F; ++k) any |= gi[k] != 0.f;
      if (!any) continue;
      const float t0 = r.starts[ray * r.stride + s], t1 = r.ends[ray * r.stride + s];
      const SamplePos gs = sample_gaussian(ox_, oy_, oz_, dx_, dy_, dz_, area, t0, t1);
      const SamplePos p = contract_gaussian(gs.x, gs.y, gs.z, gs.std, scale);
      float gx = 0.f, gy = 0.f, gz = 0.f, gstd = 0.f;
      for (int l = 0; l < g.L; ++l) {
        const float sc = g.scal[l];
        const Corners c = hash_corners(p.x, p.y, p.z, sc, mask);
        float f[8][F];
#pragma unroll
        for (int k = 0; k < 8; ++k) Entry<F, HALF>::load(table, ((uint32_t)l << g.log2T) + c.idx[k], f[k]);
        const float ox = c.ox, oy = c.oy, oz = c.oz, mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
        float dx = 0.f, dy = 0.f, dz = 0.f, dv = 0.f;
#pragma unroll
        for (int q = 0; q < F; ++q) {
          const float gq = gi[l * F + q];
          const float f03 = f[0][q] * ox + f[3][q] * mx, f12 = f[1][q] * ox + f[2][q] * mx;
          const float f56 = f[5][q] * ox + f[6][q] * mx, f47 = f[4][q] * ox + f[7][q] * mx;
          const float d03 = f[0][q] - f[3][q], d12 = f[1][q] - f[2][q], d56 = f[5][q] - f[6][q], d47 = f[4][q] - f[7][q];
          const float f0312 = f03 * oy + f12 * my, f4756 = f47 * oy + f56 * my;
          dx += gq * ((d03 * oy + d12 * my) * oz + (d47 * oy + d56 * my) * mz);
          dy += gq * ((f03 - f12) * oz + (f47 - f56) * mz);
          dz += gq * (f0312 - f4756);
          dv += gq * (f0312 * oz + f4756 * mz);
        }
        const float a2 = sc * 2.f * p.std;
        const float w = 1.f / fmaxf(a2, 1.f);
        gx += (sc * w) * dx, gy += (sc * w) * dy, gz += (sc * w) * dz;
        if (a2 > 1.f) gstd -= dv * (2.f * sc) * (w * w);
      }
      // contraction backward: (gx, gy, gz) = dL/dx01, gstd = dL/d cstd  ->  dL/d mean
      const float u[3] = {gs.x / scale, gs.y / scale, gs.z / scale};
      const float au[3] = {fabsf(u[0]), fabsf(u[1]), fabsf(u[2])};
      const int kmax = au[0] >= au[1] ? (au[0] >= au[2] ? 0 : 2) : (au[1] >= au[2] ? 1 : 2);
      const float mag = au[kmax];
      float gm[3] = {gx / 4.f, gy / 4.f, gz / 4.f};
      if (!(mag < 1.f)) {
        const float k = 2.f / mag - 1.f / (mag * mag), dk = -2.f / (mag * mag) + 2.f / (mag * mag * mag);
        const float cr = cbrtf(2.f * mag - 1.f), rr = cr / mag;
        const float dq = 2.f * rr * ((2.f / 3.f) / (cr * cr * mag) - cr / (mag * mag));
        const float g_mag = (gm[0] * u[0] + gm[1] * u[1] + gm[2] * u[2]) * dk + gstd * (gs.std / scale) * dq / 4.f;
        for (int c = 0; c < 3; ++c) gm[c] *= k;
        gm[kmax] += g_mag * (u[kmax] < 0.f ? -1.f : 1.f);
      }
      const float dist = (t1 - t0) / 2.f, t = t0 + dist;
      for (int c = 0; c < 3; ++c) {
        const float v = gm[c] / scale;
        ao[c] += v, ad[c] += v * t;
      }
    }
  }
  for (int off = 1; off < G; off <<= 1)
    for (int c = 0; c < 3; ++c) {
      ao[c] += __shfl_xor(ao[c], off, 64);
      ad[c] += __shfl_xor(ad[c], off, 64);
    }
  if (live_ray && sub == 0)
    for (int c = 0; c < 3; ++c) g_o[3 * ray + c] = ao[c], g_d[3 * ray + c] = ad[c];
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

extern "C" int nrhip_encode_bwd_rays(const nrhip_grid* g, const void* table, float static_scale, const nrhip_rays* rays,
                                     const float* grad_out, float* grad_origins, float* grad_directions, void* stream) {
  if (int e = validate_grid(g)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(table && grad_out && grad_origins && grad_directions && static_scale > 0.f, NRHIP_ERR_INVALID_ARG,
             "encode_bwd_rays: bad argument");
  if (rays->n_rays == 0) return NRHIP_OK;
  const GridDev gd = to_dev(*g);
  const RaysDev rd = to_dev(*rays);
  const int G = rd.S > 32 ? 64 : (rd.S > 16 ? 32 : 16);
  const int blocks = grid_for(rd.R * G, 256);
  const hipStream_t st = (hipStream_t)stream;
#define CALL(F)                                                                                                        \
  do {                                                                                                                 \
    if (g->param_dtype == 1)                                                                                           \
      encode_bwd_rays_kernel<F, true><<<blocks, 256, 0, st>>>(gd, table, static_scale, rd, grad_out, G, grad_origins, \
                                                              grad_directions);                                        \
    else                                                                                                               \
      encode_bwd_rays_kernel<F, false><<<blocks, 256, 0, st>>>(gd, table, static_scale, rd, grad_out, G, grad_origins, \
                                                               grad_directions);                                       \
  } while (0)
  switch (gd.F) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    default: CALL(8); break;
  }
#undef CALL
  return check_launch("encode_bwd_rays");
}
