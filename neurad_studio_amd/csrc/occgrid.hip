// S6: occupancy-grid ray march with wavefront ballot / prefix-popcount compaction.
// (VolumetricSampler.forward -> nerfacc OccGridEstimator.sampling, model_components/ray_samplers.py:483-566.)
// nerfacc is un-vendored and nothing in neurad-studio instantiates VolumetricSampler, so the marching rule is stated
// here (and restated in oracle/neurad_oracle.py:occgrid_march) -- parity unpinned, see DESIGN.md:
//   * the ray is clipped to [max(near_plane, t_min), min(far_plane, t_max)] and to the grid's AABB;
//   * candidate intervals tile that range back to back: dt = max(t * cone_angle, step)  (uniform when cone_angle = 0);
//   * a candidate is kept iff the cell containing its midpoint is occupied; kept intervals are emitted in order as
//     packed (ray_index, t_start, t_end).
// One wavefront marches one ray 64 candidates at a time: ballot(occupied) -> popcount prefix -> compacted store.
#include "common.h"

namespace nrhip {

struct OccDev {
  float lo[3], hi[3];
  int res;
  const uint8_t* bin;
};

struct March {
  float t0, t_far, step, c, t1;
  int k1;  // number of uniform steps before the cone takes over
  __device__ __forceinline__ float at(int k) const {
    if (c <= 0.f || k < k1) return t0 + (float)k * step;
    return t1 * powf(1.f + c, (float)(k - k1));
  }
};

__device__ __forceinline__ bool setup_march(const OccDev& g, const float* o, const float* d, float near, float far,
                                            float step, float cone, March& m) {
  float tn = near, tf = far;
#pragma unroll
  for (int a = 0; a < 3; ++a) {  // slab test against the grid AABB
    const float inv = 1.f / d[a];
    float ta = (g.lo[a] - o[a]) * inv, tb = (g.hi[a] - o[a]) * inv;
    if (d[a] == 0.f) {
      if (o[a] < g.lo[a] || o[a] > g.hi[a]) return false;
      continue;
    }
    if (ta > tb) { const float s = ta; ta = tb; tb = s; }
    tn = fmaxf(tn, ta), tf = fminf(tf, tb);
  }
  if (!(tn < tf)) return false;
  m.t0 = tn, m.t_far = tf, m.step = step, m.c = cone, m.k1 = 0, m.t1 = tn;
  if (cone > 0.f && tn * cone < step) {
    m.k1 = (int)ceilf((step / cone - tn) / step);
    m.t1 = tn + (float)m.k1 * step;
  }
  return true;
}

__device__ __forceinline__ bool occupied(const OccDev& g, const float* o, const float* d, float tm) {
  int idx = 0;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float u = (o[a] + d[a] * tm - g.lo[a]) / (g.hi[a] - g.lo[a]);
    const int i = min(max((int)floorf(u * (float)g.res), 0), g.res - 1);
    idx = idx * g.res + i;
  }
  return g.bin[idx] != 0;
}

// WRITE=false: counts[ray] ; WRITE=true: packed outputs at offsets[ray]
template <bool WRITE>
__global__ __launch_bounds__(256) void occgrid_march_kernel(OccDev g, const float* __restrict__ origins,
                                                            const float* __restrict__ dirs,
                                                            const float* __restrict__ t_min,
                                                            const float* __restrict__ t_max,
                                                            const float* __restrict__ t_rand, int64_t R, float step,
                                                            float near_plane, float far_plane, float cone,
                                                            int max_candidates, int32_t* __restrict__ counts,
                                                            const int64_t* __restrict__ offsets,
                                                            int64_t* __restrict__ ray_indices,
                                                            float* __restrict__ t_starts, float* __restrict__ t_ends) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float o[3] = {origins[3 * ray], origins[3 * ray + 1], origins[3 * ray + 2]};
  const float d[3] = {dirs[3 * ray], dirs[3 * ray + 1], dirs[3 * ray + 2]};
  float near = fmaxf(near_plane, t_min ? t_min[ray] : near_plane);
  const float far = fminf(far_plane, t_max ? t_max[ray] : far_plane);
  if (t_rand) near += t_rand[ray] * step;  // stratified=True (nerfacc shifts the near plane by U(0,1)*step)
  March m;
  int total = 0;
  if (setup_march(g, o, d, near, far, step, cone, m)) {
    for (int k0 = 0; k0 < max_candidates; k0 += 64) {
      const int k = k0 + lane;
      const float ts = m.at(k);
      const float te = fminf(m.at(k + 1), m.t_far);
      const bool in_range = ts < m.t_far && te > ts;
      const bool keep = in_range && occupied(g, o, d, 0.5f * (ts + te));
      const unsigned long long mk = __ballot(keep);
      if (WRITE && keep) {
        const int64_t pos = offsets[ray] + total + __popcll(mk & ((1ull << lane) - 1ull));
        ray_indices[pos] = ray;
        t_starts[pos] = ts;
        t_ends[pos] = te;
      }
      total += __popcll(mk);
      if (__ballot(in_range) != ~0ull) break;  // ran past t_far: done (early termination of the wave)
    }
  }
  if (!WRITE && lane == 0) counts[ray] = total;
}

static int to_dev(const nrhip_occgrid* g, OccDev& d) {
  NR_REQUIRE(g && g->binaries && g->resolution >= 1 && g->resolution <= 1024, NRHIP_ERR_INVALID_ARG,
             "occgrid: NULL grid or resolution outside [1,1024]");
  for (int a = 0; a < 3; ++a) {
    d.lo[a] = g->aabb[a], d.hi[a] = g->aabb[3 + a];
    NR_REQUIRE(d.hi[a] > d.lo[a], NRHIP_ERR_INVALID_ARG, "occgrid: empty AABB");
  }
  d.res = g->resolution, d.bin = g->binaries;
  return NRHIP_OK;
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_occgrid_march(const nrhip_occgrid* grid, const float* origins, const float* directions,
                                   const float* t_min, const float* t_max, const float* t_rand, int64_t r,
                                   float render_step_size, float near_plane, float far_plane, float cone_angle,
                                   int32_t max_candidates, int32_t* counts, const int64_t* offsets,
                                   int64_t* ray_indices, float* t_starts, float* t_ends, void* stream) {
  OccDev d;
  if (int e = to_dev(grid, d)) return e;
  NR_REQUIRE(r >= 0 && render_step_size > 0.f && cone_angle >= 0.f && max_candidates >= 1, NRHIP_ERR_INVALID_ARG,
             "occgrid_march: bad argument");
  if (r == 0) return NRHIP_OK;
  NR_REQUIRE(origins && directions, NRHIP_ERR_INVALID_ARG, "occgrid_march: null rays");
  const int blocks = (int)((r + 3) / 4);
  const hipStream_t st = (hipStream_t)stream;
  if (!offsets) {
    NR_REQUIRE(counts, NRHIP_ERR_INVALID_ARG, "occgrid_march: counting pass needs `counts`");
    occgrid_march_kernel<false><<<blocks, 256, 0, st>>>(d, origins, directions, t_min, t_max, t_rand, r,
                                                        render_step_size, near_plane, far_plane, cone_angle,
                                                        max_candidates, counts, nullptr, nullptr, nullptr, nullptr);
  } else {
    NR_REQUIRE(ray_indices && t_starts && t_ends, NRHIP_ERR_INVALID_ARG, "occgrid_march: write pass needs outputs");
    occgrid_march_kernel<true><<<blocks, 256, 0, st>>>(d, origins, directions, t_min, t_max, t_rand, r,
                                                       render_step_size, near_plane, far_plane, cone_angle,
                                                       max_candidates, nullptr, offsets, ray_indices, t_starts, t_ends);
  }
  return check_launch("occgrid_march");
}

namespace nrhip {
// nerfacc render_visibility_from_alpha (packed): keep sample i of a ray iff T_i >= early_stop_eps and
// alpha_i >= alpha_thre, T_i = prod_{j<i}(1 - alpha_j) over the ray's packed segment [seg[r], seg[r+1]).
__global__ __launch_bounds__(256) void packed_visibility_kernel(const float* __restrict__ alphas,
                                                                const int64_t* __restrict__ seg, int64_t R,
                                                                float early_stop_eps, float alpha_thre,
                                                                uint8_t* __restrict__ mask) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const int64_t b = seg[ray], e = seg[ray + 1];
  float carry = 1.f;
  for (int64_t i0 = b; i0 < e; i0 += 64) {
    const int64_t i = i0 + lane;
    const float a = i < e ? alphas[i] : 0.f;
    float incl = 1.f - a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const float u = __shfl_up(incl, off, 64);
      if (lane >= off) incl *= u;
    }
    float excl = __shfl_up(incl, 1, 64);
    if (lane == 0) excl = 1.f;
    const float T = carry * excl;
    if (i < e) mask[i] = (T >= early_stop_eps && a >= alpha_thre) ? 1 : 0;
    carry *= __shfl(incl, 63, 64);
    if (carry < early_stop_eps) {  // everything behind is invisible: early ray termination
      for (int64_t k = i0 + 64 + lane; k < e; k += 64) mask[k] = 0;
      break;
    }
  }
}
}  // namespace nrhip

extern "C" int nrhip_packed_visibility_from_alpha(const float* alphas, const int64_t* segments, int64_t r,
                                                  float early_stop_eps, float alpha_thre, uint8_t* mask,
                                                  void* stream) {
  NR_REQUIRE(r >= 0, NRHIP_ERR_INVALID_ARG, "packed_visibility: negative r");
  if (r == 0) return NRHIP_OK;
  NR_REQUIRE(alphas && segments && mask, NRHIP_ERR_INVALID_ARG, "packed_visibility: null pointer");
  nrhip::packed_visibility_kernel<<<(int)((r + 3) / 4), 256, 0, (hipStream_t)stream>>>(alphas, segments, r,
                                                                                       early_stop_eps, alpha_thre, mask);
  return nrhip::check_launch("packed_visibility_from_alpha");
}
