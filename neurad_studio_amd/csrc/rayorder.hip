// Ray ordering for cache locality (scheduling hint only -- results never depend on it).
//
// The fused kernels walk contiguous ranges of a processing order, one range per XCD (render.hip).  A training batch
// is 32x32 camera patches (already coherent) plus lidar rays drawn at random from whole scans, and eval chunks of lidar
// scans are incoherent too: neighbouring positions of such a batch touch unrelated hash-grid lines, so every XCD's L2
// ends up fetching every level's lines.  nrhip_ray_order computes a permutation that puts rays which look at the same
// region next to each other: key = Morton code (4 or 5 bits per axis) of the CONTRACTED position (the field's own
// ScaledSceneContraction, spatial_distortions.py:103-141) of the point o + d * t_ref on the ray, t_ref = a representative
// distance (default: the contraction boundary).  A counting sort in ONE workgroup: LDS histogram -> scan -> LDS ranks.  Nothing
// is moved: the kernels read rays through `nrhip_rays.order` and write results at the ray's own index, so the
// (unspecified) order inside a bucket never shows in any output.
#include "rayorder.h"

namespace nrhip {

template <int BITS>
__global__ __launch_bounds__(kOrderThreads) void ray_order_kernel(const float* __restrict__ o,
                                                                  const float* __restrict__ d, int64_t n, float t_ref,
                                                                  float scale, int32_t* __restrict__ order) {
  ray_order_body<BITS>(o, d, n, t_ref, scale, order);
}

// ---- the same counting sort over MANY workgroups (round 5) -----------------------------------------------------------------
// The single-workgroup pass costs ~2 us per 1024 rays: nothing at the 4096 rays of a training batch, ~120 us at a 65 536-ray
// eval chunk -- more than the ordered render stage returns there (bench.py --config c4: 0.62 -> 0.69 ms with it).  Three
// small launches instead: keys + a global histogram (integer atomics stay in the L2), one workgroup scans the buckets, every
// ray takes its slot with one more integer atomic.  The order inside a bucket is unspecified here as there.
template <int BITS>
__global__ __launch_bounds__(256) void ray_order_keys_kernel(const float* __restrict__ o, const float* __restrict__ d,
                                                             int64_t n, float t_ref, float scale,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ hist) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = order_key<BITS>(o, d, i, t_ref, scale);
  keys[i] = k;
  atomicAdd(hist + k, 1u);
}

__global__ __launch_bounds__(1024) void ray_order_scan_kernel(uint32_t* __restrict__ hist, int buckets) {
  __shared__ uint32_t part[1024];
  const int per = (buckets + 1023) / 1024, c0 = threadIdx.x * per;
  uint32_t s = 0;
  for (int k = 0; k < per; ++k)
    if (c0 + k < buckets) s += hist[c0 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for (int k = 0; k < per; ++k)
    if (c0 + k < buckets) {
      const uint32_t v = hist[c0 + k];
      hist[c0 + k] = run;
      run += v;
    }
}

__global__ __launch_bounds__(256) void ray_order_place_kernel(const uint32_t* __restrict__ keys, int64_t n,
                                                              uint32_t* __restrict__ hist, int32_t* __restrict__ order) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  order[atomicAdd(hist + keys[i], 1u)] = (int32_t)i;
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_ray_order_workspace(int64_t n_rays, int32_t key_bits, int64_t* bytes) {
  NR_REQUIRE(bytes && n_rays >= 0 && (key_bits == 0 || key_bits == 4 || key_bits == 5), NRHIP_ERR_INVALID_ARG,
             "ray_order_workspace: bad argument");
  *bytes = (n_rays + ((int64_t)1 << (3 * (key_bits == 5 ? 5 : 4)))) * (int64_t)sizeof(uint32_t);
  return NRHIP_OK;
}

extern "C" int nrhip_ray_order_large(const float* origins, const float* directions, int64_t n_rays, float t_ref,
                                     float static_scale, int32_t key_bits, void* workspace, int64_t workspace_bytes,
                                     int32_t* order, void* stream) {
  NR_REQUIRE(n_rays >= 0 && n_rays < (INT64_C(1) << 31), NRHIP_ERR_INVALID_ARG, "ray_order_large: n_rays %lld out of range",
             (long long)n_rays);
  if (n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(origins && directions && order && workspace, NRHIP_ERR_INVALID_ARG, "ray_order_large: NULL pointer");
  NR_REQUIRE(static_scale > 0.f && t_ref >= 0.f, NRHIP_ERR_INVALID_ARG, "ray_order_large: scale must be > 0 and t_ref >= 0");
  NR_REQUIRE(key_bits == 0 || key_bits == 4 || key_bits == 5, NRHIP_ERR_INVALID_ARG,
             "ray_order_large: key_bits %d not in {0 (default = 4), 4, 5}", key_bits);
  const int bits = key_bits == 5 ? 5 : 4, buckets = 1 << (3 * bits);
  NR_REQUIRE(workspace_bytes >= (n_rays + buckets) * (int64_t)sizeof(uint32_t), NRHIP_ERR_INVALID_ARG,
             "ray_order_large: workspace of %lld bytes, need %lld", (long long)workspace_bytes,
             (long long)((n_rays + buckets) * (int64_t)sizeof(uint32_t)));
  const hipStream_t st = (hipStream_t)stream;
  uint32_t* hist = static_cast<uint32_t*>(workspace);
  uint32_t* keys = hist + buckets;
  if (hipMemsetAsync(hist, 0, (size_t)buckets * sizeof(uint32_t), st) != hipSuccess) return check_launch("ray_order_large");
  const int blocks = grid_for(n_rays, 256);
  if (bits == 5)
    ray_order_keys_kernel<5><<<blocks, 256, 0, st>>>(origins, directions, n_rays, t_ref, static_scale, keys, hist);
  else
    ray_order_keys_kernel<4><<<blocks, 256, 0, st>>>(origins, directions, n_rays, t_ref, static_scale, keys, hist);
  ray_order_scan_kernel<<<1, 1024, 0, st>>>(hist, buckets);
  ray_order_place_kernel<<<blocks, 256, 0, st>>>(keys, n_rays, hist, order);
  return check_launch("ray_order_large");
}

extern "C" int nrhip_ray_order(const float* origins, const float* directions, int64_t n_rays, float t_ref,
                               float static_scale, int32_t key_bits, int32_t* order, void* stream) {
  NR_REQUIRE(n_rays >= 0 && n_rays < (INT64_C(1) << 31), NRHIP_ERR_INVALID_ARG, "ray_order: n_rays %lld out of range",
             (long long)n_rays);
  if (n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(origins && directions && order, NRHIP_ERR_INVALID_ARG, "ray_order: NULL pointer");
  NR_REQUIRE(static_scale > 0.f && t_ref >= 0.f, NRHIP_ERR_INVALID_ARG, "ray_order: scale must be > 0 and t_ref >= 0");
  NR_REQUIRE(key_bits == 0 || key_bits == 4 || key_bits == 5, NRHIP_ERR_INVALID_ARG,
             "ray_order: key_bits %d not in {0 (default = 4), 4, 5}", key_bits);
  const hipStream_t st = (hipStream_t)stream;
  if (key_bits == 5)
    ray_order_kernel<5><<<1, kOrderThreads, 0, st>>>(origins, directions, n_rays, t_ref, static_scale, order);
  else
    ray_order_kernel<4><<<1, kOrderThreads, 0, st>>>(origins, directions, n_rays, t_ref, static_scale, order);
  return check_launch("ray_order");
}
