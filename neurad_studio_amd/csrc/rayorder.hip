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

}  // namespace nrhip

using namespace nrhip;

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
