// ray generation on the device -- the step immediately before the hot path.
//
//   nrhip_camera_rays  == Cameras._generate_rays_from_coords for PERSPECTIVE cameras without lens distortion, incl. the
//                         rolling-shutter correction of origins and times (nerfstudio/cameras/cameras.py:560-968);
//   nrhip_lidar_rays   == Lidars._generate_rays_from_points (nerfstudio/cameras/lidars.py:399-460).
//
// The reference runs ~40 torch ops per batch for each (index_select of the per-sensor tables, stacks, masked selects,
// norms).  Here one thread produces one ray: the sensor row is a handful of broadcast loads (rays of a 32x32 patch share
// one camera; lidar points of a scan share one lidar), outputs go out as whole [R,3] / [R,1] rows.  Arithmetic mirrors
// torch op for op (separately rounded mul/add/div: -ffp-contract=off) so that directions agree to the ulp -- a ray's
// direction feeds floor() in every hash-grid level downstream.
#include "common.h"

namespace nrhip {

constexpr float kNormEps = 8.8817841970012523e-16f;  // camera_utils._EPS = 4 * finfo(float64).eps, cast to fp32

struct Vec3 {
  float x, y, z;
};

__device__ __forceinline__ Vec3 rotate(const float* __restrict__ m /*3x4 row major*/, float a, float b, float c) {
#pragma clang fp contract(off)
  // torch.sum(d[..., None, :] * R, dim=-1): three separately rounded products, summed left to right
  Vec3 o;
  o.x = (a * m[0] + b * m[1]) + c * m[2];
  o.y = (a * m[4] + b * m[5]) + c * m[6];
  o.z = (a * m[8] + b * m[9]) + c * m[10];
  return o;
}

__device__ __forceinline__ float norm3(Vec3 v) {
#pragma clang fp contract(off)
  return fmaxf(sqrtf((v.x * v.x + v.y * v.y) + v.z * v.z), kNormEps);
}

struct CameraTable {
  const float* c2w;    // [C,3,4]
  const float* fx;     // [C]
  const float* fy;
  const float* cx;
  const float* cy;
  const float* times;  // [C] or NULL
  // rolling shutter (all three or none): cameras.py:937-960
  const float* rs_time;         // [C] rolling_shutter_time
  const float* time_to_center;  // [C]
  const float* velocities;      // [C,3]
  const float* extent;          // [C] image height (vertical shutter) or width (horizontal)
  int rs_mode;                  // 0 none, 1 vertical (rows), 2 horizontal (cols), 3 horizontal reversed
};

__global__ __launch_bounds__(256) void camera_rays_kernel(CameraTable t, const int64_t* __restrict__ cam_idx,
                                                          const float* __restrict__ coords, int64_t n,
                                                          float* __restrict__ origins, float* __restrict__ directions,
                                                          float* __restrict__ pixel_area, float* __restrict__ dir_norm,
                                                          float* __restrict__ times) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t c = cam_idx[i];
  const float y = coords[2 * i], x = coords[2 * i + 1];
  const float fx = t.fx[c], fy = t.fy[c], cx = t.cx[c], cy = t.cy[c];
  // image-plane coordinates and the two neighbours one pixel away (cameras.py:631-633), OpenCV -> OpenGL (:668)
  const float u0 = (x - cx) / fx, u1 = ((x - cx) + 1.f) / fx;
  const float v0 = -((y - cy) / fy), v1 = -(((y - cy) + 1.f) / fy);
  const float* m = t.c2w + 12 * c;
  Vec3 d0 = rotate(m, u0, v0, -1.f), d1 = rotate(m, u1, v0, -1.f), d2 = rotate(m, u0, v1, -1.f);
  const float n0 = norm3(d0), n1 = norm3(d1), n2 = norm3(d2);
  d0.x /= n0, d0.y /= n0, d0.z /= n0;
  d1.x /= n1, d1.y /= n1, d1.z /= n1;
  d2.x /= n2, d2.y /= n2, d2.z /= n2;
  const float ax = d0.x - d1.x, ay = d0.y - d1.y, az = d0.z - d1.z;
  const float bx = d0.x - d2.x, by = d0.y - d2.y, bz = d0.z - d2.z;
  const float dx = sqrtf((ax * ax + ay * ay) + az * az), dy = sqrtf((bx * bx + by * by) + bz * bz);
  float ox = m[3], oy = m[7], oz = m[11];
  float tm = t.times ? t.times[c] : 0.f;
  if (t.rs_mode) {
    const float pos = t.rs_mode == 1 ? y : x;
    float off = (pos / t.extent[c] - 0.5f) * t.rs_time[c] + t.time_to_center[c];
    if (t.rs_mode == 3) off = -off;
    ox = ox + t.velocities[3 * c] * off, oy = oy + t.velocities[3 * c + 1] * off, oz = oz + t.velocities[3 * c + 2] * off;
    tm = tm + off;
  }
  origins[3 * i] = ox, origins[3 * i + 1] = oy, origins[3 * i + 2] = oz;
  directions[3 * i] = d0.x, directions[3 * i + 1] = d0.y, directions[3 * i + 2] = d0.z;
  pixel_area[i] = dx * dy;
  dir_norm[i] = n0;
  if (times) times[i] = tm;
}

struct LidarTable {
  const float* l2w;         // [Ln,3,4]
  const float* times;       // [Ln] or NULL
  const float* velocities;  // [Ln,3] or NULL
  const float* hdiv;        // [Ln] horizontal beam divergence
  const float* vdiv;        // [Ln]
  int ego_compensated;
  float valid_distance;
};

__global__ __launch_bounds__(256) void lidar_rays_kernel(LidarTable t, const int64_t* __restrict__ lidar_idx,
                                                         const float* __restrict__ points, int point_dim, int64_t n,
                                                         float* __restrict__ origins, float* __restrict__ directions,
                                                         float* __restrict__ pixel_area, float* __restrict__ distance,
                                                         uint8_t* __restrict__ did_return, float* __restrict__ times) {
#pragma clang fp contract(off)
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t l = lidar_idx[i];
  const float* p = points + (int64_t)point_dim * i;
  const float* m = t.l2w + 12 * l;
  // transform_points_pairwise (lidars.py:550-564): bmm then + translation
  Vec3 w = rotate(m, p[0], p[1], p[2]);
  w.x = w.x + m[3], w.y = w.y + m[7], w.z = w.z + m[11];
  float ox = m[3], oy = m[7], oz = m[11];
  const float dt = point_dim >= 5 ? p[4] : 0.f;
  if (point_dim >= 5 && t.velocities) {  // motion of the sensor during the sweep (lidars.py:420-426)
    const float vx = dt * t.velocities[3 * l], vy = dt * t.velocities[3 * l + 1], vz = dt * t.velocities[3 * l + 2];
    ox = ox + vx, oy = oy + vy, oz = oz + vz;
    if (!t.ego_compensated) w.x = w.x + vx, w.y = w.y + vy, w.z = w.z + vz;
  }
  Vec3 d{w.x - ox, w.y - oy, w.z - oz};
  const float dist = norm3(d);
  origins[3 * i] = ox, origins[3 * i + 1] = oy, origins[3 * i + 2] = oz;
  directions[3 * i] = d.x / dist, directions[3 * i + 1] = d.y / dist, directions[3 * i + 2] = d.z / dist;
  pixel_area[i] = t.hdiv[l] * t.vdiv[l];
  distance[i] = dist;
  did_return[i] = dist < t.valid_distance ? 1 : 0;
  if (times) times[i] = (t.times ? t.times[l] : 0.f) + dt;
}

// ---- ScaledPatchSampler (data/pixel_samplers.py:618-742): patch centres -> ray indices + ground-truth RGB patches ------
// One workgroup per patch.  The centre is either handed over (the sampling-weights branch draws it with
// torch.multinomial and clips it, :728-742) or derived from the three uniform draws of PixelSampler.sample_method
// (:100-103): (u * float(dim)).long() over (n_images, H - K + 1, W - K + 1), then + K/2 on the pixel axes (:722-726).
// Integer work downstream of that one fp32 product; the image gather is a row-wise copy, K*C contiguous elements per row.
template <typename Pix>
__global__ __launch_bounds__(256) void patch_sample_kernel(const float* __restrict__ uniforms,
                                                           const int64_t* __restrict__ centers, int n_images, int H, int W,
                                                           int C, int Kf, int sc, const int64_t* __restrict__ image_idx,
                                                           const Pix* __restrict__ images, int64_t* __restrict__ ray_indices,
                                                           float* __restrict__ coords, Pix* __restrict__ patches) {
#pragma clang fp contract(off)
  const int64_t p = blockIdx.x;
  const int K = Kf * sc, half = K / 2;
  int64_t c, cy, cx;
  if (centers) {
    c = centers[3 * p], cy = centers[3 * p + 1], cx = centers[3 * p + 2];
  } else {
    c = (int64_t)(uniforms[3 * p] * (float)n_images);
    cy = (int64_t)(uniforms[3 * p + 1] * (float)(H - K + 1)) + half;
    cx = (int64_t)(uniforms[3 * p + 2] * (float)(W - K + 1)) + half;
  }
  const int64_t y0 = cy - half, x0 = cx - half;  // offsets = arange(-(K//2), K//2 + K%2)  (:703)
  const int64_t global_c = image_idx ? image_idx[min(max(c, (int64_t)0), (int64_t)n_images - 1)] : c;
  for (int r = threadIdx.x; r < Kf * Kf; r += 256) {  // ray_indices = rgb_indices[:, sc//2::sc, sc//2::sc]  (:709-712)
    const int64_t y = y0 + sc / 2 + (r / Kf) * sc, x = x0 + sc / 2 + (r % Kf) * sc;
    int64_t* o = ray_indices + 3 * (p * Kf * Kf + r);
    o[0] = global_c, o[1] = y, o[2] = x;
    if (coords) {  // RayGenerator.forward's image_coords[y, x] = pixel centres (ray_generators.py:49, cameras.py:get_image_coords)
      coords[2 * (p * Kf * Kf + r)] = (float)y + 0.5f;
      coords[2 * (p * Kf * Kf + r) + 1] = (float)x + 0.5f;
    }
  }
  if (!patches) return;
  // reads are clamped into the image: a centre the caller placed too close to the border must not fault (the reference's
  // advanced indexing would raise / wrap there)
  const int64_t cc = min(max(c, (int64_t)0), (int64_t)n_images - 1);
  const int row_len = K * C;
  Pix* dst = patches + p * (int64_t)K * row_len;
  for (int e = threadIdx.x; e < K * row_len; e += 256) {
    const int r = e / row_len, q = e - r * row_len;
    const int64_t y = min(max(y0 + r, (int64_t)0), (int64_t)H - 1);
    const int64_t x = min(max(x0 + q / C, (int64_t)0), (int64_t)W - 1);
    dst[e] = images[((cc * H + y) * W + x) * C + q % C];
  }
}

// ---- LidarPointSampler.collate_image_dataset_batch (data/pixel_samplers.py:538-583), packed point clouds ----------------
// Output row r belongs to scan l = shuffle[r / rays_per_lidar] and takes its point floor(draw[l, r % rays_per_lidar] *
// points_per_lidar[l]) (fp64 product, as torch computes it): one thread per row copies the point and writes the
// (lidar_idx[l], point) index pair.  The exclusive prefix sum of the scan sizes (:549-551) is rebuilt per workgroup in LDS.
constexpr int kMaxScans = 2048;

__global__ __launch_bounds__(256) void lidar_point_sample_kernel(const int64_t* __restrict__ shuffle,
                                                                 const double* __restrict__ draws,
                                                                 const int64_t* __restrict__ points_per_lidar,
                                                                 const int64_t* __restrict__ lidar_idx,
                                                                 const float* __restrict__ lidar, int n_lidars,
                                                                 int rays_per_lidar, int D, int64_t n_rays,
                                                                 int64_t* __restrict__ indices, float* __restrict__ points) {
#pragma clang fp contract(off)
  __shared__ int64_t first_point[kMaxScans];
  if (threadIdx.x < 64) {  // one wave: chunked serial scan, n_lidars is tens to hundreds
    int64_t run = 0;
    for (int base = 0; base < n_lidars; base += 64) {
      const int i = base + threadIdx.x;
      int64_t v = i < n_lidars ? points_per_lidar[i] : 0, incl = v;
      for (int d = 1; d < 64; d <<= 1) {
        const int64_t up = __shfl_up(incl, d, 64);
        if ((int)threadIdx.x >= d) incl += up;
      }
      if (i < n_lidars) first_point[i] = run + incl - v;
      run += __shfl(incl, 63, 64);
    }
  }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= n_rays) return;
  const int64_t l = shuffle[r / rays_per_lidar];
  const int64_t k = r % rays_per_lidar;
  const int64_t p = (int64_t)floor(draws[l * rays_per_lidar + k] * (double)points_per_lidar[l]);
  indices[2 * r] = lidar_idx ? lidar_idx[l] : l;
  indices[2 * r + 1] = p;
  const float* src = lidar + (first_point[l] + p) * D;
  for (int d = 0; d < D; ++d) points[r * D + d] = src[d];
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_camera_rays(const nrhip_camera_table* cams, const int64_t* camera_indices, const float* coords,
                                 int64_t n_rays, float* origins, float* directions, float* pixel_area,
                                 float* directions_norm, float* times, void* stream) {
  NR_REQUIRE(cams && n_rays >= 0, NRHIP_ERR_INVALID_ARG, "camera_rays: bad argument");
  if (n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(cams->camera_to_worlds && cams->fx && cams->fy && cams->cx && cams->cy && camera_indices && coords && origins &&
                 directions && pixel_area && directions_norm,
             NRHIP_ERR_INVALID_ARG, "camera_rays: NULL pointer");
  NR_REQUIRE(cams->rolling_shutter >= 0 && cams->rolling_shutter <= 3, NRHIP_ERR_INVALID_ARG, "camera_rays: rolling_shutter mode");
  NR_REQUIRE(cams->rolling_shutter == 0 || (cams->rolling_shutter_time && cams->time_to_center_pixel && cams->velocities &&
                                            cams->shutter_extent && cams->times && times),
             NRHIP_ERR_INVALID_ARG, "camera_rays: rolling shutter needs duration, time_to_center_pixel, velocities, extent, times");
  CameraTable t{cams->camera_to_worlds, cams->fx, cams->fy, cams->cx, cams->cy, cams->times, cams->rolling_shutter_time,
                cams->time_to_center_pixel, cams->velocities, cams->shutter_extent, cams->rolling_shutter};
  camera_rays_kernel<<<grid_for(n_rays, 256), 256, 0, (hipStream_t)stream>>>(t, camera_indices, coords, n_rays, origins,
                                                                            directions, pixel_area, directions_norm, times);
  return check_launch("camera_rays");
}

extern "C" int nrhip_lidar_rays(const nrhip_lidar_table* lidars, const int64_t* lidar_indices, const float* points,
                                int32_t point_dim, int64_t n_rays, float* origins, float* directions, float* pixel_area,
                                float* distance, uint8_t* did_return, float* times, void* stream) {
  NR_REQUIRE(lidars && n_rays >= 0 && point_dim >= 3, NRHIP_ERR_INVALID_ARG, "lidar_rays: bad argument");
  if (n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(lidars->lidar_to_worlds && lidars->horizontal_beam_divergence && lidars->vertical_beam_divergence &&
                 lidar_indices && points && origins && directions && pixel_area && distance && did_return,
             NRHIP_ERR_INVALID_ARG, "lidar_rays: NULL pointer");
  LidarTable t{lidars->lidar_to_worlds, lidars->times, lidars->velocities, lidars->horizontal_beam_divergence,
               lidars->vertical_beam_divergence, lidars->assume_ego_compensated, lidars->valid_lidar_distance_threshold};
  lidar_rays_kernel<<<grid_for(n_rays, 256), 256, 0, (hipStream_t)stream>>>(t, lidar_indices, points, point_dim, n_rays,
                                                                           origins, directions, pixel_area, distance,
                                                                           did_return, times);
  return check_launch("lidar_rays");
}

extern "C" int nrhip_patch_sample(const float* uniforms, const int64_t* centers, int64_t n_patches, int32_t n_images,
                                  int32_t height, int32_t width, int32_t channels, int32_t patch_size, int32_t patch_scale,
                                  const int64_t* image_idx, const void* images, int32_t image_dtype, int64_t* ray_indices,
                                  float* coords, void* patches, void* stream) {
  NR_REQUIRE(n_patches >= 0 && n_images > 0 && channels > 0 && patch_size > 0 && patch_scale > 0, NRHIP_ERR_INVALID_ARG,
             "patch_sample: bad argument");
  const int64_t K = (int64_t)patch_size * patch_scale;
  NR_REQUIRE(K <= height && K <= width, NRHIP_ERR_INVALID_ARG, "patch_sample: the rgb patch (patch_size * patch_scale) exceeds the image");
  NR_REQUIRE(image_dtype == 0 || image_dtype == 1, NRHIP_ERR_UNSUPPORTED, "patch_sample: image_dtype 0 (fp32) or 1 (uint8)");
  if (n_patches == 0) return NRHIP_OK;
  NR_REQUIRE((uniforms != nullptr) != (centers != nullptr), NRHIP_ERR_INVALID_ARG,
             "patch_sample: exactly one of uniforms / centers");
  NR_REQUIRE(ray_indices && (!patches || images), NRHIP_ERR_INVALID_ARG, "patch_sample: NULL pointer");
  NR_REQUIRE(n_patches <= 0x7fffffff, NRHIP_ERR_INVALID_ARG, "patch_sample: too many patches for one launch");
  if (image_dtype == 0)
    patch_sample_kernel<float><<<(int)n_patches, 256, 0, (hipStream_t)stream>>>(
        uniforms, centers, n_images, height, width, channels, patch_size, patch_scale, image_idx, (const float*)images,
        ray_indices, coords, (float*)patches);
  else
    patch_sample_kernel<uint8_t><<<(int)n_patches, 256, 0, (hipStream_t)stream>>>(
        uniforms, centers, n_images, height, width, channels, patch_size, patch_scale, image_idx, (const uint8_t*)images,
        ray_indices, coords, (uint8_t*)patches);
  return check_launch("patch_sample");
}

extern "C" int nrhip_lidar_point_sample(const int64_t* shuffle, const double* draws, const int64_t* points_per_lidar,
                                        const int64_t* lidar_idx, const float* lidar, int32_t n_lidars,
                                        int32_t rays_per_lidar, int32_t point_dim, int64_t n_rays, int64_t* indices,
                                        float* points, void* stream) {
  NR_REQUIRE(n_rays >= 0 && n_lidars > 0 && rays_per_lidar > 0 && point_dim > 0, NRHIP_ERR_INVALID_ARG,
             "lidar_point_sample: bad argument");
  NR_REQUIRE(n_lidars <= kMaxScans, NRHIP_ERR_UNSUPPORTED, "lidar_point_sample: at most 2048 scans per batch");
  NR_REQUIRE(n_rays <= (int64_t)n_lidars * rays_per_lidar, NRHIP_ERR_INVALID_ARG,
             "lidar_point_sample: n_rays exceeds n_lidars * rays_per_lidar draws");
  if (n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(shuffle && draws && points_per_lidar && lidar && indices && points, NRHIP_ERR_INVALID_ARG,
             "lidar_point_sample: NULL pointer");
  lidar_point_sample_kernel<<<grid_for(n_rays, 256), 256, 0, (hipStream_t)stream>>>(
      shuffle, draws, points_per_lidar, lidar_idx, lidar, n_lidars, rays_per_lidar, point_dim, n_rays, indices, points);
  return check_launch("lidar_point_sample");
}
