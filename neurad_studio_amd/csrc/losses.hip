// SURVEY §8(f) row 2 -- the losses that consume the sampler outputs every training step:
//   zipnerf_interlevel_loss (model_components/losses.py:645-705) and distortion_loss (losses.py:137-156).
// The reference evaluates each with ~40 small torch kernels per proposal level (sort, two cumsums, searchsorted,
// five take_along_dims ...) on [R, 68] / [R, 129] tensors.  Here one wavefront owns one ray: the 2(Sf+1) blurred
// step-function knots never leave LDS, the "sort" is a merge of two already sorted sequences (c - r and c + r),
// the three prefix sums are wave scans, and the gradient w.r.t. the proposal weights (the only input that carries
// one: the fine histogram is detached, losses.py:678-679) is produced in the same pass.
// The prefix sums accumulate in fp64 and round every prefix to fp32, exactly what torch's CPU cumsum does for fp32
// (at::acc_type<float> = double); a plain fp32 scan lands 1e-4 away after the division by (w_p + 1e-5).
#include "common.h"
#include "wave_scan.h"

namespace nrhip {
namespace {

constexpr int kMaxFine = 128;                   // fine (field) samples per ray
constexpr int kMaxProp = 512;                   // proposal samples per ray
constexpr int kKnots = 2 * (kMaxFine + 1) + 2;  // blurred knots incl. the 0 / 1 padding
constexpr int kRaysPerBlock = 4;

// One ray's scratch, carved out of dynamic LDS for the launch's actual sample counts: sized for the maxima (6.2 KB per ray)
// the four rays of a workgroup held the CU at 6 workgroups = 24 of its 32 waves; NeuRAD's 32 fine / 128 proposal samples
// need 1.6 KB.
struct RayLds {
  float* c;    // [sf + 1]
  float* wn;   // [sf + 1]      w / (c[i+1] - c[i]), then y1
  float* x;    // [2 sf + 4]    c_  (padded knots)
  float* y;    // [2 sf + 4]    w_  (padded blurred pdf)
  float* cdf;  // [2 sf + 4]    padded cdf
  float* v;    // [sp + 1]      cdf interpolated at the proposal edges
};
__host__ __device__ constexpr int ray_lds_floats(int sf, int sp) { return 2 * (sf + 1) + 3 * (2 * sf + 4) + (sp + 1); }

// wave-wide sums / scans on DPP row shifts + readlane (wave_scan.h): the fp64 scans below were six dependent pairs of
// ds_bpermute each as shuffles, three times per ray and chunk
__device__ __forceinline__ double wave_sum_d(double v) { return wscan::reduce<wscan::Add>(v); }
__device__ __forceinline__ float wave_sum_f(float v) { return wscan::reduce<wscan::Add>(v); }
// inclusive prefix sum of arr[0..n) in place: fp64 accumulation, every prefix rounded to fp32 (see header)
__device__ __forceinline__ void scan_inplace(float* arr, int n, int lane) {
  double carry = 0.0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    double v = wscan::incl<wscan::Add>(i < n ? (double)arr[i] : 0.0, lane);
    v += carry;
    if (i < n) arr[i] = (float)v;
    carry = wscan::last(v);
  }
}
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

__global__ __launch_bounds__(64 * kRaysPerBlock) void interlevel_loss_kernel(
    const float* __restrict__ c_all, const float* __restrict__ w_all, int sf, const float* __restrict__ cp_all,
    const float* __restrict__ wp_all, int sp, float r, int64_t R, float* __restrict__ loss, float* __restrict__ gwp) {
  extern __shared__ float lds_dyn[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wid;
  if (ray >= R) return;
  RayLds L;
  L.c = lds_dyn + wid * ray_lds_floats(sf, sp);
  L.wn = L.c + (sf + 1);
  L.x = L.wn + (sf + 1);
  L.y = L.x + (2 * sf + 4);
  L.cdf = L.y + (2 * sf + 4);
  L.v = L.cdf + (2 * sf + 4);
  const float* c = c_all + ray * (sf + 1);
  const float* w = w_all + ray * sf;
  // ---- fine histogram: last weight absorbs the missing mass, heights = weight / width (losses.py:680-683) ------
  float acc = 0.f;
  for (int i = lane; i < sf; i += 64) acc += w[i];
  acc = wave_sum_f(acc);
  for (int i = lane; i <= sf; i += 64) L.c[i] = c[i];
  wave_fence();
  for (int i = lane; i < sf; i += 64) {
    const float wi = w[i] + (i == sf - 1 ? 1.f - acc : 0.f);
    L.wn[i] = wi / (L.c[i + 1] - L.c[i]);
  }
  if (lane == 0) L.wn[sf] = 0.f;
  wave_fence();
  // ---- _blur_stepfun (losses.py:645-653): knots = merge of (c - r) and (c + r); y2 = +-y1 in knot order ---------
  const int n1 = sf + 1, M = 2 * n1;
  float* xr = L.x + 1;   // knot m of the merge at xr[m]; x[0] / x[M+1] are the 0 / 1 padding
  float* y2 = L.y + 1;   // y2[m], later yr[m] shifted by one
  for (int k = lane; k < n1; k += 64) {
    const float y1 = (L.wn[k] - (k > 0 ? L.wn[k - 1] : 0.f)) / (2.f * r);
    const float a = L.c[k] - r, b = L.c[k] + r;
    int pa = k, pb = k;  // stable merge: ties keep the (c - r) element first
    for (int q = 0; q < n1; ++q) {
      pa += (L.c[q] + r < a) ? 1 : 0;
      pb += (L.c[q] - r <= b) ? 1 : 0;
    }
    xr[pa] = a, xr[pb] = b;
    L.cdf[pa] = y1, L.cdf[pb] = -y1;  // cdf[] is free until the third scan: y2 in knot order
  }
  wave_fence();
  // inner cumsum over y2[0..M-2]
  scan_inplace(L.cdf, M - 1, lane);
  wave_fence();
  // yr_inc[m] = cumsum((xr[m+1] - xr[m]) * inner[m]), clamp_min(0); yr = [0, yr_inc]
  for (int m = lane; m < M - 1; m += 64) y2[m + 1] = (xr[m + 1] - xr[m]) * L.cdf[m];
  wave_fence();
  scan_inplace(y2 + 1, M - 1, lane);
  wave_fence();
  for (int m = lane; m < M - 1; m += 64) y2[m + 1] = fmaxf(y2[m + 1], 0.f);
  if (lane == 0) y2[0] = 0.f;  // yr[0]
  wave_fence();
  // ---- piecewise-linear pdf -> piecewise-quadratic cdf (losses.py:691-693), then the 0 / 1 padding (695-698) -----
  float* cdf = L.cdf + 1;
  for (int m = lane; m < M - 1; m += 64) cdf[m + 1] = 0.5f * (y2[m + 1] + y2[m]) * (xr[m + 1] - xr[m]);
  wave_fence();
  scan_inplace(cdf + 1, M - 1, lane);
  if (lane == 0) {
    cdf[0] = 0.f;
    L.x[0] = 0.f, L.y[0] = 0.f, L.cdf[0] = 0.f;
    L.x[M + 1] = 1.f, L.y[M + 1] = 0.f, L.cdf[M + 1] = 1.f;
  }
  wave_fence();
  // ---- _sorted_interp_quad at the proposal edges (losses.py:656-669) ------------------------------------------
  const int K = M + 2;
  const float* cp = cp_all + ray * (sp + 1);
  for (int k = lane; k <= sp; k += 64) {
    const float x = cp[k];
    int lo = 0, hi = K;  // first index with c_[idx] >= x  (torch.searchsorted, right=False)
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (L.x[mid] < x) lo = mid + 1; else hi = mid;
    }
    const int left = lo > 0 ? lo - 1 : 0, right = lo < K - 1 ? lo : K - 1;
    const float xp0 = L.x[left], xp1 = L.x[right], f0 = L.y[left], f1 = L.y[right], c0 = L.cdf[left];
    float off = (x - xp0) / (xp1 - xp0);
    off = (off != off) ? 0.f : fminf(fmaxf(off, 0.f), 1.f);  // nan_to_num(nan -> 0), clip to [0, 1] (inf -> 1)
    L.v[k] = c0 + (x - xp0) * (f0 + f1 * off + f0 * (1.f - off)) * 0.5f;
  }
  wave_fence();
  // ---- loss and its gradient w.r.t. the proposal weights (losses.py:703-704) -----------------------------------
  const float* wp = wp_all + ray * sp;
  float part = 0.f;
  for (int k = lane; k < sp; k += 64) {
    const float ws = L.v[k + 1] - L.v[k];
    const float p = wp[k], den = p + 1e-5f;
    const float d = fmaxf(ws - p, 0.f);
    part += d * d / den;
    if (gwp) gwp[ray * sp + k] = -(2.f * d / den + d * d / (den * den));
  }
  part = wave_sum_f(part);
  if (lane == 0) loss[ray] = part;
}

// lossfun_distortion (losses.py:137-148): loss = sum_i w_i sum_j w_j |u_i - u_j| + sum_i w_i^2 delta_i / 3
__global__ __launch_bounds__(64 * kRaysPerBlock) void distortion_loss_kernel(const float* __restrict__ c_all,
                                                                             const float* __restrict__ w_all, int s,
                                                                             int64_t R, float* __restrict__ loss,
                                                                             float* __restrict__ gw) {
  __shared__ float u_all[kRaysPerBlock][kMaxProp], w_lds[kRaysPerBlock][kMaxProp];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wid;
  if (ray >= R) return;
  const float* c = c_all + ray * (s + 1);
  const float* w = w_all + ray * s;
  float* u = u_all[wid];
  float* wl = w_lds[wid];
  for (int i = lane; i < s; i += 64) u[i] = (c[i + 1] + c[i]) / 2.f, wl[i] = w[i];
  wave_fence();
  float part = 0.f;
  for (int i = lane; i < s; i += 64) {
    float inner = 0.f;
    for (int j = 0; j < s; ++j) inner += wl[j] * fabsf(u[i] - u[j]);
    const float wi = wl[i], delta = c[i + 1] - c[i];
    part += wi * inner + wi * wi * delta / 3.f;
    if (gw) gw[ray * s + i] = 2.f * inner + (2.f / 3.f) * wi * delta;
  }
  part = wave_sum_f(part);
  if (lane == 0) loss[ray] = part;
}

}  // namespace

// ---- carving: NeuRADModel._compute_is_close_to_lidar (models/neurad.py:677-700) and the proposal
// carving term sum((w * (is_lidar & ~close))^2) (models/neurad.py:399-408) ------------------------------------------------
// The reference spends ~7 elementwise ops per sample level on the mask and ~6 more (with autograd) on each loss term; here
// one pass per level writes the mask and, when weights are given, the per-ray loss and its gradient 2 w far.
__global__ __launch_bounds__(256) void lidar_carving_kernel(const float* __restrict__ starts,
                                                            const float* __restrict__ ends, int stride,
                                                            const float* __restrict__ weights,
                                                            const uint8_t* __restrict__ is_lidar,
                                                            const uint8_t* __restrict__ did_return,
                                                            const float* __restrict__ dist, float eps, float non_return,
                                                            int64_t R, int S, uint8_t* __restrict__ is_close,
                                                            float* __restrict__ loss_ray, float* __restrict__ grad_w) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= R) return;
  const bool lidar = is_lidar[ray] != 0;
  const bool returned = did_return ? did_return[ray] != 0 : true;
  const float dr = dist[ray];
  float acc = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float mid = (starts[ray * stride + s] + ends[ray * stride + s]) * 0.5f;
    const bool close = lidar && (returned ? fabsf(dr - mid) < eps : mid < non_return);
    if (is_close) is_close[ray * S + s] = close ? 1 : 0;
    if (weights) {
      const float wf = (lidar && !close) ? weights[ray * S + s] : 0.f;
      acc += wf * wf;
      if (grad_w) grad_w[ray * S + s] = 2.f * wf;
    }
  }
  if (loss_ray) {
    acc = wave_sum_f(acc);
    if (lane == 0) loss_ray[ray] = acc;
  }
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_interlevel_loss(const float* c, const float* w, int32_t n_fine, const float* cp, const float* wp,
                                     int32_t n_prop, float pulse_width, int64_t r, float* loss_per_ray, float* grad_wp,
                                     void* stream) {
  NR_REQUIRE(r >= 0 && n_fine >= 1 && n_prop >= 1 && pulse_width > 0.f, NRHIP_ERR_INVALID_ARG,
             "interlevel_loss: bad argument");
  NR_REQUIRE(n_fine <= kMaxFine && n_prop <= kMaxProp, NRHIP_ERR_UNSUPPORTED,
             "interlevel_loss: %d fine / %d proposal samples per ray exceed %d / %d", n_fine, n_prop, kMaxFine, kMaxProp);
  if (r == 0) return NRHIP_OK;
  NR_REQUIRE(c && w && cp && wp && loss_per_ray, NRHIP_ERR_INVALID_ARG, "interlevel_loss: null pointer");
  interlevel_loss_kernel<<<(unsigned)((r + kRaysPerBlock - 1) / kRaysPerBlock), 64 * kRaysPerBlock,
                           (size_t)kRaysPerBlock * ray_lds_floats(n_fine, n_prop) * sizeof(float), (hipStream_t)stream>>>(c, w, n_fine, cp, wp, n_prop, pulse_width, r, loss_per_ray, grad_wp);
  return check_launch("interlevel_loss");
}

extern "C" int nrhip_distortion_loss(const float* c, const float* w, int32_t n_samples, int64_t r, float* loss_per_ray,
                                     float* grad_w, void* stream) {
  NR_REQUIRE(r >= 0 && n_samples >= 1, NRHIP_ERR_INVALID_ARG, "distortion_loss: bad argument");
  NR_REQUIRE(n_samples <= kMaxProp, NRHIP_ERR_UNSUPPORTED, "distortion_loss: %d samples per ray exceed %d", n_samples,
             kMaxProp);
  if (r == 0) return NRHIP_OK;
  NR_REQUIRE(c && w && loss_per_ray, NRHIP_ERR_INVALID_ARG, "distortion_loss: null pointer");
  distortion_loss_kernel<<<(unsigned)((r + kRaysPerBlock - 1) / kRaysPerBlock), 64 * kRaysPerBlock, 0,
                           (hipStream_t)stream>>>(c, w, n_samples, r, loss_per_ray, grad_w);
  return check_launch("distortion_loss");
}

extern "C" int nrhip_lidar_carving(const float* starts, const float* ends, int32_t sample_stride, const float* weights,
                                   const uint8_t* is_lidar, const uint8_t* did_return, const float* distance,
                                   float carving_epsilon, float non_return_lidar_distance, int64_t r, int32_t s,
                                   uint8_t* is_close, float* loss_per_ray, float* grad_weights, void* stream) {
  NR_REQUIRE(starts && ends && is_lidar && distance && r >= 0 && s >= 0 && (is_close || loss_per_ray || grad_weights),
             NRHIP_ERR_INVALID_ARG, "lidar_carving: bad argument");
  NR_REQUIRE(weights || (!loss_per_ray && !grad_weights), NRHIP_ERR_INVALID_ARG, "lidar_carving: the loss needs the weights");
  if (r == 0) return NRHIP_OK;
  nrhip::lidar_carving_kernel<<<(int)((r + 3) / 4), 256, 0, (hipStream_t)stream>>>(
      starts, ends, sample_stride > 0 ? sample_stride : s, weights, is_lidar, did_return, distance, carving_epsilon,
      non_return_lidar_distance, r, s, is_close, loss_per_ray, grad_weights);
  return nrhip::check_launch("lidar_carving");
}
