// C1 (nerfacc dense-mode transmittance/weights), S3 (RaySamples.get_weights) and C2 (NeuRAD compositing).
// One 64-lane wavefront owns one ray: the exclusive product / sum scan over the ray's samples runs as a
// wave-level shuffle scan in chunks of 64 samples with a carried transmittance, reductions are wave
// shuffles.  Nothing here needs LDS or atomics.
#include "common.h"
#include "wave_scan.h"

namespace nrhip {

constexpr int kRaysPerBlock = 4;  // 4 waves / block

// wave-wide scans on DPP row shifts + readlane (wave_scan.h), not on ds_bpermute shuffles
__device__ __forceinline__ float wave_incl_scan_mul(float v, int lane) { return wscan::incl<wscan::Mul>(v, lane); }
__device__ __forceinline__ float wave_incl_scan_add(float v, int lane) { return wscan::incl<wscan::Add>(v, lane); }
__device__ __forceinline__ float wave_sum(float v) { return wscan::reduce<wscan::Add>(v); }
// suffix (reverse) inclusive scan
__device__ __forceinline__ float wave_incl_rscan_add(float v, int lane) { return wscan::rincl<wscan::Add>(v, lane); }

__device__ __forceinline__ float nan_to_num(float v) {
  if (v != v) return 0.f;
  if (v == INFINITY) return 3.4028234663852886e38f;
  if (v == -INFINITY) return -3.4028234663852886e38f;
  return v;
}

// MODE 0: alphas given.  MODE 1: sigmas + (t_ends - t_starts).  MODE 2: densities * deltas with nan_to_num (S3).
template <int MODE>
__global__ __launch_bounds__(64 * kRaysPerBlock) void weights_fwd_kernel(const float* __restrict__ a,
                                                                         const float* __restrict__ b,
                                                                         const float* __restrict__ c, int64_t R,
                                                                         int S, float* __restrict__ weights,
                                                                         float* __restrict__ trans,
                                                                         float* __restrict__ alphas_out) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  float carry = (MODE == 0) ? 1.f : 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool live = s < S;
    const int64_t i = ray * S + s;
    float alpha, step;  // step = (1-alpha) [MODE 0] or sigma*delta [MODE 1,2]
    if (MODE == 0) {
      alpha = live ? a[i] : 0.f;
      step = 1.f - alpha;
    } else if (MODE == 1) {
      step = live ? c[i] * (b[i] - a[i]) : 0.f;  // a=t_starts b=t_ends c=sigmas
      alpha = 1.f - expf(-step);
    } else {
      step = live ? a[i] * b[i] : 0.f;  // a=deltas b=densities
      alpha = 1.f - expf(-step);
    }
    float T;
    if (MODE == 0) {
      const float incl = wave_incl_scan_mul(step, lane);
      const float excl = wscan::shift_up1(incl, 1.f, lane);
      T = carry * excl;
      carry *= wscan::last(incl);
    } else {
      const float incl = wave_incl_scan_add(step, lane);
      const float excl = wscan::shift_up1(incl, 0.f, lane);
      T = expf(-(carry + excl));
      carry += wscan::last(incl);
    }
    if (live) {
      float w = alpha * T;
      if (MODE == 2) w = nan_to_num(w);
      weights[i] = w;
      if (trans) trans[i] = T;
      if (alphas_out) alphas_out[i] = alpha;
    }
  }
}

// Backward of the three modes.  gw = dL/dweights, gt = dL/dtrans (may be NULL).
//   alpha mode:   dα_i  = gw_i T_i - (Σ_{k>i} G_k T_k) / (1-α_i),  G_k = gw_k α_k + gt_k
//   density mode: dsd_i = gw_i T_i e^{-sd_i} - Σ_{k>i} (gw_k w_k + gt_k T_k);  dσ_i = dsd_i δ_i
// The suffix sums run as a reverse wave scan over chunks processed last-to-first.
template <int MODE>
__global__ __launch_bounds__(64 * kRaysPerBlock) void weights_bwd_kernel(const float* __restrict__ a,
                                                                         const float* __restrict__ b,
                                                                         const float* __restrict__ c,
                                                                         const float* __restrict__ gw,
                                                                         const float* __restrict__ gt, int64_t R,
                                                                         int S, float* __restrict__ gout) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  // pass 1: per-chunk carry-in of the forward scan (recomputed, S is small)
  const int nchunk = (S + 63) / 64;
  float suffix = 0.f;  // Σ over samples in later chunks
  for (int ch = nchunk - 1; ch >= 0; --ch) {
    // forward carry up to this chunk
    float carry = (MODE == 0) ? 1.f : 0.f;
    for (int p = 0; p < ch; ++p) {
      const int s = p * 64 + lane;
      const int64_t i = ray * S + s;
      float step;
      if (MODE == 0) step = 1.f - a[i];
      else if (MODE == 1) step = c[i] * (b[i] - a[i]);
      else step = a[i] * b[i];
      if (MODE == 0) {
        carry *= wscan::reduce<wscan::Mul>(step);
      } else {
        carry += wave_sum(step);
      }
    }
    const int s = ch * 64 + lane;
    const bool live = s < S;
    const int64_t i = ray * S + s;
    float alpha, step, delta = 1.f;
    if (MODE == 0) {
      alpha = live ? a[i] : 0.f;
      step = 1.f - alpha;
    } else if (MODE == 1) {
      delta = live ? (b[i] - a[i]) : 0.f;
      step = live ? c[i] * delta : 0.f;
      alpha = 1.f - expf(-step);
    } else {
      delta = live ? a[i] : 0.f;
      step = live ? a[i] * b[i] : 0.f;
      alpha = 1.f - expf(-step);
    }
    float T;
    if (MODE == 0) {
      const float incl = wave_incl_scan_mul(step, lane);
      const float excl = wscan::shift_up1(incl, 1.f, lane);
      T = carry * excl;
    } else {
      const float incl = wave_incl_scan_add(step, lane);
      const float excl = wscan::shift_up1(incl, 0.f, lane);
      T = expf(-(carry + excl));
    }
    const float gwi = live ? gw[i] : 0.f;
    const float gti = (live && gt) ? gt[i] : 0.f;
    const float term = (gwi * alpha + gti) * T;  // G_k T_k  (== gw_k w_k + gt_k T_k)
    const float incl_r = wave_incl_rscan_add(term, lane);
    const float after = incl_r - term + suffix;  // Σ_{k>i}
    if (live) {
      float g;
      if (MODE == 0) {
        g = gwi * T - after / fmaxf(1.f - alpha, 1e-10f);
      } else {
        g = (gwi * T * expf(-step) - after) * delta;
      }
      gout[i] = g;
    }
    suffix += wscan::first(incl_r);
  }
}

// C3: appearance embedding (models/neurad.py:423-441): out[r] = E[lo[r]] (1 - f[r]) + E[hi[r]] f[r]  (hi == NULL: a plain
// lookup).  Backward: a few dozen embedding rows receive all R gradients -- torch's embedding_dense_backward sorts the
// indices for that (2 x 150 us per step at 57 344 rays); here every workgroup sums its rays into an LDS image of the
// table and adds the image once (tables beyond 32 KB go straight to global atomics).
constexpr int kEmbedLdsFloats = 8192;
__device__ __forceinline__ int64_t clamp_row(int64_t i, int E) { return i < 0 ? 0 : (i >= E ? E - 1 : i); }

__global__ __launch_bounds__(256) void embedding_lerp_fwd_kernel(const float* __restrict__ wgt,
                                                                 const int64_t* __restrict__ lo,
                                                                 const int64_t* __restrict__ hi,
                                                                 const float* __restrict__ frac, int64_t R, int E, int D,
                                                                 float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * D) return;
  const int64_t r = i / D;
  const int d = (int)(i - r * D);
  // indices are clamped into the table: a bad sensor / time index must not read (or, in the backward, add) out of bounds
  const float a = wgt[clamp_row(lo[r], E) * D + d];
  if (!hi) {
    out[i] = a;
    return;
  }
  const float f = frac[r];
  out[i] = a * (1.f - f) + wgt[clamp_row(hi[r], E) * D + d] * f;  // the reference's e_lo * (1 - frac) + e_hi * frac
}

__global__ __launch_bounds__(256) void embedding_lerp_bwd_kernel(const float* __restrict__ g,
                                                                 const int64_t* __restrict__ lo,
                                                                 const int64_t* __restrict__ hi,
                                                                 const float* __restrict__ frac, int64_t R, int E, int D,
                                                                 int in_lds, float* __restrict__ gw) {
  extern __shared__ __attribute__((aligned(16))) float img[];
  const int cells = E * D;
  if (in_lds) {
    for (int k = threadIdx.x; k < cells; k += 256) img[k] = 0.f;
    __syncthreads();
  }
  float* acc = in_lds ? img : gw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < R * D; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    const float gv = g[i];
    if (!hi) {
      atomicAdd(acc + clamp_row(lo[r], E) * D + d, gv);
    } else {
      const float f = frac[r];
      atomicAdd(acc + clamp_row(lo[r], E) * D + d, gv * (1.f - f));
      atomicAdd(acc + clamp_row(hi[r], E) * D + d, gv * f);
    }
  }
  if (in_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < cells; k += 256)
      if (img[k] != 0.f) atomicAdd(gw + k, img[k]);
  }
}

// backward of accumulate_along_rays with values: gw[r,s] = Σ_c g[r,c] v[r,s,c];  gv[r,s,c] = w[r,s] g[r,c].  One wave per
// ray walks the [S*C] row flat (coalesced); torch autograd materialises two [R,S,C] products for this.
__global__ __launch_bounds__(64 * kRaysPerBlock) void accumulate_bwd_kernel(const float* __restrict__ w,
                                                                            const float* __restrict__ v,
                                                                            const float* __restrict__ g, int64_t R, int S,
                                                                            int C, float* __restrict__ gw,
                                                                            float* __restrict__ gv) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float* wr = w + ray * S;
  const float* vr = v + ray * (int64_t)S * C;
  const float* gr = g + ray * C;
  if (C <= 64 && (64 % C) == 0) {
    // a lane always sees the same channel; the C lanes of one sample are neighbours -> xor-reduce them for gw
    const float gc = gr[lane % C];
    const int total = S * C;
    for (int i0 = 0; i0 < total; i0 += 64) {
      const int i = i0 + lane;
      const bool live = i < total;
      const int s = live ? i / C : 0;
      float part = live ? gc * vr[i] : 0.f;
      if (gv && live) gv[ray * (int64_t)total + i] = wr[s] * gc;
      if (gw) {
        for (int off = C >> 1; off > 0; off >>= 1) part += __shfl_xor(part, off, 64);
        if (live && (lane % C) == 0) gw[ray * S + s] = part;
      }
    }
  } else {
    for (int s = lane; s < S; s += 64) {
      float acc = 0.f;
      const float ws = wr[s];
      for (int ch = 0; ch < C; ++ch) {
        acc += gr[ch] * vr[(int64_t)s * C + ch];
        if (gv) gv[(ray * S + s) * (int64_t)C + ch] = ws * gr[ch];
      }
      if (gw) gw[ray * S + s] = acc;
    }
  }
}

// accumulate_along_rays (dense): out[r,c] = Σ_s w[r,s] * v[r,s,c]   (values==NULL -> Σ_s w)
__global__ __launch_bounds__(64 * kRaysPerBlock) void accumulate_kernel(const float* __restrict__ w,
                                                                        const float* __restrict__ v, int64_t R,
                                                                        int S, int C, float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float* wr = w + ray * S;
  if (!v) {
    float acc = 0.f;
    for (int s = lane; s < S; s += 64) acc += wr[s];
    acc = wave_sum(acc);
    if (lane == 0) out[ray] = acc;
    return;
  }
  const float* vr = v + ray * (int64_t)S * C;
  if (C <= 64 && (64 % C) == 0) {
    // flat walk over [S*C]: a lane always sees the same channel
    float acc = 0.f;
    const int total = S * C;
    for (int i = lane; i < total; i += 64) acc += wr[i / C] * vr[i];
    for (int off = 32; off >= C; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane < C) out[ray * C + lane] = acc;
  } else {
    for (int ch = lane; ch < C; ch += 64) {
      float acc = 0.f;
      for (int s = 0; s < S; ++s) acc += wr[s] * vr[(int64_t)s * C + ch];
      out[ray * C + ch] = acc;
    }
  }
}

// C2 forward: acc = Σw ; sky residual on the last sample ; features over all S ; depth over S-1.
__global__ __launch_bounds__(64 * kRaysPerBlock) void composite_fwd_kernel(
    const float* __restrict__ w, const float* __restrict__ f, const float* __restrict__ starts,
    const float* __restrict__ ends, int64_t R, int S, int C, float* __restrict__ of, float* __restrict__ od,
    float* __restrict__ oa) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float* wr = w + ray * S;
  float acc = 0.f, depth = 0.f;
  for (int s = lane; s < S; s += 64) {
    const float ws = wr[s];
    acc += ws;
    if (s < S - 1) depth += ws * ((starts[ray * S + s] + ends[ray * S + s]) / 2.f);
  }
  acc = wave_sum(acc);
  depth = wave_sum(depth);
  if (lane == 0) {
    oa[ray] = acc;
    od[ray] = depth;
  }
  const float resid = 1.f - acc;
  const float* fr = f + ray * (int64_t)S * C;
  if (C <= 64 && (64 % C) == 0) {
    float a2 = 0.f;
    const int total = S * C;
    for (int i = lane; i < total; i += 64) {
      const int s = i / C;
      const float ws = wr[s] + (s == S - 1 ? resid : 0.f);
      a2 += ws * fr[i];
    }
    for (int off = 32; off >= C; off >>= 1) a2 += __shfl_xor(a2, off, 64);
    if (lane < C) of[ray * C + lane] = a2;
  } else {
    for (int ch = lane; ch < C; ch += 64) {
      float a2 = 0.f;
      for (int s = 0; s < S; ++s) a2 += (wr[s] + (s == S - 1 ? resid : 0.f)) * fr[(int64_t)s * C + ch];
      of[ray * C + ch] = a2;
    }
  }
}

// C2 backward:  df_sc = w2_s gF_c ;  dw_s = q_s - q_{S-1} + g_acc + g_depth*mid_s[s<S-1],  q_s = Σ_c gF_c f_sc
__global__ __launch_bounds__(64 * kRaysPerBlock) void composite_bwd_kernel(
    const float* __restrict__ w, const float* __restrict__ f, const float* __restrict__ starts,
    const float* __restrict__ ends, const float* __restrict__ gF, const float* __restrict__ gD,
    const float* __restrict__ gA, int64_t R, int S, int C, float* __restrict__ gw, float* __restrict__ gf) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float* wr = w + ray * S;
  const float* fr = f + ray * (int64_t)S * C;
  const float* gFr = gF + ray * C;
  float acc = 0.f;
  for (int s = lane; s < S; s += 64) acc += wr[s];
  acc = wave_sum(acc);
  const float resid = 1.f - acc;
  // q_{S-1}
  float qlast = 0.f;
  for (int ch = lane; ch < C; ch += 64) qlast += gFr[ch] * fr[(int64_t)(S - 1) * C + ch];
  qlast = wave_sum(qlast);
  const float gacc = gA ? gA[ray] : 0.f, gdep = gD ? gD[ray] : 0.f;
  // vector path: LP = C/4 lanes share a sample, each owns one float4 of the channel row -> every load/store of a
  // wave is a run of full 16-byte pieces (a lane-per-sample loop walks 64 separate lines 4 bytes at a time)
  const int LP = C >> 2;
  if ((C & 3) == 0 && LP <= 64 && (LP & (LP - 1)) == 0 && ((reinterpret_cast<uintptr_t>(f) | reinterpret_cast<uintptr_t>(gF) |
                                                            reinterpret_cast<uintptr_t>(gf)) & 15) == 0) {
    const int sub = lane & (LP - 1), sl = lane / LP, spw = 64 / LP;
    const float4 g4 = *reinterpret_cast<const float4*>(gFr + 4 * sub);
    for (int s0 = 0; s0 < S; s0 += spw) {
      const int s = s0 + sl;
      const bool live = s < S;
      const int sc = live ? s : S - 1;
      const float4 f4 = *reinterpret_cast<const float4*>(fr + (int64_t)sc * C + 4 * sub);
      float q = g4.x * f4.x;
      q = fmaf(g4.y, f4.y, q);
      q = fmaf(g4.z, f4.z, q);
      q = fmaf(g4.w, f4.w, q);
      for (int off = 1; off < LP; off <<= 1) q += __shfl_xor(q, off, 64);
      if (live) {
        const float w2 = wr[s] + (s == S - 1 ? resid : 0.f);
        if (gf) *reinterpret_cast<float4*>(gf + (ray * S + s) * (int64_t)C + 4 * sub) =
            make_float4(w2 * g4.x, w2 * g4.y, w2 * g4.z, w2 * g4.w);
        if (sub == 0) {
          float g = q - qlast + gacc;
          if (s < S - 1) g += gdep * ((starts[ray * S + s] + ends[ray * S + s]) / 2.f);
          gw[ray * S + s] = g;
        }
      }
    }
    return;
  }
  for (int s = lane; s < S; s += 64) {
    float q = 0.f;
    const float w2 = wr[s] + (s == S - 1 ? resid : 0.f);
    for (int ch = 0; ch < C; ++ch) {
      const float g = gFr[ch];
      q += g * fr[(int64_t)s * C + ch];
      if (gf) gf[(ray * S + s) * (int64_t)C + ch] = w2 * g;
    }
    float g = q - qlast + gacc;
    if (s < S - 1) g += gdep * ((starts[ray * S + s] + ends[ray * S + s]) / 2.f);
    gw[ray * S + s] = g;
  }
}

}  // namespace nrhip

using namespace nrhip;

#define LAUNCH_RAYS(KERNEL, R_, stream, ...)                                                             \
  KERNEL<<<(int)(((R_) + kRaysPerBlock - 1) / kRaysPerBlock), 64 * kRaysPerBlock, 0, (hipStream_t)stream>>>( \
      __VA_ARGS__)

extern "C" int nrhip_render_weight_from_alpha(const float* alphas, int64_t r, int32_t s, float* weights,
                                              float* trans, void* stream) {
  NR_REQUIRE(alphas && weights && r >= 0 && s >= 0, NRHIP_ERR_INVALID_ARG, "render_weight_from_alpha: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(weights_fwd_kernel<0>, r, stream, alphas, nullptr, nullptr, r, s, weights, trans, nullptr);
  return check_launch("render_weight_from_alpha");
}

extern "C" int nrhip_render_weight_from_alpha_bwd(const float* alphas, const float* grad_w, const float* grad_t,
                                                  int64_t r, int32_t s, float* grad_alphas, void* stream) {
  NR_REQUIRE(alphas && grad_w && grad_alphas && r >= 0 && s >= 0, NRHIP_ERR_INVALID_ARG,
             "render_weight_from_alpha_bwd: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(weights_bwd_kernel<0>, r, stream, alphas, nullptr, nullptr, grad_w, grad_t, r, s, grad_alphas);
  return check_launch("render_weight_from_alpha_bwd");
}

extern "C" int nrhip_render_weight_from_density(const float* t_starts, const float* t_ends, const float* sigmas,
                                                int64_t r, int32_t s, float* weights, float* trans, float* alphas,
                                                void* stream) {
  NR_REQUIRE(t_starts && t_ends && sigmas && weights && r >= 0 && s >= 0, NRHIP_ERR_INVALID_ARG,
             "render_weight_from_density: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(weights_fwd_kernel<1>, r, stream, t_starts, t_ends, sigmas, r, s, weights, trans, alphas);
  return check_launch("render_weight_from_density");
}

extern "C" int nrhip_render_weight_from_density_bwd(const float* t_starts, const float* t_ends, const float* sigmas,
                                                    const float* grad_w, int64_t r, int32_t s, float* grad_sigmas,
                                                    void* stream) {
  NR_REQUIRE(t_starts && t_ends && sigmas && grad_w && grad_sigmas && r >= 0 && s >= 0, NRHIP_ERR_INVALID_ARG,
             "render_weight_from_density_bwd: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(weights_bwd_kernel<1>, r, stream, t_starts, t_ends, sigmas, grad_w, nullptr, r, s, grad_sigmas);
  return check_launch("render_weight_from_density_bwd");
}

extern "C" int nrhip_weights_from_density(const float* deltas, const float* densities, int64_t r, int32_t s,
                                          float* weights, void* stream) {
  NR_REQUIRE(deltas && densities && weights && r >= 0 && s >= 0, NRHIP_ERR_INVALID_ARG,
             "weights_from_density: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(weights_fwd_kernel<2>, r, stream, deltas, densities, nullptr, r, s, weights, nullptr, nullptr);
  return check_launch("weights_from_density");
}

extern "C" int nrhip_weights_from_density_bwd(const float* deltas, const float* densities, const float* grad_w,
                                              int64_t r, int32_t s, float* grad_densities, void* stream) {
  NR_REQUIRE(deltas && densities && grad_w && grad_densities && r >= 0 && s >= 0, NRHIP_ERR_INVALID_ARG,
             "weights_from_density_bwd: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(weights_bwd_kernel<2>, r, stream, deltas, densities, nullptr, grad_w, nullptr, r, s, grad_densities);
  return check_launch("weights_from_density_bwd");
}

extern "C" int nrhip_accumulate_along_rays(const float* weights, const float* values, int64_t r, int32_t s,
                                           int32_t c, float* out, void* stream) {
  NR_REQUIRE(weights && out && r >= 0 && s >= 0 && c >= 1, NRHIP_ERR_INVALID_ARG, "accumulate_along_rays: bad argument");
  if (r == 0) return NRHIP_OK;
  LAUNCH_RAYS(accumulate_kernel, r, stream, weights, values, r, s, c, out);
  return check_launch("accumulate_along_rays");
}

extern "C" int nrhip_embedding_lerp_fwd(const float* weight, const int64_t* idx_lo, const int64_t* idx_hi,
                                        const float* frac, int64_t r, int32_t n_embed, int32_t dim, float* out,
                                        void* stream) {
  NR_REQUIRE(weight && idx_lo && out && r >= 0 && n_embed >= 1 && dim >= 1 && (!idx_hi || frac), NRHIP_ERR_INVALID_ARG,
             "embedding_lerp_fwd: bad argument");
  if (r == 0) return NRHIP_OK;
  embedding_lerp_fwd_kernel<<<grid_for(r * dim, 256), 256, 0, (hipStream_t)stream>>>(weight, idx_lo, idx_hi, frac, r,
                                                                                    n_embed, dim, out);
  return check_launch("embedding_lerp_fwd");
}

extern "C" int nrhip_embedding_lerp_bwd(const float* g_out, const int64_t* idx_lo, const int64_t* idx_hi,
                                        const float* frac, int64_t r, int32_t n_embed, int32_t dim, float* grad_weight,
                                        void* stream) {
  NR_REQUIRE(g_out && idx_lo && grad_weight && r >= 0 && n_embed >= 1 && dim >= 1 && (!idx_hi || frac),
             NRHIP_ERR_INVALID_ARG, "embedding_lerp_bwd: bad argument");
  if (r == 0) return NRHIP_OK;
  const int64_t cells = (int64_t)n_embed * dim;
  const bool in_lds = cells <= kEmbedLdsFloats;
  int blocks = (int)((r * dim + 256 * 16 - 1) / (256 * 16));  // >= 16 elements per thread before a table is flushed
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  embedding_lerp_bwd_kernel<<<blocks, 256, in_lds ? cells * sizeof(float) : 0, (hipStream_t)stream>>>(
      g_out, idx_lo, idx_hi, frac, r, n_embed, dim, in_lds ? 1 : 0, grad_weight);
  return check_launch("embedding_lerp_bwd");
}

extern "C" int nrhip_accumulate_along_rays_bwd(const float* weights, const float* values, const float* g_out, int64_t r,
                                               int32_t s, int32_t c, float* grad_weights, float* grad_values,
                                               void* stream) {
  NR_REQUIRE(weights && values && g_out && r >= 0 && s >= 0 && c >= 1 && (grad_weights || grad_values),
             NRHIP_ERR_INVALID_ARG, "accumulate_along_rays_bwd: bad argument");
  if (r == 0 || s == 0) return NRHIP_OK;
  LAUNCH_RAYS(accumulate_bwd_kernel, r, stream, weights, values, g_out, r, s, c, grad_weights, grad_values);
  return check_launch("accumulate_along_rays_bwd");
}

extern "C" int nrhip_composite_fwd(const float* weights, const float* features, const float* starts,
                                   const float* ends, int64_t r, int32_t s, int32_t c, float* out_features,
                                   float* out_depth, float* out_acc, void* stream) {
  NR_REQUIRE(weights && features && starts && ends && out_features && out_depth && out_acc && r >= 0 && s >= 1 &&
                 c >= 1,
             NRHIP_ERR_INVALID_ARG, "composite_fwd: bad argument");
  if (r == 0) return NRHIP_OK;
  LAUNCH_RAYS(composite_fwd_kernel, r, stream, weights, features, starts, ends, r, s, c, out_features, out_depth,
              out_acc);
  return check_launch("composite_fwd");
}

extern "C" int nrhip_composite_bwd(const float* weights, const float* features, const float* starts,
                                   const float* ends, const float* g_features, const float* g_depth,
                                   const float* g_acc, int64_t r, int32_t s, int32_t c, float* grad_weights,
                                   float* grad_features, void* stream) {
  NR_REQUIRE(weights && features && starts && ends && g_features && grad_weights && r >= 0 && s >= 1 && c >= 1,
             NRHIP_ERR_INVALID_ARG, "composite_bwd: bad argument");
  if (r == 0) return NRHIP_OK;
  LAUNCH_RAYS(composite_bwd_kernel, r, stream, weights, features, starts, ends, g_features, g_depth, g_acc, r, s, c,
              grad_weights, grad_features);
  return check_launch("composite_bwd");
}
