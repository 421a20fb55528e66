// The glue of the training step as kernels (SURVEY §8 rows S3, C1-C4 and (f)-2 in their TRAINING form).
//
// Round 2's c3 step spent 17 % of its GPU time in ~400 torch library launches per iteration: slices, cats, `where`s,
// tiny reductions, a sort for one quantile, boolean-mask indexing.  Every one of those is a per-ray or per-sample
// elementwise op next to a kernel that already walks the same data; here they are folded into that kernel:
//   prop_weights      RaySamples.get_weights (rays.py:188-210) straight from the bin EDGES + render_depth_simple of the
//                     proposal round (models/neurad.py:396,727-734) -- no deltas tensor, no [R,S,1] views
//   sdf_render        SigmoidDensity (model_components/utils.py:21-41, learnable beta read from device memory)
//                     -> render_weight_from_alpha -> accumulation -> sky residual on the last sample -> features, depth
//                     (models/neurad.py:377-395), and its backward incl. d beta; features land in a [R, 32+A] row so that
//                     the appearance embedding is written beside them (no cat)
//   appearance        models/neurad.py:423-441 with the slot arithmetic in the kernel
//   mask_compact      rows of a boolean ray mask + the inverse map, no host sync (x[mask] is nonzero + a device->host read)
//   lidar_losses      models/neurad.py:485-521: the three depth terms, the 0.95-quantile robust mean (radix select, no sort),
//                     intensity, ray-drop BCE -- a per-ray pass + one single-workgroup pass; gradients in one scatter
// One wavefront per ray for the per-ray scans, exactly like composite.hip.
#include "common.h"
#include "wave_scan.h"

namespace nrhip {
namespace {

constexpr int kRaysPerBlock = 4;  // 4 waves / block

// wave-wide scans on DPP row shifts + readlane (wave_scan.h), not on ds_bpermute shuffles
__device__ __forceinline__ float tf_scan_mul(float v, int lane) { return wscan::incl<wscan::Mul>(v, lane); }
__device__ __forceinline__ float tf_scan_add(float v, int lane) { return wscan::incl<wscan::Add>(v, lane); }
__device__ __forceinline__ float tf_rscan_add(float v, int lane) { return wscan::rincl<wscan::Add>(v, lane); }
__device__ __forceinline__ float tf_sum(float v) { return wscan::reduce<wscan::Add>(v); }
__device__ __forceinline__ float tf_nan_to_num(float v) {
  if (v != v) return 0.f;
  if (v == INFINITY) return 3.4028234663852886e38f;
  if (v == -INFINITY) return -3.4028234663852886e38f;
  return v;
}

// ---- S3 from bin edges + proposal depth -----------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kRaysPerBlock) void prop_weights_fwd_kernel(const float* __restrict__ edges, int es,
                                                                              const float* __restrict__ dens, int64_t R,
                                                                              int S, float* __restrict__ weights,
                                                                              float* __restrict__ depth) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float* e = edges + ray * es;
  float carry = 0.f, dacc = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool live = s < S;
    const float e0 = live ? e[s] : 0.f, e1 = live ? e[s + 1] : 0.f;
    const float step = live ? (e1 - e0) * dens[ray * S + s] : 0.f;  // delta_density = deltas * densities
    const float alpha = 1.f - expf(-step);
    const float incl = tf_scan_add(step, lane);
    const float excl = wscan::shift_up1(incl, 0.f, lane);
    const float T = expf(-(carry + excl));
    carry += wscan::last(incl);
    if (live) {
      const float w = tf_nan_to_num(alpha * T);
      weights[ray * S + s] = w;
      dacc += w * ((e0 + e1) / 2.f);
    }
  }
  if (depth) {
    dacc = tf_sum(dacc);
    if (lane == 0) depth[ray] = dacc;
  }
}

// dsd_i = G_i T_i e^{-sd_i} - sum_{k>i} G_k w_k,  G = gw + gdepth * mid;  d dens_i = dsd_i * delta_i
__global__ __launch_bounds__(64 * kRaysPerBlock) void prop_weights_bwd_kernel(const float* __restrict__ edges, int es,
                                                                              const float* __restrict__ dens,
                                                                              const float* __restrict__ gw,
                                                                              const float* __restrict__ gdepth, int64_t R,
                                                                              int S, float* __restrict__ gdens) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + (threadIdx.x >> 6);
  if (ray >= R) return;
  const float* e = edges + ray * es;
  const float gd = gdepth ? gdepth[ray] : 0.f;
  const int nchunk = (S + 63) / 64;
  float suffix = 0.f;
  for (int ch = nchunk - 1; ch >= 0; --ch) {
    float carry = 0.f;  // sum of the steps of the earlier chunks (S is small: recomputed)
    for (int p = 0; p < ch; ++p) {
      const int s = p * 64 + lane;  // full chunks: always < S
      carry += tf_sum((e[s + 1] - e[s]) * dens[ray * S + s]);
    }
    const int s = ch * 64 + lane;
    const bool live = s < S;
    const float e0 = live ? e[s] : 0.f, e1 = live ? e[s + 1] : 0.f;
    const float delta = e1 - e0;
    const float step = live ? delta * dens[ray * S + s] : 0.f;
    const float alpha = 1.f - expf(-step);
    const float incl = tf_scan_add(step, lane);
    const float excl = wscan::shift_up1(incl, 0.f, lane);
    const float T = expf(-(carry + excl));
    float G = 0.f;
    if (live) G = (gw ? gw[ray * S + s] : 0.f) + gd * ((e0 + e1) / 2.f);
    const float term = G * alpha * T;
    const float incl_r = tf_rscan_add(term, lane);
    const float after = incl_r - term + suffix;
    if (live) gdens[ray * S + s] = (G * T * expf(-step) - after) * delta;
    suffix += wscan::first(incl_r);
  }
}

// ---- SDF head + C1 + C2, forward ------------------------------------------------------------------------------------
// LDS: one slab of S floats per wave (the ray's final weights, read back sample-major by the feature accumulation).
__global__ __launch_bounds__(64 * kRaysPerBlock) void sdf_render_fwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ beta_ptr, float beta_min, const float* __restrict__ feat,
    const float* __restrict__ edges, int es, int64_t R, int S, int C, float* __restrict__ alpha_out,
    float* __restrict__ w_ns, float* __restrict__ of, int of_stride, float* __restrict__ od, float* __restrict__ oa) {
  extern __shared__ float tf_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  if (ray >= R) return;
  float* wsh = tf_lds + wave * S;
  // beta_ptr == NULL: the density head (use_sdf = False, fields/neurad_field.py:149-151): sigma = trunc_exp(x), and
  // render_weight_from_density's alpha = 1 - exp(-sigma (t_end - t_start)) (models/neurad.py:718-723) -- the same compositing
  const bool dens = beta_ptr == nullptr;
  const float beta = dens ? 0.f : fabsf(*beta_ptr) + beta_min;
  const float* e = edges + ray * es;
  float carry = 1.f, acc = 0.f, depth = 0.f;
  for (int s0 = 0; s0 < S; s0 += 64) {
    const int s = s0 + lane;
    const bool live = s < S;
    float a = 0.f;
    if (live) {
      const float x = sdf[ray * S + s];
      a = dens ? -expm1f(-expf(x) * (e[s + 1] - e[s])) : sigmoidf_(-x * beta);
    }
    const float incl = tf_scan_mul(1.f - a, lane);
    const float excl = wscan::shift_up1(incl, 1.f, lane);
    const float w = a * (carry * excl);
    carry *= wscan::last(incl);
    if (live) {
      alpha_out[ray * S + s] = a;
      wsh[s] = w;
      acc += w;
      if (s < S - 1) {
        depth += w * ((e[s] + e[s + 1]) / 2.f);
        w_ns[ray * (S - 1) + s] = w;
      }
    }
  }
  acc = tf_sum(acc);
  depth = tf_sum(depth);
  if (lane == 0) {
    oa[ray] = acc;
    od[ray] = depth;
    wsh[S - 1] += 1.f - acc;  // what is left behind the last sample is sky
  }
  __builtin_amdgcn_wave_barrier();
  const float* fr = feat + ray * (int64_t)S * C;
  float* o = of + ray * of_stride;
  const int LP = C >> 2;  // lanes per sample when a lane takes 4 channels
  if ((C & 3) == 0 && LP <= 64 && (LP & (LP - 1)) == 0 &&
      ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(o)) & 15) == 0) {
    // 16-byte loads: 64 / LP samples per pass (the 4-byte form moved the 4 KB of a ray's features at ~3 TB/s)
    const int sub = lane & (LP - 1), sl = lane / LP, spw = 64 / LP;
    float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s0 = 0; s0 < S; s0 += spw) {
      const int s = s0 + sl;
      if (s < S) {
        const float4 f4 = *reinterpret_cast<const float4*>(fr + (int64_t)s * C + 4 * sub);
        const float ws = wsh[s];
        a4.x = fmaf(ws, f4.x, a4.x), a4.y = fmaf(ws, f4.y, a4.y), a4.z = fmaf(ws, f4.z, a4.z), a4.w = fmaf(ws, f4.w, a4.w);
      }
    }
    for (int off = 32; off >= LP; off >>= 1) {
      a4.x += __shfl_xor(a4.x, off, 64), a4.y += __shfl_xor(a4.y, off, 64);
      a4.z += __shfl_xor(a4.z, off, 64), a4.w += __shfl_xor(a4.w, off, 64);
    }
    if (sl == 0) *reinterpret_cast<float4*>(o + 4 * sub) = a4;
  } else if (C <= 64 && (64 % C) == 0) {
    float a2 = 0.f;
    const int total = S * C;
    for (int i = lane; i < total; i += 64) a2 += wsh[i / C] * fr[i];
    for (int off = 32; off >= C; off >>= 1) a2 += __shfl_xor(a2, off, 64);
    if (lane < C) o[lane] = a2;
  } else {
    for (int ch = lane; ch < C; ch += 64) {
      float a2 = 0.f;
      for (int s = 0; s < S; ++s) a2 += wsh[s] * fr[(int64_t)s * C + ch];
      o[ch] = a2;
    }
  }
}

// Backward.  LDS per wave: w2[S] (final weights), T[S], gw[S].  gbeta_part: one partial per block.
__global__ __launch_bounds__(64 * kRaysPerBlock) void sdf_render_bwd_kernel(
    const float* __restrict__ sdf, const float* __restrict__ beta_ptr, float beta_min, const float* __restrict__ alpha,
    const float* __restrict__ feat, const float* __restrict__ edges, int es, const float* __restrict__ gF, int gF_stride,
    const float* __restrict__ gD, const float* __restrict__ gA, const float* __restrict__ gWns, int64_t R, int S, int C,
    float* __restrict__ gfeat, float* __restrict__ gsdf, float* __restrict__ gbeta_part) {
  extern __shared__ float tf_lds[];
  __shared__ float gb_wave[kRaysPerBlock];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t ray = (int64_t)blockIdx.x * kRaysPerBlock + wave;
  float gb = 0.f;
  if (ray < R) {
    float* w2 = tf_lds + wave * 3 * S;
    float* Tsh = w2 + S;
    float* gwsh = Tsh + S;
    const bool dens = beta_ptr == nullptr;  // (the density head: see the forward)
    const float beta = dens ? 0.f : fabsf(*beta_ptr) + beta_min;
    const float* e = edges + ray * es;
    const float* ar = alpha + ray * S;
    // 1. transmittance and weights from the saved alphas
    float carry = 1.f, acc = 0.f;
    for (int s0 = 0; s0 < S; s0 += 64) {
      const int s = s0 + lane;
      const bool live = s < S;
      const float a = live ? ar[s] : 0.f;
      const float incl = tf_scan_mul(1.f - a, lane);
      const float excl = wscan::shift_up1(incl, 1.f, lane);
      const float T = carry * excl;
      carry *= wscan::last(incl);
      if (live) {
        Tsh[s] = T;
        w2[s] = a * T;
        acc += a * T;
      }
    }
    acc = tf_sum(acc);
    if (lane == 0) w2[S - 1] += 1.f - acc;
    __builtin_amdgcn_wave_barrier();
    // 2. q_{S-1}
    const float* fr = feat + ray * (int64_t)S * C;
    const float* gFr = gF + ray * gF_stride;
    float qlast = 0.f;
    for (int ch = lane; ch < C; ch += 64) qlast += gFr[ch] * fr[(int64_t)(S - 1) * C + ch];
    qlast = tf_sum(qlast);
    const float gacc = gA ? gA[ray] : 0.f, gdep = gD ? gD[ray] : 0.f;
    // 3. feature gradient and dL/dw
    const int LP = C >> 2;
    if ((C & 3) == 0 && LP <= 64 && (LP & (LP - 1)) == 0 &&
        ((reinterpret_cast<uintptr_t>(feat) | reinterpret_cast<uintptr_t>(gfeat)) & 15) == 0 &&
        ((reinterpret_cast<uintptr_t>(gFr)) & 15) == 0) {
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
          const float ws = w2[s];
          *reinterpret_cast<float4*>(gfeat + (ray * S + s) * (int64_t)C + 4 * sub) =
              make_float4(ws * g4.x, ws * g4.y, ws * g4.z, ws * g4.w);
          if (sub == 0) {
            float g = q - qlast + gacc;
            if (s < S - 1) g += gdep * ((e[s] + e[s + 1]) / 2.f) + (gWns ? gWns[ray * (S - 1) + s] : 0.f);
            gwsh[s] = g;
          }
        }
      }
    } else {
      for (int s = lane; s < S; s += 64) {
        float q = 0.f;
        const float ws = w2[s];
        for (int ch = 0; ch < C; ++ch) {
          const float g = gFr[ch];
          q += g * fr[(int64_t)s * C + ch];
          gfeat[(ray * S + s) * (int64_t)C + ch] = ws * g;
        }
        float g = q - qlast + gacc;
        if (s < S - 1) g += gdep * ((e[s] + e[s + 1]) / 2.f) + (gWns ? gWns[ray * (S - 1) + s] : 0.f);
        gwsh[s] = g;
      }
    }
    __builtin_amdgcn_wave_barrier();
    // 4. alpha-mode weights backward (composite.hip weights_bwd_kernel<0>), 5. the sigmoid head incl. d beta
    const int nchunk = (S + 63) / 64;
    float suffix = 0.f;
    for (int ch = nchunk - 1; ch >= 0; --ch) {
      const int s = ch * 64 + lane;
      const bool live = s < S;
      const float a = live ? ar[s] : 0.f;
      const float T = live ? Tsh[s] : 0.f;
      const float gwi = live ? gwsh[s] : 0.f;
      const float term = gwi * a * T;
      const float incl_r = tf_rscan_add(term, lane);
      const float after = incl_r - term + suffix;
      if (live) {
        const float x = sdf[ray * S + s];
        if (dens) {
          // d L / d sigma = delta (g_i T_i exp(-sigma delta) - sum_{j > i} g_j w_j), exp(-sigma delta) = 1 - alpha;
          // trunc_exp's backward: exp(clamp(x, -15, 15)) (field_components/activations.py:37-41)
          const float gsig = (gwi * T * (1.f - a) - after) * (e[s + 1] - e[s]);
          gsdf[ray * S + s] = gsig * expf(fminf(fmaxf(x, -15.f), 15.f));
        } else {
          const float ga = gwi * T - after / fmaxf(1.f - a, 1e-10f);
          const float ds = ga * a * (1.f - a);  // sigmoid'(x), x = -sdf * beta
          gsdf[ray * S + s] = -ds * beta;
          gb -= ds * x;
        }
      }
      suffix += wscan::first(incl_r);
    }
    gb = tf_sum(gb);
  }
  if (lane == 0) gb_wave[wave] = gb;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kRaysPerBlock; ++k) t += gb_wave[k];
    gbeta_part[blockIdx.x] = t;
  }
}

// ---- the same two kernels with TWO rays per wavefront (S <= 32, C == 32) ---------------------------------------------
// At NeuRAD's 32 samples per ray the per-ray scans above leave lanes 32..63 idle; here each 32-lane half of a wave owns a
// ray: the scans run over the halves independently (DPP inside the 16-lane rows, one readlane per half for the row below /
// above), the feature passes take 4 samples per half and pass (8 lanes x 16 bytes per sample, as above).
// NRHIP_SDF_RENDER_PAIR=1 selects them (see the launchers for what was measured).
namespace half {
template <class Op>
__device__ __forceinline__ float incl(float v, int lane) {
  const float e = Op::template id<float>();
  v = Op::op(wscan::dpp<0x111>(e, v), v);
  v = Op::op(wscan::dpp<0x112>(e, v), v);
  v = Op::op(wscan::dpp<0x114>(e, v), v);
  v = Op::op(wscan::dpp<0x118>(e, v), v);
  const float t0 = wscan::lane_of(v, 15), t2 = wscan::lane_of(v, 47);
  const int row = lane >> 4;
  return Op::op(row == 1 ? t0 : (row == 3 ? t2 : e), v);
}
template <class Op>
__device__ __forceinline__ float rincl(float v, int lane) {
  const float e = Op::template id<float>();
  v = Op::op(v, wscan::dpp<0x101>(e, v));
  v = Op::op(v, wscan::dpp<0x102>(e, v));
  v = Op::op(v, wscan::dpp<0x104>(e, v));
  v = Op::op(v, wscan::dpp<0x108>(e, v));
  const float s1 = wscan::lane_of(v, 16), s3 = wscan::lane_of(v, 48);
  const int row = lane >> 4;
  return Op::op(v, row == 0 ? s1 : (row == 2 ? s3 : e));
}
// the value of lane - 1 inside the half (lanes 0 and 32: `first`)
__device__ __forceinline__ float shift_up1(float v, float first, int lane) {
  float x = wscan::dpp<0x111>(first, v);
  const float a = wscan::lane_of(v, 15), c = wscan::lane_of(v, 47);
  x = lane == 16 ? a : x;
  x = lane == 48 ? c : x;
  return x;
}
// sum over the half, the same value in each of its lanes
__device__ __forceinline__ float sum(float v, int lane) {
  v += wscan::dpp<0x111>(0.f, v);
  v += wscan::dpp<0x112>(0.f, v);
  v += wscan::dpp<0x114>(0.f, v);
  v += wscan::dpp<0x118>(0.f, v);
  const float lo = wscan::lane_of(v, 15) + wscan::lane_of(v, 31), hi = wscan::lane_of(v, 47) + wscan::lane_of(v, 63);
  return lane < 32 ? lo : hi;
}
}  // namespace half

__global__ __launch_bounds__(64 * kRaysPerBlock) void sdf_render_fwd_pair_kernel(
    const float* __restrict__ sdf, const float* __restrict__ beta_ptr, float beta_min, const float* __restrict__ feat,
    const float* __restrict__ edges, int es, int64_t R, int S, float* __restrict__ alpha_out, float* __restrict__ w_ns,
    float* __restrict__ of, int of_stride, float* __restrict__ od, float* __restrict__ oa) {
  constexpr int C = 32;
  extern __shared__ float tf_lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31;
  const int64_t ray0 = (int64_t)blockIdx.x * (2 * kRaysPerBlock) + 2 * wave;
  if (ray0 >= R) return;
  const int64_t ray = ray0 + (lane >> 5);
  const bool rv = ray < R;            // an odd ray count leaves the last wave's upper half without a ray: it computes on
  const int64_t rc = rv ? ray : ray0;  // the lower half's data and stores nothing
  float* wsh = tf_lds + (2 * wave + (lane >> 5)) * S;
  const float beta = fabsf(*beta_ptr) + beta_min;
  const float* e = edges + rc * es;
  const int s = hl;
  const bool live = s < S;
  const float a = live ? sigmoidf_(-sdf[rc * S + s] * beta) : 0.f;
  const float incl = half::incl<wscan::Mul>(1.f - a, lane);
  const float w = a * half::shift_up1(incl, 1.f, lane);
  float depth = 0.f;
  if (live) {
    wsh[s] = w;
    if (rv) alpha_out[ray * S + s] = a;
    if (s < S - 1) {
      depth = w * ((e[s] + e[s + 1]) / 2.f);
      if (rv) w_ns[ray * (S - 1) + s] = w;
    }
  }
  const float acc = half::sum(live ? w : 0.f, lane);
  depth = half::sum(depth, lane);
  if (hl == 0) {
    if (rv) oa[ray] = acc, od[ray] = depth;
    wsh[S - 1] += 1.f - acc;  // what is left behind the last sample is sky
  }
  __builtin_amdgcn_wave_barrier();
  const float* fr = feat + rc * (int64_t)S * C;
  const int sub = lane & 7, sl = hl >> 3;
  float4 a4 = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int s0 = 0; s0 < S; s0 += 4) {
    const int sk = s0 + sl;
    if (sk < S) {
      const float4 f4 = *reinterpret_cast<const float4*>(fr + (int64_t)sk * C + 4 * sub);
      const float ws = wsh[sk];
      a4.x = fmaf(ws, f4.x, a4.x), a4.y = fmaf(ws, f4.y, a4.y), a4.z = fmaf(ws, f4.z, a4.z), a4.w = fmaf(ws, f4.w, a4.w);
    }
  }
#pragma unroll
  for (int off = 16; off >= 8; off >>= 1) {
    a4.x += __shfl_xor(a4.x, off, 64), a4.y += __shfl_xor(a4.y, off, 64);
    a4.z += __shfl_xor(a4.z, off, 64), a4.w += __shfl_xor(a4.w, off, 64);
  }
  if (sl == 0 && rv) *reinterpret_cast<float4*>(of + ray * of_stride + 4 * sub) = a4;
}

__global__ __launch_bounds__(64 * kRaysPerBlock) void sdf_render_bwd_pair_kernel(
    const float* __restrict__ sdf, const float* __restrict__ beta_ptr, float beta_min, const float* __restrict__ alpha,
    const float* __restrict__ feat, const float* __restrict__ edges, int es, const float* __restrict__ gF, int gF_stride,
    const float* __restrict__ gD, const float* __restrict__ gA, const float* __restrict__ gWns, int64_t R, int S,
    float* __restrict__ gfeat, float* __restrict__ gsdf, float* __restrict__ gbeta_part) {
  constexpr int C = 32;
  extern __shared__ float tf_lds[];
  __shared__ float gb_wave[kRaysPerBlock];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, hl = lane & 31;
  const int64_t ray0 = (int64_t)blockIdx.x * (2 * kRaysPerBlock) + 2 * wave;
  float gb = 0.f;
  if (ray0 < R) {
    const int64_t ray = ray0 + (lane >> 5);
    const bool rv = ray < R;
    const int64_t rc = rv ? ray : ray0;
    float* w2 = tf_lds + (2 * wave + (lane >> 5)) * 3 * S;
    float* Tsh = w2 + S;
    float* gwsh = Tsh + S;
    const float beta = fabsf(*beta_ptr) + beta_min;
    const float* e = edges + rc * es;
    const int s = hl;
    const bool live = s < S;
    // 1. transmittance and weights from the saved alphas
    const float a = live ? alpha[rc * S + s] : 0.f;
    const float incl = half::incl<wscan::Mul>(1.f - a, lane);
    const float T = half::shift_up1(incl, 1.f, lane);
    if (live) Tsh[s] = T, w2[s] = a * T;
    const float acc = half::sum(live ? a * T : 0.f, lane);
    if (hl == 0) w2[S - 1] += 1.f - acc;
    __builtin_amdgcn_wave_barrier();
    // 2. q_{S-1}: the half's 32 lanes are the 32 channels
    const float* fr = feat + rc * (int64_t)S * C;
    const float* gFr = gF + rc * gF_stride;
    const float qlast = half::sum(gFr[hl] * fr[(int64_t)(S - 1) * C + hl], lane);
    const float gacc = gA ? gA[rc] : 0.f, gdep = gD ? gD[rc] : 0.f;
    // 3. feature gradient and dL/dw
    const int sub = lane & 7, sl = hl >> 3;
    const float4 g4 = *reinterpret_cast<const float4*>(gFr + 4 * sub);
    for (int s0 = 0; s0 < S; s0 += 4) {
      const int sk = s0 + sl;
      const bool lv = sk < S;
      const int sc = lv ? sk : S - 1;
      const float4 f4 = *reinterpret_cast<const float4*>(fr + (int64_t)sc * C + 4 * sub);
      float q = g4.x * f4.x;
      q = fmaf(g4.y, f4.y, q);
      q = fmaf(g4.z, f4.z, q);
      q = fmaf(g4.w, f4.w, q);
      q += __shfl_xor(q, 1, 64);
      q += __shfl_xor(q, 2, 64);
      q += __shfl_xor(q, 4, 64);
      if (lv) {
        const float ws = w2[sk];
        if (rv)
          *reinterpret_cast<float4*>(gfeat + (ray * S + sk) * (int64_t)C + 4 * sub) =
              make_float4(ws * g4.x, ws * g4.y, ws * g4.z, ws * g4.w);
        if (sub == 0) {
          float g = q - qlast + gacc;
          if (sk < S - 1) g += gdep * ((e[sk] + e[sk + 1]) / 2.f) + (gWns ? gWns[rc * (S - 1) + sk] : 0.f);
          gwsh[sk] = g;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // 4. alpha-mode weights backward, 5. the sigmoid head incl. d beta
    const float gwi = live ? gwsh[s] : 0.f;
    const float Tl = live ? T : 0.f;
    const float term = gwi * a * Tl;
    const float after = half::rincl<wscan::Add>(term, lane) - term;
    if (live && rv) {
      const float ga = gwi * Tl - after / fmaxf(1.f - a, 1e-10f);
      const float ds = ga * a * (1.f - a);  // sigmoid'(x), x = -sdf * beta
      gsdf[ray * S + s] = -ds * beta;
      gb -= ds * sdf[ray * S + s];
    }
    gb = tf_sum(gb);
  }
  if (lane == 0) gb_wave[wave] = gb;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < kRaysPerBlock; ++k) t += gb_wave[k];
    gbeta_part[blockIdx.x] = t;
  }
}

// d beta = sign(beta) * sum(partials): one workgroup, fixed summation order
__global__ __launch_bounds__(1024) void beta_grad_reduce_kernel(const float* __restrict__ part, int n,
                                                                const float* __restrict__ beta_ptr,
                                                                float* __restrict__ out) {
  __shared__ float sh[16];
  float t = 0.f;
  for (int i = threadIdx.x; i < n; i += 1024) t += part[i];
  t = tf_sum(t);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s += sh[k];
    const float b = *beta_ptr;
    out[0] = b > 0.f ? s : (b < 0.f ? -s : 0.f);  // d|b|/db = sign(b), 0 at 0 like torch
  }
}

// ---- C3: appearance embedding with the slot arithmetic in the kernel -------------------------------------------------
struct Slot {
  int64_t lo, hi;
  float frac;
};
__device__ __forceinline__ Slot appearance_slot(int64_t sensor, float t, float duration, int n_per, int temporal) {
#pragma clang fp contract(off)
  Slot s;
  if (!temporal) {
    s.lo = s.hi = sensor;
    s.frac = 0.f;
    return s;
  }
  const float ti = t / duration * (float)n_per;
  const float lo = fminf(fmaxf(floorf(ti), 0.f), (float)(n_per - 1));
  const float hi = fminf(fmaxf(lo + 1.f, 0.f), (float)(n_per - 1));
  s.frac = ti - lo;
  s.lo = (int64_t)lo + sensor * n_per;
  s.hi = (int64_t)hi + sensor * n_per;
  return s;
}

__global__ __launch_bounds__(256) void appearance_fwd_kernel(const float* __restrict__ wgt,
                                                             const int64_t* __restrict__ sensor,
                                                             const float* __restrict__ times, float duration, int n_per,
                                                             int temporal, int64_t R, int E, int D,
                                                             float* __restrict__ out, int out_stride) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= R * D) return;
  const int64_t r = i / D;
  const int d = (int)(i - r * D);
  Slot s = appearance_slot(sensor ? sensor[r] : 0, times ? times[r] : 0.f, duration, n_per, temporal && times);
  s.lo = s.lo < 0 ? 0 : (s.lo >= E ? E - 1 : s.lo);  // a bad sensor index must not read out of bounds
  s.hi = s.hi < 0 ? 0 : (s.hi >= E ? E - 1 : s.hi);
  const float a = wgt[s.lo * D + d];
  out[r * out_stride + d] = (temporal && times) ? a * (1.f - s.frac) + wgt[s.hi * D + d] * s.frac : a;
}

constexpr int kEmbedLds = 8192;
__global__ __launch_bounds__(256) void appearance_bwd_kernel(const float* __restrict__ g, int g_stride,
                                                             const int64_t* __restrict__ sensor,
                                                             const float* __restrict__ times, float duration, int n_per,
                                                             int temporal, int64_t R, int E, int D, int in_lds,
                                                             float* __restrict__ gw) {
  extern __shared__ float tf_lds[];
  const int cells = E * D;
  if (in_lds) {
    for (int k = threadIdx.x; k < cells; k += 256) tf_lds[k] = 0.f;
    __syncthreads();
  }
  float* acc = in_lds ? tf_lds : gw;
  const bool lerp = temporal && times;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < R * D; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / D;
    const int d = (int)(i - r * D);
    Slot s = appearance_slot(sensor ? sensor[r] : 0, times ? times[r] : 0.f, duration, n_per, lerp);
    s.lo = s.lo < 0 ? 0 : (s.lo >= E ? E - 1 : s.lo);
    s.hi = s.hi < 0 ? 0 : (s.hi >= E ? E - 1 : s.hi);
    const float gv = g[r * g_stride + d];
    if (!lerp) {
      atomicAdd(acc + s.lo * D + d, gv);
    } else {
      atomicAdd(acc + s.lo * D + d, gv * (1.f - s.frac));
      atomicAdd(acc + s.hi * D + d, gv * s.frac);
    }
  }
  if (in_lds) {
    __syncthreads();
    for (int k = threadIdx.x; k < cells; k += 256)
      if (tf_lds[k] != 0.f) atomicAdd(gw + k, tf_lds[k]);
  }
}

// ---- rows of a boolean ray mask, in order, + the inverse map: one workgroup, no host round trip ----------------------
__global__ __launch_bounds__(1024) void mask_compact_kernel(const uint8_t* __restrict__ mask, int64_t R,
                                                            int64_t* __restrict__ rows, int64_t n_out,
                                                            int32_t* __restrict__ inverse, int32_t* __restrict__ count) {
  // every thread takes 8 consecutive mask bytes per pass (one 8-byte load where the pointer allows): 8192 rays per pass
  __shared__ uint32_t wave_tot[16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool aligned = (reinterpret_cast<uintptr_t>(mask) & 7) == 0;
  uint32_t base = 0;
  for (int64_t start = 0; start < R; start += 8192) {
    const int64_t i0 = start + 8 * (int64_t)tid;
    uint32_t bits = 0;  // bit k: ray i0 + k is set
    if (i0 + 8 <= R && aligned) {
      const unsigned long long v = *reinterpret_cast<const unsigned long long*>(mask + i0);
#pragma unroll
      for (int k = 0; k < 8; ++k) bits |= ((v >> (8 * k)) & 0xffull) ? (1u << k) : 0u;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (i0 + k < R && mask[i0 + k]) bits |= 1u << k;
    }
    const uint32_t c = (uint32_t)__popc(bits);
    const uint32_t incl = wscan::incl<wscan::Add>(c, lane);  // inclusive prefix over the wave's lanes
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint32_t t = wave_tot[w];
      before += w < wave ? t : 0u;
      total += t;
    }
    uint32_t pos = base + before + incl - c;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = i0 + k;
      if (i < R) {
        const bool m = (bits >> k) & 1u;
        if (m && (int64_t)pos < n_out && rows) rows[pos] = i;
        if (inverse) inverse[i] = m ? (int32_t)pos : -1;
        pos += m ? 1u : 0u;
      }
    }
    base += total;
    __syncthreads();
  }
  if (tid == 0 && count) count[0] = (int32_t)base;
}

// ---- lidar losses (models/neurad.py:485-521) ------------------------------------------------------------------------
// pass 1 (thread per lidar ray, any number of workgroups): per-ray errors and their d/d prediction, per-workgroup partial
// sums of the plain means; pass 2 (ONE workgroup): the 0.95 quantile of the field level's errors by radix select over an
// LDS copy, the robust means, the metrics.  The kept-ray scaling of the field level's gradient happens in the backward
// scatter (it needs the threshold).
constexpr int kMaxDepthLevels = 4;
constexpr int kLossLds = 32768;  // errors kept in LDS for the select (128 KB); larger batches re-read them from memory
struct LidarLossArgs {
  const float* depth[kMaxDepthLevels];  // [R] each: level 0 = the field's depth, 1.. = proposal rounds
  int n_levels;
  const int64_t* rows;  // [n] batch row of lidar ray j
  const float* distance;
  const uint8_t* ret;  // did_return
  const float* intensity;
  const float* intensity_target;
  const float* logits;
  int64_t n;
  float nr_dist, nr_mult, q;
  float* metrics;  // [2 + n_levels]: depth_loss, intensity_loss, ray_drop_loss, depth_loss_0, ...
  float* unit;     // [(n_levels + 2), n]: d err / d prediction per ray (rows 0 and n_levels are finished by the backward)
  float* err;      // [n] the field level's per-ray depth error
  float* part;     // [blocks, kMaxDepthLevels + 1] partial sums of pass 1: bce, level sums
  float* stat;     // [4]: threshold, kept count, kept & returned count, (unused)
  int blocks;
};

__device__ __forceinline__ float block_sum(float v, float* sh16, int nwaves) {
  v = tf_sum(v);
  __syncthreads();  // sh16 may still be read by the previous reduction
  if ((threadIdx.x & 63) == 0) sh16[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int k = 0; k < nwaves; ++k) t += sh16[k];
  return t;
}

// |target - pred| with the reference's treatment of beams without a return, and d/d pred of it
__device__ __forceinline__ void depth_l1(float pred, float dist, bool ret, float nr_dist, float nr_mult, float* err,
                                         float* dpred) {
  const float target = ret ? dist : fmaxf(pred, nr_dist);
  const float diff = target - pred;
  const float m = ret ? 1.f : nr_mult;
  *err = fabsf(diff) * m;
  *dpred = (diff > 0.f ? -1.f : (diff < 0.f ? 1.f : 0.f)) * m;
}

__global__ __launch_bounds__(256) void lidar_rays_kernel(LidarLossArgs A) {
  __shared__ float sh16[16];
  const int64_t n = A.n;
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float bce = 0.f;
  float lsum[kMaxDepthLevels] = {0.f, 0.f, 0.f, 0.f};
  if (j < n) {
    const int64_t r = A.rows[j];
    const bool ret = A.ret[j] != 0;
    const float dist = A.distance[j];
#pragma unroll
    for (int l = 0; l < kMaxDepthLevels; ++l) {
      if (l >= A.n_levels) break;
      float e, dp;
      depth_l1(A.depth[l][r], dist, ret, A.nr_dist, A.nr_mult, &e, &dp);
      if (l == 0) {
        A.err[j] = e;
        A.unit[j] = dp;  // x keep / kept count in the backward
      } else {
        lsum[l] = e;
        A.unit[(int64_t)l * n + j] = dp / (float)n;
      }
    }
    const float x = A.logits[j], y = ret ? 0.f : 1.f;
    bce = fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
    A.unit[(int64_t)(A.n_levels + 1) * n + j] = (sigmoidf_(x) - y) / (float)n;
    A.unit[(int64_t)A.n_levels * n + j] = -2.f * (A.intensity_target[j] - A.intensity[j]);  // x sel / count in the backward
  }
  const float b = block_sum(bce, sh16, 4);
  if (threadIdx.x == 0) A.part[(size_t)blockIdx.x * (kMaxDepthLevels + 1)] = b;
  for (int l = 1; l < A.n_levels; ++l) {
    const float t = block_sum(lsum[l], sh16, 4);
    if (threadIdx.x == 0) A.part[(size_t)blockIdx.x * (kMaxDepthLevels + 1) + l] = t;
  }
}

__global__ __launch_bounds__(1024) void lidar_finish_kernel(LidarLossArgs A) {
  extern __shared__ float tf_lds[];  // min(n, kLossLds) errors
  __shared__ float sh16[16];
  __shared__ uint32_t hist[256];
  __shared__ uint32_t s_prefix, s_k, s_cle, s_mgt;
  const int tid = threadIdx.x;
  const int64_t n = A.n;
  const bool in_lds = n <= kLossLds;
  const float* err = in_lds ? tf_lds : A.err;
  if (in_lds)
    for (int64_t j = tid; j < n; j += 1024) tf_lds[j] = A.err[j];
  // the plain means: partial sums of pass 1, in a fixed order
  {
    float acc[kMaxDepthLevels + 1] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int b = tid; b < A.blocks; b += 1024)
#pragma unroll
      for (int k = 0; k <= kMaxDepthLevels; ++k) acc[k] += A.part[(size_t)b * (kMaxDepthLevels + 1) + k];
    const float bce = block_sum(acc[0], sh16, 16);
    if (tid == 0) A.metrics[2] = bce / (float)n;
    for (int l = 1; l < A.n_levels; ++l) {
      const float t = block_sum(acc[l], sh16, 16);
      if (tid == 0) A.metrics[2 + l] = t / (float)n;
    }
  }
  __syncthreads();  // tf_lds complete
  // ---- torch.quantile(err, q), interpolation='linear' (ATen quantile_compute): radix select of the two order stats ----
  const float rank = A.q * (float)(n - 1);  // fp32, as `q * last_index` on a float32 tensor
  const int64_t k_lo = (int64_t)rank;       // rank >= 0
  const int64_t k_hi = (int64_t)ceilf(rank);
  const float wq = rank - (float)k_lo;
  uint32_t prefix = 0;
  uint32_t k = (uint32_t)k_lo;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 24 - 8 * pass;
    const uint32_t himask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int64_t j = tid; j < n; j += 1024) {
      const uint32_t u = __float_as_uint(err[j]);
      if ((u & himask) == prefix) atomicAdd(&hist[(u >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {  // one wave: 4 bins per lane, prefix over the lanes, the lane whose range holds rank k reports
      const uint32_t c0 = hist[4 * tid], c1 = hist[4 * tid + 1], c2 = hist[4 * tid + 2], c3 = hist[4 * tid + 3];
      const uint32_t tot = c0 + c1 + c2 + c3;
      uint32_t incl = tot;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(incl, off, 64);
        if (tid >= off) incl += u;
      }
      const uint32_t excl = incl - tot;
      if (k >= excl && k < incl) {
        uint32_t kk = k - excl, b = 4 * tid;
        if (kk >= c0) { kk -= c0, ++b; if (kk >= c1) { kk -= c1, ++b; if (kk >= c2) { kk -= c2, ++b; } } }
        s_prefix = prefix | (b << shift);
        s_k = kk;
      }
    }
    __syncthreads();
    prefix = s_prefix;
    k = s_k;
  }
  if (tid == 0) {
    s_cle = 0;
    s_mgt = 0xffffffffu;
  }
  __syncthreads();
  {
    uint32_t cle = 0, mgt = 0xffffffffu;
    for (int64_t j = tid; j < n; j += 1024) {
      const uint32_t u = __float_as_uint(err[j]);
      cle += u <= prefix ? 1u : 0u;
      if (u > prefix && u < mgt) mgt = u;
    }
    atomicAdd(&s_cle, cle);
    atomicMin(&s_mgt, mgt);
  }
  __syncthreads();
  const float v_lo = __uint_as_float(prefix);
  const float v_hi = (k_hi == k_lo || (int64_t)s_cle > k_hi || s_mgt == 0xffffffffu) ? v_lo : __uint_as_float(s_mgt);
  // at::lerp: weight < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
  const float thr = wq < 0.5f ? v_lo + wq * (v_hi - v_lo) : v_hi - (v_hi - v_lo) * (1.f - wq);
  // ---- robust means ----
  float dsum = 0.f, dcnt = 0.f, isum = 0.f, icnt = 0.f;
  for (int64_t j = tid; j < n; j += 1024) {
    const float e = err[j];
    const bool keep = e < thr;
    if (keep) dsum += e, dcnt += 1.f;
    if (keep && A.ret[j] != 0) {
      const float d = A.intensity_target[j] - A.intensity[j];
      isum += d * d;
      icnt += 1.f;
    }
  }
  dsum = block_sum(dsum, sh16, 16);
  dcnt = block_sum(dcnt, sh16, 16);
  isum = block_sum(isum, sh16, 16);
  icnt = block_sum(icnt, sh16, 16);
  if (tid == 0) {
    A.metrics[0] = dsum / dcnt;
    A.metrics[1] = isum / icnt;
    A.stat[0] = thr, A.stat[1] = dcnt, A.stat[2] = icnt, A.stat[3] = 0.f;
  }
}

struct LidarBwdArgs {
  float* gdepth[kMaxDepthLevels];  // [R] each (may be NULL)
  int n_levels;
  const float* unit;
  const float* err;        // [n]
  const float* stat;       // threshold, kept count, kept & returned count
  const uint8_t* ret;      // [n]
  const int32_t* inverse;  // [R]
  const float* up;         // [2 + n_levels] upstream gradients of the metrics
  int64_t R, n;
  float* gint;    // [n] (may be NULL)
  float* glogit;  // [n] (may be NULL)
};
__global__ __launch_bounds__(256) void lidar_losses_bwd_kernel(LidarBwdArgs A) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const float thr = A.stat[0], dcnt = A.stat[1], icnt = A.stat[2];
  if (i < A.R) {
    const int32_t j = A.inverse[i];
#pragma unroll
    for (int l = 0; l < kMaxDepthLevels; ++l) {
      if (l >= A.n_levels) break;
      if (!A.gdepth[l]) continue;
      float g = 0.f;
      if (j >= 0) {
        g = A.unit[(int64_t)l * A.n + j] * A.up[l == 0 ? 0 : 2 + l];
        if (l == 0) g = A.err[j] < thr ? g / dcnt : 0.f;
      }
      A.gdepth[l][i] = g;
    }
  }
  if (i < A.n) {
    if (A.gint) {
      const bool sel = A.err[i] < thr && A.ret[i] != 0;
      A.gint[i] = sel ? A.unit[(int64_t)A.n_levels * A.n + i] / icnt * A.up[1] : 0.f;
    }
    if (A.glogit) A.glogit[i] = A.unit[(int64_t)(A.n_levels + 1) * A.n + i] * A.up[2];
  }
}

}  // namespace
}  // namespace nrhip

using namespace nrhip;

// two rays per wave (sdf_render_*_pair_kernel): NRHIP_SDF_RENDER_PAIR=1, rays of at most 32 samples with 32 channels
static bool sdf_render_pair(int s, int c) { return tuning().sdf_render_pair && s <= 32 && c == 32; }

#define TF_LAUNCH_RAYS(KERNEL, R_, LDS_, stream, ...)                                                              \
  KERNEL<<<(int)(((R_) + kRaysPerBlock - 1) / kRaysPerBlock), 64 * kRaysPerBlock, LDS_, (hipStream_t)stream>>>( \
      __VA_ARGS__)

extern "C" int nrhip_prop_weights_fwd(const float* edges, int32_t edge_stride, const float* densities, int64_t r,
                                      int32_t s, float* weights, float* depth, void* stream) {
  NR_REQUIRE(edges && densities && weights && r >= 0 && s >= 1 && edge_stride >= s + 1, NRHIP_ERR_INVALID_ARG,
             "prop_weights_fwd: bad argument");
  if (r == 0) return NRHIP_OK;
  TF_LAUNCH_RAYS(prop_weights_fwd_kernel, r, 0, stream, edges, edge_stride, densities, r, s, weights, depth);
  return check_launch("prop_weights_fwd");
}

extern "C" int nrhip_prop_weights_bwd(const float* edges, int32_t edge_stride, const float* densities,
                                      const float* grad_weights, const float* grad_depth, int64_t r, int32_t s,
                                      float* grad_densities, void* stream) {
  NR_REQUIRE(edges && densities && grad_densities && (grad_weights || grad_depth) && r >= 0 && s >= 1 &&
                 edge_stride >= s + 1,
             NRHIP_ERR_INVALID_ARG, "prop_weights_bwd: bad argument");
  if (r == 0) return NRHIP_OK;
  TF_LAUNCH_RAYS(prop_weights_bwd_kernel, r, 0, stream, edges, edge_stride, densities, grad_weights, grad_depth, r, s,
                 grad_densities);
  return check_launch("prop_weights_bwd");
}

extern "C" int nrhip_sdf_render_fwd(const float* sdf, const float* beta, float beta_min, const float* features,
                                    const float* edges, int32_t edge_stride, int64_t r, int32_t s, int32_t c,
                                    float* alpha, float* weights_ns, float* out_features, int32_t out_stride,
                                    float* out_depth, float* out_acc, void* stream) {
  NR_REQUIRE(sdf && features && edges && alpha && weights_ns && out_features && out_depth && out_acc && r >= 0 &&
                 s >= 2 && c >= 1 && edge_stride >= s + 1 && out_stride >= c,
             NRHIP_ERR_INVALID_ARG, "sdf_render_fwd: bad argument");
  NR_REQUIRE(s <= 2048, NRHIP_ERR_UNSUPPORTED, "sdf_render_fwd: %d samples per ray (max 2048)", s);
  if (r == 0) return NRHIP_OK;
  if (beta && sdf_render_pair(s, c) && out_stride % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(features) | reinterpret_cast<uintptr_t>(out_features)) & 15) == 0) {
    sdf_render_fwd_pair_kernel<<<(int)((r + 2 * kRaysPerBlock - 1) / (2 * kRaysPerBlock)), 64 * kRaysPerBlock,
                                 (size_t)2 * kRaysPerBlock * s * sizeof(float), (hipStream_t)stream>>>(
        sdf, beta, beta_min, features, edges, edge_stride, r, s, alpha, weights_ns, out_features, out_stride, out_depth,
        out_acc);
    return check_launch("sdf_render_fwd");
  }
  TF_LAUNCH_RAYS(sdf_render_fwd_kernel, r, (size_t)kRaysPerBlock * s * sizeof(float), stream, sdf, beta, beta_min,
                 features, edges, edge_stride, r, s, c, alpha, weights_ns, out_features, out_stride, out_depth, out_acc);
  return check_launch("sdf_render_fwd");
}

extern "C" int nrhip_sdf_render_bwd_workspace(int64_t r, int64_t* floats) {
  NR_REQUIRE(floats && r >= 0, NRHIP_ERR_INVALID_ARG, "sdf_render_bwd_workspace: bad argument");
  *floats = (r + kRaysPerBlock - 1) / kRaysPerBlock;
  return NRHIP_OK;
}

extern "C" int nrhip_sdf_render_bwd(const float* sdf, const float* beta, float beta_min, const float* alpha,
                                    const float* features, const float* edges, int32_t edge_stride,
                                    const float* g_features, int32_t g_stride, const float* g_depth, const float* g_acc,
                                    const float* g_weights_ns, int64_t r, int32_t s, int32_t c, float* grad_features,
                                    float* grad_sdf, float* grad_beta, float* workspace, void* stream) {
  NR_REQUIRE(sdf && alpha && features && edges && g_features && grad_features && grad_sdf && (grad_beta || !beta) &&
                 workspace && r >= 0 && s >= 2 && c >= 1 && edge_stride >= s + 1 && g_stride >= c,
             NRHIP_ERR_INVALID_ARG, "sdf_render_bwd: bad argument");
  NR_REQUIRE(s <= 1024, NRHIP_ERR_UNSUPPORTED, "sdf_render_bwd: %d samples per ray (max 1024)", s);
  if (r == 0) {
    if (grad_beta && hipMemsetAsync(grad_beta, 0, sizeof(float), (hipStream_t)stream) != hipSuccess)
      return check_launch("sdf_render_bwd");
    return NRHIP_OK;
  }
  if (beta && sdf_render_pair(s, c) && g_stride % 4 == 0 &&
      ((reinterpret_cast<uintptr_t>(features) | reinterpret_cast<uintptr_t>(grad_features) |
        reinterpret_cast<uintptr_t>(g_features)) & 15) == 0) {
    const int pblocks = (int)((r + 2 * kRaysPerBlock - 1) / (2 * kRaysPerBlock));
    sdf_render_bwd_pair_kernel<<<pblocks, 64 * kRaysPerBlock, (size_t)2 * kRaysPerBlock * 3 * s * sizeof(float),
                                 (hipStream_t)stream>>>(sdf, beta, beta_min, alpha, features, edges, edge_stride, g_features,
                                                        g_stride, g_depth, g_acc, g_weights_ns, r, s, grad_features, grad_sdf,
                                                        workspace);
    beta_grad_reduce_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(workspace, pblocks, beta, grad_beta);
    return check_launch("sdf_render_bwd");
  }
  const int blocks = (int)((r + kRaysPerBlock - 1) / kRaysPerBlock);
  sdf_render_bwd_kernel<<<blocks, 64 * kRaysPerBlock, (size_t)kRaysPerBlock * 3 * s * sizeof(float), (hipStream_t)stream>>>(
      sdf, beta, beta_min, alpha, features, edges, edge_stride, g_features, g_stride, g_depth, g_acc, g_weights_ns, r, s,
      c, grad_features, grad_sdf, workspace);
  if (beta) beta_grad_reduce_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(workspace, blocks, beta, grad_beta);
  return check_launch("sdf_render_bwd");
}

extern "C" int nrhip_appearance_fwd(const float* weight, const int64_t* sensor_idx, const float* times, float duration,
                                    int32_t n_per_sensor, int32_t temporal, int64_t r, int32_t n_embed, int32_t dim,
                                    float* out, int32_t out_stride, void* stream) {
  NR_REQUIRE(weight && out && r >= 0 && n_embed >= 1 && dim >= 1 && out_stride >= dim && n_per_sensor >= 1 &&
                 (!temporal || duration > 0.f),
             NRHIP_ERR_INVALID_ARG, "appearance_fwd: bad argument");
  if (r == 0) return NRHIP_OK;
  appearance_fwd_kernel<<<grid_for(r * dim, 256), 256, 0, (hipStream_t)stream>>>(
      weight, sensor_idx, times, duration, n_per_sensor, temporal, r, n_embed, dim, out, out_stride);
  return check_launch("appearance_fwd");
}

extern "C" int nrhip_appearance_bwd(const float* g_out, int32_t g_stride, const int64_t* sensor_idx, const float* times,
                                    float duration, int32_t n_per_sensor, int32_t temporal, int64_t r, int32_t n_embed,
                                    int32_t dim, float* grad_weight, void* stream) {
  NR_REQUIRE(g_out && grad_weight && r >= 0 && n_embed >= 1 && dim >= 1 && g_stride >= dim && n_per_sensor >= 1 &&
                 (!temporal || duration > 0.f),
             NRHIP_ERR_INVALID_ARG, "appearance_bwd: bad argument");
  const int64_t cells = (int64_t)n_embed * dim;
  if (hipMemsetAsync(grad_weight, 0, cells * sizeof(float), (hipStream_t)stream) != hipSuccess)
    return check_launch("appearance_bwd");
  if (r == 0) return NRHIP_OK;
  const bool in_lds = cells <= kEmbedLds;
  int blocks = (int)((r * dim + 256 * 16 - 1) / (256 * 16));
  blocks = blocks < 1 ? 1 : (blocks > 256 ? 256 : blocks);
  appearance_bwd_kernel<<<blocks, 256, in_lds ? cells * sizeof(float) : 0, (hipStream_t)stream>>>(
      g_out, g_stride, sensor_idx, times, duration, n_per_sensor, temporal, r, n_embed, dim, in_lds ? 1 : 0, grad_weight);
  return check_launch("appearance_bwd");
}

extern "C" int nrhip_mask_compact(const uint8_t* mask, int64_t r, int64_t* rows, int64_t n_out, int32_t* inverse,
                                  int32_t* count, void* stream) {
  NR_REQUIRE(r >= 0 && n_out >= 0 && r < (INT64_C(1) << 31) && (r == 0 || mask), NRHIP_ERR_INVALID_ARG,
             "mask_compact: bad argument");
  NR_REQUIRE(n_out == 0 || rows, NRHIP_ERR_INVALID_ARG, "mask_compact: rows is NULL");
  mask_compact_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(mask, r, rows, n_out, inverse, count);
  return check_launch("mask_compact");
}

extern "C" int nrhip_lidar_losses_workspace(int64_t n, int64_t* floats) {
  NR_REQUIRE(floats && n >= 0, NRHIP_ERR_INVALID_ARG, "lidar_losses_workspace: bad argument");
  *floats = n + 4 + ((n + 255) / 256) * (kMaxDepthLevels + 1);  // err [n], stat [4], per-workgroup partials
  return NRHIP_OK;
}

extern "C" int nrhip_lidar_losses(const float* const* depths, int32_t n_levels, const int64_t* lidar_rows,
                                  const float* distance, const uint8_t* did_return, const float* intensity,
                                  const float* intensity_target, const float* ray_drop_logits, int64_t n,
                                  float non_return_distance, float non_return_mult, float quantile, float* metrics,
                                  float* unit_grads, float* scratch, void* stream) {
  NR_REQUIRE(depths && n_levels >= 1 && n_levels <= kMaxDepthLevels, NRHIP_ERR_INVALID_ARG,
             "lidar_losses: 1..%d depth levels", kMaxDepthLevels);
  NR_REQUIRE(n >= 1 && lidar_rows && distance && did_return && intensity && intensity_target && ray_drop_logits && metrics &&
                 unit_grads && scratch,
             NRHIP_ERR_INVALID_ARG, "lidar_losses: bad argument (n must be >= 1)");
  NR_REQUIRE(quantile >= 0.f && quantile <= 1.f, NRHIP_ERR_INVALID_ARG, "lidar_losses: quantile outside [0,1]");
  LidarLossArgs a{};
  for (int l = 0; l < n_levels; ++l) {
    NR_REQUIRE(depths[l], NRHIP_ERR_INVALID_ARG, "lidar_losses: depth level %d is NULL", l);
    a.depth[l] = depths[l];
  }
  a.n_levels = n_levels, a.rows = lidar_rows, a.distance = distance, a.ret = did_return, a.intensity = intensity;
  a.intensity_target = intensity_target, a.logits = ray_drop_logits, a.n = n;
  a.nr_dist = non_return_distance, a.nr_mult = non_return_mult, a.q = quantile;
  a.metrics = metrics, a.unit = unit_grads;
  a.blocks = (int)((n + 255) / 256);
  a.err = scratch, a.stat = scratch + n, a.part = scratch + n + 4;
  static thread_local bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)lidar_finish_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                              kLossLds * (int)sizeof(float));
    configured = true;
  }
  lidar_rays_kernel<<<a.blocks, 256, 0, (hipStream_t)stream>>>(a);
  lidar_finish_kernel<<<1, 1024, n <= kLossLds ? (size_t)n * sizeof(float) : 0, (hipStream_t)stream>>>(a);
  return check_launch("lidar_losses");
}

extern "C" int nrhip_lidar_losses_bwd(const float* unit_grads, const float* scratch, const uint8_t* did_return,
                                      const int32_t* inverse, const float* upstream, int32_t n_levels, int64_t r, int64_t n,
                                      float* const* grad_depths, float* grad_intensity, float* grad_logits, void* stream) {
  NR_REQUIRE(unit_grads && scratch && did_return && inverse && upstream && grad_depths && n_levels >= 1 &&
                 n_levels <= kMaxDepthLevels && r >= 0 && n >= 0,
             NRHIP_ERR_INVALID_ARG, "lidar_losses_bwd: bad argument");
  if (r == 0 && n == 0) return NRHIP_OK;
  LidarBwdArgs a{};
  for (int l = 0; l < n_levels; ++l) a.gdepth[l] = grad_depths[l];
  a.n_levels = n_levels, a.unit = unit_grads, a.err = scratch, a.stat = scratch + n, a.ret = did_return;
  a.inverse = inverse, a.up = upstream, a.R = r, a.n = n;
  a.gint = grad_intensity, a.glogit = grad_logits;
  const int64_t m = r > n ? r : n;
  lidar_losses_bwd_kernel<<<grid_for(m, 256), 256, 0, (hipStream_t)stream>>>(a);
  return check_launch("lidar_losses_bwd");
}
