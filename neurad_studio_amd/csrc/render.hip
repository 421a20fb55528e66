// F1 (+C1+C2) fused: multi-resolution hash-grid lookup -> tiny MLPs on the matrix cores -> SDF/density
// head -> transmittance scan -> feature/depth/accumulation compositing, one wavefront per ray.
//
// Wave layout (tile = 16 consecutive samples of one ray):   lane = 16*g + j
//     j = lane & 15  : sample inside the tile          g = lane >> 4 : "k-group" (0..3)
//   * gather: the 4 lanes (j, g=0..3) of a sample each look up L/4 levels  -> 8 features per lane.  Those 8
//     registers ARE the MFMA B-fragments of geo-layer 1 (v_mfma_f32_16x16x4_f32: B[k = lane>>4][n = lane&15]),
//     the weight matrix is the A operand with its K axis permuted to match (done once, at LDS staging).
//   * transposed chaining H^T = W . X^T : the D tile comes out as D[neuron = 4g + r][sample = j]  (r = 0..3),
//     i.e. lane (j,g) again holds 4 activations of ITS sample per 16-neuron block -> they feed the next layer's
//     B operand directly.  No cross-lane traffic, no LDS round trip between the five layers.
//   * compositing: the 16 samples of a tile live in one DPP row (16 lanes) -> exclusive transmittance scan
//     with row_shr DPP ops; the four rows (g) carry identical copies, each accumulates w*feature for its
//     own 8 feature channels.
// Exact fp32 (f32-input MFMA == fmaf chain): the parity target is the reference's fp32 torch path.
#include <stdlib.h>

#include "render_common.h"


namespace nrhip {



// L levels, F features/level (L*F == 32), H hidden width, HALF = fp16 table, COMPOSITE = fuse C1+C2.
template <int L, int F, int H, bool HALF, bool COMPOSITE>
__global__ __launch_bounds__(256, 2) void render_kernel(FieldDev fd, RaysDev rays, float* __restrict__ out_feat,
                                                        float* __restrict__ out_depth, float* __restrict__ out_acc,
                                                        float* __restrict__ out_w, float* __restrict__ out_sdf,
                                                        float* __restrict__ out_alpha, SaveDev sv) {
  static_assert(L * F == 32 && L % 4 == 0, "fused kernel needs L*F == 32, L % 4 == 0");
  static_assert(H % 16 == 0 && H >= 16 && H <= 128, "hidden width");
  using Ld = Lds<H>;
  constexpr int NB = H / 16;
  constexpr int LPL = L / 4;  // levels per lane
  extern __shared__ __attribute__((aligned(16))) float lds[];

  // ---- stage weights (once per workgroup; the grid is persistent over rays) ----------------------
  stage_field_weights<H>(fd, lds);
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const uint32_t mask = (1u << fd.grid.log2T) - 1u;
  const int S = rays.S;
  const int ntile = (S + 15) >> 4;

  float scal_l[LPL];
#pragma unroll
  for (int q = 0; q < LPL; ++q) scal_l[q] = lds[Ld::SCAL + LPL * g + q];

  for (int64_t ray = (int64_t)blockIdx.x * 4 + wid; ray < rays.R; ray += (int64_t)gridDim.x * 4) {
    const float ox = rays.o[3 * ray], oy = rays.o[3 * ray + 1], oz = rays.o[3 * ray + 2];
    const float dx = rays.d[3 * ray], dy = rays.d[3 * ray + 1], dz = rays.d[3 * ray + 2];
    const float area = rays.area[ray];

    // per-ray part of feat layer 0:  rb[n] = fb0[n] + Σ_c fw0[n][32+c] * SH_c((d+1)/2)   (neurad_field.py:140-141)
    f32x4 rb[NB];
    f32x4 shq = f32x4{0.f, 0.f, 0.f, 0.f};  // SH coefficients 4g..4g+3 (only read when saving for the backward)
    {
      float sh[16];
      sh4((dx + 1.f) / 2.f, (dy + 1.f) / 2.f, (dz + 1.f) / 2.f, sh);
      if constexpr (!COMPOSITE) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if ((c >> 2) == g) shq[c & 3] = sh[c];
      }
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) rb[mb] = *reinterpret_cast<const f32x4*>(lds + Ld::BF0 + 16 * mb + 4 * g);
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(lds + Ld::SHW + c * H + 16 * mb + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) rb[mb][r] = fmaf(w[r], sh[c], rb[mb][r]);
        }
    }

    float carry = COMPOSITE ? (fd.use_sdf ? 1.f : 0.f) : 0.f;  // running transmittance (product) / optical depth (sum)
    float acc_w = 0.f, acc_d = 0.f;
    f32x4 fa[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    f32x4 flast[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

    for (int t = 0; t < ntile; ++t) {
      // The weights in LDS are loop invariant: without this opaque offset the compiler hoists ~200 LDS
      // loads out of the tile loop and spills them.  `lw` re-derives the LDS base once per tile.
      int opaque = 0;
      asm volatile("" : "+v"(opaque));
      const float* lw = lds + opaque;
      const int s = 16 * t + j;
      const bool live = s < S;
      const int64_t si = ray * rays.stride + (live ? s : S - 1);
      const float t0 = rays.starts[si], t1 = rays.ends[si];
      const SamplePos p = sample_position(ox, oy, oz, dx, dy, dz, area, t0, t1, fd.scale);

      // ---- gather: LPL levels x 8 corners, rescaled (H1 + H4) ------------------------------------
      // All 8*LPL gathers of the tile are issued before the first lerp: one memory round trip per tile
      // instead of one per level (the loads of a level would otherwise wait for the previous level's blend).
      float feat[8];
      {
        Corners cs[LPL];
        float fv[LPL][8][F];
#pragma unroll
        for (int q = 0; q < LPL; ++q) cs[q] = hash_corners(p.x, p.y, p.z, scal_l[q], mask);
#pragma unroll
        for (int q = 0; q < LPL; ++q)
#pragma unroll
          for (int k = 0; k < 8; ++k)
            Entry<F, HALF>::load(fd.table, ((uint32_t)(LPL * g + q) << fd.grid.log2T) + cs[q].idx[k], fv[q][k]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < LPL; ++q) {
          float v[F];
          lerp_corners<F>(cs[q], fv[q], v);
          const float w = rescale_weight(scal_l[q], p.std);
#pragma unroll
          for (int f = 0; f < F; ++f) feat[q * F + f] = v[f] * w;
        }
      }

      bool saving = false;
      int64_t srow = 0;
      if constexpr (!COMPOSITE) {
        saving = sv.enc != nullptr && live;
        srow = ray * S + s;
        if (saving) {
          float* ep = sv.enc + srow * 32 + 8 * g;
          *reinterpret_cast<f32x4*>(ep) = f32x4{feat[0], feat[1], feat[2], feat[3]};
          *reinterpret_cast<f32x4*>(ep + 4) = f32x4{feat[4], feat[5], feat[6], feat[7]};
        }
      }

      // ---- geo MLP layer 0 (32 -> H, ReLU) ---------------------------------------------------------
      f32x4 h[NB];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BG0 + 16 * mb + 4 * g);
      mfma_layer<NB, 8>(lw + Ld::G0, lane, feat, h);
      float hb[H / 4];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[4 * mb + r] = fmaxf(h[mb][r], 0.f);

      if constexpr (!COMPOSITE) {
        if (saving) {
#pragma unroll
          for (int mb = 0; mb < NB; ++mb)
            *reinterpret_cast<f32x4*>(sv.hg + srow * H + 16 * mb + 4 * g) =
                f32x4{hb[4 * mb], hb[4 * mb + 1], hb[4 * mb + 2], hb[4 * mb + 3]};
        }
      }

      // ---- geo MLP layer 1 (H -> 1 + 32): row 0 (sdf / raw density) on the VALU, rows 1..32 on MFMA --
      float sdf = 0.f;
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x4 w = *reinterpret_cast<const f32x4*>(lw + Ld::SDFW + 16 * mb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) sdf = fmaf(w[r], hb[4 * mb + r], sdf);
      }
      sdf += __shfl_xor(sdf, 16, 64);
      sdf += __shfl_xor(sdf, 32, 64);
      sdf += lw[Ld::BG1];
      f32x4 e[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) {
        const float* bp = lw + Ld::BG1 + 1 + 16 * mb + 4 * g;
        e[mb] = f32x4{bp[0], bp[1], bp[2], bp[3]};
      }
      mfma_layer<2, H / 4>(lw + Ld::G1, lane, hb, e);
      float eb[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) eb[k] = e[k >> 2][k & 3];

      // ---- feature MLP (32 [+16 SH folded into rb] -> H -> H -> 32), residual add -------------------
      if constexpr (!COMPOSITE) {
        if (saving) {
          float* xp = sv.xf + srow * 48;
          *reinterpret_cast<f32x4*>(xp + 4 * g) = e[0];
          *reinterpret_cast<f32x4*>(xp + 16 + 4 * g) = e[1];
          *reinterpret_cast<f32x4*>(xp + 32 + 4 * g) = shq;
        }
      }
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) h[mb] = rb[mb];
      mfma_layer<NB, 8>(lw + Ld::F0, lane, eb, h);
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[4 * mb + r] = fmaxf(h[mb][r], 0.f);
      if constexpr (!COMPOSITE) {
        if (saving) {
#pragma unroll
          for (int mb = 0; mb < NB; ++mb)
            *reinterpret_cast<f32x4*>(sv.hf + srow * (2 * H) + 16 * mb + 4 * g) =
                f32x4{hb[4 * mb], hb[4 * mb + 1], hb[4 * mb + 2], hb[4 * mb + 3]};
        }
      }
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BF1 + 16 * mb + 4 * g);
      mfma_layer<NB, H / 4>(lw + Ld::F1, lane, hb, h);
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[4 * mb + r] = fmaxf(h[mb][r], 0.f);
      if constexpr (!COMPOSITE) {
        if (saving) {
#pragma unroll
          for (int mb = 0; mb < NB; ++mb)
            *reinterpret_cast<f32x4*>(sv.hf + srow * (2 * H) + H + 16 * mb + 4 * g) =
                f32x4{hb[4 * mb], hb[4 * mb + 1], hb[4 * mb + 2], hb[4 * mb + 3]};
        }
      }
      f32x4 o[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) o[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BF2 + 16 * mb + 4 * g);
      mfma_layer<2, H / 4>(lw + Ld::F2, lane, hb, o);
      o[0] += e[0];
      o[1] += e[1];  // feature = geo_embedding + mlp_feature(...)   (neurad_field.py:141)

      // ---- head (F4 / trunc_exp) -------------------------------------------------------------------
      float a_or_d;  // alpha (sdf mode) or density
      if (fd.use_sdf) a_or_d = __builtin_amdgcn_rcpf(1.f + __expf(sdf * fd.beta));  // sigmoid(-sdf*beta)
      else a_or_d = expf(sdf);

      if constexpr (!COMPOSITE) {
        if (live) {
          float* fp = out_feat + (ray * S + s) * 32;
          *reinterpret_cast<f32x4*>(fp + 4 * g) = o[0];
          *reinterpret_cast<f32x4*>(fp + 16 + 4 * g) = o[1];
          if (g == 0) {
            out_sdf[ray * S + s] = sdf;
            out_alpha[ray * S + s] = a_or_d;
          }
        }
      } else {
        // ---- C1: transmittance scan over the 16 samples of the DPP row, carried across tiles -------
        float w, T;
        if (fd.use_sdf) {
          const float alpha = live ? a_or_d : 0.f;
          float incl = 1.f - alpha;
          incl *= row_shr<1>(incl, 1.f);
          incl *= row_shr<2>(incl, 1.f);
          incl *= row_shr<4>(incl, 1.f);
          incl *= row_shr<8>(incl, 1.f);
          const float excl = row_shr<1>(incl, 1.f);
          T = carry * excl;
          w = T * alpha;
          carry *= __shfl(incl, (lane & 48) | 15, 64);
        } else {
          const float sd = live ? a_or_d * (t1 - t0) : 0.f;
          float incl = sd;
          incl += row_shr<1>(incl, 0.f);
          incl += row_shr<2>(incl, 0.f);
          incl += row_shr<4>(incl, 0.f);
          incl += row_shr<8>(incl, 0.f);
          const float excl = row_shr<1>(incl, 0.f);
          T = expf(-(carry + excl));
          w = T * (1.f - expf(-sd));
          carry += __shfl(incl, (lane & 48) | 15, 64);
        }
        if (out_w && live && g == 0) out_w[ray * S + s] = w;
        // ---- C2 accumulation ------------------------------------------------------------------------
        acc_w += w;
        if (s < S - 1) acc_d += w * ((t0 + t1) / 2.f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fa[0][r] = fmaf(o[0][r], w, fa[0][r]);
          fa[1][r] = fmaf(o[1][r], w, fa[1][r]);
        }
        if (s == S - 1) flast[0] = o[0], flast[1] = o[1];
      }
    }

    if constexpr (COMPOSITE) {
      const float acc = row_sum16(acc_w);
      const float dep = row_sum16(acc_d);
      const float resid = 1.f - acc;  // goes onto the last (sky) sample (models/neurad.py:381)
      fa[0] += flast[0] * resid;      // flast is non-zero only in the lane that owns sample S-1
      fa[1] += flast[1] * resid;
#pragma unroll
      for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) fa[mb][r] = row_sum16(fa[mb][r]);
      if (j == 0) {
        float* fp = out_feat + ray * 32;
        *reinterpret_cast<f32x4*>(fp + 4 * g) = fa[0];
        *reinterpret_cast<f32x4*>(fp + 16 + 4 * g) = fa[1];
        if (g == 0) {
          out_acc[ray] = acc;
          out_depth[ray] = dep;
        }
      }
    }
  }
}

int validate_field(const nrhip_field* f) {
  NR_REQUIRE(f, NRHIP_ERR_INVALID_ARG, "field descriptor is NULL");
  if (int e = validate_grid(&f->grid)) return e;
  NR_REQUIRE(f->table && f->static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "field: NULL table or non-positive scale");
  const nrhip_mlp& a = f->geo;
  const nrhip_mlp& b = f->feat;
  const int in = f->grid.num_levels * f->grid.n_features;
  NR_REQUIRE(in == 32 && f->grid.num_levels % 4 == 0, NRHIP_ERR_UNSUPPORTED,
             "fused field kernel needs L*F == 32 with L %% 4 == 0 (got L=%d F=%d); use the unfused ops",
             f->grid.num_levels, f->grid.n_features);
  NR_REQUIRE(a.num_layers == 2 && b.num_layers == 3 && a.in_dim == 32 && a.out_dim == 33 && b.in_dim == 48 &&
                 b.out_dim == 32 && a.hidden_dim == b.hidden_dim,
             NRHIP_ERR_UNSUPPORTED,
             "fused field kernel needs geo 32->H->33 (2 layers) and feat 48->H->H->32 (3 layers); use the unfused ops");
  NR_REQUIRE(a.hidden_dim == 32 || a.hidden_dim == 64, NRHIP_ERR_UNSUPPORTED,
             "fused field kernel is instantiated for hidden width 32 and 64 (got %d)", a.hidden_dim);
  for (int l = 0; l < 2; ++l) NR_REQUIRE(a.weight[l], NRHIP_ERR_INVALID_ARG, "geo weight %d is NULL", l);
  for (int l = 0; l < 3; ++l) NR_REQUIRE(b.weight[l], NRHIP_ERR_INVALID_ARG, "feat weight %d is NULL", l);
  return NRHIP_OK;
}

FieldDev to_dev(const nrhip_field& f) {
  FieldDev d;
  d.grid = to_dev(f.grid);
  d.table = f.table;
  d.scale = f.static_scale;
  d.gw0 = f.geo.weight[0], d.gb0 = f.geo.bias[0];
  d.gw1 = f.geo.weight[1], d.gb1 = f.geo.bias[1];
  d.fw0 = f.feat.weight[0], d.fb0 = f.feat.bias[0];
  d.fw1 = f.feat.weight[1], d.fb1 = f.feat.bias[1];
  d.fw2 = f.feat.weight[2], d.fb2 = f.feat.bias[2];
  d.use_sdf = f.use_sdf;
  d.beta = f.beta;
  return d;
}

int persistent_blocks(const void* kernel, size_t lds_bytes, int64_t n_rays, int max_per_cu) {
  static int n_cu = 0;
  if (!n_cu) {
    int dev = 0;
    hipDeviceProp_t p;
    n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) ? p.multiProcessorCount
                                                                                              : 256;
  }
  int nb = 0;  // persistent grid: as many workgroups per CU as registers + LDS admit
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, lds_bytes) != hipSuccess || nb < 1) nb = 2;
  if (nb > max_per_cu) nb = max_per_cu;
  const int64_t cap = (int64_t)n_cu * nb, blocks = (n_rays + 3) / 4;
  return (int)(blocks < cap ? blocks : cap);
}

template <int L, int F, int H, bool HALF, bool COMPOSITE>
static int launch_render(const FieldDev& fd, const RaysDev& rd, float* of, float* od, float* oa, float* ow, float* os,
                         float* oal, const SaveDev& sv, hipStream_t st) {
  constexpr size_t lds = Lds<H>::TOTAL * sizeof(float);
  auto kern = render_kernel<L, F, H, HALF, COMPOSITE>;
  static bool configured = false;
  if (lds > 64 * 1024 && !configured) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  static int per_cu_blocks = 0;  // occupancy query once per instantiation
  if (!per_cu_blocks) per_cu_blocks = persistent_blocks((const void*)kern, lds, INT64_C(1) << 40, 4);
  const int64_t want = (rd.R + 3) / 4;
  const int blocks = (int)(want < per_cu_blocks ? want : per_cu_blocks);
  kern<<<blocks, 256, lds, st>>>(fd, rd, of, od, oa, ow, os, oal, sv);
  return check_launch("render/field fused kernel");
}

template <bool COMPOSITE>
static int dispatch_render_serial(const nrhip_field* f, const nrhip_rays* rays, float* of, float* od, float* oa,
                                  float* ow, float* os, float* oal, void* stream, const SaveDev& sv) {
  const FieldDev fd = to_dev(*f);
  const RaysDev rd = to_dev(*rays);
  const hipStream_t st = (hipStream_t)stream;
  const int L = f->grid.num_levels, F = f->grid.n_features, H = f->geo.hidden_dim;
  const bool half = f->grid.param_dtype == 1;
#define CASE(L_, F_, H_)                                                                                   \
  if (L == L_ && F == F_ && H == H_) {                                                                     \
    return half ? launch_render<L_, F_, H_, true, COMPOSITE>(fd, rd, of, od, oa, ow, os, oal, sv, st)      \
                : launch_render<L_, F_, H_, false, COMPOSITE>(fd, rd, of, od, oa, ow, os, oal, sv, st);    \
  }
  CASE(16, 2, 64)
  CASE(16, 2, 32)
  CASE(8, 4, 32)
  CASE(8, 4, 64)
  CASE(4, 8, 32)
  CASE(4, 8, 64)
#undef CASE
  set_error("fused field kernel: no instantiation for L=%d F=%d H=%d", L, F, H);
  return NRHIP_ERR_UNSUPPORTED;
}

// A/B switch (read once per process): NRHIP_RENDER_VARIANT = 1 tile-serial, 2 pipelined, 3 pipelined + deferred
// last feature layer.  Unset / 0 = the default chosen from measurements.
static int env_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("NRHIP_RENDER_VARIANT");
    v = e ? atoi(e) : 0;
    if (v < 0 || v > 3) v = 0;
  }
  return v;
}

template <bool COMPOSITE>
static int dispatch_render(const nrhip_field* f, const nrhip_rays* rays, float* of, float* od, float* oa, float* ow,
                           float* os, float* oal, void* stream, const SaveDev& sv = SaveDev{},
                           RenderOpts opts = RenderOpts{0.f, 0}) {
  if (opts.variant == 0) opts.variant = env_variant();
  if (opts.variant == 0) opts.variant = COMPOSITE ? 3 : 2;
  if (opts.variant == 1) {
    NR_REQUIRE(opts.stop_eps == 0.f, NRHIP_ERR_UNSUPPORTED, "early ray termination needs the pipelined kernel");
    return dispatch_render_serial<COMPOSITE>(f, rays, of, od, oa, ow, os, oal, stream, sv);
  }
  return dispatch_render_pipelined<COMPOSITE>(f, rays, of, od, oa, ow, os, oal, stream, sv, opts);
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_field_fwd(const nrhip_field* f, const nrhip_rays* rays, float* feature, float* sdf, float* alpha,
                               void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0 || rays->n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(feature && sdf && alpha, NRHIP_ERR_INVALID_ARG, "field_fwd: NULL output");
  return dispatch_render<false>(f, rays, feature, nullptr, nullptr, nullptr, sdf, alpha, stream);
}

extern "C" int nrhip_field_fwd_train(const nrhip_field* f, const nrhip_rays* rays, float* feature, float* sdf,
                                     float* alpha, float* save_enc, float* save_geo_hidden, float* save_feat_in,
                                     float* save_feat_hidden, void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0 || rays->n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(feature && sdf && alpha && save_enc && save_geo_hidden && save_feat_in && save_feat_hidden,
             NRHIP_ERR_INVALID_ARG, "field_fwd_train: NULL output");
  NR_REQUIRE(((reinterpret_cast<uintptr_t>(save_enc) | reinterpret_cast<uintptr_t>(save_geo_hidden) |
               reinterpret_cast<uintptr_t>(save_feat_in) | reinterpret_cast<uintptr_t>(save_feat_hidden)) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "field_fwd_train: save buffers must be 16-byte aligned");
  const SaveDev sv{save_enc, save_geo_hidden, save_feat_in, save_feat_hidden};
  return dispatch_render<false>(f, rays, feature, nullptr, nullptr, nullptr, sdf, alpha, stream, sv);
}

static int render_fwd_impl(const nrhip_field* f, const nrhip_rays* rays, float* out_features, float* out_depth,
                           float* out_acc, float* out_weights, const RenderOpts& opts, void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(out_features && out_depth && out_acc, NRHIP_ERR_INVALID_ARG, "render_fwd: NULL output");
  NR_REQUIRE(rays->n_samples >= 1, NRHIP_ERR_INVALID_ARG, "render_fwd: needs >= 1 sample per ray");
  NR_REQUIRE(opts.stop_eps >= 0.f && opts.stop_eps < 1.f, NRHIP_ERR_INVALID_ARG, "render_fwd: early_stop_eps %g not in [0,1)",
             (double)opts.stop_eps);
  NR_REQUIRE(opts.variant >= 0 && opts.variant <= 3, NRHIP_ERR_INVALID_ARG, "render_fwd: unknown kernel variant %d",
             opts.variant);
  return dispatch_render<true>(f, rays, out_features, out_depth, out_acc, out_weights, nullptr, nullptr, stream,
                               SaveDev{}, opts);
}

extern "C" int nrhip_render_fwd(const nrhip_field* f, const nrhip_rays* rays, float* out_features, float* out_depth,
                                float* out_acc, float* out_weights, void* stream) {
  return render_fwd_impl(f, rays, out_features, out_depth, out_acc, out_weights, RenderOpts{0.f, 0}, stream);
}

extern "C" int nrhip_render_fwd_ex(const nrhip_field* f, const nrhip_rays* rays, float* out_features, float* out_depth,
                                   float* out_acc, float* out_weights, float early_stop_eps, int32_t variant,
                                   void* stream) {
  return render_fwd_impl(f, rays, out_features, out_depth, out_acc, out_weights, RenderOpts{early_stop_eps, variant},
                         stream);
}
