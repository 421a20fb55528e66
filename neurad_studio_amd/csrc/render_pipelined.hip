// F1 (+C1+C2) fused, software-pipelined: the same wave-per-ray / 16-sample-tile / transposed-MFMA-chain layout as
// render.hip (read its header first), with three changes aimed at what bounds that kernel on MI355X -- the chip's
// random-line miss rate, with the fp32 MFMA chain fully exposed behind it:
//
//   * PIPELINE: a wave's (ray, tile) sequence is flattened and the 8*L/4 corner gathers of the NEXT tile (the next
//     tile of this ray, or the first tile of the wave's next ray) are issued right after the current tile's blend,
//     BEFORE its ~170 MFMAs.  The gathered entries stay in flight in VGPRs during the MLP (the kernel runs at 2 waves /
//     SIMD and has the registers), so every wave keeps requests in the memory system all the time instead of going
//     idle on memory for the 2.6 us of its MFMA phase.  Ray constants are wave-uniform -> scalar loads, no VGPRs.
//   * DEFER (composited output only): the last feature layer (H -> 32, no activation) is linear and followed only by
//     sum_s w_s * (.), so the kernel accumulates sum_s w_s*h2_s (H channels) and sum_s w_s*e_s per ray and applies
//     fw2 / fb2 ONCE per ray in the epilogue:  sum_s w_s (fw2 h2_s + fb2 + e_s) = fw2 (sum w h2) + fb2 sum w + sum w e.
//     -32 of 192 MFMAs per tile at H=64 (reassociation only: ~1e-7 relative).
//   * early ray termination (eval option, off by default = exact): once the transmittance carried across tiles falls
//     below `stop_eps` the remaining tiles of the ray are skipped (their weights are < stop_eps in total).
//
// The sky residual (models/neurad.py:381: w_{S-1} += 1 - sum w) is folded into the last tile -- the accumulated weight
// is complete there, so no copy of the last sample's features has to be kept.
#include "render_common.h"

namespace nrhip {

template <int LPL, int F>
struct TileFetch {
  float fv[LPL][8][F];  // corner entries, in flight until the blend
  float x, y, z, std;   // contracted sample position / std: the trilinear offsets and the H4 weights are re-derived from
                        // them at blend time (3 VALU per level and axis) instead of holding 4*LPL more registers
  float t0, t1;
};

// Ray-side kernel arguments as separate `const __restrict__` pointers: the per-ray constants are read at wave-uniform
// addresses, and only noalias/readonly arguments let the compiler turn those reads into scalar loads (SGPRs, no VGPRs,
// off the vector-memory counter) inside a loop that also stores.
struct RayArgs {
  int64_t R;
  int S, stride;
};

// The tile that will be issued NEXT iteration: its sample interval (vector loads, per lane) and ray constants (scalar
// loads) are requested one iteration ahead, so issuing its gathers never waits on a memory round trip.
struct PendingTile {
  int64_t ray;
  int t;
  bool valid;
  float t0, t1;
  float ox, oy, oz, dx, dy, dz, area;
};

__device__ __forceinline__ void load_pending(PendingTile& p, int64_t ray, int t, const RayArgs& ra, int j,
                                             const float* __restrict__ ro, const float* __restrict__ rd,
                                             const float* __restrict__ rarea, const float* __restrict__ rstarts,
                                             const float* __restrict__ rends) {
  // Past the end of this wave's rays the loads still go out (clamped to the last ray, tile 0) and their gathers are
  // issued and dropped: an `if (valid)` around them makes every fetched register a phi at the join, and the copies the
  // compiler puts there wait for ALL gathers before the MFMA phase -- exactly the stall the pipeline exists to remove.
  p.ray = ray, p.t = t, p.valid = ray < ra.R;
  const int64_t rc = p.valid ? ray : ra.R - 1;
  const int s = p.valid ? 16 * t + j : j;
  const int64_t si = rc * ra.stride + (s < ra.S ? s : ra.S - 1);
  p.t0 = rstarts[si];
  p.t1 = rends[si];
  p.ox = ro[3 * rc], p.oy = ro[3 * rc + 1], p.oz = ro[3 * rc + 2];
  p.dx = rd[3 * rc], p.dy = rd[3 * rc + 1], p.dz = rd[3 * rc + 2];
  p.area = rarea[rc];
}

// H2 + H3 + the hash of H1 for one pending tile, then all gathers issued back to back (no waits in here beyond the
// pending tile's own small loads, which were issued a whole tile earlier).
template <int L, int F, bool HALF>
__device__ __forceinline__ void issue_tile(const FieldDev& fd, const PendingTile& pt, int g, uint32_t mask,
                                           const float* scal_lds, TileFetch<L / 4, F>& tf) {
  constexpr int LPL = L / 4;
  tf.t0 = pt.t0;
  tf.t1 = pt.t1;
  const SamplePos p = sample_position(pt.ox, pt.oy, pt.oz, pt.dx, pt.dy, pt.dz, pt.area, pt.t0, pt.t1, fd.scale);
  tf.x = p.x, tf.y = p.y, tf.z = p.z, tf.std = p.std;
#pragma unroll
  for (int q = 0; q < LPL; ++q) {
    const Corners cs = hash_corners(p.x, p.y, p.z, scal_lds[q], mask);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      Entry<F, HALF>::load(fd.table, ((uint32_t)(LPL * g + q) << fd.grid.log2T) + cs.idx[k], tf.fv[q][k]);
  }
}

template <int LPL, int F>
__device__ __forceinline__ void blend_tile(const TileFetch<LPL, F>& tf, const float* scal_lds, float (&feat)[8]) {
  static_assert(LPL * F == 8, "8 features per lane");
#pragma unroll
  for (int q = 0; q < LPL; ++q) {
    const float sc = scal_lds[q];
    Corners c;  // offsets rounded exactly as hash_corners does (encodings.py:431-446)
    const float sx = __fmul_rn(tf.x, sc), sy = __fmul_rn(tf.y, sc), sz = __fmul_rn(tf.z, sc);
    c.ox = __fsub_rn(sx, floorf(sx)), c.oy = __fsub_rn(sy, floorf(sy)), c.oz = __fsub_rn(sz, floorf(sz));
    float v[F];
    lerp_corners<F>(c, tf.fv[q], v);
    const float rw = rescale_weight(sc, tf.std);
#pragma unroll
    for (int f = 0; f < F; ++f) feat[q * F + f] = v[f] * rw;
  }
}

// L levels, F features/level (L*F == 32), H hidden width, HALF = fp16 table, COMPOSITE = fuse C1+C2,
// DEFER = apply the last feature layer once per ray (COMPOSITE only).
template <int L, int F, int H, bool HALF, bool COMPOSITE, bool DEFER>
__global__ __launch_bounds__(256, 2) void render_pipelined_kernel(
    FieldDev fd, RayArgs rays, const float* __restrict__ ro, const float* __restrict__ rd,
    const float* __restrict__ rarea, const float* __restrict__ rstarts, const float* __restrict__ rends,
    float* __restrict__ out_feat, float* __restrict__ out_depth, float* __restrict__ out_acc, float* __restrict__ out_w,
    float* __restrict__ out_sdf, float* __restrict__ out_alpha, SaveDev sv, float stop_eps) {
  static_assert(L * F == 32 && L % 4 == 0, "fused kernel needs L*F == 32, L % 4 == 0");
  static_assert(H % 16 == 0 && H >= 16 && H <= 128, "hidden width");
  static_assert(COMPOSITE || !DEFER, "DEFER applies to the composited output");
  using Ld = Lds<H>;
  constexpr int NB = H / 16;
  constexpr int LPL = L / 4;  // levels per lane
  extern __shared__ __attribute__((aligned(16))) float lds[];

  stage_field_weights<H>(fd, lds);
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const uint32_t mask = (1u << fd.grid.log2T) - 1u;
  const int S = rays.S;
  const int ntile = (S + 15) >> 4;
  const int64_t ray_step = (int64_t)gridDim.x * 4;

  const float* scal_l = lds + Ld::SCAL + LPL * g;  // this lane's levels (re-read per tile: 1 ds_read, no VGPRs held)
  float* rbw = lds + Ld::RB + wid * H;              // this wave's per-ray bias row
  int64_t ray = (int64_t)blockIdx.x * 4 + wid;
  if (ray >= rays.R) return;
  int t = 0;
  // three-stage pipeline over this wave's flattened (ray, tile) sequence:
  //   tf = gathered corners of the CURRENT tile | q = next tile (interval + ray constants loaded, gathers not yet
  //   issued) | the tile after q has its small loads requested at the end of the issue step
  TileFetch<LPL, F> tf;
  PendingTile q;
  load_pending(q, ray, 0, rays, j, ro, rd, rarea, rstarts, rends);
  issue_tile<L, F, HALF>(fd, q, g, mask, scal_l, tf);
  {
    const bool wrap = ntile == 1;
    load_pending(q, wrap ? ray + ray_step : ray, wrap ? 0 : 1, rays, j, ro, rd, rarea, rstarts, rends);
  }

  // per-ray state
  f32x4 shq = f32x4{0.f, 0.f, 0.f, 0.f};
  float carry = 0.f, acc_w = 0.f, acc_d = 0.f;
  constexpr int NHA = DEFER ? H / 4 : 1;
  float ha[NHA];                     // DEFER: sum_s w_s * h2_s (this lane's sample column)
  f32x4 fa[2];                       // sum_s w_s * feature_s, or (DEFER) sum_s w_s * geo_embedding_s

  while (true) {
    if (t == 0) {
      // per-ray part of feat layer 0:  rb[n] = fb0[n] + sum_c fw0[n][32+c] * SH_c((d+1)/2)   (neurad_field.py:140-141)
      const float dx = rd[3 * ray], dy = rd[3 * ray + 1], dz = rd[3 * ray + 2];
      float sh[16];
      sh4((dx + 1.f) / 2.f, (dy + 1.f) / 2.f, (dz + 1.f) / 2.f, sh);
      if constexpr (!COMPOSITE) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if ((c >> 2) == g) shq[c & 3] = sh[c];
      }
      f32x4 rb[NB];
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
      // parked in this wave's LDS row (the 16 lanes of a DPP row hold identical copies; every lane stores its own ->
      // same value to the same address) and re-read as the accumulator init of feat layer 0 in every tile
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) *reinterpret_cast<f32x4*>(rbw + 16 * mb + 4 * g) = rb[mb];
      carry = COMPOSITE ? (fd.use_sdf ? 1.f : 0.f) : 0.f;  // running transmittance (product) / optical depth (sum)
      acc_w = 0.f, acc_d = 0.f;
      fa[0] = fa[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NHA; ++k) ha[k] = 0.f;
    }

    // The weights in LDS are loop invariant: without this opaque offset the compiler hoists ~200 LDS
    // loads out of the tile loop and spills them.  `lw` re-derives the LDS base once per tile.
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const float* lw = lds + opaque;
    const int s = 16 * t + j;
    const bool live = s < S;
    const float t0 = tf.t0, t1 = tf.t1;

    // ---- blend the fetched corners (H1 + H4): the only wait on the gathers ---------------------------
    float feat[8];
    blend_tile<LPL, F>(tf, scal_l, feat);
    __builtin_amdgcn_sched_barrier(0);

    // ---- early ray termination (eval option): the transmittance ENTERING this tile is already below stop_eps, so
    // everything behind it weighs less than stop_eps in total -> this (already fetched) tile is the ray's last.
    // The test lags one tile behind the carry so that the fetch below never has to be re-issued.
    bool stop_here = false;
    if constexpr (COMPOSITE) {
      if (stop_eps > 0.f) {
        const float c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, carry)));
        stop_here = fd.use_sdf ? (c < stop_eps) : (c > -__logf(stop_eps));  // wave-uniform -> scalar branch
      }
    }
    const bool last_tile = t == ntile - 1;
    const bool ray_done = last_tile || stop_here;

    // ---- issue the next tile's gathers: they fly while this tile runs through the MLPs ---------------
    if (ray_done && q.valid && q.ray == ray)  // terminated early: q still points into this ray -> skip to the next ray
      load_pending(q, ray + ray_step, 0, rays, j, ro, rd, rarea, rstarts, rends);  // (one exposed round trip per such ray)
    const bool have_next = q.valid;
    const int64_t nray = q.ray;
    const int nt = q.t;
    issue_tile<L, F, HALF>(fd, q, g, mask, scal_l, tf);  // unconditional (see load_pending)
    {
      const bool wrap = nt + 1 == ntile;  // request the small loads of the tile after it
      load_pending(q, wrap ? nray + ray_step : nray, wrap ? 0 : nt + 1, rays, j, ro, rd, rarea, rstarts, rends);
    }
    __builtin_amdgcn_sched_barrier(0);

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
    for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::RB + wid * H + 16 * mb + 4 * g);
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
    if constexpr (!DEFER) {
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) o[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BF2 + 16 * mb + 4 * g);
      mfma_layer<2, H / 4>(lw + Ld::F2, lane, hb, o);
      o[0] += e[0];
      o[1] += e[1];  // feature = geo_embedding + mlp_feature(...)   (neurad_field.py:141)
    }

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
      float wf = w;  // weight of this sample's FEATURES: the sky residual 1 - sum w goes on sample S-1
      float acc = 0.f;
      if (ray_done) {
        acc = row_sum16(acc_w);
        if (last_tile && s == S - 1) wf += 1.f - acc;
      }
      if constexpr (DEFER) {
#pragma unroll
        for (int k = 0; k < H / 4; ++k) ha[k] = fmaf(hb[k], wf, ha[k]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fa[0][r] = fmaf(e[0][r], wf, fa[0][r]);
          fa[1][r] = fmaf(e[1][r], wf, fa[1][r]);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          fa[0][r] = fmaf(o[0][r], wf, fa[0][r]);
          fa[1][r] = fmaf(o[1][r], wf, fa[1][r]);
        }
      }

      if (ray_done) {
        const float dep = row_sum16(acc_d);
        if (!last_tile && out_w)  // terminated early: the rest of the ray contributes nothing
          for (int s2 = 16 * (t + 1) + lane; s2 < S; s2 += 64) out_w[ray * S + s2] = 0.f;
        f32x4 of2[2];
        if constexpr (DEFER) {
          // features = fw2 . (sum w h2) + fb2 * sum w' + sum w' e ; sum w' = acc + (1 - acc) on a completed ray
          const float wsum = last_tile ? acc + (1.f - acc) : acc;
          of2[0] = of2[1] = f32x4{0.f, 0.f, 0.f, 0.f};
          mfma_layer<2, H / 4>(lw + Ld::F2, lane, ha, of2);
#pragma unroll
          for (int mb = 0; mb < 2; ++mb) {
            const f32x4 b2 = *reinterpret_cast<const f32x4*>(lw + Ld::BF2 + 16 * mb + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) of2[mb][r] = fmaf(b2[r], wsum, row_sum16(of2[mb][r] + fa[mb][r]));
          }
        } else {
#pragma unroll
          for (int mb = 0; mb < 2; ++mb)
#pragma unroll
            for (int r = 0; r < 4; ++r) of2[mb][r] = row_sum16(fa[mb][r]);
        }
        if (j == 0) {
          float* fp = out_feat + ray * 32;
          *reinterpret_cast<f32x4*>(fp + 4 * g) = of2[0];
          *reinterpret_cast<f32x4*>(fp + 16 + 4 * g) = of2[1];
          if (g == 0) {
            out_acc[ray] = acc;
            out_depth[ray] = dep;
          }
        }
      }
    }

    if (!have_next) break;
    ray = nray, t = nt;
  }
}

template <int L, int F, int H, bool HALF, bool COMPOSITE, bool DEFER>
static int launch_pipelined(const FieldDev& fd, const RaysDev& rd, float* of, float* od, float* oa, float* ow, float* os,
                            float* oal, const SaveDev& sv, float stop_eps, hipStream_t st) {
  constexpr size_t lds = Lds<H>::TOTAL_PIPELINED * sizeof(float);
  auto kern = render_pipelined_kernel<L, F, H, HALF, COMPOSITE, DEFER>;
  static bool configured = false;
  if (lds > 64 * 1024 && !configured) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = true;
  }
  static int cap = 0;  // persistent grid, occupancy queried once per instantiation
  if (!cap) cap = persistent_blocks((const void*)kern, lds, INT64_C(1) << 40, 4);
  const int64_t want = (rd.R + 3) / 4;
  const int blocks = (int)(want < cap ? want : cap);
  const RayArgs ra{rd.R, rd.S, rd.stride};
  kern<<<blocks, 256, lds, st>>>(fd, ra, rd.o, rd.d, rd.area, rd.starts, rd.ends, of, od, oa, ow, os, oal, sv, stop_eps);
  return check_launch("render/field fused kernel (pipelined)");
}

template <bool COMPOSITE>
int dispatch_render_pipelined(const nrhip_field* f, const nrhip_rays* rays, float* of, float* od, float* oa, float* ow,
                              float* os, float* oal, void* stream, const SaveDev& sv, const RenderOpts& opts) {
  const FieldDev fd = to_dev(*f);
  const RaysDev rd = to_dev(*rays);
  const hipStream_t st = (hipStream_t)stream;
  const int L = f->grid.num_levels, F = f->grid.n_features, H = f->geo.hidden_dim;
  const bool half = f->grid.param_dtype == 1;
  const bool defer = COMPOSITE && opts.variant == 3;
#define CASE(L_, F_, H_)                                                                                              \
  if (L == L_ && F == F_ && H == H_) {                                                                                \
    if constexpr (COMPOSITE) {                                                                                        \
      if (defer)                                                                                                      \
        return half ? launch_pipelined<L_, F_, H_, true, true, true>(fd, rd, of, od, oa, ow, os, oal, sv, opts.stop_eps, st)   \
                    : launch_pipelined<L_, F_, H_, false, true, true>(fd, rd, of, od, oa, ow, os, oal, sv, opts.stop_eps, st); \
    }                                                                                                                 \
    return half ? launch_pipelined<L_, F_, H_, true, COMPOSITE, false>(fd, rd, of, od, oa, ow, os, oal, sv, opts.stop_eps, st) \
                : launch_pipelined<L_, F_, H_, false, COMPOSITE, false>(fd, rd, of, od, oa, ow, os, oal, sv, opts.stop_eps, st); \
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

template int dispatch_render_pipelined<true>(const nrhip_field*, const nrhip_rays*, float*, float*, float*, float*, float*,
                                             float*, void*, const SaveDev&, const RenderOpts&);
template int dispatch_render_pipelined<false>(const nrhip_field*, const nrhip_rays*, float*, float*, float*, float*,
                                              float*, float*, void*, const SaveDev&, const RenderOpts&);

}  // namespace nrhip
