// Shared device/host helpers for libneurad_hip.so (gfx950 only).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/neurad_hip.h"

namespace nrhip {

void set_error(const char* fmt, ...);
int check_launch(const char* what);

// A/B switches of the library (NRHIP_* environment variables): read ONCE, when the library is loaded, into this struct --
// no getenv on any dispatch path.  A process that flips a switch afterwards (the parity tests, scripts/final_measure.sh's
// alternating runs) calls nrhip_tuning_reload().  Every switch selects between two implementations of the SAME result.
struct Tuning {
  int bin_round_log2;        // NRHIP_BIN_ROUND_LOG2 in 15..24: samples per round of the binned table gradient; default 23
  int bin_pairs;             // NRHIP_BIN_PAIRS: 1 "all", 0 "none", -1 unset (x-pair records at F = 1 only)
  bool bin_transpose;        // NRHIP_BIN_TRANSPOSE != "0": sample-index-major walk of coherent chunks (default on)
  bool bin_stats;            // NRHIP_BIN_STATS: print the record count of every round (synchronises; diagnostic)
  bool multi_bwd_runs;       // NRHIP_MULTI_BWD_RUNS != "0": run combining in the actor-grid gradient (default on)
  bool pair_bwd_runs;        // NRHIP_PAIR_BWD_RUNS != "0": run combining in the trajectory gradient (default on)
  bool mlp_generic;          // NRHIP_MLP_GENERIC: per-layer MLP kernels instead of the chained ones
  bool mlp_split_wgrad;      // NRHIP_MLP_SPLIT_WGRAD: weight gradients outside the chained backward
  bool mlp_split_bf16;       // NRHIP_MLP_SPLIT_BF16: 3-way bf16 split products in the composited render kernels
  int mlp_pairs;             // NRHIP_MLP_PAIRS: 1 / 0, -1 unset (fp16-pair products: the default of the composited kernels)
  bool mlp_pairs_train;      // NRHIP_MLP_PAIRS_TRAIN == "1": pair products in the per-sample (training forward) kernel too
  bool sampler_actor_inline; // NRHIP_SAMPLER_ACTOR_INLINE == "1": per-chunk in-box lookup in the fused sampler
  bool sdf_render_pair;      // NRHIP_SDF_RENDER_PAIR == "1": two rays per wave in sdf_render_fwd/bwd
};
const Tuning& tuning();

#define NR_REQUIRE(cond, code, ...)  \
  do {                               \
    if (!(cond)) {                   \
      ::nrhip::set_error(__VA_ARGS__); \
      return (code);                 \
    }                                \
  } while (0)

constexpr uint32_t kPrimeY = 2654435761u;  // encodings.py:419
constexpr uint32_t kPrimeZ = 805459861u;

// Device-side copy of nrhip_grid (kernel argument, by value).
struct GridDev {
  int L, F, log2T, dtype;
  float scal[NRHIP_MAX_LEVELS];
};

inline GridDev to_dev(const nrhip_grid& g) {
  GridDev d;
  d.L = g.num_levels;
  d.F = g.n_features;
  d.log2T = g.log2_table_size;
  d.dtype = g.param_dtype;
  for (int i = 0; i < NRHIP_MAX_LEVELS; ++i) d.scal[i] = i < g.num_levels ? g.scalings[i] : 0.f;
  return d;
}

int validate_grid(const nrhip_grid* g);

struct RaysDev {
  int64_t R;
  int S;
  int stride;  // row stride of starts/ends
  const float* o;
  const float* d;
  const float* area;
  const float* starts;
  const float* ends;
  const int32_t* order;  // optional processing order (permutation of the rays), honoured by the fused kernels
};
inline RaysDev to_dev(const nrhip_rays& r) {
  return RaysDev{r.n_rays, r.n_samples, r.sample_stride > 0 ? r.sample_stride : r.n_samples, r.origins, r.directions,
                 r.pixel_area, r.starts, r.ends, r.order};
}
int validate_rays(const nrhip_rays* r);

// ---------------------------------------------------------------------------------------------
// Table element loads: F features of one entry, fp32 or fp16 storage, one vector load each.
// ---------------------------------------------------------------------------------------------
template <int F, bool HALF>
struct Entry;

template <int F>
struct Entry<F, false> {
  static __device__ __forceinline__ void load(const void* table, uint32_t row, float (&v)[F]) {
    // 32-bit byte offset from a wave-uniform base -> global_load with an SGPR base + one VGPR offset
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(table) + row * (uint32_t)(F * 4));
    if constexpr (F == 1) {
      v[0] = *p;
    } else if constexpr (F == 2) {
      float2 t = *reinterpret_cast<const float2*>(p);
      v[0] = t.x, v[1] = t.y;
    } else if constexpr (F == 4) {
      float4 t = *reinterpret_cast<const float4*>(p);
      v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
    } else {
      static_assert(F == 8, "F in {1,2,4,8}");
      float4 a = *reinterpret_cast<const float4*>(p);
      float4 b = *reinterpret_cast<const float4*>(p + 4);
      v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    }
  }
};

template <int F>
struct Entry<F, true> {
  static __device__ __forceinline__ void load(const void* table, uint32_t row, float (&v)[F]) {
    const __half* p =
        reinterpret_cast<const __half*>(reinterpret_cast<const char*>(table) + row * (uint32_t)(F * 2));
    if constexpr (F == 1) {
      v[0] = __half2float(*p);
    } else if constexpr (F == 2) {
      float2 t = __half22float2(*reinterpret_cast<const __half2*>(p));
      v[0] = t.x, v[1] = t.y;
    } else if constexpr (F == 4) {
      uint2 raw = *reinterpret_cast<const uint2*>(p);
      float2 a = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
      float2 b = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
      v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
    } else {
      static_assert(F == 8, "F in {1,2,4,8}");
      uint4 raw = *reinterpret_cast<const uint4*>(p);
      const __half2* h = reinterpret_cast<const __half2*>(&raw);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float2 t = __half22float2(h[i]);
        v[2 * i] = t.x, v[2 * i + 1] = t.y;
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// One hash-grid level: HashEncoding.pytorch_fwd for a single (point, level)
// (encodings.py:425-464).  x in [0,1]^3, `scale` = scalings[l], table_l = base row of the level.
// The lerp tree and corner pairing follow the reference exactly (f03/f12/f56/f47 -> f0312/f4756).
// ---------------------------------------------------------------------------------------------
struct Corners {
  uint32_t idx[8];  // reference corner order 0..7 (encodings.py:437-444), already masked (no level offset)
  float ox, oy, oz;
};

__device__ __forceinline__ Corners hash_corners(float x, float y, float z, float scale, uint32_t mask) {
  Corners c;
  // __fmul_rn / __fsub_rn: the product must be ROUNDED before floor/ceil and before the offset is taken,
  // exactly like torch's separate mul and sub.  Letting the compiler contract x*scale - floor(..) into an
  // fma changes the offset by up to half an ulp of sx (~5e-4 at scale 8192) -- measured 7e-5 rel-L2.
  const float sx = __fmul_rn(x, scale), sy = __fmul_rn(y, scale), sz = __fmul_rn(z, scale);
  const float fx = floorf(sx), fy = floorf(sy), fz = floorf(sz);
  const uint32_t ifx = (uint32_t)(int)fx, ify = (uint32_t)(int)fy, ifz = (uint32_t)(int)fz;
  const uint32_t icx = (uint32_t)(int)ceilf(sx), icy = (uint32_t)(int)ceilf(sy), icz = (uint32_t)(int)ceilf(sz);
  c.ox = __fsub_rn(sx, fx), c.oy = __fsub_rn(sy, fy), c.oz = __fsub_rn(sz, fz);
  // int32*int64 products in the reference never wrap in 64 bit; their low log2(T) bits equal the
  // uint32-wrapped products (T is a power of two), verified in tests against exact golden indices.
  const uint32_t hyc = icy * kPrimeY, hyf = ify * kPrimeY, hzc = icz * kPrimeZ, hzf = ifz * kPrimeZ;
  c.idx[0] = (icx ^ hyc ^ hzc) & mask;  // c c c
  c.idx[1] = (icx ^ hyf ^ hzc) & mask;  // c f c
  c.idx[2] = (ifx ^ hyf ^ hzc) & mask;  // f f c
  c.idx[3] = (ifx ^ hyc ^ hzc) & mask;  // f c c
  c.idx[4] = (icx ^ hyc ^ hzf) & mask;  // c c f
  c.idx[5] = (icx ^ hyf ^ hzf) & mask;  // c f f
  c.idx[6] = (ifx ^ hyf ^ hzf) & mask;  // f f f
  c.idx[7] = (ifx ^ hyc ^ hzf) & mask;  // f c f
  return c;
}

// trilinear blend of the 8 corner entries, reference order of operations (encodings.py:446-464)
template <int F>
__device__ __forceinline__ void lerp_corners(const Corners& c, const float (&f)[8][F], float (&out)[F]) {
  const float ox = c.ox, oy = c.oy, oz = c.oz;
  const float mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
#pragma unroll
  for (int i = 0; i < F; ++i) {
    // a*o + b*(1-o) as fma(a, o, b*(1-o)): 2 VALU ops per lerp; one rounding fewer than torch's mul,mul,add
    const float f03 = fmaf(f[0][i], ox, f[3][i] * mx);
    const float f12 = fmaf(f[1][i], ox, f[2][i] * mx);
    const float f56 = fmaf(f[5][i], ox, f[6][i] * mx);
    const float f47 = fmaf(f[4][i], ox, f[7][i] * mx);
    const float f0312 = fmaf(f03, oy, f12 * my);
    const float f4756 = fmaf(f47, oy, f56 * my);
    out[i] = fmaf(f0312, oz, f4756 * mz);
  }
}

// F = 1: the (floor x, ceil x) corners of one (y, z) differ only through ifx ^ icx -- the same value for all four such pairs
// of a sample.  When it is 0 or 1 (x integral, or floor(x) even: half of the samples) both entries lie in one aligned
// 2-entry pair and ONE 8-byte (fp16: 4-byte) load serves both; the other lanes fetch their ceil corners with a second,
// exec-masked load.  Same bytes, same results; the L1 sees 6 line accesses per (sample, level) on average instead of 8,
// and the level-partitioned F = 1 kernels (every access an L2 hit) run against exactly that rate.  row0 must be even.
template <bool HALF>
__device__ __forceinline__ void load_corners_f1(const void* table, uint32_t row0, const Corners& c, float (&f)[8][1]) {
  constexpr int kF[4] = {3, 2, 7, 6}, kC[4] = {0, 1, 4, 5};  // (f, c) corner of the pairs (y, z) = cc, fc, cf, ff
  if constexpr (!HALF) {
    // fp32 (round 4): ONE aligned 16-byte load per (y, z) where floor x ^ ceil x is 0, 1 or 3 -- three quarters of the
    // samples: both entries then lie in one aligned 4-entry block.  5 line accesses per (sample, level) instead of 6; the L1
    // charges per line, not per byte (scripts/probes/l1_coalesce_probe.hip).  row0 must be a multiple of 4.
    const bool same = ((c.idx[3] ^ c.idx[0]) >> 2) == 0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const uint32_t rf = row0 + c.idx[kF[p]], rc = row0 + c.idx[kC[p]];
      const float4 t = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(table) + (rf & ~3u) * 4u);
      const uint32_t a = rf & 3u, b = rc & 3u;
      f[kF[p]][0] = a == 0u ? t.x : (a == 1u ? t.y : (a == 2u ? t.z : t.w));
      f[kC[p]][0] = b == 0u ? t.x : (b == 1u ? t.y : (b == 2u ? t.z : t.w));  // meaningful where `same`
    }
    if (!same) {
#pragma unroll
      for (int p = 0; p < 4; ++p) Entry<1, false>::load(table, row0 + c.idx[kC[p]], f[kC[p]]);
    }
    return;
  }
  const bool same = ((c.idx[3] ^ c.idx[0]) >> 1) == 0;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const uint32_t rf = row0 + c.idx[kF[p]], rc = row0 + c.idx[kC[p]];
    float lo, hi;
    if constexpr (HALF) {
      const float2 t = __half22float2(*reinterpret_cast<const __half2*>(reinterpret_cast<const char*>(table) + (rf & ~1u) * 2u));
      lo = t.x, hi = t.y;
    } else {
      const float2 t = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(table) + (rf & ~1u) * 4u);
      lo = t.x, hi = t.y;
    }
    f[kF[p]][0] = (rf & 1u) ? hi : lo;
    f[kC[p]][0] = (rc & 1u) ? hi : lo;  // meaningful where `same`
  }
  if (!same) {
#pragma unroll
    for (int p = 0; p < 4; ++p) Entry<1, HALF>::load(table, row0 + c.idx[kC[p]], f[kC[p]]);
  }
}

// PAIRED (F = 1 only): load_corners_f1.  For kernels whose threads do ONE level each (proposal_levels_lp: 286 -> 265 us on
// 2.1 M samples x 6 levels); where a thread walks all levels the exec-masked second load turns one memory round trip
// into one per level and loses (409 -> 492 us), so it is opt-in.
template <int F, bool HALF, bool PAIRED = false>
__device__ __forceinline__ void hash_level(const void* table, uint32_t level_row0, float x, float y, float z,
                                           float scale, uint32_t mask, float (&out)[F]) {
  const Corners c = hash_corners(x, y, z, scale, mask);
  float f[8][F];
  if constexpr (F == 1 && PAIRED) {
    load_corners_f1<HALF>(table, level_row0, c, f);
  } else {
#pragma unroll
    for (int k = 0; k < 8; ++k) Entry<F, HALF>::load(table, level_row0 + c.idx[k], f[k]);
  }
  lerp_corners<F>(c, f, out);
}

// Trilinear corner weights in the same corner order (for the scatter-add backward).
__device__ __forceinline__ void corner_weights(const Corners& c, float (&w)[8]) {
  const float ox = c.ox, oy = c.oy, oz = c.oz, mx = 1.f - ox, my = 1.f - oy, mz = 1.f - oz;
  w[0] = ox * oy * oz;
  w[1] = ox * my * oz;
  w[2] = mx * my * oz;
  w[3] = mx * oy * oz;
  w[4] = ox * oy * mz;
  w[5] = ox * my * mz;
  w[6] = mx * my * mz;
  w[7] = mx * oy * mz;
}

// ---------------------------------------------------------------------------------------------
// H2 + H3: sample -> gaussian (mean,std) -> contracted position in [0,1]^3 and contracted std
// (cameras/rays.py:109-124, M=1; spatial_distortions.py:103-113,126-141, order=inf)
// ---------------------------------------------------------------------------------------------
struct SamplePos {
  float x, y, z, std;
};

// x^(1/3) for x >= 0 via v_log_f32 / v_exp_f32 (~1e-6 relative).  Only the gaussian std (which feeds the smooth
// down-weighting 1/max(1, 2*s_l*std)) goes through it -- never a position -- so torch's pow(x, 1/3) need not be
// matched to the ulp; a full powf costs ~40 VALU instructions twice per sample.
__device__ __forceinline__ float fast_cbrt(float x) {
  return __builtin_amdgcn_exp2f(__builtin_amdgcn_logf(x) * (1.0f / 3.0f));
}

// H2 alone: gaussian mean (world space) and std of a sample (cameras/rays.py:109-124, M=1)
__device__ __forceinline__ SamplePos sample_gaussian(float ox, float oy, float oz, float dx, float dy, float dz,
                                                     float area, float t0, float t1) {
#pragma clang fp contract(off)
  const float dist = (t1 - t0) / 2.f;
  const float t = t0 + 1.f * dist;
  SamplePos p;
  p.x = ox + dx * t, p.y = oy + dy * t, p.z = oz + dz * t;
  p.std = fast_cbrt((area * (t * t)) * dist);
  return p;
}

// H3 alone: ScaledSceneContraction(order=inf) of a gaussian -> [0,1]^3 (spatial_distortions.py:103-141)
__device__ __forceinline__ SamplePos contract_gaussian(float mx, float my, float mz, float std, float scale) {
#pragma clang fp contract(off)
  mx /= scale, my /= scale, mz /= scale, std /= scale;
  const float mag = fmaxf(fabsf(mx), fmaxf(fabsf(my), fabsf(mz)));
  if (!(mag < 1.f)) {
    const float cm = fmaxf(mag, 1.f);
    const float k = 2.f - (1.f / cm);
    mx = k * (mx / cm), my = k * (my / cm), mz = k * (mz / cm);
    const float sc = fast_cbrt(2.f * cm - 1.f) / cm;
    std = std * (sc * sc);
  }
  SamplePos p;
  p.x = (mx + 2.f) / 4.f, p.y = (my + 2.f) / 4.f, p.z = (mz + 2.f) / 4.f;
  p.std = std / 4.f;
  return p;
}

__device__ __forceinline__ SamplePos sample_position(float ox, float oy, float oz, float dx, float dy, float dz,
                                                     float area, float t0, float t1, float scale) {
  // Position arithmetic mirrors torch op for op (separately rounded mul/add/div, no fma): one ulp of the
  // contracted coordinate is 8192 * 6e-8 = 5e-4 of a cell at the finest level, so rounding-order matters.
#pragma clang fp contract(off)
  const float dist = (t1 - t0) / 2.f;
  const float t = t0 + 1.f * dist;
  float mx = ox + dx * t, my = oy + dy * t, mz = oz + dz * t;
  float std = fast_cbrt((area * (t * t)) * dist);
  // ScaledSceneContraction: divide by scale, contract (inf-norm), map [-2,2] -> [0,1]
  mx /= scale, my /= scale, mz /= scale, std /= scale;
  const float mag = fmaxf(fabsf(mx), fmaxf(fabsf(my), fabsf(mz)));
  if (!(mag < 1.f)) {
    const float cm = fmaxf(mag, 1.f);
    const float k = 2.f - (1.f / cm);
    mx = k * (mx / cm), my = k * (my / cm), mz = k * (mz / cm);
    const float sc = fast_cbrt(2.f * cm - 1.f) / cm;
    std = std * (sc * sc);
  }
  SamplePos p;
  p.x = (mx + 2.f) / 4.f, p.y = (my + 2.f) / 4.f, p.z = (mz + 2.f) / 4.f;
  p.std = std / 4.f;
  return p;
}

// H4: 1 / max(1, 2*scalings_l*std)   (neurad_encoding.py:302)
__device__ __forceinline__ float rescale_weight(float scale_l, float std) {
  return __builtin_amdgcn_rcpf(fmaxf(scale_l * 2.f * std, 1.f));  // v_rcp_f32 (1 ulp); weights are smooth in std
}

// F3: 16 real SH components (utils/math.py:31-94) of a direction given as (dir+1)/2 -- the torch path
// evaluates the polynomials on that [0,1]-normalised vector directly (base_field.py:136-142).
__device__ __forceinline__ void sh4(float x, float y, float z, float (&c)[16]) {
  const float xx = x * x, yy = y * y, zz = z * z;
  c[0] = 0.28209479177387814f;
  c[1] = 0.4886025119029199f * y;
  c[2] = 0.4886025119029199f * z;
  c[3] = 0.4886025119029199f * x;
  c[4] = 1.0925484305920792f * x * y;
  c[5] = 1.0925484305920792f * y * z;
  c[6] = 0.9461746957575601f * zz - 0.31539156525251999f;
  c[7] = 1.0925484305920792f * x * z;
  c[8] = 0.5462742152960396f * (xx - yy);
  c[9] = 0.5900435899266435f * y * (3.f * xx - yy);
  c[10] = 2.890611442640554f * x * y * z;
  c[11] = 0.4570457994644658f * y * (5.f * zz - 1.f);
  c[12] = 0.3731763325901154f * z * (5.f * zz - 3.f);
  c[13] = 0.4570457994644658f * x * (5.f * zz - 1.f);
  c[14] = 1.445305721320277f * z * (xx - yy);
  c[15] = 0.5900435899266435f * x * (xx - 3.f * yy);
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + __expf(-x)); }

// DPP helpers over rows of 16 lanes -------------------------------------------------------------
// row_shr:n  = 0x110+n ; bound_ctrl=false keeps `old` for lanes shifted in from outside the row.
template <int N>
__device__ __forceinline__ float dpp_row_shr(float v, float fill) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x110 + N, 0xf,
                                         0xf, false));
}

// row_shl:n = 0x100+n : lane i reads lane i+n of its row; lanes whose source is outside the row get `fill`.
template <int N>
__device__ __forceinline__ float dpp_row_shl(float v, float fill) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, fill), __builtin_bit_cast(int, v), 0x100 + N, 0xf,
                                         0xf, false));
}
template <int N>
__device__ __forceinline__ uint32_t dpp_row_shl(uint32_t v, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x100 + N, 0xf, 0xf, false);
}
template <int N>
__device__ __forceinline__ uint32_t dpp_row_shr(uint32_t v, uint32_t fill) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x110 + N, 0xf, 0xf, false);
}

inline int grid_for(int64_t threads, int block) { return (int)((threads + block - 1) / block); }

// --------------------------------------------------------------------------------------------
// fp32 MFMA building block shared by the fused render kernel and the chained MLP kernels.
using f32x4 = __attribute__((ext_vector_type(4))) float;

// One MFMA layer on a 16-sample tile, weights as the A operand read from LDS in fragment order
// [mb][s4][lane][s3] (one conflict-free ds_read_b128 per 4 k-steps), samples as the B operand in registers:
// acc[mb] += Σ_s A[mb][s] * B[s],  NS k-steps, NBLK output blocks, all statically unrolled.
template <int NBLK, int NS>
__device__ __forceinline__ void mfma_layer(const float* __restrict__ wf, int lane, const float (&b)[NS],
                                           f32x4 (&acc)[NBLK]) {
#pragma unroll
  for (int s4 = 0; s4 < NS / 4; ++s4) {
    f32x4 a[NBLK];
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb)
      a[mb] = *reinterpret_cast<const f32x4*>(wf + ((mb * (NS / 4) + s4) * 64 + lane) * 4);
#pragma unroll
    for (int s3 = 0; s3 < 4; ++s3)
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[mb][s3], b[4 * s4 + s3], acc[mb], 0, 0, 0);
  }
}


}  // namespace nrhip
