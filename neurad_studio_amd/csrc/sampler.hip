// S1 / S4 / S5+M1: PowerSampler bins, PDF resampling and the fused proposal sampler.
// One wavefront marches one ray; bin edges, the CDF and the weights of a round live in a per-wave LDS
// slab, the exclusive sum scans are wave shuffles, the inverse-CDF lookups are per-lane binary searches
// in LDS (the 65 / 33 query points of a round fit one / two passes of the 64 lanes).
#include "rayorder.h"
#include "wave_scan.h"

namespace nrhip {

constexpr int kSMax = 512;  // max samples per ray per round handled by the LDS slabs

// ZipNeRF power transform (utils/math.py:541-579) as used by PowerSampler (ray_samplers.py:838-852)
// The reference raises with torch.pow(tensor, python float), and ATen evaluates the exponent -1 -- NeuRAD's lambda, and its
// own inverse -- as the RECIPROCAL (pow_tensor_scalar: reciprocal kernel; torch.pow(x, -1.0) == 1 / x bit for bit, checked in
// the build container).  So does this: one IEEE division instead of a generic powf, which was what bound the kernels below
// (one powf per bin edge).
__device__ __forceinline__ float pow_like_aten(float base, float e) { return e == -1.f ? 1.f / base : powf(base, e); }
__device__ __forceinline__ float power_fn(float x, float lam) {
  if (lam == 1.f) return x;
  if (lam == 0.f) return log1pf(x);
  const float lam_1 = fabsf(lam - 1.f);
  return (lam_1 / lam) * (pow_like_aten(x / lam_1 + 1.f, lam) - 1.f);
}
__device__ __forceinline__ float inv_power_fn(float x, float lam) {
  if (lam == 1.f) return x;
  if (lam == 0.f) return expm1f(x);
  const float lam_1 = fabsf(lam - 1.f);
  return (pow_like_aten(fmaxf(x * lam / lam_1 + 1.f, 1e-10f), 1.f / lam) - 1.f) * lam_1;
}
struct Spacing {
  float s_near, s_far, lam, scaling;
  __device__ __forceinline__ float to_euclid(float b) const {
    return inv_power_fn(b * s_far + (1.f - b) * s_near, lam) / scaling;
  }
};
__device__ __forceinline__ Spacing make_spacing(float near, float far, float lam, float scaling) {
  return Spacing{power_fn(near * scaling, lam), power_fn(far * scaling, lam), lam, scaling};
}

// torch.linspace(start, end, steps)[i] in fp32 (symmetric evaluation used by ATen)
__device__ __forceinline__ float linspace_at(float start, float end, int steps, int i) {
  if (steps == 1) return start;
  const float step = (end - start) / (float)(steps - 1);
  return i < steps / 2 ? start + step * (float)i : end - step * (float)(steps - 1 - i);
}

// S1 for one ray: edges k = lane, lane+64, ... of the S+1 bins (ray_samplers.py:100-119)
__device__ __forceinline__ float power_bin(int k, int S, const float* t_rand_row) {
  float b = linspace_at(0.f, 1.f, S + 1, k);
  if (t_rand_row) {
    const float bm = k > 0 ? linspace_at(0.f, 1.f, S + 1, k - 1) : b;
    const float bp = k < S ? linspace_at(0.f, 1.f, S + 1, k + 1) : b;
    const float upper = k < S ? (bp + b) / 2.f : b;
    const float lower = k > 0 ? (b + bm) / 2.f : b;
    b = lower + (upper - lower) * t_rand_row[k];
  }
  return b;
}

// bin edge t of the flattened [R, S+1] arrays
__device__ __forceinline__ void power_sampler_bin(const float* __restrict__ nears, const float* __restrict__ fars,
                                                  int64_t t, int S, float lam, float scaling,
                                                  const float* __restrict__ t_rand, float last_edge,
                                                  float* __restrict__ sp, float* __restrict__ eu) {
  const int64_t ray = t / (S + 1);
  const int k = (int)(t - ray * (S + 1));
  const Spacing spc = make_spacing(nears ? nears[ray] : 0.f, fars[ray], lam, scaling);
  const float b = power_bin(k, S, t_rand ? t_rand + ray * (S + 1) : nullptr);
  sp[t] = b;
  // last_edge > 0: the model's sky stretch (models/neurad.py:451-455, frustums.ends[:, -1] = sky_distance) folded in
  eu[t] = (k == S && last_edge > 0.f) ? last_edge : spc.to_euclid(b);
}

// A workgroup owns kRaysPerSamplerBlock consecutive rays.  The two power_fn evaluations that map a ray's near / far plane into
// the sampler's spacing (two powf per RAY) are done once, by thread r for ray r, and parked in LDS; the (S + 1) edges per ray
// then cost one powf each.  Evaluated per edge, as power_sampler_bin does, the spacing is 2/3 of the kernel's arithmetic, and the
// kernel is bound by it (the c3 step: 57 344 rays x 129 edges, 109 us).  Same expressions on the same inputs: bit-identical bins.
constexpr int kRaysPerSamplerBlock = 64;
__global__ __launch_bounds__(256) void power_sampler_kernel(const float* __restrict__ nears,
                                                             const float* __restrict__ fars, int64_t R, int S,
                                                             float lam, float scaling,
                                                             const float* __restrict__ t_rand, float last_edge,
                                                             float* __restrict__ sp, float* __restrict__ eu) {
  __shared__ float s_near[kRaysPerSamplerBlock], s_far[kRaysPerSamplerBlock];
  const int64_t ray0 = (int64_t)blockIdx.x * kRaysPerSamplerBlock;
  const int n_rays = (int)min((int64_t)kRaysPerSamplerBlock, R - ray0);
  if ((int)threadIdx.x < n_rays) {
    const int64_t ray = ray0 + threadIdx.x;
    const Spacing spc = make_spacing(nears ? nears[ray] : 0.f, fars[ray], lam, scaling);
    s_near[threadIdx.x] = spc.s_near;
    s_far[threadIdx.x] = spc.s_far;
  }
  __syncthreads();
  const int E = S + 1, n_edges = n_rays * E;
  for (int e = threadIdx.x; e < n_edges; e += 256) {
    const int r = e / E, k = e - r * E;
    const int64_t t = ray0 * E + e;
    const Spacing spc{s_near[r], s_far[r], lam, scaling};
    const float b = power_bin(k, S, t_rand ? t_rand + (ray0 + r) * E : nullptr);
    sp[t] = b;
    eu[t] = (k == S && last_edge > 0.f) ? last_edge : spc.to_euclid(b);
  }
}

// The same bins plus the processing order of the rays (rayorder.h) in ONE launch: workgroup 0 runs the single-workgroup
// counting sort while all the others fill bins -- the ordering pass is a latency chain on one CU (5.7 us as a launch of
// its own, which would sit in front of the render kernel) and hides behind the bins of a chip-wide kernel.
template <int BITS>
__global__ __launch_bounds__(kOrderThreads) void power_sampler_order_kernel(
    const float* __restrict__ nears, const float* __restrict__ fars, int64_t R, int S, float lam, float scaling,
    const float* __restrict__ t_rand, float last_edge, float* __restrict__ sp, float* __restrict__ eu,
    const float* __restrict__ o, const float* __restrict__ d, float t_ref, float scale, int32_t* __restrict__ order) {
  if (blockIdx.x == 0) {
    ray_order_body<BITS>(o, d, R, t_ref, scale, order);
    return;
  }
  // the block's 1024 consecutive edges belong to a handful of rays: their spacing (two powf per RAY) once, in LDS, as in
  // power_sampler_kernel -- evaluated per edge it was 2/3 of the arithmetic, and the bins, not the ordering chain (5.7 us
  // alone), set this launch's 15 us in front of the render kernel.  Same expressions on the same inputs: bit-identical bins.
  __shared__ float s_near[kOrderThreads], s_far[kOrderThreads];
  const int E = S + 1;
  const int64_t t0 = ((int64_t)blockIdx.x - 1) * kOrderThreads, n_edges = R * E;
  const int64_t ray_first = t0 / E, t_last = min(t0 + kOrderThreads, n_edges) - 1;
  const int n_rays = (int)(t_last / E - ray_first) + 1;  // <= kOrderThreads (E >= 1)
  if ((int)threadIdx.x < n_rays) {
    const int64_t ray = ray_first + threadIdx.x;
    const Spacing spc = make_spacing(nears ? nears[ray] : 0.f, fars[ray], lam, scaling);
    s_near[threadIdx.x] = spc.s_near;
    s_far[threadIdx.x] = spc.s_far;
  }
  __syncthreads();
  const int64_t t = t0 + threadIdx.x;
  if (t < n_edges) {
    const int64_t ray = t / E;
    const int k = (int)(t - ray * E), r = (int)(ray - ray_first);
    const Spacing spc{s_near[r], s_far[r], lam, scaling};
    const float b = power_bin(k, S, t_rand ? t_rand + ray * E : nullptr);
    sp[t] = b;
    eu[t] = (k == S && last_edge > 0.f) ? last_edge : spc.to_euclid(b);
  }
}

// wave-wide sum / scan on DPP row shifts + readlane (wave_scan.h), not on ds_bpermute shuffles
__device__ __forceinline__ float wsum(float v) { return wscan::reduce<wscan::Add>(v); }
__device__ __forceinline__ float wscan_add(float v, int lane) { return wscan::incl<wscan::Add>(v, lane); }
__device__ __forceinline__ void wave_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// S4 for one ray (ray_samplers.py:306-366).  w_lds[Sp] raw weights, bins_lds[Sp+1] existing spacing bins,
// cdf_lds[Sp+1] scratch.  Writes the Sn+1 new spacing bins to new_bins (LDS or global) and, if eu_out, the
// euclidean bins.  rand_row: NULL (eval) / jitter values, rand_stride 0 = single jitter.
// cdf_lds[0 .. Sp] = min(1, cumsum of the padded, normalised weights), cdf[0] = 0 (ray_samplers.py:318-331); one wave
__device__ __forceinline__ void pdf_build_cdf(const float* w_lds, float* cdf_lds, int Sp, float pad, int lane) {
  constexpr float eps = 1e-5f;
  float tot = 0.f;
  for (int k = lane; k < Sp; k += 64) tot += w_lds[k] + pad;
  tot = wsum(tot);
  const float padding = fmaxf(eps - tot, 0.f);
  const float add = padding / (float)Sp;
  tot += padding;
  float carry = 0.f;
  for (int k0 = 0; k0 < Sp; k0 += 64) {
    const int k = k0 + lane;
    const float pdf = k < Sp ? ((w_lds[k] + pad) + add) / tot : 0.f;
    const float incl = wscan_add(pdf, lane);
    if (k < Sp) cdf_lds[k + 1] = fminf(1.f, carry + incl);
    carry += wscan::last(incl);
  }
  if (lane == 0) cdf_lds[0] = 0.f;
}

// new spacing bin i of nb = Sn + 1 (ray_samplers.py:333-366): inverse CDF at u_i
__device__ __forceinline__ float pdf_sample_bin(const float* cdf_lds, const float* bins_lds, int Sp, int nb, int i,
                                                const float* rand_row, int rand_stride) {
  float u = linspace_at(0.f, 1.f - (1.f / (float)nb), nb, i);
  if (rand_row) u += rand_row[rand_stride ? i : 0] / (float)nb;
  else u += 1.f / (float)(2 * nb);
  // searchsorted(cdf, u, side="right") over Sp+1 entries = #{cdf <= u}
  int lo = 0, hi = Sp + 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (cdf_lds[mid] <= u) lo = mid + 1;
    else hi = mid;
  }
  const int below = min(max(lo - 1, 0), Sp), above = min(max(lo, 0), Sp);
  const float c0 = cdf_lds[below], c1 = cdf_lds[above];
  const float b0 = bins_lds[below], b1 = bins_lds[above];
  float tt = (u - c0) / (c1 - c0);
  if (tt != tt) tt = 0.f;  // nan_to_num(nan=0); +-inf are clipped below
  tt = fminf(fmaxf(tt, 0.f), 1.f);
  return b0 + tt * (b1 - b0);
}

// S4 for one ray (ray_samplers.py:306-366).  w_lds[Sp] raw weights, bins_lds[Sp+1] existing spacing bins,
// cdf_lds[Sp+1] scratch.  Writes the Sn+1 new spacing bins to new_bins (LDS or global) and, if eu_out, the
// euclidean bins.  rand_row: NULL (eval) / jitter values, rand_stride 0 = single jitter.
__device__ __forceinline__ void pdf_resample_ray(const float* w_lds, const float* bins_lds, float* cdf_lds, int Sp,
                                                 int Sn, float pad, const float* rand_row, int rand_stride,
                                                 const Spacing& spc, float* new_bins, float* eu_out, int lane) {
  pdf_build_cdf(w_lds, cdf_lds, Sp, pad, lane);
  wave_fence();
  const int nb = Sn + 1;
  for (int i = lane; i < nb; i += 64) {
    const float nbv = pdf_sample_bin(cdf_lds, bins_lds, Sp, nb, i, rand_row, rand_stride);
    new_bins[i] = nbv;
    if (eu_out) eu_out[i] = spc.to_euclid(nbv);
  }
}

// Standalone S4 (the training path's rounds): a workgroup = 4 rays.  Each wave builds its ray's CDF; the 4 (Sn + 1) new bins
// are then spread over all 256 threads -- with one wave per ray, the odd bin (Sn + 1 = 65, 33) costs that wave a second pass
// of searches and a powf for a single lane.
__global__ __launch_bounds__(256) void pdf_sample_kernel(const float* __restrict__ weights,
                                                          const float* __restrict__ bins,
                                                          const float* __restrict__ nears,
                                                          const float* __restrict__ fars, int64_t R, int Sp, int Sn,
                                                          float lam, float scaling, float pad,
                                                          const float* __restrict__ rand, int rand_stride,
                                                          float* __restrict__ new_sp, float* __restrict__ new_eu) {
  // three arrays of Sp + 1 floats per ray, sized for THIS launch: at the kSMax the four slabs (24 KB) held the CU at 6
  // workgroups = 24 of its 32 waves
  extern __shared__ float slab_dyn[];
  const int slab_len = 3 * (Sp + 1);
  __shared__ float s_near[4], s_far[4];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t ray0 = (int64_t)blockIdx.x * 4;
  const int n_rays = (int)min((int64_t)4, R - ray0);
  if (wid < n_rays) {
    const int64_t ray = ray0 + wid;
    float* w_lds = slab_dyn + wid * slab_len;
    float* b_lds = w_lds + Sp + 1;
    float* c_lds = b_lds + Sp + 1;
    for (int k = lane; k < Sp; k += 64) w_lds[k] = weights[ray * Sp + k];
    for (int k = lane; k <= Sp; k += 64) b_lds[k] = bins[ray * (Sp + 1) + k];
    wave_fence();
    pdf_build_cdf(w_lds, c_lds, Sp, pad, lane);
    if (lane == 0) {
      const Spacing spc = make_spacing(nears ? nears[ray] : 0.f, fars[ray], lam, scaling);
      s_near[wid] = spc.s_near;
      s_far[wid] = spc.s_far;
    }
  }
  __syncthreads();
  const int nb = Sn + 1;
  for (int j = threadIdx.x; j < n_rays * nb; j += 256) {
    const int r = j / nb, i = j - r * nb;
    const float* w_lds = slab_dyn + r * slab_len;
    const float* b_lds = w_lds + Sp + 1;
    const float* c_lds = b_lds + Sp + 1;
    const int64_t ray = ray0 + r;
    const float nbv = pdf_sample_bin(c_lds, b_lds, Sp, nb, i, rand ? rand + ray * (rand_stride ? rand_stride : 1) : nullptr,
                                     rand_stride);
    new_sp[ray * nb + i] = nbv;
    new_eu[ray * nb + i] = Spacing{s_near[r], s_far[r], lam, scaling}.to_euclid(nbv);
  }
}

// ---------------------------------------------------------------------------------------------
// S5 + M1 fused.  Per ray: power bins -> [density (S2) -> weights (S3) -> pdf resample (S4)] x rounds.
struct PropDev {
  GridDev grid;
  const void* table;
  float scale;
  const float* dec;
};
struct PropActorDev {  // the proposal field's per-actor grids (F = 1), ACT launches only
  const void* const* tables;
  GridDev grid;
  float scale;
};
struct SamplerDev {
  int n_rounds;
  int ns[3];
  float lam, scaling, pad, sky;
  PropDev prop[2];
  float* w_out[2];
  float* sp_out[3];
  float* eu_out[3];
  PropActorDev pact[2];
  int K;  // row length of the per-ray candidate lists
};

// S2 with dynamic actors (fields/neurad_field.py:208-213 over neurad_encoding.py:150-187): a sample inside an actor's box
// takes its density from THAT actor's grid at the box-frame position, density = exp(sum_{l < La} f_l w_l dec[l]) (the
// actor features are zero-padded to the static width, so only the decoder's first La weights see them).  Called by the
// whole wave after the static densities; `ncand` (wave-uniform) > 0.  Candidate walk over scalar loads, then one
// wave-uniform pass per distinct winning actor so that the table base stays in SGPRs (as in render.hip).
__device__ __forceinline__ float actor_density_override(const PropDev& p, const PropActorDev& pa, int K, int64_t ray,
                                                        int ncand, float dens, bool live, float ox, float oy, float oz,
                                                        float dx, float dy, float dz, float area, float t0, float t1,
                                                        const int32_t* __restrict__ cand_actor,
                                                        const float* __restrict__ cand_w2b,
                                                        const float* __restrict__ bounds) {
  const SamplePos gs = sample_gaussian(ox, oy, oz, dx, dy, dz, area, t0, t1);
  const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)ray * (uint32_t)K));
  int slot = -1;
  for (int c = 0; c < ncand; ++c) {  // ascending actor index: the last containing box wins (neurad_encoding.py:184-185)
    const float* w = cand_w2b + (row + (uint32_t)c) * 12u;
    const int act = cand_actor[row + (uint32_t)c];
    const float bx = w[0] * gs.x + w[1] * gs.y + w[2] * gs.z + w[3];
    const float by = w[4] * gs.x + w[5] * gs.y + w[6] * gs.z + w[7];
    const float bz = w[8] * gs.x + w[9] * gs.y + w[10] * gs.z + w[11];
    if (fabsf(bx) < bounds[3 * act] && fabsf(by) < bounds[3 * act + 1] && fabsf(bz) < bounds[3 * act + 2]) slot = c;
  }
  unsigned long long todo = __ballot(live && slot >= 0);
  const uint32_t amask = (1u << pa.grid.log2T) - 1u;
  while (todo) {
    const int c = __builtin_amdgcn_readlane(slot, (int)__builtin_ctzll(todo));
    const bool mine = live && slot == c;
    todo &= ~__ballot(mine);
    const float* w = cand_w2b + (row + (uint32_t)c) * 12u;
    const void* tb = pa.tables[cand_actor[row + (uint32_t)c]];
    if (mine) {
      const float bx = w[0] * gs.x + w[1] * gs.y + w[2] * gs.z + w[3];
      const float by = w[4] * gs.x + w[5] * gs.y + w[6] * gs.z + w[7];
      const float bz = w[8] * gs.x + w[9] * gs.y + w[10] * gs.z + w[11];
      const SamplePos q = contract_gaussian(bx, by, bz, gs.std, pa.scale);
      float acc = 0.f;
      for (int l = 0; l < pa.grid.L; ++l) {
        float v[1];
        hash_level<1, false>(tb, (uint32_t)l << pa.grid.log2T, q.x, q.y, q.z, pa.grid.scal[l], amask, v);
        acc += (v[0] * rescale_weight(pa.grid.scal[l], q.std)) * p.dec[l];
      }
      dens = expf(acc);
    }
  }
  return dens;
}

// Round 5: the in-box samples of a ray's round as a SEPARATE, dense pass.  `actor_density_override` above (kept for
// NRHIP_SAMPLER_ACTOR_INLINE=1, A/B) looks an actor grid up inside the 64-sample chunk that met it: one wave-uniform pass per
// distinct winning actor, a handful of lanes busy, four dependent gather round trips each -- for the 5 % of the samples
// that lie in a box, on the 75 % of the rays that have a candidate.  Here the chunk loop only finds WHICH candidate
// contains a sample (`actor_slot_of`) and appends (sample, slot) to a per-wave list in LDS; after the last chunk the list is
// walked 64 entries at a time with every lane busy -- per-lane world->box rows and table base, all corner gathers of up
// to four levels in flight together -- and the densities land in the round's density slab before the weights are formed.
// Same arithmetic, same summation order: bit-identical densities.
__device__ __forceinline__ int actor_slot_of(int K, int64_t ray, int ncand, float ox, float oy, float oz, float dx, float dy,
                                             float dz, float area, float t0, float t1,
                                             const int32_t* __restrict__ cand_actor, const float* __restrict__ cand_w2b,
                                             const float* __restrict__ bounds) {
  const SamplePos gs = sample_gaussian(ox, oy, oz, dx, dy, dz, area, t0, t1);
  const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)ray * (uint32_t)K));
  int slot = -1;
  for (int c = 0; c < ncand; ++c) {  // ascending actor index: the last containing box wins (neurad_encoding.py:184-185)
    const float* w = cand_w2b + (row + (uint32_t)c) * 12u;
    const int act = cand_actor[row + (uint32_t)c];
    const float bx = w[0] * gs.x + w[1] * gs.y + w[2] * gs.z + w[3];
    const float by = w[4] * gs.x + w[5] * gs.y + w[6] * gs.z + w[7];
    const float bz = w[8] * gs.x + w[9] * gs.y + w[10] * gs.z + w[11];
    if (fabsf(bx) < bounds[3 * act] && fabsf(by) < bounds[3 * act + 1] && fabsf(bz) < bounds[3 * act + 2]) slot = c;
  }
  return slot;
}

// density of one in-box sample from its actor's grid (per-lane table base and world->box rows)
__device__ __forceinline__ float actor_density_of(const PropDev& p, const PropActorDev& pa, const float* __restrict__ w,
                                                  const void* tb, float ox, float oy, float oz, float dx, float dy, float dz,
                                                  float area, float t0, float t1) {
  const SamplePos gs = sample_gaussian(ox, oy, oz, dx, dy, dz, area, t0, t1);
  const float4 w0 = *reinterpret_cast<const float4*>(w), w1 = *reinterpret_cast<const float4*>(w + 4),
               w2 = *reinterpret_cast<const float4*>(w + 8);
  const float bx = w0.x * gs.x + w0.y * gs.y + w0.z * gs.z + w0.w;
  const float by = w1.x * gs.x + w1.y * gs.y + w1.z * gs.z + w1.w;
  const float bz = w2.x * gs.x + w2.y * gs.y + w2.z * gs.z + w2.w;
  const SamplePos q = contract_gaussian(bx, by, bz, gs.std, pa.scale);
  const uint32_t amask = (1u << pa.grid.log2T) - 1u;
  float acc = 0.f;
  for (int l0 = 0; l0 < pa.grid.L; l0 += 4) {  // up to four levels' 32 gathers in flight, then their blends in level order
    float fv[4][8][1];
    float off[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (l0 + u < pa.grid.L) {  // wave-uniform
        const Corners cs = hash_corners(q.x, q.y, q.z, pa.grid.scal[l0 + u], amask);
        off[u][0] = cs.ox, off[u][1] = cs.oy, off[u][2] = cs.oz;
#pragma unroll
        for (int k = 0; k < 8; ++k) Entry<1, false>::load(tb, ((uint32_t)(l0 + u) << pa.grid.log2T) + cs.idx[k], fv[u][k]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (l0 + u < pa.grid.L) {
        Corners cs;
        cs.ox = off[u][0], cs.oy = off[u][1], cs.oz = off[u][2];
        float v[1];
        lerp_corners<1>(cs, fv[u], v);
        acc += (v[0] * rescale_weight(pa.grid.scal[l0 + u], q.std)) * p.dec[l0 + u];
      }
    }
  }
  return expf(acc);
}

// S2 for one sample.  LT > 0: the grid has exactly LT levels and ALL 8*LT corner loads are issued before the first
// blend -- one memory round trip per sample instead of one per level (the kernel is latency bound: 4-byte entries, a few
// waves per SIMD).  LT == 0: any level count, level after level.  Same arithmetic and summation order either way.
template <bool HALF, int LT>
__device__ __forceinline__ float prop_density(const PropDev& p, float ox, float oy, float oz, float dx, float dy,
                                              float dz, float area, float t0, float t1) {
  const SamplePos q = sample_position(ox, oy, oz, dx, dy, dz, area, t0, t1, p.scale);
  const uint32_t mask = (1u << p.grid.log2T) - 1u;
  float acc = 0.f;
  if constexpr (LT > 0) {
    float fv[LT][8][1];
    float off[LT][3];
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      const Corners cs = hash_corners(q.x, q.y, q.z, p.grid.scal[l], mask);
      off[l][0] = cs.ox, off[l][1] = cs.oy, off[l][2] = cs.oz;
#pragma unroll
      for (int k = 0; k < 8; ++k) Entry<1, HALF>::load(p.table, ((uint32_t)l << p.grid.log2T) + cs.idx[k], fv[l][k]);
    }
#pragma unroll
    for (int l = 0; l < LT; ++l) {
      Corners cs;
      cs.ox = off[l][0], cs.oy = off[l][1], cs.oz = off[l][2];
      float v[1];
      lerp_corners<1>(cs, fv[l], v);
      acc += (v[0] * rescale_weight(p.grid.scal[l], q.std)) * p.dec[l];
    }
  } else {
    for (int l = 0; l < p.grid.L; ++l) {
      float v[1];
      hash_level<1, HALF>(p.table, (uint32_t)l << p.grid.log2T, q.x, q.y, q.z, p.grid.scal[l], mask, v);
      acc += (v[0] * rescale_weight(p.grid.scal[l], q.std)) * p.dec[l];
    }
  }
  return expf(acc);
}

// HALF: fp16 tables; LT: level count of the proposal grids when both have the same, compile-time one (else 0).
// slab_len = (largest sample count of any round) + 1: the per-wave LDS slabs are sized to what the launch needs, not to
// kSMax, so that LDS does not cap the waves per CU (5 * 513 floats per wave allowed 12 of them).
template <bool HALF, int LT, bool ACT = false>
__global__ __launch_bounds__(256) void proposal_sampler_kernel(SamplerDev sd, const float* __restrict__ o,
                                                                const float* __restrict__ d,
                                                                const float* __restrict__ area,
                                                                const float* __restrict__ nears,
                                                                const float* __restrict__ fars, int64_t R, int slab_len,
                                                                const int32_t* __restrict__ cand_count,
                                                                const int32_t* __restrict__ cand_actor,
                                                                const float* __restrict__ cand_w2b,
                                                                const float* __restrict__ bounds, int inline_actors) {
  // per wave: spacing bins (2 buffers), euclid bins, weights, cdf
  extern __shared__ __attribute__((aligned(16))) float slab[];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  float* spA = slab + (size_t)wid * 5 * slab_len;
  float* spB = spA + slab_len;
  float* eu = spB + slab_len;
  float* wl = eu + slab_len;
  float* cdf = wl + slab_len;
  // each XCD (workgroup b runs on XCD b % 8) walks one contiguous eighth of the batch: neighbouring rays (a camera
  // patch) share ONE L2 for the fine levels instead of leaving copies of their lines in all eight
  const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = (int)blockIdx.x % nx, lb = (int)blockIdx.x / nx;
  const int nblk = ((int)gridDim.x - xcd + nx - 1) / nx;
  const int64_t r_end = R * (xcd + 1) / nx;
  for (int64_t ray = R * xcd / nx + (int64_t)lb * 4 + wid; ray < r_end; ray += (int64_t)nblk * 4) {
    const float ox = o[3 * ray], oy = o[3 * ray + 1], oz = o[3 * ray + 2];
    const float dx = d[3 * ray], dy = d[3 * ray + 1], dz = d[3 * ray + 2];
    const float ar = area[ray];
    const Spacing spc =
        make_spacing(nears ? nears[ray] : 0.f, fminf(fars ? fars[ray] : sd.sky, sd.sky), sd.lam, sd.scaling);
    wave_fence();
    // round 0 bins (eval mode: no jitter inside the fused kernel)
    int S = sd.ns[0];
    for (int k = lane; k <= S; k += 64) {
      const float b = power_bin(k, S, nullptr);
      const float e = spc.to_euclid(b);
      spA[k] = b;
      eu[k] = e;
      sd.sp_out[0][ray * (S + 1) + k] = b;
      sd.eu_out[0][ray * (S + 1) + k] = e;
    }
    wave_fence();
    float* cur = spA;
    float* nxt = spB;
    for (int rd = 0; rd < sd.n_rounds; ++rd) {
      // S2 + S3: density -> delta*density -> exclusive-sum transmittance -> weights
      float carry = 0.f;
      bool staged = false;  // ACT: the round's densities already lie in wl[] (static pass + dense in-box pass)
      if constexpr (ACT) {
        const int ncand = __builtin_amdgcn_readfirstlane(cand_count[ray]);
        if (ncand > 0 && !inline_actors) {
          staged = true;
          uint32_t* list = reinterpret_cast<uint32_t*>(cdf);  // (cdf is free until the resampling below)
          int nin = 0;                                         // wave-uniform: in-box samples of this round so far
          for (int k0 = 0; k0 < S; k0 += 64) {
            const int k = k0 + lane;
            const bool live = k < S;
            const float t0 = eu[live ? k : 0], t1 = eu[live ? k + 1 : 1];
            // (the candidate walk first: it ends in one integer, the 48 gathers of the static lookup then have the registers)
            const int slot = actor_slot_of(sd.K, ray, ncand, ox, oy, oz, dx, dy, dz, ar, t0, t1, cand_actor, cand_w2b, bounds);
            const bool inb = live && slot >= 0;
            const unsigned long long m = __ballot(inb);
            if (inb) list[nin + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)k | ((uint32_t)slot << 16);
            nin += __popcll(m);
            const float dens = prop_density<HALF, LT>(sd.prop[rd], ox, oy, oz, dx, dy, dz, ar, t0, t1);
            if (live) wl[k] = dens;
          }
          wave_fence();
          const uint32_t row = (uint32_t)ray * (uint32_t)sd.K;
          for (int i0 = 0; i0 < nin; i0 += 64) {
            const bool on = i0 + lane < nin;
            const uint32_t e = list[on ? i0 + lane : 0];
            const int k = (int)(e & 0xffffu);
            const uint32_t c = e >> 16;
            const float dens = actor_density_of(sd.prop[rd], sd.pact[rd], cand_w2b + (size_t)(row + c) * 12u,
                                                sd.pact[rd].tables[cand_actor[row + c]], ox, oy, oz, dx, dy, dz, ar, eu[k],
                                                eu[k + 1]);
            if (on) wl[k] = dens;
          }
          wave_fence();
        }
      }
      for (int k0 = 0; k0 < S; k0 += 64) {
        const int k = k0 + lane;
        const bool live = k < S;
        const float t0 = eu[live ? k : 0], t1 = eu[live ? k + 1 : 1];
        float dens;
        if (staged) {
          dens = wl[live ? k : 0];
        } else {
          dens = prop_density<HALF, LT>(sd.prop[rd], ox, oy, oz, dx, dy, dz, ar, t0, t1);
          if constexpr (ACT) {
            const int ncand = __builtin_amdgcn_readfirstlane(cand_count[ray]);
            if (ncand > 0)
              dens = actor_density_override(sd.prop[rd], sd.pact[rd], sd.K, ray, ncand, dens, live, ox, oy, oz, dx, dy, dz, ar,
                                            t0, t1, cand_actor, cand_w2b, bounds);
          }
        }
        const float dd = live ? (t1 - t0) * dens : 0.f;
        const float incl = wscan_add(dd, lane);
        float w = (1.f - expf(-dd)) * expf(-(carry + incl - dd));
        if (w != w) w = 0.f;  // nan_to_num
        w = fminf(fmaxf(w, -3.4028234663852886e38f), 3.4028234663852886e38f);
        if (live) {
          wl[k] = w;
          sd.w_out[rd][ray * S + k] = w;
        }
        carry += wscan::last(incl);
      }
      wave_fence();
      // S4: resample into the next round's bins
      const int Sn = sd.ns[rd + 1];
      pdf_resample_ray(wl, cur, cdf, S, Sn, sd.pad, nullptr, 0, spc, nxt, eu, lane);
      wave_fence();
      for (int k = lane; k <= Sn; k += 64) {
        sd.sp_out[rd + 1][ray * (Sn + 1) + k] = nxt[k];
        sd.eu_out[rd + 1][ray * (Sn + 1) + k] = eu[k];
      }
      float* tsw = cur;
      cur = nxt;
      nxt = tsw;
      S = Sn;
    }
  }
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_power_sampler(const float* nears, const float* fars, int64_t r, int32_t s, float lam,
                                   float scaling, const float* t_rand, float last_edge, float* spacing_bins,
                                   float* euclid_bins, void* stream) {
  NR_REQUIRE(fars && spacing_bins && euclid_bins && r >= 0 && s >= 1, NRHIP_ERR_INVALID_ARG,
             "power_sampler: bad argument");
  if (r == 0) return NRHIP_OK;
  NR_REQUIRE((int64_t)kRaysPerSamplerBlock * (s + 1) < (int64_t)1 << 31, NRHIP_ERR_UNSUPPORTED,
             "power_sampler: %d samples per ray", s);
  power_sampler_kernel<<<grid_for(r, kRaysPerSamplerBlock), 256, 0, (hipStream_t)stream>>>(
      nears, fars, r, s, lam, scaling, t_rand, last_edge, spacing_bins, euclid_bins);
  return check_launch("power_sampler");
}

extern "C" int nrhip_power_sampler_ordered(const float* nears, const float* fars, int64_t r, int32_t s, float lam,
                                           float scaling, const float* t_rand, float last_edge, float* spacing_bins,
                                           float* euclid_bins, const float* origins, const float* directions, float t_ref,
                                           float static_scale, int32_t key_bits, int32_t* order, void* stream) {
  NR_REQUIRE(r >= 0 && r < (INT64_C(1) << 31) && s >= 1, NRHIP_ERR_INVALID_ARG, "power_sampler_ordered: bad argument");
  if (r == 0) return NRHIP_OK;
  NR_REQUIRE(fars && spacing_bins && euclid_bins && origins && directions && order, NRHIP_ERR_INVALID_ARG,
             "power_sampler_ordered: NULL pointer");
  NR_REQUIRE(static_scale > 0.f && t_ref >= 0.f, NRHIP_ERR_INVALID_ARG,
             "power_sampler_ordered: scale must be > 0 and t_ref >= 0");
  NR_REQUIRE(key_bits == 0 || key_bits == 4 || key_bits == 5, NRHIP_ERR_INVALID_ARG,
             "power_sampler_ordered: key_bits %d not in {0 (default = 4), 4, 5}", key_bits);
  const int blocks = 1 + grid_for(r * (s + 1), kOrderThreads);
  if (key_bits == 5)
    power_sampler_order_kernel<5><<<blocks, kOrderThreads, 0, (hipStream_t)stream>>>(
        nears, fars, r, s, lam, scaling, t_rand, last_edge, spacing_bins, euclid_bins, origins, directions, t_ref,
        static_scale, order);
  else
    power_sampler_order_kernel<4><<<blocks, kOrderThreads, 0, (hipStream_t)stream>>>(
        nears, fars, r, s, lam, scaling, t_rand, last_edge, spacing_bins, euclid_bins, origins, directions, t_ref,
        static_scale, order);
  return check_launch("power_sampler_ordered");
}

extern "C" int nrhip_pdf_sample(const float* weights, const float* spacing_bins, const float* nears, const float* fars,
                                int64_t r, int32_t s_prev, int32_t s_new, float lam, float scaling,
                                float histogram_padding, const float* rand, int32_t rand_stride,
                                float* new_spacing_bins, float* new_euclid_bins, void* stream) {
  NR_REQUIRE(weights && spacing_bins && fars && new_spacing_bins && new_euclid_bins && r >= 0, NRHIP_ERR_INVALID_ARG,
             "pdf_sample: null pointer");
  NR_REQUIRE(s_prev >= 1 && s_prev <= kSMax && s_new >= 1 && s_new <= kSMax, NRHIP_ERR_UNSUPPORTED,
             "pdf_sample: sample counts (%d -> %d) outside [1,%d]", s_prev, s_new, kSMax);
  NR_REQUIRE(rand_stride == 0 || rand_stride == s_new + 1, NRHIP_ERR_INVALID_ARG,
             "pdf_sample: rand_stride must be 0 (single jitter) or s_new+1");
  if (r == 0) return NRHIP_OK;
  pdf_sample_kernel<<<(int)((r + 3) / 4), 256, (size_t)4 * 3 * (s_prev + 1) * sizeof(float), (hipStream_t)stream>>>(
      weights, spacing_bins, nears, fars, r, s_prev, s_new, lam, scaling, histogram_padding, rand, rand_stride,
      new_spacing_bins, new_euclid_bins);
  return check_launch("pdf_sample");
}

static int sampler_fwd(const nrhip_sampler_cfg* cfg, const nrhip_proposal* props, const nrhip_actors* actors,
                       const int32_t* cand_count, const int32_t* cand_actor, const float* cand_w2b, const float* origins,
                       const float* directions, const float* pixel_area, const float* nears, const float* fars, int64_t r,
                       float* const* round_weights, float* const* round_spacing, float* const* round_euclid, void* stream) {
  NR_REQUIRE(cfg && props && origins && directions && pixel_area && round_weights && round_spacing && round_euclid,
             NRHIP_ERR_INVALID_ARG, "proposal_sampler_fwd: null pointer");
  NR_REQUIRE(cfg->n_rounds >= 1 && cfg->n_rounds <= 2, NRHIP_ERR_UNSUPPORTED,
             "proposal_sampler_fwd: n_rounds %d outside [1,2]", cfg->n_rounds);
  SamplerDev sd;
  sd.n_rounds = cfg->n_rounds;
  sd.lam = cfg->lam, sd.scaling = cfg->scaling, sd.pad = cfg->histogram_padding, sd.sky = cfg->sky_distance;
  for (int i = 0; i <= cfg->n_rounds; ++i) {
    NR_REQUIRE(cfg->n_samples[i] >= 1 && cfg->n_samples[i] <= kSMax, NRHIP_ERR_UNSUPPORTED,
               "proposal_sampler_fwd: n_samples[%d]=%d outside [1,%d]", i, cfg->n_samples[i], kSMax);
    sd.ns[i] = cfg->n_samples[i];
    NR_REQUIRE(round_spacing[i] && round_euclid[i], NRHIP_ERR_INVALID_ARG, "proposal_sampler_fwd: null bins output");
    sd.sp_out[i] = round_spacing[i];
    sd.eu_out[i] = round_euclid[i];
  }
  for (int i = 0; i < cfg->n_rounds; ++i) {
    if (int e = validate_grid(&props[i].grid)) return e;
    NR_REQUIRE(props[i].grid.n_features == 1, NRHIP_ERR_UNSUPPORTED, "proposal field needs features_per_level == 1");
    NR_REQUIRE(props[i].table && props[i].decoder_weight && props[i].static_scale > 0.f && round_weights[i],
               NRHIP_ERR_INVALID_ARG, "proposal_sampler_fwd: bad proposal field %d", i);
    sd.prop[i] = PropDev{to_dev(props[i].grid), props[i].table, props[i].static_scale, props[i].decoder_weight};
    sd.w_out[i] = round_weights[i];
  }
  const float* bounds = nullptr;
  sd.K = 0;
  if (actors) {
    NR_REQUIRE(cand_count && cand_actor && cand_w2b, NRHIP_ERR_INVALID_ARG, "proposal_sampler_fwd_actors: NULL candidate lists");
    for (int i = 0; i < cfg->n_rounds; ++i) {
      const nrhip_actors& a = actors[i];
      if (int e = validate_grid(&a.grid)) return e;
      NR_REQUIRE(a.tables && a.bounds && a.actor_scale > 0.f && a.n_actors >= 1, NRHIP_ERR_INVALID_ARG,
                 "proposal_sampler_fwd_actors: bad actor descriptor %d", i);
      // (the STATIC proposal table may be fp16 storage -- round 5: on incoherent rays the kernel runs against the L2 <-> fabric
      //  bandwidth, 15.7 GB per 65 536-ray launch with fp32 tables, and half the footprint is half the misses; the small actor
      //  grids stay fp32)
      NR_REQUIRE(a.grid.n_features == 1 && a.grid.param_dtype == 0 && a.grid.num_levels <= props[i].grid.num_levels,
                 NRHIP_ERR_UNSUPPORTED,
                 "proposal_sampler_fwd_actors: actor grids need 1 feature per level, fp32 tables and at most the static "
                 "grid's levels; use the unfused ops");
      const int k = a.max_candidates > 0 ? a.max_candidates : NRHIP_DEFAULT_ACTOR_CANDIDATES;
      NR_REQUIRE(i == 0 || (k == sd.K && a.bounds == bounds), NRHIP_ERR_INVALID_ARG,
                 "proposal_sampler_fwd_actors: the rounds must share one actor set (candidate lists, bounds)");
      sd.K = k, bounds = a.bounds;
      sd.pact[i] = PropActorDev{a.tables, to_dev(a.grid), a.actor_scale};
    }
  }
  if (r == 0) return NRHIP_OK;
  int smax = 0, lt = sd.prop[0].grid.L;
  bool half = sd.prop[0].grid.dtype == 1;
  for (int i = 0; i <= cfg->n_rounds; ++i) smax = cfg->n_samples[i] > smax ? cfg->n_samples[i] : smax;
  for (int i = 1; i < cfg->n_rounds; ++i) {
    if (sd.prop[i].grid.L != lt) lt = 0;
    NR_REQUIRE((sd.prop[i].grid.dtype == 1) == half, NRHIP_ERR_UNSUPPORTED,
               "proposal_sampler_fwd: the proposal tables must share one storage type");
  }
  const int slab_len = smax + 1;
  const size_t lds = (size_t)4 * 5 * slab_len * sizeof(float);
  int64_t blocks = (r + 3) / 4;
  if (blocks > 256 * 8) blocks = 256 * 8;
  const hipStream_t st = (hipStream_t)stream;
#define LAUNCH(HALF_, LT_)                                                                                              \
  proposal_sampler_kernel<HALF_, LT_><<<(int)blocks, 256, lds, st>>>(sd, origins, directions, pixel_area, nears, fars, r, \
                                                                     slab_len, nullptr, nullptr, nullptr, nullptr, 0)
  if (actors && half) {
    if (lt == 6)
      proposal_sampler_kernel<true, 6, true><<<(int)blocks, 256, lds, st>>>(sd, origins, directions, pixel_area, nears, fars,
                                                                           r, slab_len, cand_count, cand_actor, cand_w2b,
                                                                           bounds, 0);
    else
      proposal_sampler_kernel<true, 0, true><<<(int)blocks, 256, lds, st>>>(sd, origins, directions, pixel_area, nears, fars,
                                                                           r, slab_len, cand_count, cand_actor, cand_w2b,
                                                                           bounds, 0);
  } else if (actors) {
    const int inline_actors = tuning().sampler_actor_inline ? 1 : 0;  // NRHIP_SAMPLER_ACTOR_INLINE=1: the round-2..4 per-chunk lookup (A/B)
    if (lt == 6)
      proposal_sampler_kernel<false, 6, true><<<(int)blocks, 256, lds, st>>>(sd, origins, directions, pixel_area, nears, fars,
                                                                            r, slab_len, cand_count, cand_actor, cand_w2b,
                                                                            bounds, inline_actors);
    else
      proposal_sampler_kernel<false, 0, true><<<(int)blocks, 256, lds, st>>>(sd, origins, directions, pixel_area, nears, fars,
                                                                            r, slab_len, cand_count, cand_actor, cand_w2b,
                                                                            bounds, inline_actors);
  } else if (half) {
    if (lt == 6) LAUNCH(true, 6);
    else LAUNCH(true, 0);
  } else {
    if (lt == 6) LAUNCH(false, 6);  // NeuRAD's proposal grids (fields/neurad_field.py:170-177)
    else if (lt == 4) LAUNCH(false, 4);
    else if (lt == 8) LAUNCH(false, 8);
    else LAUNCH(false, 0);
  }
#undef LAUNCH
  return check_launch("proposal_sampler_fwd");
}

extern "C" int nrhip_proposal_sampler_fwd(const nrhip_sampler_cfg* cfg, const nrhip_proposal* props,
                                          const float* origins, const float* directions, const float* pixel_area,
                                          const float* nears, const float* fars, int64_t r,
                                          float* const* round_weights, float* const* round_spacing,
                                          float* const* round_euclid, void* stream) {
  return sampler_fwd(cfg, props, nullptr, nullptr, nullptr, nullptr, origins, directions, pixel_area, nears, fars, r,
                     round_weights, round_spacing, round_euclid, stream);
}

extern "C" int nrhip_proposal_sampler_fwd_actors(const nrhip_sampler_cfg* cfg, const nrhip_proposal* props,
                                                 const nrhip_actors* actors, const int32_t* cand_count,
                                                 const int32_t* cand_actor, const float* cand_w2b, const float* origins,
                                                 const float* directions, const float* pixel_area, const float* nears,
                                                 const float* fars, int64_t r, float* const* round_weights,
                                                 float* const* round_spacing, float* const* round_euclid, void* stream) {
  NR_REQUIRE(actors, NRHIP_ERR_INVALID_ARG, "proposal_sampler_fwd_actors: actors is NULL");
  return sampler_fwd(cfg, props, actors, cand_count, cand_actor, cand_w2b, origins, directions, pixel_area, nears, fars, r,
                     round_weights, round_spacing, round_euclid, stream);
}
