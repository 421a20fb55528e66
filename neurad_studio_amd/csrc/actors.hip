// H5: dynamic actors.  Per ray: pose interpolation of every actor at the ray's time + line cull -> compact
// candidate list (one wavefront per ray, lane = actor, ballot/popcount compaction).  Per sample: in-box test over
// the ray's few candidates, then the actor's own 3-D hash grid replaces the static features (torch-path
// semantics: per-actor grids, highest actor index wins on overlap, neurad_encoding.py:184-185,256-263).
#include "common.h"

namespace nrhip {

struct ActorsDev {
  int A, Tn;
  const float* ts;
  const float* pos;
  const float* rot6;
  const uint8_t* present;
  const float* bounds;
  GridDev grid;
  const void* const* tables;
  float scale;
  const float* flip;  // [R] +-1 per ray (training-mode x flip, neurad_encoding.py:212-219) or NULL
  int K;              // row length of the per-ray candidate lists (nrhip_actors.max_candidates; == A: no ray overflows)
};

constexpr int KH = NRHIP_MAX_SAMPLE_CONTAINMENTS;  // containing boxes recorded per SAMPLE by nrhip_actor_hits

__device__ __forceinline__ void normalize3(float& x, float& y, float& z) {  // F.normalize, eps 1e-12
  const float n = fmaxf(sqrtf(x * x + y * y + z * z), 1e-12f);
  x /= n, y /= n, z /= n;
}

// poses.py:110-118: a1 = normalize(a1); a2 = normalize(a2 - (a1.a2) a1)
__device__ __forceinline__ void ortho6(const float* p, float (&o)[9], const float* t) {
  float a1x = p[0], a1y = p[1], a1z = p[2], a2x = p[3], a2y = p[4], a2z = p[5];
  normalize3(a1x, a1y, a1z);
  const float dt = a1x * a2x + a1y * a2y + a1z * a2z;
  a2x -= dt * a1x, a2y -= dt * a1y, a2z -= dt * a1z;
  normalize3(a2x, a2y, a2z);
  o[0] = a1x, o[1] = a1y, o[2] = a1z, o[3] = a2x, o[4] = a2y, o[5] = a2z, o[6] = t[0], o[7] = t[1], o[8] = t[2];
}

__global__ __launch_bounds__(256) void actor_prepare_kernel(ActorsDev a, RaysDev r, const float* __restrict__ times,
                                                             int32_t* __restrict__ cand_count,
                                                             int32_t* __restrict__ cand_actor,
                                                             float* __restrict__ cand_w2b,
                                                             int32_t* __restrict__ overflow) {
  const int lane = threadIdx.x & 63;
  const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (ray >= r.R) return;
  const float q = times[ray];
  // torch.searchsorted(pose_times, q) (side=left) = #{ts < q}
  int right = 0;
  for (int i = 0; i < a.Tn; ++i) right += a.ts[i] < q ? 1 : 0;
  const int left = max(right - 1, 0);
  right = min(right, a.Tn - 1);
  const float lt = a.ts[left], rt = a.ts[right];
  const float frac = fminf(fmaxf((q - lt) / (rt - lt + 1e-6f), 0.f), 1.f);
  // the ray's line through the means of its first and last sample (neurad_encoding.py:231-235)
  const float ox = r.o[3 * ray], oy = r.o[3 * ray + 1], oz = r.o[3 * ray + 2];
  const float dx = r.d[3 * ray], dy = r.d[3 * ray + 1], dz = r.d[3 * ray + 2];
  const float area = r.area[ray];
  const SamplePos p0 = sample_gaussian(ox, oy, oz, dx, dy, dz, area, r.starts[ray * r.stride], r.ends[ray * r.stride]);
  const SamplePos p1 = sample_gaussian(ox, oy, oz, dx, dy, dz, area, r.starts[ray * r.stride + r.S - 1],
                                       r.ends[ray * r.stride + r.S - 1]);
  float lx = p1.x - p0.x, ly = p1.y - p0.y, lz = p1.z - p0.z;
  const float ln = sqrtf(lx * lx + ly * ly + lz * lz) + 1e-7f;
  lx /= ln, ly /= ln, lz /= ln;
  int count = 0;
  for (int a0 = 0; a0 < a.A; a0 += 64) {
    const int act = a0 + lane;
    bool close = false;
    float w2b[12];
    if (act < a.A) {
      float pl[9], pr[9], ip[9];
      ortho6(a.rot6 + ((size_t)left * a.A + act) * 6, pl, a.pos + ((size_t)left * a.A + act) * 3);
      ortho6(a.rot6 + ((size_t)right * a.A + act) * 6, pr, a.pos + ((size_t)right * a.A + act) * 3);
#pragma unroll
      for (int k = 0; k < 9; ++k) ip[k] = pl[k] + (pr[k] - pl[k]) * frac;
      // rotation_6d_to_matrix (camera_utils.py:422-443): rows b1, b2, b3 = b1 x b2
      float b1x = ip[0], b1y = ip[1], b1z = ip[2], b2x = ip[3], b2y = ip[4], b2z = ip[5];
      normalize3(b1x, b1y, b1z);
      const float dt = b1x * b2x + b1y * b2y + b1z * b2z;
      b2x -= dt * b1x, b2y -= dt * b1y, b2z -= dt * b1z;
      normalize3(b2x, b2y, b2z);
      const float b3x = b1y * b2z - b1z * b2y, b3y = b1z * b2x - b1x * b2z, b3z = b1x * b2y - b1y * b2x;
      const float tx = ip[6], ty = ip[7], tz = ip[8];
      // boxes2world = [[b1],[b2],[b3] | t];  world2box = [R^T | -R^T t]  (utils/poses.py:42-55)
      w2b[0] = b1x, w2b[1] = b2x, w2b[2] = b3x, w2b[3] = -(b1x * tx + b2x * ty + b3x * tz);
      w2b[4] = b1y, w2b[5] = b2y, w2b[6] = b3y, w2b[7] = -(b1y * tx + b2y * ty + b3y * tz);
      w2b[8] = b1z, w2b[9] = b2z, w2b[10] = b3z, w2b[11] = -(b1z * tx + b2z * ty + b3z * tz);
      const bool valid = a.present[(size_t)left * a.A + act] | a.present[(size_t)right * a.A + act];
      const float vx = tx - p0.x, vy = ty - p0.y, vz = tz - p0.z;
      const float cx = vy * lz - vz * ly, cy = vz * lx - vx * lz, cz = vx * ly - vy * lx;
      const float dist = sqrtf(cx * cx + cy * cy + cz * cz);
      const float bx = a.bounds[3 * act], by = a.bounds[3 * act + 1], bz = a.bounds[3 * act + 2];
      const float radius = sqrtf(bx * bx + by * by + bz * bz);
      close = valid && dist < radius;
    }
    const unsigned long long m = __ballot(close);
    const int slot = count + __popcll(m & ((1ull << lane) - 1ull));
    if (close) {
      if (slot < a.K) {
        cand_actor[ray * a.K + slot] = act;
#pragma unroll
        for (int k = 0; k < 12; ++k) cand_w2b[(ray * a.K + slot) * 12 + k] = w2b[k];
      } else if (overflow) {
        *overflow = 1;
      }
    }
    count += __popcll(m);
  }
  if (lane == 0) cand_count[ray] = min(count, a.K);
}

// shared per-sample part: which actor (if any) contains the sample; box-frame position/direction
struct ActorHit {
  int actor;  // -1: none
  float px, py, pz, std, dx, dy, dz;
};

__device__ __forceinline__ ActorHit find_hit(const ActorsDev& a, const RaysDev& r, int64_t i,
                                             const int32_t* cand_count, const int32_t* cand_actor,
                                             const float* cand_w2b) {
  ActorHit h;
  h.actor = -1;
  const int64_t ray = i / r.S;
  const int s = (int)(i - ray * r.S);
  const int n = cand_count[ray];
  h.dx = r.d[3 * ray], h.dy = r.d[3 * ray + 1], h.dz = r.d[3 * ray + 2];
  if (n == 0) return h;
  const SamplePos g = sample_gaussian(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], h.dx, h.dy, h.dz, r.area[ray],
                                      r.starts[ray * r.stride + s], r.ends[ray * r.stride + s]);
  h.std = g.std;
  const float* wsel = nullptr;
  for (int c = 0; c < n; ++c) {  // ascending actor index; the LAST hit wins (neurad_encoding.py:184-185 on CPU)
    const float* w = cand_w2b + (ray * a.K + c) * 12;
    const int act = cand_actor[ray * a.K + c];
    const float bx = w[0] * g.x + w[1] * g.y + w[2] * g.z + w[3];
    const float by = w[4] * g.x + w[5] * g.y + w[6] * g.z + w[7];
    const float bz = w[8] * g.x + w[9] * g.y + w[10] * g.z + w[11];
    if (fabsf(bx) < a.bounds[3 * act] && fabsf(by) < a.bounds[3 * act + 1] && fabsf(bz) < a.bounds[3 * act + 2]) {
      h.actor = act, h.px = bx, h.py = by, h.pz = bz, wsel = w;
    }
  }
  if (h.actor >= 0) {
    float ddx = wsel[0] * h.dx + wsel[1] * h.dy + wsel[2] * h.dz;
    float ddy = wsel[4] * h.dx + wsel[5] * h.dy + wsel[6] * h.dz;
    float ddz = wsel[8] * h.dx + wsel[9] * h.dy + wsel[10] * h.dz;
    const float nn = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) + 1e-7f;  // neurad_encoding.py:207
    h.dx = ddx / nn, h.dy = ddy / nn, h.dz = ddz / nn;
    if (a.flip) {
      const float f = a.flip[ray];
      h.px *= f, h.dx *= f;
    }
  }
  return h;
}

template <int F>
__global__ __launch_bounds__(256) void actor_encode_kernel(ActorsDev a, RaysDev r,
                                                            const int32_t* __restrict__ cand_count,
                                                            const int32_t* __restrict__ cand_actor,
                                                            const float* __restrict__ cand_w2b, int out_dim,
                                                            float* __restrict__ feat, float* __restrict__ dirs,
                                                            int32_t* __restrict__ hit) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= r.R * r.S) return;
  const ActorHit h = find_hit(a, r, i, cand_count, cand_actor, cand_w2b);
  if (dirs) dirs[3 * i] = h.dx, dirs[3 * i + 1] = h.dy, dirs[3 * i + 2] = h.dz;
  if (hit) hit[i] = h.actor;
  if (h.actor < 0) return;
  const SamplePos p = contract_gaussian(h.px, h.py, h.pz, h.std, a.scale);
  const void* table = a.tables[h.actor];
  const uint32_t mask = (1u << a.grid.log2T) - 1u;
  float* o = feat + i * out_dim;
  for (int l = 0; l < a.grid.L; ++l) {
    float v[F];
    hash_level<F, false>(table, (uint32_t)l << a.grid.log2T, p.x, p.y, p.z, a.grid.scal[l], mask, v);
    const float w = rescale_weight(a.grid.scal[l], p.std);
#pragma unroll
    for (int f = 0; f < F; ++f) o[l * F + f] = v[f] * w;
  }
  for (int k = a.grid.L * F; k < out_dim; ++k) o[k] = 0.f;  // F.pad (neurad_encoding.py:183)
}

__global__ __launch_bounds__(256) void actor_density_kernel(ActorsDev a, RaysDev r,
                                                             const int32_t* __restrict__ cand_count,
                                                             const int32_t* __restrict__ cand_actor,
                                                             const float* __restrict__ cand_w2b,
                                                             const float* __restrict__ dec, int n_dec,
                                                             float* __restrict__ dens, int32_t* __restrict__ hit) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= r.R * r.S) return;
  const ActorHit h = find_hit(a, r, i, cand_count, cand_actor, cand_w2b);
  if (hit) hit[i] = h.actor;
  if (h.actor < 0) return;
  const SamplePos p = contract_gaussian(h.px, h.py, h.pz, h.std, a.scale);
  const void* table = a.tables[h.actor];
  const uint32_t mask = (1u << a.grid.log2T) - 1u;
  float acc = 0.f;
  for (int l = 0; l < a.grid.L && l < n_dec; ++l) {
    float v[1];
    hash_level<1, false>(table, (uint32_t)l << a.grid.log2T, p.x, p.y, p.z, a.grid.scal[l], mask, v);
    acc += (v[0] * rescale_weight(a.grid.scal[l], p.std)) * dec[l];
  }
  dens[i] = expf(acc);
}

// every (sample, candidate) containment, not only the winner: the reference's index_put backward hands the
// upstream gradient to ALL duplicate (ray, sample) rows (neurad_encoding.py:184-185,256-263), so overlapping
// actors all receive gradients -- the training path needs the full list to reproduce that.
__global__ __launch_bounds__(256) void actor_hits_kernel(ActorsDev a, RaysDev r, const int32_t* __restrict__ cand_count,
                                                          const int32_t* __restrict__ cand_actor,
                                                          const float* __restrict__ cand_w2b,
                                                          int32_t* __restrict__ hits) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= r.R * r.S) return;
  const int64_t ray = i / r.S;
  const int s = (int)(i - ray * r.S);
  const int n = cand_count[ray];
#pragma unroll
  for (int c = 0; c < KH; ++c) hits[i * KH + c] = -1;
  if (n == 0) return;
  const SamplePos g = sample_gaussian(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray], r.d[3 * ray + 1],
                                      r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + s],
                                      r.ends[ray * r.stride + s]);
  int slot = 0;  // compacted, ascending actor order; beyond KH overlapping boxes the lowest indices drop out, the
                 // winner (highest index = last entry) stays
  for (int c = 0; c < n; ++c) {
    const float* w = cand_w2b + (ray * a.K + c) * 12;
    const int act = cand_actor[ray * a.K + c];
    const float bx = w[0] * g.x + w[1] * g.y + w[2] * g.z + w[3];
    const float by = w[4] * g.x + w[5] * g.y + w[6] * g.z + w[7];
    const float bz = w[8] * g.x + w[9] * g.y + w[10] * g.z + w[11];
    if (fabsf(bx) < a.bounds[3 * act] && fabsf(by) < a.bounds[3 * act + 1] && fabsf(bz) < a.bounds[3 * act + 2]) {
      if (slot == KH) {
        for (int k = 1; k < KH; ++k) hits[i * KH + k - 1] = hits[i * KH + k];
        slot = KH - 1;
      }
      hits[i * KH + slot++] = act;
    }
  }
}

static int to_dev(const nrhip_actors* a, ActorsDev& d) {
  NR_REQUIRE(a, NRHIP_ERR_INVALID_ARG, "actors descriptor is NULL");
  NR_REQUIRE(a->n_actors >= 1 && a->n_times >= 1, NRHIP_ERR_INVALID_ARG, "actors: need >= 1 actor and timestamp");
  NR_REQUIRE(a->timestamps && a->positions && a->rotations_6d && a->present && a->bounds, NRHIP_ERR_INVALID_ARG,
             "actors descriptor has a NULL pointer");
  d.A = a->n_actors, d.Tn = a->n_times;
  d.K = a->max_candidates > 0 ? a->max_candidates : NRHIP_DEFAULT_ACTOR_CANDIDATES;
  d.ts = a->timestamps, d.pos = a->positions, d.rot6 = a->rotations_6d, d.present = a->present, d.bounds = a->bounds;
  d.grid = to_dev(a->grid);
  d.tables = a->tables;
  d.scale = a->actor_scale;
  d.flip = nullptr;
  return NRHIP_OK;
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_actor_prepare(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                   int32_t* cand_count, int32_t* cand_actor, float* cand_w2b, int32_t* overflow,
                                   void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(rays->n_samples >= 1 && times && cand_count && cand_actor && cand_w2b, NRHIP_ERR_INVALID_ARG,
             "actor_prepare: bad argument");
  NR_REQUIRE(overflow || d.K >= d.A, NRHIP_ERR_INVALID_ARG,
             "actor_prepare: without an overflow flag the candidate lists must hold all %d actors (max_candidates = %d)",
             d.A, d.K);
  actor_prepare_kernel<<<(int)((rays->n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(d, to_dev(*rays), times,
                                                                                     cand_count, cand_actor, cand_w2b,
                                                                                     overflow);
  return check_launch("actor_prepare");
}

extern "C" int nrhip_actor_encode(const nrhip_actors* a, const nrhip_rays* rays, const int32_t* cand_count,
                                  const int32_t* cand_actor, const float* cand_w2b, int32_t out_dim, float* features,
                                  float* directions, int32_t* hit, const float* ray_flip, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  d.flip = ray_flip;
  if (int e = validate_grid(&a->grid)) return e;
  if (int e = validate_rays(rays)) return e;
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(a->tables && a->actor_scale > 0.f && cand_count && cand_actor && cand_w2b && features,
             NRHIP_ERR_INVALID_ARG, "actor_encode: bad argument");
  NR_REQUIRE(a->grid.param_dtype == 0, NRHIP_ERR_UNSUPPORTED, "actor grids: fp32 tables only");
  NR_REQUIRE(out_dim >= a->grid.num_levels * a->grid.n_features, NRHIP_ERR_INVALID_ARG,
             "actor_encode: out_dim %d < actor feature dim %d", out_dim, a->grid.num_levels * a->grid.n_features);
  const RaysDev rd = to_dev(*rays);
  const int blocks = grid_for(n, 256);
  const hipStream_t st = (hipStream_t)stream;
  switch (a->grid.n_features) {
    case 1: actor_encode_kernel<1><<<blocks, 256, 0, st>>>(d, rd, cand_count, cand_actor, cand_w2b, out_dim, features, directions, hit); break;
    case 2: actor_encode_kernel<2><<<blocks, 256, 0, st>>>(d, rd, cand_count, cand_actor, cand_w2b, out_dim, features, directions, hit); break;
    case 4: actor_encode_kernel<4><<<blocks, 256, 0, st>>>(d, rd, cand_count, cand_actor, cand_w2b, out_dim, features, directions, hit); break;
    default: actor_encode_kernel<8><<<blocks, 256, 0, st>>>(d, rd, cand_count, cand_actor, cand_w2b, out_dim, features, directions, hit); break;
  }
  return check_launch("actor_encode");
}

extern "C" int nrhip_actor_hits(const nrhip_actors* a, const nrhip_rays* rays, const int32_t* cand_count,
                                const int32_t* cand_actor, const float* cand_w2b, int32_t* hits, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  if (int e = validate_rays(rays)) return e;
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(cand_count && cand_actor && cand_w2b && hits, NRHIP_ERR_INVALID_ARG, "actor_hits: null pointer");
  actor_hits_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(d, to_dev(*rays), cand_count, cand_actor,
                                                                      cand_w2b, hits);
  return check_launch("actor_hits");
}

extern "C" int nrhip_actor_density(const nrhip_actors* a, const nrhip_rays* rays, const int32_t* cand_count,
                                   const int32_t* cand_actor, const float* cand_w2b, const float* decoder_weight,
                                   int32_t n_dec, float* density, int32_t* hit, const float* ray_flip, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  d.flip = ray_flip;
  if (int e = validate_grid(&a->grid)) return e;
  if (int e = validate_rays(rays)) return e;
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(a->tables && a->actor_scale > 0.f && cand_count && cand_actor && cand_w2b && decoder_weight && density,
             NRHIP_ERR_INVALID_ARG, "actor_density: bad argument");
  NR_REQUIRE(a->grid.n_features == 1 && a->grid.param_dtype == 0, NRHIP_ERR_UNSUPPORTED,
             "actor_density: proposal actor grids have features_per_level == 1, fp32");
  actor_density_kernel<<<grid_for(n, 256), 256, 0, (hipStream_t)stream>>>(d, to_dev(*rays), cand_count, cand_actor,
                                                                         cand_w2b, decoder_weight, n_dec, density, hit);
  return check_launch("actor_density");
}
