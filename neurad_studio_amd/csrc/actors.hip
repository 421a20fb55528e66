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

// Eval-time actor edit (DynamicActors.edit_boxes2world, model_components/dynamic_actors.py:181-249, the flatten=False branch the
// hash encoding reads): the viewer's sliders and the actor-shift FID evaluation (pipelines/ad_pipeline.py:476-480) move /
// yaw the boxes AFTER the pose interpolation.  on = 0: no edit.
struct ActorEdit {
  int on, index;           // index < 0: every actor, else that actor
  int shift, turn;         // translate by (lateral, longitudinal, height) in the box frame; pre-multiply a yaw
  float lat, lon, hgt, cs, sn;
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
                                                             int32_t* __restrict__ overflow, ActorEdit ed) {
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
      float tx = ip[6], ty = ip[7], tz = ip[8];
      if (ed.on && (ed.index < 0 || ed.index == act)) {
        if (ed.shift) {  // t' = boxes2world @ (lateral, longitudinal, height, 1)   (dynamic_actors.py:205-227)
          tx += b1x * ed.lat + b1y * ed.lon + b1z * ed.hgt;
          ty += b2x * ed.lat + b2y * ed.lon + b2z * ed.hgt;
          tz += b3x * ed.lat + b3y * ed.lon + b3z * ed.hgt;
        }
        if (ed.turn) {  // R' = yaw @ R: rows 0 and 1 mix, the translation stays   (dynamic_actors.py:229-249)
          const float n1x = ed.cs * b1x - ed.sn * b2x, n1y = ed.cs * b1y - ed.sn * b2y, n1z = ed.cs * b1z - ed.sn * b2z;
          const float n2x = ed.sn * b1x + ed.cs * b2x, n2y = ed.sn * b1y + ed.cs * b2y, n2z = ed.sn * b1z + ed.cs * b2z;
          b1x = n1x, b1y = n1y, b1z = n1z, b2x = n2x, b2y = n2y, b2z = n2z;
        }
      }
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

template <int F, bool HALF>
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
    hash_level<F, HALF>(table, (uint32_t)l << a.grid.log2T, p.x, p.y, p.z, a.grid.scal[l], mask, v);
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

extern "C" int nrhip_actor_prepare_edited(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                          const nrhip_actor_edit* edit, int32_t* cand_count, int32_t* cand_actor,
                                          float* cand_w2b, int32_t* overflow, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(rays->n_samples >= 1 && times && cand_count && cand_actor && cand_w2b, NRHIP_ERR_INVALID_ARG,
             "actor_prepare: bad argument");
  NR_REQUIRE(overflow || d.K >= d.A, NRHIP_ERR_INVALID_ARG,
             "actor_prepare: without an overflow flag the candidate lists must hold all %d actors (max_candidates = %d)",
             d.A, d.K);
  ActorEdit ed = {};
  // the reference returns the poses untouched unless longitudinal, lateral or rotation is set -- a height-only edit does
  // nothing (dynamic_actors.py:182-187); the index is clamped to the last actor (:192)
  if (edit && (edit->longitudinal != 0.f || edit->lateral != 0.f || edit->rotation != 0.f)) {
    ed.on = 1;
    ed.index = edit->index < 0 ? -1 : (edit->index < d.A - 1 ? edit->index : d.A - 1);
    ed.shift = edit->longitudinal != 0.f || edit->lateral != 0.f || edit->height != 0.f;
    ed.turn = edit->rotation != 0.f;
    ed.lat = edit->lateral, ed.lon = edit->longitudinal, ed.hgt = edit->height;
    ed.cs = cosf(edit->rotation), ed.sn = sinf(edit->rotation);
  }
  actor_prepare_kernel<<<(int)((rays->n_rays + 3) / 4), 256, 0, (hipStream_t)stream>>>(d, to_dev(*rays), times,
                                                                                     cand_count, cand_actor, cand_w2b,
                                                                                     overflow, ed);
  return check_launch("actor_prepare");
}

extern "C" int nrhip_actor_prepare(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                   int32_t* cand_count, int32_t* cand_actor, float* cand_w2b, int32_t* overflow,
                                   void* stream) {
  return nrhip_actor_prepare_edited(a, rays, times, nullptr, cand_count, cand_actor, cand_w2b, overflow, stream);
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
  NR_REQUIRE(out_dim >= a->grid.num_levels * a->grid.n_features, NRHIP_ERR_INVALID_ARG,
             "actor_encode: out_dim %d < actor feature dim %d", out_dim, a->grid.num_levels * a->grid.n_features);
  const RaysDev rd = to_dev(*rays);
  const int blocks = grid_for(n, 256);
  const hipStream_t st = (hipStream_t)stream;
#define CALL(F)                                                                                                              \
  do {                                                                                                                       \
    if (a->grid.param_dtype == 1)                                                                                            \
      actor_encode_kernel<F, true><<<blocks, 256, 0, st>>>(d, rd, cand_count, cand_actor, cand_w2b, out_dim, features, directions, hit);  \
    else                                                                                                                     \
      actor_encode_kernel<F, false><<<blocks, 256, 0, st>>>(d, rd, cand_count, cand_actor, cand_w2b, out_dim, features, directions, hit); \
  } while (0)
  switch (a->grid.n_features) {
    case 1: CALL(1); break;
    case 2: CALL(2); break;
    case 4: CALL(4); break;
    default: CALL(8); break;
  }
#undef CALL
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

// ---------------------------------------------------------------------------------------------------------------------
// Training path of the in-box samples (B1): box-frame, contracted position of (sample, actor) PAIRS and its backward into
// the trajectory parameters.  One thread per pair does what the reference spreads over ~50 torch ops (and their ~100
// autograd kernels): interpolate_trajectories_6d (utils/poses.py:90-150: Gram-Schmidt of the stored 6-D rotations, lerp
// between the two bracketing poses, Gram-Schmidt again) -> rotation_6d_to_matrix (cameras/camera_utils.py:422-443) ->
// pose inverse (utils/poses.py:42-55) -> transform_points_pairwise (cameras/lidars.py:550-564) -> training flip
// (neurad_encoding.py:212-219) -> ScaledSceneContraction(inf) of the gaussian (spatial_distortions.py:103-141).
// Same arithmetic as actor_prepare_kernel / find_hit, so the forward kernels and this path agree on every position.
namespace nrhip {

struct PairFrame {
  int left, right;
  float frac;
  float u1[3], u2[3], t[3];  // interpolated (normalised) pose: two rotation rows and the translation
  float b1[3], b2[3], b3[3];
  float n1, nc, dot;         // |u1|, |u2 - (b1.u2) b1|, b1.u2
};

__device__ __forceinline__ float dot3(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

// Gram-Schmidt with torch's F.normalize (eps 1e-12): a1 = r1/|r1|, a2 = c/|c|, c = r2 - (a1.r2) a1
__device__ __forceinline__ void gram_schmidt(const float* r1, const float* r2, float* a1, float* a2, float& n1, float& nc,
                                             float& dt) {
  n1 = fmaxf(sqrtf(dot3(r1, r1)), 1e-12f);
  for (int k = 0; k < 3; ++k) a1[k] = r1[k] / n1;
  dt = dot3(a1, r2);
  float c[3];
  for (int k = 0; k < 3; ++k) c[k] = r2[k] - dt * a1[k];
  nc = fmaxf(sqrtf(dot3(c, c)), 1e-12f);
  for (int k = 0; k < 3; ++k) a2[k] = c[k] / nc;
}

// gradient of gram_schmidt: (g_a1, g_a2) -> (g_r1, g_r2)
__device__ __forceinline__ void gram_schmidt_bwd(const float* r2, const float* a1, const float* a2, float n1, float nc,
                                                 float dt, const float* g_a1_in, const float* g_a2, float* g_r1,
                                                 float* g_r2) {
  float gc[3], ga1[3];
  const float s2 = dot3(g_a2, a2);
  for (int k = 0; k < 3; ++k) gc[k] = (g_a2[k] - s2 * a2[k]) / nc;
  const float s1 = dot3(gc, a1);
  for (int k = 0; k < 3; ++k) {
    g_r2[k] = gc[k] - s1 * a1[k];
    ga1[k] = g_a1_in[k] - s1 * r2[k] - dt * gc[k];
  }
  const float s0 = dot3(ga1, a1);
  for (int k = 0; k < 3; ++k) g_r1[k] = (ga1[k] - s0 * a1[k]) / n1;
}

__device__ __forceinline__ void pair_frame(const ActorsDev& a, int act, float q, PairFrame& f) {
  int right = 0;
  for (int i = 0; i < a.Tn; ++i) right += a.ts[i] < q ? 1 : 0;  // torch.searchsorted(side=left)
  f.left = max(right - 1, 0);
  f.right = min(right, a.Tn - 1);
  const float lt = a.ts[f.left], rt = a.ts[f.right];
  f.frac = fminf(fmaxf((q - lt) / (rt - lt + 1e-6f), 0.f), 1.f);
  float pl[9], pr[9];
  ortho6(a.rot6 + ((size_t)f.left * a.A + act) * 6, pl, a.pos + ((size_t)f.left * a.A + act) * 3);
  ortho6(a.rot6 + ((size_t)f.right * a.A + act) * 6, pr, a.pos + ((size_t)f.right * a.A + act) * 3);
  float ip[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) ip[k] = pl[k] + (pr[k] - pl[k]) * f.frac;
  for (int k = 0; k < 3; ++k) f.u1[k] = ip[k], f.u2[k] = ip[3 + k], f.t[k] = ip[6 + k];
  gram_schmidt(f.u1, f.u2, f.b1, f.b2, f.n1, f.nc, f.dot);
  f.b3[0] = f.b1[1] * f.b2[2] - f.b1[2] * f.b2[1];
  f.b3[1] = f.b1[2] * f.b2[0] - f.b1[0] * f.b2[2];
  f.b3[2] = f.b1[0] * f.b2[1] - f.b1[1] * f.b2[0];
}

// world -> box: rows of world2box are (b1x, b2x, b3x | -(.)t) ... exactly actor_prepare_kernel's w2b
__device__ __forceinline__ void pair_box_position(const PairFrame& f, const SamplePos& g, float flip, float* pos) {
  for (int i = 0; i < 3; ++i) {
    const float tr = -(f.b1[i] * f.t[0] + f.b2[i] * f.t[1] + f.b3[i] * f.t[2]);
    pos[i] = f.b1[i] * g.x + f.b2[i] * g.y + f.b3[i] * g.z + tr;
  }
  pos[0] *= flip;
}

__global__ __launch_bounds__(256) void actor_pair_positions_kernel(ActorsDev a, RaysDev r, const float* __restrict__ times,
                                                                   const int64_t* __restrict__ sample_idx,
                                                                   const int32_t* __restrict__ actor_idx,
                                                                   const float* __restrict__ ray_flip, int64_t n_pairs,
                                                                   float* __restrict__ x01, float* __restrict__ cstd) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pairs) return;
  const int64_t i = sample_idx[p], ray = i / r.S;
  const int s = (int)(i - ray * r.S);
  PairFrame f;
  pair_frame(a, actor_idx[p], times[ray], f);
  const SamplePos g = sample_gaussian(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray], r.d[3 * ray + 1],
                                      r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + s],
                                      r.ends[ray * r.stride + s]);
  float pos[3];
  pair_box_position(f, g, ray_flip ? ray_flip[ray] : 1.f, pos);
  const SamplePos c = contract_gaussian(pos[0], pos[1], pos[2], g.std, a.scale);
  x01[3 * p] = c.x, x01[3 * p + 1] = c.y, x01[3 * p + 2] = c.z;
  cstd[p] = c.std;
}

__global__ __launch_bounds__(256) void actor_pair_positions_bwd_kernel(
    ActorsDev a, RaysDev r, const float* __restrict__ times, const int64_t* __restrict__ sample_idx,
    const int32_t* __restrict__ actor_idx, const float* __restrict__ ray_flip, int64_t n_pairs,
    const float* __restrict__ g_x01, const float* __restrict__ g_cstd, float* __restrict__ g_positions,
    float* __restrict__ g_rot6, float* __restrict__ g_origins, float* __restrict__ g_directions, int combine_runs) {
  // (no early exit: the lanes of a 16-lane row merge their contributions on DPP shifts below; lanes past the end work on the
  //  last pair with their values zeroed)
  const int64_t p0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = p0 < n_pairs;
  const int64_t p = live ? p0 : n_pairs - 1;
  const int lane = threadIdx.x & 63;
  const int64_t i = sample_idx[p], ray = i / r.S;
  const int s = (int)(i - ray * r.S);
  const int act = actor_idx[p];
  PairFrame f;
  pair_frame(a, act, times[ray], f);
  const SamplePos g = sample_gaussian(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray], r.d[3 * ray + 1],
                                      r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + s],
                                      r.ends[ray * r.stride + s]);
  const float flip = ray_flip ? ray_flip[ray] : 1.f;
  float pos[3];
  pair_box_position(f, g, flip, pos);
  // ---- contraction backward (spatial_distortions.py:126-141, order = inf): (g_x01, g_cstd) -> g_pos ----------------
  float m[3] = {pos[0] / a.scale, pos[1] / a.scale, pos[2] / a.scale};
  const float am[3] = {fabsf(m[0]), fabsf(m[1]), fabsf(m[2])};
  const int kmax = am[0] >= am[1] ? (am[0] >= am[2] ? 0 : 2) : (am[1] >= am[2] ? 1 : 2);
  const float mag = am[kmax];
  float gm[3] = {g_x01[3 * p] / 4.f, g_x01[3 * p + 1] / 4.f, g_x01[3 * p + 2] / 4.f};  // x01 = (m' + 2) / 4
  if (!(mag < 1.f)) {
    // m' = k m, k = 2/mag - 1/mag^2;   cstd = (std/scale) q / 4, q = ((2 mag - 1)^(1/3) / mag)^2
    const float k = 2.f / mag - 1.f / (mag * mag), dk = -2.f / (mag * mag) + 2.f / (mag * mag * mag);
    const float cr = cbrtf(2.f * mag - 1.f), rr = cr / mag;
    const float dq = 2.f * rr * ((2.f / 3.f) / (cr * cr * mag) - cr / (mag * mag));
    const float g_mag = (gm[0] * m[0] + gm[1] * m[1] + gm[2] * m[2]) * dk + g_cstd[p] * (g.std / a.scale) * dq / 4.f;
    for (int c = 0; c < 3; ++c) gm[c] *= k;
    gm[kmax] += g_mag * (m[kmax] < 0.f ? -1.f : 1.f);
  }
  float gpos[3] = {gm[0] / a.scale * flip, gm[1] / a.scale, gm[2] / a.scale};
  // ---- pos_i = b1_i v_0 + b2_i v_1 + b3_i v_2,  v = mean - t ---------------------------------------------------------
  const float v[3] = {g.x - f.t[0], g.y - f.t[1], g.z - f.t[2]};
  float gb1[3], gb2[3], gb3[3], gt[3];
  for (int c = 0; c < 3; ++c) gb1[c] = v[0] * gpos[c], gb2[c] = v[1] * gpos[c], gb3[c] = v[2] * gpos[c];
  gt[0] = -dot3(f.b1, gpos), gt[1] = -dot3(f.b2, gpos), gt[2] = -dot3(f.b3, gpos);
  if (g_origins && live) {
    // d pos / d mean = -d pos / d t: the sample's world position moves with the ray (camera optimizer,
    // cameras/camera_optimizers.py:173-182); mean = o + d t_mid with t_mid a constant (detached bins)
    const float t0 = r.starts[ray * r.stride + s], t1 = r.ends[ray * r.stride + s];
    const float tm = t0 + (t1 - t0) / 2.f;
    for (int c = 0; c < 3; ++c) {
      atomicAdd(g_origins + 3 * ray + c, -gt[c]);
      atomicAdd(g_directions + 3 * ray + c, -gt[c] * tm);
    }
  }
  // b3 = b1 x b2:  g_b1 += b2 x g_b3,  g_b2 += g_b3 x b1
  gb1[0] += f.b2[1] * gb3[2] - f.b2[2] * gb3[1];
  gb1[1] += f.b2[2] * gb3[0] - f.b2[0] * gb3[2];
  gb1[2] += f.b2[0] * gb3[1] - f.b2[1] * gb3[0];
  gb2[0] += gb3[1] * f.b1[2] - gb3[2] * f.b1[1];
  gb2[1] += gb3[2] * f.b1[0] - gb3[0] * f.b1[2];
  gb2[2] += gb3[0] * f.b1[1] - gb3[1] * f.b1[0];
  float gu1[3], gu2[3];
  gram_schmidt_bwd(f.u2, f.b1, f.b2, f.n1, f.nc, f.dot, gb1, gb2, gu1, gu2);
  // ---- lerp between the two stored poses, then each pose's own Gram-Schmidt -----------------------------------------
  const float wgt[2] = {1.f - f.frac, f.frac};
  const int tix[2] = {f.left, f.right};
  // The pairs come in sample order: consecutive lanes are consecutive samples of one ray inside one box -- the same actor
  // and the same two bracketing poses -- and the trajectory tensors are a few thousand floats, so one atomic per lane and
  // component serialises memory-side (config[4] training: 1.8 M atomics onto 2.6 K addresses, 0.44 ms).  Round 5: equal
  // (pose, actor) slots of neighbouring lanes of a 16-lane row are summed onto the run's first lane on DPP row shifts and
  // only run heads issue atomics.
  for (int e = 0; e < 2; ++e) {
    const float* raw = a.rot6 + ((size_t)tix[e] * a.A + act) * 6;
    float a1[3], a2[3], n1, nc, dt, ga1[3], ga2[3], gr1[3], gr2[3];
    gram_schmidt(raw, raw + 3, a1, a2, n1, nc, dt);
    for (int c = 0; c < 3; ++c) ga1[c] = wgt[e] * gu1[c], ga2[c] = wgt[e] * gu2[c];
    gram_schmidt_bwd(raw + 3, a1, a2, n1, nc, dt, ga1, ga2, gr1, gr2);
    float v[9];
    const bool on = live && wgt[e] != 0.f;
    for (int c = 0; c < 3; ++c) {
      v[c] = on ? gr1[c] : 0.f;
      v[3 + c] = on ? gr2[c] : 0.f;
      v[6 + c] = on ? wgt[e] * gt[c] : 0.f;
    }
    const uint32_t key = live ? (uint32_t)(tix[e] * a.A + act) : 0xffffffffu;
    const bool head = live && (!combine_runs || dpp_row_shr<1>(key, ~key) != key);
    const unsigned long long hm = __ballot(head);
    if (hm != __ballot(live)) {
      const uint32_t run = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull));
#define NR_SEG_STEP(OFF)                                                   \
  {                                                                        \
    const bool same = dpp_row_shl<OFF>(run, 0xffffffffu) == run;           \
    _Pragma("unroll") for (int j = 0; j < 9; ++j) {                        \
      const float t = dpp_row_shl<OFF>(v[j], 0.f);                         \
      if (same) v[j] += t;                                                 \
    }                                                                      \
  }
      NR_SEG_STEP(1)
      NR_SEG_STEP(2)
      NR_SEG_STEP(4)
      NR_SEG_STEP(8)
#undef NR_SEG_STEP
    }
    if (head) {
      float* gr = g_rot6 + ((size_t)tix[e] * a.A + act) * 6;
      float* gp = g_positions + ((size_t)tix[e] * a.A + act) * 3;
      for (int c = 0; c < 3; ++c) {
        atomicAdd(gr + c, v[c]);
        atomicAdd(gr + 3 + c, v[3 + c]);
        atomicAdd(gp + c, v[6 + c]);
      }
    }
  }
}

// ---- proposal density of the in-box samples, training (fields/neurad_field.py:208-213 with neurad_encoding.py:150-187) ----
// density = trunc_exp(decoder . actor features) replaces the static density of every sample inside a box; of several boxes
// containing a sample the highest actor index wins (the reference's index_put order), the shadowed pairs keep a gradient
// path into THEIR actor's features (the reference hands the merged row's gradient to every duplicate index).  One pass over
// the P (sample, actor) pairs each way instead of ~25 torch launches (two of them rocblas gemv calls on a 4-wide dot
// product: 1.7 ms each at 65 536 rays).
__global__ __launch_bounds__(256) void actor_density_splice_fwd_kernel(const float* __restrict__ rows, int la,
                                                                        const float* __restrict__ w,
                                                                        const int64_t* __restrict__ idx,
                                                                        const uint8_t* __restrict__ winner, int64_t n_pairs,
                                                                        float* __restrict__ dens, float* __restrict__ logit) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n_pairs) return;
  float acc = 0.f;
  for (int k = 0; k < la; ++k) acc = fmaf(rows[p * la + k], w[k], acc);
  logit[p] = acc;
  if (winner[p]) dens[idx[p]] = expf(acc);  // one winner per sample: no race
}

__global__ __launch_bounds__(256) void actor_density_splice_bwd_kernel(
    const float* __restrict__ rows, int la, const float* __restrict__ w, const int64_t* __restrict__ idx,
    const uint8_t* __restrict__ winner, const float* __restrict__ logit, const float* __restrict__ dens_out,
    const float* __restrict__ g_out, int64_t n_pairs, float* __restrict__ g_dens, float* __restrict__ g_rows,
    float* __restrict__ g_w) {
  __shared__ float part[4][NRHIP_MAX_LEVELS];
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float gl = 0.f;  // d loss / d logit of a winner pair (the decoder's gradient comes from the winners only)
  if (p < n_pairs) {
    const int64_t s = idx[p];
    const float go = g_out[s];
    float f;
    if (winner[p]) {
      g_dens[s] = 0.f;  // the static density of a hit sample is not in the output
      gl = go * expf(fminf(fmaxf(logit[p], -15.f), 15.f));  // trunc_exp's backward (field_components/activations.py:37-41)
      f = gl;
    } else {
      f = go * dens_out[s];  // shadowed pair: (shadow - shadow.detach()) * merged value
    }
    for (int k = 0; k < la; ++k) g_rows[p * la + k] = f * w[k];
  }
  for (int k = 0; k < la; ++k) {
    float v = p < n_pairs ? gl * rows[p * la + k] : 0.f;
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) part[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < la) atomicAdd(g_w + threadIdx.x, part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x]);
}

}  // namespace nrhip

extern "C" int nrhip_actor_pair_positions_fwd(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                              const int64_t* sample_idx, const int32_t* actor_idx, const float* ray_flip,
                                              int64_t n_pairs, float* x01, float* cstd, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(n_pairs >= 0 && a->actor_scale > 0.f, NRHIP_ERR_INVALID_ARG, "actor_pair_positions: bad argument");
  if (n_pairs == 0) return NRHIP_OK;
  NR_REQUIRE(times && sample_idx && actor_idx && x01 && cstd, NRHIP_ERR_INVALID_ARG, "actor_pair_positions: NULL pointer");
  actor_pair_positions_kernel<<<grid_for(n_pairs, 256), 256, 0, (hipStream_t)stream>>>(
      d, to_dev(*rays), times, sample_idx, actor_idx, ray_flip, n_pairs, x01, cstd);
  return check_launch("actor_pair_positions_fwd");
}

extern "C" int nrhip_actor_pair_positions_bwd(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                              const int64_t* sample_idx, const int32_t* actor_idx, const float* ray_flip,
                                              int64_t n_pairs, const float* grad_x01, const float* grad_cstd,
                                              float* grad_positions, float* grad_rotations_6d, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(n_pairs >= 0 && a->actor_scale > 0.f, NRHIP_ERR_INVALID_ARG, "actor_pair_positions_bwd: bad argument");
  if (n_pairs == 0) return NRHIP_OK;
  NR_REQUIRE(times && sample_idx && actor_idx && grad_x01 && grad_cstd && grad_positions && grad_rotations_6d,
             NRHIP_ERR_INVALID_ARG, "actor_pair_positions_bwd: NULL pointer");
  actor_pair_positions_bwd_kernel<<<grid_for(n_pairs, 256), 256, 0, (hipStream_t)stream>>>(
      d, to_dev(*rays), times, sample_idx, actor_idx, ray_flip, n_pairs, grad_x01, grad_cstd, grad_positions,
      grad_rotations_6d, nullptr, nullptr, tuning().pair_bwd_runs ? 1 : 0);
  return check_launch("actor_pair_positions_bwd");
}

extern "C" int nrhip_actor_pair_positions_bwd_rays(const nrhip_actors* a, const nrhip_rays* rays, const float* times,
                                                   const int64_t* sample_idx, const int32_t* actor_idx,
                                                   const float* ray_flip, int64_t n_pairs, const float* grad_x01,
                                                   const float* grad_cstd, float* grad_positions,
                                                   float* grad_rotations_6d, float* grad_origins,
                                                   float* grad_directions, void* stream) {
  ActorsDev d;
  if (int e = to_dev(a, d)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(n_pairs >= 0 && a->actor_scale > 0.f, NRHIP_ERR_INVALID_ARG, "actor_pair_positions_bwd_rays: bad argument");
  if (n_pairs == 0) return NRHIP_OK;
  NR_REQUIRE(times && sample_idx && actor_idx && grad_x01 && grad_cstd && grad_positions && grad_rotations_6d &&
                 grad_origins && grad_directions,
             NRHIP_ERR_INVALID_ARG, "actor_pair_positions_bwd_rays: NULL pointer");
  actor_pair_positions_bwd_kernel<<<grid_for(n_pairs, 256), 256, 0, (hipStream_t)stream>>>(
      d, to_dev(*rays), times, sample_idx, actor_idx, ray_flip, n_pairs, grad_x01, grad_cstd, grad_positions,
      grad_rotations_6d, grad_origins, grad_directions, tuning().pair_bwd_runs ? 1 : 0);
  return check_launch("actor_pair_positions_bwd_rays");
}

// ---- the (sample, actor) pairs of a hits table, in order, without torch's nonzero -----------------------------------------
// `(hits >= 0).nonzero()` + `hits[idx, slot]` is a compare, a rocprim partition (161 us at 65 536 rays x 32 samples), a block
// reduce, a gather and two casts per field per step.  Here: per-block counts, one single-workgroup scan (block offsets +
// total), then every block writes its pairs at its offset in (sample, slot) order -- two streaming passes over the table.
namespace nrhip {
namespace {
constexpr int kPairBlock = 1024;  // samples per workgroup (256 threads x 4)

__global__ __launch_bounds__(256) void actor_pairs_count_kernel(const int32_t* __restrict__ hits, int64_t n,
                                                                 uint32_t* __restrict__ block_counts) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t c = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t i = (int64_t)blockIdx.x * kPairBlock + u * 256 + threadIdx.x;
    if (i < n) {
      const int4 a = reinterpret_cast<const int4*>(hits + i * KH)[0], b = reinterpret_cast<const int4*>(hits + i * KH)[1];
      c += (a.x >= 0) + (a.y >= 0) + (a.z >= 0) + (a.w >= 0) + (b.x >= 0) + (b.y >= 0) + (b.z >= 0) + (b.w >= 0);
    }
  }
#pragma unroll
  for (int off = 32; off; off >>= 1) c += __shfl_xor(c, off, 64);
  if (lane == 0) wsum[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive prefix of block_counts (in place) + the total; one workgroup
__global__ __launch_bounds__(1024) void actor_pairs_scan_kernel(uint32_t* __restrict__ block_counts, int nblk,
                                                                 int64_t* __restrict__ total) {
  __shared__ uint32_t part[1024];
  const int per = (nblk + 1023) / 1024, c0 = threadIdx.x * per;
  uint32_t s = 0;
  for (int k = 0; k < per; ++k)
    if (c0 + k < nblk) s += block_counts[c0 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;
  for (int k = 0; k < per; ++k)
    if (c0 + k < nblk) {
      const uint32_t v = block_counts[c0 + k];
      block_counts[c0 + k] = run;
      run += v;
    }
  if (threadIdx.x == 1023) *total = (int64_t)part[1023];
}

__global__ __launch_bounds__(256) void actor_pairs_write_kernel(const int32_t* __restrict__ hits, int64_t n,
                                                                 const uint32_t* __restrict__ block_offsets,
                                                                 const int64_t* __restrict__ total,
                                                                 int64_t* __restrict__ sample_idx,
                                                                 int32_t* __restrict__ actor_idx) {
  __shared__ uint32_t wsum[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t base = block_offsets[blockIdx.x];
  // most 1024-sample blocks hold no pair at all (5 % of the samples lie in a box, clustered along the rays that meet one)
  const uint32_t next = blockIdx.x + 1 < gridDim.x ? block_offsets[blockIdx.x + 1] : (uint32_t)*total;
  if (next == base) return;
  for (int u = 0; u < 4; ++u) {  // samples in order: pass u covers 256 consecutive samples
    const int64_t i = (int64_t)blockIdx.x * kPairBlock + u * 256 + threadIdx.x;
    int h[KH];
    uint32_t c = 0;
    static_assert(KH == 8, "two 16-byte loads per row");
    if (i < n) {
      const int4 a = reinterpret_cast<const int4*>(hits + i * KH)[0], b = reinterpret_cast<const int4*>(hits + i * KH)[1];
      h[0] = a.x, h[1] = a.y, h[2] = a.z, h[3] = a.w, h[4] = b.x, h[5] = b.y, h[6] = b.z, h[7] = b.w;
    } else {
#pragma unroll
      for (int k = 0; k < KH; ++k) h[k] = -1;
    }
#pragma unroll
    for (int k = 0; k < KH; ++k) c += h[k] >= 0 ? 1u : 0u;
    uint32_t incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = __shfl_up(incl, off, 64);
      if (lane >= off) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      before += w < wave ? wsum[w] : 0u;
      tot += wsum[w];
    }
    uint32_t pos = base + before + incl - c;
#pragma unroll
    for (int k = 0; k < KH; ++k)
      if (h[k] >= 0) {
        sample_idx[pos] = i;
        actor_idx[pos] = h[k];
        ++pos;
      }
    base += tot;
    __syncthreads();
  }
}
}  // namespace
}  // namespace nrhip

extern "C" int nrhip_actor_pairs_count(const int32_t* hits, int64_t n_samples, uint32_t* block_offsets, int64_t* total,
                                       void* stream) {
  NR_REQUIRE(n_samples >= 0 && total && (n_samples == 0 || (hits && block_offsets)), NRHIP_ERR_INVALID_ARG,
             "actor_pairs_count: bad argument");
  NR_REQUIRE((reinterpret_cast<uintptr_t>(hits) & 15) == 0, NRHIP_ERR_INVALID_ARG, "actor_pairs_count: hits must be 16-byte aligned");
  const int nblk = (int)((n_samples + kPairBlock - 1) / kPairBlock);
  if (nblk == 0) {
    if (hipMemsetAsync(total, 0, sizeof(int64_t), (hipStream_t)stream) != hipSuccess) return check_launch("actor_pairs_count");
    return NRHIP_OK;
  }
  actor_pairs_count_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(hits, n_samples, block_offsets);
  actor_pairs_scan_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(block_offsets, nblk, total);
  return check_launch("actor_pairs_count");
}

extern "C" int nrhip_actor_pairs_write(const int32_t* hits, int64_t n_samples, const uint32_t* block_offsets,
                                       const int64_t* total, int64_t* sample_idx, int32_t* actor_idx, void* stream) {
  NR_REQUIRE(n_samples >= 0, NRHIP_ERR_INVALID_ARG, "actor_pairs_write: bad argument");
  if (n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(hits && block_offsets && total && sample_idx && actor_idx, NRHIP_ERR_INVALID_ARG,
             "actor_pairs_write: NULL pointer");
  const int nblk = (int)((n_samples + kPairBlock - 1) / kPairBlock);
  actor_pairs_write_kernel<<<nblk, 256, 0, (hipStream_t)stream>>>(hits, n_samples, block_offsets, total, sample_idx,
                                                                  actor_idx);
  return check_launch("actor_pairs_write");
}

extern "C" int nrhip_actor_density_splice_fwd(const float* rows, int32_t row_dim, const float* decoder_weight,
                                              const int64_t* sample_idx, const uint8_t* winner, int64_t n_pairs,
                                              float* density, float* logit, void* stream) {
  NR_REQUIRE(n_pairs >= 0 && row_dim >= 1 && row_dim <= NRHIP_MAX_LEVELS, NRHIP_ERR_INVALID_ARG,
             "actor_density_splice_fwd: bad argument");
  if (n_pairs == 0) return NRHIP_OK;
  NR_REQUIRE(rows && decoder_weight && sample_idx && winner && density && logit, NRHIP_ERR_INVALID_ARG,
             "actor_density_splice_fwd: NULL pointer");
  actor_density_splice_fwd_kernel<<<grid_for(n_pairs, 256), 256, 0, (hipStream_t)stream>>>(
      rows, row_dim, decoder_weight, sample_idx, winner, n_pairs, density, logit);
  return check_launch("actor_density_splice_fwd");
}

extern "C" int nrhip_actor_density_splice_bwd(const float* rows, int32_t row_dim, const float* decoder_weight,
                                              const int64_t* sample_idx, const uint8_t* winner, const float* logit,
                                              const float* density_out, const float* grad_out, int64_t n_pairs,
                                              float* grad_density, float* grad_rows, float* grad_decoder, void* stream) {
  NR_REQUIRE(n_pairs >= 0 && row_dim >= 1 && row_dim <= NRHIP_MAX_LEVELS, NRHIP_ERR_INVALID_ARG,
             "actor_density_splice_bwd: bad argument");
  if (n_pairs == 0) return NRHIP_OK;
  NR_REQUIRE(rows && decoder_weight && sample_idx && winner && logit && density_out && grad_out && grad_density && grad_rows &&
                 grad_decoder,
             NRHIP_ERR_INVALID_ARG, "actor_density_splice_bwd: NULL pointer");
  actor_density_splice_bwd_kernel<<<grid_for(n_pairs, 256), 256, 0, (hipStream_t)stream>>>(
      rows, row_dim, decoder_weight, sample_idx, winner, logit, density_out, grad_out, n_pairs, grad_density, grad_rows,
      grad_decoder);
  return check_launch("actor_density_splice_bwd");
}
