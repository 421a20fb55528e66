// B1 (table gradient) without device-scope atomics.
//
// On MI355X every fp32 global atomic is executed memory-side: rocprofv3 shows TCC_EA0_ATOMIC == TCC_ATOMIC for the
// scatter-add kernel (hashgrid.hip), i.e. each of its ~55 M atomic requests is one fabric transaction, and the
// fabric retires ~1.8e10 of them per second no matter where the line lives -> 3.1 ms for config 2, 44 % of a
// training step.  The per-XCD L2s are not coherent with each other, so there is no cheaper scope to fall back to.
//
// This path gives every table entry exactly one owner instead -- a radix partition by table slice:
// Samples whose incoming gradient is EXACTLY zero (samples behind an opaque surface: weight 0, transmittance 0) send no
// records at all: `prep` drops them -- each 4096-sample chunk is compacted, in order, to its live samples -- and `count`
// / `emit` walk only those.  Adding zeros changes nothing, so the result is bit-identical; on a uniformly sampled ray of
// a converged scene the silent samples are the majority (bench config[1]: 76 %).
//   count   : the table of a level is cut into slices of TS = 16384/F entries (128 KB of 64-bit accumulators).
//             A workgroup takes 4096 consecutive samples, parks their contracted positions in LDS (and in the
//             scratch, for `emit`), and for each of its levels histograms, in LDS, the records its samples send to
//             every slice.  Runs of equal entries in neighbouring lanes of a 16-lane row (consecutive samples of a
//             ray in the same coarse cell) count once: they are summed before they are emitted.
//   scan    : exclusive prefix sums over the workgroups (per slice) and over the slices -> every workgroup's write
//             position in every slice queue, exactly; queues are as long as their slice needs, whatever the sample
//             distribution (real scenes put most samples into a thin slab of the volume: a fixed per-slice capacity
//             overflowed there, and the overflow went to the memory-side atomics this path exists to avoid).
//   emit    : same walk as `count`; each record goes to base + LDS rank.  No global atomics.
//             At F = 1 a record is an X-PAIR (round 4; wider grids keep one corner term per record, see use_pairs): the
//             (floor x, ceil x) corners of one (y, z) hash to entries that differ
//             only by xm = (floor x ^ ceil x) & mask -- the same small value for all four pairs of a sample -- so they
//             lie in the same slice and one record {entry-in-slice | xm << 16, F values for the floor corner, F values
//             for the ceil corner} carries both: 4 records per (sample, level) instead of 8, half the LDS rank atomics
//             and store transactions, 12 instead of 16 bytes per corner pair at F = 1 (36 / 40 at F = 4).  Where xm
//             reaches the slice bits (only possible when a level's resolution exceeds the slice length) the pair goes
//             out as two records with xm = 0.
//   reduce  : one 1024-thread workgroup per (level, slice) streams its queue, accumulates into the slice image in
//             LDS, and adds the image to grad_table with plain 16-byte loads/stores.  The image is 64-bit FIXED
//             POINT: ds_add_f32 turned out ~10x slower than the integer LDS atomics on gfx950 (493 vs 147 us for
//             this pass), so values are scaled by a power of two chosen from the level's largest |value| (found by
//             `emit`) and the slice's record count so that the sum cannot overflow, and added with ds_add_u64.  The
//             quantum is below 2^-40 of the level's largest term -- finer than the fp32 rounding of the atomic path
//             for anything that matters -- and integer addition is associative: the result is bit-reproducible.
//             Inf/NaN values poison their entry (NaN out), as an atomic add of them would.
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

#include "common.h"

namespace nrhip {
namespace {

constexpr int kSamplesPerBlock = 4096;      // count/emit: 1024 threads x 4 samples
constexpr int kMaxSlices = 2048;            // per level (LDS histogram + base table = 16 KB)
constexpr int kTileBytes = 128 * 1024;      // slice image in LDS
constexpr int kSegChunks = 64;              // the chunk-prefix scan runs per segment of 64 chunks, then over the segments
// x-pairs of the reference corner order (common.h hash_corners): (floor x, ceil x) corner of (y, z) = cc, fc, cf, ff
__device__ constexpr int kPairF[4] = {3, 2, 7, 6};
__device__ constexpr int kPairC[4] = {0, 1, 4, 5};

// Samples per count/scan/emit/reduce round: bounds the scratch (one record slot per corner term).  Round 2 used 2^20
// (400 MB for the proposal grid); the c3 step then ran 11 rounds of seven launches each for its two proposal calls.  With
// 288 GB of HBM a round of 2^23 samples (3.4 GB for that grid) covers every call of the step in one: the fixed cost per
// round (memset, two scan launches, the tails of five kernels) is paid once, and with >= 1024 chunks a count / emit
// workgroup walks all levels of its samples instead of one (positions and per-sample factors loaded once).
// NRHIP_BIN_ROUND_LOG2 (15..24) overrides, for A/B runs.
int64_t round_samples() { return (int64_t)1 << tuning().bin_round_log2; }

// X-pair records pay where the record is small: F = 1 (12 instead of 2 x 8 bytes per corner pair, half the LDS rank atomics:
// emit<1> 539 -> 393 us per c3 call).  At F = 4 they were measured SLOWER (emit 754 -> 848, reduce 592 -> 858 us: 36-byte
// records, 8 values per DPP chain), so wider grids keep one corner term per record.  NRHIP_BIN_PAIRS=all|none overrides (A/B).
bool use_pairs(int F) {
  if (tuning().bin_pairs >= 0) return tuning().bin_pairs == 1;
  return F == 1;
}

struct BinPlan {
  int log2TS, nb;        // entries per slice (log2), slices per level
  int chunks, lgroups;   // count/emit grid: sample chunks x level groups (levels dealt round-robin)
  int nmax;              // per-level partial maxima (one per emit wave)
  int nseg;              // segments of kSegChunks chunks (two-level prefix over the chunks)
  bool pair;             // records carry x-pairs of corner terms (else one corner term each)
  int rec_slots;         // record slots per (sample, level): 4 x-pairs (8 if pairs can straddle slices), or 8 corners
  size_t off_counts, off_seg, off_totals, off_offsets, off_qmax, off_pos, off_idx, off_live, off_rec, total_bytes;
};

size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// A record is RW = 2F + 1 dwords at 4-byte alignment.  Moved as 12-byte vectors (global_load/store_dwordx3 and wider: gfx950
// takes unaligned vector accesses) instead of RW dword accesses: a wave's strided record reads touch every cache line once
// per INSTRUCTION, so nine dword loads of 36-byte records cost the L1 nine passes over the same lines (reduce<4> went from
// 592 to 849 us when the records grew from 5 to 9 dwords), three 12-byte loads three.
typedef float rec3 __attribute__((ext_vector_type(3)));
typedef rec3 rec3u __attribute__((aligned(4)));
typedef float rec2 __attribute__((ext_vector_type(2)));
typedef rec2 rec2u __attribute__((aligned(4)));
template <int RW>
__device__ __forceinline__ void load_record(const float* __restrict__ p, float (&q)[RW]) {
  constexpr int n3 = RW / 3, rem = RW - 3 * n3;
#pragma unroll
  for (int c = 0; c < n3; ++c) {
    const rec3 v = *reinterpret_cast<const rec3u*>(p + 3 * c);
    q[3 * c] = v.x, q[3 * c + 1] = v.y, q[3 * c + 2] = v.z;
  }
  if constexpr (rem == 2) {
    const rec2 v = *reinterpret_cast<const rec2u*>(p + 3 * n3);
    q[3 * n3] = v.x, q[3 * n3 + 1] = v.y;
  } else if constexpr (rem == 1) {
    q[3 * n3] = p[3 * n3];
  }
}
template <int RW>
__device__ __forceinline__ void store_record(float* __restrict__ p, const float (&q)[RW]) {
  constexpr int n3 = RW / 3, rem = RW - 3 * n3;
#pragma unroll
  for (int c = 0; c < n3; ++c) {
    rec3 v;
    v.x = q[3 * c], v.y = q[3 * c + 1], v.z = q[3 * c + 2];
    *reinterpret_cast<rec3u*>(p + 3 * c) = v;
  }
  if constexpr (rem == 2) {
    rec2 v;
    v.x = q[3 * n3], v.y = q[3 * n3 + 1];
    *reinterpret_cast<rec2u*>(p + 3 * n3) = v;
  } else if constexpr (rem == 1) {
    p[3 * n3] = q[3 * n3];
  }
}

// Returns false when the grid can not be binned (too many slices per level).
// n_slots > 1: the MULTI-GRID form (the per-actor grids, MultiSrc below) -- n_slots tables of the shape `g` side by side; a
// level's slices are then (slot, slice-of-that-table) pairs, nb = n_slots x the slices of one table, and everything downstream
// (histograms, prefix sums, queues, one `reduce` workgroup per slice) is the single-grid machinery over those columns.
bool make_plan(const GridDev& g, int64_t n_total, BinPlan* p, int n_slots = 1) {
  const int64_t n = n_total < round_samples() ? n_total : round_samples();  // larger batches go through in rounds
  int log2F = 0;
  while ((1 << log2F) < g.F) ++log2F;
  int log2TS = 14 - log2F;  // 8-byte accumulators: 16384 / F entries fill the 128 KB image (64 KB images, two `reduce`
                            // workgroups per CU: measured the same, 8.94-9.00 vs 8.95-9.03 ms per c3 step)
  if (log2TS > g.log2T) log2TS = g.log2T;
  // small tables: shrink the slices until `reduce` has ~2 workgroups per CU (one workgroup owns one slice)
  while ((((int64_t)g.L * n_slots) << (g.log2T - log2TS)) < 512 && log2TS > 9) --log2TS;
  const int64_t nb = ((int64_t)1 << (g.log2T - log2TS)) * n_slots;
  if (nb > kMaxSlices) return false;
  p->log2TS = log2TS;
  p->nb = (int)nb;
  p->chunks = (int)((n + kSamplesPerBlock - 1) / kSamplesPerBlock);
  int lg = (1024 + p->chunks - 1) / p->chunks;  // ~4 workgroups of 1024 threads per CU
  lg = lg < 1 ? 1 : (lg > g.L ? g.L : lg);
  while (g.L % lg != 0) ++lg;  // every level group the same number of levels: 6 proposal levels over 4 groups left a
                               // third of the workgroups with twice the work of the others
  p->lgroups = lg;
  p->nmax = p->chunks * (kSamplesPerBlock / 4 / 64);
  p->nseg = (p->chunks + kSegChunks - 1) / kSegChunks;
  const size_t cols = (size_t)g.L * nb;
  size_t o = 0;
  p->off_counts = o, o += align256((size_t)p->chunks * cols * sizeof(uint32_t));
  p->off_seg = o, o += align256((size_t)p->nseg * cols * sizeof(uint32_t));
  p->off_totals = o, o += align256(cols * sizeof(uint32_t));
  p->off_offsets = o, o += align256((cols + 1) * sizeof(uint32_t));
  p->off_qmax = o, o += align256((size_t)g.L * p->nmax * sizeof(float));
  p->off_pos = o, o += align256((size_t)p->chunks * kSamplesPerBlock * sizeof(float4));
  p->off_idx = o, o += align256((size_t)p->chunks * kSamplesPerBlock * sizeof(uint16_t));
  p->off_live = o, o += align256((size_t)p->chunks * sizeof(uint32_t));
  // worst case every x-pair its own record: 4 per (sample, level); 8 where a pair can straddle two slices (a level whose
  // resolution reaches the slice length: floor x ^ ceil x can then carry into the slice bits)
  float smax = 0.f;
  for (int l = 0; l < g.L; ++l) smax = g.scal[l] > smax ? g.scal[l] : smax;
  p->pair = use_pairs(g.F);
  p->rec_slots = (p->pair && smax + 2.f < (float)(1 << log2TS)) ? 4 : 8;
  p->off_rec = o, o += (size_t)n * p->rec_slots * g.L * ((p->pair ? 2 : 1) * g.F + 1) * sizeof(float);
  p->total_bytes = o;
  return true;
}

// Is the gradient row of sample (w0 + lane) all +-0?  (NaN compares unequal to 0: a poisoned row stays alive and poisons,
// as before.)  Called by whole waves, w0 wave-uniform; rows are `width` floats, [0, n_rows) are valid.  32-float rows --
// every field grid -- are read the way they lie in memory: the wave's 64 rows are 8 KB in a row, lane l takes the
// float4s l, l + 64, ... and a ballot per load tells which of its 8 rows saw a non-zero.
__device__ __forceinline__ bool rows_are_zero(const float* __restrict__ g, int width, int64_t w0, int64_t n_rows) {
  const int lane = threadIdx.x & 63;
  if (width == 32 && (reinterpret_cast<uintptr_t>(g) & 15) == 0) {
    bool alive = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t row = w0 + 8 * k + (lane >> 3);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < n_rows) v = reinterpret_cast<const float4*>(g + w0 * 32)[k * 64 + lane];
      const unsigned long long nz = __ballot(!(v.x == 0.f && v.y == 0.f && v.z == 0.f && v.w == 0.f));
      if ((lane >> 3) == k) alive = ((nz >> (8 * (lane & 7))) & 0xffull) != 0ull;
    }
    return !alive;
  }
  const int64_t row = w0 + lane;
  bool z = true;
  if (row < n_rows)
    for (int k = 0; k < width; ++k) z = z && g[row * width + k] == 0.f;
  return z;
}

// Where the samples and their feature gradients come from (pass A is otherwise identical):
//   position(i) -> (x, y, z in [0,1]^3, aux);  grad(i, l, level scale, aux, gv[F]) -> dL/d(level-l features of sample i)
__device__ __forceinline__ int coherent_rays_of(const RaysDev& r, int64_t first_sample, int64_t count);  // (see `prep`)

struct EncodeSrc {  // H2+H3+H1+H4 (nrhip_encode_bwd): positions from ray samples, gradient of the rescaled features
  static constexpr bool kMulti = false;
  RaysDev r;
  float scale;
  const float* go;
  int L;
  __device__ float4 position(int64_t i) const {
    const int64_t ray = i / r.S;
    const int s = (int)(i - ray * r.S);
    const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                        r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + s],
                                        r.ends[ray * r.stride + s], scale);
    return make_float4(p.x, p.y, p.z, p.std);
  }
  __device__ float pre(int64_t) const { return 0.f; }
  template <int F>
  __device__ void grad(int64_t i, int l, float sc, float std, float, float (&gv)[F]) const {
    const float rw = rescale_weight(sc, std);
#pragma unroll
    for (int k = 0; k < F; ++k) gv[k] = go[(i * L + l) * F + k] * rw;
  }
  int width;  // L * F
  __device__ bool silent(int64_t w0, int64_t n_rows) const { return rows_are_zero(go, width, w0, n_rows); }
  __device__ int coherent_rays(int64_t first, int64_t count) const { return coherent_rays_of(r, first, count); }
};
struct GridSrc {  // H1 (nrhip_hashgrid_bwd): positions given
  static constexpr bool kMulti = false;
  const float* x;
  const float* go;
  int L;
  __device__ float4 position(int64_t i) const { return make_float4(x[3 * i], x[3 * i + 1], x[3 * i + 2], 0.f); }
  __device__ float pre(int64_t) const { return 0.f; }
  template <int F>
  __device__ void grad(int64_t i, int l, float, float, float, float (&gv)[F]) const {
#pragma unroll
    for (int k = 0; k < F; ++k) gv[k] = go[(i * L + l) * F + k];
  }
  int width;  // L * F
  __device__ bool silent(int64_t w0, int64_t n_rows) const { return rows_are_zero(go, width, w0, n_rows); }
  __device__ int coherent_rays(int64_t, int64_t) const { return 0; }  // bare positions: no rays to compare
};
struct MultiSrc {  // H5's per-actor grids (nrhip_hashgrid_multi_bwd_binned): positions given, ONE of n_grids tables per sample
  static constexpr bool kMulti = true;  // `count` / `emit` read the slot from the position's fourth component
  const float* x;
  const float* go;
  const int* gid;      // [N] grid of each sample (< 0 or >= n_grids: the sample sends nothing)
  const int* slot_of;  // [n_grids] position of the grid's gradient in the output block, < 0: no gradient wanted
  int n_grids;
  int L;
  int width;  // L * F
  __device__ int slot(int64_t i) const {
    const int g_ = gid[i];
    return (g_ >= 0 && g_ < n_grids) ? slot_of[g_] : -1;
  }
  // (the fourth component is the std of the ray-sample sources; here it carries the slot to `count` / `emit`)
  __device__ float4 position(int64_t i) const { return make_float4(x[3 * i], x[3 * i + 1], x[3 * i + 2], __int_as_float(slot(i))); }
  __device__ float pre(int64_t) const { return 0.f; }
  template <int F>
  __device__ void grad(int64_t i, int l, float, float, float, float (&gv)[F]) const {
#pragma unroll
    for (int k = 0; k < F; ++k) gv[k] = go[(i * L + l) * F + k];
  }
  __device__ bool silent(int64_t w0, int64_t n_rows) const {
    bool z = rows_are_zero(go, width, w0, n_rows);
    const int64_t i = w0 + (threadIdx.x & 63);
    if (i < n_rows && slot(i) < 0) z = true;
    return z;
  }
  __device__ int coherent_rays(int64_t, int64_t) const { return 0; }
};
struct ProposalSrc {  // S2 (nrhip_proposal_density_bwd): density = trunc_exp(decoder . rescaled features), F = 1
  static constexpr bool kMulti = false;
  RaysDev r;
  float scale;
  const float* dec;
  const float* dens;
  const float* gd;
  __device__ float4 position(int64_t i) const {
    const int64_t ray = i / r.S;
    const int s = (int)(i - ray * r.S);
    const SamplePos p = sample_position(r.o[3 * ray], r.o[3 * ray + 1], r.o[3 * ray + 2], r.d[3 * ray],
                                        r.d[3 * ray + 1], r.d[3 * ray + 2], r.area[ray], r.starts[ray * r.stride + s],
                                        r.ends[ray * r.stride + s], scale);
    return make_float4(p.x, p.y, p.z, p.std);
  }
  // the level-independent factor of a sample's gradient, once per sample (a workgroup may walk all levels)
  __device__ float pre(int64_t i) const {
    const float xlog = logf(dens[i]);  // activations.py:37-41: g * exp(clamp(x, -15, 15))
    return gd[i] * expf(fminf(fmaxf(xlog, -15.f), 15.f));
  }
  template <int F>
  __device__ void grad(int64_t, int l, float sc, float std, float gx, float (&gv)[F]) const {
    gv[0] = gx * dec[l] * rescale_weight(sc, std);
  }
  __device__ bool silent(int64_t w0, int64_t n_rows) const {  // (exp(.) > 0: the chain factor cannot revive it)
    const int64_t i = w0 + (threadIdx.x & 63);
    return i < n_rows ? gd[i] == 0.f : true;
  }
  __device__ int coherent_rays(int64_t first, int64_t count) const { return coherent_rays_of(r, first, count); }
};

// ---- prep ----------------------------------------------------------------------------------------------------
// Positions of the round's LIVE samples, once, for `count` and `emit`: a sample that is out of range or whose incoming
// gradient is exactly zero sends no records and is dropped here.  Each 4096-sample chunk is compacted in order (so that
// consecutive samples of a ray stay neighbours and their equal entries still merge): gpos / gidx [chunk][slot] = position
// / index inside the chunk of the slot-th live sample, nlive[chunk] = their number.
// Walk order inside a chunk (round 4).  `count` / `emit` merge equal entries of NEIGHBOURING LANES, so what the lanes of a
// 16-lane row hold decides how many records go out.  Ray-major (a row = 16 consecutive samples of one ray) merges the runs of
// one ray through a coarse cell.  When the chunk's rays are neighbours themselves -- the 32 rays of a pixel row of a camera
// patch: one origin, directions a fraction of a degree apart -- walking the chunk SAMPLE-INDEX-major (a row = 16 neighbouring
// rays at one sample index) merges far more: adjacent rays sit |jitter difference| x bin width apart along the ray, a third
// of the distance between consecutive samples of one ray.  Measured on the c3 step (scripts/record_stats_probe.py,
// profiles/r04_record_stats.txt): records per corner-pair term 0.78 -> 0.48 (first proposal round), 0.64 -> 0.49 (second),
// 0.49 -> 0.35 (field) on the camera rays; lidar rays (unrelated neighbours) keep the ray-major walk.  Decided per chunk
// from the rays themselves (coherent_rays); any order is correct -- `emit` finds its source sample through gidx.
// Round 5 built a third walk -- QUADS: a thread owns four consecutive samples of a ray, neighbouring lanes are neighbouring
// rays, equal entries merge first inside the thread and then across the row (the "2-D merge") -- and measured it: records
// -3.4 % (field) / -5.6 % (proposal rounds) on the c3 step, but the count / emit structure it needs (four slots' corners live at
// once) is slower than this loop: c1 train 1.42 -> 1.48 ms, c3 step 9.11 -> 9.21 ms on one box.  Reverted; the patch and
// the numbers: profiles/r05_quad_walk_rejected.diff, r05_record_stats_walks.txt, r05_ab_table_gradient_r04_vs_r05.txt.
__device__ __forceinline__ int coherent_rays_of(const RaysDev& r, int64_t first_sample, int64_t count) {
  const int S = r.S;
  if (S < 16 || S > kSamplesPerBlock / 16 || kSamplesPerBlock % S != 0 || count < kSamplesPerBlock) return 0;
  const int nr = kSamplesPerBlock / S;
  const int64_t r0 = first_sample / S, r1 = r0 + 1, r2 = r0 + nr - 1;
  const float ox = r.o[3 * r0], oy = r.o[3 * r0 + 1], oz = r.o[3 * r0 + 2];
  const float tol = 1e-4f * (1.f + fabsf(ox) + fabsf(oy) + fabsf(oz));
  const bool same_o = fabsf(r.o[3 * r1] - ox) + fabsf(r.o[3 * r1 + 1] - oy) + fabsf(r.o[3 * r1 + 2] - oz) <= tol &&
                      fabsf(r.o[3 * r2] - ox) + fabsf(r.o[3 * r2 + 1] - oy) + fabsf(r.o[3 * r2 + 2] - oz) <= tol;
  const float dx = r.d[3 * r0], dy = r.d[3 * r0 + 1], dz = r.d[3 * r0 + 2];
  const float c1 = dx * r.d[3 * r1] + dy * r.d[3 * r1 + 1] + dz * r.d[3 * r1 + 2];
  const float c2 = dx * r.d[3 * r2] + dy * r.d[3 * r2 + 1] + dz * r.d[3 * r2 + 2];
  return (same_o && c1 > 0.9999f && c2 > 0.99f) ? nr : 0;  // (unit directions: < 0.8 deg to the next ray, < 8 deg across the chunk)
}

bool transposed_walk_enabled() { return tuning().bin_transpose; }  // NRHIP_BIN_TRANSPOSE=0: ray-major everywhere (A/B)

template <class Src>
__global__ __launch_bounds__(1024) void bin_prep_kernel(Src src, int64_t i_off, int64_t n, int64_t n_total,
                                                         float4* __restrict__ gpos, uint16_t* __restrict__ gidx,
                                                         uint32_t* __restrict__ nlive, int allow_transpose) {
  __shared__ uint32_t wave_tot[16 * (kSamplesPerBlock / 1024)];
  __shared__ uint32_t live_bits[kSamplesPerBlock / 32];
  __shared__ int s_nr;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int nt = 1024, nit = kSamplesPerBlock / nt;
  const int64_t i_blk = (int64_t)blockIdx.x * kSamplesPerBlock;  // sample i of this round = sample i_off + i of the source
  // pass 1, source order (the gradient rows are tested the way they lie in memory): which samples are live
#pragma unroll
  for (int it = 0; it < nit; ++it) {
    const int64_t i0 = i_blk + it * nt + tid;
    const bool live = !src.silent(i_off + i0 - lane, n_total) && i0 < n;
    const unsigned long long m = __ballot(live);
    if (lane == 0) {
      live_bits[(it * nt + tid) >> 5] = (uint32_t)m;
      live_bits[((it * nt + tid) >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
  if (tid == 0) s_nr = allow_transpose ? src.coherent_rays(i_off + i_blk, n - i_blk) : 0;
  __syncthreads();
  const int nr = s_nr;                                  // rays of a sample-index-major chunk, 0: ray-major
  const int spr = nr ? kSamplesPerBlock / nr : 1;       // samples per ray
  // pass 2, walk order: positions of the live samples, compacted in that order.  All four passes' wave counts go to LDS
  // first and ONE barrier orders them (a barrier pair per pass was eight workgroup syncs for 4096 samples).
  bool live_it[nit];
  int local_it[nit];
  unsigned long long m_it[nit];
#pragma unroll
  for (int it = 0; it < nit; ++it) {
    const int p = it * nt + tid;
    local_it[it] = nr ? (p % nr) * spr + p / nr : p;  // slot p of the walk holds sample `local` of the chunk
    live_it[it] = (live_bits[local_it[it] >> 5] >> (local_it[it] & 31)) & 1u;
    m_it[it] = __ballot(live_it[it]);
    if (lane == 0) wave_tot[it * 16 + wave] = (uint32_t)__popcll(m_it[it]);
  }
  __syncthreads();
  uint32_t base = 0;
#pragma unroll
  for (int it = 0; it < nit; ++it) {
    uint32_t before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
      const uint32_t c = wave_tot[it * 16 + w];
      before += w < wave ? c : 0u;
      total += c;
    }
    if (live_it[it]) {
      const uint32_t slot = base + before + (uint32_t)__popcll(m_it[it] & ((1ull << lane) - 1ull));
      gpos[i_blk + slot] = src.position(i_off + i_blk + local_it[it]);
      gidx[i_blk + slot] = (uint16_t)local_it[it];
    }
    base += total;
  }
  if (tid == 0) nlive[blockIdx.x] = base;
}

// ---- count ---------------------------------------------------------------------------------------------------
template <bool PAIR, bool MULTI>
__global__ __launch_bounds__(1024) void bin_count_kernel(GridDev g, int log2TS, int nb, uint32_t* __restrict__ counts,
                                                          const float4* __restrict__ gpos,
                                                          const uint32_t* __restrict__ nlive) {
  extern __shared__ __attribute__((aligned(16))) float4 pos[];  // x, y, z, std (MULTI: slot) of the block's samples
  uint32_t* hist = reinterpret_cast<uint32_t*>(pos + kSamplesPerBlock);
  const int tid = threadIdx.x;
  constexpr int nt = 1024, nit = kSamplesPerBlock / nt;
  const int64_t i_blk = (int64_t)blockIdx.x * kSamplesPerBlock;
  const int nl = (int)nlive[blockIdx.x], nit_live = (nl + nt - 1) / nt;  // block-uniform
  for (int it = 0; it < nit_live; ++it)
    if (it * nt + tid < nl) pos[it * nt + tid] = gpos[i_blk + it * nt + tid];
  const uint32_t mask = (1u << g.log2T) - 1u;
  for (int l = blockIdx.y; l < g.L; l += gridDim.y) {
    const float sc = g.scal[l];
    __syncthreads();  // pos[] written / previous level's histogram stored
    for (int b = tid; b < nb; b += nt) hist[b] = 0;
    __syncthreads();
    for (int it = 0; it < nit_live; ++it) {
      const bool live = it * nt + tid < nl;
      if (__ballot(live) == 0ull) continue;  // wave-uniform
      const float4 p = pos[live ? it * nt + tid : 0];
      const Corners c = hash_corners(p.x, p.y, p.z, sc, mask);
      const uint32_t sb = MULTI ? __float_as_uint(p.w) << g.log2T : 0u;  // multi-grid: entry = slot * T + hash
#pragma unroll
      for (int k = 0; k < (PAIR ? 4 : 8); ++k) {
        // a silent sample never joins a run of equal entries; the first lane of a 16-lane row always heads a run
        const uint32_t kf = live ? (c.idx[PAIR ? kPairF[k] : k] | sb) : 0xffffffffu;
        const uint32_t kc = PAIR ? (live ? (c.idx[kPairC[k]] | sb) : 0xffffffffu) : kf;
        const bool head = dpp_row_shr<1>(kf, ~kf) != kf || (PAIR && dpp_row_shr<1>(kc, ~kc) != kc);
        if (live && head) {
          atomicAdd(&hist[kf >> log2TS], 1u);
          if (PAIR && ((kf ^ kc) >> log2TS)) atomicAdd(&hist[kc >> log2TS], 1u);  // the pair straddles two slices: two records
        }
      }
    }
    __syncthreads();
    uint32_t* row = counts + ((size_t)blockIdx.x * g.L + l) * nb;
    for (int b = tid; b < nb; b += nt) row[b] = hist[b];
  }
}

// ---- scan ----------------------------------------------------------------------------------------------------
// counts[chunk][col] -> exclusive prefix over the chunks OF ONE SEGMENT (in place), segtot[seg][col] = the segment's sum.
// The loads of a column do not depend on the running sum, so they are issued 8 at a time.  (One thread per column over
// ALL chunks was the round-2 form: with 2^23-sample rounds that is a serial walk over up to 2048 rows by a few hundred
// threads; segments give the walk chunks / 64 times the parallelism.)
__global__ __launch_bounds__(256) void bin_scan_chunks_kernel(uint32_t* __restrict__ counts, int chunks, int cols,
                                                               uint32_t* __restrict__ segtot) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const int k0 = blockIdx.y * kSegChunks;
  const int k1 = k0 + kSegChunks < chunks ? k0 + kSegChunks : chunks;
  uint32_t run = 0;
  int k = k0;
  for (; k + 8 <= k1; k += 8) {
    uint32_t v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = counts[(size_t)(k + u) * cols + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      counts[(size_t)(k + u) * cols + c] = run;
      run += v[u];
    }
  }
  for (; k < k1; ++k) {
    const uint32_t v = counts[(size_t)k * cols + c];
    counts[(size_t)k * cols + c] = run;
    run += v;
  }
  segtot[(size_t)blockIdx.y * cols + c] = run;
}

// segtot[seg][col] -> exclusive prefix over the segments (in place), totals[col] = column sum
__global__ __launch_bounds__(256) void bin_scan_segments_kernel(uint32_t* __restrict__ segtot, int nseg, int cols,
                                                                 uint32_t* __restrict__ totals) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  uint32_t run = 0;
  for (int sg = 0; sg < nseg; ++sg) {
    const uint32_t v = segtot[(size_t)sg * cols + c];
    segtot[(size_t)sg * cols + c] = run;
    run += v;
  }
  totals[c] = run;
}

// offsets[0..cols] = exclusive prefix of totals (one workgroup; cols <= 32 * 2048)
__global__ __launch_bounds__(1024) void bin_scan_totals_kernel(const uint32_t* __restrict__ totals, int cols,
                                                                uint32_t* __restrict__ offsets) {
  __shared__ uint32_t part[1024];
  const int per = (cols + 1023) / 1024, c0 = threadIdx.x * per;
  uint32_t s = 0;
  for (int k = 0; k < per; ++k)
    if (c0 + k < cols) s += totals[c0 + k];
  part[threadIdx.x] = s;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele over the 1024 partial sums
    const uint32_t t = threadIdx.x >= off ? part[threadIdx.x - off] : 0u;
    __syncthreads();
    part[threadIdx.x] += t;
    __syncthreads();
  }
  uint32_t run = part[threadIdx.x] - s;  // exclusive
  for (int k = 0; k < per; ++k)
    if (c0 + k < cols) {
      offsets[c0 + k] = run;
      run += totals[c0 + k];
    }
  if (threadIdx.x == 1023) offsets[cols] = part[1023];
}

// ---- emit ----------------------------------------------------------------------------------------------------
template <int F, bool PAIR, class Src>
__global__ __launch_bounds__(1024) void bin_emit_kernel(GridDev g, Src src, int log2TS, int nb, int64_t i_off, int64_t n,
                                                         const uint32_t* __restrict__ bases,
                                                         const uint32_t* __restrict__ segbase,
                                                         const uint32_t* __restrict__ offsets,
                                                         const float4* __restrict__ gpos,
                                                         const uint16_t* __restrict__ gidx,
                                                         const uint32_t* __restrict__ nlive, float* __restrict__ qrec,
                                                         float* __restrict__ qmax, int nmax) {
  extern __shared__ __attribute__((aligned(16))) float4 pos[];
  uint32_t* rank = reinterpret_cast<uint32_t*>(pos + kSamplesPerBlock);
  uint32_t* base = rank + nb;
  const int tid = threadIdx.x, lane = tid & 63;
  constexpr int nt = 1024, nit = kSamplesPerBlock / nt;
  const int64_t i_blk = (int64_t)blockIdx.x * kSamplesPerBlock;
  const int nl = (int)nlive[blockIdx.x], nit_live = (nl + nt - 1) / nt;  // block-uniform
  int64_t src_i[nit];  // the source sample behind each of this thread's live slots
  float pre[nit];      // its level-independent gradient factor
#pragma unroll
  for (int it = 0; it < nit; ++it) {
    const bool live = it * nt + tid < nl;
    if (live) pos[it * nt + tid] = gpos[i_blk + it * nt + tid];
    src_i[it] = i_off + i_blk + (live ? (int64_t)gidx[i_blk + it * nt + tid] : 0);
    pre[it] = live ? src.pre(src_i[it]) : 0.f;
  }
  const uint32_t* seg = segbase ? segbase + (size_t)(blockIdx.x / kSegChunks) * g.L * nb : nullptr;
  const uint32_t mask = (1u << g.log2T) - 1u;
  const uint32_t tsmask = (1u << log2TS) - 1u;
  for (int l = blockIdx.y; l < g.L; l += gridDim.y) {
    const float sc = g.scal[l];
    __syncthreads();  // pos[] loaded / previous level's ranks consumed
    for (int b = tid; b < nb; b += nt) {
      rank[b] = 0;
      base[b] = offsets[l * nb + b] + bases[((size_t)blockIdx.x * g.L + l) * nb + b] + (seg ? seg[l * nb + b] : 0u);
    }
    __syncthreads();
    float vmax = 0.f;
#pragma unroll
    for (int it = 0; it < nit; ++it) {
      if (it >= nit_live) break;  // block-uniform
      const bool live = it * nt + tid < nl;
      if (__ballot(live) == 0ull) continue;
      const float4 p = pos[live ? it * nt + tid : 0];
      const Corners c = hash_corners(p.x, p.y, p.z, sc, mask);
      const uint32_t sb = Src::kMulti ? __float_as_uint(p.w) << g.log2T : 0u;
      float w[8];
      corner_weights(c, w);
      float gv[F];
      src.template grad<F>(live ? src_i[it] : i_off, l, sc, p.w, pre[it], gv);
      if (!live) {
#pragma unroll
        for (int k = 0; k < F; ++k) gv[k] = 0.f;
      }
      constexpr int NV = PAIR ? 2 * F : F, RW = NV + 1;  // values per record, record length in dwords
#pragma unroll
      for (int k = 0; k < (PAIR ? 4 : 8); ++k) {
        const uint32_t kf = live ? (c.idx[PAIR ? kPairF[k] : k] | sb) : 0xffffffffu;
        const uint32_t kc = PAIR ? (live ? (c.idx[kPairC[k]] | sb) : 0xffffffffu) : kf;
        const bool head = dpp_row_shr<1>(kf, ~kf) != kf || (PAIR && dpp_row_shr<1>(kc, ~kc) != kc);
        const unsigned long long hm = __ballot(head);
        float v[NV];  // (floor-corner terms, then ceil-corner terms)
#pragma unroll
        for (int j = 0; j < F; ++j) {
          v[j] = w[PAIR ? kPairF[k] : k] * gv[j];
          if constexpr (PAIR) v[F + j] = w[kPairC[k]] * gv[j];
        }
        if (hm != ~0ull) {
          // some run is longer than 1: segmented suffix sum onto the run heads, inside each 16-lane row, on DPP
          // row shifts (VALU rate; a 64-lane __shfl version goes through the LDS crossbar 18x per corner)
          const uint32_t run = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull));
#define NR_SEG_STEP(OFF)                                                   \
  {                                                                        \
    const bool same = dpp_row_shl<OFF>(run, 0xffffffffu) == run;           \
    _Pragma("unroll") for (int j = 0; j < NV; ++j) {                       \
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
        if (head && live) {
          const uint32_t b = kf >> log2TS;
          // (one LDS atomic per record.  A wave-aggregated add -- one atomic for all lanes that share the first lane's slice --
          //  was measured SLOWER: emit<1> 293 -> 341 us, emit<4> 693 -> 752 us; the hashed levels scatter a wave's records
          //  over the slices, so the two ballots and the readlanes buy nothing there)
          const uint32_t my_rank = atomicAdd(&rank[b], 1u);
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            const float av = fabsf(v[j]);
            if (av <= 3.402823466e38f) vmax = fmaxf(vmax, av);  // Inf/NaN do not set the scale; they poison in `reduce`
          }
          const uint32_t xm = kf ^ kc;  // (0 without pairs)
          float out[RW];
          if ((xm >> log2TS) == 0) {  // (always, unless the level's resolution reaches the slice length)
            out[0] = __uint_as_float((kf & tsmask) | (xm << 16));
#pragma unroll
            for (int j = 0; j < NV; ++j) out[1 + j] = v[j];
            store_record<RW>(qrec + (size_t)(base[b] + my_rank) * RW, out);
          } else if constexpr (PAIR) {  // the ceil corner lives in another slice: two records, each with a zero second half
            out[0] = __uint_as_float(kf & tsmask);
#pragma unroll
            for (int j = 0; j < F; ++j) out[1 + j] = v[j], out[1 + F + j] = 0.f;
            store_record<RW>(qrec + (size_t)(base[b] + my_rank) * RW, out);
            const uint32_t b2 = kc >> log2TS;
            out[0] = __uint_as_float(kc & tsmask);
#pragma unroll
            for (int j = 0; j < F; ++j) out[1 + j] = v[F + j];
            store_record<RW>(qrec + (size_t)(base[b2] + atomicAdd(&rank[b2], 1u)) * RW, out);
          }
        }
      }
    }
    // level maximum of |value|: one slot per wave, read back by `reduce`
#pragma unroll
    for (int off = 32; off; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) qmax[(size_t)l * nmax + (size_t)blockIdx.x * (nt >> 6) + (tid >> 6)] = vmax;
  }
}

// ---- reduce --------------------------------------------------------------------------------------------------
template <int F, bool PAIR>
__global__ __launch_bounds__(1024) void bin_reduce_kernel(const uint32_t* __restrict__ offsets,
                                                           const float* __restrict__ qrec,
                                                           const float* __restrict__ qmax_all, float* __restrict__ gt,
                                                           int log2T, int log2TS, int nb, int nmax, int overwrite,
                                                           int out_half, int nbg) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];
  __shared__ float smax[16];
  __shared__ uint32_t poisoned;
  const int lb = blockIdx.x;
  const uint32_t first = offsets[lb], cnt = offsets[lb + 1] - first;
  const int l = lb / nb, b = lb - l * nb;
  // where the slice's entries start in grad_table, in entries.  One grid: [L][T].  Multi-grid (nbg = slices per table and
  // level > 0): the block is grid-major, [slot][L][T] -- every slot's gradient a contiguous [L * T, F] tensor.
  size_t entry0 = ((size_t)l << log2T) + ((size_t)b << log2TS);
  if (nbg > 0) {
    const int slot = b / nbg, L = (int)gridDim.x / nb;
    entry0 = (((size_t)slot * L + l) << log2T) + ((size_t)(b - slot * nbg) << log2TS);
  }
  if (cnt == 0) {  // uniform: nothing was sent to this slice
    if (overwrite && out_half) {  // fp16 gradient (fp16-storage table): half the bytes, written as 8-byte groups
      __half* z = reinterpret_cast<__half*>(gt) + entry0 * F;
      const int nz = F << log2TS;
      for (int i = threadIdx.x; i < nz; i += 1024) z[i] = __float2half(0.f);
    } else if (overwrite) {  // the caller did not zero grad_table: this slice's part of it is ours to define
      float* z = gt + entry0 * F;
      const int nz = F << log2TS;
      if (nz % 4 == 0)
        for (int i = threadIdx.x * 4; i < nz; i += 4096) *reinterpret_cast<float4*>(z + i) = make_float4(0.f, 0.f, 0.f, 0.f);
      else
        for (int i = threadIdx.x; i < nz; i += 1024) z[i] = 0.f;
    }
    return;
  }
  // |value| < 2^(e+1) for every finite record of the level; cnt < 2^hb records -> |value * 2^sh| < 2^(61-hb), sum < 2^61
  const float* qmax = qmax_all + (size_t)l * nmax;
  float vmax = 0.f;
  for (int i = threadIdx.x; i < nmax; i += 1024) vmax = fmaxf(vmax, qmax[i]);
#pragma unroll
  for (int off = 32; off; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = vmax;
  if (threadIdx.x == 0) poisoned = 0;
  const int nacc = F << log2TS;
  uint32_t* pbits = reinterpret_cast<uint32_t*>(tile + nacc);  // one poison bit per accumulator
  for (int i = threadIdx.x; i < nacc; i += 1024) tile[i] = 0ull;
  for (int i = threadIdx.x; i < (nacc + 31) / 32; i += 1024) pbits[i] = 0u;
  __syncthreads();
  vmax = smax[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) vmax = fmaxf(vmax, smax[i]);
  const int e = (int)((__float_as_uint(vmax) >> 23) & 0xff) - 127;
  const int hb = (PAIR ? 33 : 32) - __clz(cnt);  // a pair record adds up to two terms to one accumulator (xm = 0)
  const int sh = 61 - hb - (e + 1);
  // record: {entry-in-slice | xm << 16, F floor-corner values, F ceil-corner values}; without pairs {entry, F values}
  constexpr int NV = PAIR ? 2 * F : F, RW = NV + 1;
  const float* rec = qrec + (size_t)first * RW;
  // 4 records per thread in flight: the loads are independent, only the LDS adds follow them
  for (uint32_t e0 = 0; e0 < cnt; e0 += 4096) {
    float q[4][RW];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = e0 + u * 1024 + threadIdx.x;
      load_record<RW>(rec + (size_t)(i < cnt ? i : cnt - 1) * RW, q[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u * 1024 + threadIdx.x < cnt) {
        const uint32_t w0 = __float_as_uint(q[u][0]);
        const uint32_t key = w0 & 0xffffu, key2 = key ^ (w0 >> 16);
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const float v = q[u][1 + j];
          const uint32_t at = (j < F ? key : key2) * F + (j < F ? j : j - F);
          if (fabsf(v) <= 3.402823466e38f) {
            atomicAdd(tile + at, (unsigned long long)__float2ll_rn(ldexpf(v, sh)));
          } else {
            atomicOr(&pbits[at >> 5], 1u << (at & 31));
            poisoned = 1;
          }
        }
      }
    }
  }
  __syncthreads();
  float* out = gt + entry0 * F;
  const bool any_poison = poisoned != 0;
  const float nan = __uint_as_float(0x7fc00000u);
  if (out_half) {
    // the gradient of an fp16-storage table in the table's own type (overwrite mode only: the launcher refuses the rest):
    // autograd wants it in the parameter's dtype, and the 537 MB fp32 image + its cast pass (0.2 ms on config[4]) never exist
    __half* oh = reinterpret_cast<__half*>(gt) + entry0 * F;
    for (int i = threadIdx.x; i < nacc; i += 1024) {
      float o = (float)ldexp((double)(long long)tile[i], -sh);
      if (any_poison && ((pbits[i >> 5] >> (i & 31)) & 1)) o = nan;
      oh[i] = __float2half(o);
    }
    return;
  }
  if (nacc % 4 == 0) {
    for (int i = threadIdx.x * 4; i < nacc; i += 4096) {
      float4 o = overwrite ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(out + i);
      o.x += (float)ldexp((double)(long long)tile[i], -sh);
      o.y += (float)ldexp((double)(long long)tile[i + 1], -sh);
      o.z += (float)ldexp((double)(long long)tile[i + 2], -sh);
      o.w += (float)ldexp((double)(long long)tile[i + 3], -sh);
      if (any_poison) {
        const uint32_t pw = pbits[i >> 5] >> (i & 31);
        if (pw & 1) o.x = nan;
        if (pw & 2) o.y = nan;
        if (pw & 4) o.z = nan;
        if (pw & 8) o.w = nan;
      }
      *reinterpret_cast<float4*>(out + i) = o;
    }
  } else {
    for (int i = threadIdx.x; i < nacc; i += 1024) {
      float o = (overwrite ? 0.f : out[i]) + (float)ldexp((double)(long long)tile[i], -sh);
      if (any_poison && ((pbits[i >> 5] >> (i & 31)) & 1)) o = nan;
      out[i] = o;
    }
  }
}

#define NR_DISPATCH_F(F_, CALL)      \
  do {                               \
    const int f__ = (F_);            \
    if (f__ == 1) { CALL(1); }       \
    else if (f__ == 2) { CALL(2); }  \
    else if (f__ == 4) { CALL(4); }  \
    else { CALL(8); }                \
  } while (0)

}  // namespace
}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_encode_bwd_binned_workspace(const nrhip_grid* g, int64_t n_samples, int64_t* bytes) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(bytes && n_samples >= 0, NRHIP_ERR_INVALID_ARG, "encode_bwd_binned_workspace: bad argument");
  BinPlan p;
  *bytes = (n_samples > 0 && make_plan(to_dev(*g), n_samples, &p)) ? (int64_t)p.total_bytes : 0;
  return NRHIP_OK;
}

namespace {

// All four passes for one source of samples, in rounds of round_samples().  `what` names the entry point in errors.
template <class Src>
int run_binned(const char* what, const GridDev& gd, const Src& src, int64_t n, float* grad_table, bool overwrite,
               void* workspace, int64_t workspace_bytes, hipStream_t st, int n_slots = 0) {
  // n_slots > 0: the multi-grid form (MultiSrc): grad_table is a block of n_slots gradients, [slot][L * T][F]
  const bool multi = n_slots > 0;
  NR_REQUIRE(multi == Src::kMulti, NRHIP_ERR_INVALID_ARG, "%s: n_slots goes with the multi-grid source", what);
  BinPlan p;
  NR_REQUIRE(make_plan(gd, n, &p, multi ? n_slots : 1), NRHIP_ERR_UNSUPPORTED,
             "%s: %d table(s) of 2^%d entries x %d features need more than %d slices per level; use the atomic entry point", what,
             multi ? n_slots : 1, gd.log2T, gd.F, kMaxSlices);
  const int nbg = multi ? p.nb / n_slots : 0;  // slices per table and level
  NR_REQUIRE(workspace && workspace_bytes >= (int64_t)p.total_bytes, NRHIP_ERR_INVALID_ARG,
             "%s: workspace of %lld bytes, need %zu", what, (long long)workspace_bytes, p.total_bytes);
  NR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_table) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "%s: workspace and grad_table must be 16-byte aligned", what);
  // gd.dtype describes GRAD_TABLE here -- set by the entry point, never taken from the caller's descriptor (the partition
  // does not read the table, whose storage type the descriptor's param_dtype names): 1 = an fp16 gradient for an fp16-storage
  // table, written once -- overwrite mode and a single round only (no fp16 read-modify-write)
  const bool out_half = gd.dtype == 1;
  NR_REQUIRE(!out_half || (overwrite && n <= round_samples()), NRHIP_ERR_UNSUPPORTED,
             "%s: an fp16 grad_table needs overwrite = 1 and at most %lld samples (one round)", what, (long long)round_samples());
  char* ws = static_cast<char*>(workspace);
  uint32_t* counts = reinterpret_cast<uint32_t*>(ws + p.off_counts);
  uint32_t* segtot = reinterpret_cast<uint32_t*>(ws + p.off_seg);
  uint32_t* totals = reinterpret_cast<uint32_t*>(ws + p.off_totals);
  uint32_t* offsets = reinterpret_cast<uint32_t*>(ws + p.off_offsets);
  float* qmax = reinterpret_cast<float*>(ws + p.off_qmax);
  float4* gpos = reinterpret_cast<float4*>(ws + p.off_pos);
  uint16_t* gidx = reinterpret_cast<uint16_t*>(ws + p.off_idx);
  uint32_t* nlive = reinterpret_cast<uint32_t*>(ws + p.off_live);
  float* qrec = reinterpret_cast<float*>(ws + p.off_rec);
  const int cols = gd.L * p.nb;
  const int lds_a = kSamplesPerBlock * (int)sizeof(float4) + 2 * p.nb * (int)sizeof(uint32_t);
  constexpr int lds_a_max = kSamplesPerBlock * (int)sizeof(float4) + 2 * kMaxSlices * (int)sizeof(uint32_t);
  const int nacc = gd.F << p.log2TS;
  const size_t lds_b = (size_t)nacc * sizeof(unsigned long long) + (size_t)((nacc + 31) / 32) * sizeof(uint32_t);
  static thread_local bool count_configured = false;
  if (!count_configured) {
    (void)hipFuncSetAttribute((const void*)bin_count_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a_max);
    (void)hipFuncSetAttribute((const void*)bin_count_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a_max);
    (void)hipFuncSetAttribute((const void*)bin_count_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a_max);
    (void)hipFuncSetAttribute((const void*)bin_count_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_a_max);
    count_configured = true;
  }
  const int64_t round = round_samples();
  for (int64_t i_off = 0; i_off < n; i_off += round) {
    const int64_t cnt = n - i_off < round ? n - i_off : round;
    const int chunks = (int)((cnt + kSamplesPerBlock - 1) / kSamplesPerBlock);
    const int nseg = (chunks + kSegChunks - 1) / kSegChunks;
    const dim3 grid_a((unsigned)chunks, (unsigned)p.lgroups);
    // qmax slots of chunks this round does not have stay from an earlier round otherwise
    if (hipMemsetAsync(qmax, 0, (size_t)gd.L * p.nmax * sizeof(float), st) != hipSuccess) return check_launch(what);
    bin_prep_kernel<Src><<<chunks, 1024, 0, st>>>(src, i_off, cnt, n, gpos, gidx, nlive, transposed_walk_enabled() ? 1 : 0);
    if (p.pair)
      bin_count_kernel<true, Src::kMulti><<<grid_a, 1024, lds_a, st>>>(gd, p.log2TS, p.nb, counts, gpos, nlive);
    else
      bin_count_kernel<false, Src::kMulti><<<grid_a, 1024, lds_a, st>>>(gd, p.log2TS, p.nb, counts, gpos, nlive);
    if (int e = check_launch(what)) return e;
    // one segment: its sums ARE the column totals and there is no second level
    bin_scan_chunks_kernel<<<dim3((cols + 255) / 256, nseg), 256, 0, st>>>(counts, chunks, cols, nseg > 1 ? segtot : totals);
    if (nseg > 1) bin_scan_segments_kernel<<<(cols + 255) / 256, 256, 0, st>>>(segtot, nseg, cols, totals);
    bin_scan_totals_kernel<<<1, 1024, 0, st>>>(totals, cols, offsets);
    if (int e = check_launch(what)) return e;
    if (tuning().bin_stats) {  // diagnostic (synchronises): how many records this round sends
      uint32_t total = 0;
      (void)hipStreamSynchronize(st);
      (void)hipMemcpy(&total, offsets + cols, sizeof(total), hipMemcpyDeviceToHost);
      fprintf(stderr, "[nrhip bin stats] %s: %lld samples x %d levels, F=%d, transposed walk %d: %u records (%.3f per corner%s term)\n",
              what, (long long)cnt, gd.L, gd.F, transposed_walk_enabled() ? 1 : 0, total,
              (double)total / ((double)cnt * gd.L * (p.pair ? 4 : 8)), p.pair ? "-pair" : "");
      // per level: records, and the fullest slice against the mean (one `reduce` workgroup owns one slice)
      std::vector<uint32_t> off((size_t)cols + 1);
      (void)hipMemcpy(off.data(), offsets, off.size() * sizeof(uint32_t), hipMemcpyDeviceToHost);
      for (int l = 0; l < gd.L; ++l) {
        uint32_t mx = 0;
        const uint32_t sum = off[(size_t)(l + 1) * p.nb] - off[(size_t)l * p.nb];
        for (int b = 0; b < p.nb; ++b) mx = std::max(mx, off[(size_t)l * p.nb + b + 1] - off[(size_t)l * p.nb + b]);
        fprintf(stderr, "[nrhip bin stats]   level %d: %u records in %d slices, fullest %u (%.1f x the mean)\n", l, sum, p.nb, mx,
                sum ? (double)mx * p.nb / sum : 0.0);
      }
    }
#define CALL2(F, P)                                                                                                 \
  do {                                                                                                              \
    static thread_local bool configured = false;                                                                    \
    if (!configured) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)bin_emit_kernel<F, P, Src>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                lds_a_max);                                                                         \
      (void)hipFuncSetAttribute((const void*)bin_reduce_kernel<F, P>, hipFuncAttributeMaxDynamicSharedMemorySize,   \
                                kTileBytes + 4096);                                                                 \
      configured = true;                                                                                            \
    }                                                                                                               \
    bin_emit_kernel<F, P, Src><<<grid_a, 1024, lds_a, st>>>(gd, src, p.log2TS, p.nb, i_off, cnt, counts,             \
                                                            nseg > 1 ? segtot : nullptr, offsets, gpos, gidx,       \
                                                            nlive, qrec, qmax, p.nmax);                             \
    bin_reduce_kernel<F, P><<<cols, 1024, lds_b, st>>>(offsets, qrec, qmax, grad_table, gd.log2T, p.log2TS, p.nb,   \
                                                       p.nmax, (overwrite && i_off == 0) ? 1 : 0, out_half ? 1 : 0, \
                                                       nbg);                                                        \
  } while (0)
#define CALL(F)              \
  do {                       \
    if (p.pair) CALL2(F, true);  \
    else CALL2(F, false);    \
  } while (0)
    NR_DISPATCH_F(gd.F, CALL);
#undef CALL
#undef CALL2
    if (int e = check_launch(what)) return e;
  }
  return NRHIP_OK;
}

}  // namespace

extern "C" int nrhip_encode_bwd_binned(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                                       const float* grad_out, float* grad_table, int32_t overwrite, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
  if (int e = validate_grid(g)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(grad_out && grad_table && static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "encode_bwd_binned: bad argument");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  GridDev gd = to_dev(*g);
  gd.dtype = 0;  // fp32 grad_table, whatever the table's storage type (g->param_dtype) is
  const EncodeSrc src{to_dev(*rays), static_scale, grad_out, gd.L, gd.L * gd.F};
  return run_binned("encode_bwd_binned", gd, src, n, grad_table, overwrite != 0, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

extern "C" int nrhip_encode_bwd_binned_f16(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                                           const float* grad_out, void* grad_table_fp16, void* workspace,
                                           int64_t workspace_bytes, void* stream) {
  if (int e = validate_grid(g)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(grad_out && grad_table_fp16 && static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "encode_bwd_binned_f16: bad argument");
  const int64_t n = rays->n_rays * rays->n_samples;
  NR_REQUIRE(n > 0, NRHIP_ERR_INVALID_ARG, "encode_bwd_binned_f16: every element is written: at least one sample");
  GridDev gd = to_dev(*g);
  gd.dtype = 1;  // fp16 grad_table
  const EncodeSrc src{to_dev(*rays), static_scale, grad_out, gd.L, gd.L * gd.F};
  return run_binned("encode_bwd_binned_f16", gd, src, n, static_cast<float*>(grad_table_fp16), true, workspace,
                    workspace_bytes, (hipStream_t)stream);
}

extern "C" int nrhip_hashgrid_bwd_binned(const nrhip_grid* g, const float* x, const float* grad_out, int64_t n,
                                         float* grad_table, int32_t overwrite, void* workspace,
                                         int64_t workspace_bytes, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(x && grad_out && grad_table && n >= 0, NRHIP_ERR_INVALID_ARG, "hashgrid_bwd_binned: bad argument");
  if (n == 0) return NRHIP_OK;
  GridDev gd = to_dev(*g);
  gd.dtype = 0;  // fp32 grad_table
  const GridSrc src{x, grad_out, gd.L, gd.L * gd.F};
  return run_binned("hashgrid_bwd_binned", gd, src, n, grad_table, overwrite != 0, workspace, workspace_bytes,
                    (hipStream_t)stream);
}

extern "C" int nrhip_hashgrid_multi_bwd_binned_workspace(const nrhip_grid* g, int32_t n_slots, int64_t n, int64_t* bytes) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(bytes && n >= 0 && n_slots >= 1, NRHIP_ERR_INVALID_ARG, "hashgrid_multi_bwd_binned_workspace: bad argument");
  BinPlan p;
  *bytes = (n > 0 && make_plan(to_dev(*g), n, &p, n_slots)) ? (int64_t)p.total_bytes : 0;
  return NRHIP_OK;
}

extern "C" int nrhip_hashgrid_multi_bwd_binned(const nrhip_grid* g, int32_t n_grids, const int32_t* grid_id,
                                               const int32_t* slot_of, int32_t n_slots, const float* x, const float* grad_out,
                                               int64_t n, void* grad_block, int32_t block_dtype, void* workspace,
                                               int64_t workspace_bytes, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(grid_id && slot_of && x && grad_out && grad_block && n > 0 && n_grids >= 1 && n_slots >= 1 && n_slots <= n_grids &&
                 (block_dtype == 0 || block_dtype == 1),
             NRHIP_ERR_INVALID_ARG, "hashgrid_multi_bwd_binned: bad argument (every element of the block is written: n > 0)");
  GridDev gd = to_dev(*g);
  NR_REQUIRE(((int64_t)n_slots << gd.log2T) < ((int64_t)1 << 32) - 1, NRHIP_ERR_UNSUPPORTED,
             "hashgrid_multi_bwd_binned: %d tables of 2^%d entries exceed 32-bit entry numbers", n_slots, gd.log2T);
  gd.dtype = block_dtype;  // the block's type: 1 = fp16 gradients for fp16-storage tables (one round only)
  const MultiSrc src{x, grad_out, grid_id, slot_of, n_grids, gd.L, gd.L * gd.F};
  return run_binned("hashgrid_multi_bwd_binned", gd, src, n, static_cast<float*>(grad_block), true, workspace, workspace_bytes,
                    (hipStream_t)stream, n_slots);
}

// grad_table part of nrhip_proposal_density_bwd (the decoder gradient stays with that entry point's kernel)
namespace nrhip {
int proposal_table_grad_binned(const nrhip_proposal* p, const nrhip_rays* rays, const float* density,
                               const float* grad_density, float* grad_table, bool overwrite, void* workspace,
                               int64_t workspace_bytes, void* stream) {
  const int64_t n = rays->n_rays * rays->n_samples;
  GridDev gd = to_dev(p->grid);
  gd.dtype = 0;  // (here the descriptor's param_dtype is the TABLE's storage type; this entry point's grad_table is fp32)
  const ProposalSrc src{to_dev(*rays), p->static_scale, p->decoder_weight, density, grad_density};
  return run_binned("proposal_density_bwd_binned", gd, src, n, grad_table, overwrite, workspace, workspace_bytes,
                    (hipStream_t)stream);
}
}  // namespace nrhip
