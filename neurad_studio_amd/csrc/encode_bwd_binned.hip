// B1 (table gradient) without device-scope atomics.
//
// On MI355X every fp32 global atomic is executed memory-side: rocprofv3 shows TCC_EA0_ATOMIC == TCC_ATOMIC for the
// scatter-add kernel (hashgrid.hip), i.e. each of its ~55 M atomic requests is one fabric transaction, and the
// fabric retires ~1.8e10 of them per second no matter where the line lives -> 3.1 ms for config 2, 44 % of a
// training step.  The per-XCD L2s are not coherent with each other, so there is no cheaper scope to fall back to.
//
// This path gives every table entry exactly one owner instead:
//   pass A (bin_scatter):  the table of a level is cut into LDS-sized slices of TS = 16384/F entries (128 KB of
//       64-bit accumulators).  A workgroup takes 1024 or 4096 consecutive samples, parks their contracted positions in LDS, and for
//       each level bins the (entry, F gradient values) records of its samples by slice: LDS histogram -> one
//       returning global atomic per non-empty slice to reserve a range of that slice's queue -> records written
//       at (range base + LDS rank).  Runs of equal entries in neighbouring lanes of a 16-lane row (consecutive
//       samples of a ray in the same coarse cell) are summed first.
//   pass B (bin_reduce):   one 1024-thread workgroup per (level, slice) streams its queue, accumulates into the
//       slice image in LDS, and adds the image to grad_table with plain 16-byte loads/stores.  The image is 64-bit
//       FIXED POINT: ds_add_f32 turned out ~10x slower than the integer LDS atomics on gfx950 (493 vs 147 us for
//       this kernel), so values are scaled by a power of two chosen from the level's largest |value| (found by
//       pass A) and the slice's record count so that the sum cannot overflow, and added with ds_add_u64.  The
//       quantum is below 2^-40 of the level's largest term -- finer than the fp32 rounding of the atomic path for
//       anything that matters -- and integer addition is associative: the result is bit-reproducible run to run.
// A queue holds 2x the mean record count of its level; records beyond that (never seen in the tests) fall back to
// the global atomic in pass A, which pass B's read-modify-write then picks up (stream order).
#include "common.h"

namespace nrhip {
namespace {

constexpr int kMaxSamplesPerBlock = 4096;  // pass A: 4 samples per thread, 256 or 1024 threads
constexpr int kMaxSlices = 2048;        // per level (LDS histogram + base table = 16 KB)
constexpr int kTileBytes = 128 * 1024;   // slice image in LDS
constexpr int64_t kChunkSamples = 1 << 20;  // samples per (pass A, pass B) round: bounds the scratch (3.2 GB on config 2)

struct BinPlan {
  int log2TS;          // entries per slice
  int nb;              // slices per level
  uint32_t cap;        // records per queue
  int spb;             // pass A: samples per workgroup (threads = spb / 4)
  int lgroups;         // pass A: the levels are dealt round-robin to this many workgroups per sample chunk
  int nmax;            // partial maxima per level (pass A waves)
  size_t counter_bytes;
  size_t total_bytes;
};

// Returns false when the grid can not be binned (too many slices per level).
bool make_plan(const GridDev& g, int64_t n_total, BinPlan* p) {
  const int64_t n = n_total < kChunkSamples ? n_total : kChunkSamples;  // larger batches go through in rounds
  int log2F = 0;
  while ((1 << log2F) < g.F) ++log2F;
  int log2TS = 14 - log2F;  // 8-byte accumulators: 16384 / F entries fill the 128 KB image
  if (log2TS > g.log2T) log2TS = g.log2T;
  // small tables: shrink the slices until pass B has ~2 workgroups per CU (one workgroup owns one slice)
  while (((int64_t)g.L << (g.log2T - log2TS)) < 512 && log2TS > 9) --log2TS;
  const int64_t nb = (int64_t)1 << (g.log2T - log2TS);
  if (nb > kMaxSlices) return false;
  const int64_t mean = (n * 8 + nb - 1) / nb;
  const int64_t cap = 2 * mean + 256;
  if (cap > 0x7fffffff) return false;
  p->log2TS = log2TS;
  p->nb = (int)nb;
  p->cap = (uint32_t)cap;
  // few slices: small chunks already give long contiguous runs per (workgroup, slice); many slices: bigger chunks
  // so that a reservation atomic still buys >= ~64 records.  Levels are split over workgroups until the grid has
  // ~4 workgroups per CU.
  p->spb = nb <= 128 ? 1024 : kMaxSamplesPerBlock;
  const int64_t chunks = (n + p->spb - 1) / p->spb;
  int lg = (int)((1024 + chunks - 1) / chunks);
  p->lgroups = lg < 1 ? 1 : (lg > g.L ? g.L : lg);
  // header: [L * nb] queue fill counters, then [L][chunks * waves] partial maxima of |value| (plain stores, one
  // slot per wave of pass A: a contended atomicMax on L words costs more than reading the slots back)
  p->nmax = (int)(chunks * (p->spb / 4 / 64));
  p->counter_bytes = (((size_t)g.L * (nb + p->nmax) * sizeof(uint32_t)) + 255) & ~(size_t)255;
  p->total_bytes = p->counter_bytes + (size_t)g.L * nb * cap * (g.F + 1) * sizeof(float);
  return true;
}

// Where the samples and their feature gradients come from (pass A is otherwise identical):
//   position(i) -> (x, y, z in [0,1]^3, aux);  grad(i, l, level scale, aux, gv[F]) -> dL/d(level-l features of sample i)
struct EncodeSrc {  // H2+H3+H1+H4 (nrhip_encode_bwd): positions from ray samples, gradient of the rescaled features
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
  template <int F>
  __device__ void grad(int64_t i, int l, float sc, float std, float (&gv)[F]) const {
    const float rw = rescale_weight(sc, std);
#pragma unroll
    for (int k = 0; k < F; ++k) gv[k] = go[(i * L + l) * F + k] * rw;
  }
};
struct GridSrc {  // H1 (nrhip_hashgrid_bwd): positions given
  const float* x;
  const float* go;
  int L;
  __device__ float4 position(int64_t i) const { return make_float4(x[3 * i], x[3 * i + 1], x[3 * i + 2], 0.f); }
  template <int F>
  __device__ void grad(int64_t i, int l, float, float, float (&gv)[F]) const {
#pragma unroll
    for (int k = 0; k < F; ++k) gv[k] = go[(i * L + l) * F + k];
  }
};
struct ProposalSrc {  // S2 (nrhip_proposal_density_bwd): density = trunc_exp(decoder . rescaled features), F = 1
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
  template <int F>
  __device__ void grad(int64_t i, int l, float sc, float std, float (&gv)[F]) const {
    const float xlog = logf(dens[i]);  // activations.py:37-41: g * exp(clamp(x, -15, 15))
    const float gx = gd[i] * expf(fminf(fmaxf(xlog, -15.f), 15.f));
    gv[0] = gx * dec[l] * rescale_weight(sc, std);
  }
};

template <int F, class Src>
__global__ __launch_bounds__(1024) void bin_scatter_kernel(GridDev g, Src src, float* __restrict__ gt,
                                                            uint32_t* __restrict__ qcount, float* __restrict__ qrec,
                                                            int log2TS, int nb, uint32_t cap, int spb,
                                                            int64_t i_off, int64_t n, int nmax) {
  extern __shared__ __attribute__((aligned(16))) float4 pos[];  // x, y, z, std of the block's samples
  uint32_t* hist = reinterpret_cast<uint32_t*>(pos + spb);
  uint32_t* base = hist + nb;
  const int tid = threadIdx.x, lane = tid & 63, nt = blockDim.x;
  const int64_t i_blk = (int64_t)blockIdx.x * spb;  // sample i of this round is sample i_off + i of the source
  const int nit = (int)(((n - i_blk < spb ? n - i_blk : spb) + nt - 1) / nt);
  for (int it = 0; it < nit; ++it) {
    const int64_t i0 = i_blk + it * nt + tid;
    pos[it * nt + tid] = src.position(i_off + (i0 < n ? i0 : n - 1));
  }
  const uint32_t mask = (1u << g.log2T) - 1u;
  const uint32_t tsmask = (1u << log2TS) - 1u;
  for (int l = blockIdx.y; l < g.L; l += gridDim.y) {
    const float sc = g.scal[l];
    __syncthreads();  // pos[] written / previous level's ranks consumed
    for (int b = tid; b < nb; b += nt) hist[b] = 0;
    __syncthreads();
    // ---- count: how many records does this block send to each slice -------------------------------
    for (int it = 0; it < nit; ++it) {
      const bool live = i_blk + it * nt + tid < n;
      const float4 p = pos[it * nt + tid];
      const Corners c = hash_corners(p.x, p.y, p.z, sc, mask);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t key = c.idx[k];
        const uint32_t prev = dpp_row_shr<1>(key, ~key);  // first lane of a 16-lane row always heads a run
        if (live && prev != key) atomicAdd(&hist[key >> log2TS], 1u);
      }
    }
    __syncthreads();
    // ---- reserve a range of every non-empty slice queue -------------------------------------------
    for (int b = tid; b < nb; b += nt) {
      const uint32_t cnt = hist[b];
      base[b] = cnt ? atomicAdd(&qcount[l * nb + b], cnt) : 0u;
      hist[b] = 0;
    }
    __syncthreads();
    // ---- emit ---------------------------------------------------------------------------------------
    float* gl = gt + ((size_t)l << g.log2T) * F;
    float vmax = 0.f;
    for (int it = 0; it < nit; ++it) {
      const int64_t i0 = i_blk + it * nt + tid;
      const bool live = i0 < n;
      const int64_t i = live ? i0 : n - 1;
      const float4 p = pos[it * nt + tid];
      const Corners c = hash_corners(p.x, p.y, p.z, sc, mask);
      float w[8];
      corner_weights(c, w);
      float gv[F];
      src.template grad<F>(i_off + i, l, sc, p.w, gv);
      if (!live) {
#pragma unroll
        for (int k = 0; k < F; ++k) gv[k] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint32_t key = c.idx[k];
        const uint32_t prev = dpp_row_shr<1>(key, ~key);
        const bool head = prev != key;
        const unsigned long long hm = __ballot(head);
        float v[F];
#pragma unroll
        for (int j = 0; j < F; ++j) v[j] = w[k] * gv[j];
        if (hm != ~0ull) {
          // some run is longer than 1: segmented suffix sum onto the run heads, inside each 16-lane row, on DPP
          // row shifts (VALU rate; the 64-lane __shfl version went through the LDS crossbar 18x per corner)
          const uint32_t run = (uint32_t)__popcll(hm & ((2ull << lane) - 1ull));
#define NR_SEG_STEP(OFF)                                                   \
  {                                                                        \
    const bool same = dpp_row_shl<OFF>(run, 0xffffffffu) == run;           \
    _Pragma("unroll") for (int j = 0; j < F; ++j) {                        \
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
          bool finite = true;
#pragma unroll
          for (int j = 0; j < F; ++j) finite = finite && fabsf(v[j]) <= 3.402823466e38f;
          if (!finite) {  // Inf/NaN can not go through the fixed-point image: hand them to the atomic directly
#pragma unroll
            for (int j = 0; j < F; ++j) unsafeAtomicAdd(gl + (size_t)key * F + j, v[j]), v[j] = 0.f;
          }
          const uint32_t b = key >> log2TS;
          const uint32_t at = base[b] + atomicAdd(&hist[b], 1u);
          if (at < cap) {
            float* rec = qrec + ((size_t)(l * nb + b) * cap + at) * (F + 1);
            rec[0] = __uint_as_float(key & tsmask);
#pragma unroll
            for (int j = 0; j < F; ++j) rec[1 + j] = v[j], vmax = fmaxf(vmax, fabsf(v[j]));
          } else {  // queue full: fall back to the memory-side atomic (pass B adds on top of it)
#pragma unroll
            for (int j = 0; j < F; ++j) unsafeAtomicAdd(gl + (size_t)key * F + j, v[j]);
          }
        }
      }
    }
    // level maximum of |value| (non-negative floats order like their bit patterns); NaN/Inf propagate as "huge"
#pragma unroll
    for (int off = 32; off; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if (lane == 0) {
      const int nwave = nt >> 6;
      float* qmax = reinterpret_cast<float*>(qcount + (size_t)g.L * nb);
      qmax[(size_t)l * nmax + (size_t)blockIdx.x * nwave + (tid >> 6)] = vmax;
    }
  }
}

template <int F>
__global__ __launch_bounds__(1024) void bin_reduce_kernel(const uint32_t* __restrict__ qcount,
                                                           const float* __restrict__ qrec, float* __restrict__ gt,
                                                           int L, int log2T, int log2TS, int nb, uint32_t cap,
                                                           int nmax) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long tile[];
  __shared__ float smax[16];
  const int lb = blockIdx.x;
  const uint32_t filled = qcount[lb];
  if (filled == 0) return;  // uniform: nothing was sent to this slice
  const uint32_t cnt = filled < cap ? filled : cap;
  const int l = lb / nb, b = lb - l * nb;
  // |value| < 2^(e+1) for every record of the level; cnt < 2^hb records -> |value * 2^sh| < 2^(61-hb), sum < 2^61
  const float* qmax = reinterpret_cast<const float*>(qcount + (size_t)L * nb) + (size_t)l * nmax;
  float vmax = 0.f;
  for (int i = threadIdx.x; i < nmax; i += 1024) vmax = fmaxf(vmax, qmax[i]);
#pragma unroll
  for (int off = 32; off; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
  if ((threadIdx.x & 63) == 0) smax[threadIdx.x >> 6] = vmax;
  __syncthreads();
  vmax = smax[0];
#pragma unroll
  for (int i = 1; i < 16; ++i) vmax = fmaxf(vmax, smax[i]);
  const int e = (int)((__float_as_uint(vmax) >> 23) & 0xff) - 127;
  const int hb = 32 - __clz(cnt);
  const int sh = 61 - hb - (e + 1);
  const int nacc = F << log2TS;
  for (int i = threadIdx.x; i < nacc; i += 1024) tile[i] = 0ull;
  __syncthreads();
  const float* rec = qrec + (size_t)lb * cap * (F + 1);
  // 4 records per thread in flight: the loads are independent, only the LDS adds follow them
  for (uint32_t e0 = 0; e0 < cnt; e0 += 4096) {
    float q[4][F + 1];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t i = e0 + u * 1024 + threadIdx.x;
      const float* src = rec + (size_t)(i < cnt ? i : cnt - 1) * (F + 1);
#pragma unroll
      for (int j = 0; j <= F; ++j) q[u][j] = src[j];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (e0 + u * 1024 + threadIdx.x < cnt) {
        const uint32_t key = __float_as_uint(q[u][0]);
#pragma unroll
        for (int j = 0; j < F; ++j) {
          const long long fx = __float2ll_rn(ldexpf(q[u][1 + j], sh));
          atomicAdd(tile + key * F + j, (unsigned long long)fx);
        }
      }
    }
  }
  __syncthreads();
  float* out = gt + (((size_t)l << log2T) + ((size_t)b << log2TS)) * F;
  if (nacc % 4 == 0) {
    for (int i = threadIdx.x * 4; i < nacc; i += 4096) {
      float4 o = *reinterpret_cast<const float4*>(out + i);
      o.x += (float)ldexp((double)(long long)tile[i], -sh);
      o.y += (float)ldexp((double)(long long)tile[i + 1], -sh);
      o.z += (float)ldexp((double)(long long)tile[i + 2], -sh);
      o.w += (float)ldexp((double)(long long)tile[i + 3], -sh);
      *reinterpret_cast<float4*>(out + i) = o;
    }
  } else {
    for (int i = threadIdx.x; i < nacc; i += 1024) out[i] += (float)ldexp((double)(long long)tile[i], -sh);
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

// Both passes for one source of samples.  `what` names the entry point in error messages.
template <class Src>
int run_binned(const char* what, const GridDev& gd, const Src& src, int64_t n, float* grad_table, void* workspace,
               int64_t workspace_bytes, hipStream_t st) {
  BinPlan p;
  NR_REQUIRE(make_plan(gd, n, &p), NRHIP_ERR_UNSUPPORTED,
             "%s: 2^%d entries x %d features need more than %d slices per level; use the atomic entry point", what,
             gd.log2T, gd.F, kMaxSlices);
  NR_REQUIRE(workspace && workspace_bytes >= (int64_t)p.total_bytes, NRHIP_ERR_INVALID_ARG,
             "%s: workspace of %lld bytes, need %zu", what, (long long)workspace_bytes, p.total_bytes);
  NR_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 15) == 0 && (reinterpret_cast<uintptr_t>(grad_table) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "%s: workspace and grad_table must be 16-byte aligned", what);
  uint32_t* qcount = static_cast<uint32_t*>(workspace);
  float* qrec = reinterpret_cast<float*>(static_cast<char*>(workspace) + p.counter_bytes);
  const int lds_a = p.spb * (int)sizeof(float4) + 2 * p.nb * (int)sizeof(uint32_t);
  constexpr int lds_a_max = kMaxSamplesPerBlock * (int)sizeof(float4) + 2 * kMaxSlices * (int)sizeof(uint32_t);
  const size_t lds_b = (size_t)(gd.F << p.log2TS) * sizeof(unsigned long long);
  for (int64_t i_off = 0; i_off < n; i_off += kChunkSamples) {
    const int64_t cnt = n - i_off < kChunkSamples ? n - i_off : kChunkSamples;
    if (hipMemsetAsync(qcount, 0, p.counter_bytes, st) != hipSuccess) return check_launch(what);
    const dim3 grid_a((unsigned)((cnt + p.spb - 1) / p.spb), (unsigned)p.lgroups);
#define CALL(F)                                                                                                     \
  do {                                                                                                              \
    static thread_local bool configured = false;                                                                    \
    if (!configured) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)bin_scatter_kernel<F, Src>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                lds_a_max);                                                                         \
      configured = true;                                                                                            \
    }                                                                                                               \
    bin_scatter_kernel<F, Src><<<grid_a, p.spb / 4, lds_a, st>>>(gd, src, grad_table, qcount, qrec, p.log2TS, p.nb, \
                                                                 p.cap, p.spb, i_off, cnt, p.nmax);                 \
  } while (0)
    NR_DISPATCH_F(gd.F, CALL);
#undef CALL
    if (int e = check_launch(what)) return e;
#define CALL(F)                                                                                                    \
  do {                                                                                                             \
    static thread_local bool configured = false;                                                                   \
    if (!configured) {                                                                                             \
      (void)hipFuncSetAttribute((const void*)bin_reduce_kernel<F>, hipFuncAttributeMaxDynamicSharedMemorySize,     \
                                kTileBytes);                                                                       \
      configured = true;                                                                                           \
    }                                                                                                              \
    bin_reduce_kernel<F><<<gd.L * p.nb, 1024, lds_b, st>>>(qcount, qrec, grad_table, gd.L, gd.log2T, p.log2TS,      \
                                                           p.nb, p.cap, p.nmax);                                   \
  } while (0)
    NR_DISPATCH_F(gd.F, CALL);
#undef CALL
    if (int e = check_launch(what)) return e;
  }
  return NRHIP_OK;
}

}  // namespace

extern "C" int nrhip_encode_bwd_binned(const nrhip_grid* g, float static_scale, const nrhip_rays* rays,
                                       const float* grad_out, float* grad_table, void* workspace,
                                       int64_t workspace_bytes, void* stream) {
  if (int e = validate_grid(g)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(grad_out && grad_table && static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "encode_bwd_binned: bad argument");
  const int64_t n = rays->n_rays * rays->n_samples;
  if (n == 0) return NRHIP_OK;
  const GridDev gd = to_dev(*g);
  const EncodeSrc src{to_dev(*rays), static_scale, grad_out, gd.L};
  return run_binned("encode_bwd_binned", gd, src, n, grad_table, workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int nrhip_hashgrid_bwd_binned(const nrhip_grid* g, const float* x, const float* grad_out, int64_t n,
                                         float* grad_table, void* workspace, int64_t workspace_bytes, void* stream) {
  if (int e = validate_grid(g)) return e;
  NR_REQUIRE(x && grad_out && grad_table && n >= 0, NRHIP_ERR_INVALID_ARG, "hashgrid_bwd_binned: bad argument");
  if (n == 0) return NRHIP_OK;
  const GridDev gd = to_dev(*g);
  const GridSrc src{x, grad_out, gd.L};
  return run_binned("hashgrid_bwd_binned", gd, src, n, grad_table, workspace, workspace_bytes, (hipStream_t)stream);
}

// grad_table part of nrhip_proposal_density_bwd (the decoder gradient stays with that entry point's kernel)
namespace nrhip {
int proposal_table_grad_binned(const nrhip_proposal* p, const nrhip_rays* rays, const float* density,
                               const float* grad_density, float* grad_table, void* workspace, int64_t workspace_bytes,
                               void* stream) {
  const int64_t n = rays->n_rays * rays->n_samples;
  const GridDev gd = to_dev(p->grid);
  const ProposalSrc src{to_dev(*rays), p->static_scale, p->decoder_weight, density, grad_density};
  return run_binned("proposal_density_bwd_binned", gd, src, n, grad_table, workspace, workspace_bytes,
                    (hipStream_t)stream);
}
}  // namespace nrhip
