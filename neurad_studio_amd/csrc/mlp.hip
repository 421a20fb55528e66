// F2: generic tiny-MLP on the matrix cores, exact fp32 (v_mfma_f32_16x16x4_f32), any layer sizes.
//
// Transposed formulation  H_l^T = W_l . H_{l-1}^T : the WEIGHTS are the MFMA A operand (M = 16 output
// neurons), the SAMPLES are the B operand (N = 16 samples per wavefront tile), so the D tile comes out
// as [neuron][sample].  Weights are staged once per workgroup in LDS in *fragment order*
// (Wf[(mb*nstep+s)*64 + lane] = W[16mb + (lane&15)][4s + (lane>>4)]) so an A fragment is one
// conflict-free, lane-linear ds_read_b32; activations ping-pong through a small per-wave LDS tile.
// fp32 MFMA is bit-for-bit an fmaf chain, so results match the reference's fp32 Linear to rounding order.
#include <cstdlib>

#include "common.h"

namespace nrhip {


// Wave-private LDS tiles: ds ops of one wave execute in order, so only the COMPILER must be kept from
// moving a lane's reads above other lanes' writes.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

struct MlpDev {
  int in_dim, hidden, out_dim, nl;
  const float* w[NRHIP_MAX_LAYERS];
  const float* b[NRHIP_MAX_LAYERS];
};

__host__ __device__ inline int pad4(int v) { return (v + 3) & ~3; }
__host__ __device__ inline int pad16(int v) { return (v + 15) & ~15; }
__host__ __device__ inline int layer_in(const MlpDev& m, int l) { return l == 0 ? m.in_dim : m.hidden; }
__host__ __device__ inline int layer_out(const MlpDev& m, int l) { return l == m.nl - 1 ? m.out_dim : m.hidden; }

// activation row stride (floats): == 2 mod 32 -> the B-fragment read (j*ld + 4s + g) is conflict free
__host__ __device__ inline int act_ld(const MlpDev& m) {
  int k = m.in_dim > m.hidden ? m.in_dim : m.hidden;
  k = k > m.out_dim ? k : m.out_dim;
  k = pad16(k);
  return ((k + 31) / 32) * 32 + 2;
}

// TRANSPOSED=false: A = W   (rows = out neurons, k = in neurons)   -> forward
// TRANSPOSED=true : A = W^T (rows = in neurons,  k = out neurons)  -> data gradient
template <bool TRANSPOSED>
__host__ __device__ inline int frag_floats(const MlpDev& m, int l) {
  const int rows = TRANSPOSED ? layer_in(m, l) : layer_out(m, l);
  const int k = TRANSPOSED ? layer_out(m, l) : layer_in(m, l);
  return pad16(rows) * pad4(k);
}

template <bool TRANSPOSED>
__device__ void stage_weights(const MlpDev& m, float* lds_w) {
  int off = 0;
  for (int l = 0; l < m.nl; ++l) {
    const int in = layer_in(m, l), out = layer_out(m, l);
    const int rows = TRANSPOSED ? in : out, kk = TRANSPOSED ? out : in;
    const int nstep = pad4(kk) / 4, nblk = pad16(rows) / 16;
    const int total = nblk * nstep * 64;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
      const int lane = e & 63, fs = e >> 6;
      const int s = fs % nstep, mb = fs / nstep;
      const int row = 16 * mb + (lane & 15), k = 4 * s + (lane >> 4);
      float v = 0.f;
      if (row < rows && k < kk) v = TRANSPOSED ? m.w[l][(size_t)k * in + row] : m.w[l][(size_t)row * in + k];
      lds_w[off + e] = v;
    }
    off += total;
  }
}

// one layer on one 16-sample tile: act_out[j][n] = (relu)(bias[n] + Σ_k A[n][k] act_in[j][k])
// RELU_MASK != nullptr (data-gradient pass): out = value * (mask[j][n] > 0)
__device__ __forceinline__ void layer_tile(const float* __restrict__ wf, int rows, int kk, const float* bias,
                                           const float* act_in, float* act_out, int ld, bool relu, int lane) {
  const int nstep = pad4(kk) / 4, nblk = pad16(rows) / 16;
  const int j = lane & 15, g = lane >> 4;
  const float* bin = act_in + j * ld + g;
  for (int mb = 0; mb < nblk; mb += 2) {
    const bool two = mb + 1 < nblk;
    f32x4 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n0 = 16 * mb + 4 * g + r, n1 = n0 + 16;
      acc0[r] = (bias && n0 < rows) ? bias[n0] : 0.f;
      acc1[r] = (bias && two && n1 < rows) ? bias[n1] : 0.f;
    }
    const float* w0 = wf + (size_t)mb * nstep * 64 + lane;
    const float* w1 = w0 + (size_t)nstep * 64;
    if (two) {
      for (int s = 0; s < nstep; ++s) {
        const float b = bin[4 * s];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[s * 64], b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w1[s * 64], b, acc1, 0, 0, 0);
      }
    } else {
      for (int s = 0; s < nstep; ++s)
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w0[s * 64], bin[4 * s], acc0, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v0 = acc0[r], v1 = acc1[r];
      if (relu) v0 = fmaxf(v0, 0.f), v1 = fmaxf(v1, 0.f);
      act_out[j * ld + 16 * mb + 4 * g + r] = v0;
      if (two) act_out[j * ld + 16 * mb + 16 + 4 * g + r] = v1;
    }
  }
}

// copy a [16][cols] tile between global (row stride gld) and the LDS activation tile (zero padded to kpad)
__device__ __forceinline__ void load_tile(const float* __restrict__ g, int64_t row0, int64_t nrows, int cols, int gld,
                                          float* act, int ld, int kpad, int lane) {
  for (int e = lane; e < 16 * kpad; e += 64) {
    const int j = e / kpad, k = e - j * kpad;
    float v = 0.f;
    if (k < cols && row0 + j < nrows) v = g[(row0 + j) * gld + k];
    act[j * ld + k] = v;
  }
}
__device__ __forceinline__ void store_tile(float* __restrict__ g, int64_t row0, int64_t nrows, int cols, int gld,
                                           const float* act, int ld, int lane) {
  for (int e = lane; e < 16 * cols; e += 64) {
    const int j = e / cols, k = e - j * cols;
    if (row0 + j < nrows) g[(row0 + j) * gld + k] = act[j * ld + k];
  }
}

__global__ __launch_bounds__(256) void mlp_fwd_kernel(MlpDev m, const float* __restrict__ x, int64_t n,
                                                       float* __restrict__ y, float* __restrict__ hidden) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int wtot = 0;
  for (int l = 0; l < m.nl; ++l) wtot += frag_floats<false>(m, l);
  const int ld = act_ld(m);
  float* lds_w = lds;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* actA = lds + wtot + wid * 2 * 16 * ld;
  float* actB = actA + 16 * ld;
  stage_weights<false>(m, lds_w);
  __syncthreads();
  const int64_t ntiles = (n + 15) / 16;
  const int hid_ld = (m.nl - 1) * m.hidden;
  const int nw = blockDim.x >> 6;
  for (int64_t tile = (int64_t)blockIdx.x * nw + wid; tile < ntiles; tile += (int64_t)gridDim.x * nw) {
    const int64_t row0 = tile * 16;
    wave_lds_fence();
    load_tile(x, row0, n, m.in_dim, m.in_dim, actA, ld, pad4(m.in_dim), lane);
    wave_lds_fence();
    float* ain = actA;
    float* aout = actB;
    int woff = 0;
    for (int l = 0; l < m.nl; ++l) {
      const int in = layer_in(m, l), out = layer_out(m, l);
      const bool last = l == m.nl - 1;
      layer_tile(lds_w + woff, out, in, m.b[l], ain, aout, ld, !last, lane);
      wave_lds_fence();
      woff += frag_floats<false>(m, l);
      if (!last && hidden) store_tile(hidden + l * m.hidden, row0, n, m.hidden, hid_ld, aout, ld, lane);
      float* t = ain;
      ain = aout;
      aout = t;
    }
    store_tile(y, row0, n, m.out_dim, m.out_dim, ain, ld, lane);
  }
}

// data-gradient chain: dZ_last = grad_y ; for l = last..1: dH_{l-1} = W_l^T dZ_l ; dZ_{l-1} = dH_{l-1} * (h_{l-1} > 0)
// dZ of hidden layers are written to `dz` ([N,(nl-1)*hidden], same layout as `hidden`) for the weight-grad pass.
__global__ __launch_bounds__(256) void mlp_bwd_data_kernel(MlpDev m, const float* __restrict__ hidden,
                                                            const float* __restrict__ gy, int64_t n,
                                                            float* __restrict__ gx, float* __restrict__ dz) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  int wtot = 0;
  for (int l = 0; l < m.nl; ++l) wtot += frag_floats<true>(m, l);
  const int ld = act_ld(m);
  float* lds_w = lds;
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float* actA = lds + wtot + wid * 2 * 16 * ld;
  float* actB = actA + 16 * ld;
  stage_weights<true>(m, lds_w);
  __syncthreads();
  const int64_t ntiles = (n + 15) / 16;
  const int hid_ld = (m.nl - 1) * m.hidden;
  const int nw = blockDim.x >> 6;
  for (int64_t tile = (int64_t)blockIdx.x * nw + wid; tile < ntiles; tile += (int64_t)gridDim.x * nw) {
    const int64_t row0 = tile * 16;
    wave_lds_fence();
    load_tile(gy, row0, n, m.out_dim, m.out_dim, actA, ld, pad4(m.out_dim), lane);
    wave_lds_fence();
    float* ain = actA;
    float* aout = actB;
    for (int l = m.nl - 1; l >= (gx ? 0 : 1); --l) {
      const int in = layer_in(m, l), out = layer_out(m, l);
      int woff = 0;
      for (int q = 0; q < l; ++q) woff += frag_floats<true>(m, q);
      layer_tile(lds_w + woff, in, out, nullptr, ain, aout, ld, false, lane);
      wave_lds_fence();
      if (l > 0) {
        // ReLU mask of layer l-1, then publish dZ_{l-1}
        for (int e = lane; e < 16 * m.hidden; e += 64) {
          const int j = e / m.hidden, k = e - j * m.hidden;
          if (row0 + j < n) {
            const float h = hidden[(row0 + j) * hid_ld + (l - 1) * m.hidden + k];
            const float v = h > 0.f ? aout[j * ld + k] : 0.f;
            aout[j * ld + k] = v;
            dz[(row0 + j) * hid_ld + (l - 1) * m.hidden + k] = v;
          } else {
            aout[j * ld + k] = 0.f;
          }
        }
        // zero the K padding of the next step (rows of pad4(hidden) beyond hidden)
        for (int e = lane; e < 16 * (pad4(m.hidden) - m.hidden); e += 64) {
          const int pw = pad4(m.hidden) - m.hidden;
          aout[(e / pw) * ld + m.hidden + e % pw] = 0.f;
        }
        wave_lds_fence();
      } else {
        store_tile(gx, row0, n, m.in_dim, m.in_dim, aout, ld, lane);
      }
      float* t = ain;
      ain = aout;
      aout = t;
    }
  }
}

// weight gradient: dW[o][i] += Σ_n dz[n][o] * h[n][i] ; db[o] += Σ_n dz[n][o].
// MFMA with K = samples: A[o][n] = dz (lane: o = lane&15, n = lane>>4), B[n][i] = h.  One workgroup owns a
// 64x64 sub-matrix (blockIdx.y) and a slice of the samples (blockIdx.x); fp32 atomics merge the slices.
struct WgradLayer {
  const float* dz;  // [N, dz_ld]: gradient w.r.t. the layer's pre-activation
  const float* h;   // [N, h_ld]:  the layer's input
  float* dW;        // [out][in], accumulated into
  float* db;        // [out] or nullptr
  int dz_ld, h_ld, out, in, nb_in;
  int sub0;         // first blockIdx.y of this layer (64x64 sub-matrices are numbered layer by layer)
};
struct WgradArgs {
  WgradLayer layer[NRHIP_MAX_LAYERS];
  int nl;
};

// All layers of an MLP in ONE launch: a single layer's grid (<= 256 sample slices) is one wave per SIMD and runs at
// memory latency; with the layers side by side the CU overlaps them.
__global__ __launch_bounds__(256) void mlp_wgrad_kernel(WgradArgs a, int64_t n) {
  int li = 0;
#pragma unroll
  for (int l = 1; l < NRHIP_MAX_LAYERS; ++l)
    if (l < a.nl && (int)blockIdx.y >= a.layer[l].sub0) li = l;
  const WgradLayer& L = a.layer[li];
  const float* __restrict__ dzp = L.dz;
  const float* __restrict__ hp = L.h;
  float* __restrict__ dW = L.dW;
  float* __restrict__ db = L.db;
  const int dz_ld = L.dz_ld, h_ld = L.h_ld, out = L.out, in = L.in, nb_in = L.nb_in;
  const int sub = (int)blockIdx.y - L.sub0;
  const int ob = sub / nb_in, ib = sub % nb_in;  // 64-wide sub-matrix coordinates
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int i16 = lane & 15, g = lane >> 4;
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t nquads = (n + 3) / 4;
  constexpr int U = 4;  // sample quads in flight per wave: 8*U independent loads before the first MFMA needs one
  const int64_t qstep = (int64_t)gridDim.x * 4;
  for (int64_t q0 = (int64_t)blockIdx.x * 4 + wid; q0 < nquads; q0 += qstep * U) {
    float a[U][4], b[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = (q0 + u * qstep) * 4 + g;
      const bool live = row < n;  // also false for quads past the end
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int o = ob * 64 + 16 * k + i16, i = ib * 64 + 16 * k + i16;
        a[u][k] = (live && o < out) ? dzp[row * dz_ld + o] : 0.f;
        b[u][k] = (live && i < in) ? hp[row * h_ld + i] : 0.f;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int k = 0; k < 4; ++k) bsum[k] += a[u][k];
#pragma unroll
      for (int x = 0; x < 4; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y)
          acc[x][y] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][x], b[u][y], acc[x][y], 0, 0, 0);
    }
  }
  // merge the 4 waves of the workgroup in LDS, then one set of atomics per workgroup
  __shared__ float red[3][64][64];
  if (wid > 0) {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wid - 1][(x * 4 + y) * 4 + r][lane] = acc[x][y][r];
  }
  __syncthreads();
  // D[o][i]: lane holds rows o = 4g + r, col i = lane&15.  One 16 x 16 block at a time (fenced: read all at once the three
  // other waves' 192 partials per lane set the kernel's register count -- 256 + 64, one wave per SIMD)
  if (wid == 0)
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 4; ++y) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = (x * 4 + y) * 4 + r;
          const float v = acc[x][y][r] + red[0][e][lane] + red[1][e][lane] + red[2][e][lane];
          const int o = ob * 64 + 16 * x + 4 * g + r, i = ib * 64 + 16 * y + i16;
          if (o < out && i < in) unsafeAtomicAdd(dW + (size_t)o * in + i, v);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
  if (db && ib == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = bsum[k];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int o = ob * 64 + 16 * k + i16;
      if (g == 0 && o < out) unsafeAtomicAdd(db + o, v);
    }
  }
}

// mlp_chain.hip: register-chained kernels for NeuRAD's own MLP shapes; NRHIP_ERR_UNSUPPORTED = not covered
int mlp_chain_fwd(const nrhip_mlp* m, const float* x, int64_t n, float* y, float* hidden, void* stream);
int mlp_chain_bwd(const nrhip_mlp* m, const float* x, const float* hidden, const float* gy, int64_t n, float* gx,
                  float* dz, float* part, int64_t part_floats, float* const* gw, float* const* gbias, int* done_mask,
                  void* stream);
int mlp_chain_bwd_residual(const nrhip_mlp* m, const float* x, const float* hidden, const float* gy, const float* col0,
                           int64_t n, float* g_geo, float* dz, float* part, int64_t part_floats, float* const* gw,
                           float* const* gbias, int* done_mask, void* stream);
int64_t mlp_chain_part_floats(const nrhip_mlp* m);
static bool use_chain() { return !tuning().mlp_generic; }  // NRHIP_MLP_GENERIC: A/B switch (tests run both paths)

int validate_mlp(const nrhip_mlp* m) {
  NR_REQUIRE(m, NRHIP_ERR_INVALID_ARG, "mlp descriptor is NULL");
  NR_REQUIRE(m->num_layers >= 1 && m->num_layers <= NRHIP_MAX_LAYERS, NRHIP_ERR_INVALID_ARG,
             "mlp num_layers %d outside [1,%d]", m->num_layers, NRHIP_MAX_LAYERS);
  NR_REQUIRE(m->in_dim >= 1 && m->out_dim >= 1 && (m->num_layers == 1 || m->hidden_dim >= 1) &&
                 m->in_dim <= 256 && m->out_dim <= 256 && m->hidden_dim <= 256,
             NRHIP_ERR_INVALID_ARG, "mlp dims (%d,%d,%d) outside [1,256]", m->in_dim, m->hidden_dim, m->out_dim);
  for (int l = 0; l < m->num_layers; ++l)
    NR_REQUIRE(m->weight[l], NRHIP_ERR_INVALID_ARG, "mlp weight[%d] is NULL", l);
  return NRHIP_OK;
}

MlpDev to_dev(const nrhip_mlp& m) {
  MlpDev d;
  d.in_dim = m.in_dim, d.hidden = m.hidden_dim, d.out_dim = m.out_dim, d.nl = m.num_layers;
  for (int l = 0; l < NRHIP_MAX_LAYERS; ++l) {
    d.w[l] = l < m.num_layers ? m.weight[l] : nullptr;
    d.b[l] = l < m.num_layers ? m.bias[l] : nullptr;
  }
  return d;
}

template <bool TRANSPOSED>
static size_t lds_bytes(const MlpDev& d, int waves) {
  size_t w = 0;
  for (int l = 0; l < d.nl; ++l) w += frag_floats<TRANSPOSED>(d, l);
  return (w + (size_t)waves * 2 * 16 * act_ld(d)) * sizeof(float);
}
// 4 waves per workgroup unless the weights leave too little LDS for 4 activation slabs
template <bool TRANSPOSED>
static int pick_waves(const MlpDev& d) {
  for (int w = 4; w >= 1; w >>= 1)
    if (lds_bytes<TRANSPOSED>(d, w) <= 160 * 1024) return w;
  return 0;
}

static int blocks_for_tiles(int64_t n, int waves) {
  const int64_t tiles = (n + 15) / 16;
  int64_t b = (tiles + waves - 1) / waves;
  if (b > 2048) b = 2048;  // grid-stride beyond 8 workgroups per CU
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_mlp_fwd(const nrhip_mlp* m, const float* x, int64_t n, float* y, float* hidden, void* stream) {
  if (int e = validate_mlp(m)) return e;
  NR_REQUIRE(n >= 0, NRHIP_ERR_INVALID_ARG, "mlp_fwd: negative n");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(x && y, NRHIP_ERR_INVALID_ARG, "mlp_fwd: null pointer");
  if (use_chain()) {
    const int rc = mlp_chain_fwd(m, x, n, y, hidden, stream);
    if (rc != NRHIP_ERR_UNSUPPORTED) return rc;
  }
  const MlpDev d = to_dev(*m);
  const int waves = pick_waves<false>(d);
  NR_REQUIRE(waves > 0, NRHIP_ERR_UNSUPPORTED, "mlp_fwd: weights need %zu B of LDS (> 160 KiB)",
             lds_bytes<false>(d, 1));
  const size_t lds = lds_bytes<false>(d, waves);
  static thread_local size_t configured = 0;
  if (lds > 64 * 1024 && lds > configured) {
    (void)hipFuncSetAttribute((const void*)mlp_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    configured = lds;
  }
  mlp_fwd_kernel<<<blocks_for_tiles(n, waves), 64 * waves, lds, (hipStream_t)stream>>>(d, x, n, y, hidden);
  return check_launch("mlp_fwd");
}

namespace {
int64_t dz_floats(const nrhip_mlp* m, int64_t n) { return n * (int64_t)(m->num_layers - 1) * m->hidden_dim; }
}  // namespace

extern "C" int nrhip_mlp_bwd_workspace(const nrhip_mlp* m, int64_t n, int64_t* floats) {
  if (int e = validate_mlp(m)) return e;
  NR_REQUIRE(floats && n >= 0, NRHIP_ERR_INVALID_ARG, "mlp_bwd_workspace: bad argument");
  *floats = ((dz_floats(m, n) + 3) & ~(int64_t)3) + mlp_chain_part_floats(m);
  return NRHIP_OK;
}

namespace {
// weight / bias gradients of the layers the chained kernel did not cover (bit l of wgrad_done = layer l is done)
int run_wgrad(const MlpDev& d, const float* x, const float* hidden, const float* grad_y, const float* workspace,
              float* const* grad_weight, float* const* grad_bias, int wgrad_done, int64_t n, hipStream_t st) {
  const int hid_ld = (d.nl - 1) * d.hidden;
  WgradArgs wa{};
  int nsub = 0;
  for (int l = 0; l < d.nl; ++l) {
    if (!grad_weight[l] || ((wgrad_done >> l) & 1)) continue;
    const int in = layer_in(d, l), out = layer_out(d, l);
    WgradLayer& L = wa.layer[wa.nl++];
    L.dz = (l == d.nl - 1) ? grad_y : workspace + (size_t)l * d.hidden;
    L.dz_ld = (l == d.nl - 1) ? d.out_dim : hid_ld;
    L.h = (l == 0) ? x : hidden + (size_t)(l - 1) * d.hidden;
    L.h_ld = (l == 0) ? d.in_dim : hid_ld;
    L.dW = grad_weight[l];
    L.db = grad_bias ? grad_bias[l] : nullptr;
    L.out = out, L.in = in, L.nb_in = (in + 63) / 64;
    L.sub0 = nsub;
    nsub += ((out + 63) / 64) * L.nb_in;
  }
  if (wa.nl > 0) {
    int64_t bx = ((n + 3) / 4 + 4 * 64 - 1) / (4 * 64);  // >= 64 sample-quads per wave
    // every workgroup ends in out*in memory-side atomics: measured optimum on 524 288 samples (re-measured once the kernel fit
    // three waves per SIMD: 512 .. 1024 workgroups are 18 us slower)
    if (bx > 256) bx = 256;
    if (bx < 1) bx = 1;
    mlp_wgrad_kernel<<<dim3((unsigned)bx, (unsigned)nsub), 256, 0, st>>>(wa, n);
    if (int e = check_launch("mlp_wgrad")) return e;
  }
  return NRHIP_OK;
}
}  // namespace

extern "C" int nrhip_mlp_bwd(const nrhip_mlp* m, const float* x, const float* hidden, const float* grad_y, int64_t n,
                             float* grad_x, float* const* grad_weight, float* const* grad_bias, float* workspace,
                             int64_t workspace_floats, void* stream) {
  if (int e = validate_mlp(m)) return e;
  NR_REQUIRE(n >= 0 && grad_weight, NRHIP_ERR_INVALID_ARG, "mlp_bwd: bad argument");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(x && grad_y, NRHIP_ERR_INVALID_ARG, "mlp_bwd: null pointer");
  NR_REQUIRE(m->num_layers == 1 || (hidden && workspace && workspace_floats >= dz_floats(m, n)),
             NRHIP_ERR_INVALID_ARG,
             "mlp_bwd: hidden activations and a workspace of >= N*(num_layers-1)*hidden_dim floats are required");
  const MlpDev d = to_dev(*m);
  const hipStream_t st = (hipStream_t)stream;
  int chained = NRHIP_ERR_UNSUPPORTED;
  int wgrad_done = 0;  // bit l: layer l's weight gradient came out of the chained kernel
  if (use_chain() && d.nl > 1) {
    // room behind the dZ block (nrhip_mlp_bwd_workspace) lets the chained kernel produce the weight gradients too
    const int64_t part_off = (dz_floats(m, n) + 3) & ~(int64_t)3;
    const bool no_fused_wgrad = tuning().mlp_split_wgrad;  // NRHIP_MLP_SPLIT_WGRAD: A/B switch
    const int64_t part_floats = no_fused_wgrad ? 0 : workspace_floats - part_off;
    chained = mlp_chain_bwd(m, x, hidden, grad_y, n, grad_x, workspace, part_floats > 0 ? workspace + part_off : nullptr,
                            part_floats > 0 ? part_floats : 0, grad_weight, grad_bias, &wgrad_done, stream);
    if (chained != NRHIP_OK && chained != NRHIP_ERR_UNSUPPORTED) return chained;
  }
  if (chained == NRHIP_ERR_UNSUPPORTED && (d.nl > 1 || grad_x)) {
    const int waves = pick_waves<true>(d);
    NR_REQUIRE(waves > 0, NRHIP_ERR_UNSUPPORTED, "mlp_bwd: weights need %zu B of LDS (> 160 KiB)",
               lds_bytes<true>(d, 1));
    const size_t lds = lds_bytes<true>(d, waves);
    static thread_local size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
      (void)hipFuncSetAttribute((const void*)mlp_bwd_data_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      configured = lds;
    }
    mlp_bwd_data_kernel<<<blocks_for_tiles(n, waves), 64 * waves, lds, st>>>(d, hidden, grad_y, n, grad_x, workspace);
    if (int e = check_launch("mlp_bwd_data")) return e;
  }
  return run_wgrad(d, x, hidden, grad_y, workspace, grad_weight, grad_bias, wgrad_done, n, st);
}

extern "C" int nrhip_field_feature_bwd(const nrhip_mlp* m, const float* x, const float* hidden, const float* grad_feature,
                                       const float* grad_geo0, int64_t n, float* grad_geo, float* const* grad_weight,
                                       float* const* grad_bias, float* workspace, int64_t workspace_floats, void* stream) {
  if (int e = validate_mlp(m)) return e;
  NR_REQUIRE(n >= 0 && grad_weight, NRHIP_ERR_INVALID_ARG, "field_feature_bwd: bad argument");
  NR_REQUIRE(m->in_dim == 48 && m->out_dim == 32 && m->num_layers == 3 && (m->hidden_dim == 32 || m->hidden_dim == 64),
             NRHIP_ERR_UNSUPPORTED, "field_feature_bwd: covers the feature head 48 -> {32,64} -> {32,64} -> 32");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(x && hidden && grad_feature && grad_geo0 && grad_geo && workspace, NRHIP_ERR_INVALID_ARG,
             "field_feature_bwd: null pointer");
  for (int l = 0; l < 3; ++l)
    NR_REQUIRE(grad_weight[l], NRHIP_ERR_INVALID_ARG, "field_feature_bwd: every layer's weight gradient is formed here");
  const int64_t part_off = (dz_floats(m, n) + 3) & ~(int64_t)3;
  NR_REQUIRE(workspace_floats > part_off, NRHIP_ERR_INVALID_ARG,
             "field_feature_bwd: workspace smaller than nrhip_mlp_bwd_workspace asks for");
  int wgrad_done = 0;
  const int rc = mlp_chain_bwd_residual(m, x, hidden, grad_feature, grad_geo0, n, grad_geo, workspace, workspace + part_off,
                                        workspace_floats - part_off, grad_weight, grad_bias, &wgrad_done, stream);
  NR_REQUIRE(rc != NRHIP_ERR_UNSUPPORTED, NRHIP_ERR_UNSUPPORTED,
             "field_feature_bwd: pointers must be 16-byte aligned and the workspace sized by nrhip_mlp_bwd_workspace");
  if (rc != NRHIP_OK) return rc;
  return run_wgrad(to_dev(*m), x, hidden, grad_feature, workspace, grad_weight, grad_bias, wgrad_done, n, (hipStream_t)stream);
}
