// F2, fast path: NeuRAD's two MLP shapes (geo 32 -> H -> 33, feature 48|64 -> H -> H -> 32, H = 32|64) with the
// layers CHAINED IN REGISTERS, forward and data-gradient, for the operator-level (training) path.
//
// Same transposed MFMA formulation as the fused render kernel (render.hip): weights are the A operand (read from
// LDS in fragment order, one ds_read_b128 per 4 k-steps), the 16 samples of a wave tile are the B operand, and the
// D tile [neuron 16mb+4g+r][sample j] of one layer is, lane for lane, the B operand of the next -- no LDS round
// trip for activations, no cross-lane traffic.  The generic kernels in mlp.hip (any shape, activations through
// LDS, runtime loops) stay as the fallback; on config 2 they cost 360 / 440 us per call, these take a fraction.
// Exact fp32 throughout (v_mfma_f32_16x16x4_f32).
#include "common.h"

namespace nrhip {
namespace {

// Fragment image of one weight matrix.  Element e = [mb][s4][lane][s3] holds A[16mb + i][col(g, s)], s = 4*s4+s3,
// (i, g) = (lane & 15, lane >> 4).  CHAIN: col = 16*(s/4) + 4g + s%4 (the input is the previous layer's D tile);
// else col = NSTEP*g + s (the input is NSTEP consecutive floats of the sample's row, loaded from global).
// TRANS: A = W^T (data gradient).  Rows/cols beyond the matrix are zero.
template <bool CHAIN, bool TRANS, int NSTEP>
__device__ __forceinline__ float frag_src(const float* __restrict__ W, int ldw, int rows, int cols, int e) {
  const int s3 = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
  const int s4 = rest % (NSTEP / 4), mb = rest / (NSTEP / 4);
  const int s = 4 * s4 + s3, i = lane & 15, g = lane >> 4;
  const int col = CHAIN ? (16 * (s >> 2) + 4 * g + (s & 3)) : (NSTEP * g + s);
  const int row = 16 * mb + i;
  if (row >= rows || col >= cols) return 0.f;
  return TRANS ? W[(size_t)col * ldw + row] : W[(size_t)row * ldw + col];
}

// Stage one matrix: all of a thread's global loads first, then the LDS stores (one memory round trip).
template <bool CHAIN, bool TRANS, int NBLK, int NSTEP>
__device__ __forceinline__ void stage_matrix(float* dst, const float* __restrict__ W, int ldw, int rows, int cols) {
  constexpr int T = 256, COUNT = NBLK * NSTEP * 64, IT = COUNT / T;
  static_assert(COUNT % T == 0, "fragment image is a whole number of block passes");
  float v[IT];
#pragma unroll
  for (int it = 0; it < IT; ++it) v[it] = frag_src<CHAIN, TRANS, NSTEP>(W, ldw, rows, cols, it * T + threadIdx.x);
#pragma unroll
  for (int it = 0; it < IT; ++it) dst[it * T + threadIdx.x] = v[it];
}

struct ChainArgs {
  const float* w[3];
  const float* b[3];
};

template <int IN, int H, int OUT, int NL>
struct Shape {
  static_assert(NL == 2 || NL == 3, "two or three layers");
  static_assert(IN % 16 == 0 && H % 16 == 0, "input and hidden widths are whole MFMA blocks");
  static constexpr int NB = H / 16, IB = IN / 16, OB = (OUT + 15) / 16, KP = OB * 16;
  // forward image: W0 [H x IN], (W1 [H x H]), WL [KP x H], biases
  static constexpr int F_W0 = 0;
  static constexpr int F_W1 = F_W0 + H * IN;
  static constexpr int F_WL = F_W1 + (NL == 3 ? H * H : 0);
  static constexpr int F_B0 = F_WL + KP * H;
  static constexpr int F_B1 = F_B0 + H;
  static constexpr int F_BL = F_B1 + (NL == 3 ? H : 0);
  static constexpr int F_TOTAL = F_BL + KP;
  // backward image: WL^T [H x KP], (W1^T [H x H]), W0^T [IN x H]
  static constexpr int B_TL = 0;
  static constexpr int B_T1 = B_TL + H * KP;
  static constexpr int B_T0 = B_T1 + (NL == 3 ? H * H : 0);
  static constexpr int B_TOTAL = B_T0 + IN * H;
};

template <int IN, int H, int OUT, int NL>
__global__ __launch_bounds__(256) void mlp_chain_fwd_kernel(ChainArgs a, const float* __restrict__ x, int64_t n,
                                                             float* __restrict__ y, float* __restrict__ hidden) {
  using S = Shape<IN, H, OUT, NL>;
  constexpr int NB = S::NB, OB = S::OB;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_matrix<false, false, NB, IN / 4>(lds + S::F_W0, a.w[0], IN, H, IN);
  if constexpr (NL == 3) stage_matrix<true, false, NB, H / 4>(lds + S::F_W1, a.w[1], H, H, H);
  stage_matrix<true, false, OB, H / 4>(lds + S::F_WL, a.w[NL - 1], H, OUT, H);
  for (int e = threadIdx.x; e < H; e += 256) {
    lds[S::F_B0 + e] = a.b[0] ? a.b[0][e] : 0.f;
    if constexpr (NL == 3) lds[S::F_B1 + e] = a.b[1] ? a.b[1][e] : 0.f;
  }
  for (int e = threadIdx.x; e < S::KP; e += 256) lds[S::F_BL + e] = (a.b[NL - 1] && e < OUT) ? a.b[NL - 1][e] : 0.f;
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t ntiles = (n + 15) / 16;
  constexpr int HID_LD = (NL - 1) * H;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wid; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int opaque = 0;  // keeps the loop-invariant LDS weight reads inside the loop (see render.hip)
    asm volatile("" : "+v"(opaque));
    const float* lw = lds + opaque;
    const int64_t row = tile * 16 + j;
    const bool live = row < n;
    const int64_t rc = live ? row : n - 1;
    float xb[IN / 4];
    const float* xp = x + rc * IN + (IN / 4) * g;
#pragma unroll
    for (int q = 0; q < IN / 16; ++q) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xp + 4 * q);
#pragma unroll
      for (int r = 0; r < 4; ++r) xb[4 * q + r] = v[r];
    }
    f32x4 h[NB];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + S::F_B0 + 16 * mb + 4 * g);
    mfma_layer<NB, IN / 4>(lw + S::F_W0, lane, xb, h);
    float hb[H / 4];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[4 * mb + r] = h[mb][r] = fmaxf(h[mb][r], 0.f);
      if (hidden && live) *reinterpret_cast<f32x4*>(hidden + row * HID_LD + 16 * mb + 4 * g) = h[mb];
    }
    if constexpr (NL == 3) {
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + S::F_B1 + 16 * mb + 4 * g);
      mfma_layer<NB, H / 4>(lw + S::F_W1, lane, hb, h);
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) hb[4 * mb + r] = h[mb][r] = fmaxf(h[mb][r], 0.f);
        if (hidden && live) *reinterpret_cast<f32x4*>(hidden + row * HID_LD + H + 16 * mb + 4 * g) = h[mb];
      }
    }
    f32x4 o[OB];
#pragma unroll
    for (int mb = 0; mb < OB; ++mb) o[mb] = *reinterpret_cast<const f32x4*>(lw + S::F_BL + 16 * mb + 4 * g);
    mfma_layer<OB, H / 4>(lw + S::F_WL, lane, hb, o);
    if (live) {
      float* yp = y + row * OUT;
#pragma unroll
      for (int mb = 0; mb < OB; ++mb) {
        const int c0 = 16 * mb + 4 * g;
        if constexpr (OUT % 4 == 0) {
          if (c0 < OUT) *reinterpret_cast<f32x4*>(yp + c0) = o[mb];
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (c0 + r < OUT) yp[c0 + r] = o[mb][r];
        }
      }
    }
  }
}

// dZ_last = grad_y;  dH_{l-1} = W_l^T dZ_l;  dZ_{l-1} = dH_{l-1} * (h_{l-1} > 0);  grad_x = W_0^T dZ_0.
// dZ of the hidden layers go to `dz` ([N,(NL-1)*H], the layout of `hidden`) for the weight-gradient pass.
template <int IN, int H, int OUT, int NL>
__global__ __launch_bounds__(256) void mlp_chain_bwd_kernel(ChainArgs a, const float* __restrict__ hidden,
                                                             const float* __restrict__ gy, int64_t n,
                                                             float* __restrict__ gx, float* __restrict__ dz) {
  using S = Shape<IN, H, OUT, NL>;
  constexpr int NB = S::NB, IB = S::IB, KP = S::KP;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_matrix<false, true, NB, KP / 4>(lds + S::B_TL, a.w[NL - 1], H, H, OUT);
  if constexpr (NL == 3) stage_matrix<true, true, NB, H / 4>(lds + S::B_T1, a.w[1], H, H, H);
  if (gx) stage_matrix<true, true, IB, H / 4>(lds + S::B_T0, a.w[0], IN, IN, H);
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t ntiles = (n + 15) / 16;
  constexpr int HID_LD = (NL - 1) * H;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + wid; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const float* lw = lds + opaque;
    const int64_t row = tile * 16 + j;
    const bool live = row < n;
    const int64_t rc = live ? row : n - 1;
    // grad_y row: KP/4 consecutive floats per lane, zero beyond OUT
    float gb[KP / 4];
    const float* gp = gy + rc * OUT;
#pragma unroll
    for (int s = 0; s < KP / 4; ++s) {
      const int c = (KP / 4) * g + s;
      gb[s] = c < OUT ? gp[c] : 0.f;
    }
    f32x4 d[NB];
    float db[H / 4];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
    mfma_layer<NB, KP / 4>(lw + S::B_TL, lane, gb, d);
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(hidden + rc * HID_LD + (NL - 2) * H + 16 * mb + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) db[4 * mb + r] = d[mb][r] = hv[r] > 0.f ? d[mb][r] : 0.f;
      if (live) *reinterpret_cast<f32x4*>(dz + row * HID_LD + (NL - 2) * H + 16 * mb + 4 * g) = d[mb];
    }
    if constexpr (NL == 3) {
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) d[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      mfma_layer<NB, H / 4>(lw + S::B_T1, lane, db, d);
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hidden + rc * HID_LD + 16 * mb + 4 * g);
#pragma unroll
        for (int r = 0; r < 4; ++r) db[4 * mb + r] = d[mb][r] = hv[r] > 0.f ? d[mb][r] : 0.f;
        if (live) *reinterpret_cast<f32x4*>(dz + row * HID_LD + 16 * mb + 4 * g) = d[mb];
      }
    }
    if (gx) {
      f32x4 dx[IB];
#pragma unroll
      for (int mb = 0; mb < IB; ++mb) dx[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      mfma_layer<IB, H / 4>(lw + S::B_T0, lane, db, dx);
      if (live) {
#pragma unroll
        for (int mb = 0; mb < IB; ++mb) *reinterpret_cast<f32x4*>(gx + row * IN + 16 * mb + 4 * g) = dx[mb];
      }
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int grid_blocks(int64_t n) {
  int n_cu = 256;
  hipDeviceProp_t prop;
  int dev = 0;
  static thread_local int cached_cu = 0;
  if (!cached_cu) {
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
      cached_cu = prop.multiProcessorCount;
    else
      cached_cu = n_cu;
  }
  const int64_t want = ((n + 15) / 16 + 3) / 4;
  const int64_t cap = (int64_t)cached_cu * 4;  // persistent: the weight image is staged once per workgroup
  return (int)(want < cap ? (want < 1 ? 1 : want) : cap);
}

template <int IN, int H, int OUT, int NL>
int launch_fwd(const ChainArgs& a, const float* x, int64_t n, float* y, float* hidden, hipStream_t st) {
  using S = Shape<IN, H, OUT, NL>;
  auto kern = mlp_chain_fwd_kernel<IN, H, OUT, NL>;
  constexpr int lds = S::F_TOTAL * (int)sizeof(float);
  static thread_local bool configured = false;
  if (lds > 64 * 1024 && !configured) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    configured = true;
  }
  kern<<<grid_blocks(n), 256, lds, st>>>(a, x, n, y, hidden);
  return check_launch("mlp_chain_fwd");
}

template <int IN, int H, int OUT, int NL>
int launch_bwd(const ChainArgs& a, const float* hidden, const float* gy, int64_t n, float* gx, float* dz,
               hipStream_t st) {
  using S = Shape<IN, H, OUT, NL>;
  auto kern = mlp_chain_bwd_kernel<IN, H, OUT, NL>;
  constexpr int lds = S::B_TOTAL * (int)sizeof(float);
  static thread_local bool configured = false;
  if (lds > 64 * 1024 && !configured) {
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    configured = true;
  }
  kern<<<grid_blocks(n), 256, lds, st>>>(a, hidden, gy, n, gx, dz);
  return check_launch("mlp_chain_bwd");
}

// the shapes NeuRADField builds (neurad_field.py:108-133): geo 32 -> H -> 1+32, feature (32+16 | 32+16+16) -> H -> H -> 32
#define NR_CHAIN_SHAPES(X) \
  X(32, 64, 33, 2) X(32, 32, 33, 2) X(48, 64, 32, 3) X(48, 32, 32, 3) X(64, 64, 32, 3) X(64, 32, 32, 3)

}  // namespace

// Both return NRHIP_ERR_UNSUPPORTED (without setting the error string) when the shape or the pointer alignment
// is not covered; the caller then runs the generic kernels.
int mlp_chain_fwd(const nrhip_mlp* m, const float* x, int64_t n, float* y, float* hidden, void* stream) {
  if (!aligned16(x) || !aligned16(hidden) || (m->out_dim % 4 == 0 && !aligned16(y))) return NRHIP_ERR_UNSUPPORTED;
  ChainArgs a{};
  for (int l = 0; l < m->num_layers && l < 3; ++l) a.w[l] = m->weight[l], a.b[l] = m->bias[l];
#define X(IN_, H_, OUT_, NL_)                                                                       \
  if (m->in_dim == IN_ && m->hidden_dim == H_ && m->out_dim == OUT_ && m->num_layers == NL_)         \
    return launch_fwd<IN_, H_, OUT_, NL_>(a, x, n, y, hidden, (hipStream_t)stream);
  NR_CHAIN_SHAPES(X)
#undef X
  return NRHIP_ERR_UNSUPPORTED;
}

int mlp_chain_bwd(const nrhip_mlp* m, const float* hidden, const float* gy, int64_t n, float* gx, float* dz,
                  void* stream) {
  if (!aligned16(hidden) || !aligned16(dz) || !aligned16(gx)) return NRHIP_ERR_UNSUPPORTED;
  ChainArgs a{};
  for (int l = 0; l < m->num_layers && l < 3; ++l) a.w[l] = m->weight[l], a.b[l] = m->bias[l];
#define X(IN_, H_, OUT_, NL_)                                                                       \
  if (m->in_dim == IN_ && m->hidden_dim == H_ && m->out_dim == OUT_ && m->num_layers == NL_)         \
    return launch_bwd<IN_, H_, OUT_, NL_>(a, hidden, gy, n, gx, dz, (hipStream_t)stream);
  NR_CHAIN_SHAPES(X)
#undef X
  return NRHIP_ERR_UNSUPPORTED;
}

}  // namespace nrhip
