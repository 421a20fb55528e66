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

// ---- data gradient + weight gradient in one pass ---------------------------------------------------------------
// The weight gradient dW_l[o][i] = sum_n dZ_l[n][o] * a_{l-1}[n][i] is an MFMA with the SAMPLES on the K axis.  The
// separate kernel (mlp.hip) re-reads dZ and the activations of every layer from HBM (1.1 GB per step on config 2)
// and runs at memory latency; here the tile is already in flight: dZ_l goes through a wave-private LDS tile to
// turn its [neuron 4g+r][sample j] register layout into the A operand's (row = neuron, k = sample), the B operand
// (k = sample, col = input) comes straight from the rows the wave has just touched (L1/L2 hits), and the products
// accumulate in registers over the wave's whole tile loop.  Each workgroup parks ONE partial (all layers) in the
// caller's workspace; wgrad_merge_kernel sums them into dW/db -- no atomics.
// Which layers accumulate in registers: all of them when that is <= 24 accumulator tiles (96 VGPRs); the 64-wide
// feature MLP (12 + 16 + 8 tiles) keeps layers 1 and 2 and leaves layer 0 to the separate kernel in mlp.hip.
template <int IN, int H, int OUT, int NL>
constexpr int wg_mask() {
  constexpr int nb = H / 16, ib = IN / 16, ob = (OUT + 15) / 16;
  constexpr int t0 = nb * ib, t1 = NL == 3 ? nb * nb : 0, tl = ob * nb;
  if (t0 + t1 + tl <= 24) return (1 << NL) - 1;
  if (t1 + tl <= 24) return ((1 << NL) - 1) & ~1;
  return 0;
}

template <int IN, int H, int OUT, int NL>
struct WgShape {
  using S = Shape<IN, H, OUT, NL>;
  static constexpr int MASK = wg_mask<IN, H, OUT, NL>();
  static constexpr bool L0 = (MASK & 1) != 0, L1 = NL == 3 && (MASK & 2) != 0, LL = (MASK >> (NL - 1)) & 1;
  static constexpr int NB = S::NB, IB = S::IB, OB = S::OB, KS = S::KP / 4;
  static constexpr int A0 = 0;                            // layer 0  [H x IN]   NB x IB blocks
  static constexpr int A1 = A0 + (L0 ? NB * IB : 0);      // layer 1  [H x H]    NB x NB blocks (NL == 3)
  static constexpr int AL = A1 + (L1 ? NB * NB : 0);      // last     [OUT x H]  OB x NB blocks
  static constexpr int NACC = AL + (LL ? OB * NB : 0);    // f32x4 accumulators per lane
  // bias partial sums, per lane: hidden layers in D layout (NB f32x4 each), last layer as the KS grad_y columns
  static constexpr int B0 = NACC * 4;                     // float slots
  static constexpr int B1 = B0 + (L0 ? NB * 4 : 0);
  static constexpr int BL = B1 + (L1 ? NB * 4 : 0);
  static constexpr int NSLOT = BL + (LL ? KS : 0);        // float slots per lane
  static constexpr int PART_FLOATS = NSLOT * 64;          // one workgroup's partial
  static constexpr int LD = H + 16;                       // dZ tile row stride: rows 4s+k land in distinct banks
  static constexpr int TILE = 16 * LD;
};

// RES (the feature head of NeuRADField, feature = geo[:, 1:] + mlp([geo[:, 1:] | sh]), neurad_field.py:146-152): instead
// of grad_x the kernel writes the geometry MLP's whole output gradient, gx[n][0] = col0[n] (the sdf / density logit's
// gradient) and gx[n][1 + c] = gy[n][c] + grad_x[n][c] for the OUT embedding columns -- rows of OUT + 1 floats.  The
// gradient of the SH columns has no consumer and is not formed.
template <int IN, int H, int OUT, int NL, bool RES = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(H == 32 ? 3 : 2, H == 32 ? 3 : 2))) void mlp_chain_bwd_wg_kernel(ChainArgs a, const float* __restrict__ x,
                                                                const float* __restrict__ hidden,
                                                                const float* __restrict__ gy, int64_t n,
                                                                float* __restrict__ gx, float* __restrict__ dz,
                                                                float* __restrict__ part,
                                                                const float* __restrict__ col0 = nullptr) {
  static_assert(!RES || (OUT % 16 == 0 && OUT <= IN), "residual head: whole blocks of embedding columns");
  using S = Shape<IN, H, OUT, NL>;
  using W = WgShape<IN, H, OUT, NL>;
  constexpr int NB = S::NB, IB = S::IB, OB = S::OB, KP = S::KP, KS = W::KS;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  stage_matrix<false, true, NB, KP / 4>(lds + S::B_TL, a.w[NL - 1], H, H, OUT);
  if constexpr (NL == 3) stage_matrix<true, true, NB, H / 4>(lds + S::B_T1, a.w[1], H, H, H);
  if (gx) stage_matrix<true, true, IB, H / 4>(lds + S::B_T0, a.w[0], IN, IN, H);
  __syncthreads();

  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int64_t ntiles = (n + 15) / 16;
  constexpr int HID_LD = (NL - 1) * H;
  float* tz = lds + S::B_TOTAL + wid * W::TILE;  // wave-private dZ tile [16 samples][LD]
  constexpr int RL = OUT + 1;                     // RES: row length of the geometry MLP's output gradient

  constexpr bool WL0 = W::L0, WL1 = W::L1, WLL = W::LL;
  // the weight gradient of the layer whose dZ is produced first / second in the chain (layer NL-2 / layer 0 of 3)
  constexpr bool WFIRST = NL == 3 ? WL1 : WL0;
  f32x4 acc[W::NACC];
#pragma unroll
  for (int q = 0; q < W::NACC; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 bs0[NB], bs1[NL == 3 ? NB : 1];
  float bsl[KS];
#pragma unroll
  for (int mb = 0; mb < NB; ++mb) bs0[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mb = 0; mb < (NL == 3 ? NB : 1); ++mb) bs1[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < KS; ++q) bsl[q] = 0.f;

  for (int64_t tile = (int64_t)blockIdx.x * 4 + wid; tile < ntiles; tile += (int64_t)gridDim.x * 4) {
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const float* lw = lds + opaque;
    const int64_t row0 = tile * 16;
    const int64_t row = row0 + j;
    const bool live = row < n;
    const int64_t rc = live ? row : n - 1;
    // rows of the four k-steps of a weight-gradient MFMA: sample 4s + g, for the lane's operand column j
    int64_t rk[4];
    bool lk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t rr = row0 + 4 * q + g;
      lk[q] = rr < n;
      rk[q] = lk[q] ? rr : n - 1;
    }
    float gb[KP / 4];
    const float* gp = gy + rc * OUT;
#pragma unroll
    for (int q = 0; q < KP / 4; ++q) {
      const int c = (KP / 4) * g + q;
      gb[q] = (c < OUT && live) ? gp[c] : 0.f;  // dead rows contribute nothing (data path never stores them)
      if constexpr (WLL) bsl[q] += gb[q];
    }
    // ---- last layer: dW_L[o][i] += gy[n][o] * h_{NL-2}[n][i], both operands straight from global ------------
    if constexpr (WLL)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float av[OB], bv[NB];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob) av[ob] = (lk[q] && 16 * ob + j < OUT) ? gy[rk[q] * OUT + 16 * ob + j] : 0.f;
#pragma unroll
      for (int ib = 0; ib < NB; ++ib) bv[ib] = hidden[rk[q] * HID_LD + (NL - 2) * H + 16 * ib + j];
#pragma unroll
      for (int ob = 0; ob < OB; ++ob)
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
          acc[W::AL + ob * NB + ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ob], bv[ib], acc[W::AL + ob * NB + ib], 0, 0, 0);
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
      // dZ goes to memory only for a layer whose weight gradient is NOT formed in here (the separate kernel reads it).
      // (48 -> 64 -> 64 -> 32 keeps the store: without it the allocator needs 12 more VGPRs, 152 + 116 AGPRs > 256, and
      //  the kernel drops to one wave per SIMD: 277 -> 304 us.)
      constexpr bool kStoreFirstDz = !WFIRST || (IN == 48 && H == 64);
      if constexpr (kStoreFirstDz)
        if (live) *reinterpret_cast<f32x4*>(dz + row * HID_LD + (NL - 2) * H + 16 * mb + 4 * g) = d[mb];
      if constexpr (WFIRST) {
        *reinterpret_cast<f32x4*>(tz + j * W::LD + 16 * mb + 4 * g) = d[mb];  // zero for dead rows (gb was zeroed)
        if constexpr (NL == 3) bs1[mb] += d[mb]; else bs0[mb] += d[mb];
      }
    }
    // ---- layer NL-2: dW[o][i] += dZ[n][o] * (NL == 3 ? h_0 : x)[n][i] ------------------------------------------
    if constexpr (WFIRST) {
      constexpr int NI = NL == 3 ? NB : IB;
      constexpr int ABASE = NL == 3 ? W::A1 : W::A0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float av[NB], bv[NI];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) av[ob] = tz[(4 * q + g) * W::LD + 16 * ob + j];
#pragma unroll
        for (int ib = 0; ib < NI; ++ib)
          bv[ib] = NL == 3 ? hidden[rk[q] * HID_LD + 16 * ib + j] : x[rk[q] * IN + 16 * ib + j];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
          for (int ib = 0; ib < NI; ++ib)
            acc[ABASE + ob * NI + ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ob], bv[ib], acc[ABASE + ob * NI + ib], 0, 0, 0);
      }
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
        if constexpr (!WL0)
          if (live) *reinterpret_cast<f32x4*>(dz + row * HID_LD + 16 * mb + 4 * g) = d[mb];
        if constexpr (WL0) {
          *reinterpret_cast<f32x4*>(tz + j * W::LD + 16 * mb + 4 * g) = d[mb];
          bs0[mb] += d[mb];
        }
      }
      // ---- layer 0: dW_0[o][i] += dZ_0[n][o] * x[n][i] ---------------------------------------------------------
      if constexpr (WL0)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float av[NB], bv[IB];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob) av[ob] = tz[(4 * q + g) * W::LD + 16 * ob + j];
#pragma unroll
        for (int ib = 0; ib < IB; ++ib) bv[ib] = x[rk[q] * IN + 16 * ib + j];
#pragma unroll
        for (int ob = 0; ob < NB; ++ob)
#pragma unroll
          for (int ib = 0; ib < IB; ++ib)
            acc[W::A0 + ob * IB + ib] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ob], bv[ib], acc[W::A0 + ob * IB + ib], 0, 0, 0);
      }
    }
    if (gx) {
      constexpr int XB = RES ? OUT / 16 : IB;  // RES: the SH blocks of grad_x are never formed (block-major weight image)
      f32x4 dx[XB];
#pragma unroll
      for (int mb = 0; mb < XB; ++mb) dx[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
      mfma_layer<XB, H / 4>(lw + S::B_T0, lane, db, dx);
      if constexpr (RES) {
        // Straight from the registers: each lane owns 4 consecutive columns per block of its row; the odd row stride makes
        // them four 4-byte stores.  (Gathering the tile's 16 x RL run in LDS first -- residual parked there at tile start,
        // grad_x added with ds_add_f32, 16-byte stores -- measured SLOWER: 734 vs 536 us at the c3 size, 426 vs 378 us at
        // c1; this kernel is latency-bound at two waves per SIMD, and the LDS round trip is one more dependent chain.)
        // The MFMA chain leaves no register to spare at H = 64 (140 + 116 AGPRs): one block's grad_y at a time.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mb = 0; mb < XB; ++mb) {
          const f32x4 gf = *reinterpret_cast<const f32x4*>(gy + rc * OUT + 16 * mb + 4 * g);
          if (live) {
#pragma unroll
            for (int r = 0; r < 4; ++r) gx[row * RL + 1 + 16 * mb + 4 * g + r] = dx[mb][r] + gf[r];
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        if (g == 0 && live) gx[row * RL] = col0[rc];
      } else if (live) {
#pragma unroll
        for (int mb = 0; mb < IB; ++mb) *reinterpret_cast<f32x4*>(gx + row * IN + 16 * mb + 4 * g) = dx[mb];
      }
    }
  }

  // ---- merge the four waves (through LDS, one wave at a time), then park the workgroup's partial ---------------
  __syncthreads();  // all waves are done with the weight image: reuse it
  float* red = lds;
  for (int w = 1; w < 4; ++w) {
    if (wid == w) {
#pragma unroll
      for (int q = 0; q < W::NACC; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[(q * 4 + r) * 64 + lane] = acc[q][r];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (WL0) red[(W::B0 + mb * 4 + r) * 64 + lane] = bs0[mb][r];
          if constexpr (WL1) red[(W::B1 + mb * 4 + r) * 64 + lane] = bs1[mb][r];
        }
      if constexpr (WLL)
#pragma unroll
        for (int q = 0; q < KS; ++q) red[(W::BL + q) * 64 + lane] = bsl[q];
    }
    __syncthreads();
    if (wid == 0) {
#pragma unroll
      for (int q = 0; q < W::NACC; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[q][r] += red[(q * 4 + r) * 64 + lane];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (WL0) bs0[mb][r] += red[(W::B0 + mb * 4 + r) * 64 + lane];
          if constexpr (WL1) bs1[mb][r] += red[(W::B1 + mb * 4 + r) * 64 + lane];
        }
      if constexpr (WLL)
#pragma unroll
        for (int q = 0; q < KS; ++q) bsl[q] += red[(W::BL + q) * 64 + lane];
    }
    __syncthreads();
  }
  if (wid == 0) {
    float* pp = part + (size_t)blockIdx.x * W::PART_FLOATS;
#pragma unroll
    for (int q = 0; q < W::NACC; ++q)
#pragma unroll
      for (int r = 0; r < 4; ++r) pp[(q * 4 + r) * 64 + lane] = acc[q][r];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (WL0) pp[(W::B0 + mb * 4 + r) * 64 + lane] = bs0[mb][r];
        if constexpr (WL1) pp[(W::B1 + mb * 4 + r) * 64 + lane] = bs1[mb][r];
      }
    if constexpr (WLL)
#pragma unroll
      for (int q = 0; q < KS; ++q) pp[(W::BL + q) * 64 + lane] = bsl[q];
  }
}

// Second stage.  A 256-thread workgroup owns 64 consecutive float slots of the partial (one (block,r) row of an
// accumulator, or one bias slot); its four waves split the parked partials, an LDS step joins them, and lane
// arithmetic maps the slot back to (o, i).  dW/db are ACCUMULATED into (plain read-modify-write: one owner each).
struct MergeLayer {
  float* dW;
  float* db;
  int out, in, nbi;  // nbi = 16-wide input blocks of this layer
  int acc0;          // first accumulator of the layer
  int bias0, bias_kind, bias_n;  // float-slot base; 0 = D layout (o = 16mb + 4g + r), 1 = column layout (o = n*g + s)
};
struct MergeArgs {
  MergeLayer layer[3];
  int nl, nacc, nslot;
};

__global__ __launch_bounds__(256) void wgrad_merge_kernel(MergeArgs m, const float* __restrict__ part, int nparts,
                                                           int part_floats) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int slot = blockIdx.x;  // float slot (64 lanes wide)
  const float* pp = part + (size_t)slot * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int b = wid;
  for (; b + 12 < nparts; b += 16) {
    s0 += pp[(size_t)b * part_floats];
    s1 += pp[(size_t)(b + 4) * part_floats];
    s2 += pp[(size_t)(b + 8) * part_floats];
    s3 += pp[(size_t)(b + 12) * part_floats];
  }
  for (; b < nparts; b += 4) s0 += pp[(size_t)b * part_floats];
  red[wid][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wid != 0) return;
  float v = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
  const int j = lane & 15, g = lane >> 4;
  if (slot < m.nacc * 4) {  // accumulator row: slot = acc * 4 + r; D[o = 16ob + 4g + r][i = 16ib + j]
    const int q = slot >> 2, r = slot & 3;
    int li = 0;
    for (int l = 1; l < m.nl; ++l)
      if (q >= m.layer[l].acc0) li = l;
    const MergeLayer& L = m.layer[li];
    const int rel = q - L.acc0, ob = rel / L.nbi, ib = rel % L.nbi;
    const int o = 16 * ob + 4 * g + r, i = 16 * ib + j;
    if (o < L.out && i < L.in) L.dW[(size_t)o * L.in + i] += v;
    return;
  }
  // bias slot: the 16 lanes of a row hold the same neuron for 16 different samples
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  for (int l = 0; l < m.nl; ++l) {
    const MergeLayer& L = m.layer[l];
    const int cnt = L.bias_kind == 0 ? L.bias_n * 4 : L.bias_n;
    if (slot >= L.bias0 && slot < L.bias0 + cnt && L.db) {
      const int e = slot - L.bias0;
      const int o = L.bias_kind == 0 ? 16 * (e >> 2) + 4 * g + (e & 3) : L.bias_n * g + e;
      if (j == 0 && o < L.out) L.db[o] += v;
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int cu_count() {
  static thread_local int cached = 0;
  if (!cached) {
    hipDeviceProp_t prop;
    int dev = 0;
    cached = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                 ? prop.multiProcessorCount
                 : 256;
  }
  return cached;
}

// persistent grid (the weight image is staged once per workgroup): one wave per 16-sample tile, at most per_cu
// workgroups per CU
int grid_blocks(int64_t n, int per_cu = 4) {
  const int64_t want = ((n + 15) / 16 + 3) / 4;
  const int64_t cap = (int64_t)cu_count() * per_cu;
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

// Returns NRHIP_ERR_UNSUPPORTED (without setting the error string) when the shape or the pointer alignment
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

namespace {

template <int IN, int H, int OUT, int NL, bool RES = false>
int launch_bwd_wg(const nrhip_mlp* m, const ChainArgs& a, const float* x, const float* hidden, const float* gy,
                  int64_t n, float* gx, float* dz, float* part, int64_t part_floats, float* const* gw,
                  float* const* gbias, int* done_mask, hipStream_t st, const float* col0 = nullptr) {
  using S = Shape<IN, H, OUT, NL>;
  using W = WgShape<IN, H, OUT, NL>;
  // (all 36 accumulator tiles of 48|64 -> 64 -> 64 -> 32 in registers leave one wave per SIMD and measured slower
  // than the separate weight-gradient kernel, 0.44 vs 0.43 ms per call: hence wg_mask())
  if (W::MASK == 0) return NRHIP_ERR_UNSUPPORTED;
  const int64_t fit = part_floats / W::PART_FLOATS;
  if (fit < 1) return NRHIP_ERR_UNSUPPORTED;
  auto kern = mlp_chain_bwd_wg_kernel<IN, H, OUT, NL, RES>;
  constexpr int lds = (S::B_TOTAL + 4 * W::TILE > W::PART_FLOATS ? S::B_TOTAL + 4 * W::TILE : W::PART_FLOATS) *
                      (int)sizeof(float);
  static thread_local int resident = 0;  // workgroups per CU the accumulators + LDS admit: one partial per resident one
  if (!resident) {
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 1;
    resident = nb > 4 ? 4 : nb;
  }
  int blocks = grid_blocks(n, resident);
  if (blocks > fit) blocks = (int)fit;
  kern<<<blocks, 256, lds, st>>>(a, x, hidden, gy, n, gx, dz, part, col0);
  if (int e = check_launch("mlp_chain_bwd_wg")) return e;
  MergeArgs ma{};
  ma.nacc = W::NACC, ma.nslot = W::NSLOT;
  if (W::L0) ma.layer[ma.nl++] = MergeLayer{gw[0], gbias ? gbias[0] : nullptr, H, IN, W::IB, W::A0, W::B0, 0, W::NB};
  if (W::L1) ma.layer[ma.nl++] = MergeLayer{gw[1], gbias ? gbias[1] : nullptr, H, H, W::NB, W::A1, W::B1, 0, W::NB};
  if (W::LL)
    ma.layer[ma.nl++] = MergeLayer{gw[NL - 1], gbias ? gbias[NL - 1] : nullptr, OUT, H, W::NB, W::AL, W::BL, 1, W::KS};
  wgrad_merge_kernel<<<W::NSLOT, 256, 0, st>>>(ma, part, blocks, W::PART_FLOATS);
  if (int e = check_launch("mlp_wgrad_merge")) return e;
  *done_mask = W::MASK;
  return NRHIP_OK;
}

}  // namespace

// floats of scratch the fused weight gradient wants for this MLP (0: shape not covered)
int64_t mlp_chain_part_floats(const nrhip_mlp* m) {
#define X(IN_, H_, OUT_, NL_)                                                                  \
  if (m->in_dim == IN_ && m->hidden_dim == H_ && m->out_dim == OUT_ && m->num_layers == NL_)    \
    return (int64_t)WgShape<IN_, H_, OUT_, NL_>::PART_FLOATS * 1024;
  NR_CHAIN_SHAPES(X)
#undef X
  return 0;
}

// Data gradient; when `part` has room and every layer wants a weight gradient, also the weight gradients of the
// layers in *done_mask (bit l = layer l); the caller runs the separate kernel for the others.
int mlp_chain_bwd(const nrhip_mlp* m, const float* x, const float* hidden, const float* gy, int64_t n, float* gx,
                  float* dz, float* part, int64_t part_floats, float* const* gw, float* const* gbias, int* done_mask,
                  void* stream) {
  *done_mask = 0;
  if (!aligned16(hidden) || !aligned16(dz) || !aligned16(gx)) return NRHIP_ERR_UNSUPPORTED;
  ChainArgs a{};
  for (int l = 0; l < m->num_layers && l < 3; ++l) a.w[l] = m->weight[l], a.b[l] = m->bias[l];
  bool all_w = part != nullptr && gw != nullptr;
  for (int l = 0; all_w && l < m->num_layers; ++l) all_w = gw[l] != nullptr;
  if (all_w) {
#define X(IN_, H_, OUT_, NL_)                                                                      \
  if (m->in_dim == IN_ && m->hidden_dim == H_ && m->out_dim == OUT_ && m->num_layers == NL_) {       \
    const int rc = launch_bwd_wg<IN_, H_, OUT_, NL_>(m, a, x, hidden, gy, n, gx, dz, part, part_floats, gw, gbias, \
                                                     done_mask, (hipStream_t)stream);               \
    if (rc != NRHIP_ERR_UNSUPPORTED) return rc;                                                     \
  }
    NR_CHAIN_SHAPES(X)
#undef X
  }
#define X(IN_, H_, OUT_, NL_)                                                                       \
  if (m->in_dim == IN_ && m->hidden_dim == H_ && m->out_dim == OUT_ && m->num_layers == NL_)         \
    return launch_bwd<IN_, H_, OUT_, NL_>(a, hidden, gy, n, gx, dz, (hipStream_t)stream);
  NR_CHAIN_SHAPES(X)
#undef X
  return NRHIP_ERR_UNSUPPORTED;
}

// Feature head backward (see RES above): x [n,48] = (embedding | sh), gy [n,32], col0 [n] -> g_geo [n,33] + all three
// weight / bias gradients; done_mask as in mlp_chain_bwd.  Covers the 48 -> H -> H -> 32 shapes only.
int mlp_chain_bwd_residual(const nrhip_mlp* m, const float* x, const float* hidden, const float* gy, const float* col0,
                           int64_t n, float* g_geo, float* dz, float* part, int64_t part_floats, float* const* gw,
                           float* const* gbias, int* done_mask, void* stream) {
  *done_mask = 0;
  if (!aligned16(hidden) || !aligned16(dz) || !aligned16(g_geo) || !aligned16(gy) || !part) return NRHIP_ERR_UNSUPPORTED;
  ChainArgs a{};
  for (int l = 0; l < m->num_layers && l < 3; ++l) a.w[l] = m->weight[l], a.b[l] = m->bias[l];
#define X(H_)                                                                                                     \
  if (m->in_dim == 48 && m->hidden_dim == H_ && m->out_dim == 32 && m->num_layers == 3)                             \
    return launch_bwd_wg<48, H_, 32, 3, true>(m, a, x, hidden, gy, n, g_geo, dz, part, part_floats, gw, gbias, done_mask, \
                                              (hipStream_t)stream, col0);
  X(64) X(32)
#undef X
  return NRHIP_ERR_UNSUPPORTED;
}

}  // namespace nrhip
