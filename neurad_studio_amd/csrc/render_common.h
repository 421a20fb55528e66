// Pieces shared by the fused field/render kernels (render.hip: tile-serial kernel, render_pipelined.hip: the
// software-pipelined kernel): device-side field descriptor, LDS carve, weight staging, host validation.
#pragma once
#include "common.h"

namespace nrhip {

struct FieldDev {
  GridDev grid;
  const void* table;
  float scale;
  const float* gw0; const float* gb0;   // geo layer 0: [H][32], [H]
  const float* gw1; const float* gb1;   // geo layer 1: [33][H], [33]
  const float* fw0; const float* fb0;   // feat layer 0: [H][48]
  const float* fw1; const float* fb1;   // feat layer 1: [H][H]
  const float* fw2; const float* fb2;   // feat layer 2: [32][H]
  int use_sdf;
  float beta;
};

// Training forward (nrhip_field_fwd_train): what the hand-written backward needs, written in the layouts the
// operator-level kernels read ([N, width] row-major).  All null for inference.
struct SaveDev {
  float* enc;  // [N, 32]   rescaled grid features = input of the geometry MLP
  float* hg;   // [N, H]    geometry MLP hidden activations (post-ReLU)
  float* xf;   // [N, 48]   feature MLP input: geometry embedding (32) | SH of the ray direction (16)
  float* hf;   // [N, 2H]   feature MLP hidden activations, layer 0 | layer 1
};

// LDS carve (floats), H = hidden width.  Fragment-ordered weights use [mb][s/4][lane][s%4] so that one
// ds_read_b128 fetches the A fragments of 4 consecutive k-steps.
template <int H>
struct Lds {
  static constexpr int NB = H / 16;        // 16-neuron blocks of a hidden layer
  static constexpr int G0 = 0;             // geo L0 : NB blocks x 8 steps
  static constexpr int G1 = G0 + H * 32;   // geo L1 (rows 1..32): 2 blocks x H/4 steps
  static constexpr int F0 = G1 + 32 * H;   // feat L0 (geo part): NB blocks x 8 steps
  static constexpr int F1 = F0 + H * 32;   // feat L1: NB blocks x H/4 steps
  static constexpr int F2 = F1 + H * H;    // feat L2: 2 blocks x H/4 steps
  static constexpr int SHW = F2 + 32 * H;  // feat L0 SH part: [16 c][NB][4 g][4 r]
  static constexpr int SDFW = SHW + 16 * H;  // geo L1 row 0: [NB][4 g][4 r]
  static constexpr int BG0 = SDFW + H;     // biases, [blk][g][r] == natural order
  static constexpr int BG1 = BG0 + H;      // 33 -> [0] = sdf bias, [1..32]
  static constexpr int BF0 = BG1 + 36;
  static constexpr int BF1 = BF0 + H;
  static constexpr int BF2 = BF1 + H;
  static constexpr int SCAL = BF2 + 32;    // per-level scalings
  static constexpr int TOTAL = SCAL + NRHIP_MAX_LEVELS;
  static constexpr int RB = TOTAL;         // pipelined kernel: per-wave, per-ray bias of feat L0 (bias + SH part), 4 x H
  static constexpr int TOTAL_PIPELINED = RB + 4 * H;
};

// Weight staging.  W[row_off + 16mb + i][col(g,s)] goes to fragment order [mb][s4][lane][s3]; CHAIN: col =
// 16*(s/4) + 4g + s%4 (input is a D tile of the previous layer), else col = 8g + s (input is the gathered feature
// registers).
template <bool CHAIN, int NBLK, int NSTEP>
__device__ __forceinline__ float frag_src(const float* __restrict__ W, int ldw, int row_off, int e) {
  const int s3 = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
  const int s4 = rest % (NSTEP / 4), mb = rest / (NSTEP / 4);
  const int s = 4 * s4 + s3, i = lane & 15, g = lane >> 4;
  const int col = CHAIN ? (16 * (s >> 2) + 4 * g + (s & 3)) : (8 * g + s);
  return W[(size_t)(row_off + 16 * mb + i) * ldw + col];
}

// Stage all weights of the field into LDS (256-thread workgroup; caller barriers afterwards).  Every thread first
// ISSUES all of its global loads (one register each, ~60 in flight), then stores: one memory round trip for the whole
// 54 KB image instead of one per loop iteration.
template <int H>
__device__ __forceinline__ void stage_field_weights(const FieldDev& fd, float* __restrict__ lds) {
  using Ld = Lds<H>;
  constexpr int NB = H / 16;
  constexpr int T = 256;  // == blockDim.x
  constexpr int N_G0 = H * 32 / T, N_G1 = 32 * H / T, N_F0 = H * 32 / T, N_F1 = H * H / T, N_F2 = 32 * H / T,
                N_SH = 16 * H / T;
  static_assert((H * 32) % T == 0 && (H * H) % T == 0 && (16 * H) % T == 0, "regions are whole passes of the block");
  const int tid = threadIdx.x;
  float vg0[N_G0], vg1[N_G1], vf0[N_F0], vf1[N_F1], vf2[N_F2], vsh[N_SH], vs[7];
#pragma unroll
  for (int it = 0; it < N_G0; ++it) vg0[it] = frag_src<false, NB, 8>(fd.gw0, 32, 0, it * T + tid);
#pragma unroll
  for (int it = 0; it < N_G1; ++it) vg1[it] = frag_src<true, 2, H / 4>(fd.gw1, H, 1, it * T + tid);
#pragma unroll
  for (int it = 0; it < N_F0; ++it) vf0[it] = frag_src<true, NB, 8>(fd.fw0, 48, 0, it * T + tid);
#pragma unroll
  for (int it = 0; it < N_F1; ++it) vf1[it] = frag_src<true, NB, H / 4>(fd.fw1, H, 0, it * T + tid);
#pragma unroll
  for (int it = 0; it < N_F2; ++it) vf2[it] = frag_src<true, 2, H / 4>(fd.fw2, H, 0, it * T + tid);
#pragma unroll
  for (int it = 0; it < N_SH; ++it) {  // SHW[c][n] = fw0[n][32+c]
    const int e = it * T + tid, c = e / H, n = e - c * H;
    vsh[it] = fd.fw0[(size_t)n * 48 + 32 + c];
  }
  const int th = tid < H ? tid : 0, t33 = tid < 33 ? tid : 0, t32 = tid & 31;
  vs[0] = fd.gw1[th];
  vs[1] = fd.gb0 ? fd.gb0[th] : 0.f;
  vs[2] = fd.fb0 ? fd.fb0[th] : 0.f;
  vs[3] = fd.fb1 ? fd.fb1[th] : 0.f;
  vs[4] = fd.gb1 ? fd.gb1[t33] : 0.f;
  vs[5] = fd.fb2 ? fd.fb2[t32] : 0.f;
  vs[6] = fd.grid.scal[t32];
#pragma unroll
  for (int it = 0; it < N_G0; ++it) lds[Ld::G0 + it * T + tid] = vg0[it];
#pragma unroll
  for (int it = 0; it < N_G1; ++it) lds[Ld::G1 + it * T + tid] = vg1[it];
#pragma unroll
  for (int it = 0; it < N_F0; ++it) lds[Ld::F0 + it * T + tid] = vf0[it];
#pragma unroll
  for (int it = 0; it < N_F1; ++it) lds[Ld::F1 + it * T + tid] = vf1[it];
#pragma unroll
  for (int it = 0; it < N_F2; ++it) lds[Ld::F2 + it * T + tid] = vf2[it];
#pragma unroll
  for (int it = 0; it < N_SH; ++it) lds[Ld::SHW + it * T + tid] = vsh[it];
  if (tid < H) {
    lds[Ld::SDFW + tid] = vs[0];
    lds[Ld::BG0 + tid] = vs[1];
    lds[Ld::BF0 + tid] = vs[2];
    lds[Ld::BF1 + tid] = vs[3];
  }
  if (tid < 33) lds[Ld::BG1 + tid] = vs[4];
  if (tid < 32) {
    lds[Ld::BF2 + tid] = vs[5];
    lds[Ld::SCAL + tid] = vs[6];
  }
}

template <int N>
__device__ __forceinline__ float row_shr(float v, float fill) {
  return dpp_row_shr<N>(v, fill);
}
// sum over the 16 lanes of a DPP row (result valid in every lane of the row)
__device__ __forceinline__ float row_sum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}

// ---- host side ---------------------------------------------------------------------------------
int validate_field(const nrhip_field* f);
FieldDev to_dev(const nrhip_field& f);
// persistent grid size for a 256-thread fused kernel: min(ceil(R/4), CUs x resident workgroups per CU)
int persistent_blocks(const void* kernel, size_t lds_bytes, int64_t n_rays, int max_per_cu);

// Options of the fused kernels (nrhip_render_fwd_ex).
struct RenderOpts {
  float stop_eps;  // > 0: a ray stops once its transmittance falls below it (eval only); 0 = exact
  int variant;     // 0 auto, 1 tile-serial kernel, 2 pipelined, 3 pipelined + deferred last feature layer
};

// render_pipelined.hip
template <bool COMPOSITE>
int dispatch_render_pipelined(const nrhip_field* f, const nrhip_rays* rays, float* of, float* od, float* oa, float* ow,
                              float* os, float* oal, void* stream, const SaveDev& sv, const RenderOpts& opts);

}  // namespace nrhip
