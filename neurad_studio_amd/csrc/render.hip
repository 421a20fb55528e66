// F1 (+C1+C2) fused: multi-resolution hash-grid lookup -> tiny MLPs on the matrix cores -> SDF/density
// head -> transmittance scan -> feature/depth/accumulation compositing, one wavefront per ray.
//
// Wave layout (tile = 16 consecutive samples of one ray):   lane = 16*g + j
//     j = lane & 15  : sample inside the tile          g = lane >> 4 : "k-group" (0..3)
//   * gather: the 4 lanes (j, g=0..3) of a sample each look up L/4 levels  -> 8 features per lane.  Those 8
//     registers ARE the MFMA B-fragments of geo-layer 1 (v_mfma_f32_16x16x4_f32: B[k = lane>>4][n = lane&15]),
//     the weight matrix is the A operand with its K axis permuted to match (done once, at LDS staging).
//   * transposed chaining H^T = W . X^T : the D tile comes out as D[neuron = 4g + r][sample = j]  (r = 0..3),
//     i.e. lane (j,g) again holds 4 activations of ITS sample per 16-neuron block -> they feed the next layer's
//     B operand directly.  No cross-lane traffic, no LDS round trip between the five layers.
//   * compositing: the 16 samples of a tile live in one DPP row (16 lanes) -> exclusive transmittance scan
//     with row_shr DPP ops; the four rows (g) carry identical copies, each accumulates for its own channels.
// Exact fp32 (f32-input MFMA == fmaf chain): the parity target is the reference's fp32 torch path.
//
// What bounds this kernel on MI355X is the chip's random-line miss rate (every fine-level corner is a new 128-byte
// line for 8-16 useful bytes), with the fp32 MFMA chain behind it.  Hence:
//   * SOFTWARE PIPELINE.  A wave's (ray, tile) sequence is flattened; the 8*L/4 corner gathers of the NEXT tile (the
//     next tile of this ray or the first tile of the wave's next ray) are issued right after the current tile's blend,
//     BEFORE its ~170 MFMAs, and stay in flight in VGPRs during the MLP (2 waves / SIMD, ~220 VGPRs); the tile after
//     that already has its sample interval and ray constants requested.  A wave therefore keeps requests in the
//     memory system all the time instead of idling on memory for the ~2 us of its MFMA phase.  Ray constants are
//     read at wave-uniform addresses from `const __restrict__` arguments -> scalar loads, no VGPRs, no vmcnt.
//   * DEFERRED LAST LAYER (composited output).  The last feature layer (H -> 32, no activation) is linear and followed
//     only by sum_s w_s * (.), so the kernel accumulates sum_s w_s*h2_s (H channels) and sum_s w_s*e_s per ray and
//     applies fw2 / fb2 ONCE per ray:  sum_s w_s (fw2 h2_s + fb2 + e_s) = fw2 (sum w h2) + fb2 sum w + sum w e.
//     -32 of 192 MFMAs per tile at H=64; reassociation only (~1e-7 relative).
//   * XCD-COHERENT RAY RANGES.  Workgroup b runs on XCD b % 8 (dispatch order; a wrong guess costs speed only), so the
//     processing order is cut into 8 contiguous ranges and XCD x walks range x with its own workgroups: rays that are
//     neighbours in the order (a 32x32 camera patch; any batch after nrhip_ray_order) share ONE L2 instead of leaving
//     copies of their lines in all eight.  `rays.order` (optional permutation) supplies that order without moving data.
//   * EARLY RAY TERMINATION (eval option, off by default = exact): a ray stops once the transmittance entering a tile
//     is below `stop_eps`; what is skipped weighs less than stop_eps in total.
//   * DYNAMIC ACTORS (ACT instantiations, eval): a sample inside an actor's box reads THAT actor's hash grid at its
//     box-frame position instead of the static grid (neurad_encoding.py:150-187: per-sample table select, highest actor
//     index wins, features zero-padded to 32) and feeds its box-frame direction to the SH encoding.  The ray's candidate
//     list (nrhip_actor_prepare) is walked with scalar loads -- empty for most rays -- and only tiles that contain a hit
//     replace the per-ray SH bias row by four extra k-steps with per-sample SH values.
// The sky residual (models/neurad.py:381: w_{S-1} += 1 - sum w) is folded into the last tile -- the accumulated weight is
// complete there, so no copy of the last sample's features has to be kept.
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
  uint32_t lay[NRHIP_MAX_LEVELS * 4];  // RELAY kernels: {mulY, mulZ, mask, row0} per level of the eval table (eval_layout.hip)
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
// SPLIT = 1: the four per-tile matrices are stored as 3-way bf16 splits (6 bytes per weight instead of 4, see
// mfma_layer_split); SPLIT = 2: as fp16 pairs (hi image | lo image, 4 bytes per weight, see mfma_layer_pairs).
template <int H, int SPLIT = 0>
struct Lds {
  static constexpr int NB = H / 16;        // 16-neuron blocks of a hidden layer
  static constexpr int WS(int n) { return SPLIT == 1 ? n + n / 2 : n; }
  static constexpr int G0 = 0;             // geo L0 : NB blocks x 8 steps
  static constexpr int G1 = G0 + WS(H * 32);   // geo L1 (rows 1..32): 2 blocks x H/4 steps
  static constexpr int F0 = G1 + WS(32 * H);   // feat L0 (geo part): NB blocks x 8 steps
  static constexpr int F1 = F0 + WS(H * 32);   // feat L1: NB blocks x H/4 steps
  static constexpr int F2 = F1 + WS(H * H);    // feat L2: 2 blocks x H/4 steps (always fp32: applied once per ray)
  static constexpr int SHW = F2 + 32 * H;  // feat L0 SH part: [16 c][NB][4 g][4 r]
  static constexpr int SDFW = SHW + 16 * H;  // geo L1 row 0: [NB][4 g][4 r]
  static constexpr int BG0 = SDFW + H;     // biases, [blk][g][r] == natural order
  static constexpr int BG1 = BG0 + H;      // 33 -> [0] = sdf bias, [1..32]
  static constexpr int BF0 = BG1 + 36;
  static constexpr int BF1 = BF0 + H;
  static constexpr int BF2 = BF1 + H;
  static constexpr int SCAL = BF2 + 32;    // per-level scalings
  static constexpr int RB = SCAL + NRHIP_MAX_LEVELS;  // per wave: this ray's bias of feat L0 (fb0 + SH part), 4 x H
  static constexpr int LAY = RB + 4 * H;   // RELAY: per-level {mulY, mulZ, mask, row0} (uint32), 16-byte aligned
  static constexpr int FLG = LAY + 4 * NRHIP_MAX_LEVELS;  // SPLIT = 2: a weight does not fit the fp16 pair (int, 0 / 1)
  static constexpr int TOTAL = FLG + 4;
  // ACT instantiations only:
  static constexpr int SHF = TOTAL;             // feat L0 SH part in fragment order: NB blocks x 4 steps
  static constexpr int ASCAL = SHF + 16 * H;    // actor grid: per-level scalings
  static constexpr int TOTAL_ACT = ASCAL + NRHIP_MAX_LEVELS;
};

// Fused-kernel view of the dynamic actors.  The arrays that are read at wave-uniform addresses are separate
// `const __restrict__` kernel arguments (scalar loads); this struct carries the scalars.
struct ActorFieldDev {
  int K;        // row length of the candidate lists
  int La;       // actor grid levels (<= L; same features per level as the static grid)
  int log2T;    // actor table rows per level
  float scale;  // actor_scale of the actor-frame contraction
  float scal[NRHIP_MAX_LEVELS];  // actor grid scalings
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

// ---- the matrix work on the matrix cores: 3-way split bf16 ------------------------------------------------------------
// On gfx950 v_mfma_f32_16x16x4_f32 executes at the fp32 VECTOR rate on the vector ALU and overlaps with no other VALU work
// (profiles/r02_mfma_overlap_probe.txt); v_mfma_f32_16x16x16_bf16 is a real second pipe (8 cycles, hides behind VALU
// work).  x = h + m + l with h, m, l bf16 obtained by TRUNCATION (each remainder is exact in fp32), likewise w; the six
// products h.h, h.m, m.h, h.l, l.h, m.m carry every term above 2^-24 of the full product and each is exact in the MFMA's
// fp32 accumulation -- an fp32-equivalent dot product (measured 6e-8 representational error vs 1.4e-7 rounding error of an
// fp32 GEMM), unlike plain bf16 (4e-3) or a 2-way split (3e-5).
using bf16x4 = __attribute__((ext_vector_type(4))) short;

struct Split4 {  // four consecutive k values of one lane, as the three bf16 operands
  bf16x4 h, m, l;
};

__device__ __forceinline__ uint32_t pack_hi16(uint32_t lo_src, uint32_t hi_src) {  // {hi_src[31:16], lo_src[31:16]}
  return __builtin_amdgcn_perm(hi_src, lo_src, 0x07060302u);
}

__device__ __forceinline__ Split4 split4(float x0, float x1, float x2, float x3) {
  const float x[4] = {x0, x1, x2, x3};
  uint32_t h[4], m[4], l[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    h[k] = __float_as_uint(x[k]) & 0xffff0000u;
    const float r1 = x[k] - __uint_as_float(h[k]);  // exact
    m[k] = __float_as_uint(r1) & 0xffff0000u;
    l[k] = __float_as_uint(r1 - __uint_as_float(m[k]));  // exact; its upper half is taken at the packing
  }
  Split4 s;
  s.h = __builtin_bit_cast(bf16x4, uint2{pack_hi16(h[0], h[1]), pack_hi16(h[2], h[3])});
  s.m = __builtin_bit_cast(bf16x4, uint2{pack_hi16(m[0], m[1]), pack_hi16(m[2], m[3])});
  s.l = __builtin_bit_cast(bf16x4, uint2{pack_hi16(l[0], l[1]), pack_hi16(l[2], l[3])});
  return s;
}

// One layer on a 16-sample tile: acc[mb] += W[16 mb .. +15][:] . x, K = 16 KB inputs.  Weights: LDS image
// [mb][kb][split h|m|l][lane] of 8-byte A fragments (lane (i, g): row 16 mb + i, k = 16 kb + 4 g + 0..3 in the layer's own
// input order, see stage_split_matrix; stored as a (h | m) 16-byte image followed by an l 8-byte image); activations
// b[4 kb + v]: this lane's inputs of k-block kb.
template <int NBLK, int KB>
__device__ __forceinline__ void mfma_layer_split(const float* __restrict__ wf, int lane, const float (&b)[4 * KB],
                                                 f32x4 (&acc)[NBLK]) {
  const uint2* w2 = reinterpret_cast<const uint2*>(wf);
  // Consecutive MFMAs go to DIFFERENT accumulators: a dependent bf16 MFMA waits ~3 issue slots for its predecessor, so the
  // six terms of one block issued back to back run no faster than the fp32 MFMA they replace.  Two-block layers get a second
  // accumulator set for that (terms 0-2 | 3-5), summed at the end.
  constexpr bool kTwoSets = NBLK < 4;
  f32x4 acc2[kTwoSets ? NBLK : 1];
  if constexpr (kTwoSets) {
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb) acc2[mb] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    const Split4 x = split4(b[4 * kb], b[4 * kb + 1], b[4 * kb + 2], b[4 * kb + 3]);
    bf16x4 ah[NBLK], am[NBLK], al[NBLK];
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb) {  // one 16-byte read (h | m) + one 8-byte read (l) per block
      const uint4 hm = reinterpret_cast<const uint4*>(wf)[(mb * KB + kb) * 64 + lane];
      ah[mb] = __builtin_bit_cast(bf16x4, uint2{hm.x, hm.y});
      am[mb] = __builtin_bit_cast(bf16x4, uint2{hm.z, hm.w});
      al[mb] = __builtin_bit_cast(bf16x4, w2[NBLK * KB * 128 + (mb * KB + kb) * 64 + lane]);
    }
#define NR_TERM(ACC, A_, X_)                                                                          \
  _Pragma("unroll") for (int mb = 0; mb < NBLK; ++mb) ACC[mb] =                                        \
      __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A_[mb], X_, ACC[mb], 0, 0, 0);
    if constexpr (kTwoSets) {
      NR_TERM(acc2, al, x.h) NR_TERM(acc, ah, x.h) NR_TERM(acc2, ah, x.l) NR_TERM(acc, ah, x.m)
      NR_TERM(acc2, am, x.m) NR_TERM(acc, am, x.h)
    } else {  // small terms first
      NR_TERM(acc, al, x.h) NR_TERM(acc, ah, x.l) NR_TERM(acc, am, x.m) NR_TERM(acc, am, x.h) NR_TERM(acc, ah, x.m)
      NR_TERM(acc, ah, x.h)
    }
#undef NR_TERM
  }
  if constexpr (kTwoSets) {
#pragma unroll
    for (int mb = 0; mb < NBLK; ++mb) acc[mb] += acc2[mb];
  }
}

// W[row_off + 16 mb + i][col] -> the split image above.  CHAIN: the layer's input is the previous layer's D tile (lane (j,g)
// holds neurons 16 kb + 4 g + v of block kb): col = 16 kb + 4 g + v; else the gathered features (lane holds 8 g .. 8 g + 7,
// k-block kb takes its features 4 kb .. 4 kb + 3): col = 8 g + 4 kb + v.
template <bool CHAIN, int NBLK, int KB>
__device__ __forceinline__ void stage_split_matrix(float* __restrict__ dst, const float* __restrict__ W, int ldw,
                                                   int row_off) {
  unsigned short* d16 = reinterpret_cast<unsigned short*>(dst);
  for (int e = threadIdx.x; e < NBLK * KB * 64 * 4; e += 256) {
    const int v = e & 3, lane = (e >> 2) & 63, rest = e >> 8;
    const int kb = rest % KB, mb = rest / KB;
    const int i = lane & 15, g = lane >> 4;
    const int col = CHAIN ? (16 * kb + 4 * g + v) : (8 * g + 4 * kb + v);
    const float w = W[(size_t)(row_off + 16 * mb + i) * ldw + col];
    const uint32_t h = __float_as_uint(w) & 0xffff0000u;
    const float r1 = w - __uint_as_float(h);
    const uint32_t m = __float_as_uint(r1) & 0xffff0000u;
    const uint32_t l = __float_as_uint(r1 - __uint_as_float(m));
    // in bf16 units: [mb][kb][lane][h0..h3 m0..m3] (16 bytes per lane), then [mb][kb][lane][l0..l3] (8 bytes per lane)
    const int blk = (mb * KB + kb) * 64 + lane;
    d16[blk * 8 + v] = (unsigned short)(h >> 16);
    d16[blk * 8 + 4 + v] = (unsigned short)(m >> 16);
    d16[NBLK * KB * 64 * 8 + blk * 4 + v] = (unsigned short)(l >> 16);
  }
}

// ---- the matrix work on the matrix cores, second form: fp16 PAIRS -------------------------------------------------------
// x = h + l with h = fp16(x) and l = fp16(x - h) (both round-to-nearest; the remainder is exact in fp32): 22-24 significant
// bits, and the three products h.h, l.h, h.l carry every term above 2^-22 of the full product, each exact in the MFMA's
// fp32 accumulation.  One v_mfma_f32_16x16x32_f16 per 32 inputs and term: 60 MFMAs per 16-sample tile at H = 64, on a pipe of
// their own, where the fp32 form issues 160 v_mfma_f32_16x16x4_f32 on the vector ALU and the 3-way bf16 split 240 of K = 16;
// 2.5 VALU instructions per activation for the split instead of the bf16 split's 5.4 (gfx950 converts and subtracts PAIRS:
// v_cvt_pk_f16_f32, v_pk_add_f32).  fp16's exponent range is the price; it is spent where the network lives --
//   * the whole tile runs in units of kPairAct = 2^6 (ReLU MLPs are positively homogeneous: the biases and the feature inputs
//     are staged / blended x 2^6, the three exits -- sdf row, deferred last layer, embedding sum -- x 2^-6; all exact), so an
//     activation keeps its full precision from 1e-6 up to 1000;
//   * the weights are staged x kPairW = 2^7 and the biases x 2^7 on top, so that the scaled sum accumulates onto the bias in
//     place and comes back through one multiplication by 2^-7 per output;
//   * a tile with an input that does not fit (|x| >= 1000 in true units, infinities) or a matrix with |w| >= 500 takes the
//     fp32 MFMA with its A fragments read from global memory: slow, and the fp32-MFMA kernel's numbers.
// numpy emulation (scripts/pairs_emulation.py, profiles/r05_pairs_emulation.txt): rel. error of a 64-term product sum
// 0.7-1.0e-7 for activations >= 1e-2 (2.7e-7 at 1e-3) -- the fp32 product sum's own 1.0e-7 -- against 1.7e-6 .. 1.7e-4
// without the two recentrings.  On the bench batch the two forms' outputs are 1.0e-7 apart (bench.py: mlp_products).
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f32x2 = __attribute__((ext_vector_type(2))) float;
constexpr float kPairAct = 64.f, kPairW = 128.f, kPairFit = 65000.f;
constexpr bool kPairsDefault = true;  // NRHIP_MLP_PAIRS unset (=0: the fp32 MFMA)

// source of element e of a pair image: [mb][q][lane][t] halves (lane (i, g) slot t of the K = 32 step q: row 16 mb + i,
// input of (g, 8 q + t) in the layer's own order -- CHAIN: the previous layer's D tiles (lane (j, g) holds neurons
// 16 kb + 4 g + v of block kb = 2 q + t / 4, v = t % 4), else the gathered features 8 g + t)
template <bool CHAIN, int NBLK, int KQ>
__device__ __forceinline__ float pair_src(const float* __restrict__ W, int ldw, int row_off, int e) {
  const int t = e & 7, lane = (e >> 3) & 63, rest = e >> 9;
  const int q = rest % KQ, mb = rest / KQ;
  const int i = lane & 15, g = lane >> 4;
  const int col = CHAIN ? (16 * (2 * q + (t >> 2)) + 4 * g + (t & 3)) : (8 * g + t);
  return W[(size_t)(row_off + 16 * mb + i) * ldw + col];
}
// element e of a region of N weights: hi image at halves [0, N), lo image at [N, 2 N).  -> does not fit
__device__ __forceinline__ bool pair_store(float* __restrict__ region, int N, int e, float w) {
  _Float16* d16 = reinterpret_cast<_Float16*>(region);
  const float ws = w * kPairW;
  const _Float16 h = (_Float16)ws;
  d16[e] = h;
  d16[N + e] = (_Float16)(ws - (float)h);
  return !(fabsf(ws) < kPairFit);
}

// One layer on a 16-sample tile: acc[mb] = (acc[mb] + kPairW W[row_off + 16 mb .. +15][:] . b) / kPairW, K = 32 KQ inputs
// (b in the layer's own order, see pair_src); acc arrives holding kPairW x the bias (staged so), which makes the scaled
// product sum accumulate in place: no second accumulator set in a kernel at 213 VGPRs.  The accumulation order (hi.hi,
// lo.hi, hi.lo per K step) is immaterial: every addition rounds at 2^-24 of the running sum, as in any fp32 product sum.
// W / ldw / row_off: the fp32 matrix in global memory for the tiles that do not fit.
template <bool CHAIN, int NBLK, int KQ>
__device__ __forceinline__ void mfma_layer_pairs(const float* __restrict__ wf, const float* __restrict__ W, int ldw,
                                                 int row_off, bool wbad, int lane, const float (&b)[8 * KQ],
                                                 f32x4 (&acc)[NBLK]) {
  float mx = 0.f;
#pragma unroll
  for (int k = 0; k < 8 * KQ; ++k) mx = fmaxf(mx, fabsf(b[k]));
  if (__builtin_expect(wbad || __builtin_amdgcn_ballot_w64(!(mx < kPairFit)) != 0ull, 0)) {  // wave-uniform, rare
    // (the fragment addresses depend on the lane only: without the opaque offset they are hoisted out of the TILE loop,
    // 2 VGPRs each, and the fast path spills)
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const int g = lane >> 4;
    const float* Wl = W + (size_t)(row_off + (lane & 15)) * ldw + opaque;
#pragma unroll
    for (int s = 0; s < 8 * KQ; ++s) {
      const int col = CHAIN ? (16 * (s >> 2) + 4 * g + (s & 3)) : (8 * g + s);
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(Wl[(size_t)16 * mb * ldw + col] * kPairW, b[s], acc[mb], 0, 0, 0);
    }
  } else {
    const f16x8* w8 = reinterpret_cast<const f16x8*>(wf);
    constexpr int LO = NBLK * KQ * 64;  // f16x8 units from the hi image to the lo image
#pragma unroll
    for (int q = 0; q < KQ; ++q) {
      f16x8 xh, xl;
#pragma unroll
      for (int k = 0; k < 8; k += 2) {
        const f32x2 v = {b[8 * q + k], b[8 * q + k + 1]};
        const f16x2 h = __builtin_convertvector(v, f16x2);
        const f16x2 l = __builtin_convertvector(v - __builtin_convertvector(h, f32x2), f16x2);  // (the remainder is exact)
        xh[k] = h[0], xh[k + 1] = h[1], xl[k] = l[0], xl[k + 1] = l[1];
      }
      // consecutive MFMAs go to different accumulators (a dependent one waits for its predecessor's passes)
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w8[(mb * KQ + q) * 64 + lane], xh, acc[mb], 0, 0, 0);
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w8[LO + (mb * KQ + q) * 64 + lane], xh, acc[mb], 0, 0, 0);
#pragma unroll
      for (int mb = 0; mb < NBLK; ++mb)
        acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w8[(mb * KQ + q) * 64 + lane], xl, acc[mb], 0, 0, 0);
    }
  }
#pragma unroll
  for (int mb = 0; mb < NBLK; ++mb) acc[mb] *= 1.f / kPairW;
}

// Stage all weights of the field into LDS (256-thread workgroup; caller barriers afterwards).  Every thread first
// ISSUES all of its global loads (one register each, ~60 in flight), then stores: one memory round trip for the whole
// 54 KB image instead of one per loop iteration.
template <int H, int SPLIT = 0>
__device__ __forceinline__ void stage_field_weights(const FieldDev& fd, float* __restrict__ lds) {
  using Ld = Lds<H, SPLIT>;
  constexpr int NB = H / 16;
  constexpr int T = 256;  // == blockDim.x
  constexpr int N_G0 = H * 32 / T, N_G1 = 32 * H / T, N_F0 = H * 32 / T, N_F1 = H * H / T, N_F2 = 32 * H / T,
                N_SH = 16 * H / T;
  static_assert((H * 32) % T == 0 && (H * H) % T == 0 && (16 * H) % T == 0, "regions are whole passes of the block");
  const int tid = threadIdx.x;
  float vg0[N_G0], vg1[N_G1], vf0[N_F0], vf1[N_F1], vf2[N_F2], vsh[N_SH], vs[7];
  if constexpr (SPLIT == 1) {
    stage_split_matrix<false, NB, 2>(lds + Ld::G0, fd.gw0, 32, 0);
    stage_split_matrix<true, 2, NB>(lds + Ld::G1, fd.gw1, H, 1);
    stage_split_matrix<true, NB, 2>(lds + Ld::F0, fd.fw0, 48, 0);
    stage_split_matrix<true, NB, NB>(lds + Ld::F1, fd.fw1, H, 0);
  } else if constexpr (SPLIT == 2) {
    static_assert(H % 32 == 0, "fp16 pairs: K = 32 steps");
    if (tid == 0) reinterpret_cast<int*>(lds)[Ld::FLG] = 0;
    __syncthreads();
#pragma unroll
    for (int it = 0; it < N_G0; ++it) vg0[it] = pair_src<false, NB, 1>(fd.gw0, 32, 0, it * T + tid);
#pragma unroll
    for (int it = 0; it < N_G1; ++it) vg1[it] = pair_src<true, 2, H / 32>(fd.gw1, H, 1, it * T + tid);
#pragma unroll
    for (int it = 0; it < N_F0; ++it) vf0[it] = pair_src<true, NB, 1>(fd.fw0, 48, 0, it * T + tid);
#pragma unroll
    for (int it = 0; it < N_F1; ++it) vf1[it] = pair_src<true, NB, H / 32>(fd.fw1, H, 0, it * T + tid);
  } else {
#pragma unroll
    for (int it = 0; it < N_G0; ++it) vg0[it] = frag_src<false, NB, 8>(fd.gw0, 32, 0, it * T + tid);
#pragma unroll
    for (int it = 0; it < N_G1; ++it) vg1[it] = frag_src<true, 2, H / 4>(fd.gw1, H, 1, it * T + tid);
#pragma unroll
    for (int it = 0; it < N_F0; ++it) vf0[it] = frag_src<true, NB, 8>(fd.fw0, 48, 0, it * T + tid);
#pragma unroll
    for (int it = 0; it < N_F1; ++it) vf1[it] = frag_src<true, NB, H / 4>(fd.fw1, H, 0, it * T + tid);
  }
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
  if constexpr (SPLIT == 2) {
    // the tile runs in units of kPairAct (see mfma_layer_pairs): biases in, the three exits out -- all powers of two
    // ... and the biases of the four pair layers x kPairW on top (mfma_layer_pairs accumulates onto them in place)
    constexpr float kB = kPairAct * kPairW;
    vs[0] *= 1.f / kPairAct, vs[1] *= kB, vs[2] *= kB, vs[3] *= kB;
    if (t33 != 0) vs[4] *= kB;
#pragma unroll
    for (int it = 0; it < N_F2; ++it) vf2[it] *= 1.f / kPairAct;
#pragma unroll
    for (int it = 0; it < N_SH; ++it) vsh[it] *= kB;
    bool bad = false;
#pragma unroll
    for (int it = 0; it < N_G0; ++it) bad |= pair_store(lds + Ld::G0, H * 32, it * T + tid, vg0[it]);
#pragma unroll
    for (int it = 0; it < N_G1; ++it) bad |= pair_store(lds + Ld::G1, 32 * H, it * T + tid, vg1[it]);
#pragma unroll
    for (int it = 0; it < N_F0; ++it) bad |= pair_store(lds + Ld::F0, H * 32, it * T + tid, vf0[it]);
#pragma unroll
    for (int it = 0; it < N_F1; ++it) bad |= pair_store(lds + Ld::F1, H * H, it * T + tid, vf1[it]);
    if (bad) reinterpret_cast<int*>(lds)[Ld::FLG] = 1;
  }
  if constexpr (SPLIT == 0) {
#pragma unroll
    for (int it = 0; it < N_G0; ++it) lds[Ld::G0 + it * T + tid] = vg0[it];
#pragma unroll
    for (int it = 0; it < N_G1; ++it) lds[Ld::G1 + it * T + tid] = vg1[it];
#pragma unroll
    for (int it = 0; it < N_F0; ++it) lds[Ld::F0 + it * T + tid] = vf0[it];
#pragma unroll
    for (int it = 0; it < N_F1; ++it) lds[Ld::F1 + it * T + tid] = vf1[it];
  }
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

// Streaming store of 16 bytes: the activations a training forward saves (1.2 KB per sample) are read again only by the
// backward, a whole kernel later -- marked non-temporal so that they do not evict the hash-table lines the gathers live on
// from the L2.
__device__ __forceinline__ void stream_store(float* p, f32x4 v) { __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p)); }

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

// ---- the three pipeline stages' state ------------------------------------------------------------------------------
template <int LPL, int F>
struct TileFetch {
  float fv[LPL][8][F];  // corner entries, in flight until the blend
  float x, y, z, std;   // contracted sample position / std: the trilinear offsets and the H4 weights are re-derived from
                        // them at blend time (3 VALU per level and axis) instead of holding 4*LPL more registers
  float t0, t1;
};
// An ACT tile carries, next to its fetch, one register: the slot (in the ray's candidate list) of the actor whose box
// contains the lane's sample, -1 = static scene.

// Range of the processing order one wave walks.  The ray-side pointers are separate `const __restrict__` kernel
// arguments: the per-ray constants are read at wave-uniform addresses, and only noalias/readonly arguments let the
// compiler turn those reads into scalar loads inside a loop that also stores.
struct RayRange {
  int64_t end;  // one past the last position of this wave's range
  int S, stride;
};

// The tile that will be issued NEXT iteration: its sample interval (vector loads, per lane) and ray constants (scalar
// loads) are requested one iteration ahead, so issuing its gathers never waits on a memory round trip.
struct PendingTile {
  int64_t pos;  // position in the processing order
  int64_t ray;  // ray index = order[pos] (or pos)
  int t;
  bool valid;
  float t0, t1;
  float ox, oy, oz, dx, dy, dz, area;
  int ncand;  // ACT: number of candidate actors of the ray (wave-uniform)
};

__device__ __forceinline__ void load_pending(PendingTile& p, int64_t pos, int t, const RayRange& rr, int j,
                                             const int32_t* __restrict__ order, const float* __restrict__ ro,
                                             const float* __restrict__ rd, const float* __restrict__ rarea,
                                             const float* __restrict__ rstarts, const float* __restrict__ rends,
                                             const int32_t* __restrict__ cand_count = nullptr) {
  // Past the end of this wave's range the loads still go out (clamped to the last position, tile 0) and their gathers
  // are issued and dropped: an `if (valid)` around them makes every fetched register a phi at the join, and the copies
  // the compiler puts there wait for ALL gathers before the MFMA phase -- exactly the stall the pipeline removes.
  p.pos = pos, p.t = t, p.valid = pos < rr.end;
  const int64_t pc = p.valid ? pos : rr.end - 1;
  const int64_t ray = order ? (int64_t)order[pc] : pc;
  p.ray = ray;
  const int s = p.valid ? 16 * t + j : j;
  const int64_t si = ray * rr.stride + (s < rr.S ? s : rr.S - 1);
  p.t0 = rstarts[si];
  p.t1 = rends[si];
  p.ox = ro[3 * ray], p.oy = ro[3 * ray + 1], p.oz = ro[3 * ray + 2];
  p.dx = rd[3 * ray], p.dy = rd[3 * ray + 1], p.dz = rd[3 * ray + 2];
  p.area = rarea[ray];
  p.ncand = cand_count ? cand_count[ray] : 0;
}

// H2 + H3 + the hash of H1 for one pending tile, then all gathers issued back to back (no waits in here beyond the
// pending tile's own small loads, which were issued a whole tile earlier).
// hash_corners with the level's own multipliers and mask (eval_layout.hip): the reference hash for the levels that keep it,
// ix | iy << s | iz << 2s for the shadow levels -- one code path; the offsets are re-derived at blend time as before
__device__ __forceinline__ void layout_corners(float x, float y, float z, float scale, uint32_t my, uint32_t mz,
                                               uint32_t mask, uint32_t (&idx)[8]) {
  const float sx = __fmul_rn(x, scale), sy = __fmul_rn(y, scale), sz = __fmul_rn(z, scale);
  const uint32_t ifx = (uint32_t)(int)floorf(sx), ify = (uint32_t)(int)floorf(sy), ifz = (uint32_t)(int)floorf(sz);
  const uint32_t icx = (uint32_t)(int)ceilf(sx), icy = (uint32_t)(int)ceilf(sy), icz = (uint32_t)(int)ceilf(sz);
  const uint32_t hyc = icy * my, hyf = ify * my, hzc = icz * mz, hzf = ifz * mz;
  idx[0] = (icx ^ hyc ^ hzc) & mask;
  idx[1] = (icx ^ hyf ^ hzc) & mask;
  idx[2] = (ifx ^ hyf ^ hzc) & mask;
  idx[3] = (ifx ^ hyc ^ hzc) & mask;
  idx[4] = (icx ^ hyc ^ hzf) & mask;
  idx[5] = (icx ^ hyf ^ hzf) & mask;
  idx[6] = (ifx ^ hyf ^ hzf) & mask;
  idx[7] = (ifx ^ hyc ^ hzf) & mask;
}

template <int L, int F, bool HALF, bool RELAY = false>
__device__ __forceinline__ void issue_tile(const FieldDev& fd, const PendingTile& pt, int g, uint32_t mask,
                                           const float* scal_lds, TileFetch<L / 4, F>& tf,
                                           const uint32_t* lay_lds = nullptr) {
  constexpr int LPL = L / 4;
  tf.t0 = pt.t0;
  tf.t1 = pt.t1;
  const SamplePos p = sample_position(pt.ox, pt.oy, pt.oz, pt.dx, pt.dy, pt.dz, pt.area, pt.t0, pt.t1, fd.scale);
  tf.x = p.x, tf.y = p.y, tf.z = p.z, tf.std = p.std;
#pragma unroll
  for (int q = 0; q < LPL; ++q) {
    if constexpr (RELAY) {
      const uint4 ly = *reinterpret_cast<const uint4*>(lay_lds + 4 * q);  // mulY, mulZ, mask, row0 of this lane's level
      uint32_t idx[8];
      layout_corners(p.x, p.y, p.z, scal_lds[q], ly.x, ly.y, ly.z, idx);
#pragma unroll
      for (int k = 0; k < 8; ++k) Entry<F, HALF>::load(fd.table, ly.w + idx[k], tf.fv[q][k]);
    } else {
      const Corners cs = hash_corners(p.x, p.y, p.z, scal_lds[q], mask);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        Entry<F, HALF>::load(fd.table, ((uint32_t)(LPL * g + q) << fd.grid.log2T) + cs.idx[k], tf.fv[q][k]);
    }
  }
}

// ACT: (1) the ray's candidate actors are walked -- a wave-uniform loop over scalar loads, empty for most rays -- and
// every lane notes the LAST candidate whose box contains its sample (ascending actor index: the highest wins,
// neurad_encoding.py:184-185 on CPU); (2) the static gathers go out for every lane exactly as above; (3) in the rare tile
// with a hit, the lanes inside a box re-issue their gathers against that actor's table, one wave-uniform pass per
// distinct winner, so the table base stays in SGPRs and the loads keep the SGPR-base + 32-bit-offset form.  Vector loads
// return in order: the later (actor) data is what the registers end up holding.  Nothing of (1)/(3) is live across the
// gathers on the common path except one register (the winning slot).
template <int L, int F, bool HALF>
__device__ __forceinline__ void issue_tile_actors(const FieldDev& fd, const ActorFieldDev& ad, const PendingTile& pt, int g,
                                                  uint32_t mask, const float* scal_lds, const float* ascal_lds,
                                                  const int32_t* __restrict__ cand_actor,
                                                  const float* __restrict__ cand_w2b, const float* __restrict__ bounds,
                                                  const void* const* __restrict__ tables, TileFetch<L / 4, F>& tf,
                                                  int& slot_out) {
  constexpr int LPL = L / 4;
  const int n = __builtin_amdgcn_readfirstlane(pt.ncand);
  // 32-bit row index, made scalar explicitly: a 64-bit multiply has no scalar form
  const uint32_t row = (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)pt.ray * (uint32_t)ad.K));
  int slot = -1;
  if (n > 0) {
    const SamplePos gs = sample_gaussian(pt.ox, pt.oy, pt.oz, pt.dx, pt.dy, pt.dz, pt.area, pt.t0, pt.t1);
    for (int c = 0; c < n; ++c) {
      const float* w = cand_w2b + (row + (uint32_t)c) * 12u;
      const int act = cand_actor[row + (uint32_t)c];
      const float bx = w[0] * gs.x + w[1] * gs.y + w[2] * gs.z + w[3];
      const float by = w[4] * gs.x + w[5] * gs.y + w[6] * gs.z + w[7];
      const float bz = w[8] * gs.x + w[9] * gs.y + w[10] * gs.z + w[11];
      if (fabsf(bx) < bounds[3 * act] && fabsf(by) < bounds[3 * act + 1] && fabsf(bz) < bounds[3 * act + 2]) slot = c;
    }
  }
  slot_out = slot;
  tf.t0 = pt.t0;
  tf.t1 = pt.t1;
  unsigned long long todo = __ballot(slot >= 0);
  const SamplePos gs = sample_gaussian(pt.ox, pt.oy, pt.oz, pt.dx, pt.dy, pt.dz, pt.area, pt.t0, pt.t1);
  const uint32_t amask = (1u << ad.log2T) - 1u;
  while (todo) {  // rare: the lanes inside a box, one pass per distinct winner
    const int c = __builtin_amdgcn_readlane(slot, (int)__builtin_ctzll(todo));
    const bool mine = slot == c;
    todo &= ~__ballot(mine);
    const float* w = cand_w2b + (row + (uint32_t)c) * 12u;
    const int act = cand_actor[row + (uint32_t)c];
    const void* tb = tables[act];  // scalar load: the table base stays in SGPRs
    if (mine) {
      const float bx = w[0] * gs.x + w[1] * gs.y + w[2] * gs.z + w[3];
      const float by = w[4] * gs.x + w[5] * gs.y + w[6] * gs.z + w[7];
      const float bz = w[8] * gs.x + w[9] * gs.y + w[10] * gs.z + w[11];
      const SamplePos p = contract_gaussian(bx, by, bz, gs.std, ad.scale);
      tf.x = p.x, tf.y = p.y, tf.z = p.z, tf.std = p.std;
#pragma unroll
      for (int q = 0; q < LPL; ++q) {
        // levels beyond the actor grid are the zero padding of F.pad (neurad_encoding.py:183): their fetch goes to the
        // last actor level and is weighted 0 at the blend
        const int lv = LPL * g + q, alv = lv < ad.La ? lv : ad.La - 1;
        const Corners cs = hash_corners(p.x, p.y, p.z, ascal_lds[alv], amask);
#pragma unroll
        for (int k = 0; k < 8; ++k) Entry<F, HALF>::load(tb, ((uint32_t)alv << ad.log2T) + cs.idx[k], tf.fv[q][k]);
      }
    }
  }
  // the static scene: the lanes outside every box, i.e. all lanes of nearly every tile.  Issued LAST so that whatever
  // bookkeeping the compiler attaches to the rare passes above (waits, copies) sits in front of these gathers, where
  // nothing is in flight, and not between them and the MLPs.
  if (slot < 0) {
    const SamplePos p = contract_gaussian(gs.x, gs.y, gs.z, gs.std, fd.scale);
    tf.x = p.x, tf.y = p.y, tf.z = p.z, tf.std = p.std;
#pragma unroll
    for (int q = 0; q < LPL; ++q) {
      const Corners cs = hash_corners(p.x, p.y, p.z, scal_lds[q], mask);
#pragma unroll
      for (int k = 0; k < 8; ++k)
        Entry<F, HALF>::load(fd.table, ((uint32_t)(LPL * g + q) << fd.grid.log2T) + cs.idx[k], tf.fv[q][k]);
    }
  }
}

// ACT, tile with a hit: a sample inside an actor box sees the view direction in the box frame (neurad_encoding.py:203-208),
// so the tile takes its SH inputs per sample.  -> this lane's B values of the four extra k-steps of feat layer 0: SH
// components 4g..4g+3 of its sample's direction.  One wave-uniform pass per distinct winning actor (scalar loads of w2b).
__device__ __forceinline__ void actor_tile_sh(int slot, uint32_t row, int g, float dx, float dy, float dz,
                                              const float* __restrict__ cand_w2b, float (&shb)[4]) {
  float bx = dx, by = dy, bz = dz;
  unsigned long long todo = __ballot(slot >= 0);
  while (todo) {
    const int c = __builtin_amdgcn_readlane(slot, (int)__builtin_ctzll(todo));
    const bool mine = slot == c;
    todo &= ~__ballot(mine);
    const float* w = cand_w2b + (row + (uint32_t)c) * 12u;
    const float ddx = w[0] * dx + w[1] * dy + w[2] * dz;
    const float ddy = w[4] * dx + w[5] * dy + w[6] * dz;
    const float ddz = w[8] * dx + w[9] * dy + w[10] * dz;
    const float nn = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz) + 1e-7f;  // neurad_encoding.py:207
    if (mine) bx = ddx / nn, by = ddy / nn, bz = ddz / nn;
  }
  float sh[16];
  sh4((bx + 1.f) / 2.f, (by + 1.f) / 2.f, (bz + 1.f) / 2.f, sh);
#pragma unroll
  for (int c = 0; c < 16; ++c)
    if ((c >> 2) == g) shb[c & 3] = sh[c];
}

template <int LPL, int F>
__device__ __forceinline__ void blend_tile_actors(const TileFetch<LPL, F>& tf, int slot, int g, int La,
                                                  const float* scal_lds, const float* ascal_lds, float (&feat)[8]) {
  const bool in_box = slot >= 0;
#pragma unroll
  for (int q = 0; q < LPL; ++q) {
    const int lv = LPL * g + q, alv = lv < La ? lv : La - 1;
    const float sc = in_box ? ascal_lds[alv] : scal_lds[q];
    Corners c;
    const float sx = __fmul_rn(tf.x, sc), sy = __fmul_rn(tf.y, sc), sz = __fmul_rn(tf.z, sc);
    c.ox = __fsub_rn(sx, floorf(sx)), c.oy = __fsub_rn(sy, floorf(sy)), c.oz = __fsub_rn(sz, floorf(sz));
    float v[F];
    lerp_corners<F>(c, tf.fv[q], v);
    const float rw = (in_box && lv >= La) ? 0.f : rescale_weight(sc, tf.std);
#pragma unroll
    for (int f = 0; f < F; ++f) feat[q * F + f] = v[f] * rw;
  }
}

template <int LPL, int F>
__device__ __forceinline__ void blend_tile(const TileFetch<LPL, F>& tf, const float* scal_lds, float (&feat)[8]) {
  static_assert(LPL * F == 8, "8 features per lane");
#pragma unroll
  for (int q = 0; q < LPL; ++q) {
    const float sc = scal_lds[q];
    Corners c;  // offsets rounded exactly as hash_corners does (encodings.py:431-446)
    const float sx = __fmul_rn(tf.x, sc), sy = __fmul_rn(tf.y, sc), sz = __fmul_rn(tf.z, sc);
    c.ox = __fsub_rn(sx, floorf(sx)), c.oy = __fsub_rn(sy, floorf(sy)), c.oz = __fsub_rn(sz, floorf(sz));
    float v[F];
    lerp_corners<F>(c, tf.fv[q], v);
    const float rw = rescale_weight(sc, tf.std);
#pragma unroll
    for (int f = 0; f < F; ++f) feat[q * F + f] = v[f] * rw;
  }
}

// L levels, F features/level (L*F == 32), H hidden width, HALF = fp16 table, COMPOSITE = fuse C1+C2, ACT = dynamic actors.
template <int L, int F, int H, bool HALF, bool COMPOSITE, bool ACT = false, int SPLIT = 0, bool RELAY = false,
          bool OVR = false>
__global__ __launch_bounds__(256, 2) void render_kernel(
    FieldDev fd, int64_t n_rays, int S, int stride, const int32_t* __restrict__ order, const float* __restrict__ ro,
    const float* __restrict__ rd, const float* __restrict__ rarea, const float* __restrict__ rstarts,
    const float* __restrict__ rends, float* __restrict__ out_feat, float* __restrict__ out_depth,
    float* __restrict__ out_acc, float* __restrict__ out_w, float* __restrict__ out_sdf, float* __restrict__ out_alpha,
    SaveDev sv, float stop_eps, const int32_t* __restrict__ range, ActorFieldDev ad,
    const int32_t* __restrict__ cand_count,
    const int32_t* __restrict__ cand_actor, const float* __restrict__ cand_w2b, const float* __restrict__ bounds,
    const void* const* __restrict__ tables) {
  static_assert(L * F == 32 && L % 4 == 0, "fused kernel needs L*F == 32, L % 4 == 0");
  static_assert(!ACT || COMPOSITE, "actors: composited eval kernel (static and actor tables share one storage type)");
  static_assert(H % 16 == 0 && H >= 16 && H <= 128, "hidden width");
  // SPLIT = 1 (3-way bf16): the composited static-scene kernel.  SPLIT = 2 (fp16 pairs): also the per-sample kernel of the
  // static scene -- the training forward: the tile runs in units of kPairAct, every store of an activation undoes it
  static_assert(!SPLIT || ((COMPOSITE || SPLIT == 2) && !ACT), "split matrix products: static-scene kernels");
  static_assert(!RELAY || (COMPOSITE && !ACT && !SPLIT), "eval-table layout: the composited static-scene kernel");
  // OVR (training forward of a scene with dynamic actors): samples inside an actor box take their encoding row and view
  // direction from the caller (the differentiable actor branch computed them for the few hit samples) instead of the
  // static lookup.  The three ACT-only pointer arguments carry the overrides: cand_count = ovr_row [N] (row index or -1),
  // cand_w2b = ovr_rows [P,32], bounds = ovr_dirs [P,3].
  static_assert(!OVR || (!COMPOSITE && !ACT && !SPLIT && !RELAY), "row overrides: the per-sample training forward");
  using Ld = Lds<H, SPLIT>;
  constexpr int NB = H / 16;
  constexpr int LPL = L / 4;         // levels per lane
  constexpr bool DEFER = COMPOSITE;  // last feature layer applied once per ray
  extern __shared__ __attribute__((aligned(16))) float lds[];

  // ---- stage weights (once per workgroup; the grid is persistent over rays) ----------------------
  stage_field_weights<H, SPLIT>(fd, lds);
  if constexpr (ACT || OVR) {
    for (int e = threadIdx.x; e < 16 * H; e += 256) lds[Ld::SHF + e] = frag_src<true, NB, 4>(fd.fw0 + 32, 48, 0, e);
  }
  if constexpr (ACT) {
    if (threadIdx.x < ad.La) lds[Ld::ASCAL + threadIdx.x] = ad.scal[threadIdx.x];
  }
  if constexpr (RELAY) {
    if (threadIdx.x < 4 * L) reinterpret_cast<uint32_t*>(lds + Ld::LAY)[threadIdx.x] = fd.lay[threadIdx.x];
  }
  __syncthreads();
  bool wbad = false;  // SPLIT = 2: some weight does not fit its fp16 pair -> every tile takes the fp32 products
  if constexpr (SPLIT == 2) wbad = __builtin_amdgcn_readfirstlane(reinterpret_cast<const int*>(lds)[Ld::FLG]) != 0;

  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const uint32_t mask = (1u << fd.grid.log2T) - 1u;
  const int ntile = (S + 15) >> 4;

  // XCD-coherent ranges of the processing order (see the header)
  const int nx = (int)gridDim.x < 8 ? (int)gridDim.x : 8;
  const int xcd = (int)blockIdx.x % nx, lb = (int)blockIdx.x / nx;
  const int nblk = ((int)gridDim.x - xcd + nx - 1) / nx;  // workgroups that share this range
  // `range` (device memory, optional): the slice [range[0], range[1]) of the processing order this launch renders -- a
  // scene with actors is rendered as two launches over one order, split on the device (nrhip_render_fwd_actors)
  const int64_t first = range ? range[0] : 0, count = range ? range[1] - range[0] : n_rays;
  const int64_t lo = first + count * xcd / nx;
  const RayRange rr{first + count * (xcd + 1) / nx, S, stride};
  const int64_t pos_step = (int64_t)nblk * 4;

  const float* scal_l = lds + Ld::SCAL + LPL * g;  // this lane's levels (re-read per tile: 1 ds_read, no VGPRs held)
  const uint32_t* lay_l = reinterpret_cast<const uint32_t*>(lds + Ld::LAY) + 4 * LPL * g;  // RELAY: their layout constants
  float* rbw = lds + Ld::RB + wid * H;              // this wave's per-ray bias row
  int64_t pos = lo + (int64_t)lb * 4 + wid;
  if (pos >= rr.end) return;
  int t = 0;
  // three-stage pipeline over this wave's flattened (ray, tile) sequence:
  //   tf = gathered corners of the CURRENT tile | q = next tile (interval + ray constants loaded, gathers not yet
  //   issued) | the tile after q has its small loads requested at the end of the issue step
  TileFetch<LPL, F> tf;
  int ta = -1;  // ACT: candidate slot of the actor containing this lane's sample of the tile in `tf` (-1: none)
  const float* ascal_l = lds + Ld::ASCAL;
  PendingTile q;
  load_pending(q, pos, 0, rr, j, order, ro, rd, rarea, rstarts, rends, cand_count);
  int64_t ray = q.ray;
  if constexpr (ACT) issue_tile_actors<L, F, HALF>(fd, ad, q, g, mask, scal_l, ascal_l, cand_actor, cand_w2b, bounds, tables, tf, ta);
  else issue_tile<L, F, HALF, RELAY>(fd, q, g, mask, scal_l, tf, lay_l);
  if constexpr (OVR) ta = (q.valid && j < S) ? cand_count[q.ray * S + j] : -1;  // override row of this lane's sample of `tf`
  {
    const bool wrap = ntile == 1;
    load_pending(q, wrap ? pos + pos_step : pos, wrap ? 0 : 1, rr, j, order, ro, rd, rarea, rstarts, rends, cand_count);
  }

  // per-ray state
  f32x4 shq = f32x4{0.f, 0.f, 0.f, 0.f};
  float carry = 0.f, acc_w = 0.f, acc_d = 0.f;
  constexpr int NHA = DEFER ? H / 4 : 1;
  float ha[NHA];  // DEFER: sum_s w_s * h2_s (this lane's sample column)
  f32x4 fa[2];    // DEFER: sum_s w_s * geo_embedding_s

  while (true) {
    if (t == 0) {
      // per-ray part of feat layer 0:  rb[n] = fb0[n] + sum_c fw0[n][32+c] * SH_c((d+1)/2)   (neurad_field.py:140-141)
      const float dx = rd[3 * ray], dy = rd[3 * ray + 1], dz = rd[3 * ray + 2];
      float sh[16];
      sh4((dx + 1.f) / 2.f, (dy + 1.f) / 2.f, (dz + 1.f) / 2.f, sh);
      if constexpr (!COMPOSITE) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
          if ((c >> 2) == g) shq[c & 3] = sh[c];
      }
      f32x4 rb[NB];
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) rb[mb] = *reinterpret_cast<const f32x4*>(lds + Ld::BF0 + 16 * mb + 4 * g);
#pragma unroll
      for (int c = 0; c < 16; ++c)
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) {
          const f32x4 w = *reinterpret_cast<const f32x4*>(lds + Ld::SHW + c * H + 16 * mb + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) rb[mb][r] = fmaf(w[r], sh[c], rb[mb][r]);
        }
      // parked in this wave's LDS row (the 16 lanes of a DPP row hold identical copies; every lane stores its own ->
      // same value to the same address) and re-read as the accumulator init of feat layer 0 in every tile
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) *reinterpret_cast<f32x4*>(rbw + 16 * mb + 4 * g) = rb[mb];
      carry = COMPOSITE ? (fd.use_sdf ? 1.f : 0.f) : 0.f;  // running transmittance (product) / optical depth (sum)
      acc_w = 0.f, acc_d = 0.f;
      fa[0] = fa[1] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < NHA; ++k) ha[k] = 0.f;
    }

    // The weights in LDS are loop invariant: without this opaque offset the compiler hoists ~200 LDS
    // loads out of the tile loop and spills them.  `lw` re-derives the LDS base once per tile.
    int opaque = 0;
    asm volatile("" : "+v"(opaque));
    const float* lw = lds + opaque;
    const int s = 16 * t + j;
    const bool live = s < S;
    const float t0 = tf.t0, t1 = tf.t1;

    // ---- blend the fetched corners (H1 + H4): the only wait on the gathers ---------------------------
    float feat[8];
    bool tile_hit = false;
    float shb[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (ACT) {
      blend_tile_actors<LPL, F>(tf, ta, g, ad.La, scal_l, ascal_l, feat);
      tile_hit = __ballot(ta >= 0) != 0ull;  // wave-uniform
      if (tile_hit)  // here, where few registers are live -- not in the middle of the MLPs
        actor_tile_sh(ta, (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)ray * (uint32_t)ad.K)), g, rd[3 * ray],
                      rd[3 * ray + 1], rd[3 * ray + 2], cand_w2b, shb);
    } else {
      blend_tile<LPL, F>(tf, scal_l, feat);
      if constexpr (SPLIT == 2) {
#pragma unroll
        for (int k = 0; k < 8; ++k) feat[k] *= kPairAct;
      }
      if constexpr (OVR) {
        tile_hit = __ballot(ta >= 0) != 0ull;  // wave-uniform; rare
        if (tile_hit) {
          float bx = rd[3 * ray], by = rd[3 * ray + 1], bz = rd[3 * ray + 2];
          if (ta >= 0) {
            const float* rp = cand_w2b + (size_t)ta * 32 + 8 * g;  // this lane's 8 columns of the override row
            const f32x4 r0 = *reinterpret_cast<const f32x4*>(rp), r1 = *reinterpret_cast<const f32x4*>(rp + 4);
#pragma unroll
            for (int k = 0; k < 4; ++k) feat[k] = r0[k], feat[4 + k] = r1[k];
            bx = bounds[3 * (size_t)ta], by = bounds[3 * (size_t)ta + 1], bz = bounds[3 * (size_t)ta + 2];
          }
          float sh[16];
          sh4((bx + 1.f) / 2.f, (by + 1.f) / 2.f, (bz + 1.f) / 2.f, sh);
#pragma unroll
          for (int c = 0; c < 16; ++c)
            if ((c >> 2) == g) shb[c & 3] = sh[c];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- early ray termination (eval option): the transmittance ENTERING this tile is already below stop_eps, so
    // everything behind it weighs less than stop_eps in total -> this (already fetched) tile is the ray's last.
    // The test lags one tile behind the carry so that the fetch below never has to be re-issued.
    bool stop_here = false;
    if constexpr (COMPOSITE) {
      if (stop_eps > 0.f) {
        const float c = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, carry)));
        stop_here = fd.use_sdf ? (c < stop_eps) : (c > -__logf(stop_eps));  // wave-uniform -> scalar branch
      }
    }
    const bool last_tile = t == ntile - 1;
    const bool ray_done = last_tile || stop_here;

    // ---- issue the next tile's gathers: they fly while this tile runs through the MLPs ---------------
    if (ray_done && q.valid && q.pos == pos)  // terminated early: q still points into this ray -> skip to the next ray
      load_pending(q, pos + pos_step, 0, rr, j, order, ro, rd, rarea, rstarts, rends, cand_count);  // (one exposed round trip)
    const bool have_next = q.valid;
    const int64_t npos = q.pos, nray = q.ray;
    const int nt = q.t;
    // unconditional (see load_pending)
    if constexpr (ACT) issue_tile_actors<L, F, HALF>(fd, ad, q, g, mask, scal_l, ascal_l, cand_actor, cand_w2b, bounds, tables, tf, ta);
    else issue_tile<L, F, HALF, RELAY>(fd, q, g, mask, scal_l, tf, lay_l);
    if constexpr (OVR) ta = (q.valid && 16 * q.t + j < S) ? cand_count[q.ray * S + 16 * q.t + j] : -1;
    {
      const bool wrap = nt + 1 == ntile;  // request the small loads of the tile after it
      load_pending(q, wrap ? npos + pos_step : npos, wrap ? 0 : nt + 1, rr, j, order, ro, rd, rarea, rstarts, rends,
                   cand_count);
    }
    __builtin_amdgcn_sched_barrier(0);

    bool saving = false;
    int64_t srow = 0;
    if constexpr (!COMPOSITE) {
      saving = sv.enc != nullptr && live;
      srow = ray * S + s;
      if (saving) {
        constexpr float u = SPLIT == 2 ? 1.f / kPairAct : 1.f;  // (tile units -> true units: a power of two, exact)
        float* ep = sv.enc + srow * 32 + 8 * g;
        stream_store(ep, f32x4{feat[0] * u, feat[1] * u, feat[2] * u, feat[3] * u});
        stream_store(ep + 4, f32x4{feat[4] * u, feat[5] * u, feat[6] * u, feat[7] * u});
      }
    }

    // ---- geo MLP layer 0 (32 -> H, ReLU) ---------------------------------------------------------
    f32x4 h[NB];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BG0 + 16 * mb + 4 * g);
    if constexpr (SPLIT == 2) mfma_layer_pairs<false, NB, 1>(lw + Ld::G0, fd.gw0, 32, 0, wbad, lane, feat, h);
    else if constexpr (SPLIT == 1) mfma_layer_split<NB, 2>(lw + Ld::G0, lane, feat, h);
    else mfma_layer<NB, 8>(lw + Ld::G0, lane, feat, h);
    float hb[H / 4];
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[4 * mb + r] = fmaxf(h[mb][r], 0.f);

    if constexpr (!COMPOSITE) {
      if (saving) {
        constexpr float u = SPLIT == 2 ? 1.f / kPairAct : 1.f;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
          stream_store(sv.hg + srow * H + 16 * mb + 4 * g,
                       f32x4{hb[4 * mb] * u, hb[4 * mb + 1] * u, hb[4 * mb + 2] * u, hb[4 * mb + 3] * u});
      }
    }

    // ---- geo MLP layer 1 (H -> 1 + 32): row 0 (sdf / raw density) on the VALU, rows 1..32 on MFMA --
    float sdf = 0.f;
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(lw + Ld::SDFW + 16 * mb + 4 * g);
#pragma unroll
      for (int r = 0; r < 4; ++r) sdf = fmaf(w[r], hb[4 * mb + r], sdf);
    }
    sdf += __shfl_xor(sdf, 16, 64);
    sdf += __shfl_xor(sdf, 32, 64);
    sdf += lw[Ld::BG1];
    f32x4 e[2];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
      const float* bp = lw + Ld::BG1 + 1 + 16 * mb + 4 * g;
      e[mb] = f32x4{bp[0], bp[1], bp[2], bp[3]};
    }
    if constexpr (SPLIT == 2) mfma_layer_pairs<true, 2, H / 32>(lw + Ld::G1, fd.gw1, H, 1, wbad, lane, hb, e);
    else if constexpr (SPLIT == 1) mfma_layer_split<2, NB>(lw + Ld::G1, lane, hb, e);
    else mfma_layer<2, H / 4>(lw + Ld::G1, lane, hb, e);
    float eb[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) eb[k] = e[k >> 2][k & 3];

    // ---- feature MLP (32 [+16 SH folded into the per-ray bias] -> H -> H -> 32), residual add ------
    if constexpr (!COMPOSITE) {
      if (saving) {
        constexpr float u = SPLIT == 2 ? 1.f / kPairAct : 1.f;
        float* xp = sv.xf + srow * 48;
        stream_store(xp + 4 * g, e[0] * u);
        stream_store(xp + 16 + 4 * g, e[1] * u);
        stream_store(xp + 32 + 4 * g, tile_hit ? f32x4{shb[0], shb[1], shb[2], shb[3]} : shq);
      }
    }
    if (!tile_hit) {
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::RB + wid * H + 16 * mb + 4 * g);
    } else {
      // per-sample SH inputs (actor_tile_sh): four more k-steps instead of the per-ray bias row
#pragma unroll
      for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BF0 + 16 * mb + 4 * g);
      mfma_layer<NB, 4>(lw + Ld::SHF, lane, shb, h);
    }
    if constexpr (SPLIT == 2) mfma_layer_pairs<true, NB, 1>(lw + Ld::F0, fd.fw0, 48, 0, wbad, lane, eb, h);
    else if constexpr (SPLIT == 1) mfma_layer_split<NB, 2>(lw + Ld::F0, lane, eb, h);
    else mfma_layer<NB, 8>(lw + Ld::F0, lane, eb, h);
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[4 * mb + r] = fmaxf(h[mb][r], 0.f);
    if constexpr (!COMPOSITE) {
      if (saving) {
        constexpr float u = SPLIT == 2 ? 1.f / kPairAct : 1.f;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
          stream_store(sv.hf + srow * (2 * H) + 16 * mb + 4 * g,
                       f32x4{hb[4 * mb] * u, hb[4 * mb + 1] * u, hb[4 * mb + 2] * u, hb[4 * mb + 3] * u});
      }
    }
#pragma unroll
    for (int mb = 0; mb < NB; ++mb) h[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BF1 + 16 * mb + 4 * g);
    if constexpr (SPLIT == 2) mfma_layer_pairs<true, NB, H / 32>(lw + Ld::F1, fd.fw1, H, 0, wbad, lane, hb, h);
    else if constexpr (SPLIT == 1) mfma_layer_split<NB, NB>(lw + Ld::F1, lane, hb, h);
    else mfma_layer<NB, H / 4>(lw + Ld::F1, lane, hb, h);
#pragma unroll
    for (int mb = 0; mb < NB; ++mb)
#pragma unroll
      for (int r = 0; r < 4; ++r) hb[4 * mb + r] = fmaxf(h[mb][r], 0.f);
    if constexpr (!COMPOSITE) {
      if (saving) {
        constexpr float u = SPLIT == 2 ? 1.f / kPairAct : 1.f;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb)
          stream_store(sv.hf + srow * (2 * H) + H + 16 * mb + 4 * g,
                       f32x4{hb[4 * mb] * u, hb[4 * mb + 1] * u, hb[4 * mb + 2] * u, hb[4 * mb + 3] * u});
      }
    }

    // ---- head (F4 / trunc_exp) -------------------------------------------------------------------
    float a_or_d;  // alpha (sdf mode) or density
    if (fd.use_sdf) a_or_d = __builtin_amdgcn_rcpf(1.f + __expf(sdf * fd.beta));  // sigmoid(-sdf*beta)
    else a_or_d = expf(sdf);

    if constexpr (!COMPOSITE) {
      // per-sample output: feature = geo_embedding + mlp_feature(...)   (neurad_field.py:141)
      f32x4 o[2];
#pragma unroll
      for (int mb = 0; mb < 2; ++mb) o[mb] = *reinterpret_cast<const f32x4*>(lw + Ld::BF2 + 16 * mb + 4 * g);
      mfma_layer<2, H / 4>(lw + Ld::F2, lane, hb, o);  // (SPLIT = 2: fw2 is staged x 1 / kPairAct -- true units)
      if constexpr (SPLIT == 2) {
        o[0] += e[0] * (1.f / kPairAct);
        o[1] += e[1] * (1.f / kPairAct);
      } else {
        o[0] += e[0];
        o[1] += e[1];
      }
      if (live) {
        float* fp = out_feat + (ray * S + s) * 32;
        stream_store(fp + 4 * g, o[0]);
        stream_store(fp + 16 + 4 * g, o[1]);
        if (g == 0) {
          out_sdf[ray * S + s] = sdf;
          out_alpha[ray * S + s] = a_or_d;
        }
      }
    } else {
      // ---- C1: transmittance scan over the 16 samples of the DPP row, carried across tiles -------
      float w, T;
      if (fd.use_sdf) {
        const float alpha = live ? a_or_d : 0.f;
        float incl = 1.f - alpha;
        incl *= row_shr<1>(incl, 1.f);
        incl *= row_shr<2>(incl, 1.f);
        incl *= row_shr<4>(incl, 1.f);
        incl *= row_shr<8>(incl, 1.f);
        const float excl = row_shr<1>(incl, 1.f);
        T = carry * excl;
        w = T * alpha;
        carry *= __shfl(incl, (lane & 48) | 15, 64);
      } else {
        const float sd = live ? a_or_d * (t1 - t0) : 0.f;
        float incl = sd;
        incl += row_shr<1>(incl, 0.f);
        incl += row_shr<2>(incl, 0.f);
        incl += row_shr<4>(incl, 0.f);
        incl += row_shr<8>(incl, 0.f);
        const float excl = row_shr<1>(incl, 0.f);
        T = expf(-(carry + excl));
        w = T * (1.f - expf(-sd));
        carry += __shfl(incl, (lane & 48) | 15, 64);
      }
      if (out_w && live && g == 0) out_w[ray * S + s] = w;
      // ---- C2 accumulation ------------------------------------------------------------------------
      acc_w += w;
      if (s < S - 1) acc_d += w * ((t0 + t1) / 2.f);
      float wf = w;  // weight of this sample's FEATURES: the sky residual 1 - sum w goes on sample S-1
      float acc = 0.f;
      if (ray_done) {
        acc = row_sum16(acc_w);
        if (last_tile && s == S - 1) wf += 1.f - acc;
      }
#pragma unroll
      for (int k = 0; k < H / 4; ++k) ha[k] = fmaf(hb[k], wf, ha[k]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        fa[0][r] = fmaf(e[0][r], wf, fa[0][r]);
        fa[1][r] = fmaf(e[1][r], wf, fa[1][r]);
      }

      if (ray_done) {
        const float dep = row_sum16(acc_d);
        if (!last_tile && out_w)  // terminated early: the rest of the ray contributes nothing
          for (int s2 = 16 * (t + 1) + lane; s2 < S; s2 += 64) out_w[ray * S + s2] = 0.f;
        // features = fw2 . (sum w h2) + fb2 * sum w' + sum w' e ;  sum w' = acc + (1 - acc) on a completed ray
        const float wsum = last_tile ? acc + (1.f - acc) : acc;
        f32x4 of2[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        mfma_layer<2, H / 4>(lw + Ld::F2, lane, ha, of2);
#pragma unroll
        for (int mb = 0; mb < 2; ++mb) {
          const f32x4 b2 = *reinterpret_cast<const f32x4*>(lw + Ld::BF2 + 16 * mb + 4 * g);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float es = SPLIT == 2 ? fa[mb][r] * (1.f / kPairAct) : fa[mb][r];  // (the embedding sum ran in tile units)
            of2[mb][r] = fmaf(b2[r], wsum, row_sum16(of2[mb][r] + es));
          }
        }
        if (j == 0) {
          float* fp = out_feat + ray * 32;
          *reinterpret_cast<f32x4*>(fp + 4 * g) = of2[0];
          *reinterpret_cast<f32x4*>(fp + 16 + 4 * g) = of2[1];
          if (g == 0) {
            out_acc[ray] = acc;
            out_depth[ray] = dep;
          }
        }
      }
    }

    if (!have_next) break;
    pos = npos, ray = nray, t = nt;
  }
}

static int validate_field(const nrhip_field* f) {
  NR_REQUIRE(f, NRHIP_ERR_INVALID_ARG, "field descriptor is NULL");
  if (int e = validate_grid(&f->grid)) return e;
  NR_REQUIRE(f->table && f->static_scale > 0.f, NRHIP_ERR_INVALID_ARG, "field: NULL table or non-positive scale");
  const nrhip_mlp& a = f->geo;
  const nrhip_mlp& b = f->feat;
  const int in = f->grid.num_levels * f->grid.n_features;
  NR_REQUIRE(in == 32 && f->grid.num_levels % 4 == 0, NRHIP_ERR_UNSUPPORTED,
             "fused field kernel needs L*F == 32 with L %% 4 == 0 (got L=%d F=%d); use the unfused ops",
             f->grid.num_levels, f->grid.n_features);
  NR_REQUIRE(a.num_layers == 2 && b.num_layers == 3 && a.in_dim == 32 && a.out_dim == 33 && b.in_dim == 48 &&
                 b.out_dim == 32 && a.hidden_dim == b.hidden_dim,
             NRHIP_ERR_UNSUPPORTED,
             "fused field kernel needs geo 32->H->33 (2 layers) and feat 48->H->H->32 (3 layers); use the unfused ops");
  NR_REQUIRE(a.hidden_dim == 32 || a.hidden_dim == 64, NRHIP_ERR_UNSUPPORTED,
             "fused field kernel is instantiated for hidden width 32 and 64 (got %d)", a.hidden_dim);
  for (int l = 0; l < 2; ++l) NR_REQUIRE(a.weight[l], NRHIP_ERR_INVALID_ARG, "geo weight %d is NULL", l);
  for (int l = 0; l < 3; ++l) NR_REQUIRE(b.weight[l], NRHIP_ERR_INVALID_ARG, "feat weight %d is NULL", l);
  return NRHIP_OK;
}

static FieldDev to_dev(const nrhip_field& f) {
  FieldDev d;
  d.grid = to_dev(f.grid);
  d.table = f.table;
  d.scale = f.static_scale;
  d.gw0 = f.geo.weight[0], d.gb0 = f.geo.bias[0];
  d.gw1 = f.geo.weight[1], d.gb1 = f.geo.bias[1];
  d.fw0 = f.feat.weight[0], d.fb0 = f.feat.bias[0];
  d.fw1 = f.feat.weight[1], d.fb1 = f.feat.bias[1];
  d.fw2 = f.feat.weight[2], d.fb2 = f.feat.bias[2];
  d.use_sdf = f.use_sdf;
  d.beta = f.beta;
  for (int i = 0; i < NRHIP_MAX_LEVELS * 4; ++i)
    d.lay[i] = (f.eval_table && f.eval_layout && i < 4 * f.grid.num_levels) ? f.eval_layout[i] : 0u;
  return d;
}

// the candidate lists + actor tables an ACT launch reads (all device pointers)
struct ActorLaunch {
  const int32_t* range;  // [2] device: slice of the processing order (NULL = all rays)
  ActorFieldDev ad;
  const int32_t* cand_count;
  const int32_t* cand_actor;
  const float* cand_w2b;
  const float* bounds;
  const void* const* tables;
};

template <int L, int F, int H, bool HALF, bool COMPOSITE, bool ACT = false, int SPLIT = 0, bool RELAY = false,
          bool OVR = false>
static int launch_render(const FieldDev& fd, const RaysDev& rd, float* of, float* od, float* oa, float* ow, float* os,
                         float* oal, const SaveDev& sv, float stop_eps, hipStream_t st,
                         const ActorLaunch& al = ActorLaunch()) {
  constexpr size_t lds = ((ACT || OVR) ? Lds<H, SPLIT>::TOTAL_ACT : Lds<H, SPLIT>::TOTAL) * sizeof(float);
  auto kern = render_kernel<L, F, H, HALF, COMPOSITE, ACT, SPLIT, RELAY, OVR>;
  static int cap = 0;  // persistent grid: CUs x resident workgroups per CU, queried once per instantiation
  if (!cap) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int dev = 0, n_cu = 256, nb = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n_cu = p.multiProcessorCount;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)kern, 256, lds) != hipSuccess || nb < 1) nb = 2;
    cap = n_cu * (nb > 4 ? 4 : nb);
  }
  const int64_t want = (rd.R + 3) / 4;
  const int blocks = (int)(want < cap ? want : cap);
  kern<<<blocks, 256, lds, st>>>(fd, rd.R, rd.S, rd.stride, rd.order, rd.o, rd.d, rd.area, rd.starts, rd.ends, of, od, oa,
                                 ow, os, oal, sv, stop_eps, al.range, al.ad, al.cand_count, al.cand_actor, al.cand_w2b,
                                 al.bounds, al.tables);
  return check_launch("render/field fused kernel");
}

template <bool COMPOSITE>
static int dispatch_render(const nrhip_field* f, const nrhip_rays* rays, float* of, float* od, float* oa, float* ow,
                           float* os, float* oal, void* stream, const SaveDev& sv = SaveDev{}, float stop_eps = 0.f,
                           const int32_t* range = nullptr) {
  ActorLaunch al = ActorLaunch();
  al.range = range;
  const FieldDev fd = to_dev(*f);
  const RaysDev rd = to_dev(*rays);
  const hipStream_t st = (hipStream_t)stream;
  const int L = f->grid.num_levels, F = f->grid.n_features, H = f->geo.hidden_dim;
  const bool half = f->grid.param_dtype == 1;
  // NRHIP_MLP_SPLIT_BF16=1 (64-wide MLPs, composited output): the per-tile matrix products run as 3-way
  // split bf16 on the matrix cores (mfma_layer_split) instead of the fp32 MFMA.  Same results to fp32 accuracy; NOT the
  // default because it measured no faster (163 vs 163 us on config[1]: the splitting costs the vector ALU what the matrix
  // pipe saves -- DESIGN.md §9).
  const bool split_bf16 = tuning().mlp_split_bf16;
  // NRHIP_MLP_PAIRS (default 1): the same products as fp16 pairs (mfma_layer_pairs; composited output)
  const bool pairs = tuning().mlp_pairs >= 0 ? tuning().mlp_pairs == 1 : kPairsDefault;
  if constexpr (COMPOSITE) {
    // eval table given (nrhip_field.eval_table / eval_layout): the coarse levels are read from their shadow copies.  Decided
    // FIRST: a caller that hands the table over asked for this kernel; it exists with fp32-MFMA products only (the
    // static_assert above: the relayout and the split / pair products do not combine), so it wins over the pair default.
    if (f->eval_table && f->eval_layout) {
      FieldDev fe = fd;
      fe.table = f->eval_table;
#define RCASE(L_, F_, H_)                                                                                              \
  if (L == L_ && F == F_ && H == H_)                                                                                   \
    return half ? launch_render<L_, F_, H_, true, true, false, false, true>(fe, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al)  \
                : launch_render<L_, F_, H_, false, true, false, false, true>(fe, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al);
      RCASE(16, 2, 64)
      RCASE(8, 4, 32)
      RCASE(8, 4, 64)
#undef RCASE
    }
  }
  if constexpr (COMPOSITE) {
    if (H == 64 && pairs && !split_bf16) {
#define PCASE(L_, F_)                                                                                                  \
  if (L == L_ && F == F_)                                                                                              \
    return half ? launch_render<L_, F_, 64, true, true, false, 2>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al)      \
                : launch_render<L_, F_, 64, false, true, false, 2>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al);
      PCASE(16, 2)
      PCASE(8, 4)
      PCASE(4, 8)
#undef PCASE
    }
    // ... and at NeuRAD's own width, where the arithmetic mostly hid under the memory time already (c2's render stage
    // 72 -> 60-67 us)
    if (H == 32 && pairs && !split_bf16) {
#define PCASE(L_, F_)                                                                                                  \
  if (L == L_ && F == F_)                                                                                              \
    return half ? launch_render<L_, F_, 32, true, true, false, 2>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al)      \
                : launch_render<L_, F_, 32, false, true, false, 2>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al);
      PCASE(16, 2)
      PCASE(8, 4)
      PCASE(4, 8)
#undef PCASE
    }
  }
  if constexpr (!COMPOSITE) {
    // the per-sample kernel (field forward / training forward with saved activations): the same pair products, OPT-IN
    // (NRHIP_MLP_PAIRS_TRAIN=1).  Round 6 measured no gain: the kernel stores 516 B of activations per sample beside its
    // gathers and is bound by them -- 1.03-1.09 ms with pairs against 1.01-1.03 ms with the fp32 MFMA on the c3 batch,
    // the step unchanged (profiles/r06_ab_pairs_train.txt)
    if (pairs && tuning().mlp_pairs_train && !split_bf16) {
#define TCASE(L_, F_, H_)                                                                                              \
  if (L == L_ && F == F_ && H == H_)                                                                                   \
    return half ? launch_render<L_, F_, H_, true, false, false, 2>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al)     \
                : launch_render<L_, F_, H_, false, false, false, 2>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al);
      TCASE(8, 4, 32)
      TCASE(16, 2, 64)
      TCASE(8, 4, 64)
      TCASE(16, 2, 32)
#undef TCASE
    }
  }
  if constexpr (COMPOSITE) {
    if (H == 64 && split_bf16) {
#define SCASE(L_, F_)                                                                                                  \
  if (L == L_ && F == F_)                                                                                              \
    return half ? launch_render<L_, F_, 64, true, true, false, 1>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al)      \
                : launch_render<L_, F_, 64, false, true, false, 1>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al);
      SCASE(16, 2)
      SCASE(8, 4)
      SCASE(4, 8)
#undef SCASE
    }
  }
#define CASE(L_, F_, H_)                                                                                          \
  if (L == L_ && F == F_ && H == H_) {                                                                            \
    return half ? launch_render<L_, F_, H_, true, COMPOSITE>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al)   \
                : launch_render<L_, F_, H_, false, COMPOSITE>(fd, rd, of, od, oa, ow, os, oal, sv, stop_eps, st, al); \
  }
  CASE(16, 2, 64)
  CASE(16, 2, 32)
  CASE(8, 4, 32)
  CASE(8, 4, 64)
  CASE(4, 8, 32)
  CASE(4, 8, 64)
#undef CASE
  set_error("fused field kernel: no instantiation for L=%d F=%d H=%d", L, F, H);
  return NRHIP_ERR_UNSUPPORTED;
}

// Stable partition of the processing order into [rays no actor can touch | rays with candidate actors] + the two slices
// as device-side ranges {0, n0, n0, R}.  One workgroup: per-thread chunk counts -> block scan -> ordered writes, so each
// half keeps the locality order it was given.
constexpr int kPartThreads = 1024;
__global__ __launch_bounds__(kPartThreads) void actor_partition_kernel(const int32_t* __restrict__ cand_count,
                                                                       const int32_t* __restrict__ order_in, int64_t n,
                                                                       int32_t* __restrict__ order_out,
                                                                       int32_t* __restrict__ ranges) {
  __shared__ uint32_t wave_tot[kPartThreads / 64];
  const int tid = threadIdx.x;
  const int64_t per = (n + kPartThreads - 1) / kPartThreads, b = per * tid, e = b + per < n ? b + per : n;
  uint32_t free_rays = 0;
  for (int64_t i = b; i < e; ++i) free_rays += cand_count[order_in ? order_in[i] : i] == 0 ? 1u : 0u;
  uint32_t incl = free_rays;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t up = __shfl_up(incl, off, 64);
    if ((tid & 63) >= off) incl += up;
  }
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  uint32_t before = incl - free_rays, total = 0;
  for (int w = 0; w < kPartThreads / 64; ++w) {
    if (w < (tid >> 6)) before += wave_tot[w];
    total += wave_tot[w];
  }
  // position i of the input is preceded by `before` free rays and (b - before) actor rays
  int64_t pf = before, pa = (int64_t)total + (b < n ? b : n) - before;
  for (int64_t i = b; i < e; ++i) {
    const int32_t r = order_in ? order_in[i] : (int32_t)i;
    if (cand_count[r] == 0) order_out[pf++] = r;
    else order_out[pa++] = r;
  }
  if (tid == 0) ranges[0] = 0, ranges[1] = (int32_t)total, ranges[2] = (int32_t)total, ranges[3] = (int32_t)n;
}

// composited eval with dynamic actors: the static shapes NeuRAD uses with actors, fp32 or fp16 tables
static int dispatch_render_actors(const nrhip_field* f, const nrhip_rays* rays, const ActorLaunch& al, float* of, float* od,
                                  float* oa, float* ow, float stop_eps, void* stream) {
  const FieldDev fd = to_dev(*f);
  const RaysDev rd = to_dev(*rays);
  const hipStream_t st = (hipStream_t)stream;
  const int L = f->grid.num_levels, F = f->grid.n_features, H = f->geo.hidden_dim;
  const bool half = f->grid.param_dtype == 1;
#define CASE(L_, F_, H_)                                                                                                     \
  if (L == L_ && F == F_ && H == H_)                                                                                         \
    return half ? launch_render<L_, F_, H_, true, true, true>(fd, rd, of, od, oa, ow, nullptr, nullptr, SaveDev{}, stop_eps, \
                                                              st, al)                                                        \
                : launch_render<L_, F_, H_, false, true, true>(fd, rd, of, od, oa, ow, nullptr, nullptr, SaveDev{},          \
                                                               stop_eps, st, al);
  CASE(8, 4, 32)
  CASE(8, 4, 64)
  CASE(16, 2, 64)
#undef CASE
  set_error("fused field kernel with actors: no instantiation for L=%d F=%d H=%d", L, F, H);
  return NRHIP_ERR_UNSUPPORTED;
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_field_fwd(const nrhip_field* f, const nrhip_rays* rays, float* feature, float* sdf, float* alpha,
                               void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0 || rays->n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(feature && sdf && alpha, NRHIP_ERR_INVALID_ARG, "field_fwd: NULL output");
  return dispatch_render<false>(f, rays, feature, nullptr, nullptr, nullptr, sdf, alpha, stream);
}

extern "C" int nrhip_field_fwd_train(const nrhip_field* f, const nrhip_rays* rays, float* feature, float* sdf,
                                     float* alpha, float* save_enc, float* save_geo_hidden, float* save_feat_in,
                                     float* save_feat_hidden, void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0 || rays->n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(feature && sdf && alpha && save_enc && save_geo_hidden && save_feat_in && save_feat_hidden,
             NRHIP_ERR_INVALID_ARG, "field_fwd_train: NULL output");
  NR_REQUIRE(((reinterpret_cast<uintptr_t>(save_enc) | reinterpret_cast<uintptr_t>(save_geo_hidden) |
               reinterpret_cast<uintptr_t>(save_feat_in) | reinterpret_cast<uintptr_t>(save_feat_hidden)) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "field_fwd_train: save buffers must be 16-byte aligned");
  const SaveDev sv{save_enc, save_geo_hidden, save_feat_in, save_feat_hidden};
  return dispatch_render<false>(f, rays, feature, nullptr, nullptr, nullptr, sdf, alpha, stream, sv);
}

extern "C" int nrhip_field_fwd_train_ovr(const nrhip_field* f, const nrhip_rays* rays, const int32_t* ovr_row,
                                         const float* ovr_rows, const float* ovr_dirs, float* feature, float* sdf,
                                         float* alpha, float* save_enc, float* save_geo_hidden, float* save_feat_in,
                                         float* save_feat_hidden, void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0 || rays->n_samples == 0) return NRHIP_OK;
  NR_REQUIRE(feature && sdf && alpha && save_enc && save_geo_hidden && save_feat_in && save_feat_hidden && ovr_row &&
                 ovr_rows && ovr_dirs,
             NRHIP_ERR_INVALID_ARG, "field_fwd_train_ovr: NULL pointer");
  NR_REQUIRE(((reinterpret_cast<uintptr_t>(save_enc) | reinterpret_cast<uintptr_t>(save_geo_hidden) |
               reinterpret_cast<uintptr_t>(save_feat_in) | reinterpret_cast<uintptr_t>(save_feat_hidden) |
               reinterpret_cast<uintptr_t>(ovr_rows)) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "field_fwd_train_ovr: save buffers and override rows must be 16-byte aligned");
  NR_REQUIRE(rays->n_rays * rays->n_samples < (INT64_C(1) << 31), NRHIP_ERR_UNSUPPORTED, "field_fwd_train_ovr: N >= 2^31");
  const SaveDev sv{save_enc, save_geo_hidden, save_feat_in, save_feat_hidden};
  ActorLaunch al = ActorLaunch();
  al.cand_count = ovr_row, al.cand_w2b = ovr_rows, al.bounds = ovr_dirs;
  const FieldDev fd = to_dev(*f);
  const RaysDev rd = to_dev(*rays);
  const int L = f->grid.num_levels, F = f->grid.n_features, H = f->geo.hidden_dim;
  NR_REQUIRE(f->grid.param_dtype == 0 || f->grid.param_dtype == 1, NRHIP_ERR_INVALID_ARG, "field_fwd_train_ovr: dtype");
  const bool half = f->grid.param_dtype == 1;
#define OCASE(L_, F_, H_)                                                                                                   \
  if (L == L_ && F == F_ && H == H_)                                                                                        \
    return half ? launch_render<L_, F_, H_, true, false, false, false, false, true>(fd, rd, feature, nullptr, nullptr, nullptr, \
                                                                                    sdf, alpha, sv, 0.f, (hipStream_t)stream, al) \
                : launch_render<L_, F_, H_, false, false, false, false, false, true>(fd, rd, feature, nullptr, nullptr,     \
                                                                                     nullptr, sdf, alpha, sv, 0.f,          \
                                                                                     (hipStream_t)stream, al);
  OCASE(8, 4, 32)
  OCASE(8, 4, 64)
  OCASE(16, 2, 64)
#undef OCASE
  set_error("field_fwd_train_ovr: no instantiation for L=%d F=%d H=%d", L, F, H);
  return NRHIP_ERR_UNSUPPORTED;
}

extern "C" int nrhip_render_fwd_ex(const nrhip_field* f, const nrhip_rays* rays, float* out_features, float* out_depth,
                                   float* out_acc, float* out_weights, float early_stop_eps, void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  if (rays->n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(out_features && out_depth && out_acc, NRHIP_ERR_INVALID_ARG, "render_fwd: NULL output");
  NR_REQUIRE(rays->n_samples >= 1, NRHIP_ERR_INVALID_ARG, "render_fwd: needs >= 1 sample per ray");
  NR_REQUIRE(early_stop_eps >= 0.f && early_stop_eps < 1.f, NRHIP_ERR_INVALID_ARG,
             "render_fwd: early_stop_eps %g not in [0,1)", (double)early_stop_eps);
  return dispatch_render<true>(f, rays, out_features, out_depth, out_acc, out_weights, nullptr, nullptr, stream,
                               SaveDev{}, early_stop_eps);
}

extern "C" int nrhip_render_fwd(const nrhip_field* f, const nrhip_rays* rays, float* out_features, float* out_depth,
                                float* out_acc, float* out_weights, void* stream) {
  return nrhip_render_fwd_ex(f, rays, out_features, out_depth, out_acc, out_weights, 0.f, stream);
}

extern "C" int nrhip_render_fwd_actors(const nrhip_field* f, const nrhip_actors* a, const nrhip_rays* rays,
                                       const int32_t* cand_count, const int32_t* cand_actor, const float* cand_w2b,
                                       float* out_features, float* out_depth, float* out_acc, float* out_weights,
                                       float early_stop_eps, int32_t* workspace, void* stream) {
  if (int e = validate_field(f)) return e;
  if (int e = validate_rays(rays)) return e;
  NR_REQUIRE(a, NRHIP_ERR_INVALID_ARG, "render_fwd_actors: actors descriptor is NULL");
  if (int e = validate_grid(&a->grid)) return e;
  if (rays->n_rays == 0) return NRHIP_OK;
  NR_REQUIRE(out_features && out_depth && out_acc && cand_count && cand_actor && cand_w2b && workspace && a->bounds &&
                 a->tables && a->n_actors >= 1 && a->actor_scale > 0.f,
             NRHIP_ERR_INVALID_ARG, "render_fwd_actors: NULL pointer / empty actor set");
  NR_REQUIRE(rays->n_samples >= 1, NRHIP_ERR_INVALID_ARG, "render_fwd_actors: needs >= 1 sample per ray");
  NR_REQUIRE(early_stop_eps >= 0.f && early_stop_eps < 1.f, NRHIP_ERR_INVALID_ARG,
             "render_fwd_actors: early_stop_eps %g not in [0,1)", (double)early_stop_eps);
  // per-sample table select keeps the gather shape: same features per level, no more levels than the static grid
  NR_REQUIRE(f->grid.param_dtype == a->grid.param_dtype, NRHIP_ERR_UNSUPPORTED,
             "render_fwd_actors: the static and the actor tables must share one storage type (fp32 or fp16)");
  NR_REQUIRE(a->grid.n_features == f->grid.n_features && a->grid.num_levels <= f->grid.num_levels, NRHIP_ERR_UNSUPPORTED,
             "render_fwd_actors: actor grid (L=%d F=%d) must have the static grid's features per level (F=%d) and at most "
             "its levels (L=%d); use the unfused ops",
             a->grid.num_levels, a->grid.n_features, f->grid.n_features, f->grid.num_levels);
  ActorLaunch al;
  al.ad.K = a->max_candidates > 0 ? a->max_candidates : NRHIP_DEFAULT_ACTOR_CANDIDATES;
  al.ad.La = a->grid.num_levels, al.ad.log2T = a->grid.log2_table_size, al.ad.scale = a->actor_scale;
  for (int l = 0; l < NRHIP_MAX_LEVELS; ++l) al.ad.scal[l] = a->grid.scalings[l];
  al.cand_count = cand_count, al.cand_actor = cand_actor, al.cand_w2b = cand_w2b, al.bounds = a->bounds, al.tables = a->tables;
  // Most rays of a street scene pass no actor at all.  The processing order is split on the device into those rays and
  // the rays with candidates (no host round trip): the first slice runs through the plain static kernel at full speed,
  // the second through the ACT instantiation, which pays for the candidate walk in registers and latency.
  int32_t* order2 = workspace;
  int32_t* ranges = workspace + rays->n_rays;
  actor_partition_kernel<<<1, kPartThreads, 0, (hipStream_t)stream>>>(cand_count, rays->order, rays->n_rays, order2, ranges);
  if (int e = check_launch("actor_partition")) return e;
  nrhip_rays split = *rays;
  split.order = order2;
  if (int e = dispatch_render<true>(f, &split, out_features, out_depth, out_acc, out_weights, nullptr, nullptr, stream,
                                    SaveDev{}, early_stop_eps, ranges))
    return e;
  al.range = ranges + 2;
  return dispatch_render_actors(f, &split, al, out_features, out_depth, out_acc, out_weights, early_stop_eps, stream);
}
