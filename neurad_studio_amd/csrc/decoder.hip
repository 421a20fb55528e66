// RGB CNN decoder: rendered [patch, 48] feature rows -> 3x upsampled rgb patch
// (models/neurad.py:198-216,359-366; model_components/cnns.py:20-46).  The reference trains it under fp16 autocast
// (configs/method_configs.py: mixed_precision=True): convolutions take fp16 operands and accumulate in fp32, BatchNorm keeps
// fp32 statistics.  Same arithmetic here, on the matrix cores:
//   * activations live in HBM as NHWC fp16 -- the rendered feature rows ARE [pixel][channel] already, so the reference's
//     permute to NCHW never happens;
//   * conv7x7 (32 -> 32 channels, the 8 convolutions that are > 99 % of the decoder's FLOPs) is an implicit GEMM on
//     v_mfma_f32_32x32x16_f16: M = 32 pixels of one image row, N = 32 output channels, K = 32 input channels per tap.  A
//     workgroup stages its input tile (with the 3-pixel halo) in LDS ONCE; a wave owns R output rows and walks the taps
//     column by column: the seven B fragments of a tap column (weights, pre-packed in fragment order, L1/L2 resident) are
//     held in registers, the A fragments (one input row at one horizontal shift) are read from LDS once and serve the up to
//     R (output row, ky) pairs that touch them.  0.6 LDS/L1 reads of 16 B per MFMA;
//     Measured at 40 x 96 x 96 (scripts/bench_decoder_kernels.py, profiles/r03_decoder_conv7x7.json): 44 us = 840 TFLOP/s
//     (MIOpen's implicit GEMM: 233 us); with the tile load switched off 34 us, with the stores off 36 us, with both off
//     28.5 us (1.3 PFLOP/s): the two WGs of a CU run in lockstep, so the load and store phases are not hidden yet;
//   * the input gradient of a convolution is the same kernel on flipped / transposed weights (pack mode 1).
#include "common.h"

namespace nrhip {
namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kC = 32;                // channels of the decoder's hidden layers
constexpr int kTW = 32;               // tile width = the MFMA's 32 rows
constexpr int kCols = kTW + 6;        // + halo
constexpr int kPix = 64;              // bytes per pixel in LDS, no padding: the four 16-byte channel groups of tile column c sit
                                      // at (group ^ ((c >> 2) & 3)) -- ds_read_b128 of 32 consecutive columns is conflict-free,
                                      // and a 22 x 38 tile is 53.5 KB

union Frag {
  uint4 u;
  half8 h;
};

// weights [out][in][7][7] fp32 (torch Conv2d) -> B fragments in the order the kernel reads them:
// wfrag[kx][ky][h][lane] = 8 halves B[k = 16 h + 8 (lane >> 5) + e][j = lane & 31] of tap (ky, kx).
// mode 0: forward, B[k][j] = w[j][k][ky][kx].  mode 1: input gradient, B[k][j] = w[k][j][6 - ky][6 - kx].
__global__ void conv7_pack_kernel(const float* __restrict__ w, int mode, uint4* __restrict__ wfrag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 49 * 2 * 64) return;
  const int lane = t & 63, h = (t >> 6) & 1, tap = t >> 7;
  const int kx = tap / 7, ky = tap - kx * 7;
  const int j = lane & 31, k0 = 16 * h + 8 * (lane >> 5);
  Frag f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    const float v = mode == 0 ? w[((j * kC + k) * 7 + ky) * 7 + kx] : w[((k * kC + j) * 7 + (6 - ky)) * 7 + (6 - kx)];
    f.h[e] = (_Float16)v;
  }
  wfrag[t] = f.u;
}

struct PackMany {
  const float* w[8];
};
// all convolutions of the decoder in one launch: wfrag[(i * 2 + mode)] for i < n, mode in {0, 1}
__global__ void conv7_pack_many_kernel(PackMany src, int n, uint4* __restrict__ wfrag) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int which = blockIdx.y;  // i * 2 + mode
  if (t >= 49 * 2 * 64 || which >= 2 * n) return;
  const float* w = src.w[which >> 1];
  const int mode = which & 1;
  const int lane = t & 63, h = (t >> 6) & 1, tap = t >> 7;
  const int kx = tap / 7, ky = tap - kx * 7;
  const int j = lane & 31, k0 = 16 * h + 8 * (lane >> 5);
  Frag f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    const float v = mode == 0 ? w[((j * kC + k) * 7 + ky) * 7 + kx] : w[((k * kC + j) * 7 + (6 - ky)) * 7 + (6 - kx)];
    f.h[e] = (_Float16)v;
  }
  wfrag[(size_t)which * (49 * 2 * 64) + t] = f.u;
}

template <int R, bool STATS>
__global__ __launch_bounds__(256) void conv7_kernel(const _Float16* __restrict__ in, const uint4* __restrict__ wfrag,
                                                    const float* __restrict__ bias, _Float16* __restrict__ out,
                                                    float* __restrict__ partial, int H, int W) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  constexpr int TH = 4 * R, ROWS = TH + 6;
  const int tiles_x = (W + kTW - 1) / kTW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, b = blockIdx.y;
  const int x0 = tx * kTW, y0 = ty * TH;
  const _Float16* img = in + (size_t)b * H * W * kC;
  {  // all of this thread's 16-byte pieces of the tile are requested before the first one is written to LDS
    constexpr int NCH = ROWS * kCols * 4, NIT = (NCH + 255) / 256;
    uint4 v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = threadIdx.x + it * 256;
      const int q = c & 3, p = c >> 2;
      const int row = p / kCols, col = p - row * kCols;
      const int y = y0 - 3 + row, x = x0 - 3 + col;
      v[it] = uint4{0u, 0u, 0u, 0u};
      if (c < NCH && y >= 0 && y < H && x >= 0 && x < W)
        v[it] = *(const uint4*)(img + ((size_t)y * W + x) * kC + q * 8);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int c = threadIdx.x + it * 256;
      const int p = c >> 2, col = p % kCols;
      if (c < NCH) *(uint4*)(lds + p * kPix + (((c & 3) ^ ((col >> 2) & 3)) << 4)) = v[it];
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int px = lane & 31, kb = lane >> 5;
  f32x16 acc[R];
  {
    const float bj = bias ? bias[px] : 0.f;
#pragma unroll
    for (int y = 0; y < R; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[y][r] = bj;
  }
  const unsigned char* abase = lds + ((wave * R) * kCols + px) * kPix;  // + the swizzled channel group, per tap column
  // Software pipeline, pinned with scheduling barriers (left alone, the compiler sinks every load to its first use and the
  // wave then waits an L2 round trip per four MFMAs): a tap's two weight fragments are re-requested for the NEXT tap column
  // as soon as this column's MFMAs on them are issued (6 steps = 12 R MFMAs before their use), and the one input row a step
  // needs is read from LDS one step (2 R MFMAs) ahead.
  Frag Bf[7][2];
  Frag A[R + 6][2];
  // channel groups kb (k = 0..15) and 2 + kb (k = 16..31) of tile column px + kx, at their swizzled places
  auto col_base = [&](int kx, const unsigned char*& c0, const unsigned char*& c1) {
    const int g = kb ^ (((px + kx) >> 2) & 3);
    c0 = abase + kx * kPix + (g << 4);
    c1 = abase + kx * kPix + ((g ^ 2) << 4);
  };
#pragma unroll
  for (int ky = 0; ky < 7; ++ky) {
    Bf[ky][0].u = wfrag[(ky * 2 + 0) * 64 + lane];
    Bf[ky][1].u = wfrag[(ky * 2 + 1) * 64 + lane];
  }
  const unsigned char *a0, *a1;
  col_base(0, a0, a1);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    A[r][0].u = *(const uint4*)(a0 + (r * kCols) * kPix);
    A[r][1].u = *(const uint4*)(a1 + (r * kCols) * kPix);
  }
#pragma unroll
  for (int kx = 0; kx < 7; ++kx) {
    const unsigned char *n0, *n1;
    col_base(kx + 1 < 7 ? kx + 1 : kx, n0, n1);
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      if (ky < 6) {  // the row step ky + 1 adds
        A[ky + R][0].u = *(const uint4*)(a0 + ((ky + R) * kCols) * kPix);
        A[ky + R][1].u = *(const uint4*)(a1 + ((ky + R) * kCols) * kPix);
      } else if (kx + 1 < 7) {  // the next column's first R rows (rows 0 .. R-1 of this column are dead by now)
#pragma unroll
        for (int r = 0; r < R; ++r) {
          A[r][0].u = *(const uint4*)(n0 + (r * kCols) * kPix);
          A[r][1].u = *(const uint4*)(n1 + (r * kCols) * kPix);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // the R accumulators take turns: no MFMA waits for the one before it
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int y = 0; y < R; ++y)
          acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[y + ky][h].h, Bf[ky][h].h, acc[y], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (kx + 1 < 7) {  // this tap's registers are free: the next column's tap ky is requested 6 steps before its use
        const uint4* wp = wfrag + (size_t)((kx + 1) * 7 + ky) * 2 * 64 + lane;
        Bf[ky][0].u = wp[0];
        Bf[ky][1].u = wp[64];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    a0 = n0;
    a1 = n1;
  }
  // Two channels per store: lanes j and j ^ 1 (same pixels, neighbouring channels) swap one value per register pair, the
  // even lane then holds channels (j, j + 1) of the pair's first pixel, the odd lane (j - 1, j) of its second.
  float s1 = 0.f, s2 = 0.f;
  _Float16* oimg = out + (size_t)b * H * W * kC;
  const bool odd = lane & 1;
#pragma unroll
  for (int y = 0; y < R; ++y) {
    const int yy = y0 + wave * R + y;
    if (yy >= H) continue;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const _Float16 h0 = (_Float16)acc[y][r], h1 = (_Float16)acc[y][r + 1];
      if (STATS) {  // BatchNorm sees the rounded activation
        const int xa = x0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
        const float v0 = xa < W ? (float)h0 : 0.f, v1 = xa + 1 < W ? (float)h1 : 0.f;
        s1 += v0 + v1;
        s2 += v0 * v0 + v1 * v1;
      }
      union {
        _Float16 h[2];
        uint32_t u;
      } pk;
      union {
        _Float16 h;
        unsigned short s;
      } snd, rcv;
      snd.h = odd ? h0 : h1;
      rcv.s = (unsigned short)__shfl_xor((int)snd.s, 1, 64);
      pk.h[0] = odd ? rcv.h : h0;
      pk.h[1] = odd ? h1 : rcv.h;
      const int x = x0 + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2) + 4 * kb;
      if (x < W) *(uint32_t*)(oimg + ((size_t)yy * W + x) * kC + (px & ~1)) = pk.u;
    }
  }
  if (STATS) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    __syncthreads();  // every wave is done with the input tile
    float* red = (float*)lds;
    if (kb == 0) {
      red[wave * 64 + px] = s1;
      red[wave * 64 + 32 + px] = s2;
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int t = threadIdx.x;
      partial[((size_t)b * gridDim.x + blockIdx.x) * 64 + t] = (red[t] + red[64 + t]) + (red[128 + t] + red[192 + t]);
    }
  }
}

template <int R>
int launch_conv7(const void* in, const void* wfrag, const float* bias, void* out, float* partial, int B, int H, int W,
                 hipStream_t st) {
  constexpr int TH = 4 * R;
  const size_t smem = (size_t)(TH + 6) * kCols * kPix;
  const dim3 grid(((W + kTW - 1) / kTW) * ((H + TH - 1) / TH), B);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv7_kernel<R, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute((const void*)conv7_kernel<R, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (partial)
    hipLaunchKernelGGL((conv7_kernel<R, true>), grid, dim3(256), smem, st, (const _Float16*)in, (const uint4*)wfrag, bias,
                       (_Float16*)out, partial, H, W);
  else
    hipLaunchKernelGGL((conv7_kernel<R, false>), grid, dim3(256), smem, st, (const _Float16*)in, (const uint4*)wfrag, bias,
                       (_Float16*)out, partial, H, W);
  return check_launch("conv7x7");
}


// ---- weight gradient ------------------------------------------------------------------------------------------------
// dW[tap][ci][co] = sum over pixels p of X[p + tap - 3][ci] * G[p][co]: a GEMM that contracts over PIXELS, while NHWC keeps
// channels contiguous.  The rows are therefore turned channel-major on their way into LDS -- by the matrix core itself: an
// MFMA against the identity returns its A operand in the C layout, i.e. with pixel and channel swapped between lanes and
// registers (exact: fp16 x 1 accumulated in fp32).  A workgroup owns a strip of image rows; wave ky (7 waves) accumulates
// the seven taps (ky, 0..6) in 112 accumulator registers.  Per 16 pixels of a row it reads ONE G fragment and two aligned
// X blocks; the seven horizontal shifts are funnel shifts of those two blocks (v_alignbit, the even ones are renames).
// LDS: a ring of 8 transposed X rows (rows y .. y+6 are live) and 2 G rows; one barrier per image row.  Partial sums per
// workgroup are reduced by a second kernel in a fixed order (no atomics).
constexpr int kWgradWaves = 7;

__device__ __forceinline__ int pitch16(int halves) {  // row pitch in halves: a multiple of 8 whose 16-byte count is odd
  int n = (halves + 7) / 8;
  if ((n & 1) == 0) ++n;
  return n * 8;
}

// one 32-pixel block of an NHWC row -> channel-major LDS row, in two halves so that the global load can be issued a whole
// row of MFMAs ahead of its use: block_load (lane = pixel: two 16-byte loads), block_store (identity MFMA, pack, 4 x 8 B)
struct BlockRegs {
  Frag a0, a1;
};
__device__ __forceinline__ BlockRegs block_load(const _Float16* __restrict__ row /* may be null: zeros */, int x_first,
                                                int W, int lane) {
  const int i = lane & 31, kb = lane >> 5;
  const int x = x_first + i;
  BlockRegs r;
  r.a0.u = r.a1.u = uint4{0u, 0u, 0u, 0u};
  if (row && x >= 0 && x < W) {
    const uint4* p = (const uint4*)(row + (size_t)x * kC);
    r.a0.u = p[kb];      // channels 8 kb .. 8 kb + 7
    r.a1.u = p[2 + kb];  // channels 16 + 8 kb ..
  }
  return r;
}
__device__ __forceinline__ float block_store(const BlockRegs& in, _Float16* dst /* [32][pitch], at the block's first column */,
                                             int pitch, int lane, const Frag (&ident)[2]) {
  const int i = lane & 31, kb = lane >> 5;
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(in.a0.h, ident[0].h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(in.a1.h, ident[1].h, c, 0, 0, 0);
  // lane: channel i, pixels (r & 3) + 8 (r >> 2) + 4 kb
  float s = 0.f;
  _Float16* d = dst + i * pitch + 4 * kb;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    union {
      _Float16 h[4];
      uint2 u;
    } v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v.h[e] = (_Float16)c[4 * g + e];
      s += c[4 * g + e];
    }
    *(uint2*)(d + 8 * g) = v.u;
  }
  return s;
}

__global__ __launch_bounds__(kWgradWaves * 64) void conv7_wgrad_kernel(const _Float16* __restrict__ x,
                                                                      const _Float16* __restrict__ g,
                                                                      float* __restrict__ partial,
                                                                      float* __restrict__ partial_db, int H, int W,
                                                                      int rows_per_strip, int strips_per_image) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  // 16-pixel K blocks per row; the X fragment of block xb reads columns 16 xb + 8 kb .. + 22, G's 16 xb + 8 kb .. + 7
  const int nkb = (W + 15) / 16, nxb = (16 * nkb + 8 + 31) / 32, ngb = (16 * nkb + 31) / 32;
  const int pitch_x = pitch16(32 * nxb), pitch_g = pitch16(32 * ngb);
  _Float16* XT = (_Float16*)lds;                 // [8][32][pitch_x]
  _Float16* GT = XT + 8 * 32 * pitch_x;          // [2][32][pitch_g]
  const int b = blockIdx.x / strips_per_image;
  const int ya = (blockIdx.x - b * strips_per_image) * rows_per_strip;
  const int yb = min(H, ya + rows_per_strip);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ch = lane & 31, kb = lane >> 5;
  const _Float16* ximg = x + (size_t)b * H * W * kC;
  const _Float16* gimg = g + (size_t)b * H * W * kC;
  Frag ident[2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int e = 0; e < 8; ++e) ident[h].h[e] = (ch == 16 * h + 8 * kb + e) ? (_Float16)1.f : (_Float16)0.f;
  float db = 0.f;
  // task t of a row set: X row q (halo coordinates: image row q - 3) block t, then G row block t - nxb
  auto load_x = [&](int q, int t) {
    const int yin = q - 3;
    return block_load((yin >= 0 && yin < H) ? ximg + (size_t)yin * W * kC : nullptr, 32 * t - 3, W, lane);
  };
  auto store_x = [&](const BlockRegs& r, int q, int t) {
    block_store(r, XT + (size_t)(q & 7) * 32 * pitch_x + 32 * t, pitch_x, lane, ident);
  };
  auto load_g = [&](int y, int t) {
    return block_load((y >= 0 && y < H) ? gimg + (size_t)y * W * kC : nullptr, 32 * t, W, lane);
  };
  auto store_g = [&](const BlockRegs& r, int y, int t) {
    db += block_store(r, GT + (size_t)(y & 1) * 32 * pitch_g + 32 * t, pitch_g, lane, ident);
  };
  if (ya < yb) {
    const int n_pro = 7 * nxb + ngb;
    for (int t = wave; t < n_pro; t += kWgradWaves) {
      if (t < 7 * nxb)
        store_x(load_x(ya + t / nxb, t % nxb), ya + t / nxb, t % nxb);
      else
        store_g(load_g(ya, t - 7 * nxb), ya, t - 7 * nxb);
    }
  }
  __syncthreads();
  f32x16 acc[7];
#pragma unroll
  for (int k = 0; k < 7; ++k)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;
  const int n_tasks = nxb + ngb;
  for (int y = ya; y < yb; ++y) {
    // next iteration's rows (X row q = y + 7, G row y + 1) go to slots nobody reads now.  This wave's first task is loaded
    // BEFORE the row's MFMAs and stored behind them: its global latency hides under 6 x 7 MFMAs
    const bool more = y + 1 < yb;
    BlockRegs pre;
    if (more && wave < n_tasks) pre = wave < nxb ? load_x(y + 7, wave) : load_g(y + 1, wave - nxb);
    const int yin = y + wave - 3;  // this wave's X row for G row y
    if (yin >= 0 && yin < H) {
      const _Float16* xt = XT + ((size_t)((y + wave) & 7) * 32 + ch) * pitch_x + 8 * kb;
      const _Float16* gt = GT + ((size_t)(y & 1) * 32 + ch) * pitch_g + 8 * kb;
      for (int xb = 0; xb < nkb; ++xb) {
        Frag gf;
        gf.u = *(const uint4*)(gt + 16 * xb);
        const uint4 lo = *(const uint4*)(xt + 16 * xb), hi = *(const uint4*)(xt + 16 * xb + 8);
        const uint32_t S[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) {
          Frag a;
          uint32_t v[4];
#pragma unroll
          for (int d = 0; d < 4; ++d)
            v[d] = (kx & 1) ? __builtin_amdgcn_alignbit(S[d + (kx + 1) / 2], S[d + (kx - 1) / 2], 16) : S[d + kx / 2];
          a.u = uint4{v[0], v[1], v[2], v[3]};
          acc[kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, gf.h, acc[kx], 0, 0, 0);
        }
      }
    }
    if (more) {
      if (wave < n_tasks) {
        if (wave < nxb)
          store_x(pre, y + 7, wave);
        else
          store_g(pre, y + 1, wave - nxb);
      }
      for (int t = wave + kWgradWaves; t < n_tasks; t += kWgradWaves) {  // wide images: the rest, unprefetched
        if (t < nxb)
          store_x(load_x(y + 7, t), y + 7, t);
        else
          store_g(load_g(y + 1, t - nxb), y + 1, t - nxb);
      }
    }
    __syncthreads();
  }
  // C layout: row (ci) = (r & 3) + 8 (r >> 2) + 4 kb, column (co) = lane & 31
  float* pw = partial + ((size_t)blockIdx.x * 49 + wave * 7) * 1024;
#pragma unroll
  for (int kx = 0; kx < 7; ++kx)
#pragma unroll
    for (int r = 0; r < 16; ++r) pw[(size_t)kx * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + ch] = acc[kx][r];
  // bias gradient: every G block was transposed by exactly one wave
  db += __shfl_xor(db, 32, 64);
  float* red = (float*)lds;
  if (kb == 0) red[wave * 32 + ch] = db;
  __syncthreads();
  if (threadIdx.x < 32) {
    float s = 0.f;
#pragma unroll
    for (int w2 = 0; w2 < kWgradWaves; ++w2) s += red[w2 * 32 + threadIdx.x];
    partial_db[(size_t)blockIdx.x * 32 + threadIdx.x] = s;
  }
}

// partial [n][49][ci][co] -> grad_weight [co][ci][7][7] (+=), partial_db [n][32] -> grad_bias (+=); fixed order.
// 256 threads = 32 outputs x 8 slices of the n partials (independent load chains), combined through LDS.
__global__ __launch_bounds__(256) void conv7_wgrad_reduce_kernel(const float* __restrict__ partial,
                                                                 const float* __restrict__ partial_db, int n,
                                                                 float* __restrict__ grad_weight,
                                                                 float* __restrict__ grad_bias,
                                                                 const float* __restrict__ gscale) {
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const bool is_bias = blockIdx.x == 49 * 32;
  if (is_bias && !grad_bias) return;
  const float* src = is_bias ? partial_db + o : partial + (size_t)blockIdx.x * 32 + o;
  const size_t stride = is_bias ? 32 : 49 * 1024;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = sl;
  for (; k + 24 < n; k += 32) {
    s0 += src[(size_t)k * stride];
    s1 += src[(size_t)(k + 8) * stride];
    s2 += src[(size_t)(k + 16) * stride];
    s3 += src[(size_t)(k + 24) * stride];
  }
  for (; k < n; k += 8) s0 += src[(size_t)k * stride];
  red[sl][o] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (sl == 0) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][o];
    if (gscale) s *= gscale[1];
    if (is_bias) {
      grad_bias[o] += s;
    } else {
      const int t = blockIdx.x * 32 + o;  // (tap, ci, co)
      const int co = t & 31, ci = (t >> 5) & 31, tap = t >> 10;
      grad_weight[(co * kC + ci) * 49 + tap] += s;
    }
  }
}

inline int host_pitch16(int halves) {
  int n = (halves + 7) / 8;
  if ((n & 1) == 0) ++n;
  return n * 8;
}
// every image is cut into strips of equal height: at least 16 rows (the 6 halo rows and the 200 KB of partial sums per strip
// are per-strip costs), more when that still gives more than one strip per CU
inline void wgrad_plan(int b, int h, int* rows_per_strip, int* strips_per_image) {
  int rps = (int)(((int64_t)b * h + 255) / 256);
  if (rps < 16) rps = 16;
  if (rps > h) rps = h;
  *rows_per_strip = rps;
  *strips_per_image = (h + rps - 1) / rps;
}


// ---- the decoder's other layers --------------------------------------------------------------------------------------
// Elementwise kernels on NHWC-32 fp16 tensors: a thread owns 8 channels of one pixel (16 bytes).  Reductions over pixels
// write per-workgroup partial sums that a second kernel adds in a fixed order (bit-reproducible, no atomics).
__device__ __forceinline__ void load8(const _Float16* p, float (&f)[8]) {
  Frag t;
  t.u = *(const uint4*)p;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = (float)t.h[e];
}
__device__ __forceinline__ void store8(_Float16* p, const float (&f)[8]) {
  Frag t;
#pragma unroll
  for (int e = 0; e < 8; ++e) t.h[e] = (_Float16)f[e];
  *(uint4*)p = t.u;
}
__device__ __forceinline__ float round_h(float v) { return (float)(_Float16)v; }

// sum of column (threadIdx.x & 63) of partial [n][64] over all n rows, valid in threads 0..63; 1024 threads
__device__ __forceinline__ double colsum64(const float* __restrict__ partial, int n, double* red /*[16][64]*/) {
  const int col = threadIdx.x & 63, sl = threadIdx.x >> 6;
  double a = 0.0, b = 0.0;
  int k = sl;
  for (; k + 16 < n; k += 32) {
    a += (double)partial[(size_t)k * 64 + col];
    b += (double)partial[(size_t)(k + 16) * 64 + col];
  }
  if (k < n) a += (double)partial[(size_t)k * 64 + col];
  red[sl * 64 + col] = a + b;
  __syncthreads();
  double s = 0.0;
  if (threadIdx.x < 64)
#pragma unroll
    for (int q = 0; q < 16; ++q) s += red[q * 64 + threadIdx.x];
  return s;
}

// BatchNorm2d in training mode (cnns.py:40,43; torch.nn.functional.batch_norm): batch statistics from the convolution's
// per-workgroup sums, coefficients for the apply pass, running statistics updated like torch (unbiased variance).
// coef [4][32] = scale, shift, mean, rstd
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const float* __restrict__ partial, int n, double count,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ coef) {
  __shared__ double red[16 * 64];
  __shared__ double tot[64];
  const double s = colsum64(partial, n, red);
  if (threadIdx.x < 64) tot[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int c = threadIdx.x;
    const double mean = tot[c] / count;
    double var = tot[32 + c] / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    coef[c] = sc;
    coef[32 + c] = beta[c] - (float)mean * sc;
    coef[64 + c] = (float)mean;
    coef[96 + c] = rstd;
    if (running_mean) {
      running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
      running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * count / (count > 1.0 ? count - 1.0 : 1.0));
    }
  }
}

// out = relu(bn(c) [+ skip]) with the roundings of the reference's fp16 chain: BatchNorm's output is rounded to fp16, the
// residual sum is an fp16 add (cnns.py:31, 38-44)
__global__ __launch_bounds__(256) void bn_act_kernel(const _Float16* __restrict__ c, const float* __restrict__ coef,
                                                    const _Float16* __restrict__ skip, _Float16* __restrict__ out,
                                                    int64_t n8) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n8) return;
  const int q = (int)(idx & 3);
  float v[8], k[8];
  load8(c + idx * 8, v);
  if (skip) load8(skip + idx * 8, k);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float y = round_h(fmaf(v[e], coef[8 * q + e], coef[32 + 8 * q + e]));
    if (skip) y = round_h(y + k[e]);
    v[e] = fmaxf(y, 0.f);
  }
  store8(out + idx * 8, v);
}

// backward of relu(bn(c) [+ skip]): g = dout * (act > 0); per channel sum g and sum g * c  -> partial [blocks][64]
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const _Float16* __restrict__ dout,
                                                           const _Float16* __restrict__ act,
                                                           const _Float16* __restrict__ c, int64_t n8,
                                                           float* __restrict__ partial) {
  __shared__ float red[256][17];
  float sg[8], sgc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) sg[e] = sgc[e] = 0.f;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < n8; idx += (int64_t)gridDim.x * 256) {
    float d[8], a[8], v[8];
    load8(dout + idx * 8, d);
    load8(act + idx * 8, a);
    load8(c + idx * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float g = a[e] > 0.f ? d[e] : 0.f;
      sg[e] += g;
      sgc[e] = fmaf(g, v[e], sgc[e]);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[threadIdx.x][e] = sg[e];
    red[threadIdx.x][8 + e] = sgc[e];
  }
  __syncthreads();
  if (threadIdx.x < 64) {  // the stride of the loop is a multiple of 4: thread t always owns channel octet t & 3
    const int q = threadIdx.x >> 4, j = threadIdx.x & 15;
    float s = 0.f;
    for (int k = 0; k < 64; ++k) s += red[4 * k + q][j];
    partial[(size_t)blockIdx.x * 64 + (j < 8 ? 0 : 32) + 8 * q + (j & 7)] = s;
  }
}

// -> bcoef [3][32] with dc = A g + B c + C (the batch-norm backward as an affine map per channel), dgamma, dbeta (+=)
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ partial, int n, double count,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ coef, float* __restrict__ bcoef,
                                                              float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                              const float* __restrict__ gscale, int eval_mode) {
  __shared__ double red[16 * 64];
  __shared__ double tot[64];
  const double s = colsum64(partial, n, red);
  if (threadIdx.x < 64) tot[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < 32) {
    const int c = threadIdx.x;
    const double mean = coef[64 + c], rstd = coef[96 + c];
    const double sum_g = tot[c], sum_gc = tot[32 + c];
    const double dg = rstd * (sum_gc - mean * sum_g);  // sum g * xhat
    const double A = (double)gamma[c] * rstd;
    // eval mode: mean and rstd are the running statistics, constants of the step -- the gradient is A g alone
    const double B = eval_mode ? 0.0 : -A * rstd * dg / count;
    bcoef[c] = (float)A;
    bcoef[32 + c] = (float)B;
    bcoef[64 + c] = eval_mode ? 0.f : (float)(-A * sum_g / count - B * mean);
    const double inv = gscale ? (double)gscale[1] : 1.0;
    if (dgamma) dgamma[c] += (float)(dg * inv);
    if (dbeta) dbeta[c] += (float)(sum_g * inv);
  }
}

// mode 0: out = A g + B c + C (gradient of the convolution's output); mode 1: out = a + g (the block's input gradient:
// convolution path + skip path); g = dout * (act > 0)
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const _Float16* __restrict__ dout,
                                                          const _Float16* __restrict__ act,
                                                          const _Float16* __restrict__ c /* mode 1: a */,
                                                          const float* __restrict__ bcoef, int mode,
                                                          _Float16* __restrict__ out, int64_t n8) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n8) return;
  const int q = (int)(idx & 3);
  float d[8], a[8], v[8];
  load8(dout + idx * 8, d);
  load8(act + idx * 8, a);
  load8(c + idx * 8, v);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const float g = a[e] > 0.f ? d[e] : 0.f;
    v[e] = mode == 0 ? fmaf(bcoef[8 * q + e], g, fmaf(bcoef[32 + 8 * q + e], v[e], bcoef[64 + 8 * q + e])) : v[e] + g;
  }
  store8(out + idx * 8, v);
}

// the first `cols` columns of partial [n][k] -> out [cols] (= or +=), fixed order; 256 threads = 32 columns x 8 slices.
// Columns >= cols0 go to out1 [cols - cols0] (a layer's bias gradient behind its weight gradient: one launch for both).
__global__ __launch_bounds__(256) void reduce_cols_kernel(const float* __restrict__ partial, int n, int k, int cols,
                                                         float* __restrict__ out, int accumulate,
                                                         const float* __restrict__ gscale, int cols0 = 0x7fffffff,
                                                         float* __restrict__ out1 = nullptr) {
  __shared__ float red[8][32];
  const int o = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + o;
  float s0 = 0.f, s1 = 0.f;
  if (col < cols) {
    int r = sl;
    for (; r + 8 < n; r += 16) {
      s0 += partial[(size_t)r * k + col];
      s1 += partial[(size_t)(r + 8) * k + col];
    }
    if (r < n) s0 += partial[(size_t)r * k + col];
  }
  red[sl][o] = s0 + s1;
  __syncthreads();
  if (sl == 0 && col < cols) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) s += red[q][o];
    if (gscale) s *= gscale[1];
    float* dst = col < cols0 ? out + col : out1 + (col - cols0);
    *dst = accumulate ? *dst + s : s;
  }
}

// first layer: Conv2d(cin, 32, 1) + ReLU on the rendered feature rows (models/neurad.py:201-203): features [n][cin] fp32 ->
// h [n][32] fp16.  fp16 operands, fp32 accumulation.  256 threads = 64 pixels x 4 channel octets.
constexpr int kMaxCin = 64;
__global__ __launch_bounds__(256) void conv1x1_in_fwd_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                                            const float* __restrict__ bias, _Float16* __restrict__ out,
                                                            int64_t n, int cin) {
  __shared__ float W[kMaxCin][32];       // [i][o]
  __shared__ float X[64][kMaxCin + 1];
  for (int t = threadIdx.x; t < 32 * cin; t += 256) {
    const int o = t / cin, i = t - o * cin;
    W[i][o] = round_h(w[t]);
  }
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  for (int t = threadIdx.x; t < 64 * cin; t += 256) {
    const int p = t / cin, i = t - p * cin;
    X[p][i] = p0 + p < n ? round_h(feat[(p0 + p) * cin + i]) : 0.f;
  }
  __syncthreads();
  const int p = threadIdx.x >> 2, q = threadIdx.x & 3;
  if (p0 + p >= n) return;
  float acc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = bias[8 * q + e];
  for (int i = 0; i < cin; ++i) {
    const float x = X[p][i];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = fmaf(x, W[i][8 * q + e], acc[e]);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) acc[e] = fmaxf(acc[e], 0.f);
  store8(out + ((p0 + p) * 32 + 8 * q), acc);
}

// its backward: g = dh * (h > 0); dfeat [n][cin] fp32 = g W; partial [blocks][32 * cin + 32] = (g^T feat, sum g)
__global__ __launch_bounds__(256) void conv1x1_in_bwd_kernel(const float* __restrict__ feat, const _Float16* __restrict__ h,
                                                            const _Float16* __restrict__ dh, const float* __restrict__ w,
                                                            float* __restrict__ dfeat, float* __restrict__ partial,
                                                            const float* __restrict__ gscale, int64_t n, int cin) {
  __shared__ float W[32][kMaxCin + 1];   // [o][i]
  __shared__ float X[64][kMaxCin + 1];
  __shared__ float G[64][33];
  for (int t = threadIdx.x; t < 32 * cin; t += 256) W[t / cin][t % cin] = round_h(w[t]);
  const int64_t p0 = (int64_t)blockIdx.x * 64;
  for (int t = threadIdx.x; t < 64 * cin; t += 256) {
    const int p = t / cin, i = t - p * cin;
    X[p][i] = p0 + p < n ? round_h(feat[(p0 + p) * cin + i]) : 0.f;
  }
  {
    const int p = threadIdx.x >> 2, q = threadIdx.x & 3;
    float a[8], d[8];
    if (p0 + p < n) {
      load8(h + (p0 + p) * 32 + 8 * q, a);
      load8(dh + (p0 + p) * 32 + 8 * q, d);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) G[p][8 * q + e] = (p0 + p < n && a[e] > 0.f) ? d[e] : 0.f;
  }
  __syncthreads();
  {  // data gradient: thread = (pixel, quarter of the inputs)
    const int p = threadIdx.x >> 2, q = threadIdx.x & 3;
    const int per = (cin + 3) / 4, i0 = q * per, i1 = min(cin, i0 + per);
    if (p0 + p < n)
      for (int i = i0; i < i1; ++i) {
        float s = 0.f;
#pragma unroll 8
        for (int o = 0; o < 32; ++o) s = fmaf(G[p][o], W[o][i], s);
        dfeat[(p0 + p) * cin + i] = gscale ? s * gscale[1] : s;
      }
  }
  {  // weight gradient of this block's 64 pixels: thread = (o, eighth of the inputs)
    const int o = threadIdx.x >> 3, q = threadIdx.x & 7;
    const int per = (cin + 7) / 8, i0 = q * per;
    float acc[8], sb = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int p = 0; p < 64; ++p) {
      const float g = G[p][o];
      sb += g;
#pragma unroll
      for (int k = 0; k < 8; ++k)
        if (k < per && i0 + k < cin) acc[k] = fmaf(g, X[p][i0 + k], acc[k]);
    }
    float* dst = partial + (size_t)blockIdx.x * (32 * cin + 32);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < per && i0 + k < cin) dst[o * cin + i0 + k] = acc[k];
    if (q == 0) dst[32 * cin + o] = sb;
  }
}

// ConvTranspose2d(32, 32, kernel = stride = 3) (models/neurad.py:206-211): every input pixel writes its own 3x3 block of
// output pixels, i.e. nine [pixels, 32] x [32, 32] products -- on the matrix cores, one wave per 32 input pixels.
// weight [ci][co][3][3] -> B fragments wup[dir][tap][h][lane]: dir 0 (forward) B[k = ci][j = co] = w[k][j][tap];
// dir 1 (input gradient) B[k = co][j = ci] = w[j][k][tap]
__global__ void upsample_pack_kernel(const float* __restrict__ w, uint4* __restrict__ wup) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= 2 * 9 * 2 * 64) return;
  const int lane = t & 63, h = (t >> 6) & 1, tap = (t >> 7) % 9, dir = t / (9 * 128);
  const int j = lane & 31, k0 = 16 * h + 8 * (lane >> 5);
  Frag f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = k0 + e;
    f.h[e] = (_Float16)(dir == 0 ? w[(k * 32 + j) * 9 + tap] : w[(j * 32 + k) * 9 + tap]);
  }
  wup[t] = f.u;
}

// element offset of the 3x3 output block of flattened input pixel p
__device__ __forceinline__ int up_block_offset(int64_t p, int H, int W) {
  const int x = (int)(p % W), y = (int)((p / W) % H), b = (int)(p / ((int64_t)W * H));
  return (((b * 3 * H + 3 * y) * 3 * W) + 3 * x) * 32;
}

__global__ __launch_bounds__(256) void upsample_fwd_kernel(const _Float16* __restrict__ h, const uint4* __restrict__ wup,
                                                          const float* __restrict__ bias, _Float16* __restrict__ out,
                                                          int H, int W, int64_t npix) {
  const int lane = threadIdx.x & 63, ch = lane & 31, kb = lane >> 5;
  Frag Bf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    Bf[t][0].u = wup[(t * 2 + 0) * 64 + lane];
    Bf[t][1].u = wup[(t * 2 + 1) * 64 + lane];
  }
  const float bj = bias[ch];
  const int64_t ngroups = (npix + 31) / 32;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t p = grp * 32 + ch;
    Frag a0, a1;
    a0.u = a1.u = uint4{0u, 0u, 0u, 0u};
    int base = 0;
    if (p < npix) {
      const uint4* src = (const uint4*)(h + p * 32);
      a0.u = src[kb];
      a1.u = src[2 + kb];
      base = up_block_offset(p, H, W);
    }
    int obase[16];  // C layout: register r of this lane belongs to pixel (r & 3) + 8 (r >> 2) + 4 kb of the group
#pragma unroll
    for (int r = 0; r < 16; ++r) obase[r] = __shfl(base, (r & 3) + 8 * (r >> 2) + 4 * kb, 64);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = bj;
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, Bf[t][0].h, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, Bf[t][1].h, acc, 0, 0, 0);
      const int toff = ((t / 3) * 3 * W + (t % 3)) * 32 + ch;
#pragma unroll
      for (int r = 0; r < 16; ++r)
        if (grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb < npix) out[obase[r] + toff] = (_Float16)acc[r];
    }
  }
}

// dh [pixel][ci] = sum over taps and co of dup[3y+dy, 3x+dx][co] w[ci][co][tap]
__global__ __launch_bounds__(256) void upsample_bwd_data_kernel(const _Float16* __restrict__ dup,
                                                               const uint4* __restrict__ wup /* dir 1 */,
                                                               _Float16* __restrict__ dh, int H, int W, int64_t npix) {
  const int lane = threadIdx.x & 63, ch = lane & 31, kb = lane >> 5;
  Frag Bf[9][2];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    Bf[t][0].u = wup[(t * 2 + 0) * 64 + lane];
    Bf[t][1].u = wup[(t * 2 + 1) * 64 + lane];
  }
  const int64_t ngroups = (npix + 31) / 32;
  for (int64_t grp = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t p = grp * 32 + ch;
    const bool ok = p < npix;
    const int base = ok ? up_block_offset(p, H, W) : 0;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      Frag a0, a1;
      a0.u = a1.u = uint4{0u, 0u, 0u, 0u};
      if (ok) {
        const uint4* src = (const uint4*)(dup + base + ((t / 3) * 3 * W + (t % 3)) * 32);
        a0.u = src[kb];
        a1.u = src[2 + kb];
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, Bf[t][0].h, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, Bf[t][1].h, acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t pp = grp * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      if (pp < npix) dh[pp * 32 + ch] = (_Float16)acc[r];
    }
  }
}

// dw [ci][co][tap] = sum over pixels of h[p][ci] dup[p, tap][co], dbias [co] = sum of dup: contraction over pixels, so both
// operands are transposed by identity MFMAs (as in conv7_wgrad_kernel) -- their C-layout registers, rounded back to fp16,
// ARE the next MFMA's operands (both sides hold the same pixels in the same slots).  Per WAVE partial [9*1024 + 32].
__device__ __forceinline__ void transpose_regs(const Frag& a0, const Frag& a1, const Frag (&ident)[2], Frag (&out)[2],
                                               float* sum) {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0.h, ident[0].h, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1.h, ident[1].h, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    out[r >> 3].h[r & 7] = (_Float16)c[r];
    if (sum) *sum += c[r];
  }
}

__global__ __launch_bounds__(256) void upsample_bwd_weight_kernel(const _Float16* __restrict__ h,
                                                                 const _Float16* __restrict__ dup,
                                                                 float* __restrict__ partial, int H, int W, int64_t npix) {
  const int lane = threadIdx.x & 63, ch = lane & 31, kb = lane >> 5;
  Frag ident[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh)
#pragma unroll
    for (int e = 0; e < 8; ++e) ident[hh].h[e] = (ch == 16 * hh + 8 * kb + e) ? (_Float16)1.f : (_Float16)0.f;
  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float db = 0.f;
  const int64_t ngroups = (npix + 31) / 32;
  const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6);
  for (int64_t grp = wave_id; grp < ngroups; grp += (int64_t)gridDim.x * 4) {
    const int64_t p = grp * 32 + ch;
    const bool ok = p < npix;
    const int base = ok ? up_block_offset(p, H, W) : 0;
    Frag a0, a1, hT[2];
    a0.u = a1.u = uint4{0u, 0u, 0u, 0u};
    if (ok) {
      const uint4* src = (const uint4*)(h + p * 32);
      a0.u = src[kb];
      a1.u = src[2 + kb];
    }
    transpose_regs(a0, a1, ident, hT, nullptr);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      Frag g0, g1, gT[2];
      g0.u = g1.u = uint4{0u, 0u, 0u, 0u};
      if (ok) {
        const uint4* src = (const uint4*)(dup + base + ((t / 3) * 3 * W + (t % 3)) * 32);
        g0.u = src[kb];
        g1.u = src[2 + kb];
      }
      transpose_regs(g0, g1, ident, gT, &db);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hT[0].h, gT[0].h, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hT[1].h, gT[1].h, acc[t], 0, 0, 0);
    }
  }
  float* dst = partial + (size_t)wave_id * (9 * 1024 + 32);
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + ch) * 9 + t] = acc[t][r];
  db += __shfl_xor(db, 32, 64);
  if (kb == 0) dst[9 * 1024 + ch] = db;
}

// the backward's working scale (mixed-precision training keeps fp16 gradients away from the subnormals with a loss scale,
// engine/trainer.py:553; inside the decoder the same is done per call): scale = {S, 1/S}, S the power of two that brings
// max |grad_rgb| to [0.5, 1)
__global__ __launch_bounds__(256) void grad_amax_kernel(const float* __restrict__ g, int64_t n, uint32_t* __restrict__ amax) {
  __shared__ float red[4];
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float v = fabsf(g[i]);
    if (v < INFINITY) m = fmaxf(m, v);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    if (m > 0.f) atomicMax(amax, __float_as_uint(m));  // non-negative floats order like their bits; one atomic per block
  }
}
__global__ void grad_scale_kernel(const uint32_t* __restrict__ amax, float* __restrict__ scale) {
  const float m = __uint_as_float(*amax);
  int e = 0;
  if (m > 0.f) {
    frexpf(m, &e);  // m = f * 2^e, f in [0.5, 1)
    e = -e;
    e = e > 60 ? 60 : (e < -60 ? -60 : e);
  }
  scale[0] = ldexpf(1.f, e);
  scale[1] = ldexpf(1.f, -e);
}

// last layer: Conv2d(32, 3, 1) + Sigmoid (models/neurad.py:214-215): rgb [n][3] fp32.  thread = pixel
__global__ __launch_bounds__(256) void rgb_fwd_kernel(const _Float16* __restrict__ h, const float* __restrict__ w,
                                                     const float* __restrict__ bias, float* __restrict__ rgb, int64_t n) {
  __shared__ float Wl[3][32];
  if (threadIdx.x < 96) Wl[threadIdx.x >> 5][threadIdx.x & 31] = round_h(w[threadIdx.x]);
  __syncthreads();
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= n) return;
  float a0 = bias[0], a1 = bias[1], a2 = bias[2];
#pragma unroll
  for (int c8 = 0; c8 < 4; ++c8) {
    float v[8];
    load8(h + p * 32 + 8 * c8, v);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      a0 = fmaf(v[k], Wl[0][8 * c8 + k], a0);
      a1 = fmaf(v[k], Wl[1][8 * c8 + k], a1);
      a2 = fmaf(v[k], Wl[2][8 * c8 + k], a2);
    }
  }
  rgb[p * 3 + 0] = 1.f / (1.f + __expf(-a0));
  rgb[p * 3 + 1] = 1.f / (1.f + __expf(-a1));
  rgb[p * 3 + 2] = 1.f / (1.f + __expf(-a2));
}

constexpr int kRgbChunks = 4;
// its backward: dl = drgb * rgb (1 - rgb); dh = W^T dl (fp16); partial [blocks of 1024 pixels][99] = (dl^T h [3][32], sum dl [3])
__global__ __launch_bounds__(256) void rgb_bwd_kernel(const _Float16* __restrict__ h, const float* __restrict__ rgb,
                                                     const float* __restrict__ drgb, const float* __restrict__ w,
                                                     _Float16* __restrict__ dh, float* __restrict__ partial,
                                                     const float* __restrict__ gscale, int64_t n) {
  __shared__ float Wl[3][32];
  __shared__ float Hs[256][33];
  __shared__ float Ds[256][3];
  if (threadIdx.x < 96) Wl[threadIdx.x >> 5][threadIdx.x & 31] = round_h(w[threadIdx.x]);
  float s0 = 0.f, s1 = 0.f;
  const float up = gscale ? gscale[0] : 1.f;  // the fp16 gradients below carry this factor, parameter gradients undo it
  for (int chunk = 0; chunk < kRgbChunks; ++chunk) {
    const int64_t p = ((int64_t)blockIdx.x * kRgbChunks + chunk) * 256 + threadIdx.x;
    __syncthreads();
    float dl[3] = {0.f, 0.f, 0.f};
    if (p < n) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float r = rgb[p * 3 + k];
        dl[k] = drgb[p * 3 + k] * up * r * (1.f - r);
      }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) Ds[threadIdx.x][k] = dl[k];
#pragma unroll
    for (int c8 = 0; c8 < 4; ++c8) {
      float v[8], o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if (p < n) load8(h + p * 32 + 8 * c8, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        Hs[threadIdx.x][8 * c8 + e] = v[e];
        o[e] = fmaf(dl[0], Wl[0][8 * c8 + e], fmaf(dl[1], Wl[1][8 * c8 + e], dl[2] * Wl[2][8 * c8 + e]));
      }
      if (p < n) store8(dh + p * 32 + 8 * c8, o);
    }
    __syncthreads();
    if (threadIdx.x < 99) {
      const int k = threadIdx.x < 96 ? threadIdx.x >> 5 : threadIdx.x - 96, c = threadIdx.x & 31;
      if (threadIdx.x < 96) {
        for (int q = 0; q < 256; q += 2) {
          s0 = fmaf(Ds[q][k], Hs[q][c], s0);
          s1 = fmaf(Ds[q + 1][k], Hs[q + 1][c], s1);
        }
      } else {
        for (int q = 0; q < 256; q += 2) {
          s0 += Ds[q][k];
          s1 += Ds[q + 1][k];
        }
      }
    }
  }
  if (threadIdx.x < 99) partial[(size_t)blockIdx.x * 99 + threadIdx.x] = s0 + s1;
}

// BatchNorm2d in eval mode: coefficients from the running statistics
__global__ void bn_eval_coef_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                    const float* __restrict__ running_mean, const float* __restrict__ running_var, float eps,
                                    float* __restrict__ coef) {
  const int c = threadIdx.x;
  if (c >= 32) return;
  const float rstd = 1.f / sqrtf(running_var[c] + eps);
  const float sc = gamma[c] * rstd;
  coef[c] = sc;
  coef[32 + c] = beta[c] - running_mean[c] * sc;
  coef[64 + c] = running_mean[c];
  coef[96 + c] = rstd;
}

inline int ew_blocks(int64_t n8) { return (int)((n8 + 255) / 256); }
inline int reduce_blocks(int64_t n8) {
  int64_t b = (n8 + 256 * 16 - 1) / (256 * 16);  // >= 16 items per thread
  return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

}  // namespace
}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_conv7x7_pack(const float* weight, int32_t mode, void* wfrag, void* stream) {
  NR_REQUIRE(weight && wfrag && (mode == 0 || mode == 1), NRHIP_ERR_INVALID_ARG, "conv7x7_pack: bad argument");
  hipLaunchKernelGGL(conv7_pack_kernel, dim3((49 * 2 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, mode,
                     (uint4*)wfrag);
  return check_launch("conv7x7_pack");
}

extern "C" int nrhip_conv7x7_pack_many(const float* const* weights, int32_t n, void* wfrag, void* stream) {
  NR_REQUIRE(weights && wfrag && n >= 1 && n <= 8, NRHIP_ERR_INVALID_ARG, "conv7x7_pack_many: bad argument");
  PackMany src;
  for (int i = 0; i < 8; ++i) src.w[i] = i < n ? weights[i] : nullptr;
  for (int i = 0; i < n; ++i) NR_REQUIRE(src.w[i], NRHIP_ERR_INVALID_ARG, "conv7x7_pack_many: weight %d is null", i);
  hipLaunchKernelGGL(conv7_pack_many_kernel, dim3((49 * 2 * 64 + 255) / 256, 2 * n), dim3(256), 0, (hipStream_t)stream, src,
                     n, (uint4*)wfrag);
  return check_launch("conv7x7_pack_many");
}

extern "C" int nrhip_conv7x7_tiles(int32_t h, int32_t w, int32_t rows_per_wave, int32_t* tiles) {
  NR_REQUIRE(tiles && h > 0 && w > 0 && (rows_per_wave == 1 || rows_per_wave == 2 || rows_per_wave == 4),
             NRHIP_ERR_INVALID_ARG, "conv7x7_tiles: bad argument");
  *tiles = ((w + kTW - 1) / kTW) * ((h + 4 * rows_per_wave - 1) / (4 * rows_per_wave));
  return NRHIP_OK;
}

extern "C" int nrhip_conv7x7(const void* in, const void* wfrag, const float* bias, void* out, float* stats_partial,
                             int32_t b, int32_t h, int32_t w, int32_t rows_per_wave, void* stream) {
  NR_REQUIRE(in && wfrag && out && b >= 0 && h > 0 && w > 0, NRHIP_ERR_INVALID_ARG, "conv7x7: bad argument");
  NR_REQUIRE(b <= 65535, NRHIP_ERR_UNSUPPORTED, "conv7x7: %d images per call (max 65535)", b);
  if (b == 0) return NRHIP_OK;
  hipStream_t st = (hipStream_t)stream;
  switch (rows_per_wave) {
    case 1: return launch_conv7<1>(in, wfrag, bias, out, stats_partial, b, h, w, st);
    case 2: return launch_conv7<2>(in, wfrag, bias, out, stats_partial, b, h, w, st);
    case 4: return launch_conv7<4>(in, wfrag, bias, out, stats_partial, b, h, w, st);
  }
  set_error("conv7x7: rows_per_wave must be 1, 2 or 4");
  return NRHIP_ERR_INVALID_ARG;
}

extern "C" int nrhip_conv7x7_wgrad_workspace(int32_t b, int32_t h, int32_t w, int64_t* floats) {
  NR_REQUIRE(floats && b >= 0 && h > 0 && w > 0, NRHIP_ERR_INVALID_ARG, "conv7x7_wgrad_workspace: bad argument");
  int rps, spi;
  wgrad_plan(b > 0 ? b : 1, h, &rps, &spi);
  *floats = (int64_t)b * spi * (49 * 1024 + 32);
  return NRHIP_OK;
}

extern "C" int nrhip_conv7x7_wgrad(const void* x, const void* grad_out, float* workspace, float* grad_weight,
                                   float* grad_bias, const float* grad_scale, int32_t b, int32_t h, int32_t w,
                                   void* stream) {
  NR_REQUIRE(x && grad_out && workspace && grad_weight && b >= 0 && h > 0 && w > 0, NRHIP_ERR_INVALID_ARG,
             "conv7x7_wgrad: bad argument");
  if (b == 0) return NRHIP_OK;
  int rps, spi;
  wgrad_plan(b, h, &rps, &spi);
  const int n = b * spi;
  const int nkb = (w + 15) / 16, nxb = (16 * nkb + 8 + 31) / 32, ngb = (16 * nkb + 31) / 32;
  const size_t smem = (size_t)(8 * 32 * host_pitch16(32 * nxb) + 2 * 32 * host_pitch16(32 * ngb)) * 2;
  NR_REQUIRE(smem <= 160 * 1024, NRHIP_ERR_UNSUPPORTED, "conv7x7_wgrad: image width %d needs %zu bytes of LDS", w, smem);
  hipStream_t st = (hipStream_t)stream;
  float* partial_db = workspace + (size_t)n * 49 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)conv7_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(conv7_wgrad_kernel, dim3(n), dim3(kWgradWaves * 64), smem, st, (const _Float16*)x,
                     (const _Float16*)grad_out, workspace, partial_db, h, w, rps, spi);
  hipLaunchKernelGGL(conv7_wgrad_reduce_kernel, dim3(49 * 32 + 1), dim3(256), 0, st, workspace, partial_db, n, grad_weight,
                     grad_bias, grad_scale);
  return check_launch("conv7x7_wgrad");
}

// ---- entry points of the other layers -------------------------------------------------------------------------------
extern "C" int nrhip_dec_bn_finalize(const float* stats_partial, int32_t n_partial, int64_t count, const float* gamma,
                                     const float* beta, float eps, float momentum, float* running_mean,
                                     float* running_var, float* coef, void* stream) {
  NR_REQUIRE(stats_partial && n_partial > 0 && count > 0 && gamma && beta && coef && (!running_mean == !running_var),
             NRHIP_ERR_INVALID_ARG, "dec_bn_finalize: bad argument");
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stats_partial, n_partial,
                     (double)count, gamma, beta, eps, momentum, running_mean, running_var, coef);
  return check_launch("dec_bn_finalize");
}

extern "C" int nrhip_dec_bn_act(const void* c, const float* coef, const void* skip, void* out, int64_t n_pixels,
                                void* stream) {
  NR_REQUIRE(c && coef && out && n_pixels >= 0, NRHIP_ERR_INVALID_ARG, "dec_bn_act: bad argument");
  if (n_pixels == 0) return NRHIP_OK;
  hipLaunchKernelGGL(bn_act_kernel, dim3(ew_blocks(n_pixels * 4)), dim3(256), 0, (hipStream_t)stream, (const _Float16*)c,
                     coef, (const _Float16*)skip, (_Float16*)out, n_pixels * 4);
  return check_launch("dec_bn_act");
}

namespace {
int bn_bwd_impl(const void* grad_out, const void* act, const void* c, const float* gamma, const float* coef, float* workspace,
                float* grad_gamma, float* grad_beta, const float* grad_scale, void* grad_c, int64_t n_pixels, int eval_mode,
                void* stream);
}

extern "C" int nrhip_dec_bn_bwd(const void* grad_out, const void* act, const void* c, const float* gamma,
                                const float* coef, float* workspace, float* grad_gamma, float* grad_beta,
                                const float* grad_scale, void* grad_c, int64_t n_pixels, void* stream) {
  return bn_bwd_impl(grad_out, act, c, gamma, coef, workspace, grad_gamma, grad_beta, grad_scale, grad_c, n_pixels, 0, stream);
}

namespace {
int bn_bwd_impl(const void* grad_out, const void* act, const void* c, const float* gamma, const float* coef, float* workspace,
                float* grad_gamma, float* grad_beta, const float* grad_scale, void* grad_c, int64_t n_pixels, int eval_mode,
                void* stream) {
  NR_REQUIRE(grad_out && act && c && gamma && coef && workspace && grad_c && n_pixels > 0, NRHIP_ERR_INVALID_ARG,
             "dec_bn_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nb = reduce_blocks(n_pixels * 4);
  float* bcoef = workspace + (size_t)nb * 64;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(nb), dim3(256), 0, st, (const _Float16*)grad_out, (const _Float16*)act,
                     (const _Float16*)c, n_pixels * 4, workspace);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(1), dim3(1024), 0, st, workspace, nb, (double)n_pixels, gamma, coef, bcoef,
                     grad_gamma, grad_beta, grad_scale, eval_mode);
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n_pixels * 4)), dim3(256), 0, st, (const _Float16*)grad_out,
                     (const _Float16*)act, (const _Float16*)c, bcoef, 0, (_Float16*)grad_c, n_pixels * 4);
  return check_launch("dec_bn_bwd");
}
}  // namespace

extern "C" int nrhip_dec_bn_bwd_workspace(int64_t n_pixels, int64_t* floats) {
  NR_REQUIRE(floats && n_pixels >= 0, NRHIP_ERR_INVALID_ARG, "dec_bn_bwd_workspace: bad argument");
  *floats = (int64_t)reduce_blocks(n_pixels * 4) * 64 + 96;
  return NRHIP_OK;
}

extern "C" int nrhip_dec_add_masked(const void* a, const void* grad_out, const void* act, void* out, int64_t n_pixels,
                                    void* stream) {
  NR_REQUIRE(a && grad_out && act && out && n_pixels >= 0, NRHIP_ERR_INVALID_ARG, "dec_add_masked: bad argument");
  if (n_pixels == 0) return NRHIP_OK;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(ew_blocks(n_pixels * 4)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)grad_out, (const _Float16*)act, (const _Float16*)a, (const float*)nullptr, 1,
                     (_Float16*)out, n_pixels * 4);
  return check_launch("dec_add_masked");
}

extern "C" int nrhip_dec_conv1x1_in_fwd(const float* features, const float* weight, const float* bias, void* out,
                                        int64_t n, int32_t cin, void* stream) {
  NR_REQUIRE(features && weight && bias && out && n >= 0, NRHIP_ERR_INVALID_ARG, "dec_conv1x1_in_fwd: bad argument");
  NR_REQUIRE(cin >= 1 && cin <= kMaxCin, NRHIP_ERR_UNSUPPORTED, "dec_conv1x1_in_fwd: %d input channels (max %d)", cin,
             kMaxCin);
  if (n == 0) return NRHIP_OK;
  hipLaunchKernelGGL(conv1x1_in_fwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, features,
                     weight, bias, (_Float16*)out, n, cin);
  return check_launch("dec_conv1x1_in_fwd");
}

extern "C" int nrhip_dec_conv1x1_in_bwd_workspace(int64_t n, int32_t cin, int64_t* floats) {
  NR_REQUIRE(floats && n >= 0 && cin >= 1 && cin <= kMaxCin, NRHIP_ERR_INVALID_ARG, "dec_conv1x1_in_bwd_workspace: bad argument");
  *floats = ((n + 63) / 64) * (32 * (int64_t)cin + 32);
  return NRHIP_OK;
}

extern "C" int nrhip_dec_conv1x1_in_bwd(const float* features, const void* h, const void* grad_h, const float* weight,
                                        float* workspace, float* grad_features, float* grad_weight, float* grad_bias,
                                        const float* grad_scale, int64_t n, int32_t cin, void* stream) {
  NR_REQUIRE(features && h && grad_h && weight && workspace && grad_features && grad_weight && grad_bias && n > 0,
             NRHIP_ERR_INVALID_ARG, "dec_conv1x1_in_bwd: bad argument");
  NR_REQUIRE(cin >= 1 && cin <= kMaxCin, NRHIP_ERR_UNSUPPORTED, "dec_conv1x1_in_bwd: %d input channels (max %d)", cin,
             kMaxCin);
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((n + 63) / 64), k = 32 * cin + 32;
  hipLaunchKernelGGL(conv1x1_in_bwd_kernel, dim3(nb), dim3(256), 0, st, features, (const _Float16*)h,
                     (const _Float16*)grad_h, weight, grad_features, workspace, grad_scale, n, cin);
  // columns [0, 32 cin) -> grad_weight, [32 cin, +32) -> grad_bias (accumulated)
  hipLaunchKernelGGL(reduce_cols_kernel, dim3(cin + 1), dim3(256), 0, st, workspace, nb, k, 32 * cin + 32, grad_weight, 1,
                     grad_scale, 32 * cin, grad_bias);  // weight columns, then the 32 bias columns
  return check_launch("dec_conv1x1_in_bwd");
}

constexpr int kUpWgs = 64;  // workgroups of the transposed convolution's weight gradient: 256 waves, one partial each

extern "C" int nrhip_dec_upsample_pack(const float* weight, void* wup, void* stream) {
  NR_REQUIRE(weight && wup, NRHIP_ERR_INVALID_ARG, "dec_upsample_pack: bad argument");
  hipLaunchKernelGGL(upsample_pack_kernel, dim3((2 * 9 * 2 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight,
                     (uint4*)wup);
  return check_launch("dec_upsample_pack");
}

extern "C" int nrhip_dec_upsample_fwd(const void* h, const void* wup, const float* bias, void* out, int32_t b,
                                      int32_t hh, int32_t w, void* stream) {
  NR_REQUIRE(h && wup && bias && out && b >= 0 && hh > 0 && w > 0, NRHIP_ERR_INVALID_ARG, "dec_upsample_fwd: bad argument");
  const int64_t npix = (int64_t)b * hh * w;
  NR_REQUIRE(npix * 9 * 32 < (int64_t)1 << 31, NRHIP_ERR_UNSUPPORTED, "dec_upsample_fwd: %lld output elements (32-bit offsets)",
             (long long)(npix * 9 * 32));
  if (b == 0) return NRHIP_OK;
  const int64_t wgs = ((npix + 31) / 32 + 3) / 4;
  hipLaunchKernelGGL(upsample_fwd_kernel, dim3((unsigned)(wgs > 1024 ? 1024 : wgs)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)h, (const uint4*)wup, bias, (_Float16*)out, hh, w, npix);
  return check_launch("dec_upsample_fwd");
}

extern "C" int nrhip_dec_upsample_bwd_workspace(int32_t b, int32_t hh, int32_t w, int64_t* floats) {
  NR_REQUIRE(floats && b >= 0 && hh > 0 && w > 0, NRHIP_ERR_INVALID_ARG, "dec_upsample_bwd_workspace: bad argument");
  *floats = (int64_t)kUpWgs * 4 * (9 * 1024 + 32);
  return NRHIP_OK;
}

extern "C" int nrhip_dec_upsample_bwd(const void* h, const void* grad_out, const void* wup, float* workspace, void* grad_h,
                                      float* grad_weight, float* grad_bias, const float* grad_scale, int32_t b, int32_t hh,
                                      int32_t w, void* stream) {
  NR_REQUIRE(h && grad_out && wup && workspace && grad_h && grad_weight && grad_bias && b > 0 && hh > 0 && w > 0,
             NRHIP_ERR_INVALID_ARG, "dec_upsample_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int64_t npix = (int64_t)b * hh * w;
  NR_REQUIRE(npix * 9 * 32 < (int64_t)1 << 31, NRHIP_ERR_UNSUPPORTED, "dec_upsample_bwd: %lld output elements (32-bit offsets)",
             (long long)(npix * 9 * 32));
  const int k = 9 * 1024 + 32;
  const int64_t wgs = ((npix + 31) / 32 + 3) / 4;
  hipLaunchKernelGGL(upsample_bwd_data_kernel, dim3((unsigned)(wgs > 1024 ? 1024 : wgs)), dim3(256), 0, st,
                     (const _Float16*)grad_out, (const uint4*)wup + 9 * 2 * 64, (_Float16*)grad_h, hh, w, npix);
  hipLaunchKernelGGL(upsample_bwd_weight_kernel, dim3(kUpWgs), dim3(256), 0, st, (const _Float16*)h,
                     (const _Float16*)grad_out, workspace, hh, w, npix);
  hipLaunchKernelGGL(reduce_cols_kernel, dim3(9 * 32 + 1), dim3(256), 0, st, workspace, kUpWgs * 4, k, 9 * 1024 + 32, grad_weight,
                     1, grad_scale, 9 * 1024, grad_bias);
  return check_launch("dec_upsample_bwd");
}

extern "C" int nrhip_dec_grad_scale(const float* grad, int64_t n, float* scale, void* stream) {
  NR_REQUIRE(grad && scale && n >= 0, NRHIP_ERR_INVALID_ARG, "dec_grad_scale: bad argument");
  hipStream_t st = (hipStream_t)stream;
  uint32_t* amax = (uint32_t*)(scale + 2);
  if (hipMemsetAsync(amax, 0, sizeof(uint32_t), st) != hipSuccess) {
    set_error("dec_grad_scale: memset failed");
    return NRHIP_ERR_LAUNCH;
  }
  if (n > 0) {
    const int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(grad_amax_kernel, dim3((unsigned)(blocks > 128 ? 128 : blocks)), dim3(256), 0, st, grad, n, amax);
  }
  hipLaunchKernelGGL(grad_scale_kernel, dim3(1), dim3(1), 0, st, amax, scale);
  return check_launch("dec_grad_scale");
}

extern "C" int nrhip_dec_rgb_fwd(const void* h, const float* weight, const float* bias, float* rgb, int64_t n_pixels,
                                 void* stream) {
  NR_REQUIRE(h && weight && bias && rgb && n_pixels >= 0, NRHIP_ERR_INVALID_ARG, "dec_rgb_fwd: bad argument");
  if (n_pixels == 0) return NRHIP_OK;
  hipLaunchKernelGGL(rgb_fwd_kernel, dim3((unsigned)((n_pixels + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     (const _Float16*)h, weight, bias, rgb, n_pixels);
  return check_launch("dec_rgb_fwd");
}

extern "C" int nrhip_dec_rgb_bwd_workspace(int64_t n_pixels, int64_t* floats) {
  NR_REQUIRE(floats && n_pixels >= 0, NRHIP_ERR_INVALID_ARG, "dec_rgb_bwd_workspace: bad argument");
  *floats = ((n_pixels + 256 * kRgbChunks - 1) / (256 * kRgbChunks)) * 99;
  return NRHIP_OK;
}

extern "C" int nrhip_dec_rgb_bwd(const void* h, const float* rgb, const float* grad_rgb, const float* weight,
                                 float* workspace, void* grad_h, float* grad_weight, float* grad_bias,
                                 const float* grad_scale, int64_t n_pixels, void* stream) {
  NR_REQUIRE(h && rgb && grad_rgb && weight && workspace && grad_h && grad_weight && grad_bias && n_pixels > 0,
             NRHIP_ERR_INVALID_ARG, "dec_rgb_bwd: bad argument");
  hipStream_t st = (hipStream_t)stream;
  const int nb = (int)((n_pixels + 256 * kRgbChunks - 1) / (256 * kRgbChunks));
  hipLaunchKernelGGL(rgb_bwd_kernel, dim3(nb), dim3(256), 0, st, (const _Float16*)h, rgb, grad_rgb, weight,
                     (_Float16*)grad_h, workspace, grad_scale, n_pixels);
  hipLaunchKernelGGL(reduce_cols_kernel, dim3(4), dim3(256), 0, st, workspace, nb, 99, 99, grad_weight, 1, grad_scale, 96,
                     grad_bias);
  return check_launch("dec_rgb_bwd");
}

// ---- the whole decoder behind one entry point per direction ----------------------------------------------------------
namespace {

inline int64_t align256(int64_t b) { return (b + 255) / 256 * 256; }
constexpr int64_t kWfragBytes = 49 * 2 * 64 * 16;
constexpr int64_t kWupBytes = 2 * 9 * 2 * 64 * 16;

struct DecLayout {
  int64_t n_lo, n_hi;                  // pixels at the patch resolution / at 3x
  int64_t h0, blk[4][4], up;           // byte offsets in `saved`: per block c1, u1, c2, out
  int64_t coef, packed, wup, saved_bytes;
  int rows_lo, rows_hi;                // conv7x7 rows per wave
  int64_t stats_floats, ws_floats, grad_buf_bytes, workspace_bytes;
};

inline int rows_per_wave_for(int b, int h, int w) {
  const int r[3] = {4, 2, 1};
  for (int i = 0; i < 3; ++i)
    if ((int64_t)b * ((w + 31) / 32) * ((h + 4 * r[i] - 1) / (4 * r[i])) >= 512) return r[i];
  return 1;
}

int dec_layout(const nrhip_rgb_decoder* d, DecLayout* L) {
  const int b = d->n_patches, h = d->patch_h, w = d->patch_w;
  L->n_lo = (int64_t)b * h * w;
  L->n_hi = 9 * L->n_lo;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    const int64_t o = off;
    off += align256(bytes);
    return o;
  };
  L->h0 = take(L->n_lo * 64);
  for (int k = 0; k < 4; ++k)
    for (int j = 0; j < 4; ++j) L->blk[k][j] = take((k < 2 ? L->n_lo : L->n_hi) * 64);
  L->up = take(L->n_hi * 64);
  L->coef = take(8 * 128 * sizeof(float));
  L->packed = take(8 * 2 * kWfragBytes);
  L->wup = take(kWupBytes);
  L->saved_bytes = off;
  L->rows_lo = rows_per_wave_for(b, h, w);
  L->rows_hi = rows_per_wave_for(b, 3 * h, 3 * w);
  int32_t t_lo = 0, t_hi = 0;
  nrhip_conv7x7_tiles(h, w, L->rows_lo, &t_lo);
  nrhip_conv7x7_tiles(3 * h, 3 * w, L->rows_hi, &t_hi);
  L->stats_floats = (int64_t)b * (t_lo > t_hi ? t_lo : t_hi) * 64;
  int64_t m = L->stats_floats, v = 0;
  nrhip_conv7x7_wgrad_workspace(b, 3 * h, 3 * w, &v);
  m = v > m ? v : m;
  nrhip_conv7x7_wgrad_workspace(b, h, w, &v);
  m = v > m ? v : m;
  nrhip_dec_bn_bwd_workspace(L->n_hi, &v);
  m = v > m ? v : m;
  nrhip_dec_upsample_bwd_workspace(b, h, w, &v);
  m = v > m ? v : m;
  nrhip_dec_rgb_bwd_workspace(L->n_hi, &v);
  m = v > m ? v : m;
  nrhip_dec_conv1x1_in_bwd_workspace(L->n_lo, d->cin, &v);
  m = v > m ? v : m;
  L->ws_floats = m;
  L->grad_buf_bytes = align256(L->n_hi * 64);
  L->workspace_bytes = align256(m * 4) + 3 * L->grad_buf_bytes + 256;
  return NRHIP_OK;
}

int dec_validate(const nrhip_rgb_decoder* d, const char* who) {
  NR_REQUIRE(d, NRHIP_ERR_INVALID_ARG, "%s: null decoder", who);
  NR_REQUIRE(d->n_patches >= 1 && d->patch_h >= 1 && d->patch_w >= 1 && d->cin >= 1 && d->cin <= kMaxCin,
             NRHIP_ERR_INVALID_ARG, "%s: bad shape (%d patches of %d x %d, %d channels)", who, d->n_patches, d->patch_h,
             d->patch_w, d->cin);
  NR_REQUIRE((int64_t)d->n_patches * d->patch_h * d->patch_w * 9 * 32 < (int64_t)1 << 31, NRHIP_ERR_UNSUPPORTED,
             "%s: batch too large for 32-bit offsets", who);
  bool ok = d->conv_in_w && d->conv_in_b && d->up_w && d->up_b && d->out_w && d->out_b;
  for (int i = 0; i < 8; ++i)
    ok = ok && d->conv_w[i] && d->conv_b[i] && d->bn_gamma[i] && d->bn_beta[i] && d->bn_running_mean[i] && d->bn_running_var[i];
  NR_REQUIRE(ok, NRHIP_ERR_INVALID_ARG, "%s: null parameter", who);
  return NRHIP_OK;
}

#define DEC_TRY(call)            \
  do {                           \
    const int rc_ = (call);      \
    if (rc_ != NRHIP_OK) return rc_; \
  } while (0)

}  // namespace

extern "C" int nrhip_rgb_decoder_sizes(const nrhip_rgb_decoder* d, int64_t* saved_bytes, int64_t* workspace_bytes,
                                       int64_t* grad_param_floats) {
  DEC_TRY(dec_validate(d, "rgb_decoder_sizes"));
  NR_REQUIRE(saved_bytes && workspace_bytes && grad_param_floats, NRHIP_ERR_INVALID_ARG, "rgb_decoder_sizes: null output");
  DecLayout L;
  dec_layout(d, &L);
  *saved_bytes = L.saved_bytes;
  *workspace_bytes = L.workspace_bytes;
  *grad_param_floats = 32 * (int64_t)d->cin + 32 + 8 * (32 * 32 * 49 + 32 + 32 + 32) + (32 * 32 * 9 + 32) + (96 + 3);
  return NRHIP_OK;
}

extern "C" int nrhip_rgb_decoder_fwd(const nrhip_rgb_decoder* d, const float* features, void* saved, void* workspace,
                                     float* rgb, void* stream) {
  DEC_TRY(dec_validate(d, "rgb_decoder_fwd"));
  NR_REQUIRE(features && saved && workspace && rgb, NRHIP_ERR_INVALID_ARG, "rgb_decoder_fwd: null buffer");
  DecLayout L;
  dec_layout(d, &L);
  char* sv = (char*)saved;
  float* stats = (float*)workspace;
  float* coef = (float*)(sv + L.coef);
  const int b = d->n_patches;
  DEC_TRY(nrhip_conv7x7_pack_many(d->conv_w, 8, sv + L.packed, stream));
  DEC_TRY(nrhip_dec_upsample_pack(d->up_w, sv + L.wup, stream));
  DEC_TRY(nrhip_dec_conv1x1_in_fwd(features, d->conv_in_w, d->conv_in_b, sv + L.h0, L.n_lo, d->cin, stream));
  const void* x = sv + L.h0;
  for (int k = 0; k < 4; ++k) {
    const bool hi = k >= 2;
    const int h = hi ? 3 * d->patch_h : d->patch_h, w = hi ? 3 * d->patch_w : d->patch_w, rows = hi ? L.rows_hi : L.rows_lo;
    const int64_t npix = hi ? L.n_hi : L.n_lo;
    if (k == 2) {
      DEC_TRY(nrhip_dec_upsample_fwd(x, sv + L.wup, d->up_b, sv + L.up, b, d->patch_h, d->patch_w, stream));
      x = sv + L.up;
    }
    int32_t tiles = 0;
    nrhip_conv7x7_tiles(h, w, rows, &tiles);
    void *c1 = sv + L.blk[k][0], *u1 = sv + L.blk[k][1], *c2 = sv + L.blk[k][2], *out = sv + L.blk[k][3];
    for (int j = 0; j < 2; ++j) {  // conv a -> bn -> relu, conv b -> bn -> + skip -> relu
      const int i = 2 * k + j;
      float* cf = coef + 128 * i;
      DEC_TRY(nrhip_conv7x7(j == 0 ? x : u1, sv + L.packed + (int64_t)(2 * i) * kWfragBytes, d->conv_b[i], j == 0 ? c1 : c2,
                            d->training ? stats : nullptr, b, h, w, rows, stream));
      if (d->training) {
        DEC_TRY(nrhip_dec_bn_finalize(stats, b * tiles, npix, d->bn_gamma[i], d->bn_beta[i], d->bn_eps[i],
                                      d->bn_momentum[i], d->bn_running_mean[i], d->bn_running_var[i], cf, stream));
      } else {
        hipLaunchKernelGGL(bn_eval_coef_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, d->bn_gamma[i], d->bn_beta[i],
                           d->bn_running_mean[i], d->bn_running_var[i], d->bn_eps[i], cf);
      }
      DEC_TRY(nrhip_dec_bn_act(j == 0 ? c1 : c2, cf, j == 0 ? nullptr : x, j == 0 ? u1 : out, npix, stream));
    }
    x = out;
  }
  DEC_TRY(nrhip_dec_rgb_fwd(x, d->out_w, d->out_b, rgb, L.n_hi, stream));
  return check_launch("rgb_decoder_fwd");
}

extern "C" int nrhip_rgb_decoder_bwd(const nrhip_rgb_decoder* d, const float* features, const void* saved,
                                     const float* rgb, const float* grad_rgb, void* workspace, float* grad_features,
                                     float* grad_params, void* stream) {
  DEC_TRY(dec_validate(d, "rgb_decoder_bwd"));
  NR_REQUIRE(features && saved && rgb && grad_rgb && workspace && grad_features && grad_params, NRHIP_ERR_INVALID_ARG,
             "rgb_decoder_bwd: null buffer");
  const int eval_mode = d->training ? 0 : 1;  // eval: BatchNorm normalises with its running statistics (constants)
  DecLayout L;
  dec_layout(d, &L);
  hipStream_t st = (hipStream_t)stream;
  const char* sv = (const char*)saved;
  char* wsb = (char*)workspace;
  float* ws = (float*)wsb;
  char* gbuf[3] = {wsb + align256(L.ws_floats * 4), wsb + align256(L.ws_floats * 4) + L.grad_buf_bytes,
                   wsb + align256(L.ws_floats * 4) + 2 * L.grad_buf_bytes};
  float* gscale = (float*)(wsb + align256(L.ws_floats * 4) + 3 * L.grad_buf_bytes);
  const float* coef = (const float*)(sv + L.coef);
  const int b = d->n_patches;
  // grad_params: conv_in (w, b), 4 x (wa, ba, gamma1, beta1, wb, bb, gamma2, beta2), up (w, b), out (w, b)
  int64_t n_par = 0, sb = 0, wb = 0;
  nrhip_rgb_decoder_sizes(d, &sb, &wb, &n_par);
  if (hipMemsetAsync(grad_params, 0, n_par * sizeof(float), st) != hipSuccess) {
    set_error("rgb_decoder_bwd: memset failed");
    return NRHIP_ERR_LAUNCH;
  }
  float* g_in_w = grad_params;
  float* g_in_b = g_in_w + 32 * d->cin;
  float* g_blk = g_in_b + 32;
  const int64_t per_conv = 32 * 32 * 49 + 32 + 32 + 32;
  float* g_up_w = g_blk + 8 * per_conv;
  float* g_up_b = g_up_w + 32 * 32 * 9;
  float* g_out_w = g_up_b + 32;
  float* g_out_b = g_out_w + 96;
  DEC_TRY(nrhip_dec_grad_scale(grad_rgb, L.n_hi * 3, gscale, stream));
  const void* h4 = sv + L.blk[3][3];
  void* dcur = gbuf[0];  // gradient w.r.t. the current block's output
  DEC_TRY(nrhip_dec_rgb_bwd(h4, rgb, grad_rgb, d->out_w, ws, dcur, g_out_w, g_out_b, gscale, L.n_hi, stream));
  for (int k = 3; k >= 0; --k) {
    const bool hi = k >= 2;
    const int h = hi ? 3 * d->patch_h : d->patch_h, w = hi ? 3 * d->patch_w : d->patch_w, rows = hi ? L.rows_hi : L.rows_lo;
    const int64_t npix = hi ? L.n_hi : L.n_lo;
    const void* x = k == 0 ? sv + L.h0 : (k == 2 ? sv + L.up : sv + L.blk[k - 1][3]);
    const void *c1 = sv + L.blk[k][0], *u1 = sv + L.blk[k][1], *c2 = sv + L.blk[k][2], *out = sv + L.blk[k][3];
    void* t1 = dcur == gbuf[0] ? gbuf[1] : gbuf[0];
    void* t2 = (dcur != gbuf[2] && t1 != gbuf[2]) ? gbuf[2] : (dcur != gbuf[1] && t1 != gbuf[1] ? gbuf[1] : gbuf[0]);
    float* ga = g_blk + (2 * k) * per_conv;  // conv a: w, b, gamma, beta
    float* gb = g_blk + (2 * k + 1) * per_conv;
    const int ia = 2 * k, ib = 2 * k + 1;
    // dc2 -> t1; du1 = conv^T(dc2) -> t2; wgrad b
    DEC_TRY(bn_bwd_impl(dcur, out, c2, d->bn_gamma[ib], coef + 128 * ib, ws, gb + 32 * 32 * 49 + 32,
                        gb + 32 * 32 * 49 + 64, gscale, t1, npix, eval_mode, stream));
    DEC_TRY(nrhip_conv7x7(t1, sv + L.packed + (int64_t)(2 * ib + 1) * kWfragBytes, nullptr, t2, nullptr, b, h, w, rows, stream));
    DEC_TRY(nrhip_conv7x7_wgrad(u1, t1, ws, gb, gb + 32 * 32 * 49, gscale, b, h, w, stream));
    // dc1 -> t1 (dc2 is dead); dx = conv^T(dc1) -> t2 (du1 is dead after bn_bwd); wgrad a
    DEC_TRY(bn_bwd_impl(t2, u1, c1, d->bn_gamma[ia], coef + 128 * ia, ws, ga + 32 * 32 * 49 + 32,
                        ga + 32 * 32 * 49 + 64, gscale, t1, npix, eval_mode, stream));
    DEC_TRY(nrhip_conv7x7(t1, sv + L.packed + (int64_t)(2 * ia + 1) * kWfragBytes, nullptr, t2, nullptr, b, h, w, rows, stream));
    DEC_TRY(nrhip_conv7x7_wgrad(x, t1, ws, ga, ga + 32 * 32 * 49, gscale, b, h, w, stream));
    // block input gradient = convolution path + skip path -> t1
    DEC_TRY(nrhip_dec_add_masked(t2, dcur, out, t1, npix, stream));
    dcur = t1;
    if (k == 2) {  // through the transposed convolution
      void* dlo = dcur == gbuf[0] ? gbuf[1] : gbuf[0];
      DEC_TRY(nrhip_dec_upsample_bwd(sv + L.blk[1][3], dcur, sv + L.wup, ws, dlo, g_up_w, g_up_b, gscale, b, d->patch_h,
                                     d->patch_w, stream));
      dcur = dlo;
    }
  }
  DEC_TRY(nrhip_dec_conv1x1_in_bwd(features, sv + L.h0, dcur, d->conv_in_w, ws, grad_features, g_in_w, g_in_b, gscale, L.n_lo,
                                   d->cin, stream));
  return check_launch("rgb_decoder_bwd");
}
