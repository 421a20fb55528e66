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
//   * the input gradient of a convolution is the same kernel on flipped / transposed weights (pack mode 1).
#include "common.h"

namespace nrhip {
namespace {

using half8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kC = 32;                // channels of the decoder's hidden layers
constexpr int kTW = 32;               // tile width = the MFMA's 32 rows
constexpr int kCols = kTW + 6;        // + halo
constexpr int kPix = 80;              // bytes per pixel in LDS: 64 + 16 of padding (ds_read_b128 conflict-free, stride 20 dwords)

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
  for (int c = threadIdx.x; c < ROWS * kCols * 4; c += 256) {
    const int q = c & 3, p = c >> 2;
    const int row = p / kCols, col = p - row * kCols;
    const int y = y0 - 3 + row, x = x0 - 3 + col;
    uint4 v = {0u, 0u, 0u, 0u};
    if (y >= 0 && y < H && x >= 0 && x < W) v = *(const uint4*)(img + ((size_t)y * W + x) * kC + q * 8);
    *(uint4*)(lds + p * kPix + q * 16) = v;
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
  const unsigned char* abase = lds + ((wave * R) * kCols + px) * kPix + kb * 16;
#pragma unroll 1
  for (int kx = 0; kx < 7; ++kx) {
    Frag Bf[7][2];
    const uint4* wp = wfrag + (size_t)kx * 7 * 2 * 64 + lane;
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
      Bf[ky][0].u = wp[(ky * 2 + 0) * 64];
      Bf[ky][1].u = wp[(ky * 2 + 1) * 64];
    }
    Frag A[R + 6][2];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky) {
#pragma unroll
      for (int y = 0; y < R; ++y) {
        const int row = y + ky;
        if (ky == 0 || y == R - 1) {  // first use of this input row at this shift
          const unsigned char* ap = abase + (row * kCols + kx) * kPix;
          A[row][0].u = *(const uint4*)(ap);
          A[row][1].u = *(const uint4*)(ap + 32);
        }
        acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[row][0].h, Bf[ky][0].h, acc[y], 0, 0, 0);
        acc[y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[row][1].h, Bf[ky][1].h, acc[y], 0, 0, 0);
      }
    }
  }
  // C layout: column (channel) = lane & 31, row (pixel) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  float s1 = 0.f, s2 = 0.f;
  _Float16* oimg = out + (size_t)b * H * W * kC;
#pragma unroll
  for (int y = 0; y < R; ++y) {
    const int yy = y0 + wave * R + y;
    if (yy >= H) continue;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * kb;
      if (x < W) {
        const _Float16 hv = (_Float16)acc[y][r];
        oimg[((size_t)yy * W + x) * kC + px] = hv;
        if (STATS) {  // BatchNorm sees the rounded activation
          const float v = (float)hv;
          s1 += v;
          s2 += v * v;
        }
      }
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
                                                                 float* __restrict__ grad_bias) {
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

}  // namespace
}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_conv7x7_pack(const float* weight, int32_t mode, void* wfrag, void* stream) {
  NR_REQUIRE(weight && wfrag && (mode == 0 || mode == 1), NRHIP_ERR_INVALID_ARG, "conv7x7_pack: bad argument");
  hipLaunchKernelGGL(conv7_pack_kernel, dim3((49 * 2 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, weight, mode,
                     (uint4*)wfrag);
  return check_launch("conv7x7_pack");
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
                                   float* grad_bias, int32_t b, int32_t h, int32_t w, void* stream) {
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
                     grad_bias);
  return check_launch("conv7x7_wgrad");
}
