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
  if (partial)
    hipLaunchKernelGGL((conv7_kernel<R, true>), grid, dim3(256), smem, st, (const _Float16*)in, (const uint4*)wfrag, bias,
                       (_Float16*)out, partial, H, W);
  else
    hipLaunchKernelGGL((conv7_kernel<R, false>), grid, dim3(256), smem, st, (const _Float16*)in, (const uint4*)wfrag, bias,
                       (_Float16*)out, partial, H, W);
  return check_launch("conv7x7");
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
