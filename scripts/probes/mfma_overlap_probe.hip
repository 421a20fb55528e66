// Do MFMA and ordinary VALU instructions overlap on a CDNA4 SIMD?  One wave per SIMD (and, second run, two), three loops
// each: MFMAs only, VALU FMAs only, both interleaved -- for the fp32 MFMA (v_mfma_f32_16x16x4_f32) and the bf16 one
// (v_mfma_f32_16x16x16_bf16).  overlap = (t_mfma + t_valu - t_both) / min(t_mfma, t_valu): 1 = fully hidden, 0 = serial.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/mfma_overlap_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x4 = __attribute__((ext_vector_type(4))) short;

template <int MODE, bool BF16>  // MODE bit0: MFMA, bit1: VALU
__global__ __launch_bounds__(64) void probe(float* out, int iters) {
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = threadIdx.x * 0.001f + k;
  const float a = 1.0001f, b = 0.5f + threadIdx.x * 1e-6f;
  bf16x4 ab = {(short)0x3f80, (short)0x3f80, (short)0x3f80, (short)0x3f80};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE & 1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if constexpr (BF16) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ab, ab, acc[q], 0, 0, 0);
          else acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[q], 0, 0, 0);
        }
      }
      if (MODE & 2) {
#pragma unroll
        for (int r = 0; r < (BF16 ? 1 : 4); ++r)  // fp32 MFMA: 32 cycles each -> 4x the VALU work per MFMA
#pragma unroll
          for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(a), "v"(b));
      }
    }
  }
  float s = 0.f;
  for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int MODE, bool BF16>
float run(int blocks, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  probe<MODE, BF16><<<blocks, 64>>>(out, iters);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  probe<MODE, BF16><<<blocks, 64>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3f;
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 64 * sizeof(float));
  const int iters = 20000;
  for (int wps = 1; wps <= 2; ++wps) {
    const int blocks = 256 * 4 * wps;  // one / two single-wave workgroups per SIMD
    const float m32 = run<1, false>(blocks, iters, out), v32 = run<2, false>(blocks, iters, out), b32 = run<3, false>(blocks, iters, out);
    const float m16 = run<1, true>(blocks, iters, out), v16 = run<2, true>(blocks, iters, out), b16 = run<3, true>(blocks, iters, out);
    printf("%d wave(s)/SIMD  fp32 MFMA: mfma %.0f us, valu %.0f us, both %.0f us -> overlap %.2f\n", wps, m32, v32, b32,
           (m32 + v32 - b32) / (m32 < v32 ? m32 : v32));
    printf("%d wave(s)/SIMD  bf16 MFMA: mfma %.0f us, valu %.0f us, both %.0f us -> overlap %.2f\n", wps, m16, v16, b16,
           (m16 + v16 - b16) / (m16 < v16 ? m16 : v16));
    printf("   cycles per MFMA at 2.4 GHz: fp32 %.1f, bf16 %.1f\n", m32 * 2400.f / (iters * 16.f) / wps, m16 * 2400.f / (iters * 16.f) / wps);
  }
  return 0;
}
