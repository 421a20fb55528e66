// Do an fp32-MFMA wave and a gather (or VALU) wave that share a SIMD overlap on gfx950?
// One 512-thread workgroup per CU slot: waves 0-3 (one per SIMD) run a chain of v_mfma_f32_16x16x4_f32 on 4
// independent accumulators; waves 4-7 run either random 8-byte gathers (8 in flight) or a dependent-free VALU stream.
// mode bit 0: MFMA waves active, bit 1: second role active; role = 0 gather, 1 VALU.
// build: hipcc --offload-arch=gfx950 -O3 -o simd_overlap simd_overlap.hip ; run: ./simd_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
using f32x4 = __attribute__((ext_vector_type(4))) float;

__global__ __launch_bounds__(512) void k(const float2* __restrict__ table, uint32_t mask, int iters_m, int iters_o,
                                         int mode, int role, float* sink) {
  int wid = threadIdx.x >> 6;
  const int lane = threadIdx.x & 63;
  if (mode & 8) wid ^= 4;                                   // swap roles: the "other" role gets the older waves
  if ((mode & 4) && wid >= 4) __builtin_amdgcn_s_setprio(3);  // raise the priority of the non-MFMA role
  if (wid < 4) {
    if (!(mode & 1)) return;
    f32x4 a0{0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = lane * 1e-3f, y = 1.f + lane * 1e-4f;
    for (int i = 0; i < iters_m; ++i) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
        if (mode & 16) __builtin_amdgcn_s_sleep(1);
        if (mode & 32) asm volatile("s_nop 0");
        if (mode & 64) { __builtin_amdgcn_s_setprio(0); }
      }
    }
    if (a0[0] + a1[1] + a2[2] + a3[3] == 123.456f) sink[0] = 1.f;
    return;
  }
  if (!(mode & 2)) return;
  if (role == 0) {
    uint32_t h = (blockIdx.x * 512 + threadIdx.x) * 2654435761u;
    float acc = 0.f;
    for (int i = 0; i < iters_o; ++i) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        h = h * 1664525u + 1013904223u;
        v[u] = table[(h >> 7) & mask];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u].x * v[u].y;
    }
    if (acc == 123.456f) sink[1] = acc;
  } else {
    float a = lane, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int i = 0; i < iters_o; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        a = fmaf(a, b, c);
        d = fmaf(d, b, a);
        c = fmaf(c, b, d);
        b = fmaf(b, 0.999f, 1e-6f);
      }
    }
    if (a + d + c == 123.456f) sink[2] = a;
  }
}

static float run(const float2* t, uint32_t mask, int im, int io, int mode, int role, float* sink, int blocks) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<<<blocks, 512>>>(t, mask, im, io, mode, role, sink);
  hipEventRecord(e0);
  for (int r = 0; r < 3; ++r) k<<<blocks, 512>>>(t, mask, im, io, mode, role, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3 * 1e3f;
}

int main() {
  const size_t n = (size_t)1 << 23;  // 64 MB of float2
  float2* t;
  float* sink;
  hipMalloc(&t, n * sizeof(float2));
  hipMalloc(&sink, 64);
  hipMemset(t, 0, n * sizeof(float2));
  const int blocks = 256;  // one workgroup per CU: 2 waves per SIMD (one of each role)
  const int im = 2000, iog = 120, iov = 1500;
  for (int role = 0; role < 2; ++role) {
    const int io = role == 0 ? iog : iov;
    const float m = run(t, (uint32_t)(n - 1), im, io, 1, role, sink, blocks);
    const float o = run(t, (uint32_t)(n - 1), im, io, 2, role, sink, blocks);
    const float b = run(t, (uint32_t)(n - 1), im, io, 3, role, sink, blocks);
    printf("%s: mfma-only %.1f us, %s-only %.1f us, both %.1f us  (sum %.1f, max %.1f)\n", role == 0 ? "gather" : "valu", m,
           role == 0 ? "gather" : "valu", o, b, m + o, m > o ? m : o);
  }
  for (int role = 0; role < 2; ++role) {
    const int io = role == 0 ? iog : iov;
    printf("%s + mfma: plain %.1f us, other role at s_setprio 3: %.1f us, other role in the older waves: %.1f us\n",
           role == 0 ? "gather" : "valu", run(t, (uint32_t)(n - 1), im, io, 3, role, sink, blocks),
           run(t, (uint32_t)(n - 1), im, io, 3 | 4, role, sink, blocks), run(t, (uint32_t)(n - 1), im, io, 3 | 8, role, sink, blocks));
  }
  for (int y = 16; y <= 64; y *= 2)
    for (int role = 0; role < 2; ++role) {
      const int io = role == 0 ? iog : iov;
      printf("yield %s (%s): mfma-only %.1f us, both %.1f us\n", y == 16 ? "s_sleep 1" : y == 32 ? "s_nop 0" : "s_setprio 0",
             role == 0 ? "gather" : "valu", run(t, (uint32_t)(n - 1), im, io, 1 | y, role, sink, blocks),
             run(t, (uint32_t)(n - 1), im, io, 3 | y, role, sink, blocks));
    }
  // the same with TWO workgroups per CU slot pair (4 waves per SIMD)
  for (int role = 0; role < 2; ++role) {
    const int io = role == 0 ? iog : iov;
    const float m = run(t, (uint32_t)(n - 1), im, io, 1, role, sink, 512);
    const float o = run(t, (uint32_t)(n - 1), im, io, 2, role, sink, 512);
    const float b = run(t, (uint32_t)(n - 1), im, io, 3, role, sink, 512);
    printf("2 WG/CU %s: mfma-only %.1f us, other-only %.1f us, both %.1f us  (sum %.1f, max %.1f)\n",
           role == 0 ? "gather" : "valu", m, o, b, m + o, m > o ? m : o);
  }
  return 0;
}
