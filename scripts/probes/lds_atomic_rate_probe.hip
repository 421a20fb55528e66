// LDS atomic throughput on gfx950 by operand width and return mode: is `ds_add_u64` (what bin_reduce accumulates with)
// slower per lane than `ds_add_u32`?  One 1024-thread workgroup per CU, random addresses in a 128 KB image.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/lds_atomic_rate_probe.hip -o /tmp/lds_probe && /tmp/lds_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(1024) void probe(uint32_t* out, int iters) {
  extern __shared__ unsigned long long img[];
  for (int i = threadIdx.x; i < 16384; i += 1024) img[i] = 0;
  __syncthreads();
  uint32_t s = threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;
    const uint32_t a = (s >> 8);
    if (MODE == 0) atomicAdd(reinterpret_cast<uint32_t*>(img) + (a & 32767u), 1u);                      // u32, no return
    if (MODE == 1) atomicAdd(img + (a & 16383u), 1ull);                                                  // u64, no return
    if (MODE == 2) acc += atomicAdd(reinterpret_cast<uint32_t*>(img) + (a & 32767u), 1u);               // u32, returning
    if (MODE == 3) acc += (uint32_t)atomicAdd(img + (a & 16383u), 1ull);                                 // u64, returning
    if (MODE == 5) {  // bin_reduce<4>'s entry-major image: 4 features of a random entry, one instruction per feature
      const uint32_t key = a & 4095u;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(img + key * 4 + j, (unsigned long long)(s >> j));
    }
    if (MODE == 6) {  // feature-major image
      const uint32_t key = a & 4095u;
#pragma unroll
      for (int j = 0; j < 4; ++j) atomicAdd(img + j * 4096 + key, (unsigned long long)(s >> j));
    }
    if (MODE == 7) {  // entry-major, all four lanes' features from 16 lanes x 4 features (lane = entry * 4 + feature)
      const uint32_t key = __shfl(a, threadIdx.x & ~3u) & 4095u;
      atomicAdd(img + key * 4 + (threadIdx.x & 3u), (unsigned long long)s);
    }
    if (MODE == 4) {                                                                                     // two u32 (lo, hi)
      atomicAdd(reinterpret_cast<uint32_t*>(img) + (a & 16383u), 1u);
      atomicAdd(reinterpret_cast<uint32_t*>(img) + 16384u + (a & 16383u), 1u);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = acc + (uint32_t)img[5];
}

template <int MODE>
float run(uint32_t* out, int iters) {
  hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b;
  hipEventCreate(&a), hipEventCreate(&b);
  probe<MODE><<<256, 1024, 131072>>>(out, 64);
  hipEventRecord(a);
  probe<MODE><<<256, 1024, 131072>>>(out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  return ms;
}

int main() {
  uint32_t* out;
  hipMalloc(&out, 4096);
  const int iters = 4096;
  const char* names[8] = {"ds_add_u32", "ds_add_u64", "ds_add_rtn_u32", "ds_add_rtn_u64", "2 x ds_add_u32",
                          "4 x u64 entry-major", "4 x u64 feature-major", "u64, 4 lanes per entry"};
  float ms[8] = {run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters), run<4>(out, iters),
                 run<5>(out, iters), run<6>(out, iters), run<7>(out, iters)};
  for (int m = 0; m < 8; ++m)
    printf("%-24s %8.3f ms  -> %.2f lane-updates / ns / CU-image (1024 lanes x %d per CU)\n", names[m], ms[m],
           1024.0 * iters / (ms[m] * 1e6), iters);
  return 0;
}
