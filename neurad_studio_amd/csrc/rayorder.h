// The ordering pass of rayorder.hip as a device function, so that the PowerSampler launch can carry it along
// (sampler.hip: nrhip_power_sampler_ordered) -- see rayorder.hip for what it computes and why.
#pragma once
#include "common.h"
#include "wave_scan.h"

namespace nrhip {

constexpr int kOrderThreads = 1024;  // BITS per axis: 4 -> 4096 buckets (16 KB of LDS), 5 -> 32768 buckets (128 KB)
constexpr int kOrderCached = 8;                     // keys a thread keeps in registers between the two passes

__device__ __forceinline__ uint32_t spread3(uint32_t v) {  // up to 5 bits -> every third bit
  return (v & 1u) | ((v & 2u) << 2) | ((v & 4u) << 4) | ((v & 8u) << 6) | ((v & 16u) << 8);
}

template <int BITS>
__device__ __forceinline__ uint32_t order_key(const float* __restrict__ o, const float* __restrict__ d, int64_t i,
                                              float t_ref, float scale) {
  const float x = o[3 * i] + d[3 * i] * t_ref, y = o[3 * i + 1] + d[3 * i + 1] * t_ref,
              z = o[3 * i + 2] + d[3 * i + 2] * t_ref;
  const SamplePos p = contract_gaussian(x, y, z, 0.f, scale);
  const float q = (float)(1 << BITS);
  auto cell = [q](float v) {
    const float c = fminf(fmaxf(v * q, 0.f), q - 1.f);  // NaN -> 0 through fmaxf
    return (uint32_t)(int)c;
  };
  return spread3(cell(p.x)) | (spread3(cell(p.y)) << 1) | (spread3(cell(p.z)) << 2);
}

// the whole pass, executed by ONE workgroup of kOrderThreads threads (all of them must call it)
template <int BITS>
__device__ __forceinline__ void ray_order_body(const float* __restrict__ o, const float* __restrict__ d, int64_t n,
                                               float t_ref, float scale, int32_t* __restrict__ order) {
  constexpr int kOrderKeys = 1 << (3 * BITS);
  __shared__ uint32_t hist[kOrderKeys];
  __shared__ uint32_t wave_tot[kOrderThreads / 64];
  const int tid = threadIdx.x;
  // (the whole pass is a latency chain on one CU in front of the render kernel: the rays' global loads are issued before the
  //  histogram is cleared, so the clear and its barrier run under them)
  uint32_t cached[kOrderCached];  // batches up to 8192 rays: the second pass needs no global reads
#pragma unroll
  for (int it = 0; it < kOrderCached; ++it) {
    const int64_t i = tid + (int64_t)it * kOrderThreads;
    cached[it] = i < n ? order_key<BITS>(o, d, i, t_ref, scale) : 0u;
  }
  for (int k = tid; k < kOrderKeys; k += kOrderThreads) hist[k] = 0;
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kOrderCached; ++it) {
    const int64_t i = tid + (int64_t)it * kOrderThreads;
    if (i < n) atomicAdd(&hist[cached[it]], 1u);
  }
  for (int64_t i = tid + (int64_t)kOrderCached * kOrderThreads; i < n; i += kOrderThreads)
    atomicAdd(&hist[order_key<BITS>(o, d, i, t_ref, scale)], 1u);
  __syncthreads();
  // exclusive scan of the bins: PER consecutive bins per thread, wave scan, scan of the 16 wave totals
  constexpr int PER = kOrderKeys / kOrderThreads;
  uint32_t v[PER], sum = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) v[k] = hist[PER * tid + k], sum += v[k];
  const uint32_t incl = wscan::incl<wscan::Add>(sum, tid & 63);  // DPP row shifts + 3 readlanes, not 6 ds_bpermute
  if ((tid & 63) == 63) wave_tot[tid >> 6] = incl;
  __syncthreads();
  uint32_t base = incl - sum;
#pragma unroll
  for (int w = 0; w < kOrderThreads / 64; ++w) base += w < (tid >> 6) ? wave_tot[w] : 0u;  // (independent broadcast reads)
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    hist[PER * tid + k] = base;
    base += v[k];
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < kOrderCached; ++it) {
    const int64_t i = tid + (int64_t)it * kOrderThreads;
    if (i < n) order[atomicAdd(&hist[cached[it]], 1u)] = (int32_t)i;
  }
  for (int64_t i = tid + (int64_t)kOrderCached * kOrderThreads; i < n; i += kOrderThreads)
    order[atomicAdd(&hist[order_key<BITS>(o, d, i, t_ref, scale)], 1u)] = (int32_t)i;
}

}  // namespace nrhip
