// Wave-wide (64-lane) prefix scans and reductions for the one-wave-per-ray kernels (losses.hip, train_fused.hip,
// composite.hip, the PDF sampler): DPP row shifts inside the four 16-lane rows -- VALU rate -- and three v_readlane for the
// row totals, instead of six dependent ds_bpermute per scan (what __shfl_up compiles to: each a round trip through the LDS
// crossbar, and these kernels are chains of such scans: the interlevel loss runs six fp64 ones per ray).
// The association of the sums differs from a Hillis-Steele scan only across rows (prefix = total of the rows below + in-row
// prefix); every caller is an fp32 / fp64 sum or product with tolerances, none needs a particular association.
#pragma once
#include <hip/hip_runtime.h>

namespace nrhip {
namespace wscan {

template <int CTRL>
__device__ __forceinline__ int dpp(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);  // lanes without a source keep `old`
}
template <int CTRL>
__device__ __forceinline__ float dpp(float old, float v) {
  return __builtin_bit_cast(float, dpp<CTRL>(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v)));
}
template <int CTRL>
__device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t v) {
  return (uint32_t)dpp<CTRL>((int)old, (int)v);
}
template <int CTRL>
__device__ __forceinline__ double dpp(double old, double v) {
  const long long o = __builtin_bit_cast(long long, old), x = __builtin_bit_cast(long long, v);
  const uint32_t lo = (uint32_t)dpp<CTRL>((int)(uint32_t)o, (int)(uint32_t)x);
  const uint32_t hi = (uint32_t)dpp<CTRL>((int)(uint32_t)(o >> 32), (int)(uint32_t)(x >> 32));
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ float lane_of(float v, int l) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ uint32_t lane_of(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ double lane_of(double v, int l) {
  const long long x = __builtin_bit_cast(long long, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)x, l);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(x >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

struct Add {
  template <class T>
  static __device__ __forceinline__ T id() { return T(0); }
  template <class T>
  static __device__ __forceinline__ T op(T a, T b) { return a + b; }
};
struct Mul {
  template <class T>
  static __device__ __forceinline__ T id() { return T(1); }
  template <class T>
  static __device__ __forceinline__ T op(T a, T b) { return a * b; }
};

// inclusive scan over the 64 lanes, lane 0 first
template <class Op, class T>
__device__ __forceinline__ T incl(T v, int lane) {
  const T e = Op::template id<T>();
  v = Op::op(dpp<0x111>(e, v), v);  // row_shr:1
  v = Op::op(dpp<0x112>(e, v), v);  // row_shr:2
  v = Op::op(dpp<0x114>(e, v), v);  // row_shr:4
  v = Op::op(dpp<0x118>(e, v), v);  // row_shr:8
  const T t0 = lane_of(v, 15), t1 = lane_of(v, 31), t2 = lane_of(v, 47);
  const T t01 = Op::op(t0, t1);
  const int row = lane >> 4;
  const T below = row == 0 ? e : (row == 1 ? t0 : (row == 2 ? t01 : Op::op(t01, t2)));
  return Op::op(below, v);
}
// suffix (reverse) inclusive scan: lane i = op over lanes i..63
template <class Op, class T>
__device__ __forceinline__ T rincl(T v, int lane) {
  const T e = Op::template id<T>();
  v = Op::op(v, dpp<0x101>(e, v));  // row_shl:1
  v = Op::op(v, dpp<0x102>(e, v));
  v = Op::op(v, dpp<0x104>(e, v));
  v = Op::op(v, dpp<0x108>(e, v));
  const T s1 = lane_of(v, 16), s2 = lane_of(v, 32), s3 = lane_of(v, 48);
  const T s23 = Op::op(s2, s3);
  const int row = lane >> 4;
  const T above = row == 3 ? e : (row == 2 ? s3 : (row == 1 ? s23 : Op::op(s1, s23)));
  return Op::op(v, above);
}
// the value of lane - 1 (lane 0: `first`): exclusive scans from inclusive ones
template <class T>
__device__ __forceinline__ T shift_up1(T v, T first, int lane) {
  T x = dpp<0x111>(first, v);  // in-row; lanes 16, 32, 48 take the last lane of the row below
  const T a = lane_of(v, 15), b = lane_of(v, 31), c = lane_of(v, 47);
  x = lane == 16 ? a : x;
  x = lane == 32 ? b : x;
  x = lane == 48 ? c : x;
  return x;
}
// reduction over the 64 lanes, the same value in every lane
template <class Op, class T>
__device__ __forceinline__ T reduce(T v) {
  const T e = Op::template id<T>();
  v = Op::op(dpp<0x111>(e, v), v);
  v = Op::op(dpp<0x112>(e, v), v);
  v = Op::op(dpp<0x114>(e, v), v);
  v = Op::op(dpp<0x118>(e, v), v);
  return Op::op(Op::op(lane_of(v, 15), lane_of(v, 31)), Op::op(lane_of(v, 47), lane_of(v, 63)));
}
template <class T>
__device__ __forceinline__ T last(T v) { return lane_of(v, 63); }
template <class T>
__device__ __forceinline__ T first(T v) { return lane_of(v, 0); }

}  // namespace wscan
}  // namespace nrhip
