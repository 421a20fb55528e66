// the optimizer step of the hash tables.
//
// neurad-studio trains its tables with torch.optim.Adam(lr=1e-2, eps=1e-15) -- dense, over ~150 M parameters
// (configs/method_configs.py:423-426, engine/optimizers.py:168-181).  This is that update as ONE streaming HIP kernel
// per table, torch.optim.Adam's arithmetic (same operation order as its fused implementation, fp32):
//     m = m + (g - m) * (1 - b1);  v = b2 * v + (1 - b2) * g * g;
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          [AdamW: p *= 1 - lr * wd first]
// plus the one thing a hash table allows: rows that have NEVER received a gradient have g = m = v = 0, for which the
// update is exactly a no-op -- the kernel reads (g, m, v), sees three zeros and touches nothing else (no read of p, no
// writes): 12 instead of 28 bytes per such element, with bit-identical results.  Coarse levels (res^3 << T: most of
// their hashed rows are never addressed) and sparsely observed scenes are mostly such rows.
// 16-byte accesses, grid-stride, no atomics; the state (step, exp_avg, exp_avg_sq) stays in torch.optim.Adam's
// state_dict layout (neurad_studio_amd/optim.py), so checkpoints interchange.
#include <hip/hip_fp16.h>

#include "common.h"

namespace nrhip {

struct AdamArgs {
  float step_size;     // lr / (1 - b1^t)
  float b2;
  float omb1, omb2;    // 1 - b1, 1 - b2 rounded from double like torch's Python scalars (1.f - 0.999f is 1.3e-5 off)
  float inv_bc2_sqrt;  // 1 / sqrt(1 - b2^t)
  float eps;
  float decay;         // 1 - lr * weight_decay (decoupled), 1 = off
  float grad_scale;    // multiplies the gradient first (e.g. 1 / GradScaler scale), 1 = off
};

__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, const AdamArgs& a) {
  g *= a.grad_scale;
  m = fmaf(g - m, a.omb1, m);
  v = fmaf(a.b2, v, a.omb2 * g * g);
  const float denom = sqrtf(v) * a.inv_bc2_sqrt + a.eps;
  p = p * a.decay - a.step_size * (m / denom);
}

// One tensor of a launch.  param is the fp32 tensor the update runs on (an fp32 table itself, or the fp32 MASTER copy of an
// fp16-storage table); image (optional) is the fp16 table: the rounded new value is written there in the same pass.  The
// gradient is fp32 or fp16 (autograd hands an fp16 parameter an fp16 gradient): converted in registers, no .float() pass.
struct AdamTensor {
  float* p;
  const void* g;
  float* m;
  float* v;
  __half* image;
  int64_t n;
  AdamArgs a;        // per tensor: untouched tables keep their step count, so bias corrections differ
  int32_t grad_half;
  int32_t block0;    // first workgroup of this tensor in a multi-tensor launch
};

template <bool GH>
__device__ __forceinline__ float4 load_grad4(const void* g, int64_t i) {
  if constexpr (GH) {
    const uint2 raw = reinterpret_cast<const uint2*>(g)[i];
    const float2 lo = __half22float2(*reinterpret_cast<const __half2*>(&raw.x));
    const float2 hi = __half22float2(*reinterpret_cast<const __half2*>(&raw.y));
    return make_float4(lo.x, lo.y, hi.x, hi.y);
  } else {
    return reinterpret_cast<const float4*>(g)[i];
  }
}

template <bool GH>
__device__ __forceinline__ void adam_tensor(const AdamTensor& t, int64_t first, int64_t stride) {
  const int64_t n4 = t.n >> 2;
  for (int64_t i = first; i < n4; i += stride) {
    const float4 gv = load_grad4<GH>(t.g, i);
    float4 mv = reinterpret_cast<const float4*>(t.m)[i];
    float4 vv = reinterpret_cast<const float4*>(t.v)[i];
    const bool dead = gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f && mv.x == 0.f && mv.y == 0.f &&
                      mv.z == 0.f && mv.w == 0.f && vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f;
    if (dead && t.a.decay == 1.f) continue;  // exact no-op: never-touched rows cost three reads
    float4 pv = reinterpret_cast<float4*>(t.p)[i];
    adam_elem(pv.x, gv.x, mv.x, vv.x, t.a);
    adam_elem(pv.y, gv.y, mv.y, vv.y, t.a);
    adam_elem(pv.z, gv.z, mv.z, vv.z, t.a);
    adam_elem(pv.w, gv.w, mv.w, vv.w, t.a);
    reinterpret_cast<float4*>(t.p)[i] = pv;
    reinterpret_cast<float4*>(t.m)[i] = mv;
    reinterpret_cast<float4*>(t.v)[i] = vv;
    if (t.image) {  // the fp16 table = the rounded master copy (round to nearest even, like Tensor.copy_)
      const __half2 lo = __floats2half2_rn(pv.x, pv.y), hi = __floats2half2_rn(pv.z, pv.w);
      uint2 raw;
      raw.x = *reinterpret_cast<const uint32_t*>(&lo), raw.y = *reinterpret_cast<const uint32_t*>(&hi);
      reinterpret_cast<uint2*>(t.image)[i] = raw;
    }
  }
  if (first < (t.n & 3)) {  // tail elements, one lane each
    const int64_t i = (n4 << 2) + first;
    float g;
    if constexpr (GH) g = __half2float(reinterpret_cast<const __half*>(t.g)[i]);
    else g = reinterpret_cast<const float*>(t.g)[i];
    adam_elem(t.p[i], g, t.m[i], t.v[i], t.a);
    if (t.image) t.image[i] = __float2half_rn(t.p[i]);
  }
}

__global__ __launch_bounds__(256) void adam_kernel(AdamTensor t) {
  const int64_t first = (int64_t)blockIdx.x * 256 + threadIdx.x, stride = (int64_t)gridDim.x * 256;
  if (t.grad_half) adam_tensor<true>(t, first, stride);
  else adam_tensor<false>(t, first, stride);
}

// Many small tensors (the per-actor grids: 32 + 32 tables of 0.1 - 2 M elements) in ONE launch: the descriptors ride in the
// kernel arguments, workgroup b serves the tensor whose [block0, next block0) range contains it.
constexpr int kAdamMany = 24;  // descriptors per launch (24 x 104 B < the 4 KB kernel-argument limit)
struct AdamMany {
  AdamTensor t[kAdamMany];
  int32_t count;
  int32_t total_blocks;
};

__global__ __launch_bounds__(256) void adam_many_kernel(AdamMany many) {
  int k = 0;
  while (k + 1 < many.count && (int)blockIdx.x >= many.t[k + 1].block0) ++k;  // (block-uniform; <= 24 steps)
  const AdamTensor& t = many.t[k];
  const int nblk = (k + 1 < many.count ? many.t[k + 1].block0 : many.total_blocks) - t.block0;
  const int64_t first = (int64_t)((int)blockIdx.x - t.block0) * 256 + threadIdx.x, stride = (int64_t)nblk * 256;
  if (t.grad_half) adam_tensor<true>(t, first, stride);
  else adam_tensor<false>(t, first, stride);
}

// ---- device-controlled form: torch.amp.GradScaler's optimizer protocol + HIP-graph capture ------------------------------
// engine/trainer.py:550-576 drives every optimizer through GradScaler.step(); an optimizer that sets
// `_step_supports_amp_scaling` receives the scale S and the found-inf flag as DEVICE tensors (torch/amp/grad_scaler.py: what
// torch's fused Adam consumes) and must, without a host read: leave everything untouched when found_inf != 0 (step counts
// included), otherwise divide the gradients by S inside the update.  The same form serves a captured step (HIP graph): the
// step counts live on the device (torch's capturable=True layout: one fp32 scalar per tensor), the learning rate may be a
// device scalar (a scheduler fills it between replays), so a replay needs no new kernel arguments.
// `adam_prepare_kernel` (one thread per tensor) advances the counts and derives each tensor's AdamArgs in double, rounded
// once, exactly like make_args on the host; the streaming kernel reads them from the workspace.
struct AdamCtl {
  AdamArgs a;
  int32_t skip;  // found_inf != 0: this launch must not touch anything
  int32_t pad;
};

struct AdamPrepare {
  float* step[kAdamMany];
  AdamCtl* ctl;  // [count]
  const float* lr_dev;
  const float* grad_scale;
  const float* found_inf;
  double lr, beta1, beta2, eps, weight_decay, host_grad_scale;
  int32_t count;
};

__global__ void adam_prepare_kernel(AdamPrepare pr) {
  const int k = threadIdx.x;
  if (k >= pr.count) return;
  AdamCtl c;
  c.pad = 0;
  c.skip = (pr.found_inf && *pr.found_inf != 0.f) ? 1 : 0;
  double step = (double)*pr.step[k];
  if (!c.skip) {
    step += 1.0;
    *pr.step[k] = (float)step;
  }
  if (step < 1.0) step = 1.0;  // (skipped very first step: the arguments are never used)
  const double lr = pr.lr_dev ? (double)*pr.lr_dev : pr.lr;
  const double bc1 = 1.0 - pow(pr.beta1, step), bc2 = 1.0 - pow(pr.beta2, step);
  c.a.step_size = (float)(lr / bc1);
  c.a.b2 = (float)pr.beta2;
  c.a.omb1 = (float)(1.0 - pr.beta1), c.a.omb2 = (float)(1.0 - pr.beta2);
  c.a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  c.a.eps = (float)pr.eps;
  c.a.decay = (float)(1.0 - lr * pr.weight_decay);
  // GradScaler's scales are powers of two (init 2^16, growth 2, backoff 1/2): the reciprocal is exact, g * (1/S) == g / S
  c.a.grad_scale = (float)(pr.host_grad_scale / (pr.grad_scale ? (double)*pr.grad_scale : 1.0));
  pr.ctl[k] = c;
}

struct AdamManyDev {
  AdamTensor t[kAdamMany];  // (t[k].a unused: the arguments come from ctl[k])
  const AdamCtl* ctl;
  int32_t count;
  int32_t total_blocks;
};

__global__ __launch_bounds__(256) void adam_many_dev_kernel(AdamManyDev many) {
  int k = 0;
  while (k + 1 < many.count && (int)blockIdx.x >= many.t[k + 1].block0) ++k;
  const AdamCtl c = many.ctl[k];
  if (c.skip) return;
  AdamTensor t = many.t[k];
  t.a = c.a;
  const int nblk = (k + 1 < many.count ? many.t[k + 1].block0 : many.total_blocks) - t.block0;
  const int64_t first = (int64_t)((int)blockIdx.x - t.block0) * 256 + threadIdx.x, stride = (int64_t)nblk * 256;
  if (t.grad_half) adam_tensor<true>(t, first, stride);
  else adam_tensor<false>(t, first, stride);
}

// ---- GradScaler's inf check over the table gradients, read-only --------------------------------------------------------
// GradScaler.step runs `_check_inf_per_device` on every optimizer that takes the scale itself (torch/amp/grad_scaler.py):
// torch's `_amp_foreach_non_finite_check_and_unscale_` with a scale of 1 -- a read AND a write of every gradient element
// (1.2 GB per c3 step for 600 MB of table gradients, 0.33 ms).  The check alone is a read: an element is non-finite iff its
// exponent field is all ones; adding one exponent unit carries into the sign position exactly then, so the test of four (or
// eight fp16) elements is three integer ops per dword, OR-ed over the thread's share; any hit stores 1.0f (never cleared
// here: several launches and several optimizers' tensors accumulate into the same flag, as in torch).
constexpr int kCheckMany = 24;
struct CheckTensor {
  const void* p;
  int64_t n;
  int32_t half;
  int32_t block0;
};
struct CheckMany {
  CheckTensor t[kCheckMany];
  float* found;
  int32_t count;
  int32_t total_blocks;
};

__device__ __forceinline__ uint32_t nonfinite_bits(uint32_t x, bool half) {
  return half ? (((x & 0x7c007c00u) + 0x04000400u) & 0x80008000u) : (((x & 0x7f800000u) + 0x00800000u) & 0x80000000u);
}

__global__ __launch_bounds__(256) void nonfinite_check_kernel(CheckMany many) {
  int k = 0;
  while (k + 1 < many.count && (int)blockIdx.x >= many.t[k + 1].block0) ++k;
  const CheckTensor t = many.t[k];
  const int nblk = (k + 1 < many.count ? many.t[k + 1].block0 : many.total_blocks) - t.block0;
  const int64_t first = (int64_t)((int)blockIdx.x - t.block0) * 256 + threadIdx.x, stride = (int64_t)nblk * 256;
  const bool half = t.half != 0;
  const int per16 = half ? 8 : 4;  // elements per 16-byte load
  const int64_t n16 = t.n / per16;
  const uint4* p = reinterpret_cast<const uint4*>(t.p);
  uint32_t bad = 0;
  int64_t i = first;
  for (; i + 3 * stride < n16; i += 4 * stride) {  // four independent loads in flight per lane
    const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
    bad |= nonfinite_bits(a.x, half) | nonfinite_bits(a.y, half) | nonfinite_bits(a.z, half) | nonfinite_bits(a.w, half);
    bad |= nonfinite_bits(b.x, half) | nonfinite_bits(b.y, half) | nonfinite_bits(b.z, half) | nonfinite_bits(b.w, half);
    bad |= nonfinite_bits(c.x, half) | nonfinite_bits(c.y, half) | nonfinite_bits(c.z, half) | nonfinite_bits(c.w, half);
    bad |= nonfinite_bits(d.x, half) | nonfinite_bits(d.y, half) | nonfinite_bits(d.z, half) | nonfinite_bits(d.w, half);
  }
  for (; i < n16; i += stride) {
    const uint4 a = p[i];
    bad |= nonfinite_bits(a.x, half) | nonfinite_bits(a.y, half) | nonfinite_bits(a.z, half) | nonfinite_bits(a.w, half);
  }
  if (first < t.n - n16 * per16) {  // tail elements, one lane each
    const int64_t e = n16 * per16 + first;
    if (half) bad |= nonfinite_bits(reinterpret_cast<const uint16_t*>(t.p)[e], true);
    else bad |= nonfinite_bits(reinterpret_cast<const uint32_t*>(t.p)[e], false);
  }
  if (bad) *many.found = 1.0f;
}

}  // namespace nrhip

using namespace nrhip;

namespace {

int make_args(const char* what, int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay,
              double grad_scale, AdamArgs* a) {
  NR_REQUIRE(step >= 1, NRHIP_ERR_INVALID_ARG, "%s: step >= 1 required", what);
  NR_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., NRHIP_ERR_INVALID_ARG,
             "%s: bad hyper-parameter", what);
  // hyper-parameters arrive as doubles and every derived scalar is formed in double, then rounded once -- exactly what
  // torch does with its Python floats
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a->step_size = (float)(lr / bc1);
  a->b2 = (float)beta2;
  a->omb1 = (float)(1.0 - beta1), a->omb2 = (float)(1.0 - beta2);
  a->inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a->eps = (float)eps;
  a->decay = (float)(1.0 - lr * weight_decay);
  a->grad_scale = (float)grad_scale;
  return NRHIP_OK;
}

int blocks_for(int64_t n) {
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride: 16 workgroups per CU
  return (int)blocks;
}

int check_tensor(const char* what, const float* param, const void* grad, const float* m, const float* v, const void* image) {
  NR_REQUIRE(param && grad && m && v, NRHIP_ERR_INVALID_ARG, "%s: NULL pointer", what);
  NR_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(m) |
               reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(image)) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "%s: tensors must be 16-byte aligned", what);
  return NRHIP_OK;
}

}  // namespace

extern "C" int nrhip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                               double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                               void* stream) {
  NR_REQUIRE(n >= 0 && step >= 1, NRHIP_ERR_INVALID_ARG, "adam_step: n >= 0 and step >= 1 required");
  if (n == 0) return NRHIP_OK;
  if (int e = check_tensor("adam_step", param, grad, exp_avg, exp_avg_sq, nullptr)) return e;
  AdamTensor t{param, grad, exp_avg, exp_avg_sq, nullptr, n, {}, 0, 0};
  if (int e = make_args("adam_step", step, lr, beta1, beta2, eps, weight_decay, grad_scale, &t.a)) return e;
  adam_kernel<<<blocks_for(n), 256, 0, (hipStream_t)stream>>>(t);
  return check_launch("adam_step");
}

extern "C" int nrhip_adam_step_many(const nrhip_adam_tensor* tensors, int32_t n_tensors, double lr, double beta1, double beta2,
                                    double eps, double weight_decay, double grad_scale, void* stream) {
  NR_REQUIRE(n_tensors >= 0 && (tensors || n_tensors == 0), NRHIP_ERR_INVALID_ARG, "adam_step_many: bad argument");
  int k = 0;
  while (k < n_tensors) {
    AdamMany many;
    many.count = 0;
    int blocks = 0;
    int live = 0;  // non-empty tensors this launch will hold
    for (int j = k; j < n_tensors && live < kAdamMany; ++j) live += tensors[j].n > 0 ? 1 : 0;
    for (; k < n_tensors && many.count < kAdamMany; ++k) {
      const nrhip_adam_tensor& in = tensors[k];
      NR_REQUIRE(in.n >= 0 && (in.grad_dtype == 0 || in.grad_dtype == 1), NRHIP_ERR_INVALID_ARG,
                 "adam_step_many: tensor %d: n >= 0, grad_dtype 0 (fp32) or 1 (fp16)", k);
      if (in.n == 0) continue;
      if (int e = check_tensor("adam_step_many", in.param, in.grad, in.exp_avg, in.exp_avg_sq, in.image_fp16)) return e;
      AdamTensor& t = many.t[many.count++];
      t = AdamTensor{in.param, in.grad, in.exp_avg, in.exp_avg_sq, reinterpret_cast<__half*>(in.image_fp16), in.n, {},
                     in.grad_dtype, blocks};
      if (int e = make_args("adam_step_many", in.step, lr, beta1, beta2, eps, weight_decay, grad_scale, &t.a)) return e;
      int b = blocks_for(in.n);
      // many tensors share the machine: 4 workgroups per CU each at most; a tensor alone in its launch (the large tables
      // HashGridAdam sends one by one) keeps nrhip_adam_step's 16 per CU
      if (b > 1024 && live > 1) b = 1024;
      blocks += b;
    }
    if (many.count == 0) continue;
    many.total_blocks = blocks;
    adam_many_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(many);
    if (int e = check_launch("adam_step_many")) return e;
  }
  return NRHIP_OK;
}

extern "C" int nrhip_adam_step_many_workspace(int32_t n_tensors, int64_t* bytes) {
  NR_REQUIRE(n_tensors >= 0 && bytes, NRHIP_ERR_INVALID_ARG, "adam_step_many_workspace: bad argument");
  *bytes = (int64_t)sizeof(AdamCtl) * (n_tensors > 0 ? n_tensors : 1);
  return NRHIP_OK;
}

extern "C" int nrhip_adam_step_many_dev(const nrhip_adam_tensor_dev* tensors, int32_t n_tensors, double lr, const float* lr_dev,
                                        double beta1, double beta2, double eps, double weight_decay, double host_grad_scale,
                                        const float* grad_scale, const float* found_inf, void* workspace, void* stream) {
  NR_REQUIRE(n_tensors >= 0 && (tensors || n_tensors == 0), NRHIP_ERR_INVALID_ARG, "adam_step_many_dev: bad argument");
  NR_REQUIRE(workspace || n_tensors == 0, NRHIP_ERR_INVALID_ARG, "adam_step_many_dev: workspace required");
  NR_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., NRHIP_ERR_INVALID_ARG,
             "adam_step_many_dev: bad hyper-parameter");
  AdamCtl* ctl = reinterpret_cast<AdamCtl*>(workspace);
  int k = 0, slot = 0;
  while (k < n_tensors) {
    AdamManyDev many;
    AdamPrepare pr;
    many.count = 0;
    int blocks = 0;
    int live = 0;
    for (int j = k; j < n_tensors && live < kAdamMany; ++j) live += tensors[j].n > 0 ? 1 : 0;
    for (; k < n_tensors && many.count < kAdamMany; ++k) {
      const nrhip_adam_tensor_dev& in = tensors[k];
      NR_REQUIRE(in.n >= 0 && (in.grad_dtype == 0 || in.grad_dtype == 1) && in.step, NRHIP_ERR_INVALID_ARG,
                 "adam_step_many_dev: tensor %d: n >= 0, grad_dtype 0 (fp32) or 1 (fp16), a device step scalar", k);
      if (in.n == 0) continue;
      if (int e = check_tensor("adam_step_many_dev", in.param, in.grad, in.exp_avg, in.exp_avg_sq, in.image_fp16)) return e;
      pr.step[many.count] = in.step;
      AdamTensor& t = many.t[many.count++];
      t = AdamTensor{in.param, in.grad, in.exp_avg, in.exp_avg_sq, reinterpret_cast<__half*>(in.image_fp16), in.n, {},
                     in.grad_dtype, blocks};
      int b = blocks_for(in.n);
      if (b > 1024 && live > 1) b = 1024;
      blocks += b;
    }
    if (many.count == 0) continue;
    many.total_blocks = blocks;
    many.ctl = ctl + slot;
    pr.ctl = ctl + slot;
    pr.count = many.count;
    pr.lr = lr, pr.lr_dev = lr_dev, pr.beta1 = beta1, pr.beta2 = beta2, pr.eps = eps, pr.weight_decay = weight_decay;
    pr.host_grad_scale = host_grad_scale, pr.grad_scale = grad_scale, pr.found_inf = found_inf;
    slot += many.count;
    adam_prepare_kernel<<<1, 64, 0, (hipStream_t)stream>>>(pr);
    if (int e = check_launch("adam_step_many_dev (prepare)")) return e;
    adam_many_dev_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(many);
    if (int e = check_launch("adam_step_many_dev")) return e;
  }
  return NRHIP_OK;
}

extern "C" int nrhip_nonfinite_check_many(const nrhip_check_tensor* tensors, int32_t n_tensors, float* found_inf, void* stream) {
  NR_REQUIRE(n_tensors >= 0 && (tensors || n_tensors == 0) && found_inf, NRHIP_ERR_INVALID_ARG,
             "nonfinite_check_many: bad argument");
  int k = 0;
  while (k < n_tensors) {
    CheckMany many;
    many.count = 0;
    many.found = found_inf;
    int blocks = 0, live = 0;
    for (int j = k; j < n_tensors && live < kCheckMany; ++j) live += tensors[j].n > 0 ? 1 : 0;
    for (; k < n_tensors && many.count < kCheckMany; ++k) {
      const nrhip_check_tensor& in = tensors[k];
      NR_REQUIRE(in.n >= 0 && (in.dtype == 0 || in.dtype == 1), NRHIP_ERR_INVALID_ARG,
                 "nonfinite_check_many: tensor %d: n >= 0, dtype 0 (fp32) or 1 (fp16)", k);
      if (in.n == 0) continue;
      NR_REQUIRE(in.data && (reinterpret_cast<uintptr_t>(in.data) & 15) == 0, NRHIP_ERR_INVALID_ARG,
                 "nonfinite_check_many: tensor %d must be a 16-byte aligned device pointer", k);
      int b = blocks_for(in.dtype ? in.n >> 1 : in.n);
      if (b > 1024 && live > 1) b = 1024;
      many.t[many.count++] = CheckTensor{in.data, in.n, in.dtype, blocks};
      blocks += b;
    }
    if (many.count == 0) continue;
    many.total_blocks = blocks;
    nonfinite_check_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(many);
    if (int e = check_launch("nonfinite_check_many")) return e;
  }
  return NRHIP_OK;
}
