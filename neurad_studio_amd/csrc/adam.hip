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

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, int64_t n, AdamArgs a) {
  const int64_t n4 = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<const float4*>(m)[i];
    float4 vv = reinterpret_cast<const float4*>(v)[i];
    const bool dead = gv.x == 0.f && gv.y == 0.f && gv.z == 0.f && gv.w == 0.f && mv.x == 0.f && mv.y == 0.f &&
                      mv.z == 0.f && mv.w == 0.f && vv.x == 0.f && vv.y == 0.f && vv.z == 0.f && vv.w == 0.f;
    if (dead && a.decay == 1.f) continue;  // exact no-op: never-touched rows cost three reads
    float4 pv = reinterpret_cast<float4*>(p)[i];
    adam_elem(pv.x, gv.x, mv.x, vv.x, a);
    adam_elem(pv.y, gv.y, mv.y, vv.y, a);
    adam_elem(pv.z, gv.z, mv.z, vv.z, a);
    adam_elem(pv.w, gv.w, mv.w, vv.w, a);
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {  // tail
    const int64_t i = (n4 << 2) + threadIdx.x;
    adam_elem(p[i], g[i], m[i], v[i], a);
  }
}

}  // namespace nrhip

using namespace nrhip;

extern "C" int nrhip_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, int64_t step,
                               double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                               void* stream) {
  NR_REQUIRE(n >= 0 && step >= 1, NRHIP_ERR_INVALID_ARG, "adam_step: n >= 0 and step >= 1 required");
  if (n == 0) return NRHIP_OK;
  NR_REQUIRE(param && grad && exp_avg && exp_avg_sq, NRHIP_ERR_INVALID_ARG, "adam_step: NULL pointer");
  NR_REQUIRE(((reinterpret_cast<uintptr_t>(param) | reinterpret_cast<uintptr_t>(grad) | reinterpret_cast<uintptr_t>(exp_avg) |
               reinterpret_cast<uintptr_t>(exp_avg_sq)) & 15) == 0,
             NRHIP_ERR_INVALID_ARG, "adam_step: tensors must be 16-byte aligned");
  NR_REQUIRE(lr >= 0. && beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., NRHIP_ERR_INVALID_ARG,
             "adam_step: bad hyper-parameter");
  // hyper-parameters arrive as doubles and every derived scalar is formed in double, then rounded once -- exactly what
  // torch does with its Python floats
  AdamArgs a;
  const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
  a.step_size = (float)(lr / bc1);
  a.b2 = (float)beta2;
  a.omb1 = (float)(1.0 - beta1), a.omb2 = (float)(1.0 - beta2);
  a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
  a.eps = (float)eps;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.grad_scale = (float)grad_scale;
  int64_t blocks = ((n >> 2) + 255) / 256;
  if (blocks < 1) blocks = 1;
  if (blocks > 256 * 16) blocks = 256 * 16;  // grid-stride: 16 workgroups per CU
  adam_kernel<<<(int)blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, a);
  return check_launch("adam_step");
}
