// Fused AdamW over one flat float32 parameter buffer - replaces the per-tensor loop of torch.optim.AdamW constructed
// at train.py:81 and stepped at train.py:254 of the reference (104 parameter tensors -> one launch).
// Decoupled weight decay, bias-corrected moments, the arithmetic order of torch's single-tensor AdamW:
//   p *= 1 - lr*wd;  m = b1*m + (1-b1)*g;  v = b2*v + (1-b2)*g*g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
#include "common.h"
#include <math.h>

namespace {

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, int64_t n, float lr, float b1, float b2, float eps,
                                                    float wd, float step_size, float inv_sqrt_bc2, float grad_scale) {
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float* pe = &pp.x;
    const float* ge = &gg.x;
    float* me = &mm.x;
    float* ve = &vv.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = ge[e] * grad_scale;
      float x = pe[e] * (1.f - lr * wd);
      me[e] = b1 * me[e] + (1.f - b1) * gr;
      ve[e] = b2 * ve[e] + (1.f - b2) * gr * gr;
      x -= step_size * (me[e] / (sqrtf(ve[e]) * inv_sqrt_bc2 + eps));
      pe[e] = x;
    }
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail (n not a multiple of 4)
  for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float gr = g[i] * grad_scale;
    float x = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    x -= step_size * (mi / (sqrtf(vi) * inv_sqrt_bc2 + eps));
    p[i] = x; m[i] = mi; v[i] = vi;
  }
}

}  // namespace

extern "C" int snws_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr,
                               float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                               void* stream) {
  SNWS_CHECK_ARG(n >= 0 && step >= 1, "snws_adamw_step: bad size or step");
  if (n == 0) return SNWS_OK;
  SNWS_CHECK_ARG(param && grad && exp_avg && exp_avg_sq, "snws_adamw_step: NULL pointer");
  SNWS_CHECK_ARG((((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0,
                 "snws_adamw_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  const float step_size = (float)((double)lr / bc1), inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
  int64_t blocks = cdiv64(n >> 2 > 0 ? n >> 2 : 1, 256);
  if (blocks > 4096) blocks = 4096;
  adamw_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(param, grad, exp_avg, exp_avg_sq, n, lr, beta1, beta2, eps,
                                                                   weight_decay, step_size, inv_sqrt_bc2, grad_scale);
  SNWS_LAUNCH_CHECK("snws_adamw_step");
  return SNWS_OK;
}
