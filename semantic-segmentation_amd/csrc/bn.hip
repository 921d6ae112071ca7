// Batch normalisation on the SwitchNorm streaming kernels - replaces nn.BatchNorm2d (+ReLU) of the OCR head
// (modeling/UNet3p_res_ocr_SNws.py:52-79,124,257,263) and the statistics of _SynchronizedBatchNorm
// (modeling/sync_batchnorm/batchnorm.py:45-65,80-115: per-channel sum / square-sum / count, mean and inv_std from
// the all-reduced triple, running buffers with the unbiased variance).
//
// The heavy passes are shared with SwitchNorm: snws_sn_stats + snws_sn_finalize give mean_in / var_in per (n, c);
// the kernels here fold them into the packed [sum_n mean, sum_n E[x^2], N] vector that a data-parallel job
// all-reduces, turn the reduced vector into the per-(n,c) coefficient arrays snws_sn_apply reads, and do the same
// for backward around snws_sn_bwd_reduce / snws_sn_bwd_apply (dx = A*dy' + B*x + K per channel).
#include "common.h"

namespace {

// ctx / ctxb layouts of sn.hip (floats)
struct CtxOff { int64_t mean_in, var_in, mean, rstd, scale; };
__device__ __host__ inline CtxOff ctx_off(int N, int C) {
  const int64_t NC = (int64_t)N * C;
  return CtxOff{0, NC, 2 * NC, 3 * NC, 4 * NC};
}
struct CtxbOff { int64_t B, K; };
__device__ __host__ inline CtxbOff ctxb_off(int N, int C) {
  const int64_t NC = (int64_t)N * C;
  return CtxbOff{4 * NC, 5 * NC};
}

__global__ __launch_bounds__(256) void bn_pack_kernel(const float* __restrict__ ctx, int N, int64_t HW, int C,
                                                      float* __restrict__ pack) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c == 0) pack[2 * C] = (float)N;
  if (c >= C) return;
  const CtxOff o = ctx_off(N, C);
  double s1 = 0, s2 = 0;
  const double f = (double)(HW - 1) / (double)HW;
  for (int n = 0; n < N; ++n) {
    const double m = ctx[o.mean_in + (int64_t)n * C + c], v = ctx[o.var_in + (int64_t)n * C + c];
    s1 += m;
    s2 += v * f + m * m;
  }
  pack[c] = (float)s1;
  pack[C + c] = (float)s2;
}

__global__ __launch_bounds__(256) void bn_coeff_kernel(float* __restrict__ ctx, const float* __restrict__ pack,
                                                       const float* __restrict__ gamma, float* __restrict__ rm,
                                                       float* __restrict__ rv, int N, int64_t HW, int C, float eps,
                                                       float momentum, int training, int clamp_eps) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double mean, var;
  if (training) {
    const double Ng = pack[2 * C];
    mean = (double)pack[c] / Ng;
    var = (double)pack[C + c] / Ng - mean * mean;
    var = var > 0 ? var : 0;
    const double M = Ng * (double)HW;
    if (rm) rm[c] = (float)((1.0 - momentum) * rm[c] + momentum * mean);
    if (rv) rv[c] = (float)((1.0 - momentum) * rv[c] + momentum * var * (M / (M - 1.0)));
  } else {
    mean = rm[c];
    var = rv[c];
  }
  const float rstd = clamp_eps ? rsqrtf(fmaxf((float)var, eps)) : rsqrtf((float)var + eps);
  const float g = gamma ? gamma[c] : 1.f;
  const CtxOff o = ctx_off(N, C);
  for (int n = 0; n < N; ++n) {
    const int64_t i = (int64_t)n * C + c;
    ctx[o.mean + i] = (float)mean;
    ctx[o.rstd + i] = rstd;
    ctx[o.scale + i] = g * rstd;
  }
}

// post: optional per-(image, channel) multiplier of the forward's output (see snws_sn_apply_post): the partial sums of
// image n were taken over the unscaled gradient and are scaled here.
// 64 channels x 16 row lanes per workgroup, four rows in flight per thread (the rows are N*T short strided reads).
constexpr int BP_LANES = 16;
__global__ __launch_bounds__(64 * BP_LANES) void bn_bwd_pack_kernel(const float* __restrict__ part, int T, int N, int C,
                                                                   float* __restrict__ packb, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, const float* __restrict__ post) {
  __shared__ double red[2][BP_LANES][64];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  double s1 = 0, s2 = 0;
  if (c < C) {
    const int64_t rows = (int64_t)N * T;
    int64_t r = tl;
    for (; r + 3 * BP_LANES < rows; r += 4 * BP_LANES) {
      float2 v[4];
      float pm[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t ru = r + u * BP_LANES;
        v[u] = *reinterpret_cast<const float2*>(part + (ru * C + c) * 2);
        pm[u] = post ? post[(ru / T) * C + c] : 1.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        s1 += (double)pm[u] * (double)v[u].x;
        s2 += (double)pm[u] * (double)v[u].y;
      }
    }
    for (; r < rows; r += BP_LANES) {
      const float2 v = *reinterpret_cast<const float2*>(part + (r * C + c) * 2);
      const double pm = post ? (double)post[(r / T) * C + c] : 1.0;
      s1 += pm * (double)v.x;
      s2 += pm * (double)v.y;
    }
  }
  red[0][tl][cl] = s1;
  red[1][tl][cl] = s2;
  __syncthreads();
  if (tl == 0 && c < C) {
    s1 = s2 = 0;
#pragma unroll
    for (int k = 0; k < BP_LANES; ++k) {     // fixed order: bit-reproducible
      s1 += red[0][k][cl];
      s2 += red[1][k][cl];
    }
    packb[c] = (float)s1;
    packb[C + c] = (float)s2;
    if (dbeta) dbeta[c] = (float)s1;
    if (dgamma) dgamma[c] = (float)s2;
  }
}

__global__ __launch_bounds__(256) void bn_bwd_coeff_kernel(const float* __restrict__ ctx, float* __restrict__ ctxb,
                                                           const float* __restrict__ packb, int N, int C, float count,
                                                           int training) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const CtxOff o = ctx_off(N, C);
  const CtxbOff ob = ctxb_off(N, C);
  float B = 0.f, K = 0.f;
  if (training) {   // dx = A*(dy' - S1/M - xhat*S2/M),  xhat = (x - mu)*r,  A = gamma*r
    const float A = ctx[o.scale + c], r = ctx[o.rstd + c], mu = ctx[o.mean + c];
    const float S1 = packb[c], S2 = packb[C + c];
    B = -A * r * S2 / count;
    K = -A * S1 / count - B * mu;
  }
  for (int n = 0; n < N; ++n) {
    ctxb[ob.B + (int64_t)n * C + c] = B;
    ctxb[ob.K + (int64_t)n * C + c] = K;
  }
}

}  // namespace

extern "C" int snws_bn_pack(const float* ctx, int N, int64_t HW, int C, float* pack, void* stream) {
  SNWS_CHECK_ARG(ctx && pack && N >= 1 && C >= 1 && HW >= 1, "snws_bn_pack: bad arguments");
  bn_pack_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(ctx, N, HW, C, pack);
  SNWS_LAUNCH_CHECK("snws_bn_pack");
  return SNWS_OK;
}

extern "C" int snws_bn_coeff(float* ctx, const float* pack, const float* gamma, float* running_mean, float* running_var,
                             int N, int64_t HW, int C, float eps, float momentum, int training, int clamp_eps,
                             void* stream) {
  SNWS_CHECK_ARG(ctx && N >= 1 && C >= 1 && HW >= 1, "snws_bn_coeff: bad arguments");
  SNWS_CHECK_ARG(training ? pack != nullptr : (running_mean && running_var),
                 "snws_bn_coeff: training needs pack, evaluation needs the running buffers");
  bn_coeff_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(ctx, pack, gamma, running_mean, running_var, N, HW, C, eps,
                                                                  momentum, training, clamp_eps);
  SNWS_LAUNCH_CHECK("snws_bn_coeff");
  return SNWS_OK;
}

extern "C" int snws_bn_bwd_pack(const float* part, int T, int N, int C, float* packb, float* dgamma, float* dbeta,
                                void* stream) {
  SNWS_CHECK_ARG(part && packb && T >= 1 && N >= 1 && C >= 1, "snws_bn_bwd_pack: bad arguments");
  bn_bwd_pack_kernel<<<cdiv(C, 64), 64 * BP_LANES, 0, (hipStream_t)stream>>>(part, T, N, C, packb, dgamma, dbeta, nullptr);
  SNWS_LAUNCH_CHECK("snws_bn_bwd_pack");
  return SNWS_OK;
}

extern "C" int snws_bn_bwd_pack_post(const float* part, int T, int N, int C, const float* post, float* packb, float* dgamma,
                                     float* dbeta, void* stream) {
  SNWS_CHECK_ARG(part && packb && post && T >= 1 && N >= 1 && C >= 1, "snws_bn_bwd_pack_post: bad arguments");
  bn_bwd_pack_kernel<<<cdiv(C, 64), 64 * BP_LANES, 0, (hipStream_t)stream>>>(part, T, N, C, packb, dgamma, dbeta, post);
  SNWS_LAUNCH_CHECK("snws_bn_bwd_pack_post");
  return SNWS_OK;
}

extern "C" int snws_bn_bwd_coeff(const float* ctx, float* ctxb, const float* packb, int N, int C, float count,
                                 int training, void* stream) {
  SNWS_CHECK_ARG(ctx && ctxb && packb && N >= 1 && C >= 1 && count >= 1.f, "snws_bn_bwd_coeff: bad arguments");
  bn_bwd_coeff_kernel<<<cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(ctx, ctxb, packb, N, C, count, training);
  SNWS_LAUNCH_CHECK("snws_bn_bwd_coeff");
  return SNWS_OK;
}
