// Direct (non-MFMA) NHWC convolution: forward, dgrad, wgrad for ANY stride / padding / dilation / groups in
// f32 or bf16 with f32 accumulation.  This is the exact-f32 tier and the shape-completeness path behind
// Conv2dws (modeling/weightstd.py:18 calls F.conv2d with all of those arguments); the MFMA kernels in
// conv_mfma.hip take the shapes the UNet3+ stack actually uses.  One thread per output element, channel
// fastest, so stores coalesce; no tiling - it is not the performance path.
#include "common.h"

namespace {

template <typename T>
__global__ __launch_bounds__(256) void direct_fwd_kernel(snws_conv_desc d, int OH, int OW, const T* __restrict__ x,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         T* __restrict__ y) {
  const int64_t total = (int64_t)d.N * OH * OW * d.C_out;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int co = (int)(idx % d.C_out);
  int64_t p = idx / d.C_out;
  const int ox = (int)(p % OW); p /= OW;
  const int oy = (int)(p % OH);
  const int n = (int)(p / OH);
  const int cig = d.C_in / d.groups, cog = d.C_out / d.groups;
  const int g = co / cog;
  float acc = bias ? bias[co] : 0.f;
  for (int kh = 0; kh < d.kh; ++kh) {
    const int iy = oy * d.stride - d.pad + kh * d.dil;
    if (iy < 0 || iy >= d.H) continue;
    for (int kw = 0; kw < d.kw; ++kw) {
      const int ix = ox * d.stride - d.pad + kw * d.dil;
      if (ix < 0 || ix >= d.W) continue;
      const T* xp = x + (((int64_t)n * d.H + iy) * d.W + ix) * d.ldx + g * cig;
      const float* wp = w + (((int64_t)co * d.kh + kh) * d.kw + kw) * cig;
      for (int ci = 0; ci < cig; ++ci) acc = fmaf((float)xp[ci], wp[ci], acc);
    }
  }
  y[(((int64_t)n * OH + oy) * OW + ox) * d.ldy + co] = (T)acc;
}

template <typename T>
__global__ __launch_bounds__(256) void direct_dgrad_kernel(snws_conv_desc d, int OH, int OW, const T* __restrict__ dy,
                                                           const float* __restrict__ w, T* __restrict__ dx) {
  const int64_t total = (int64_t)d.N * d.H * d.W * d.C_in;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ci = (int)(idx % d.C_in);
  int64_t p = idx / d.C_in;
  const int ix = (int)(p % d.W); p /= d.W;
  const int iy = (int)(p % d.H);
  const int n = (int)(p / d.H);
  const int cig = d.C_in / d.groups, cog = d.C_out / d.groups;
  const int g = ci / cig, cil = ci % cig;
  float acc = 0.f;
  for (int kh = 0; kh < d.kh; ++kh) {
    const int ty = iy + d.pad - kh * d.dil;
    if (ty < 0 || ty % d.stride) continue;
    const int oy = ty / d.stride;
    if (oy >= OH) continue;
    for (int kw = 0; kw < d.kw; ++kw) {
      const int tx = ix + d.pad - kw * d.dil;
      if (tx < 0 || tx % d.stride) continue;
      const int ox = tx / d.stride;
      if (ox >= OW) continue;
      const T* gp = dy + (((int64_t)n * OH + oy) * OW + ox) * d.ldy + g * cog;
      for (int co = 0; co < cog; ++co)
        acc = fmaf((float)gp[co], w[((((int64_t)(g * cog + co)) * d.kh + kh) * d.kw + kw) * cig + cil], acc);
    }
  }
  dx[(((int64_t)n * d.H + iy) * d.W + ix) * d.ldx + ci] = (T)acc;
}

// one wave per filter element (co, kh, kw, ci_local): reduction over all output pixels
template <typename T>
__global__ __launch_bounds__(256) void direct_wgrad_kernel(snws_conv_desc d, int OH, int OW, const T* __restrict__ x,
                                                           const T* __restrict__ dy, float* __restrict__ dw) {
  const int cig = d.C_in / d.groups, cog = d.C_out / d.groups;
  const int64_t total = (int64_t)d.C_out * d.kh * d.kw * cig;
  const int64_t e = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (e >= total) return;
  const int lane = threadIdx.x & 63;
  const int cil = (int)(e % cig);
  int64_t r = e / cig;
  const int kw = (int)(r % d.kw); r /= d.kw;
  const int kh = (int)(r % d.kh);
  const int co = (int)(r / d.kh);
  const int g = co / cog;
  const int64_t npix = (int64_t)d.N * OH * OW;
  float acc = 0.f;
  for (int64_t p = lane; p < npix; p += 64) {
    const int ox = (int)(p % OW);
    int64_t q = p / OW;
    const int oy = (int)(q % OH);
    const int n = (int)(q / OH);
    const int iy = oy * d.stride - d.pad + kh * d.dil, ix = ox * d.stride - d.pad + kw * d.dil;
    if (iy < 0 || iy >= d.H || ix < 0 || ix >= d.W) continue;
    acc = fmaf((float)dy[p * d.ldy + co], (float)x[(((int64_t)n * d.H + iy) * d.W + ix) * d.ldx + g * cig + cil], acc);
  }
  acc = wave_sum(acc);
  if (lane == 0) dw[e] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void direct_dbias_kernel(int64_t npix, int C_out, int64_t ldy, const T* __restrict__ dy,
                                                           float* __restrict__ dbias) {
  const int co = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (co >= C_out) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int64_t p = lane; p < npix; p += 64) acc += (float)dy[p * ldy + co];
  acc = wave_sum(acc);
  if (lane == 0) dbias[co] = acc;
}

int check_desc(const snws_conv_desc* d, const char* fn, int* OH, int* OW) {
  SNWS_CHECK_ARG(d, "%s: desc is NULL", fn);
  SNWS_CHECK_ARG(d->N >= 1 && d->H >= 1 && d->W >= 1 && d->C_in >= 1 && d->C_out >= 1, "%s: non-positive size", fn);
  SNWS_CHECK_ARG(d->kh >= 1 && d->kw >= 1 && d->stride >= 1 && d->dil >= 1 && d->pad >= 0 && d->groups >= 1,
                 "%s: bad kernel/stride/pad/dilation/groups", fn);
  SNWS_CHECK_ARG(d->C_in % d->groups == 0 && d->C_out % d->groups == 0, "%s: channels not divisible by groups", fn);
  SNWS_CHECK_ARG(d->ldx >= d->C_in && d->ldy >= d->C_out, "%s: pixel stride smaller than channel count", fn);
  *OH = snws_conv_out_size(d->H, d->kh, d->stride, d->pad, d->dil);
  *OW = snws_conv_out_size(d->W, d->kw, d->stride, d->pad, d->dil);
  SNWS_CHECK_ARG(*OH >= 1 && *OW >= 1, "%s: empty output", fn);
  return SNWS_OK;
}

}  // namespace

extern "C" int snws_conv_out_size(int in, int k, int stride, int pad, int dil) {
  return (in + 2 * pad - dil * (k - 1) - 1) / stride + 1;
}

extern "C" int snws_conv_direct_fwd(const snws_conv_desc* d, int dtype, const void* x, const float* w_ohwi,
                                    const float* bias, void* y, void* stream) {
  int OH, OW;
  if (int rc = check_desc(d, "snws_conv_direct_fwd", &OH, &OW)) return rc;
  SNWS_CHECK_ARG(x && w_ohwi && y, "snws_conv_direct_fwd: NULL pointer");
  const int64_t total = (int64_t)d->N * OH * OW * d->C_out;
  const unsigned grid = (unsigned)cdiv64(total, 256);
  if (dtype == SNWS_F32)
    direct_fwd_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(*d, OH, OW, (const float*)x, w_ohwi, bias, (float*)y);
  else if (dtype == SNWS_BF16)
    direct_fwd_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(*d, OH, OW, (const bf16_t*)x, w_ohwi, bias, (bf16_t*)y);
  else
    SNWS_CHECK_ARG(false, "snws_conv_direct_fwd: dtype %d is not F32/BF16", dtype);
  SNWS_LAUNCH_CHECK("snws_conv_direct_fwd");
  return SNWS_OK;
}

extern "C" int snws_conv_direct_dgrad(const snws_conv_desc* d, int dtype, const void* dy, const float* w_ohwi, void* dx,
                                      void* stream) {
  int OH, OW;
  if (int rc = check_desc(d, "snws_conv_direct_dgrad", &OH, &OW)) return rc;
  SNWS_CHECK_ARG(dy && w_ohwi && dx, "snws_conv_direct_dgrad: NULL pointer");
  const int64_t total = (int64_t)d->N * d->H * d->W * d->C_in;
  const unsigned grid = (unsigned)cdiv64(total, 256);
  if (dtype == SNWS_F32)
    direct_dgrad_kernel<float><<<grid, 256, 0, (hipStream_t)stream>>>(*d, OH, OW, (const float*)dy, w_ohwi, (float*)dx);
  else if (dtype == SNWS_BF16)
    direct_dgrad_kernel<bf16_t><<<grid, 256, 0, (hipStream_t)stream>>>(*d, OH, OW, (const bf16_t*)dy, w_ohwi, (bf16_t*)dx);
  else
    SNWS_CHECK_ARG(false, "snws_conv_direct_dgrad: dtype %d is not F32/BF16", dtype);
  SNWS_LAUNCH_CHECK("snws_conv_direct_dgrad");
  return SNWS_OK;
}

extern "C" int snws_conv_direct_wgrad(const snws_conv_desc* d, int dtype, const void* x, const void* dy, float* dw_ohwi,
                                      float* dbias, void* stream) {
  int OH, OW;
  if (int rc = check_desc(d, "snws_conv_direct_wgrad", &OH, &OW)) return rc;
  SNWS_CHECK_ARG(x && dy && dw_ohwi, "snws_conv_direct_wgrad: NULL pointer");
  const int64_t total = (int64_t)d->C_out * d->kh * d->kw * (d->C_in / d->groups);
  const unsigned grid = (unsigned)cdiv64(total, 4);
  const int64_t npix = (int64_t)d->N * OH * OW;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    direct_wgrad_kernel<float><<<grid, 256, 0, st>>>(*d, OH, OW, (const float*)x, (const float*)dy, dw_ohwi);
    if (dbias) direct_dbias_kernel<float><<<cdiv(d->C_out, 4), 256, 0, st>>>(npix, d->C_out, d->ldy, (const float*)dy, dbias);
  } else if (dtype == SNWS_BF16) {
    direct_wgrad_kernel<bf16_t><<<grid, 256, 0, st>>>(*d, OH, OW, (const bf16_t*)x, (const bf16_t*)dy, dw_ohwi);
    if (dbias) direct_dbias_kernel<bf16_t><<<cdiv(d->C_out, 4), 256, 0, st>>>(npix, d->C_out, d->ldy, (const bf16_t*)dy, dbias);
  } else {
    SNWS_CHECK_ARG(false, "snws_conv_direct_wgrad: dtype %d is not F32/BF16", dtype);
  }
  SNWS_LAUNCH_CHECK("snws_conv_direct_wgrad");
  return SNWS_OK;
}
