// Convolution of a bilinearly up-sampled map WITHOUT materialising the up-sampled map.
//
// The UNet3+ decoder (modeling/UNet3p_SNws.py:131-155 of the reference) runs Conv2dws 3x3 on
// nn.Upsample(scale k, bilinear, align_corners=False)(x) for k = 2..16.  Both operators are linear and the
// interpolation acts per channel, so the channel contraction commutes with it:
//
//   conv(up_k(x))(p)[o] = b[o] + sum_{taps t} [p+t inside] * up_k(Z_t)(p+t)[o],   Z_t = W_t x  (a 1x1 conv at LOW resolution)
//
// Z = [Z_t]_t has taps*C_out channels at the source resolution and is produced by the MFMA 1x1 kernel once per
// source map (all scales share it).  This file holds the two memory-bound halves that remain:
//   forward : y(p)   = b + sum_t bilerp(Z_t, p+t)                  (gather from the small, cache-resident Z)
//   backward: dZ_t(s) = sum_q w(q -> s) * dy(q - t)                (gather from dy; no atomics, bit-reproducible)
// k^2 x fewer matrix FLOPs than convolving the up-sampled tensor, and the (up to 8.6 GB) up-sampled tensor and
// its gradient never exist.  Interpolation follows torch: src = max(0, (dst+0.5)/k - 0.5), i0 = floor(src),
// i1 = min(i0+1, in-1); taps that fall outside the image contribute zero (the conv's zero padding).
#include "common.h"

namespace {

__device__ __forceinline__ void up_tap(int o, int k, int in, int& i0, int& i1, float& lam) {
  float src = ((float)o + 0.5f) / (float)k - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  lam = src - (float)i0;
}

// weight with which high-res coordinate o reads low-res index s
__device__ __forceinline__ float up_weight(int o, int k, int in, int s) {
  int i0, i1;
  float lam;
  up_tap(o, k, in, i0, i1, lam);
  return (i0 == s ? 1.f - lam : 0.f) + (i1 == s ? lam : 0.f);
}

template <typename TZ, typename TY, int V>
__global__ __launch_bounds__(256) void upconv_fwd_kernel(const TZ* __restrict__ Z, int64_t ldz, TY* __restrict__ y,
                                                         int64_t ldy, const float* __restrict__ bias, int N, int h,
                                                         int w, int Co, int k, int KH) {
  const int H = h * k, W = w * k, CG = Co / V, pad = (KH - 1) / 2;
  const int64_t total = (int64_t)N * H * W * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t p = idx / CG;
    const int px = (int)(p % W); p /= W;
    const int py = (int)(p % H);
    const int n = (int)(p / H);
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = bias ? bias[cg * V + v] : 0.f;
    const TZ* zn = Z + (int64_t)n * h * w * ldz + (int64_t)cg * V;
    for (int dy = 0; dy < KH; ++dy) {
      const int qy = py + dy - pad;
      if (qy < 0 || qy >= H) continue;
      int ya, yb;
      float ly;
      up_tap(qy, k, h, ya, yb, ly);
      for (int dx = 0; dx < KH; ++dx) {
        const int qx = px + dx - pad;
        if (qx < 0 || qx >= W) continue;
        int xa, xb;
        float lx;
        up_tap(qx, k, w, xa, xb, lx);
        const int64_t co = (int64_t)(dy * KH + dx) * Co;
        float a[V], b[V], c[V], d[V];
        Vec<TZ, V>::load(zn + ((int64_t)ya * w + xa) * ldz + co, a);
        Vec<TZ, V>::load(zn + ((int64_t)ya * w + xb) * ldz + co, b);
        Vec<TZ, V>::load(zn + ((int64_t)yb * w + xa) * ldz + co, c);
        Vec<TZ, V>::load(zn + ((int64_t)yb * w + xb) * ldz + co, d);
        const float w00 = (1.f - ly) * (1.f - lx), w01 = (1.f - ly) * lx, w10 = ly * (1.f - lx), w11 = ly * lx;
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += w00 * a[v] + w01 * b[v] + w10 * c[v] + w11 * d[v];
      }
    }
    Vec<TY, V>::store(y + (((int64_t)n * H + py) * W + px) * ldy + (int64_t)cg * V, acc);
  }
}

// one thread per (source pixel, tap, channel group)
template <typename TZ, typename TY, int V>
__global__ __launch_bounds__(256) void upconv_bwd_kernel(const TY* __restrict__ dy, int64_t lddy, TZ* __restrict__ dZ,
                                                         int64_t lddz, int N, int h, int w, int Co, int k, int KH) {
  const int H = h * k, W = w * k, CG = Co / V, pad = (KH - 1) / 2, T = KH * KH;
  const int64_t total = (int64_t)N * h * w * T * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t r = idx / CG;
    const int t = (int)(r % T); r /= T;
    const int sx = (int)(r % w); r /= w;
    const int sy = (int)(r % h);
    const int n = (int)(r / h);
    const int ty = t / KH - pad, tx = t % KH - pad;   // q = p + (ty, tx)
    int qy0 = k * sy - k / 2 - 1, qy1 = k * sy + (3 * k) / 2 + 1;
    int qx0 = k * sx - k / 2 - 1, qx1 = k * sx + (3 * k) / 2 + 1;
    if (sy == h - 1) qy1 = H - 1;
    if (sx == w - 1) qx1 = W - 1;
    qy0 = qy0 < 0 ? 0 : qy0; qx0 = qx0 < 0 ? 0 : qx0;
    qy1 = qy1 > H - 1 ? H - 1 : qy1; qx1 = qx1 > W - 1 ? W - 1 : qx1;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    const TY* gn = dy + (int64_t)n * H * W * lddy + (int64_t)cg * V;
    for (int qy = qy0; qy <= qy1; ++qy) {
      const int py = qy - ty;
      if (py < 0 || py >= H) continue;
      const float wy = up_weight(qy, k, h, sy);
      if (wy == 0.f) continue;
      for (int qx = qx0; qx <= qx1; ++qx) {
        const int px = qx - tx;
        if (px < 0 || px >= W) continue;
        const float wx = up_weight(qx, k, w, sx);
        if (wx == 0.f) continue;
        float g[V];
        Vec<TY, V>::load(gn + ((int64_t)py * W + px) * lddy, g);
        const float wgt = wy * wx;
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = fmaf(wgt, g[v], acc[v]);
      }
    }
    Vec<TZ, V>::store(dZ + (((int64_t)n * h + sy) * w + sx) * lddz + (int64_t)t * Co + (int64_t)cg * V, acc);
  }
}

unsigned grid_for(int64_t total) {
  int64_t b = cdiv64(total, 256);
  if (b > 262144) b = 262144;
  if (b < 1) b = 1;
  return (unsigned)b;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check(const char* fn, int dtype, int N, int h, int w, int Co, int k, int ksize) {
  SNWS_CHECK_ARG(N >= 1 && h >= 1 && w >= 1 && Co >= 1 && k >= 1, "%s: non-positive size", fn);
  SNWS_CHECK_ARG(ksize == 1 || ksize == 3, "%s: kernel size %d is not 1 or 3", fn, ksize);
  SNWS_CHECK_ARG(dtype == SNWS_F32 || dtype == SNWS_BF16, "%s: dtype %d is not F32/BF16", fn, dtype);
  return SNWS_OK;
}

}  // namespace

extern "C" int snws_upconv_fwd(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, int dtype, int N,
                               int h, int w, int C_out, int k, int ksize, void* stream) {
  if (int rc = check("snws_upconv_fwd", dtype, N, h, w, C_out, k, ksize)) return rc;
  SNWS_CHECK_ARG(Z && y && ldz >= (int64_t)ksize * ksize * C_out && ldy >= C_out, "snws_upconv_fwd: NULL pointer or bad stride");
  hipStream_t st = (hipStream_t)stream;
  const int64_t pix = (int64_t)N * h * k * w * k;
  if (dtype == SNWS_F32) {
    if (C_out % 4 == 0 && ldz % 4 == 0 && ldy % 4 == 0 && al16(Z) && al16(y))
      upconv_fwd_kernel<float, float, 4><<<grid_for(pix * (C_out / 4)), 256, 0, st>>>((const float*)Z, ldz, (float*)y, ldy, bias, N, h, w, C_out, k, ksize);
    else
      upconv_fwd_kernel<float, float, 1><<<grid_for(pix * C_out), 256, 0, st>>>((const float*)Z, ldz, (float*)y, ldy, bias, N, h, w, C_out, k, ksize);
  } else {
    if (C_out % 8 == 0 && ldz % 8 == 0 && ldy % 8 == 0 && al16(Z) && al16(y))
      upconv_fwd_kernel<bf16_t, bf16_t, 8><<<grid_for(pix * (C_out / 8)), 256, 0, st>>>((const bf16_t*)Z, ldz, (bf16_t*)y, ldy, bias, N, h, w, C_out, k, ksize);
    else
      upconv_fwd_kernel<bf16_t, bf16_t, 1><<<grid_for(pix * C_out), 256, 0, st>>>((const bf16_t*)Z, ldz, (bf16_t*)y, ldy, bias, N, h, w, C_out, k, ksize);
  }
  SNWS_LAUNCH_CHECK("snws_upconv_fwd");
  return SNWS_OK;
}

extern "C" int snws_upconv_bwd(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int dtype, int N, int h, int w,
                               int C_out, int k, int ksize, void* stream) {
  if (int rc = check("snws_upconv_bwd", dtype, N, h, w, C_out, k, ksize)) return rc;
  SNWS_CHECK_ARG(dy && dZ && lddz >= (int64_t)ksize * ksize * C_out && lddy >= C_out, "snws_upconv_bwd: NULL pointer or bad stride");
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = (int64_t)N * h * w * ksize * ksize;
  if (dtype == SNWS_F32) {
    if (C_out % 4 == 0 && lddz % 4 == 0 && lddy % 4 == 0 && al16(dZ) && al16(dy))
      upconv_bwd_kernel<float, float, 4><<<grid_for(items * (C_out / 4)), 256, 0, st>>>((const float*)dy, lddy, (float*)dZ, lddz, N, h, w, C_out, k, ksize);
    else
      upconv_bwd_kernel<float, float, 1><<<grid_for(items * C_out), 256, 0, st>>>((const float*)dy, lddy, (float*)dZ, lddz, N, h, w, C_out, k, ksize);
  } else {
    if (C_out % 8 == 0 && lddz % 8 == 0 && lddy % 8 == 0 && al16(dZ) && al16(dy))
      upconv_bwd_kernel<bf16_t, bf16_t, 8><<<grid_for(items * (C_out / 8)), 256, 0, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dZ, lddz, N, h, w, C_out, k, ksize);
    else
      upconv_bwd_kernel<bf16_t, bf16_t, 1><<<grid_for(items * C_out), 256, 0, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dZ, lddz, N, h, w, C_out, k, ksize);
  }
  SNWS_LAUNCH_CHECK("snws_upconv_bwd");
  return SNWS_OK;
}
