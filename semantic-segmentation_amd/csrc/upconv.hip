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

// weight of high-res coordinate q on low-res index s, 0 when q is outside [0, n_hi)
__device__ __forceinline__ float tap_weight(int q, int n_hi, float inv_k, int n_lo, int s) {
  if (q < 0 || q >= n_hi) return 0.f;
  float src = ((float)q + 0.5f) * inv_k - 0.5f;
  src = fminf(fmaxf(src, 0.f), (float)(n_lo - 1));
  const float d = 1.f - fabsf(src - (float)s);
  return d > 0.f ? d : 0.f;
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
    // per-axis interpolation taps of the (up to) three rows / columns this pixel's filter window touches
    int ya[3], yb[3], xa[3], xb[3];
    float ly[3], lx[3];
    bool yok[3], xok[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      const int qy = py + d - pad, qx = px + d - pad;
      yok[d] = d < KH && qy >= 0 && qy < H;
      xok[d] = d < KH && qx >= 0 && qx < W;
      up_tap(yok[d] ? qy : 0, k, h, ya[d], yb[d], ly[d]);
      up_tap(xok[d] ? qx : 0, k, w, xa[d], xb[d], lx[d]);
    }
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      if (!yok[dy]) continue;
      const TZ* ra = zn + (int64_t)ya[dy] * w * ldz;
      const TZ* rb = zn + (int64_t)yb[dy] * w * ldz;
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        if (!xok[dx]) continue;
        const int64_t co = (int64_t)(dy * KH + dx) * Co;
        float a[V], b[V], c[V], d[V];
        Vec<TZ, V>::load(ra + (int64_t)xa[dx] * ldz + co, a);
        Vec<TZ, V>::load(ra + (int64_t)xb[dx] * ldz + co, b);
        Vec<TZ, V>::load(rb + (int64_t)xa[dx] * ldz + co, c);
        Vec<TZ, V>::load(rb + (int64_t)xb[dx] * ldz + co, d);
        const float w00 = (1.f - ly[dy]) * (1.f - lx[dx]), w01 = (1.f - ly[dy]) * lx[dx];
        const float w10 = ly[dy] * (1.f - lx[dx]), w11 = ly[dy] * lx[dx];
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] += w00 * a[v] + w01 * b[v] + w10 * c[v] + w11 * d[v];
      }
    }
    Vec<TY, V>::store(y + (((int64_t)n * H + py) * W + px) * ldy + (int64_t)cg * V, acc);
  }
}

// Backward: one thread per (source pixel, channel group) for ALL taps: every dy value of the (2k+2)^2 window is
// loaded once and scattered into the taps' register accumulators with separable weights.
template <typename TZ, typename TY, int V, int KH>
__global__ __launch_bounds__(256) void upconv_bwd_kernel(const TY* __restrict__ dy, int64_t lddy, TZ* __restrict__ dZ,
                                                         int64_t lddz, int N, int h, int w, int Co, int k) {
  constexpr int pad = (KH - 1) / 2, T = KH * KH;
  const int H = h * k, W = w * k, CG = Co / V;
  const float inv_k = 1.f / (float)k;
  const int64_t total = (int64_t)N * h * w * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t r = idx / CG;
    const int sx = (int)(r % w); r /= w;
    const int sy = (int)(r % h);
    const int n = (int)(r / h);
    // pixels p whose window positions q = p + t can touch source (sy, sx)
    int py0 = k * sy - k / 2 - 1 - pad, py1 = k * sy + (3 * k) / 2 + 1 + pad;
    int px0 = k * sx - k / 2 - 1 - pad, px1 = k * sx + (3 * k) / 2 + 1 + pad;
    if (sy == h - 1) py1 = H - 1;
    if (sx == w - 1) px1 = W - 1;
    py0 = py0 < 0 ? 0 : py0; px0 = px0 < 0 ? 0 : px0;
    py1 = py1 > H - 1 ? H - 1 : py1; px1 = px1 > W - 1 ? W - 1 : px1;
    float acc[T][V];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int v = 0; v < V; ++v) acc[t][v] = 0.f;
    const TY* gn = dy + (int64_t)n * H * W * lddy + (int64_t)cg * V;
    for (int py = py0; py <= py1; ++py) {
      float wy[KH];
      bool any = false;
#pragma unroll
      for (int d = 0; d < KH; ++d) {
        wy[d] = tap_weight(py + d - pad, H, inv_k, h, sy);
        any |= wy[d] != 0.f;
      }
      if (!any) continue;
      for (int px = px0; px <= px1; ++px) {
        float wx[KH];
        bool anyx = false;
#pragma unroll
        for (int d = 0; d < KH; ++d) {
          wx[d] = tap_weight(px + d - pad, W, inv_k, w, sx);
          anyx |= wx[d] != 0.f;
        }
        if (!anyx) continue;
        float g[V];
        Vec<TY, V>::load(gn + ((int64_t)py * W + px) * lddy, g);
#pragma unroll
        for (int a = 0; a < KH; ++a)
#pragma unroll
          for (int b = 0; b < KH; ++b) {
            const float wgt = wy[a] * wx[b];
#pragma unroll
            for (int v = 0; v < V; ++v) acc[a * KH + b][v] = fmaf(wgt, g[v], acc[a * KH + b][v]);
          }
      }
    }
    TZ* o = dZ + (((int64_t)n * h + sy) * w + sx) * lddz + (int64_t)cg * V;
#pragma unroll
    for (int t = 0; t < T; ++t) Vec<TZ, V>::store(o + (int64_t)t * Co, acc[t]);
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
  const int64_t items = (int64_t)N * h * w;
#define UPB(TT, VV, KK) upconv_bwd_kernel<TT, TT, VV, KK><<<grid_for(items * (C_out / VV)), 256, 0, st>>>((const TT*)dy, lddy, (TT*)dZ, lddz, N, h, w, C_out, k)
  if (dtype == SNWS_F32) {
    const bool vec = C_out % 4 == 0 && lddz % 4 == 0 && lddy % 4 == 0 && al16(dZ) && al16(dy);
    if (ksize == 3) { if (vec) UPB(float, 4, 3); else UPB(float, 1, 3); }
    else { if (vec) UPB(float, 4, 1); else UPB(float, 1, 1); }
  } else {
    const bool vec = C_out % 8 == 0 && lddz % 8 == 0 && lddy % 8 == 0 && al16(dZ) && al16(dy);
    if (ksize == 3) { if (vec) UPB(bf16_t, 8, 3); else UPB(bf16_t, 1, 3); }
    else { if (vec) UPB(bf16_t, 8, 1); else UPB(bf16_t, 1, 1); }
  }
#undef UPB
  SNWS_LAUNCH_CHECK("snws_upconv_bwd");
  return SNWS_OK;
}
