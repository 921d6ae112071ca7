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
#include <stdlib.h>

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

// ---- separable forward for the 3x3 case, k in {2,4,8,16}, channel counts that are multiples of 8.
// Bilinear interpolation is separable, so  y(py,px) = b + sum_ty [row ok] sum_{r in 2 rows} wy(py+ty-1; r) * G_ty[r][px],
// G_ty[r][px] = sum_tx [col ok] sum_{x in 2 cols} wx(px+tx-1; x) * Z_{ty,tx}[r][x].
// A thread owns one column px and 8 channels of an 8-row output strip: per ty it builds the (at most 8/k + 2) rows
// of G it needs (6 loads each) and folds them into its 8 row accumulators.  The strip origin is a multiple of 8, so
// the row indices and row weights are compile-time constants (k = 16 has two strip phases); border clamping of the
// interpolation equals reading a clamped row / column with the unclamped weights.  6 + 18/k row-column terms per
// output instead of 36.
constexpr int cfloor(double v) { return ((double)(int)v > v) ? (int)v - 1 : (int)v; }
template <int K> constexpr int up_fl(int q) { return cfloor(((double)q + 0.5) / K - 0.5); }
template <int K> constexpr float up_lam(int q) { return (float)((((double)q + 0.5) / K - 0.5) - (double)up_fl<K>(q)); }

template <typename T> __device__ __forceinline__ void load8(const T* p, float (&f)[8]);
template <> __device__ __forceinline__ void load8<bf16_t>(const bf16_t* p, float (&f)[8]) { Vec<bf16_t, 8>::load(p, f); }
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&f)[8]);
template <> __device__ __forceinline__ void store8<bf16_t>(bf16_t* p, const float (&f)[8]) { Vec<bf16_t, 8>::store(p, f); }
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&f)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(f[4], f[5], f[6], f[7]);
}

constexpr int SEP_TH = 8, SEP_TW = 32;

template <typename T, int K, int PH>
__device__ __forceinline__ void upconv_sep_body(const T* __restrict__ Z, int64_t ldz, T* __restrict__ y, int64_t ldy,
                                                const float* __restrict__ bias, int h, int w, int Co, int n, int py0,
                                                int px, int ch) {
  // rows of this strip relative to the K-aligned origin: q = PH*8 + py + ty - 1
  constexpr int FLMIN = up_fl<K>(PH * SEP_TH - 1);
  constexpr int R = up_fl<K>(PH * SEP_TH + SEP_TH) + 2 - FLMIN;
  const int H = h * K, W = w * K;
  const int base_lo = (py0 - PH * SEP_TH) / K + FLMIN;
  // columns: runtime per thread, shared by every row and ty
  int xa[3], xb[3];
  float lx[3];
  bool xok[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int qx = px + tx - 1;
    xok[tx] = qx >= 0 && qx < W;
    up_tap(xok[tx] ? qx : 0, K, w, xa[tx], xb[tx], lx[tx]);
  }
  float out[SEP_TH][8];
#pragma unroll
  for (int py = 0; py < SEP_TH; ++py)
#pragma unroll
    for (int v = 0; v < 8; ++v) out[py][v] = bias ? bias[ch + v] : 0.f;
  const T* zn = Z + (int64_t)n * h * w * ldz + ch;
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    constexpr int dummy = 0; (void)dummy;
    const int lo = up_fl<K>(PH * SEP_TH + ty - 1) - FLMIN, hi = up_fl<K>(PH * SEP_TH + SEP_TH - 1 + ty - 1) + 1 - FLMIN;
    float G[R][8];
#pragma unroll
    for (int rr = 0; rr < R; ++rr) {
      if (rr < lo || rr > hi) continue;
#pragma unroll
      for (int v = 0; v < 8; ++v) G[rr][v] = 0.f;
      int row = base_lo + rr;
      row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
      const T* zr = zn + (int64_t)row * w * ldz;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        if (!xok[tx]) continue;
        float a[8], b[8];
        const int64_t co = (int64_t)(ty * 3 + tx) * Co;
        load8<T>(zr + (int64_t)xa[tx] * ldz + co, a);
        load8<T>(zr + (int64_t)xb[tx] * ldz + co, b);
        const float w0 = 1.f - lx[tx], w1 = lx[tx];
#pragma unroll
        for (int v = 0; v < 8; ++v) G[rr][v] += w0 * a[v] + w1 * b[v];
      }
    }
#pragma unroll
    for (int py = 0; py < SEP_TH; ++py) {
      const int qy = py0 + py + ty - 1;
      if (qy < 0 || qy >= H) continue;     // the conv's zero padding
      const int ia = up_fl<K>(PH * SEP_TH + py + ty - 1) - FLMIN;
      const float lam = up_lam<K>(PH * SEP_TH + py + ty - 1);
#pragma unroll
      for (int v = 0; v < 8; ++v) out[py][v] += (1.f - lam) * G[ia][v] + lam * G[ia + 1][v];
    }
  }
#pragma unroll
  for (int py = 0; py < SEP_TH; ++py)
    if (py0 + py < H) store8<T>(y + (((int64_t)n * H + py0 + py) * W + px) * ldy + ch, out[py]);
}

template <typename T, int K>
__global__ __launch_bounds__(256) void upconv_fwd_sep_kernel(const T* __restrict__ Z, int64_t ldz, T* __restrict__ y,
                                                             int64_t ldy, const float* __restrict__ bias, int h, int w,
                                                             int Co, int tilesX) {
  const int c8 = threadIdx.x & 7, pxl = threadIdx.x >> 3;
  const int tx_ = blockIdx.x % tilesX, cchunk = blockIdx.x / tilesX;
  const int px = tx_ * SEP_TW + pxl, ch = cchunk * 64 + c8 * 8;
  const int py0 = blockIdx.y * SEP_TH, n = blockIdx.z;
  if (px >= w * K || ch >= Co) return;
  if constexpr (K == 16) {
    if (blockIdx.y & 1) upconv_sep_body<T, K, 1>(Z, ldz, y, ldy, bias, h, w, Co, n, py0, px, ch);
    else upconv_sep_body<T, K, 0>(Z, ldz, y, ldy, bias, h, w, Co, n, py0, px, ch);
  } else {
    upconv_sep_body<T, K, 0>(Z, ldz, y, ldy, bias, h, w, Co, n, py0, px, ch);
  }
}

template <typename T>
bool launch_fwd_sep(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, int N, int h, int w, int Co, int k,
                    hipStream_t st) {
  const int H = h * k, W = w * k;
  const int tilesX = cdiv(W, SEP_TW), tilesY = cdiv(H, SEP_TH);
  if (tilesY > 65535 || N > 65535) return false;
  dim3 grid((unsigned)(tilesX * cdiv(Co, 64)), (unsigned)tilesY, (unsigned)N);
#define SEP(KK) upconv_fwd_sep_kernel<T, KK><<<grid, 256, 0, st>>>((const T*)Z, ldz, (T*)y, ldy, bias, h, w, Co, tilesX)
  // measured at 16 x 512^2 x 64 outputs: k = 8 / 16 run 1.5x / 1.8x faster than the per-pixel gather kernel, k = 2 / 4
  // do not (both kernels sit at the L1 rate there), unless forced for testing
  static const bool all_k = getenv("SNWS_UPCONV_SEP_ALL") != nullptr;
  if (k < 8 && !all_k) return false;
  switch (k) {
    case 2: SEP(2); return true;
    case 4: SEP(4); return true;
    case 8: SEP(8); return true;
    case 16: SEP(16); return true;
    default: return false;
  }
#undef SEP
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
  static const bool no_sep = getenv("SNWS_UPCONV_GATHER") != nullptr;   // A/B switch: the per-pixel gather kernel
  if (!no_sep && ksize == 3 && C_out % 8 == 0 && ldz % 8 == 0 && ldy % 8 == 0 && al16(Z) && al16(y)) {
    const bool done = dtype == SNWS_F32 ? launch_fwd_sep<float>(Z, ldz, y, ldy, bias, N, h, w, C_out, k, st)
                                        : launch_fwd_sep<bf16_t>(Z, ldz, y, ldy, bias, N, h, w, C_out, k, st);
    if (done) {
      SNWS_LAUNCH_CHECK("snws_upconv_fwd");
      return SNWS_OK;
    }
  }
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
