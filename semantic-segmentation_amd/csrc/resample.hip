// Full-scale skip resampling for the UNet3+ decoder (replaces nn.MaxPool2d(k,k,ceil_mode=True),
// nn.Upsample(scale,'bilinear',align_corners=False) and torch.cat of modeling/UNet3p_SNws.py:49-55,127-156 and
// their autograd).  One gather kernel per source: it reads the source once and writes straight into the
// destination's channel slice of the concat buffer (pixel stride ldd), so no intermediate is materialised and
// no separate concat pass exists.  HBM-bound; 16 B per lane along the channel axis.
//
// Backward is also a gather (no atomics, bit-reproducible): max-pool windows do not overlap (stride = k), so the
// thread that owns a window routes the gradient to its first maximum; for bilinear, each SOURCE pixel sums the
// weights of the <= (2k+2)^2 destination pixels that touched it.
#include "common.h"

namespace {

enum { MODE_COPY = 0, MODE_MAXPOOL = 1, MODE_BILINEAR = 2 };

__device__ __forceinline__ void bilinear_tap(int o, int k, int in, int& i0, int& i1, float& lam) {
  // src = (dst + 0.5) / scale - 0.5, clamped below at 0 (align_corners=False)
  float src = ((float)o + 0.5f) / (float)k - 0.5f;
  src = src < 0.f ? 0.f : src;
  i0 = (int)src;
  i0 = i0 < in - 1 ? i0 : in - 1;
  i1 = i0 + 1 < in ? i0 + 1 : in - 1;
  lam = src - (float)i0;
}

template <typename T, int V>
__global__ __launch_bounds__(256) void resample_fwd_kernel(const T* __restrict__ src, int64_t lds_, T* __restrict__ dst,
                                                           int64_t ldd, int N, int H, int W, int OH, int OW, int C,
                                                           int mode, int k) {
  const int CG = C / V;
  const int64_t total = (int64_t)N * OH * OW * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t p = idx / CG;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    const T* sn = src + (int64_t)n * H * W * lds_ + (int64_t)cg * V;
    float out[V];
    if (mode == MODE_COPY) {
      Vec<T, V>::load(sn + ((int64_t)oy * W + ox) * lds_, out);
    } else if (mode == MODE_MAXPOOL) {
      const int y0 = oy * k, x0 = ox * k;
      const int y1 = y0 + k < H ? y0 + k : H, x1 = x0 + k < W ? x0 + k : W;
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = -INFINITY;
      for (int yy = y0; yy < y1; ++yy)
        for (int xx = x0; xx < x1; ++xx) {
          float a[V];
          Vec<T, V>::load(sn + ((int64_t)yy * W + xx) * lds_, a);
#pragma unroll
          for (int v = 0; v < V; ++v) out[v] = (a[v] > out[v] || a[v] != a[v]) ? a[v] : out[v];
        }
    } else {
      int ya, yb, xa, xb;
      float ly, lx;
      bilinear_tap(oy, k, H, ya, yb, ly);
      bilinear_tap(ox, k, W, xa, xb, lx);
      float a[V], b[V], c[V], d[V];
      Vec<T, V>::load(sn + ((int64_t)ya * W + xa) * lds_, a);
      Vec<T, V>::load(sn + ((int64_t)ya * W + xb) * lds_, b);
      Vec<T, V>::load(sn + ((int64_t)yb * W + xa) * lds_, c);
      Vec<T, V>::load(sn + ((int64_t)yb * W + xb) * lds_, d);
      const float hy = 1.f - ly, hx = 1.f - lx;
#pragma unroll
      for (int v = 0; v < V; ++v) out[v] = hy * (hx * a[v] + lx * b[v]) + ly * (hx * c[v] + lx * d[v]);
    }
    Vec<T, V>::store(dst + (((int64_t)n * OH + oy) * OW + ox) * ldd + (int64_t)cg * V, out);
  }
}

// copy / max-pool backward: one thread per destination pixel (= per source window) and channel group
template <typename T, int V>
__global__ __launch_bounds__(256) void resample_bwd_window_kernel(const T* __restrict__ ddst, int64_t ldd,
                                                                  const T* __restrict__ src, int64_t lds_,
                                                                  T* __restrict__ dsrc, int64_t ldds, int N, int H,
                                                                  int W, int OH, int OW, int C, int mode, int k, int accumulate) {
  const int CG = C / V;
  const int64_t total = (int64_t)N * OH * OW * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t p = idx / CG;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    float g[V];
    Vec<T, V>::load(ddst + (((int64_t)n * OH + oy) * OW + ox) * ldd + (int64_t)cg * V, g);
    T* dn = dsrc + (int64_t)n * H * W * ldds + (int64_t)cg * V;
    if (mode == MODE_COPY) {
      float a[V];
      T* q = dn + ((int64_t)oy * W + ox) * ldds;
      if (accumulate) {
        Vec<T, V>::load(q, a);
#pragma unroll
        for (int v = 0; v < V; ++v) a[v] += g[v];
        Vec<T, V>::store(q, a);
      } else {
        Vec<T, V>::store(q, g);
      }
      continue;
    }
    const T* sn = src + (int64_t)n * H * W * lds_ + (int64_t)cg * V;
    const int y0 = oy * k, x0 = ox * k;
    const int y1 = y0 + k < H ? y0 + k : H, x1 = x0 + k < W ? x0 + k : W;
    float best[V];
    int arg[V];
#pragma unroll
    for (int v = 0; v < V; ++v) { best[v] = -INFINITY; arg[v] = -1; }
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        float a[V];
        Vec<T, V>::load(sn + ((int64_t)yy * W + xx) * lds_, a);
        const int pos = (yy - y0) * k + (xx - x0);
#pragma unroll
        for (int v = 0; v < V; ++v)
          if (arg[v] < 0 || a[v] > best[v] || (a[v] != a[v] && best[v] == best[v])) { best[v] = a[v]; arg[v] = pos; }
      }
    for (int yy = y0; yy < y1; ++yy)
      for (int xx = x0; xx < x1; ++xx) {
        const int pos = (yy - y0) * k + (xx - x0);
        bool any = false;
#pragma unroll
        for (int v = 0; v < V; ++v) any |= (arg[v] == pos);
        float a[V];
        T* q = dn + ((int64_t)yy * W + xx) * ldds;
        if (accumulate) {
          if (!any) continue;
          Vec<T, V>::load(q, a);
#pragma unroll
          for (int v = 0; v < V; ++v) a[v] += (arg[v] == pos) ? g[v] : 0.f;
        } else {      // every source pixel belongs to exactly one window: write the whole gradient, zeros included
#pragma unroll
          for (int v = 0; v < V; ++v) a[v] = (arg[v] == pos) ? g[v] : 0.f;
        }
        Vec<T, V>::store(q, a);
      }
  }
}

// 2x2 max-pool backward (every pooling step of the encoder and of the pooled skip pyramid): as the window kernel, but
// the four source vectors, the incoming gradient and - when accumulating - the four destination vectors are all
// requested before the first comparison, instead of one dependent load per loop iteration.
template <typename T, int V>
__global__ __launch_bounds__(256) void pool2_bwd_kernel(const T* __restrict__ ddst, int64_t ldd,
                                                        const T* __restrict__ src, int64_t lds_, T* __restrict__ dsrc,
                                                        int64_t ldds, int N, int H, int W, int OH, int OW, int C,
                                                        int accumulate) {
  const int CG = C / V;
  const int64_t total = (int64_t)N * OH * OW * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t p = idx / CG;
    const int ox = (int)(p % OW); p /= OW;
    const int oy = (int)(p % OH);
    const int n = (int)(p / OH);
    const int y0 = 2 * oy, x0 = 2 * ox;
    const bool ok[4] = {true, x0 + 1 < W, y0 + 1 < H, x0 + 1 < W && y0 + 1 < H};      // ceil mode: ragged last windows
    const int64_t base = (int64_t)y0 * W + x0;
    const int64_t off[4] = {base, base + (ok[1] ? 1 : 0), base + (ok[2] ? W : 0), base + (ok[3] ? W + 1 : 0)};
    const T* sn = src + (int64_t)n * H * W * lds_ + (int64_t)cg * V;
    T* dn = dsrc + (int64_t)n * H * W * ldds + (int64_t)cg * V;
    float g[V], a[4][V], d[4][V];
    Vec<T, V>::load(ddst + (((int64_t)n * OH + oy) * OW + ox) * ldd + (int64_t)cg * V, g);
#pragma unroll
    for (int q = 0; q < 4; ++q) VecNT<T, V>::load(sn + off[q] * lds_, a[q]);
    if (accumulate) {
#pragma unroll
      for (int q = 0; q < 4; ++q) Vec<T, V>::load(dn + off[q] * ldds, d[q]);
    }
    int arg[V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
      float best = a[0][v];
      arg[v] = 0;
#pragma unroll
      for (int q = 1; q < 4; ++q)
        if (ok[q] && (a[q][v] > best || (a[q][v] != a[q][v] && best == best))) { best = a[q][v]; arg[v] = q; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (!ok[q]) continue;
      float r[V];
#pragma unroll
      for (int v = 0; v < V; ++v) r[v] = (accumulate ? d[q][v] : 0.f) + (arg[v] == q ? g[v] : 0.f);
      VecNT<T, V>::store(dn + off[q] * ldds, r);
    }
  }
}

// bilinear backward: one thread per SOURCE pixel and channel group
template <typename T, int V>
__global__ __launch_bounds__(256) void resample_bwd_bilinear_kernel(const T* __restrict__ ddst, int64_t ldd,
                                                                    T* __restrict__ dsrc, int64_t ldds, int N, int H,
                                                                    int W, int OH, int OW, int C, int k, int accumulate) {
  const int CG = C / V;
  const int64_t total = (int64_t)N * H * W * CG;
  for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
    const int cg = (int)(idx % CG);
    int64_t p = idx / CG;
    const int ix = (int)(p % W); p /= W;
    const int iy = (int)(p % H);
    const int n = (int)(p / H);
    int oy0 = k * iy - k / 2 - 1, oy1 = k * iy + (3 * k) / 2 + 1;
    int ox0 = k * ix - k / 2 - 1, ox1 = k * ix + (3 * k) / 2 + 1;
    oy0 = oy0 < 0 ? 0 : oy0; ox0 = ox0 < 0 ? 0 : ox0;
    oy1 = oy1 > OH - 1 ? OH - 1 : oy1; ox1 = ox1 > OW - 1 ? OW - 1 : ox1;
    if (iy == H - 1) oy1 = OH - 1;
    if (ix == W - 1) ox1 = OW - 1;
    float acc[V];
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = 0.f;
    const T* gn = ddst + (int64_t)n * OH * OW * ldd + (int64_t)cg * V;
    for (int oy = oy0; oy <= oy1; ++oy) {
      int ya, yb;
      float ly;
      bilinear_tap(oy, k, H, ya, yb, ly);
      const float wy = (ya == iy ? 1.f - ly : 0.f) + (yb == iy ? ly : 0.f);
      if (wy == 0.f) continue;
      for (int ox = ox0; ox <= ox1; ++ox) {
        int xa, xb;
        float lx;
        bilinear_tap(ox, k, W, xa, xb, lx);
        const float wx = (xa == ix ? 1.f - lx : 0.f) + (xb == ix ? lx : 0.f);
        if (wx == 0.f) continue;
        float g[V];
        Vec<T, V>::load(gn + ((int64_t)oy * OW + ox) * ldd, g);
        const float wgt = wy * wx;
#pragma unroll
        for (int v = 0; v < V; ++v) acc[v] = fmaf(wgt, g[v], acc[v]);
      }
    }
    float a[V];
    T* q = dsrc + (((int64_t)n * H + iy) * W + ix) * ldds + (int64_t)cg * V;
    if (accumulate) {
      Vec<T, V>::load(q, a);
#pragma unroll
      for (int v = 0; v < V; ++v) a[v] += acc[v];
      Vec<T, V>::store(q, a);
    } else {
      Vec<T, V>::store(q, acc);
    }
  }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int out_size(int in, int mode, int k) {
  if (mode == MODE_COPY) return in;
  if (mode == MODE_MAXPOOL) return (in + k - 1) / k;
  return in * k;
}

unsigned grid_for(int64_t total) {
  int64_t b = cdiv64(total, 256);
  if (b > 65536) b = 65536;
  if (b < 1) b = 1;
  return (unsigned)b;
}

template <typename T, int V>
int run_fwd(const void* src, int64_t lds_, void* dst, int64_t ldd, int N, int H, int W, int C, int mode, int k,
            hipStream_t st) {
  const int OH = out_size(H, mode, k), OW = out_size(W, mode, k);
  const int64_t total = (int64_t)N * OH * OW * (C / V);
  resample_fwd_kernel<T, V><<<grid_for(total), 256, 0, st>>>((const T*)src, lds_, (T*)dst, ldd, N, H, W, OH, OW, C, mode, k);
  SNWS_LAUNCH_CHECK("snws_resample_fwd");
  return SNWS_OK;
}

template <typename T, int V>
int run_bwd(const void* ddst, int64_t ldd, const void* src, int64_t lds_, void* dsrc, int64_t ldds, int N, int H, int W,
            int C, int mode, int k, int accumulate, hipStream_t st) {
  const int OH = out_size(H, mode, k), OW = out_size(W, mode, k);
  if (mode == MODE_BILINEAR) {
    const int64_t total = (int64_t)N * H * W * (C / V);
    resample_bwd_bilinear_kernel<T, V><<<grid_for(total), 256, 0, st>>>((const T*)ddst, ldd, (T*)dsrc, ldds, N, H, W, OH, OW, C, k, accumulate);
  } else {
    const int64_t total = (int64_t)N * OH * OW * (C / V);
    if (mode == MODE_MAXPOOL && k == 2)
      pool2_bwd_kernel<T, V><<<grid_for(total), 256, 0, st>>>((const T*)ddst, ldd, (const T*)src, lds_, (T*)dsrc, ldds, N, H, W, OH, OW, C, accumulate);
    else
      resample_bwd_window_kernel<T, V><<<grid_for(total), 256, 0, st>>>((const T*)ddst, ldd, (const T*)src, lds_, (T*)dsrc, ldds, N, H, W, OH, OW, C, mode, k, accumulate);
  }
  SNWS_LAUNCH_CHECK("snws_resample_bwd");
  return SNWS_OK;
}

int check(const char* fn, int dtype, int N, int H, int W, int C, int mode, int k) {
  SNWS_CHECK_ARG(N >= 1 && H >= 1 && W >= 1 && C >= 1, "%s: non-positive size", fn);
  SNWS_CHECK_ARG(dtype == SNWS_F32 || dtype == SNWS_BF16, "%s: dtype %d is not F32/BF16", fn, dtype);
  SNWS_CHECK_ARG(mode >= 0 && mode <= 2, "%s: mode %d is not copy/maxpool/bilinear", fn, mode);
  SNWS_CHECK_ARG(mode == MODE_COPY || k >= 1, "%s: factor %d < 1", fn, k);
  return SNWS_OK;
}

}  // namespace

extern "C" int snws_resample_fwd(const void* src, int64_t lds_, void* dst, int64_t ldd, int dtype, int N, int H, int W,
                                 int C, int mode, int k, void* stream) {
  if (int rc = check("snws_resample_fwd", dtype, N, H, W, C, mode, k)) return rc;
  SNWS_CHECK_ARG(src && dst && lds_ >= C && ldd >= C, "snws_resample_fwd: NULL pointer or pixel stride < C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (C % 4 == 0 && lds_ % 4 == 0 && ldd % 4 == 0 && al16(src) && al16(dst))
      return run_fwd<float, 4>(src, lds_, dst, ldd, N, H, W, C, mode, k, st);
    return run_fwd<float, 1>(src, lds_, dst, ldd, N, H, W, C, mode, k, st);
  }
  if (C % 8 == 0 && lds_ % 8 == 0 && ldd % 8 == 0 && al16(src) && al16(dst))
    return run_fwd<bf16_t, 8>(src, lds_, dst, ldd, N, H, W, C, mode, k, st);
  return run_fwd<bf16_t, 1>(src, lds_, dst, ldd, N, H, W, C, mode, k, st);
}

extern "C" int snws_resample_bwd(const void* ddst, int64_t ldd, const void* src, int64_t lds_, void* dsrc, int64_t ldds,
                                 int dtype, int N, int H, int W, int C, int mode, int k, int accumulate, void* stream) {
  if (int rc = check("snws_resample_bwd", dtype, N, H, W, C, mode, k)) return rc;
  SNWS_CHECK_ARG(ddst && dsrc && ldd >= C && ldds >= C, "snws_resample_bwd: NULL pointer or pixel stride < C");
  SNWS_CHECK_ARG(mode != MODE_MAXPOOL || (src && lds_ >= C), "snws_resample_bwd: max-pool backward needs the source");
  hipStream_t st = (hipStream_t)stream;
  const bool src_ok = mode != MODE_MAXPOOL || (al16(src));
  if (dtype == SNWS_F32) {
    if (C % 4 == 0 && ldd % 4 == 0 && ldds % 4 == 0 && (mode != MODE_MAXPOOL || lds_ % 4 == 0) && al16(ddst) && al16(dsrc) && src_ok)
      return run_bwd<float, 4>(ddst, ldd, src, lds_, dsrc, ldds, N, H, W, C, mode, k, accumulate, st);
    return run_bwd<float, 1>(ddst, ldd, src, lds_, dsrc, ldds, N, H, W, C, mode, k, accumulate, st);
  }
  if (C % 8 == 0 && ldd % 8 == 0 && ldds % 8 == 0 && (mode != MODE_MAXPOOL || lds_ % 8 == 0) && al16(ddst) && al16(dsrc) && src_ok)
    return run_bwd<bf16_t, 8>(ddst, ldd, src, lds_, dsrc, ldds, N, H, W, C, mode, k, accumulate, st);
  return run_bwd<bf16_t, 1>(ddst, ldd, src, lds_, dsrc, ldds, N, H, W, C, mode, k, accumulate, st);
}


// dst += src over an NHWC tensor (pixel strides may differ): the in-place sum of a gradient that several consumers
// contribute to, for contributors whose kernels cannot add on the fly.  f32 accumulation, one rounding.
namespace {
template <typename T, int V>
__global__ __launch_bounds__(256) void add_inplace_kernel(T* __restrict__ dst, int64_t ldd, const T* __restrict__ src, int64_t lds,
                                                          int64_t pixels, int C) {
  const int cg = C / V;
  const int64_t total = pixels * cg;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t px = i / cg;
    const int c = (int)(i - px * cg) * V;
    float a[V], b[V];
    Vec<T, V>::load(dst + px * ldd + c, a);
    Vec<T, V>::load(src + px * lds + c, b);
#pragma unroll
    for (int v = 0; v < V; ++v) a[v] += b[v];
    Vec<T, V>::store(dst + px * ldd + c, a);
  }
}
}  // namespace

extern "C" int snws_add_inplace(void* dst, int64_t ldd, const void* src, int64_t lds, int dtype, int64_t pixels, int C,
                                void* stream) {
  SNWS_CHECK_ARG(dst && src && pixels >= 0 && C >= 1 && ldd >= C && lds >= C, "snws_add_inplace: bad arguments");
  SNWS_CHECK_ARG(dtype == SNWS_F32 || dtype == SNWS_BF16, "snws_add_inplace: dtype %d is not F32/BF16", dtype);
  if (pixels == 0) return SNWS_OK;
  hipStream_t st = (hipStream_t)stream;
  const bool a16 = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
  if (dtype == SNWS_BF16) {
    const bool vec = a16 && C % 8 == 0 && ldd % 8 == 0 && lds % 8 == 0;
    const int64_t items = pixels * (vec ? C / 8 : C);
    const unsigned grid = (unsigned)(cdiv64(items, 256) < 65536 * 4 ? cdiv64(items, 256) : 65536 * 4);
    if (vec) add_inplace_kernel<bf16_t, 8><<<grid, 256, 0, st>>>((bf16_t*)dst, ldd, (const bf16_t*)src, lds, pixels, C);
    else add_inplace_kernel<bf16_t, 1><<<grid, 256, 0, st>>>((bf16_t*)dst, ldd, (const bf16_t*)src, lds, pixels, C);
  } else {
    const bool vec = a16 && C % 4 == 0 && ldd % 4 == 0 && lds % 4 == 0;
    const int64_t items = pixels * (vec ? C / 4 : C);
    const unsigned grid = (unsigned)(cdiv64(items, 256) < 65536 * 4 ? cdiv64(items, 256) : 65536 * 4);
    if (vec) add_inplace_kernel<float, 4><<<grid, 256, 0, st>>>((float*)dst, ldd, (const float*)src, lds, pixels, C);
    else add_inplace_kernel<float, 1><<<grid, 256, 0, st>>>((float*)dst, ldd, (const float*)src, lds, pixels, C);
  }
  SNWS_LAUNCH_CHECK("snws_add_inplace");
  return SNWS_OK;
}
