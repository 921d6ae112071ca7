// Per-pixel classification kernels over NHWC logits (HBM-bound, one thread per pixel):
//  * fused soft-max cross-entropy forward + backward - replaces nn.CrossEntropyLoss(ignore_index=255) and its
//    autograd at utils/loss.py:146-159 of the reference (called at train.py:252);
//  * arg-max + stitch of one inference tile into the full-disk prediction - replaces torch.max(outputs, 1) and the
//    pred_arr[yoff:yoff+B, xoff:xoff+B] = tile assignment at predict.py:196,248-253.
#include "common.h"

namespace {

constexpr int CE_THREADS = 256;
constexpr int CE_MAX_C = 64;

template <typename T>
__device__ __forceinline__ float ld1(const T* p) { return (float)(*p); }

// loss_part[block] = sum over the block's pixels of w[target] * (logsumexp - logit[target]) for non-ignored pixels
// (fixed summation order: bit-reproducible); dlogits = w[target] * (softmax - onehot) * scale for them, 0 for ignored
// pixels and for the padding channels C..c_write-1.  w = class_weight (nn.CrossEntropyLoss(weight=...),
// utils/loss.py:148) or 1 when NULL; a label outside [0, C) that is not ignore_index weighs 0.
// MAXC: compile-time bound of the class loops (16 for the usual <= 16 classes: a quarter of the predicated
// instructions of the general 64-class form, which made the pass instruction-bound at 2 TB/s)
template <typename T, int V, int MAXC>
__global__ __launch_bounds__(CE_THREADS) void ce_kernel(const T* __restrict__ logits, int64_t ld, const float* __restrict__ target,
                                                        int64_t npix, int C, int ignore_index, const float* __restrict__ scale_ptr,
                                                        const float* __restrict__ class_weight,
                                                        float* __restrict__ loss_part, T* __restrict__ dlogits, int64_t ldd,
                                                        int c_write) {
  __shared__ float red[16];
  const float scale0 = scale_ptr ? *scale_ptr : 1.f;
  float acc = 0.f;
  for (int64_t p = (int64_t)blockIdx.x * CE_THREADS + threadIdx.x; p < npix; p += (int64_t)gridDim.x * CE_THREADS) {
    const T* row = logits + p * ld;
    float v[MAXC];
    if (V == 8) {
#pragma unroll
      for (int c0 = 0; c0 < MAXC; c0 += 8) {
        if (c0 < C) {
          float f[8];
          Vec<T, V>::load(row + c0, *reinterpret_cast<float(*)[V]>(f));
#pragma unroll
          for (int e = 0; e < 8; ++e) v[c0 + e] = f[e];
        }
      }
    } else {
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (c < C) v[c] = ld1(row + c);
    }
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) m = fmaxf(m, v[c]);
    const float tf = target[p];
    const int t = (int)tf;                       // .long(): truncation toward zero
    const bool valid = t != ignore_index;
    const float wt = class_weight ? ((t >= 0 && t < C) ? class_weight[t] : 0.f) : 1.f;
    const float scale = scale0 * wt;
    float s = 0.f, gap = 0.f;                    // gap = logit[t] - max, taken BEFORE exponentiating: exp(gap) may underflow
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
      if (c < C) {
        const float d = v[c] - m;
        if (c == t) gap = d;
        v[c] = __expf(d);
        s += v[c];
      }
    if (valid) acc += wt * (__logf(s) - gap);            // = logsumexp - logit[t], finite for any finite logits
    if (dlogits) {
      const float k = valid ? scale / s : 0.f;
      T* drow = dlogits + p * ldd;
      if (V == 8) {
#pragma unroll
        for (int c0 = 0; c0 < MAXC; c0 += 8) {
          if (c0 < c_write) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const int c = c0 + e;
              f[e] = (c < C) ? v[c] * k - ((valid && c == t) ? scale : 0.f) : 0.f;
            }
            Vec<T, V>::store(drow + c0, *reinterpret_cast<float(*)[V]>(f));
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < c_write) drow[c] = (T)((c < C) ? v[c] * k - ((valid && c == t) ? scale : 0.f) : 0.f);
      }
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_stitch_kernel(const T* __restrict__ logits, int64_t ld, int h, int w, int C,
                                                            int64_t* __restrict__ pred, int H, int W, int y0, int x0,
                                                            int remap) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (int64_t)h * w) return;
  const int y = (int)(p / w), x = (int)(p - (int64_t)y * w);
  const int gy = y0 + y, gx = x0 + x;
  if (gy < 0 || gy >= H || gx < 0 || gx >= W) return;
  const T* row = logits + p * ld;
  // first maximum wins; the first NaN beats everything (torch.max / numpy.argmax)
  float best = ld1(row);
  int bi = 0;
  for (int c = 1; c < C; ++c) {
    const float f = ld1(row + c);
    if (!(best != best) && (f > best || f != f)) { best = f; bi = c; }
  }
  if (remap && (bi == 0 || bi == 10)) bi = 255;
  pred[(int64_t)gy * W + gx] = bi;
}

}  // namespace

extern "C" int snws_ce_blocks(int64_t n_pixels) {
  const int64_t b = cdiv64(n_pixels > 0 ? n_pixels : 1, CE_THREADS);
  return (int)(b < 2048 ? b : 2048);
}

extern "C" int snws_ce_fwd_bwd_w(const void* logits, int dtype, int64_t ld, const float* target, int64_t n_pixels, int C,
                                 int ignore_index, const float* scale, const float* class_weight, float* loss_part,
                                 void* dlogits, int64_t ldd, int c_write, void* stream) {
  SNWS_CHECK_ARG(logits && target && loss_part, "snws_ce_fwd_bwd: NULL pointer");
  SNWS_CHECK_ARG(C >= 1 && C <= CE_MAX_C, "snws_ce_fwd_bwd: %d classes outside [1,%d]", C, CE_MAX_C);
  SNWS_CHECK_ARG(n_pixels >= 0 && ld >= C, "snws_ce_fwd_bwd: bad sizes");
  SNWS_CHECK_ARG(!dlogits || (c_write >= C && c_write <= CE_MAX_C && ldd >= c_write), "snws_ce_fwd_bwd: bad gradient layout");
  SNWS_CHECK_ARG(dtype == SNWS_F32 || dtype == SNWS_BF16, "snws_ce_fwd_bwd: dtype %d not F32/BF16", dtype);
  hipStream_t st = (hipStream_t)stream;
  const int grid = snws_ce_blocks(n_pixels);
  // vector path: bf16 rows of whole 16-byte groups
  const bool vec = dtype == SNWS_BF16 && ld % 8 == 0 && ((uintptr_t)logits & 15) == 0 &&
                   (!dlogits || (ldd % 8 == 0 && c_write % 8 == 0 && ((uintptr_t)dlogits & 15) == 0)) &&
                   ((C + 7) / 8) * 8 <= ld;
#define CE_LAUNCH(T, V, MAXC)                                                                                       \
  ce_kernel<T, V, MAXC><<<grid, CE_THREADS, 0, st>>>((const T*)logits, ld, target, n_pixels, C, ignore_index, scale,  \
                                                     class_weight, loss_part, (T*)dlogits, ldd, c_write)
  const bool small = C <= 16 && (!dlogits || c_write <= 16);
  if (dtype == SNWS_BF16) {
    if (vec && small) CE_LAUNCH(bf16_t, 8, 16);
    else if (vec) CE_LAUNCH(bf16_t, 8, 64);
    else if (small) CE_LAUNCH(bf16_t, 1, 16);
    else CE_LAUNCH(bf16_t, 1, 64);
  } else {
    if (small) CE_LAUNCH(float, 1, 16);
    else CE_LAUNCH(float, 1, 64);
  }
#undef CE_LAUNCH
  SNWS_LAUNCH_CHECK("snws_ce_fwd_bwd");
  return SNWS_OK;
}

extern "C" int snws_ce_fwd_bwd(const void* logits, int dtype, int64_t ld, const float* target, int64_t n_pixels, int C,
                               int ignore_index, const float* scale, float* loss_part, void* dlogits, int64_t ldd,
                               int c_write, void* stream) {
  return snws_ce_fwd_bwd_w(logits, dtype, ld, target, n_pixels, C, ignore_index, scale, nullptr, loss_part, dlogits, ldd,
                           c_write, stream);
}

extern "C" int snws_argmax_stitch(const void* logits, int dtype, int64_t ld, int h, int w, int C, int64_t* pred, int H,
                                  int W, int y0, int x0, int remap_background, void* stream) {
  SNWS_CHECK_ARG(logits && pred, "snws_argmax_stitch: NULL pointer");
  SNWS_CHECK_ARG(h >= 1 && w >= 1 && H >= 1 && W >= 1 && C >= 1 && ld >= C, "snws_argmax_stitch: bad sizes");
  SNWS_CHECK_ARG(dtype == SNWS_F32 || dtype == SNWS_BF16, "snws_argmax_stitch: dtype %d not F32/BF16", dtype);
  hipStream_t st = (hipStream_t)stream;
  const unsigned grid = (unsigned)cdiv64((int64_t)h * w, 256);
  if (dtype == SNWS_BF16)
    argmax_stitch_kernel<bf16_t><<<grid, 256, 0, st>>>((const bf16_t*)logits, ld, h, w, C, pred, H, W, y0, x0, remap_background);
  else
    argmax_stitch_kernel<float><<<grid, 256, 0, st>>>((const float*)logits, ld, h, w, C, pred, H, W, y0, x0, remap_background);
  SNWS_LAUNCH_CHECK("snws_argmax_stitch");
  return SNWS_OK;
}
