// Confusion-matrix histogram for the Evaluator (replaces utils/metrics.py:40-45 of the reference).
//
// HBM-bound integer work: every pixel is read once (label + prediction, 12 B/pixel for f32 + i64) and costs
// one LDS integer atomic into a per-wave private histogram; a block flushes its non-zero bins with 64-bit
// global atomics at the end, so the result is exact and independent of scheduling.
#include "common.h"

namespace {

constexpr int HIST_THREADS = 256;
constexpr int HIST_WAVES = HIST_THREADS / 64;
constexpr int HIST_MAX_BLOCKS = 1024;

template <typename G>
__device__ __forceinline__ bool label_ok(G g, int C, int& gi) {
  // (gt >= 0) & (gt < C); NaN fails both compares.  astype('int') truncates toward zero.
  bool ok = (g >= (G)0) && (g < (G)C);
  gi = ok ? (int)g : 0;
  return ok;
}

__device__ __forceinline__ void count(unsigned* h, int C2, bool ok, int gi, int64_t pred, int C) {
  if (!ok) return;
  int64_t idx = (int64_t)C * gi + pred;
  int bin = idx < 0 ? C2 + 1 : (idx >= C2 ? C2 : (int)idx);
  atomicAdd(&h[bin], 1u);
}

template <typename G, typename P>
__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(const G* __restrict__ gt, const P* __restrict__ pred,
                                                            int64_t n, int C, unsigned long long* counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned hist_lds[];
  const int C2 = C * C, bins = C2 + 2;
  for (int i = threadIdx.x; i < bins * HIST_WAVES; i += HIST_THREADS) hist_lds[i] = 0;
  __syncthreads();
  unsigned* h = hist_lds + (threadIdx.x >> 6) * bins;
  const int64_t stride = (int64_t)gridDim.x * HIST_THREADS;
  // 4 pixels per thread per step keeps several independent loads in flight
  int64_t i = ((int64_t)blockIdx.x * HIST_THREADS + threadIdx.x);
  for (; i + 3 * stride < n; i += 4 * stride) {
    G g0 = gt[i], g1 = gt[i + stride], g2 = gt[i + 2 * stride], g3 = gt[i + 3 * stride];
    P p0 = pred[i], p1 = pred[i + stride], p2 = pred[i + 2 * stride], p3 = pred[i + 3 * stride];
    int a0, a1, a2, a3;
    bool k0 = label_ok(g0, C, a0), k1 = label_ok(g1, C, a1), k2 = label_ok(g2, C, a2), k3 = label_ok(g3, C, a3);
    count(h, C2, k0, a0, (int64_t)p0, C);
    count(h, C2, k1, a1, (int64_t)p1, C);
    count(h, C2, k2, a2, (int64_t)p2, C);
    count(h, C2, k3, a3, (int64_t)p3, C);
  }
  for (; i < n; i += stride) {
    int a;
    bool k = label_ok(gt[i], C, a);
    count(h, C2, k, a, (int64_t)pred[i], C);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += HIST_THREADS) {
    unsigned long long s = 0;
    for (int w = 0; w < HIST_WAVES; ++w) s += hist_lds[w * bins + b];
    if (s) atomicAdd(&counts[b], s);
  }
}

template <typename G, typename L>
__global__ __launch_bounds__(HIST_THREADS) void hist_argmax_kernel(const G* __restrict__ gt,
                                                                   const L* __restrict__ logits, int64_t ld,
                                                                   int64_t n, int C, unsigned long long* counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned hist_lds[];
  const int C2 = C * C, bins = C2 + 2;
  for (int i = threadIdx.x; i < bins * HIST_WAVES; i += HIST_THREADS) hist_lds[i] = 0;
  __syncthreads();
  unsigned* h = hist_lds + (threadIdx.x >> 6) * bins;
  const int64_t stride = (int64_t)gridDim.x * HIST_THREADS;
  for (int64_t i = (int64_t)blockIdx.x * HIST_THREADS + threadIdx.x; i < n; i += stride) {
    int gi;
    bool ok = label_ok(gt[i], C, gi);
    const L* row = logits + i * ld;
    float best = (float)row[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
      float v = (float)row[c];
      // first maximum wins; the first NaN beats everything (numpy.argmax)
      if ((v > best) || (v != v && best == best)) { best = v; arg = c; }
    }
    count(h, C2, ok, gi, arg, C);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += HIST_THREADS) {
    unsigned long long s = 0;
    for (int w = 0; w < HIST_WAVES; ++w) s += hist_lds[w * bins + b];
    if (s) atomicAdd(&counts[b], s);
  }
}

int hist_grid(int64_t n) {
  int64_t b = cdiv64(n, (int64_t)HIST_THREADS * 16);
  if (b < 1) b = 1;
  if (b > HIST_MAX_BLOCKS) b = HIST_MAX_BLOCKS;
  return (int)b;
}

template <typename G>
int launch_hist_p(const void* gt, const void* pred, int pred_dtype, int64_t n, int C, int64_t* counts,
                  hipStream_t st) {
  const size_t lds = (size_t)(C * C + 2) * HIST_WAVES * sizeof(unsigned);
  auto* out = reinterpret_cast<unsigned long long*>(counts);
  dim3 grid(hist_grid(n)), block(HIST_THREADS);
  switch (pred_dtype) {
    case SNWS_I64: hist_kernel<G, int64_t><<<grid, block, lds, st>>>((const G*)gt, (const int64_t*)pred, n, C, out); break;
    case SNWS_I32: hist_kernel<G, int32_t><<<grid, block, lds, st>>>((const G*)gt, (const int32_t*)pred, n, C, out); break;
    case SNWS_U8: hist_kernel<G, uint8_t><<<grid, block, lds, st>>>((const G*)gt, (const uint8_t*)pred, n, C, out); break;
    default: snws_set_error("snws_confusion_hist: pred_dtype %d not one of I64/I32/U8", pred_dtype); return SNWS_ERR_INVALID;
  }
  SNWS_LAUNCH_CHECK("snws_confusion_hist");
  return SNWS_OK;
}

template <typename G>
int launch_hist_argmax_l(const void* gt, const void* logits, int logit_dtype, int64_t ld, int64_t n, int C,
                         int64_t* counts, hipStream_t st) {
  const size_t lds = (size_t)(C * C + 2) * HIST_WAVES * sizeof(unsigned);
  auto* out = reinterpret_cast<unsigned long long*>(counts);
  int64_t b = cdiv64(n, HIST_THREADS * 4);
  dim3 grid((unsigned)(b < 1 ? 1 : (b > HIST_MAX_BLOCKS ? HIST_MAX_BLOCKS : b))), block(HIST_THREADS);
  switch (logit_dtype) {
    case SNWS_F32: hist_argmax_kernel<G, float><<<grid, block, lds, st>>>((const G*)gt, (const float*)logits, ld, n, C, out); break;
    case SNWS_BF16: hist_argmax_kernel<G, bf16_t><<<grid, block, lds, st>>>((const G*)gt, (const bf16_t*)logits, ld, n, C, out); break;
    default: snws_set_error("snws_confusion_hist_argmax: logit_dtype %d not F32/BF16", logit_dtype); return SNWS_ERR_INVALID;
  }
  SNWS_LAUNCH_CHECK("snws_confusion_hist_argmax");
  return SNWS_OK;
}

}  // namespace

extern "C" int snws_confusion_hist(const void* gt, int gt_dtype, const void* pred, int pred_dtype, int64_t n_pixels,
                                   int num_class, int64_t* counts, void* stream) {
  SNWS_CHECK_ARG(num_class >= 1 && num_class <= 64, "snws_confusion_hist: num_class %d outside [1,64]", num_class);
  SNWS_CHECK_ARG(n_pixels >= 0, "snws_confusion_hist: negative pixel count");
  SNWS_CHECK_ARG(counts != nullptr, "snws_confusion_hist: counts is NULL");
  if (n_pixels == 0) return SNWS_OK;
  SNWS_CHECK_ARG(gt && pred, "snws_confusion_hist: NULL input");
  hipStream_t st = (hipStream_t)stream;
  switch (gt_dtype) {
    case SNWS_F32: return launch_hist_p<float>(gt, pred, pred_dtype, n_pixels, num_class, counts, st);
    case SNWS_F64: return launch_hist_p<double>(gt, pred, pred_dtype, n_pixels, num_class, counts, st);
    case SNWS_I64: return launch_hist_p<int64_t>(gt, pred, pred_dtype, n_pixels, num_class, counts, st);
    case SNWS_I32: return launch_hist_p<int32_t>(gt, pred, pred_dtype, n_pixels, num_class, counts, st);
    case SNWS_U8: return launch_hist_p<uint8_t>(gt, pred, pred_dtype, n_pixels, num_class, counts, st);
    default: snws_set_error("snws_confusion_hist: gt_dtype %d unsupported", gt_dtype); return SNWS_ERR_INVALID;
  }
}

extern "C" int snws_confusion_hist_argmax(const void* gt, int gt_dtype, const void* logits, int logit_dtype,
                                          int64_t ld, int64_t n_pixels, int num_class, int64_t* counts,
                                          void* stream) {
  SNWS_CHECK_ARG(num_class >= 1 && num_class <= 64, "snws_confusion_hist_argmax: num_class %d outside [1,64]", num_class);
  SNWS_CHECK_ARG(n_pixels >= 0 && ld >= num_class, "snws_confusion_hist_argmax: bad sizes");
  SNWS_CHECK_ARG(counts != nullptr, "snws_confusion_hist_argmax: counts is NULL");
  if (n_pixels == 0) return SNWS_OK;
  SNWS_CHECK_ARG(gt && logits, "snws_confusion_hist_argmax: NULL input");
  hipStream_t st = (hipStream_t)stream;
  switch (gt_dtype) {
    case SNWS_F32: return launch_hist_argmax_l<float>(gt, logits, logit_dtype, ld, n_pixels, num_class, counts, st);
    case SNWS_F64: return launch_hist_argmax_l<double>(gt, logits, logit_dtype, ld, n_pixels, num_class, counts, st);
    case SNWS_I64: return launch_hist_argmax_l<int64_t>(gt, logits, logit_dtype, ld, n_pixels, num_class, counts, st);
    case SNWS_I32: return launch_hist_argmax_l<int32_t>(gt, logits, logit_dtype, ld, n_pixels, num_class, counts, st);
    case SNWS_U8: return launch_hist_argmax_l<uint8_t>(gt, logits, logit_dtype, ld, n_pixels, num_class, counts, st);
    default: snws_set_error("snws_confusion_hist_argmax: gt_dtype %d unsupported", gt_dtype); return SNWS_ERR_INVALID;
  }
}
