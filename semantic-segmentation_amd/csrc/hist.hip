// Confusion-matrix histogram for the Evaluator (replaces utils/metrics.py:40-45 of the reference).
//
// HBM-bound integer work: every pixel is read once (label + prediction, 12 B/pixel for f32 + i64; 16/32-byte
// loads, 4 pixels per lane) and costs one LDS integer atomic into a per-wave private histogram; one 1024-thread
// workgroup per CU flushes its non-zero bins with 64-bit global atomics at the end, so the result is exact and
// independent of scheduling.
#include "common.h"

namespace {

constexpr int HIST_THREADS = 1024;              // 16 waves per workgroup, one workgroup per CU
constexpr int HIST_WAVES = HIST_THREADS / 64;
constexpr int HIST_MAX_BLOCKS = 256;            // few blocks: the final same-address 64-bit atomics serialise per bin
                                                // (a full disk: 22 us with 256 workgroups, 25 with 128 or 512, 29 with 1024)

template <typename T>
struct alignas(sizeof(T) * 4) Pack4 {
  T v[4];
};

template <typename G>
__device__ __forceinline__ bool label_ok(G g, int C, int& gi) {
  // (gt >= 0) & (gt < C); NaN fails both compares.  astype('int') truncates toward zero.
  bool ok = (g >= (G)0) && (g < (G)C);
  gi = ok ? (int)g : 0;
  return ok;
}

// bin of one pixel: C * gt + pred inside the matrix, the two overflow bins beside it, or -1 for a pixel the mask drops
__device__ __forceinline__ int bin_of(bool ok, int gi, int64_t pred, int C, int C2) {
  if (!ok) return -1;
  const int64_t idx = (int64_t)C * gi + pred;
  return idx < 0 ? C2 + 1 : (idx >= C2 ? C2 : (int)idx);
}

__device__ __forceinline__ void count(unsigned* h, int C2, bool ok, int gi, int64_t pred, int C) {
  const int bin = bin_of(ok, gi, pred, C, C2);
  if (bin >= 0) atomicAdd(&h[bin], 1u);
}

// Four consecutive pixels of a lane.  Label maps are piecewise constant, so the four usually share a bin - one atomic
// of +4 - and often the whole wave does: then one lane adds the wave's count (same-address LDS atomics of a wave are
// served one lane at a time, which made a constant map 1.5x slower than random pixels).
__device__ __forceinline__ void count4(unsigned* h, const int (&bin)[4]) {
  const bool same = bin[0] == bin[1] && bin[1] == bin[2] && bin[2] == bin[3];
  const int first = __builtin_amdgcn_readfirstlane(bin[0]);
  if (__all(same && bin[0] == first)) {      // wave-uniform (every lane of a VEC iteration is active)
    if (first >= 0 && (threadIdx.x & 63) == 0) atomicAdd(&h[first], 4u * 64u);
    return;
  }
  if (same) {
    if (bin[0] >= 0) atomicAdd(&h[bin[0]], 4u);
    return;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (bin[e] >= 0) atomicAdd(&h[bin[e]], 1u);
}

template <typename G, typename P, bool VEC>
__global__ __launch_bounds__(HIST_THREADS) void hist_kernel(const G* __restrict__ gt, const P* __restrict__ pred,
                                                            int64_t n, int C, int copies, unsigned long long* counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned hist_lds[];
  const int C2 = C * C, bins = C2 + 2;
  for (int i = threadIdx.x; i < bins * copies; i += HIST_THREADS) hist_lds[i] = 0;
  __syncthreads();
  unsigned* h = hist_lds + ((threadIdx.x >> 6) % copies) * bins;
  const int64_t nthreads = (int64_t)gridDim.x * HIST_THREADS;
  const int64_t tid = (int64_t)blockIdx.x * HIST_THREADS + threadIdx.x;
  int64_t done = 0;
  if (VEC) {
    // 4 consecutive pixels per lane per load (16 B of f32 labels, 32 B of i64 predictions), U loads per tensor in flight
    constexpr int U = 2;
    const int64_t n4 = n >> 2;
    const Pack4<G>* g4 = reinterpret_cast<const Pack4<G>*>(gt);
    const Pack4<P>* p4 = reinterpret_cast<const Pack4<P>*>(pred);
    auto tally = [&](const Pack4<G>& ga, const Pack4<P>& pa) {
      int bin[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int a;
        const bool k = label_ok(ga.v[e], C, a);
        bin[e] = bin_of(k, a, (int64_t)pa.v[e], C, C2);
      }
      count4(h, bin);
    };
    // whole waves only in the unrolled loops (count4 assumes a full wave): the bound is wave-uniform because a wave's
    // lanes hold consecutive i and n4w is a multiple of 64
    const int64_t n4w = n4 & ~(int64_t)63;
    int64_t i = tid;
    for (; i + (U - 1) * nthreads < n4w; i += U * nthreads) {
      Pack4<G> ga[U];
      Pack4<P> pa[U];
#pragma unroll
      for (int u = 0; u < U; ++u) ga[u] = g4[i + u * nthreads];
#pragma unroll
      for (int u = 0; u < U; ++u) pa[u] = p4[i + u * nthreads];
#pragma unroll
      for (int u = 0; u < U; ++u) tally(ga[u], pa[u]);
    }
    for (; i < n4w; i += nthreads) tally(g4[i], p4[i]);
    done = n4w << 2;
  }
  for (int64_t i = done + tid; i < n; i += nthreads) {
    int a;
    bool k = label_ok(gt[i], C, a);
    count(h, C2, k, a, (int64_t)pred[i], C);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < bins; b += HIST_THREADS) {
    unsigned long long s = 0;
    for (int w = 0; w < copies; ++w) s += hist_lds[w * bins + b];
    if (s) atomicAdd(&counts[b], s);
  }
}

template <typename G, typename L>
__global__ __launch_bounds__(HIST_THREADS) void hist_argmax_kernel(const G* __restrict__ gt,
                                                                   const L* __restrict__ logits, int64_t ld,
                                                                   int64_t n, int C, int copies, unsigned long long* counts) {
  extern __shared__ __attribute__((aligned(16))) unsigned hist_lds[];
  const int C2 = C * C, bins = C2 + 2;
  for (int i = threadIdx.x; i < bins * copies; i += HIST_THREADS) hist_lds[i] = 0;
  __syncthreads();
  unsigned* h = hist_lds + ((threadIdx.x >> 6) % copies) * bins;
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
    for (int w = 0; w < copies; ++w) s += hist_lds[w * bins + b];
    if (s) atomicAdd(&counts[b], s);
  }
}

// private histogram copies per workgroup: one per wave while they fit in 48 KB of LDS
int hist_copies(int C) {
  int c = (48 * 1024) / ((C * C + 2) * 4);
  return c < 1 ? 1 : (c > HIST_WAVES ? HIST_WAVES : c);
}

int hist_grid(int64_t n) {
  int64_t b = cdiv64(n, (int64_t)HIST_THREADS * 8);
  if (b < 1) b = 1;
  if (b > HIST_MAX_BLOCKS) b = HIST_MAX_BLOCKS;
  return (int)b;
}

template <typename G>
int launch_hist_p(const void* gt, const void* pred, int pred_dtype, int64_t n, int C, int64_t* counts,
                  hipStream_t st) {
  const int copies = hist_copies(C);
  const size_t lds = (size_t)(C * C + 2) * copies * sizeof(unsigned);
  auto* out = reinterpret_cast<unsigned long long*>(counts);
  dim3 grid(hist_grid(n)), block(HIST_THREADS);
  auto aligned = [](const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; };
#define HK(PT)                                                                                             \
  if (aligned(gt, sizeof(G) * 4) && aligned(pred, sizeof(PT) * 4))                                         \
    hist_kernel<G, PT, true><<<grid, block, lds, st>>>((const G*)gt, (const PT*)pred, n, C, copies, out);          \
  else                                                                                                     \
    hist_kernel<G, PT, false><<<grid, block, lds, st>>>((const G*)gt, (const PT*)pred, n, C, copies, out);
  switch (pred_dtype) {
    case SNWS_I64: HK(int64_t) break;
    case SNWS_I32: HK(int32_t) break;
    case SNWS_U8: HK(uint8_t) break;
    default: snws_set_error("snws_confusion_hist: pred_dtype %d not one of I64/I32/U8", pred_dtype); return SNWS_ERR_INVALID;
  }
#undef HK
  SNWS_LAUNCH_CHECK("snws_confusion_hist");
  return SNWS_OK;
}

template <typename G>
int launch_hist_argmax_l(const void* gt, const void* logits, int logit_dtype, int64_t ld, int64_t n, int C,
                         int64_t* counts, hipStream_t st) {
  const int copies = hist_copies(C);
  const size_t lds = (size_t)(C * C + 2) * copies * sizeof(unsigned);
  auto* out = reinterpret_cast<unsigned long long*>(counts);
  int64_t b = cdiv64(n, HIST_THREADS * 4);
  dim3 grid((unsigned)(b < 1 ? 1 : (b > HIST_MAX_BLOCKS ? HIST_MAX_BLOCKS : b))), block(HIST_THREADS);
  switch (logit_dtype) {
    case SNWS_F32: hist_argmax_kernel<G, float><<<grid, block, lds, st>>>((const G*)gt, (const float*)logits, ld, n, C, copies, out); break;
    case SNWS_BF16: hist_argmax_kernel<G, bf16_t><<<grid, block, lds, st>>>((const G*)gt, (const bf16_t*)logits, ld, n, C, copies, out); break;
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
