// Weight standardisation (replaces modeling/weightstd.py:13-17 of the reference and its autograd).
//
// One workgroup per output channel: the fan-in (I*kh*kw <= a few thousand floats) is read once into registers,
// mean and unbiased std are exact two-pass f64 reductions, and the standardized filter is written in every
// layout its consumers want: f32 OIHW / OHWI for the direct kernels and the bf16 MFMA operand images
// (forward, and flipped + transposed for dgrad).  HBM-bound: 4 B read + <= 12 B written per parameter.
#include "common.h"
#include "pack.h"

namespace {

constexpr int WS_THREADS = 256;
constexpr int WS_MAX_PER_THREAD = 64;  // fan-in up to 16384

// PT = elements of the fan-in per thread (4, 16 or 64): the loops are fully unrolled over it, so a small fan-in must
// not pay for 64 predicated iterations.
template <bool STANDARDIZE, int PT>
__global__ __launch_bounds__(WS_THREADS) void ws_fwd_kernel(const float* __restrict__ w, int O, int I, int kh, int kw,
                                                            float* __restrict__ what_oihw,
                                                            float* __restrict__ what_ohwi, bf16_t* __restrict__ pk_fwd,
                                                            bf16_t* __restrict__ pk_dgrad) {
  __shared__ double scratch[16];
  const int o = blockIdx.x;
  const int taps = kh * kw, n = I * taps;
  const float* wo = w + (int64_t)o * n;
  float v[PT];
  double s = 0;
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = threadIdx.x + k * WS_THREADS;
    v[k] = i < n ? wo[i] : 0.f;
    s += v[k];
  }
  float sd = 1.f;
  if (STANDARDIZE) {
    const float mu = (float)(block_sum(s, scratch) / n);
    double ss = 0;
#pragma unroll
    for (int k = 0; k < PT; ++k) {
      const int i = threadIdx.x + k * WS_THREADS;
      v[k] = i < n ? v[k] - mu : 0.f;
      ss += (double)v[k] * v[k];
    }
    sd = (float)sqrt(block_sum(ss, scratch) / (n - 1)) + 1e-5f;
  }
  const PackGeom gf = pack_geom(I, O, taps);   // forward image: contraction over input channels
  const PackGeom gd = pack_geom(O, I, taps);   // dgrad image: contraction over output channels
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = threadIdx.x + k * WS_THREADS;
    if (i >= n) break;
    const float h = STANDARDIZE ? v[k] / sd : v[k];
    const int ci = i / taps, tap = i % taps;
    if (what_oihw) what_oihw[(int64_t)o * n + i] = h;
    if (what_ohwi) what_ohwi[((int64_t)o * taps + tap) * I + ci] = h;
    if (pk_fwd) pk_fwd[pack_index(gf, o, ci, tap)] = (bf16_t)h;
    if (pk_dgrad) pk_dgrad[pack_index(gd, ci, o, taps - 1 - tap)] = (bf16_t)h;
  }
}

// The gradient arrives in the weight-gradient kernels' layout (out ch, tap, in ch) while w and dw are (out ch, in ch,
// tap): the channel's gradient row is read the way it lies in memory - consecutive lanes, consecutive addresses - into
// LDS at its transposed position (stride `taps` words: conflict-free for the odd tap counts 1 and 9) and picked up from
// there; a gather straight from global touched nine rows per wave instruction and used 28 bytes of every line.
template <int PT>
__global__ __launch_bounds__(WS_THREADS) void ws_bwd_kernel(const float* __restrict__ w, const float* __restrict__ g,
                                                            int64_t ldg, float* __restrict__ dw, int O, int I, int kh,
                                                            int kw, int accumulate) {
  __shared__ double scratch[16];
  extern __shared__ float gt[];     // n floats: the gradient row in (in ch, tap) order
  const int o = blockIdx.x;
  const int taps = kh * kw, n = I * taps;
  const float* wo = w + (int64_t)o * n;
  float c[PT], gg[PT];
  double s = 0;
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = threadIdx.x + k * WS_THREADS;
    c[k] = i < n ? wo[i] : 0.f;
    s += c[k];
  }
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int j = threadIdx.x + k * WS_THREADS;     // (tap, in ch) order
    if (j < n) {
      const int tap = j / I, ci = j - tap * I;
      gg[k] = g[((int64_t)o * taps + tap) * ldg + ci];
    }
  }
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int j = threadIdx.x + k * WS_THREADS;
    if (j < n) {
      const int tap = j / I, ci = j - tap * I;
      gt[ci * taps + tap] = gg[k];
    }
  }
  const float mu = (float)(block_sum(s, scratch) / n);      // (its barriers also publish gt)
  double ss = 0, gc = 0;
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = threadIdx.x + k * WS_THREADS;
    if (i < n) {
      c[k] -= mu;
      gg[k] = gt[i];
    } else {
      c[k] = 0.f;
      gg[k] = 0.f;
    }
    ss += (double)c[k] * c[k];
    gc += (double)gg[k] * c[k];
  }
  const double sdev = sqrt(block_sum(ss, scratch) / (n - 1));
  gc = block_sum(gc, scratch);
  const double d = sdev + 1e-5;
  const double k2 = gc / (d * d * (n - 1) * sdev);
  const double inv_d = 1.0 / d;     // one f64 division per channel instead of one per element
  double ts = 0;
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = threadIdx.x + k * WS_THREADS;
    const double t = i < n ? (double)gg[k] * inv_d - k2 * c[k] : 0.0;
    gg[k] = (float)t;
    ts += t;
  }
  const float tm = (float)(block_sum(ts, scratch) / n);
#pragma unroll
  for (int k = 0; k < PT; ++k) {
    const int i = threadIdx.x + k * WS_THREADS;
    if (i < n) dw[(int64_t)o * n + i] = (accumulate ? dw[(int64_t)o * n + i] : 0.f) + (gg[k] - tm);
  }
}

}  // namespace

extern "C" int64_t snws_conv_pack_bytes(int c_in, int c_out, int kh, int kw) {
  if (c_in < 1 || c_out < 1 || kh < 1 || kw < 1) return 0;
  const PackGeom g = pack_geom(c_in, c_out, kh * kw);
  return pack_elems(g) * 2;
}

extern "C" int snws_ws_fwd(const float* w, int O, int I, int kh, int kw, float* what_oihw, float* what_ohwi,
                           void* pack_fwd, void* pack_dgrad, void* stream) {
  SNWS_CHECK_ARG(w, "snws_ws_fwd: w is NULL");
  SNWS_CHECK_ARG(O >= 1 && I >= 1 && kh >= 1 && kw >= 1, "snws_ws_fwd: bad shape (%d,%d,%d,%d)", O, I, kh, kw);
  const int64_t n = (int64_t)I * kh * kw;
  SNWS_CHECK_ARG(n >= 2, "snws_ws_fwd: fan-in %lld < 2 has no unbiased std", (long long)n);
  if (n > (int64_t)WS_THREADS * WS_MAX_PER_THREAD) SNWS_UNSUPPORTED("snws_ws_fwd: fan-in %lld > %d", (long long)n, WS_THREADS * WS_MAX_PER_THREAD);
#define WS_FWD(PT) ws_fwd_kernel<true, PT><<<O, WS_THREADS, 0, (hipStream_t)stream>>>(w, O, I, kh, kw, what_oihw, what_ohwi, (bf16_t*)pack_fwd, (bf16_t*)pack_dgrad)
  if (n <= 4 * WS_THREADS) WS_FWD(4); else if (n <= 16 * WS_THREADS) WS_FWD(16); else WS_FWD(64);
#undef WS_FWD
  SNWS_LAUNCH_CHECK("snws_ws_fwd");
  return SNWS_OK;
}

extern "C" int snws_conv_pack(const float* w, int O, int I, int kh, int kw, float* w_ohwi, void* pack_fwd,
                              void* pack_dgrad, void* stream) {
  SNWS_CHECK_ARG(w, "snws_conv_pack: w is NULL");
  SNWS_CHECK_ARG(O >= 1 && I >= 1 && kh >= 1 && kw >= 1, "snws_conv_pack: bad shape (%d,%d,%d,%d)", O, I, kh, kw);
  const int64_t n = (int64_t)I * kh * kw;
  if (n > (int64_t)WS_THREADS * WS_MAX_PER_THREAD) SNWS_UNSUPPORTED("snws_conv_pack: fan-in %lld > %d", (long long)n, WS_THREADS * WS_MAX_PER_THREAD);
#define WS_PACK(PT) ws_fwd_kernel<false, PT><<<O, WS_THREADS, 0, (hipStream_t)stream>>>(w, O, I, kh, kw, nullptr, w_ohwi, (bf16_t*)pack_fwd, (bf16_t*)pack_dgrad)
  if (n <= 4 * WS_THREADS) WS_PACK(4); else if (n <= 16 * WS_THREADS) WS_PACK(16); else WS_PACK(64);
#undef WS_PACK
  SNWS_LAUNCH_CHECK("snws_conv_pack");
  return SNWS_OK;
}

extern "C" int snws_ws_bwd(const float* w, const float* g_ohwi, int64_t ldg, float* dw, int O, int I, int kh, int kw,
                           int accumulate, void* stream) {
  SNWS_CHECK_ARG(w && g_ohwi && dw, "snws_ws_bwd: NULL pointer");
  SNWS_CHECK_ARG(O >= 1 && I >= 1 && kh >= 1 && kw >= 1 && ldg >= I, "snws_ws_bwd: bad shape");
  const int64_t n = (int64_t)I * kh * kw;
  SNWS_CHECK_ARG(n >= 2, "snws_ws_bwd: fan-in %lld < 2", (long long)n);
  if (n > (int64_t)WS_THREADS * WS_MAX_PER_THREAD) SNWS_UNSUPPORTED("snws_ws_bwd: fan-in %lld too large", (long long)n);
#define WS_BWD(PT) ws_bwd_kernel<PT><<<O, WS_THREADS, (size_t)n * sizeof(float), (hipStream_t)stream>>>(w, g_ohwi, ldg, dw, O, I, kh, kw, accumulate)
  if (n <= 4 * WS_THREADS) WS_BWD(4);
  else if (n <= 16 * WS_THREADS) WS_BWD(16);
  else {
    // up to 64 KB for the staged gradient row beside the static reduction scratch: above the default limit
    static PerDeviceOnce once;
    std::atomic<bool>* const attr_set = once.flag();
    if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_bwd_kernel<64>),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, WS_THREADS * WS_MAX_PER_THREAD * (int)sizeof(float));
      if (e != hipSuccess) {
        snws_set_error("snws_ws_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
        return SNWS_ERR_HIP;
      }
      if (attr_set) attr_set->store(true, std::memory_order_release);
    }
    WS_BWD(64);
  }
#undef WS_BWD
  SNWS_LAUNCH_CHECK("snws_ws_bwd");
  return SNWS_OK;
}
