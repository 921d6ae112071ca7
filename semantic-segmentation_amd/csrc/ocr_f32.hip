// Float32 form of the two OCR attention steps - SpatialGather_Module.forward and the attention inside
// _ObjectAttentionBlock.forward (modeling/UNet3p_res_ocr_SNws.py:21-28,87-100 of the reference) - for the float32
// parity tier of UNet3p_res_ocr_SNws and for any channel / class count (ocr.hip holds the bf16 matrix-core kernels for
// the benchmark's shapes).  Plain vector arithmetic in f32, every reduction in a fixed order: one wave per pixel with
// its lanes over the channels where a pixel is the unit of work, one workgroup per (image, class, channel block) where
// a sum runs over all pixels.  All tensors are pixel-major (NHWC): feats / q / dout (B, HW, C) with a pixel stride,
// class scores (B, HW, K) with a pixel stride, object-region tensors (B, K, C) dense.
#include "common.h"

namespace {

constexpr int O32_MAX_K = 64;      // classes (registers per lane in the per-pixel kernels)

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// stat[b][k] = (max over pixels of scale * score, sum over pixels of exp(scale * score - max)): one workgroup per (b, k)
__global__ __launch_bounds__(256) void ocr32_stats_kernel(const float* __restrict__ probs, int64_t ldp, int64_t HW, int K,
                                                          float scale, float* __restrict__ stat) {
  __shared__ float red[16];
  const int b = blockIdx.x / K, k = blockIdx.x % K;
  const float* p = probs + (int64_t)b * HW * ldp + k;
  float m = -INFINITY;
  for (int64_t i = threadIdx.x; i < HW; i += 256) m = fmaxf(m, scale * p[i * ldp]);
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < HW; i += 256) s += expf(scale * p[i * ldp] - m);
  s = block_sum(s, red);
  if (threadIdx.x == 0) {
    stat[2 * blockIdx.x] = m;
    stat[2 * blockIdx.x + 1] = s;
  }
}

// ctx[b][k][c] = sum_p softmax_p(scale * score[b][p][k]) * feats[b][p][c]: workgroup (b, k, block of 64 channels), four
// pixel lanes per channel, combined in a fixed order
__global__ __launch_bounds__(256) void ocr32_gather_fwd_kernel(const float* __restrict__ feats, int64_t ldf,
                                                               const float* __restrict__ probs, int64_t ldp,
                                                               const float* __restrict__ stat, float* __restrict__ ctx,
                                                               int64_t HW, int C, int K, float scale) {
  __shared__ float red[4][64];
  const int b = blockIdx.z, k = blockIdx.y, c = blockIdx.x * 64 + (threadIdx.x & 63), pl = threadIdx.x >> 6;
  const float m = stat[2 * (b * K + k)], inv = 1.f / stat[2 * (b * K + k) + 1];
  const float* p = probs + (int64_t)b * HW * ldp + k;
  const float* f = feats + (int64_t)b * HW * ldf + (c < C ? c : 0);
  float acc = 0.f;
  for (int64_t i = pl; i < HW; i += 4) acc = fmaf(expf(scale * p[i * ldp] - m) * inv, f[i * ldf], acc);
  red[pl][threadIdx.x & 63] = acc;
  __syncthreads();
  if (pl == 0 && c < C) ctx[((int64_t)b * K + k) * C + c] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// One wave per pixel: dP[k] = sum_c dctx[b][k][c] * feats[p][c]; dprobs[p][k] = scale * P[k] * (dP[k] - dot[b][k]) with
// dot[b][k] = <ctx, dctx>; dfeats[p][c] (+)= sum_k P[k] * dctx[b][k][c].
__global__ __launch_bounds__(256) void ocr32_gather_bwd_kernel(const float* __restrict__ feats, int64_t ldf,
                                                               const float* __restrict__ probs, int64_t ldp,
                                                               const float* __restrict__ stat, const float* __restrict__ dctx,
                                                               const float* __restrict__ dot, float* __restrict__ dfeats,
                                                               int64_t lddf, float* __restrict__ dprobs, int64_t lddp,
                                                               int B, int64_t HW, int C, int K, float scale, int accumulate) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (int64_t)B * HW) return;
  const int b = (int)(pix / HW);
  const float* f = feats + pix * ldf;
  const float* d = dctx + (int64_t)b * K * C;
  float P[O32_MAX_K];
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    float dp = 0.f;
    for (int c = lane; c < C; c += 64) dp = fmaf(d[(int64_t)k * C + c], f[c], dp);
    dp = wave_sum(dp);
    const float pk = expf(scale * probs[pix * ldp + k] - stat[2 * (b * K + k)]) / stat[2 * (b * K + k) + 1];
    P[k] = pk;
    if (lane == 0) dprobs[pix * lddp + k] = scale * pk * (dp - dot[b * K + k]);
  }
  for (int c = lane; c < C; c += 64) {
    float g = 0.f;
#pragma unroll 1
    for (int k = 0; k < K; ++k) g = fmaf(P[k], d[(int64_t)k * C + c], g);
    float* o = dfeats + pix * lddf + c;
    *o = accumulate ? *o + g : g;
  }
}

// One wave per pixel: sim = softmax_k(scale * q[p] . key[b][k]); out[p][c] = sum_k sim[k] * val[b][k][c]
__device__ __forceinline__ void attend_sim(const float* q, const float* key, int C, int K, float scale, int lane, float* sim) {
  float mx = -INFINITY;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(q[c], key[(int64_t)k * C + c], s);
    s = wave_sum(s) * scale;
    sim[k] = s;
    mx = fmaxf(mx, s);
  }
  float den = 0.f;
#pragma unroll 1
  for (int k = 0; k < K; ++k) { sim[k] = expf(sim[k] - mx); den += sim[k]; }
  const float inv = 1.f / den;
#pragma unroll 1
  for (int k = 0; k < K; ++k) sim[k] *= inv;
}

__global__ __launch_bounds__(256) void ocr32_attend_fwd_kernel(const float* __restrict__ q, int64_t ldq,
                                                               const float* __restrict__ key, const float* __restrict__ val,
                                                               float* __restrict__ out, int64_t ldo, int B, int64_t HW, int C,
                                                               int K, float scale) {
  const int lane = threadIdx.x & 63;
  const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pix >= (int64_t)B * HW) return;
  const int b = (int)(pix / HW);
  float sim[O32_MAX_K];
  attend_sim(q + pix * ldq, key + (int64_t)b * K * C, C, K, scale, lane, sim);
  const float* v = val + (int64_t)b * K * C;
  for (int c = lane; c < C; c += 64) {
    float g = 0.f;
#pragma unroll 1
    for (int k = 0; k < K; ++k) g = fmaf(sim[k], v[(int64_t)k * C + c], g);
    out[pix * ldo + c] = g;
  }
}

// Workgroup = (run of pixels of one image, block of 256 channels); its four waves take the run's pixels in turn.  Per
// pixel: sim again, dsim[k] = dout[p] . val[k], dl[k] = scale * sim[k] * (dsim[k] - sum_j sim[j] dsim[j]),
// dq[p][c] = sum_k dl[k] * key[k][c]; the wave accumulates dkey[k][c] += dl[k] * q[p][c] and dval[k][c] += sim[k] *
// dout[p][c] for its four channels of the block and all classes; the four waves' sums are combined through LDS in a
// fixed order and leave as this run's partials dkey_part / dval_part [B][runs][K][C].
__global__ __launch_bounds__(256) void ocr32_attend_bwd_kernel(const float* __restrict__ q, int64_t ldq,
                                                               const float* __restrict__ dout, int64_t lddo,
                                                               const float* __restrict__ key, const float* __restrict__ val,
                                                               float* __restrict__ dq, int64_t lddq, float* __restrict__ dkp,
                                                               float* __restrict__ dvp, int64_t HW, int C, int K, float scale,
                                                               int runs, int64_t run_len) {
  extern __shared__ float lds[];      // [4 waves][2][K][256]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.z, run = blockIdx.x, c0 = blockIdx.y * 256;
  const float* kb = key + (int64_t)b * K * C;
  const float* vb = val + (int64_t)b * K * C;
  float ak[4][16], av[4][16];          // [channel c0 + lane + 64 j][class chunk of 16]
  const int64_t p_lo = run * run_len, p_hi = p_lo + run_len < HW ? p_lo + run_len : HW;
  for (int k0 = 0; k0 < K; k0 += 16) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) ak[j][k] = av[j][k] = 0.f;
    for (int64_t p = p_lo + wave; p < p_hi; p += 4) {
      const int64_t pix = (int64_t)b * HW + p;
      const float* qp = q + pix * ldq;
      const float* dp = dout + pix * lddo;
      float sim[O32_MAX_K], dl[O32_MAX_K];
      attend_sim(qp, kb, C, K, scale, lane, sim);
      float mix = 0.f;
#pragma unroll 1
      for (int k = 0; k < K; ++k) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(dp[c], vb[(int64_t)k * C + c], s);
        s = wave_sum(s);
        dl[k] = s;
        mix = fmaf(sim[k], s, mix);
      }
#pragma unroll 1
      for (int k = 0; k < K; ++k) dl[k] = scale * sim[k] * (dl[k] - mix);
      if (k0 == 0 && blockIdx.y == 0) {        // dq once per pixel: by the first channel block's workgroup, all channels
        for (int c = lane; c < C; c += 64) {
          float g = 0.f;
#pragma unroll 1
          for (int k = 0; k < K; ++k) g = fmaf(dl[k], kb[(int64_t)k * C + c], g);
          dq[pix * lddq + c] = g;
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int c = c0 + lane + 64 * j;
        const float qv = c < C ? qp[c] : 0.f, dv = c < C ? dp[c] : 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
          if (k0 + k < K) {
            ak[j][k] = fmaf(dl[k0 + k], qv, ak[j][k]);
            av[j][k] = fmaf(sim[k0 + k], dv, av[j][k]);
          }
        }
      }
    }
    // combine the four waves (fixed order) and store this class chunk of the run's partials
    float* mine = lds + (size_t)wave * 2 * 16 * 256;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        mine[k * 256 + lane + 64 * j] = ak[j][k];
        mine[(16 + k) * 256 + lane + 64 * j] = av[j][k];
      }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * 16 * 256; i += 256) {
      const int which = i / (16 * 256), k = (i / 256) % 16, cc = i % 256;
      const int c = c0 + cc;
      if (k0 + k < K && c < C) {
        const float s = (lds[i] + lds[i + 2 * 16 * 256]) + (lds[i + 4 * 16 * 256] + lds[i + 6 * 16 * 256]);
        float* dst = which == 0 ? dkp : dvp;
        dst[(((int64_t)b * runs + run) * K + (k0 + k)) * C + c] = s;
      }
    }
    __syncthreads();
  }
}

}  // namespace

extern "C" int snws_ocr32_runs(int64_t HW) {
  int64_t r = (HW + 255) / 256;
  if (r < 1) r = 1;
  if (r > 128) r = 128;
  return (int)r;
}

#define O32_CHECK(fn)                                                                                                  \
  SNWS_CHECK_ARG(B >= 1 && HW >= 1 && C >= 1 && K >= 1 && K <= O32_MAX_K, "%s: B, HW, C >= 1 and 1 <= K <= %d", fn, \
                 O32_MAX_K);                                                                                           \
  SNWS_CHECK_ARG((int64_t)B * HW < (1ll << 33), "%s: too many pixels", fn)

extern "C" int snws_ocr32_gather_fwd(const float* feats, int64_t ldf, const float* probs, int64_t ldp, float* stat, float* ctx,
                                     int B, int64_t HW, int C, int K, float scale, void* stream) {
  O32_CHECK("snws_ocr32_gather_fwd");
  SNWS_CHECK_ARG(feats && probs && stat && ctx && ldf >= C && ldp >= K, "snws_ocr32_gather_fwd: NULL pointer or bad stride");
  hipStream_t st = (hipStream_t)stream;
  ocr32_stats_kernel<<<B * K, 256, 0, st>>>(probs, ldp, HW, K, scale, stat);
  SNWS_LAUNCH_CHECK("snws_ocr32_gather_fwd(1)");
  ocr32_gather_fwd_kernel<<<dim3(cdiv(C, 64), K, B), 256, 0, st>>>(feats, ldf, probs, ldp, stat, ctx, HW, C, K, scale);
  SNWS_LAUNCH_CHECK("snws_ocr32_gather_fwd(2)");
  return SNWS_OK;
}

extern "C" int snws_ocr32_gather_bwd(const float* feats, int64_t ldf, const float* probs, int64_t ldp, const float* stat,
                                     const float* dctx, const float* dot, float* dfeats, int64_t lddf, float* dprobs,
                                     int64_t lddp, int B, int64_t HW, int C, int K, float scale, int accumulate, void* stream) {
  O32_CHECK("snws_ocr32_gather_bwd");
  SNWS_CHECK_ARG(feats && probs && stat && dctx && dot && dfeats && dprobs && ldf >= C && ldp >= K && lddf >= C && lddp >= K,
                 "snws_ocr32_gather_bwd: NULL pointer or bad stride");
  const int64_t blocks = cdiv64((int64_t)B * HW, 4);
  ocr32_gather_bwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(feats, ldf, probs, ldp, stat, dctx, dot, dfeats, lddf,
                                                                             dprobs, lddp, B, HW, C, K, scale, accumulate);
  SNWS_LAUNCH_CHECK("snws_ocr32_gather_bwd");
  return SNWS_OK;
}

extern "C" int snws_ocr32_attend_fwd(const float* q, int64_t ldq, const float* key, const float* val, float* out, int64_t ldo,
                                     int B, int64_t HW, int C, int K, float scale, void* stream) {
  O32_CHECK("snws_ocr32_attend_fwd");
  SNWS_CHECK_ARG(q && key && val && out && ldq >= C && ldo >= C, "snws_ocr32_attend_fwd: NULL pointer or bad stride");
  const int64_t blocks = cdiv64((int64_t)B * HW, 4);
  ocr32_attend_fwd_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(q, ldq, key, val, out, ldo, B, HW, C, K, scale);
  SNWS_LAUNCH_CHECK("snws_ocr32_attend_fwd");
  return SNWS_OK;
}

extern "C" int snws_ocr32_attend_bwd(const float* q, int64_t ldq, const float* dout, int64_t lddo, const float* key,
                                     const float* val, float* dq, int64_t lddq, float* dkey_part, float* dval_part, int B,
                                     int64_t HW, int C, int K, float scale, void* stream) {
  O32_CHECK("snws_ocr32_attend_bwd");
  SNWS_CHECK_ARG(q && dout && key && val && dq && dkey_part && dval_part && ldq >= C && lddo >= C && lddq >= C,
                 "snws_ocr32_attend_bwd: NULL pointer or bad stride");
  const int runs = snws_ocr32_runs(HW);
  const int64_t run_len = cdiv64(HW, runs);
  const size_t lds = (size_t)4 * 2 * 16 * 256 * sizeof(float);      // 128 KB
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ocr32_attend_bwd_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
      snws_set_error("snws_ocr32_attend_bwd: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  ocr32_attend_bwd_kernel<<<dim3(runs, cdiv(C, 256), B), 256, lds, (hipStream_t)stream>>>(q, ldq, dout, lddo, key, val, dq, lddq,
                                                                                          dkey_part, dval_part, HW, C, K, scale,
                                                                                          runs, run_len);
  SNWS_LAUNCH_CHECK("snws_ocr32_attend_bwd");
  return SNWS_OK;
}
