// SwitchNorm forward/backward for NHWC activations (replaces modeling/snorm.py:41-81 of the reference and its
// autograd, fused with the ReLU that follows it in ConvSnRelu).
//
// HBM-bound streaming kernels.  The two big passes per direction read 16 B per lane with a fixed channel group
// per thread, so the per-(n,c) coefficients live in registers and consecutive lanes touch consecutive bytes:
//   forward : [stats (or the conv epilogue's partial moments)] -> finalize -> mix_a -> (all-reduce) -> mix_b -> apply
//   backward: bwd_reduce -> bwd_mix_a -> (all-reduce) -> bwd_mix_b -> bwd_apply
// Moments are accumulated as sums of (x - shift) and (x - shift)^2 per tile in f32 and combined across tiles in
// f64, in a fixed order, so results are bit-reproducible run to run.
#include "common.h"
#include <stdlib.h>
#include "sn_ctx.h"

namespace {

constexpr int SN_THREADS = 256;
// pixel rows a thread of the backward apply pass keeps in flight (three tensors per row with a stored-output mask)
template <bool YMASK>
constexpr int SN_BWD_ROWS = YMASK ? 2 : 4;

struct SnMap {  // how 256 threads cover (pixel rows) x (channel groups)
  int CG, cgb, PR, active;
};
inline SnMap sn_map(int C, int V) {
  SnMap m;
  m.CG = C / V;
  m.cgb = m.CG < SN_THREADS ? m.CG : SN_THREADS;
  m.PR = SN_THREADS / m.cgb;
  m.active = m.cgb * m.PR;
  return m;
}

inline int sn_tiles(int N, int64_t HW, int C) {
  int64_t per_img = HW * (int64_t)C;
  int64_t T = cdiv64(per_img, 32768);
  int64_t cap = 4096 / (N > 0 ? N : 1);
  if (cap < 1) cap = 1;
  if (T > cap) T = cap;
  if (T < 1) T = 1;
  int64_t PT = cdiv64(HW, T);
  return (int)cdiv64(HW, PT);
}

// V consecutive per-channel coefficients: 16-byte loads when the address allows (a workgroup's first data loads wait
// behind these, so their count is the start-up latency of every workgroup of the streaming passes)
template <int V>
__device__ __forceinline__ void load_coef(const float* __restrict__ p, float (&out)[V]) {
  if constexpr (V % 4 == 0) {
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
#pragma unroll
      for (int k = 0; k < V / 4; ++k) {
        const float4 q = reinterpret_cast<const float4*>(p)[k];
        out[4 * k] = q.x; out[4 * k + 1] = q.y; out[4 * k + 2] = q.z; out[4 * k + 3] = q.w;
      }
      return;
    }
  }
#pragma unroll
  for (int v = 0; v < V; ++v) out[v] = p[v];
}

// ------------------------------------------------------------------------------------------- pass 1: moments
template <typename T, int V>
__global__ __launch_bounds__(SN_THREADS) void sn_stats_kernel(const T* __restrict__ x, int64_t HW, int C, int64_t ldx,
                                                              int PT, int Tn, SnMap m, float* __restrict__ part,
                                                              float* __restrict__ shift_out) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][PR][cgb*V]
  // PT == 0: the workgroups of an image interleave pieces of 4 PR pixels (see sn_apply_kernel); a partial-sum row then
  // belongs to a strided set of pixels instead of a contiguous range - its consumers only ever sum the rows
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int64_t PS = 4 * (int64_t)m.PR;
  const int64_t p0 = PT > 0 ? (int64_t)t * PT : (int64_t)t * PS;
  const int64_t p1 = PT > 0 ? (p0 + PT < HW ? p0 + PT : HW) : HW;
  const int64_t pstep = PT > 0 ? PS : (int64_t)gridDim.x * PS;
  const T* xn = x + (int64_t)n * HW * ldx;
  const int cgl = tid % m.cgb, prow = tid / m.cgb;
  const int row = m.cgb * V;
  for (int cg0 = 0; cg0 < m.CG; cg0 += m.cgb) {
    const int cg = cg0 + cgl;
    const bool valid = tid < m.active && cg < m.CG;
    float S[V], Q[V], sh[V];
#pragma unroll
    for (int v = 0; v < V; ++v) S[v] = Q[v] = sh[v] = 0.f;
    if (valid) {
      Vec<T, V>::load(xn + (int64_t)cg * V, sh);  // first pixel of the image: the offset of the shifted sums
      if (t == 0 && prow == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) shift_out[(int64_t)n * C + cg * V + v] = sh[v];
      }
      int64_t p = p0 + prow;
      for (; p + 3 * m.PR < p1; p += pstep) {
        float a[V], b[V], c[V], d[V];
        Vec<T, V>::load(xn + p * ldx + (int64_t)cg * V, a);
        Vec<T, V>::load(xn + (p + m.PR) * ldx + (int64_t)cg * V, b);
        Vec<T, V>::load(xn + (p + 2 * m.PR) * ldx + (int64_t)cg * V, c);
        Vec<T, V>::load(xn + (p + 3 * m.PR) * ldx + (int64_t)cg * V, d);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float e0 = a[v] - sh[v], e1 = b[v] - sh[v], e2 = c[v] - sh[v], e3 = d[v] - sh[v];
          S[v] += (e0 + e1) + (e2 + e3);
          Q[v] += (e0 * e0 + e1 * e1) + (e2 * e2 + e3 * e3);
        }
      }
      for (; p < p1; p += m.PR) {
        float a[V];
        Vec<T, V>::load(xn + p * ldx + (int64_t)cg * V, a);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float e = a[v] - sh[v];
          S[v] += e;
          Q[v] += e * e;
        }
      }
    }
    __syncthreads();
    if (tid < m.active) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        red[prow * row + cgl * V + v] = S[v];
        red[m.PR * row + prow * row + cgl * V + v] = Q[v];
      }
    }
    __syncthreads();
    for (int j = tid; j < row; j += SN_THREADS) {
      const int ch = cg0 * V + j;
      if (ch < C) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < m.PR; ++r) {
          s += red[r * row + j];
          q += red[m.PR * row + r * row + j];
        }
        float* o = part + (((int64_t)n * Tn + t) * C + ch) * 2;
        o[0] = s;
        o[1] = q;
      }
    }
  }
}

// partial moments -> mean_in, var_in (unbiased)
__device__ __forceinline__ void sn_finalize_body(const float* __restrict__ part, int T,
                                                          const float* __restrict__ shift, int shift_per_image, int N,
                                                          int64_t HW, int C, float* __restrict__ ctx) {
  __shared__ double red[2][8][64];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  double S = 0, Q = 0;
  if (c < C) {
    // eight, then four independent loads in flight per thread (the loop is latency-bound: 16 workgroups read the
    // whole buffer); the order of the additions is fixed by T alone
    const float2* p = reinterpret_cast<const float2*>(part) + ((int64_t)n * T) * C + c;
    int t = tl;
    for (; t + 56 < T; t += 64) {
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(t + 8 * u) * C];
      S += (((double)v[0].x + (double)v[1].x) + ((double)v[2].x + (double)v[3].x)) +
           (((double)v[4].x + (double)v[5].x) + ((double)v[6].x + (double)v[7].x));
      Q += (((double)v[0].y + (double)v[1].y) + ((double)v[2].y + (double)v[3].y)) +
           (((double)v[4].y + (double)v[5].y) + ((double)v[6].y + (double)v[7].y));
    }
    for (; t + 24 < T; t += 32) {
      const float2 v0 = p[(int64_t)t * C], v1 = p[(int64_t)(t + 8) * C], v2 = p[(int64_t)(t + 16) * C],
                   v3 = p[(int64_t)(t + 24) * C];
      S += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      Q += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; t < T; t += 8) {
      const float2 v = p[(int64_t)t * C];
      S += (double)v.x;
      Q += (double)v.y;
    }
  }
  red[0][tl][cl] = S;
  red[1][tl][cl] = Q;
  __syncthreads();
  if (tl == 0 && c < C) {
    S = 0; Q = 0;
    for (int i = 0; i < 8; ++i) { S += red[0][i][cl]; Q += red[1][i][cl]; }
    const CtxOff o = ctx_off(N, C);
    const double sh = (double)shift[shift_per_image ? (int64_t)n * C + c : c];
    const double m = S / (double)HW;
    const double var = (Q - S * m) / (double)(HW - 1);
    ctx[o.mean_in + (int64_t)n * C + c] = (float)(sh + m);
    ctx[o.var_in + (int64_t)n * C + c] = (float)var;
  }
}

// LN moments per image, local BN sums per channel
__device__ __forceinline__ void sn_mix_a_body(float* __restrict__ ctx, float* __restrict__ pack, int N, int C) {
  __shared__ double scratch[16];
  const CtxOff o = ctx_off(N, C);
  if ((int)blockIdx.x < N) {
    const int n = blockIdx.x;
    double s1 = 0, s2 = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
      double mi = ctx[o.mean_in + (int64_t)n * C + c], vi = ctx[o.var_in + (int64_t)n * C + c];
      s1 += mi;
      s2 += vi + mi * mi;
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
      double ml = s1 / C;
      ctx[o.ln + 2 * n] = (float)ml;
      ctx[o.ln + 2 * n + 1] = (float)(s2 / C - ml * ml);
    }
  } else {
    const int c = (blockIdx.x - N) * 256 + threadIdx.x;
    if (c < C) {
      double s1 = 0, s2 = 0;
      for (int n = 0; n < N; ++n) {
        double mi = ctx[o.mean_in + (int64_t)n * C + c], vi = ctx[o.var_in + (int64_t)n * C + c];
        s1 += mi;
        s2 += vi + mi * mi;
      }
      pack[c] = (float)s1;
      pack[C + c] = (float)s2;
    }
    if (blockIdx.x == (unsigned)N && threadIdx.x == 0) pack[2 * C] = (float)N;
  }
}

__device__ __forceinline__ void softmax_k(const float* w, int K, float* out) {
  float mx = w[0];
  for (int i = 1; i < K; ++i) mx = fmaxf(mx, w[i]);
  float e[3], s = 0.f;
  for (int i = 0; i < K; ++i) { e[i] = expf(w[i] - mx); s += e[i]; }
  for (int i = 0; i < 3; ++i) out[i] = i < K ? e[i] / s : 0.f;
}

__device__ __forceinline__ void sn_mix_b_body(float* __restrict__ ctx, const float* __restrict__ pack,
                                                       const float* __restrict__ gamma, const float* __restrict__ mw,
                                                       const float* __restrict__ vw, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, int N, int C, float eps, float momentum,
                                                       int training, int using_movavg, int using_bn) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)N * C) return;
  const int n = (int)(i / C), c = (int)(i % C);
  const CtxOff o = ctx_off(N, C);
  const int K = using_bn ? 3 : 2;
  float a[3], b[3];
  softmax_k(mw, K, a);
  softmax_k(vw, K, b);
  const float mean_in = ctx[o.mean_in + i], var_in = ctx[o.var_in + i];
  const float mean_ln = ctx[o.ln + 2 * n], var_ln = ctx[o.ln + 2 * n + 1];
  float mean = a[0] * mean_in + a[1] * mean_ln;
  float var = b[0] * var_in + b[1] * var_ln;
  if (using_bn) {
    float mean_bn, var_bn;
    if (training) {
      const float Ng = pack[2 * C];
      mean_bn = pack[c] / Ng;
      var_bn = pack[C + c] / Ng - mean_bn * mean_bn;
      if (n == 0) {
        if (using_movavg) {
          rmean[c] = rmean[c] * momentum + (1.f - momentum) * mean_bn;
          rvar[c] = rvar[c] * momentum + (1.f - momentum) * var_bn;
        } else {
          rmean[c] += mean_bn;
          rvar[c] += mean_bn * mean_bn + var_bn;
        }
      }
    } else {
      mean_bn = rmean[c];
      var_bn = rvar[c];
    }
    if (n == 0) {
      ctx[o.bn + c] = mean_bn;
      ctx[o.bn + C + c] = var_bn;
    }
    mean += a[2] * mean_bn;
    var += b[2] * var_bn;
  }
  const float rstd = 1.0f / sqrtf(var + eps);
  ctx[o.mean + i] = mean;
  ctx[o.rstd + i] = rstd;
  ctx[o.scale + i] = gamma[c] * rstd;
}

// ------------------------------------------------------------------------------------------- pass 2: normalise
template <typename T, int V, bool RELU, bool RES>
__device__ __forceinline__ void sn_apply_body(const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                              int64_t ldy, const T* __restrict__ res, int64_t ldr,
                                                              const float* __restrict__ ctx,
                                                              const float* __restrict__ beta, int N, int64_t HW, int C,
                                                              int PT, SnMap m, const float* __restrict__ post = nullptr) {
  // post: optional non-negative multiplier per (image, channel) on the OUTPUT (a feature-dropout mask after the
  // activation): act(z) * s = act(z * s) for s >= 0, so it is folded into the coefficients (no residual then)
  // PT > 0: the workgroup streams the contiguous pixel range [t PT, (t+1) PT).  PT == 0: the workgroups of an image
  // interleave pieces of 4 PR pixels (piece t, t + T, ...), so the resident workgroups sweep one compact window of
  // each tensor instead of thousands of far-apart streams (DRAM page locality).
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int64_t PS = 4 * (int64_t)m.PR;
  const int64_t p0 = PT > 0 ? (int64_t)t * PT : (int64_t)t * PS;
  const int64_t p1 = PT > 0 ? (p0 + PT < HW ? p0 + PT : HW) : HW;
  const int64_t pstep = PT > 0 ? PS : (int64_t)gridDim.x * PS;
  const CtxOff o = ctx_off(N, C);
  const T* xn = x + (int64_t)n * HW * ldx;
  T* yn = y + (int64_t)n * HW * ldy;
  const T* rn = RES ? res + (int64_t)n * HW * ldr : nullptr;
  const int cgl = tid % m.cgb, prow = tid / m.cgb;
  if (tid >= m.active) return;
  for (int cg0 = 0; cg0 < m.CG; cg0 += m.cgb) {
    const int cg = cg0 + cgl;
    if (cg >= m.CG) continue;
    float mu[V], sc[V], be[V];
    {
      const int64_t i0 = (int64_t)n * C + cg * V;
      load_coef<V>(ctx + o.mean + i0, mu);
      load_coef<V>(ctx + o.scale + i0, sc);
      load_coef<V>(beta + cg * V, be);
      if (post) {
        float pm[V];
        load_coef<V>(post + i0, pm);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          sc[v] *= pm[v];
          be[v] *= pm[v];
        }
      }
    }
    const int64_t co = (int64_t)cg * V;
    int64_t p = p0 + prow;
    for (; p + 3 * m.PR < p1; p += pstep) {
      float a[4][V];
#pragma unroll
      for (int u = 0; u < 4; ++u) VecNT<T, V>::load(xn + (p + u * m.PR) * ldx + co, a[u]);
      float r[4][V];
      if (RES) {
#pragma unroll
        for (int u = 0; u < 4; ++u) VecNT<T, V>::load(rn + (p + u * m.PR) * ldr + co, r[u]);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float z = fmaf(a[u][v] - mu[v], sc[v], be[v]);
          if (RES) z += r[u][v];
          a[u][v] = RELU ? fmaxf(z, 0.f) : z;
        }
        VecNT<T, V>::store(yn + (p + u * m.PR) * ldy + co, a[u]);
      }
    }
    // (the ragged end: at most one piece, and only in the workgroup whose walk reaches it)
    for (; p < p1; p += m.PR) {
      float a[V], r[V];
      VecNT<T, V>::load(xn + p * ldx + co, a);
      if (RES) VecNT<T, V>::load(rn + p * ldr + co, r);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        float z = fmaf(a[v] - mu[v], sc[v], be[v]);
        if (RES) z += r[v];
        a[v] = RELU ? fmaxf(z, 0.f) : z;
      }
      VecNT<T, V>::store(yn + p * ldy + co, a);
    }
  }
}

// ------------------------------------------------------------------------------------------- backward pass 1
template <typename T, int V, bool RELU, bool YMASK>
__global__ __launch_bounds__(SN_THREADS) void sn_bwd_reduce_kernel(const T* __restrict__ dy, int64_t lddy,
                                                                   const T* __restrict__ x, int64_t ldx,
                                                                   const T* __restrict__ ym, int64_t ldy,
                                                                   const float* __restrict__ ctx,
                                                                   const float* __restrict__ beta, int N, int64_t HW,
                                                                   int C, int PT, int Tn, SnMap m,
                                                                   float* __restrict__ part) {
  extern __shared__ __attribute__((aligned(16))) float red[];
  // PT == 0: interleaved pieces of U PR pixels (see sn_stats_kernel)
  constexpr int U = SN_BWD_ROWS<YMASK>;
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int64_t PS = U * (int64_t)m.PR;
  const int64_t p0 = PT > 0 ? (int64_t)t * PT : (int64_t)t * PS;
  const int64_t p1 = PT > 0 ? (p0 + PT < HW ? p0 + PT : HW) : HW;
  const int64_t pstep = PT > 0 ? PS : (int64_t)gridDim.x * PS;
  const CtxOff o = ctx_off(N, C);
  const T* xn = x + (int64_t)n * HW * ldx;
  const T* dn = dy + (int64_t)n * HW * lddy;
  const T* yn = YMASK ? ym + (int64_t)n * HW * ldy : nullptr;
  const int cgl = tid % m.cgb, prow = tid / m.cgb;
  const int row = m.cgb * V;
  for (int cg0 = 0; cg0 < m.CG; cg0 += m.cgb) {
    const int cg = cg0 + cgl;
    const bool valid = tid < m.active && cg < m.CG;
    float P[V], Q[V];
#pragma unroll
    for (int v = 0; v < V; ++v) P[v] = Q[v] = 0.f;
    if (valid) {
      float mu[V], sc[V], rs[V], be[V];
      {
        const int64_t i0 = (int64_t)n * C + cg * V;
        load_coef<V>(ctx + o.mean + i0, mu);
        load_coef<V>(ctx + o.scale + i0, sc);
        load_coef<V>(ctx + o.rstd + i0, rs);
        load_coef<V>(beta + cg * V, be);
      }
      const int64_t co = (int64_t)cg * V;
      int64_t p = p0 + prow;
      for (; p + (U - 1) * m.PR < p1; p += pstep) {
        float a[U][V], g[U][V], yv[YMASK ? U : 1][V];
#pragma unroll
        for (int u = 0; u < U; ++u) VecNT<T, V>::load(xn + (p + u * m.PR) * ldx + co, a[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) VecNT<T, V>::load(dn + (p + u * m.PR) * lddy + co, g[u]);
        if (YMASK) {
#pragma unroll
          for (int u = 0; u < U; ++u) VecNT<T, V>::load(yn + (p + u * m.PR) * ldy + co, yv[YMASK ? u : 0]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const float e = a[u][v] - mu[v];
            float gg = g[u][v];
            if (RELU) {
              const bool k = YMASK ? (yv[YMASK ? u : 0][v] > 0.f) : (fmaf(e, sc[v], be[v]) > 0.f);
              gg = k ? gg : 0.f;
            }
            P[v] += gg;
            Q[v] += gg * (e * rs[v]);
          }
        }
      }
      for (; p < p1; p += m.PR) {
        float a0[V], g0[V], y0[V];
        VecNT<T, V>::load(xn + p * ldx + co, a0);
        VecNT<T, V>::load(dn + p * lddy + co, g0);
        if (YMASK) VecNT<T, V>::load(yn + p * ldy + co, y0);
#pragma unroll
        for (int v = 0; v < V; ++v) {
          float e0 = a0[v] - mu[v];
          if (RELU) {
            bool k0 = YMASK ? (y0[v] > 0.f) : (fmaf(e0, sc[v], be[v]) > 0.f);
            g0[v] = k0 ? g0[v] : 0.f;
          }
          P[v] += g0[v];
          Q[v] += g0[v] * (e0 * rs[v]);
        }
      }
    }
    __syncthreads();
    if (tid < m.active) {
#pragma unroll
      for (int v = 0; v < V; ++v) {
        red[prow * row + cgl * V + v] = P[v];
        red[m.PR * row + prow * row + cgl * V + v] = Q[v];
      }
    }
    __syncthreads();
    for (int j = tid; j < row; j += SN_THREADS) {
      const int ch = cg0 * V + j;
      if (ch < C) {
        float s = 0.f, q = 0.f;
        for (int r = 0; r < m.PR; ++r) {
          s += red[r * row + j];
          q += red[m.PR * row + r * row + j];
        }
        float* op = part + (((int64_t)n * Tn + t) * C + ch) * 2;
        op[0] = s;
        op[1] = q;
      }
    }
  }
}

// partials -> P, Q, dm, dv per (n,c)
__device__ __forceinline__ void sn_bwd_mix_a1_body(const float* __restrict__ part, int T,
                                                            const float* __restrict__ ctx,
                                                            const float* __restrict__ gamma, float* __restrict__ ctxb,
                                                            int N, int C) {
  __shared__ double red[2][8][64];
  const int cl = threadIdx.x & 63, tl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl, n = blockIdx.y;
  double P = 0, Q = 0;
  if (c < C) {
    const float2* p = reinterpret_cast<const float2*>(part) + ((int64_t)n * T) * C + c;
    int t = tl;
    for (; t + 56 < T; t += 64) {       // eight, then four independent loads in flight per thread
      float2 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = p[(int64_t)(t + 8 * u) * C];
      P += (((double)v[0].x + (double)v[1].x) + ((double)v[2].x + (double)v[3].x)) +
           (((double)v[4].x + (double)v[5].x) + ((double)v[6].x + (double)v[7].x));
      Q += (((double)v[0].y + (double)v[1].y) + ((double)v[2].y + (double)v[3].y)) +
           (((double)v[4].y + (double)v[5].y) + ((double)v[6].y + (double)v[7].y));
    }
    for (; t + 24 < T; t += 32) {
      const float2 v0 = p[(int64_t)t * C], v1 = p[(int64_t)(t + 8) * C], v2 = p[(int64_t)(t + 16) * C],
                   v3 = p[(int64_t)(t + 24) * C];
      P += ((double)v0.x + (double)v1.x) + ((double)v2.x + (double)v3.x);
      Q += ((double)v0.y + (double)v1.y) + ((double)v2.y + (double)v3.y);
    }
    for (; t < T; t += 8) {
      const float2 v = p[(int64_t)t * C];
      P += (double)v.x;
      Q += (double)v.y;
    }
  }
  red[0][tl][cl] = P;
  red[1][tl][cl] = Q;
  __syncthreads();
  if (tl == 0 && c < C) {
    P = 0; Q = 0;
    for (int i = 0; i < 8; ++i) { P += red[0][i][cl]; Q += red[1][i][cl]; }
    const CtxOff o = ctx_off(N, C);
    const CtxbOff ob = ctxb_off(N, C);
    const int64_t i = (int64_t)n * C + c;
    const double r = ctx[o.rstd + i], g = gamma[c];
    ctxb[ob.P + i] = (float)P;
    ctxb[ob.Q + i] = (float)Q;
    ctxb[ob.dm + i] = (float)(-r * g * P);
    ctxb[ob.dv + i] = (float)(-0.5 * r * r * g * Q);
  }
}

__device__ __forceinline__ void sn_bwd_mix_a2_body(float* __restrict__ ctxb, float* __restrict__ packb,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                            int C, int accumulate) {
  __shared__ double scratch[16];
  const CtxbOff ob = ctxb_off(N, C);
  if ((int)blockIdx.x < N) {
    const int n = blockIdx.x;
    double s1 = 0, s2 = 0;
    for (int c = threadIdx.x; c < C; c += 256) {
      s1 += (double)ctxb[ob.dm + (int64_t)n * C + c];
      s2 += (double)ctxb[ob.dv + (int64_t)n * C + c];
    }
    s1 = block_sum(s1, scratch);
    s2 = block_sum(s2, scratch);
    if (threadIdx.x == 0) {
      ctxb[ob.lnb + 2 * n] = (float)s1;
      ctxb[ob.lnb + 2 * n + 1] = (float)s2;
    }
  } else {
    const int c = (blockIdx.x - N) * 256 + threadIdx.x;
    if (c < C) {
      double s1 = 0, s2 = 0, sp = 0, sq = 0;
      for (int n = 0; n < N; ++n) {
        const int64_t i = (int64_t)n * C + c;
        s1 += (double)ctxb[ob.dm + i];
        s2 += (double)ctxb[ob.dv + i];
        sp += (double)ctxb[ob.P + i];
        sq += (double)ctxb[ob.Q + i];
      }
      packb[c] = (float)s1;
      packb[C + c] = (float)s2;
      dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sq;
      dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)sp;
    }
  }
}

__device__ __forceinline__ void sn_bwd_mix_b_body(const float* __restrict__ ctx, float* __restrict__ ctxb,
                                                            const float* __restrict__ packb,
                                                            const float* __restrict__ mw, const float* __restrict__ vw,
                                                            float* __restrict__ dmw, float* __restrict__ dvw,
                                                            float* __restrict__ dxsum, int N, int64_t HW, int C,
                                                            float Ng, int training, int using_bn, int accumulate) {
  const CtxOff o = ctx_off(N, C);
  const CtxbOff ob = ctxb_off(N, C);
  const int K = using_bn ? 3 : 2;
  float a[3], b[3];
  softmax_k(mw, K, a);
  softmax_k(vw, K, b);
  const bool bn_batch = using_bn && training;
  double da[3] = {0, 0, 0}, db[3] = {0, 0, 0};
  const int NC = N * C;      // the host checks N * C < 2^31
  constexpr int U = 4;       // entries per thread per round: all their loads are issued before the first store
  for (int i0 = threadIdx.x; i0 < NC; i0 += 1024 * U) {
    float dm[U], dv[U], mean_in[U], var_in[U], mean_ln[U], var_ln[U], lnb0[U], lnb1[U], mean_bn[U], var_bn[U], pk0[U],
        pk1[U], sc[U], Pv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 1024;
      const bool ok = i < NC;
      const int ii = ok ? i : 0;
      const int n = ii / C, c = ii - n * C;
      dm[u] = ok ? ctxb[ob.dm + ii] : 0.f;
      dv[u] = ok ? ctxb[ob.dv + ii] : 0.f;
      mean_in[u] = ctx[o.mean_in + ii];
      var_in[u] = ctx[o.var_in + ii];
      mean_ln[u] = ctx[o.ln + 2 * n];
      var_ln[u] = ctx[o.ln + 2 * n + 1];
      lnb0[u] = ctxb[ob.lnb + 2 * n];
      lnb1[u] = ctxb[ob.lnb + 2 * n + 1];
      mean_bn[u] = using_bn ? ctx[o.bn + c] : 0.f;
      var_bn[u] = using_bn ? ctx[o.bn + C + c] : 0.f;
      pk0[u] = bn_batch ? packb[c] : 0.f;
      pk1[u] = bn_batch ? packb[C + c] : 0.f;
      sc[u] = ctx[o.scale + ii];
      Pv[u] = ctxb[ob.P + ii];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = i0 + u * 1024;
      if (i >= NC) break;
      da[0] += (double)dm[u] * mean_in[u];
      da[1] += (double)dm[u] * mean_ln[u];
      db[0] += (double)dv[u] * var_in[u];
      db[1] += (double)dv[u] * var_ln[u];
      const float dvar_ln = b[1] * lnb1[u];
      const float dmean_ln = a[1] * lnb0[u] - 2.f * mean_ln[u] * dvar_ln;
      float dtemp = dvar_ln / C;
      float dmean_in = a[0] * dm[u] + dmean_ln / C;
      if (using_bn) {
        da[2] += (double)dm[u] * mean_bn[u];
        db[2] += (double)dv[u] * var_bn[u];
        if (bn_batch) {
          const float dvar_bn = b[2] * pk1[u];
          const float dmean_bn = a[2] * pk0[u] - 2.f * mean_bn[u] * dvar_bn;
          dtemp += dvar_bn / Ng;
          dmean_in += dmean_bn / Ng;
        }
      }
      const float dvar_in = b[0] * dv[u] + dtemp;
      dmean_in += 2.f * mean_in[u] * dtemp;
      const float B = 2.f * dvar_in / (float)(HW - 1);
      ctxb[ob.B + i] = B;
      ctxb[ob.K + i] = dmean_in / (float)HW - B * mean_in[u];
      // sum over the plane of dx = A*dy' + B*x + K is A*P + dmean_in; parked in the (now consumed) dm slot
      ctxb[ob.dm + i] = sc[u] * Pv[u] + dmean_in;
    }
  }
  {   // the six sums in one round trip through LDS
    __shared__ double red6[6][16];
    const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      da[k] = wave_sum(da[k]);
      db[k] = wave_sum(db[k]);
    }
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        red6[k][wid] = da[k];
        red6[3 + k][wid] = db[k];
      }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      double sa = 0, sb = 0;
      for (int i = 0; i < nw; ++i) {
        sa += red6[k][i];
        sb += red6[3 + k][i];
      }
      da[k] = sa;
      db[k] = sb;
    }
  }
  if (dxsum) {
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 1024) {
      double s = 0;
      for (int n = 0; n < N; ++n) s += (double)ctxb[ob.dm + (int64_t)n * C + c];
      dxsum[c] = ((accumulate & 2) ? dxsum[c] : 0.f) + (float)s;
    }
  }
  if (threadIdx.x == 0) {
    double sa = 0, sb = 0;
    for (int k = 0; k < K; ++k) { sa += a[k] * da[k]; sb += b[k] * db[k]; }
    for (int k = 0; k < K; ++k) {
      dmw[k] = ((accumulate & 1) ? dmw[k] : 0.f) + (float)(a[k] * (da[k] - sa));
      dvw[k] = ((accumulate & 1) ? dvw[k] : 0.f) + (float)(b[k] * (db[k] - sb));
    }
  }
}

// ------------------------------------------------------------------------------------------- backward pass 2
// DRES: also store the masked incoming gradient (the gradient of a residual input added before the ReLU)
template <typename T, int V, bool RELU, bool YMASK, bool DRES = false>
__device__ __forceinline__ void sn_bwd_apply_body(const T* __restrict__ dy, int64_t lddy,
                                                                  const T* __restrict__ x, int64_t ldx,
                                                                  const T* __restrict__ ym, int64_t ldy,
                                                                  T* __restrict__ dx, int64_t lddx,
                                                                  const float* __restrict__ ctx,
                                                                  const float* __restrict__ ctxb,
                                                                  const float* __restrict__ beta, int N, int64_t HW,
                                                                  int C, int PT, SnMap m,
                                                                  T* __restrict__ dres = nullptr, int64_t lddr = 0,
                                                                  const float* __restrict__ post = nullptr) {
  // post: the forward's per-(image, channel) output multiplier; the incoming gradient is scaled by it after the mask
  // PT == 0: interleaved pieces of U PR pixels (see sn_apply_kernel)
  constexpr int U = SN_BWD_ROWS<YMASK>;
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x;
  const int64_t PS = U * (int64_t)m.PR;
  const int64_t p0 = PT > 0 ? (int64_t)t * PT : (int64_t)t * PS;
  const int64_t p1 = PT > 0 ? (p0 + PT < HW ? p0 + PT : HW) : HW;
  const int64_t pstep = PT > 0 ? PS : (int64_t)gridDim.x * PS;
  const CtxOff o = ctx_off(N, C);
  const CtxbOff ob = ctxb_off(N, C);
  const T* xn = x + (int64_t)n * HW * ldx;
  const T* dn = dy + (int64_t)n * HW * lddy;
  const T* yn = YMASK ? ym + (int64_t)n * HW * ldy : nullptr;
  T* on = dx + (int64_t)n * HW * lddx;
  T* rn = DRES ? dres + (int64_t)n * HW * lddr : nullptr;
  const int cgl = tid % m.cgb, prow = tid / m.cgb;
  if (tid >= m.active) return;
  for (int cg0 = 0; cg0 < m.CG; cg0 += m.cgb) {
    const int cg = cg0 + cgl;
    if (cg >= m.CG) continue;
    float mu[V], sc[V], be[V], Bc[V], Kc[V], sg[V];
    {
      const int64_t i0 = (int64_t)n * C + cg * V;
      load_coef<V>(ctx + o.mean + i0, mu);
      load_coef<V>(ctx + o.scale + i0, sc);
      load_coef<V>(beta + cg * V, be);
      load_coef<V>(ctxb + ob.B + i0, Bc);
      load_coef<V>(ctxb + ob.K + i0, Kc);
#pragma unroll
      for (int v = 0; v < V; ++v) sg[v] = sc[v];
      if (post) {
        float pm[V];
        load_coef<V>(post + i0, pm);
#pragma unroll
        for (int v = 0; v < V; ++v) sg[v] = sc[v] * pm[v];
      }
    }
    const int64_t co = (int64_t)cg * V;
    int64_t p = p0 + prow;
    for (; p + (U - 1) * m.PR < p1; p += pstep) {
      float a[U][V], g[U][V], yv[YMASK ? U : 1][V];
#pragma unroll
      for (int u = 0; u < U; ++u) VecNT<T, V>::load(xn + (p + u * m.PR) * ldx + co, a[u]);
#pragma unroll
      for (int u = 0; u < U; ++u) VecNT<T, V>::load(dn + (p + u * m.PR) * lddy + co, g[u]);
      if (YMASK) {
#pragma unroll
        for (int u = 0; u < U; ++u) VecNT<T, V>::load(yn + (p + u * m.PR) * ldy + co, yv[YMASK ? u : 0]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (RELU) {
#pragma unroll
          for (int v = 0; v < V; ++v) {
            const bool k = YMASK ? (yv[YMASK ? u : 0][v] > 0.f) : (fmaf(a[u][v] - mu[v], sc[v], be[v]) > 0.f);
            g[u][v] = k ? g[u][v] : 0.f;
          }
        }
        if (DRES) VecNT<T, V>::store(rn + (p + u * m.PR) * lddr + co, g[u]);
#pragma unroll
        for (int v = 0; v < V; ++v) g[u][v] = fmaf(sg[v], g[u][v], fmaf(Bc[v], a[u][v], Kc[v]));
        VecNT<T, V>::store(on + (p + u * m.PR) * lddx + co, g[u]);
      }
    }
    for (; p < p1; p += m.PR) {
      float a0[V], g0[V], y0[V];
      VecNT<T, V>::load(xn + p * ldx + co, a0);
      VecNT<T, V>::load(dn + p * lddy + co, g0);
      if (YMASK) VecNT<T, V>::load(yn + p * ldy + co, y0);
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (RELU) {
          bool k0 = YMASK ? (y0[v] > 0.f) : (fmaf(a0[v] - mu[v], sc[v], be[v]) > 0.f);
          g0[v] = k0 ? g0[v] : 0.f;
        }
      }
      if (DRES) VecNT<T, V>::store(rn + p * lddr + co, g0);
#pragma unroll
      for (int v = 0; v < V; ++v) g0[v] = fmaf(sg[v], g0[v], fmaf(Bc[v], a0[v], Kc[v]));
      VecNT<T, V>::store(on + p * lddx + co, g0);
    }
  }
}


// ------------------------------------------------------------------------------------------- kernels
// Every pass exists as a plain kernel and as a BATCHED kernel: up to SNWS_SN_BATCH_MAX independent SwitchNorm calls of
// the same (N, HW, C, dtype) - the five branch blocks of a UNet3+ decoder scale - run as ONE launch per pass, the call
// picked by blockIdx.z.  The per-(image, channel) passes are latency-bound (a few microseconds whatever their size), so
// five calls cost what one does; the streaming passes save their ramps.
__global__ __launch_bounds__(512) void sn_finalize_kernel(const float* __restrict__ part, int T,
                                                          const float* __restrict__ shift, int shift_per_image, int N,
                                                          int64_t HW, int C, float* __restrict__ ctx) {
  sn_finalize_body(part, T, shift, shift_per_image, N, HW, C, ctx);
}
__global__ __launch_bounds__(256) void sn_mix_a_kernel(float* __restrict__ ctx, float* __restrict__ pack, int N, int C) {
  sn_mix_a_body(ctx, pack, N, C);
}
__global__ __launch_bounds__(256) void sn_mix_b_kernel(float* __restrict__ ctx, const float* __restrict__ pack,
                                                       const float* __restrict__ gamma, const float* __restrict__ mw,
                                                       const float* __restrict__ vw, float* __restrict__ rmean,
                                                       float* __restrict__ rvar, int N, int C, float eps, float momentum,
                                                       int training, int using_movavg, int using_bn) {
  sn_mix_b_body(ctx, pack, gamma, mw, vw, rmean, rvar, N, C, eps, momentum, training, using_movavg, using_bn);
}
template <typename T, int V, bool RELU, bool RES>
__global__ __launch_bounds__(SN_THREADS) void sn_apply_kernel(const T* __restrict__ x, int64_t ldx, T* __restrict__ y,
                                                              int64_t ldy, const T* __restrict__ res, int64_t ldr,
                                                              const float* __restrict__ ctx,
                                                              const float* __restrict__ beta, int N, int64_t HW, int C,
                                                              int PT, SnMap m, const float* __restrict__ post = nullptr) {
  sn_apply_body<T, V, RELU, RES>(x, ldx, y, ldy, res, ldr, ctx, beta, N, HW, C, PT, m, post);
}
__global__ __launch_bounds__(512) void sn_bwd_mix_a1_kernel(const float* __restrict__ part, int T,
                                                            const float* __restrict__ ctx,
                                                            const float* __restrict__ gamma, float* __restrict__ ctxb,
                                                            int N, int C) {
  sn_bwd_mix_a1_body(part, T, ctx, gamma, ctxb, N, C);
}
__global__ __launch_bounds__(256) void sn_bwd_mix_a2_kernel(float* __restrict__ ctxb, float* __restrict__ packb,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int N,
                                                            int C, int accumulate) {
  sn_bwd_mix_a2_body(ctxb, packb, dgamma, dbeta, N, C, accumulate);
}
__global__ __launch_bounds__(1024) void sn_bwd_mix_b_kernel(const float* __restrict__ ctx, float* __restrict__ ctxb,
                                                            const float* __restrict__ packb,
                                                            const float* __restrict__ mw, const float* __restrict__ vw,
                                                            float* __restrict__ dmw, float* __restrict__ dvw,
                                                            float* __restrict__ dxsum, int N, int64_t HW, int C,
                                                            float Ng, int training, int using_bn, int accumulate) {
  sn_bwd_mix_b_body(ctx, ctxb, packb, mw, vw, dmw, dvw, dxsum, N, HW, C, Ng, training, using_bn, accumulate);
}
template <typename T, int V, bool RELU, bool YMASK, bool DRES = false>
__global__ __launch_bounds__(SN_THREADS) void sn_bwd_apply_kernel(const T* __restrict__ dy, int64_t lddy,
                                                                  const T* __restrict__ x, int64_t ldx,
                                                                  const T* __restrict__ ym, int64_t ldy,
                                                                  T* __restrict__ dx, int64_t lddx,
                                                                  const float* __restrict__ ctx,
                                                                  const float* __restrict__ ctxb,
                                                                  const float* __restrict__ beta, int N, int64_t HW,
                                                                  int C, int PT, SnMap m,
                                                                  T* __restrict__ dres = nullptr, int64_t lddr = 0,
                                                                  const float* __restrict__ post = nullptr) {
  sn_bwd_apply_body<T, V, RELU, YMASK, DRES>(dy, lddy, x, ldx, ym, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, PT, m, dres,
                                             lddr, post);
}

struct SnFwdBatch { snws_sn_fwd_item item[SNWS_SN_BATCH_MAX]; };
struct SnBwdBatch { snws_sn_bwd_item item[SNWS_SN_BATCH_MAX]; };

__global__ __launch_bounds__(512) void sn_finalize_batched_kernel(const SnFwdBatch b, int N, int64_t HW, int C) {
  const snws_sn_fwd_item& d = b.item[blockIdx.z];
  sn_finalize_body(d.part, d.T, d.shift, d.shift_per_image, N, HW, C, d.ctx);
}
__global__ __launch_bounds__(256) void sn_mix_a_batched_kernel(const SnFwdBatch b, int N, int C) {
  const snws_sn_fwd_item& d = b.item[blockIdx.z];
  sn_mix_a_body(d.ctx, d.pack, N, C);
}
__global__ __launch_bounds__(256) void sn_mix_b_batched_kernel(const SnFwdBatch b, int N, int C, float eps,
                                                               float momentum, int training, int using_movavg,
                                                               int using_bn) {
  const snws_sn_fwd_item& d = b.item[blockIdx.z];
  sn_mix_b_body(d.ctx, d.pack, d.gamma, d.mean_weight, d.var_weight, d.running_mean, d.running_var, N, C, eps, momentum,
                training, using_movavg, using_bn);
}
template <typename T, int V, bool RELU>
__global__ __launch_bounds__(SN_THREADS) void sn_apply_batched_kernel(const SnFwdBatch b, int N, int64_t HW, int C,
                                                                      int PT, SnMap m) {
  const snws_sn_fwd_item& d = b.item[blockIdx.z];
  sn_apply_body<T, V, RELU, false>((const T*)d.x, d.ldx, (T*)d.y, d.ldy, nullptr, 0, d.ctx, d.beta, N, HW, C, PT, m,
                                   nullptr);
}
__global__ __launch_bounds__(512) void sn_bwd_mix_a1_batched_kernel(const SnBwdBatch b, int N, int C) {
  const snws_sn_bwd_item& d = b.item[blockIdx.z];
  sn_bwd_mix_a1_body(d.part, d.T, d.ctx, d.gamma, d.ctxb, N, C);
}
__global__ __launch_bounds__(256) void sn_bwd_mix_a2_batched_kernel(const SnBwdBatch b, int N, int C) {
  const snws_sn_bwd_item& d = b.item[blockIdx.z];
  sn_bwd_mix_a2_body(d.ctxb, d.packb, d.dgamma, d.dbeta, N, C, d.accumulate_a);
}
__global__ __launch_bounds__(1024) void sn_bwd_mix_b_batched_kernel(const SnBwdBatch b, int N, int64_t HW, int C,
                                                                    float Ng, int training, int using_bn) {
  const snws_sn_bwd_item& d = b.item[blockIdx.z];
  sn_bwd_mix_b_body(d.ctx, d.ctxb, d.packb, d.mean_weight, d.var_weight, d.dmean_weight, d.dvar_weight, d.dx_sum, N, HW,
                    C, Ng, training, using_bn, d.accumulate_b);
}
template <typename T, int V, bool RELU>
__global__ __launch_bounds__(SN_THREADS) void sn_bwd_apply_batched_kernel(const SnBwdBatch b, int N, int64_t HW, int C,
                                                                          int PT, SnMap m) {
  const snws_sn_bwd_item& d = b.item[blockIdx.z];
  sn_bwd_apply_body<T, V, RELU, false, false>((const T*)d.dy, d.lddy, (const T*)d.x, d.ldx, nullptr, 0, (T*)d.dx, d.lddx,
                                              d.ctx, d.ctxb, d.beta, N, HW, C, PT, m, nullptr, 0, nullptr);
}

// ------------------------------------------------------------------------------------------- host side
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <typename T>
constexpr int full_vec() { return 16 / sizeof(T); }

// can every tensor in the call be accessed 16 bytes at a time?
template <typename T>
bool can_vec(int C, std::initializer_list<const void*> ptrs, std::initializer_list<int64_t> lds) {
  const int V = full_vec<T>();
  if (C % V) return false;
  for (const void* p : ptrs)
    if (p && !aligned16(p)) return false;
  for (int64_t l : lds)
    if (l % V) return false;
  return true;
}

#define SN_COMMON_CHECKS(fn)                                                                       \
  SNWS_CHECK_ARG(N >= 1 && C >= 1 && HW >= 1, fn ": N, C, HW must be positive (N=%d C=%d HW=%lld)", N, C, \
                 (long long)HW);                                                                   \
  SNWS_CHECK_ARG(dtype == SNWS_F32 || dtype == SNWS_BF16, fn ": dtype %d is not F32/BF16", dtype);

// SNWS_SN_INTERLEAVE = 0 gives every workgroup of the apply passes one contiguous pixel range instead of interleaved
// pieces (A/B measurements)
inline bool sn_interleave() {
  static const bool on = !(getenv("SNWS_SN_INTERLEAVE") && atoi(getenv("SNWS_SN_INTERLEAVE")) == 0);
  return on;
}
template <typename T, int V>
int run_stats(const void* x, int N, int64_t HW, int C, int64_t ldx, float* part, float* shift, hipStream_t st) {
  const SnMap m = sn_map(C, V);
  const int T_ = sn_tiles(N, HW, C);        // = the partial-sum rows per image the caller allocated
  const int PT = sn_interleave() ? 0 : (int)cdiv64(HW, T_);
  const size_t lds = (size_t)2 * m.PR * m.cgb * V * sizeof(float);
  sn_stats_kernel<T, V><<<dim3(T_, N), SN_THREADS, lds, st>>>((const T*)x, HW, C, ldx, PT, T_, m, part, shift);
  SNWS_LAUNCH_CHECK("snws_sn_stats");
  return SNWS_OK;
}

// grid of an apply pass whose workgroups interleave pieces of `piece` pixels: the workgroup count of the contiguous
// walk (64 KB of a tensor per workgroup, at most 4096 workgroups: shorter workgroups spend their time loading
// coefficients), never more than there are pieces
inline int sn_apply_tiles(int N, int64_t HW, int C, int64_t piece) {
  const int64_t t = sn_tiles(N, HW, C), np = cdiv64(HW, piece);
  return (int)(np < t ? np : t);
}

template <typename T, int V>
int run_apply(const void* x, int64_t ldx, void* y, int64_t ldy, const void* res, int64_t ldr, const float* ctx,
              const float* beta, int N, int64_t HW, int C, int relu, hipStream_t st, const float* post = nullptr) {
  const SnMap m = sn_map(C, V);
  const bool il = sn_interleave();
  const int T_ = il ? sn_apply_tiles(N, HW, C, 4 * m.PR) : sn_tiles(N, HW, C);
  const int PT = il ? 0 : (int)cdiv64(HW, T_);
  dim3 grid(T_, N);
#define AP(R, S) sn_apply_kernel<T, V, R, S><<<grid, SN_THREADS, 0, st>>>((const T*)x, ldx, (T*)y, ldy, (const T*)res, ldr, ctx, beta, N, HW, C, PT, m, post)
  if (relu) { if (res) AP(true, true); else AP(true, false); }
  else { if (res) AP(false, true); else AP(false, false); }
#undef AP
  SNWS_LAUNCH_CHECK("snws_sn_apply");
  return SNWS_OK;
}

template <typename T, int V>
int run_bwd_reduce(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* ym, int64_t ldy,
                   const float* ctx, const float* beta, int N, int64_t HW, int C, int relu, float* part,
                   hipStream_t st) {
  const SnMap m = sn_map(C, V);
  const int T_ = sn_tiles(N, HW, C);        // = the partial-sum rows per image the caller allocated
  const int PT = sn_interleave() ? 0 : (int)cdiv64(HW, T_);
  const size_t lds = (size_t)2 * m.PR * m.cgb * V * sizeof(float);
  dim3 grid(T_, N);
#define BR(R, Y) sn_bwd_reduce_kernel<T, V, R, Y><<<grid, SN_THREADS, lds, st>>>((const T*)dy, lddy, (const T*)x, ldx, (const T*)ym, ldy, ctx, beta, N, HW, C, PT, T_, m, part)
  if (relu) { if (ym) BR(true, true); else BR(true, false); }
  else BR(false, false);
#undef BR
  SNWS_LAUNCH_CHECK("snws_sn_bwd_reduce");
  return SNWS_OK;
}

template <typename T, int V>
int run_bwd_apply(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* ym, int64_t ldy, void* dx,
                  int64_t lddx, const float* ctx, const float* ctxb, const float* beta, int N, int64_t HW, int C,
                  int relu, hipStream_t st, void* dres = nullptr, int64_t lddr = 0, const float* post = nullptr) {
  const SnMap m = sn_map(C, V);
  const bool il = sn_interleave();
  const int T_ = il ? sn_apply_tiles(N, HW, C, (ym ? SN_BWD_ROWS<true> : SN_BWD_ROWS<false>) * m.PR) : sn_tiles(N, HW, C);
  const int PT = il ? 0 : (int)cdiv64(HW, T_);
  dim3 grid(T_, N);
#define BA(R, Y) sn_bwd_apply_kernel<T, V, R, Y><<<grid, SN_THREADS, 0, st>>>((const T*)dy, lddy, (const T*)x, ldx, (const T*)ym, ldy, (T*)dx, lddx, ctx, ctxb, beta, N, HW, C, PT, m, nullptr, 0, post)
  if (dres)    // the caller checked relu && ym
    sn_bwd_apply_kernel<T, V, true, true, true><<<grid, SN_THREADS, 0, st>>>((const T*)dy, lddy, (const T*)x, ldx, (const T*)ym, ldy, (T*)dx, lddx, ctx, ctxb, beta, N, HW, C, PT, m, (T*)dres, lddr);
  else if (relu) { if (ym) BA(true, true); else BA(true, false); }
  else BA(false, false);
#undef BA
  SNWS_LAUNCH_CHECK("snws_sn_bwd_apply");
  return SNWS_OK;
}

}  // namespace

extern "C" int64_t snws_sn_ctx_floats(int N, int C) { return ctx_off(N, C).total; }
extern "C" int64_t snws_sn_ctxb_floats(int N, int C) { return ctxb_off(N, C).total; }
extern "C" int snws_sn_stats_tiles(int N, int64_t HW, int C) { return sn_tiles(N, HW, C); }

extern "C" int snws_sn_stats(const void* x, int dtype, int N, int64_t HW, int C, int64_t ldx, float* part,
                             float* shift, void* stream) {
  SN_COMMON_CHECKS("snws_sn_stats");
  SNWS_CHECK_ARG(x && part && shift, "snws_sn_stats: NULL pointer");
  SNWS_CHECK_ARG(ldx >= C, "snws_sn_stats: ldx %lld < C %d", (long long)ldx, C);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {x}, {ldx})) return run_stats<float, 4>(x, N, HW, C, ldx, part, shift, st);
    return run_stats<float, 1>(x, N, HW, C, ldx, part, shift, st);
  }
  if (can_vec<bf16_t>(C, {x}, {ldx})) return run_stats<bf16_t, 8>(x, N, HW, C, ldx, part, shift, st);
  return run_stats<bf16_t, 1>(x, N, HW, C, ldx, part, shift, st);
}

extern "C" int snws_sn_finalize(const float* part, int T, const float* shift, int shift_per_image, int N, int64_t HW,
                                int C, float* ctx, void* stream) {
  SNWS_CHECK_ARG(part && shift && ctx, "snws_sn_finalize: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && C >= 1 && HW >= 1 && T >= 1, "snws_sn_finalize: bad sizes");
  sn_finalize_kernel<<<dim3(cdiv(C, 64), N), 512, 0, (hipStream_t)stream>>>(part, T, shift, shift_per_image, N, HW, C,
                                                                          ctx);
  SNWS_LAUNCH_CHECK("snws_sn_finalize");
  return SNWS_OK;
}

extern "C" int snws_sn_mix_a(float* ctx, float* pack, int N, int C, void* stream) {
  SNWS_CHECK_ARG(ctx && pack, "snws_sn_mix_a: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && C >= 1, "snws_sn_mix_a: bad sizes");
  sn_mix_a_kernel<<<N + cdiv(C, 256), 256, 0, (hipStream_t)stream>>>(ctx, pack, N, C);
  SNWS_LAUNCH_CHECK("snws_sn_mix_a");
  return SNWS_OK;
}

extern "C" int snws_sn_mix_b(float* ctx, const float* pack, const float* gamma, const float* mean_weight,
                             const float* var_weight, float* running_mean, float* running_var, int N, int C, float eps,
                             float momentum, int training, int using_movavg, int using_bn, void* stream) {
  SNWS_CHECK_ARG(ctx && gamma && mean_weight && var_weight, "snws_sn_mix_b: NULL pointer");
  SNWS_CHECK_ARG(!using_bn || (running_mean && running_var), "snws_sn_mix_b: using_bn needs the running buffers");
  SNWS_CHECK_ARG(!(using_bn && training) || pack, "snws_sn_mix_b: training with using_bn needs pack");
  SNWS_CHECK_ARG(N >= 1 && C >= 1, "snws_sn_mix_b: bad sizes");
  sn_mix_b_kernel<<<(unsigned)cdiv64((int64_t)N * C, 256), 256, 0, (hipStream_t)stream>>>(
      ctx, pack, gamma, mean_weight, var_weight, running_mean, running_var, N, C, eps, momentum, training,
      using_movavg, using_bn);
  SNWS_LAUNCH_CHECK("snws_sn_mix_b");
  return SNWS_OK;
}

extern "C" int snws_sn_apply(const void* x, int64_t ldx, void* y, int64_t ldy, const void* residual, int64_t ldr,
                             int dtype, const float* ctx, const float* beta, int N, int64_t HW, int C, int relu,
                             void* stream) {
  SN_COMMON_CHECKS("snws_sn_apply");
  SNWS_CHECK_ARG(x && y && ctx && beta, "snws_sn_apply: NULL pointer");
  SNWS_CHECK_ARG(ldx >= C && ldy >= C && (!residual || ldr >= C), "snws_sn_apply: pixel stride smaller than C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {x, y, residual}, {ldx, ldy, residual ? ldr : 0}))
      return run_apply<float, 4>(x, ldx, y, ldy, residual, ldr, ctx, beta, N, HW, C, relu, st);
    return run_apply<float, 1>(x, ldx, y, ldy, residual, ldr, ctx, beta, N, HW, C, relu, st);
  }
  if (can_vec<bf16_t>(C, {x, y, residual}, {ldx, ldy, residual ? ldr : 0}))
    return run_apply<bf16_t, 8>(x, ldx, y, ldy, residual, ldr, ctx, beta, N, HW, C, relu, st);
  return run_apply<bf16_t, 1>(x, ldx, y, ldy, residual, ldr, ctx, beta, N, HW, C, relu, st);
}

extern "C" int snws_sn_bwd_reduce(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y_for_mask,
                                  int64_t ldy, int dtype, const float* ctx, const float* gamma, const float* beta, int N,
                                  int64_t HW, int C, int relu, float* part, void* stream) {
  SN_COMMON_CHECKS("snws_sn_bwd_reduce");
  (void)gamma;
  SNWS_CHECK_ARG(dy && x && ctx && beta && part, "snws_sn_bwd_reduce: NULL pointer");
  SNWS_CHECK_ARG(lddy >= C && ldx >= C && (!y_for_mask || ldy >= C), "snws_sn_bwd_reduce: pixel stride smaller than C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {dy, x, y_for_mask}, {lddy, ldx, y_for_mask ? ldy : 0}))
      return run_bwd_reduce<float, 4>(dy, lddy, x, ldx, y_for_mask, ldy, ctx, beta, N, HW, C, relu, part, st);
    return run_bwd_reduce<float, 1>(dy, lddy, x, ldx, y_for_mask, ldy, ctx, beta, N, HW, C, relu, part, st);
  }
  if (can_vec<bf16_t>(C, {dy, x, y_for_mask}, {lddy, ldx, y_for_mask ? ldy : 0}))
    return run_bwd_reduce<bf16_t, 8>(dy, lddy, x, ldx, y_for_mask, ldy, ctx, beta, N, HW, C, relu, part, st);
  return run_bwd_reduce<bf16_t, 1>(dy, lddy, x, ldx, y_for_mask, ldy, ctx, beta, N, HW, C, relu, part, st);
}

extern "C" int snws_sn_bwd_mix_a(const float* part, int T, const float* ctx, const float* gamma, float* ctxb,
                                 float* packb, float* dgamma, float* dbeta, int N, int C, int accumulate,
                                 void* stream) {
  SNWS_CHECK_ARG(part && ctx && gamma && ctxb && packb && dgamma && dbeta, "snws_sn_bwd_mix_a: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && C >= 1 && T >= 1, "snws_sn_bwd_mix_a: bad sizes");
  hipStream_t st = (hipStream_t)stream;
  sn_bwd_mix_a1_kernel<<<dim3(cdiv(C, 64), N), 512, 0, st>>>(part, T, ctx, gamma, ctxb, N, C);
  SNWS_LAUNCH_CHECK("snws_sn_bwd_mix_a(1)");
  sn_bwd_mix_a2_kernel<<<N + cdiv(C, 256), 256, 0, st>>>(ctxb, packb, dgamma, dbeta, N, C, accumulate);
  SNWS_LAUNCH_CHECK("snws_sn_bwd_mix_a(2)");
  return SNWS_OK;
}

extern "C" int snws_sn_bwd_mix_b(const float* ctx, float* ctxb, const float* packb, const float* mean_weight,
                                 const float* var_weight, const float* running_mean, const float* running_var,
                                 float* dmean_weight, float* dvar_weight, float* dx_sum, int N, int64_t HW, int C,
                                 float n_global, int training, int using_bn, int accumulate, void* stream) {
  (void)running_mean;
  (void)running_var;
  SNWS_CHECK_ARG(ctx && ctxb && packb && mean_weight && var_weight && dmean_weight && dvar_weight,
                 "snws_sn_bwd_mix_b: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && C >= 1 && HW >= 1 && n_global >= 1.f && (int64_t)N * C < (1ll << 31),
                 "snws_sn_bwd_mix_b: bad sizes");
  sn_bwd_mix_b_kernel<<<1, 1024, 0, (hipStream_t)stream>>>(ctx, ctxb, packb, mean_weight, var_weight, dmean_weight,
                                                         dvar_weight, dx_sum, N, HW, C, n_global, training, using_bn, accumulate);
  SNWS_LAUNCH_CHECK("snws_sn_bwd_mix_b");
  return SNWS_OK;
}

extern "C" int snws_sn_bwd_apply(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y_for_mask,
                                 int64_t ldy, void* dx, int64_t lddx, int dtype, const float* ctx, const float* ctxb,
                                 const float* beta, int N, int64_t HW, int C, int relu, void* stream) {
  SN_COMMON_CHECKS("snws_sn_bwd_apply");
  SNWS_CHECK_ARG(dy && x && dx && ctx && ctxb && beta, "snws_sn_bwd_apply: NULL pointer");
  SNWS_CHECK_ARG(lddy >= C && ldx >= C && lddx >= C && (!y_for_mask || ldy >= C),
                 "snws_sn_bwd_apply: pixel stride smaller than C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {dy, x, y_for_mask, dx}, {lddy, ldx, y_for_mask ? ldy : 0, lddx}))
      return run_bwd_apply<float, 4>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st);
    return run_bwd_apply<float, 1>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st);
  }
  if (can_vec<bf16_t>(C, {dy, x, y_for_mask, dx}, {lddy, ldx, y_for_mask ? ldy : 0, lddx}))
    return run_bwd_apply<bf16_t, 8>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st);
  return run_bwd_apply<bf16_t, 1>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st);
}

extern "C" int snws_sn_bwd_apply_res(const void* dy, int64_t lddy, const void* x, int64_t ldx, const void* y_for_mask,
                                     int64_t ldy, void* dx, int64_t lddx, void* dres, int64_t lddres, int dtype,
                                     const float* ctx, const float* ctxb, const float* beta, int N, int64_t HW, int C,
                                     void* stream) {
  SN_COMMON_CHECKS("snws_sn_bwd_apply_res");
  SNWS_CHECK_ARG(dy && x && y_for_mask && dx && dres && ctx && ctxb && beta, "snws_sn_bwd_apply_res: NULL pointer");
  SNWS_CHECK_ARG(lddy >= C && ldx >= C && lddx >= C && ldy >= C && lddres >= C,
                 "snws_sn_bwd_apply_res: pixel stride smaller than C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {dy, x, y_for_mask, dx, dres}, {lddy, ldx, ldy, lddx, lddres}))
      return run_bwd_apply<float, 4>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, 1, st, dres, lddres);
    return run_bwd_apply<float, 1>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, 1, st, dres, lddres);
  }
  if (can_vec<bf16_t>(C, {dy, x, y_for_mask, dx, dres}, {lddy, ldx, ldy, lddx, lddres}))
    return run_bwd_apply<bf16_t, 8>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, 1, st, dres, lddres);
  return run_bwd_apply<bf16_t, 1>(dy, lddy, x, ldx, y_for_mask, ldy, dx, lddx, ctx, ctxb, beta, N, HW, C, 1, st, dres, lddres);
}

// ------------------------------------------------------------------------------------------- batched passes
namespace {
template <typename Item>
int copy_batch(const Item* items, int count, Item* dst, const char* fn) {
  SNWS_CHECK_ARG(items && count >= 1 && count <= SNWS_SN_BATCH_MAX, "%s: count %d outside 1..%d (or NULL items)", fn, count,
                 SNWS_SN_BATCH_MAX);
  for (int i = 0; i < count; ++i) dst[i] = items[i];
  for (int i = count; i < SNWS_SN_BATCH_MAX; ++i) dst[i] = items[0];
  return SNWS_OK;
}
}  // namespace

extern "C" int snws_sn_fwd_a_batched(const snws_sn_fwd_item* items, int count, int N, int64_t HW, int C, void* stream) {
  SnFwdBatch b;
  if (int rc = copy_batch(items, count, b.item, "snws_sn_fwd_a_batched")) return rc;
  SNWS_CHECK_ARG(N >= 1 && C >= 1 && HW >= 1, "snws_sn_fwd_a_batched: bad sizes");
  for (int i = 0; i < count; ++i)
    SNWS_CHECK_ARG(items[i].part && items[i].shift && items[i].ctx && items[i].pack && items[i].T >= 1,
                   "snws_sn_fwd_a_batched: item %d: NULL pointer or T < 1", i);
  hipStream_t st = (hipStream_t)stream;
  sn_finalize_batched_kernel<<<dim3(cdiv(C, 64), N, count), 512, 0, st>>>(b, N, HW, C);
  SNWS_LAUNCH_CHECK("snws_sn_fwd_a_batched(finalize)");
  sn_mix_a_batched_kernel<<<dim3(N + cdiv(C, 256), 1, count), 256, 0, st>>>(b, N, C);
  SNWS_LAUNCH_CHECK("snws_sn_fwd_a_batched(mix_a)");
  return SNWS_OK;
}

extern "C" int snws_sn_mix_b_batched(const snws_sn_fwd_item* items, int count, int N, int C, float eps, float momentum,
                                     int training, int using_movavg, int using_bn, void* stream) {
  SnFwdBatch b;
  if (int rc = copy_batch(items, count, b.item, "snws_sn_mix_b_batched")) return rc;
  SNWS_CHECK_ARG(N >= 1 && C >= 1, "snws_sn_mix_b_batched: bad sizes");
  for (int i = 0; i < count; ++i) {
    const snws_sn_fwd_item& d = items[i];
    SNWS_CHECK_ARG(d.ctx && d.gamma && d.mean_weight && d.var_weight, "snws_sn_mix_b_batched: item %d: NULL pointer", i);
    SNWS_CHECK_ARG(!using_bn || (d.running_mean && d.running_var), "snws_sn_mix_b_batched: item %d: using_bn needs the running buffers", i);
    SNWS_CHECK_ARG(!(using_bn && training) || d.pack, "snws_sn_mix_b_batched: item %d: training with using_bn needs pack", i);
    // two calls of one launch that update the same running buffers would race
    for (int j = 0; j < i; ++j)
      SNWS_CHECK_ARG(!(using_bn && training) || d.running_mean != items[j].running_mean,
                     "snws_sn_mix_b_batched: items %d and %d share their running buffers", j, i);
  }
  sn_mix_b_batched_kernel<<<dim3((unsigned)cdiv64((int64_t)N * C, 256), 1, count), 256, 0, (hipStream_t)stream>>>(
      b, N, C, eps, momentum, training, using_movavg, using_bn);
  SNWS_LAUNCH_CHECK("snws_sn_mix_b_batched");
  return SNWS_OK;
}

extern "C" int snws_sn_apply_batched(const snws_sn_fwd_item* items, int count, int dtype, int N, int64_t HW, int C,
                                     int relu, void* stream) {
  SnFwdBatch b;
  if (int rc = copy_batch(items, count, b.item, "snws_sn_apply_batched")) return rc;
  SN_COMMON_CHECKS("snws_sn_apply_batched");
  bool vec = true;
  for (int i = 0; i < count; ++i) {
    const snws_sn_fwd_item& d = items[i];
    SNWS_CHECK_ARG(d.x && d.y && d.ctx && d.beta, "snws_sn_apply_batched: item %d: NULL pointer", i);
    SNWS_CHECK_ARG(d.ldx >= C && d.ldy >= C, "snws_sn_apply_batched: item %d: pixel stride smaller than C", i);
    vec = vec && (dtype == SNWS_F32 ? can_vec<float>(C, {d.x, d.y}, {d.ldx, d.ldy})
                                    : can_vec<bf16_t>(C, {d.x, d.y}, {d.ldx, d.ldy}));
  }
  if (!vec) SNWS_UNSUPPORTED("snws_sn_apply_batched: needs 16-byte aligned tensors, strides and channel count");
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == SNWS_F32 ? 4 : 8;
  const SnMap m = sn_map(C, V);
  const bool il = sn_interleave();
  const int T_ = il ? sn_apply_tiles(N, HW, C, 4 * m.PR) : sn_tiles(N, HW, C);
  const int PT = il ? 0 : (int)cdiv64(HW, T_);
  const dim3 grid(T_, N, count);
  if (dtype == SNWS_F32) {
    if (relu) sn_apply_batched_kernel<float, 4, true><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
    else sn_apply_batched_kernel<float, 4, false><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
  } else {
    if (relu) sn_apply_batched_kernel<bf16_t, 8, true><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
    else sn_apply_batched_kernel<bf16_t, 8, false><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
  }
  SNWS_LAUNCH_CHECK("snws_sn_apply_batched");
  return SNWS_OK;
}

extern "C" int snws_sn_bwd_a_batched(const snws_sn_bwd_item* items, int count, int N, int C, void* stream) {
  SnBwdBatch b;
  if (int rc = copy_batch(items, count, b.item, "snws_sn_bwd_a_batched")) return rc;
  SNWS_CHECK_ARG(N >= 1 && C >= 1, "snws_sn_bwd_a_batched: bad sizes");
  for (int i = 0; i < count; ++i) {
    const snws_sn_bwd_item& d = items[i];
    SNWS_CHECK_ARG(d.part && d.ctx && d.gamma && d.ctxb && d.packb && d.dgamma && d.dbeta && d.T >= 1,
                   "snws_sn_bwd_a_batched: item %d: NULL pointer or T < 1", i);
    for (int j = 0; j < i; ++j)
      SNWS_CHECK_ARG(d.dgamma != items[j].dgamma && d.dbeta != items[j].dbeta,
                     "snws_sn_bwd_a_batched: items %d and %d share their gradient buffers", j, i);
  }
  hipStream_t st = (hipStream_t)stream;
  sn_bwd_mix_a1_batched_kernel<<<dim3(cdiv(C, 64), N, count), 512, 0, st>>>(b, N, C);
  SNWS_LAUNCH_CHECK("snws_sn_bwd_a_batched(1)");
  sn_bwd_mix_a2_batched_kernel<<<dim3(N + cdiv(C, 256), 1, count), 256, 0, st>>>(b, N, C);
  SNWS_LAUNCH_CHECK("snws_sn_bwd_a_batched(2)");
  return SNWS_OK;
}

extern "C" int snws_sn_bwd_mix_b_batched(const snws_sn_bwd_item* items, int count, int N, int64_t HW, int C,
                                         float n_global, int training, int using_bn, void* stream) {
  SnBwdBatch b;
  if (int rc = copy_batch(items, count, b.item, "snws_sn_bwd_mix_b_batched")) return rc;
  SNWS_CHECK_ARG(N >= 1 && C >= 1 && HW >= 1 && n_global >= 1.f && (int64_t)N * C < (1ll << 31),
                 "snws_sn_bwd_mix_b_batched: bad sizes");
  for (int i = 0; i < count; ++i) {
    const snws_sn_bwd_item& d = items[i];
    SNWS_CHECK_ARG(d.ctx && d.ctxb && d.packb && d.mean_weight && d.var_weight && d.dmean_weight && d.dvar_weight,
                   "snws_sn_bwd_mix_b_batched: item %d: NULL pointer", i);
    for (int j = 0; j < i; ++j)
      SNWS_CHECK_ARG(d.dmean_weight != items[j].dmean_weight && d.dvar_weight != items[j].dvar_weight &&
                         (!d.dx_sum || d.dx_sum != items[j].dx_sum),
                     "snws_sn_bwd_mix_b_batched: items %d and %d share their gradient buffers", j, i);
  }
  sn_bwd_mix_b_batched_kernel<<<dim3(1, 1, count), 1024, 0, (hipStream_t)stream>>>(b, N, HW, C, n_global, training,
                                                                                 using_bn);
  SNWS_LAUNCH_CHECK("snws_sn_bwd_mix_b_batched");
  return SNWS_OK;
}

extern "C" int snws_sn_bwd_apply_batched(const snws_sn_bwd_item* items, int count, int dtype, int N, int64_t HW, int C,
                                         int relu, void* stream) {
  SnBwdBatch b;
  if (int rc = copy_batch(items, count, b.item, "snws_sn_bwd_apply_batched")) return rc;
  SN_COMMON_CHECKS("snws_sn_bwd_apply_batched");
  bool vec = true;
  for (int i = 0; i < count; ++i) {
    const snws_sn_bwd_item& d = items[i];
    SNWS_CHECK_ARG(d.dy && d.x && d.dx && d.ctx && d.ctxb && d.beta, "snws_sn_bwd_apply_batched: item %d: NULL pointer", i);
    SNWS_CHECK_ARG(d.lddy >= C && d.ldx >= C && d.lddx >= C, "snws_sn_bwd_apply_batched: item %d: pixel stride smaller than C", i);
    vec = vec && (dtype == SNWS_F32 ? can_vec<float>(C, {d.dy, d.x, d.dx}, {d.lddy, d.ldx, d.lddx})
                                    : can_vec<bf16_t>(C, {d.dy, d.x, d.dx}, {d.lddy, d.ldx, d.lddx}));
  }
  if (!vec) SNWS_UNSUPPORTED("snws_sn_bwd_apply_batched: needs 16-byte aligned tensors, strides and channel count");
  hipStream_t st = (hipStream_t)stream;
  const int V = dtype == SNWS_F32 ? 4 : 8;
  const SnMap m = sn_map(C, V);
  const bool il = sn_interleave();
  const int T_ = il ? sn_apply_tiles(N, HW, C, SN_BWD_ROWS<false> * m.PR) : sn_tiles(N, HW, C);
  const int PT = il ? 0 : (int)cdiv64(HW, T_);
  const dim3 grid(T_, N, count);
  if (dtype == SNWS_F32) {
    if (relu) sn_bwd_apply_batched_kernel<float, 4, true><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
    else sn_bwd_apply_batched_kernel<float, 4, false><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
  } else {
    if (relu) sn_bwd_apply_batched_kernel<bf16_t, 8, true><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
    else sn_bwd_apply_batched_kernel<bf16_t, 8, false><<<grid, SN_THREADS, 0, st>>>(b, N, HW, C, PT, m);
  }
  SNWS_LAUNCH_CHECK("snws_sn_bwd_apply_batched");
  return SNWS_OK;
}

// The apply passes with a per-(image, channel) output multiplier `post` (N x C floats, >= 0) - a feature-dropout mask
// that follows the normalisation (+ ReLU) folded into the same pass in both directions.
extern "C" int snws_sn_apply_post(const void* x, int64_t ldx, void* y, int64_t ldy, int dtype, const float* ctx,
                                  const float* beta, const float* post, int N, int64_t HW, int C, int relu, void* stream) {
  SN_COMMON_CHECKS("snws_sn_apply_post");
  SNWS_CHECK_ARG(x && y && ctx && beta && post, "snws_sn_apply_post: NULL pointer");
  SNWS_CHECK_ARG(ldx >= C && ldy >= C, "snws_sn_apply_post: pixel stride smaller than C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {x, y}, {ldx, ldy})) return run_apply<float, 4>(x, ldx, y, ldy, nullptr, 0, ctx, beta, N, HW, C, relu, st, post);
    return run_apply<float, 1>(x, ldx, y, ldy, nullptr, 0, ctx, beta, N, HW, C, relu, st, post);
  }
  if (can_vec<bf16_t>(C, {x, y}, {ldx, ldy})) return run_apply<bf16_t, 8>(x, ldx, y, ldy, nullptr, 0, ctx, beta, N, HW, C, relu, st, post);
  return run_apply<bf16_t, 1>(x, ldx, y, ldy, nullptr, 0, ctx, beta, N, HW, C, relu, st, post);
}

extern "C" int snws_sn_bwd_apply_post(const void* dy, int64_t lddy, const void* x, int64_t ldx, void* dx, int64_t lddx,
                                      int dtype, const float* ctx, const float* ctxb, const float* beta, const float* post,
                                      int N, int64_t HW, int C, int relu, void* stream) {
  SN_COMMON_CHECKS("snws_sn_bwd_apply_post");
  SNWS_CHECK_ARG(dy && x && dx && ctx && ctxb && beta && post, "snws_sn_bwd_apply_post: NULL pointer");
  SNWS_CHECK_ARG(lddy >= C && ldx >= C && lddx >= C, "snws_sn_bwd_apply_post: pixel stride smaller than C");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == SNWS_F32) {
    if (can_vec<float>(C, {dy, x, dx}, {lddy, ldx, lddx}))
      return run_bwd_apply<float, 4>(dy, lddy, x, ldx, nullptr, 0, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st, nullptr, 0, post);
    return run_bwd_apply<float, 1>(dy, lddy, x, ldx, nullptr, 0, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st, nullptr, 0, post);
  }
  if (can_vec<bf16_t>(C, {dy, x, dx}, {lddy, ldx, lddx}))
    return run_bwd_apply<bf16_t, 8>(dy, lddy, x, ldx, nullptr, 0, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st, nullptr, 0, post);
  return run_bwd_apply<bf16_t, 1>(dy, lddy, x, ldx, nullptr, 0, dx, lddx, ctx, ctxb, beta, N, HW, C, relu, st, nullptr, 0, post);
}

// ------------------------------------------------------------------------------------------- no-allocation forward
// SwitchNorm forward for callers that cannot allocate device memory (HIP graph capture, tight serving loops): the
// side buffers the stand-alone entry points take from the caller (part, shift, ctx, pack) live in one file-scope
// device array instead.  Each GPU holds its own copy of the array at its own address; the host keeps that address
// per device and passes it to the kernels like any other buffer.
namespace {

constexpr int64_t SN_SCRATCH_FLOATS = (int64_t)4 << 20;  // 16 MiB per GPU
static __device__ __align__(256) float scratch[SN_SCRATCH_FLOATS];

struct ScratchTable {
  std::atomic<float*> ptr[64];
};
ScratchTable g_scratch;

// the device address of `scratch` on the current device (looked up once per device); nullptr + error text on failure
float* scratch_on_current_device() {
  int d = -1;
  hipError_t e = hipGetDevice(&d);
  if (e != hipSuccess || d < 0 || d >= 64) {
    snws_set_error("snws_sn_apply_noalloc: cannot determine the current device (%s)", hipGetErrorString(e));
    return nullptr;
  }
  float* p = g_scratch.ptr[d].load(std::memory_order_acquire);
  if (p) return p;
  void* sym = nullptr;
  e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
  if (e != hipSuccess || !sym) {
    snws_set_error("snws_sn_apply_noalloc: hipGetSymbolAddress failed on device %d: %s", d, hipGetErrorString(e));
    return nullptr;
  }
  p = static_cast<float*>(sym);
  g_scratch.ptr[d].store(p, std::memory_order_release);
  return p;
}

inline int64_t round64(int64_t v) { return (v + 63) / 64 * 64; }

}  // namespace

// Also the warm-up call: hipGetSymbolAddress is not a capturable stream operation, so a caller that will use
// snws_sn_apply_noalloc under HIP graph capture calls this once per device (with that device current) beforehand;
// it resolves and caches the array's address when a device is present and is a plain size query otherwise.
extern "C" int64_t snws_sn_noalloc_max_floats(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) == hipSuccess && n > 0) (void)scratch_on_current_device();
  return SN_SCRATCH_FLOATS;
}

extern "C" int64_t snws_sn_noalloc_floats(int N, int64_t HW, int C) {
  if (N < 1 || HW < 1 || C < 1) return -1;
  const int T = sn_tiles(N, HW, C);
  return round64((int64_t)N * T * C * 2) + round64((int64_t)N * C) + round64(ctx_off(N, C).total) + round64(2 * (int64_t)C + 1);
}

extern "C" int snws_sn_apply_noalloc(const void* x, int64_t ldx, void* y, int64_t ldy, const void* residual,
                                     int64_t ldr, int dtype, const float* gamma, const float* beta,
                                     const float* mean_weight, const float* var_weight, float* running_mean,
                                     float* running_var, int N, int64_t HW, int C, float eps, float momentum,
                                     int training, int using_movavg, int using_bn, int relu, void* stream) {
  SN_COMMON_CHECKS("snws_sn_apply_noalloc");
  SNWS_CHECK_ARG(x && y && gamma && beta && mean_weight && var_weight, "snws_sn_apply_noalloc: NULL pointer");
  SNWS_CHECK_ARG(HW >= 2, "snws_sn_apply_noalloc: the unbiased variance needs at least 2 positions per image");
  SNWS_CHECK_ARG(!using_bn || (running_mean && running_var), "snws_sn_apply_noalloc: using_bn needs the running buffers");
  const int64_t need = snws_sn_noalloc_floats(N, HW, C);
  SNWS_CHECK_ARG(need <= SN_SCRATCH_FLOATS,
                 "snws_sn_apply_noalloc: N=%d HW=%lld C=%d needs %lld floats of side buffers, the static scratch array "
                 "holds %lld; use the stand-alone entry points with caller-provided buffers",
                 N, (long long)HW, C, (long long)need, (long long)SN_SCRATCH_FLOATS);
  float* base = scratch_on_current_device();
  if (!base) return SNWS_ERR_HIP;
  const int T = sn_tiles(N, HW, C);
  float* part = base;
  float* shift = part + round64((int64_t)N * T * C * 2);
  float* ctx = shift + round64((int64_t)N * C);
  float* pack = ctx + round64(ctx_off(N, C).total);
  // every float of part / shift / ctx / pack that a later kernel reads is written by an earlier one of this call:
  // nothing has to start at zero, whatever the previous call left behind
  int rc = snws_sn_stats(x, dtype, N, HW, C, ldx, part, shift, stream);
  if (rc) return rc;
  rc = snws_sn_finalize(part, T, shift, 1, N, HW, C, ctx, stream);
  if (rc) return rc;
  rc = snws_sn_mix_a(ctx, pack, N, C, stream);
  if (rc) return rc;
  rc = snws_sn_mix_b(ctx, pack, gamma, mean_weight, var_weight, running_mean, running_var, N, C, eps, momentum,
                     training, using_movavg, using_bn, stream);
  if (rc) return rc;
  return snws_sn_apply(x, ldx, y, ldy, residual, ldr, dtype, ctx, beta, N, HW, C, relu, stream);
}
