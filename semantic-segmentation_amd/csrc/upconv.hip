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
#include <atomic>

namespace {

__device__ __attribute__((aligned(256))) unsigned int up_zero_page[64];   // never written: all zero

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


// v_dot2c_f32_bf16: acc + a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (products exact, f32 accumulation); with
// v_perm_b32 to pair up the same channel of two pixels it halves the ALU work of unpack + 2 FMA per pair.
typedef __attribute__((ext_vector_type(2))) __bf16 up_bf16x2;
__device__ __forceinline__ float dot2_bf16(unsigned a, unsigned b, float acc) {
  return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(up_bf16x2, a), __builtin_bit_cast(up_bf16x2, b), acc, false);
}
// (a.lo, b.lo) and (a.hi, b.hi) as packed pairs: element 0 from a, element 1 from b
__device__ __forceinline__ unsigned pair_lo(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ unsigned pair_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// ---- the separable forward with the source footprint staged in LDS (bf16, 3x3, k in {2,4,8,16}, 64-channel chunks).
// The per-pixel kernels above issue 36 (gather) or ~15 (separable) 16-byte global loads per 8 outputs and run at the
// L1 rate; here a workgroup copies the footprint of its 8x32 output strip - (8/k + 2..3) x (32/k + 2) source pixels,
// the three taps of one filter row at a time (192 channels) - global->LDS with 16-byte LDS-DMA loads and every thread
// reads its operands from there; the column interpolation runs on packed bf16 pairs (v_dot2c_f32_bf16: exact products,
// f32 accumulation), the row interpolation in f32 as in upconv_sep_body.  Source pixels are clamped into the image while staging (= the interpolation's border clamp);
// a 16-byte pad slot per pixel (400-byte pixel stride) spreads the pixels over the LDS banks.
// Optionally the strip's per-channel sum / sum of squares of the ROUNDED outputs minus the bias are written out, in
// the layout of the convolution kernels' statistics epilogue, so SwitchNorm does not re-read the tensor.
template <int K, int PH>
struct UpStage {
  static constexpr int FLMIN = up_fl<K>(PH * SEP_TH - 1);
  static constexpr int R = up_fl<K>(PH * SEP_TH + SEP_TH) + 2 - FLMIN;
  static constexpr int CW = up_fl<K>(SEP_TW) + 3;          // columns fl(-1) = -1 ... fl(32) + 1
  static constexpr int QS = 25;                            // 16-byte slots per staged pixel: 3 taps x 8 + 1 pad
  static constexpr int NS = R * CW * QS;
  static constexpr int NI = (NS + 63) / 64;                // LDS-DMA instructions per stage (1 KiB each)
  static constexpr int JN = (NI + 3) / 4;                  // per wave
  static constexpr int LDS_BYTES = NI * 1024;
};

__device__ __forceinline__ void up_glds16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

template <int K, int PH>
__device__ __forceinline__ void upconv_lds_body(const bf16_t* __restrict__ Z, int64_t ldz, bf16_t* __restrict__ y,
                                                int64_t ldy, const float* __restrict__ bias, float* __restrict__ stat,
                                                int h, int w, int Co, int tilesX, unsigned char* smem) {
  using G = UpStage<K, PH>;
  constexpr int LOG2K2 = K == 2 ? 2 : K == 4 ? 3 : K == 8 ? 4 : 5;      // log2(2K)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c8 = tid & 7, pxl = tid >> 3;
  const int tx_ = blockIdx.x % tilesX, cchunk = blockIdx.x / tilesX;
  const int px0 = tx_ * SEP_TW, px = px0 + pxl, ch = cchunk * 64 + c8 * 8;
  const int py0 = blockIdx.y * SEP_TH, n = blockIdx.z;
  const int H = h * K, W = w * K;
  const int base_lo = (py0 - PH * SEP_TH) / K + G::FLMIN;      // first staged source row (before clamping)
  const int col_lo = px0 / K - 1;                               // first staged source column (before clamping)

  // per-lane source of every LDS-DMA load of a stage, as a byte offset from the image's tap-row-0 base
  const unsigned char* zimg = reinterpret_cast<const unsigned char*>(Z + (int64_t)n * h * w * ldz + cchunk * 64);
  unsigned src[G::JN];
#pragma unroll
  for (int j = 0; j < G::JN; ++j) {
    const int s = (wave + 4 * j) * 64 + lane;
    const int q = s / G::QS, c = s - q * G::QS;
    const int qr = q / G::CW, qc = q - qr * G::CW;
    int row = base_lo + qr, col = col_lo + qc;
    row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
    col = col < 0 ? 0 : (col > w - 1 ? w - 1 : col);
    const int tx = c >> 3, jj = c & 7;
    const bool real = s < G::NS && c < 24;
    src[j] = real ? (unsigned)((((int64_t)row * w + col) * ldz + (int64_t)tx * Co + jj * 8) * 2) : 0u;
  }

  // columns this thread's three filter columns touch: local staged column and interpolation weight
  int xl[3];
  float lx[3];
  bool xok[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int qx = px + tx - 1;
    xok[tx] = qx >= 0 && qx < W && px < W;
    const int t2 = 2 * qx + 1 - K;                              // (qx + 0.5)/K - 0.5 = t2 / (2K)
    xl[tx] = (t2 >> LOG2K2) - col_lo;
    lx[tx] = (float)(t2 & (2 * K - 1)) * (1.f / (float)(2 * K));
    if (!xok[tx]) xl[tx] = 0;
  }
  unsigned wpk[3];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) wpk[tx] = pack_bf16x2(1.f - lx[tx], lx[tx]);
  float out[SEP_TH][8];
#pragma unroll
  for (int py = 0; py < SEP_TH; ++py)
#pragma unroll
    for (int v = 0; v < 8; ++v) out[py][v] = bias ? bias[ch + v] : 0.f;

#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    if (ty) __syncthreads();                                    // every wave is done reading the previous stage
    const unsigned char* zt = zimg + (int64_t)ty * 3 * Co * 2;
#pragma unroll
    for (int j = 0; j < G::JN; ++j)
      if (wave + 4 * j < G::NI) up_glds16(zt + src[j], smem + (wave + 4 * j) * 1024);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    constexpr int dummy = 0; (void)dummy;
    const int lo = up_fl<K>(PH * SEP_TH + ty - 1) - G::FLMIN, hi = up_fl<K>(PH * SEP_TH + SEP_TH - 1 + ty - 1) + 1 - G::FLMIN;
    float Gr[G::R][8];
#pragma unroll
    for (int rr = 0; rr < G::R; ++rr) {
      if (rr < lo || rr > hi) continue;
#pragma unroll
      for (int v = 0; v < 8; ++v) Gr[rr][v] = 0.f;
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        if (!xok[tx]) continue;
        const unsigned char* pa = smem + ((rr * G::CW + xl[tx]) * G::QS + tx * 8 + c8) * 16;
        const uint4 ua = *reinterpret_cast<const uint4*>(pa), ub = *reinterpret_cast<const uint4*>(pa + G::QS * 16);
        const unsigned a4[4] = {ua.x, ua.y, ua.z, ua.w}, b4[4] = {ub.x, ub.y, ub.z, ub.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {       // (1 - lx) * a + lx * b per channel: the two weights are exact in bf16
          Gr[rr][2 * e] = dot2_bf16(pair_lo(a4[e], b4[e]), wpk[tx], Gr[rr][2 * e]);
          Gr[rr][2 * e + 1] = dot2_bf16(pair_hi(a4[e], b4[e]), wpk[tx], Gr[rr][2 * e + 1]);
        }
      }
    }
#pragma unroll
    for (int py = 0; py < SEP_TH; ++py) {
      const int qy = py0 + py + ty - 1;
      if (qy < 0 || qy >= H) continue;     // the conv's zero padding
      const int ia = up_fl<K>(PH * SEP_TH + py + ty - 1) - G::FLMIN;
      const float lam = up_lam<K>(PH * SEP_TH + py + ty - 1);
#pragma unroll
      for (int v = 0; v < 8; ++v) out[py][v] += (1.f - lam) * Gr[ia][v] + lam * Gr[ia + 1][v];
    }
  }

  // ---- stores (8 lanes x 16 B = one pixel's 64 channels) and the optional per-channel partial moments
  float S[8], Q[8], sh[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    S[v] = Q[v] = 0.f;
    sh[v] = (stat && bias) ? bias[ch + v] : 0.f;
  }
  const bool col_ok = px < W;
#pragma unroll
  for (int py = 0; py < SEP_TH; ++py) {
    if (col_ok && py0 + py < H) {
      uint4 v;
      v.x = pack_bf16x2(out[py][0], out[py][1]);
      v.y = pack_bf16x2(out[py][2], out[py][3]);
      v.z = pack_bf16x2(out[py][4], out[py][5]);
      v.w = pack_bf16x2(out[py][6], out[py][7]);
      *reinterpret_cast<uint4*>(y + (((int64_t)n * H + py0 + py) * W + px) * ldy + ch) = v;
      if (stat) {
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f0 = __uint_as_float(u[e] << 16) - sh[2 * e];
          const float f1 = __uint_as_float(u[e] & 0xffff0000u) - sh[2 * e + 1];
          S[2 * e] += f0; Q[2 * e] += f0 * f0;
          S[2 * e + 1] += f1; Q[2 * e + 1] += f1 * f1;
        }
      }
    }
  }
  if (stat) {
    __syncthreads();                        // the last stage's readers are done: its LDS is reused for the partials
    float* st_lds = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        S[e] += __shfl_xor(S[e], o, 64);
        Q[e] += __shfl_xor(Q[e], o, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        st_lds[(wave * 64 + lane * 8 + e) * 2] = S[e];
        st_lds[(wave * 64 + lane * 8 + e) * 2 + 1] = Q[e];
      }
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        s += st_lds[(wv * 64 + tid) * 2];
        q += st_lds[(wv * 64 + tid) * 2 + 1];
      }
      const int64_t tile = (int64_t)blockIdx.y * tilesX + tx_;
      const int64_t T = (int64_t)gridDim.y * tilesX;
      float* o = stat + (((int64_t)n * T + tile) * Co + cchunk * 64 + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

template <int K>
__global__ __launch_bounds__(256, 3) void upconv_fwd_lds_kernel(const bf16_t* __restrict__ Z, int64_t ldz,
                                                             bf16_t* __restrict__ y, int64_t ldy,
                                                             const float* __restrict__ bias, float* __restrict__ stat,
                                                             int h, int w, int Co, int tilesX) {
  extern __shared__ __attribute__((aligned(16))) unsigned char up_smem[];
  if constexpr (K == 16) {
    if (blockIdx.y & 1) upconv_lds_body<K, 1>(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, up_smem);
    else upconv_lds_body<K, 0>(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, up_smem);
  } else {
    upconv_lds_body<K, 0>(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, up_smem);
  }
}

template <int K>
constexpr int up_lds_bytes() {
  return UpStage<K, 0>::LDS_BYTES > UpStage<K, (K == 16 ? 1 : 0)>::LDS_BYTES ? UpStage<K, 0>::LDS_BYTES
                                                                             : UpStage<K, (K == 16 ? 1 : 0)>::LDS_BYTES;
}

// true when the LDS-staged kernel takes this problem
bool lds_fwd_ok(int dtype, const void* Z, int64_t ldz, const void* y, int64_t ldy, int N, int h, int w, int Co, int k, int ksize) {
  if (dtype != SNWS_BF16 || ksize != 3 || Co % 64 || ldz % 8 || ldy % 8) return false;
  if (k != 2 && k != 4 && k != 8 && k != 16) return false;
  if ((reinterpret_cast<uintptr_t>(Z) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return false;
  if ((int64_t)h * w * ldz * 2 >= 0xFFFFFFFFll) return false;                  // 32-bit source offsets per image
  if (cdiv(h * k, SEP_TH) > 65535 || N > 65535) return false;
  return true;
}

template <int K>
int launch_fwd_lds_k(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, float* stat, int N, int h, int w,
                     int Co, hipStream_t st) {
  constexpr int LDS = up_lds_bytes<K>() < 2048 ? 2048 : up_lds_bytes<K>();
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_fwd_lds_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("upconv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  const int H = h * K, W = w * K;
  const int tilesX = cdiv(W, SEP_TW), tilesY = cdiv(H, SEP_TH);
  dim3 grid((unsigned)(tilesX * (Co / 64)), (unsigned)tilesY, (unsigned)N);
  upconv_fwd_lds_kernel<K><<<grid, 256, LDS, st>>>((const bf16_t*)Z, ldz, (bf16_t*)y, ldy, bias, stat, h, w, Co, tilesX);
  return SNWS_OK;
}

int launch_fwd_lds(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, float* stat, int N, int h, int w,
                   int Co, int k, hipStream_t st) {
  switch (k) {
    case 2: return launch_fwd_lds_k<2>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
    case 4: return launch_fwd_lds_k<4>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
    case 8: return launch_fwd_lds_k<8>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
    default: return launch_fwd_lds_k<16>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
  }
}

// ---- the LDS-staged forward on the matrix cores (bf16, 3x3, k in {2,4,8,16}, 64-channel chunks).
// y(p) = b + sum_ty sum_{(tx, r, c)} wy(py+ty-1; r) wx(px+tx-1; c) Z_{ty,tx}[r][c] is a small matrix product per run of
// output pixels: a 4x4 block of outputs touches at most NRC x NRC source pixels per tap (3x3; 4x4 for k = 2), so per
// filter row ty it is  D[64 channels][16 pixels] = Zrows[64][KIDX] * Wt[KIDX][16],  KIDX = 3 taps x NRC x NRC <= 32
// (64 for k = 2).  Zrows comes straight from the staged footprint with the transposing LDS read (a contraction row =
// one source pixel's 16 channels of one tap), Wt - products of two interpolation weights, multiples of 1/(4 k^2) -
// is split into two bf16 halves that hold it EXACTLY (8 + 2 significant bits), so every product is exact and only the
// f32 accumulation order differs from the separable kernels above.  Weights depend on the output position only
// through its phase inside the source cell (and on the image border through zeroed taps): a workgroup walks a run of
// 8x32 strips along one strip row, builds the weights once and keeps them in registers.  The interpolation work moves
// from ~25 VALU operations per output to 1.5 matrix instructions per 16 outputs; what remains per output is the
// rounding, the store and the statistics.
typedef __attribute__((ext_vector_type(8))) __bf16 up_bf16x8;
typedef __attribute__((ext_vector_type(4))) float up_f32x4;
typedef __attribute__((ext_vector_type(4))) short up_short4;
typedef __attribute__((ext_vector_type(8))) short up_short8;

__device__ __forceinline__ up_bf16x8 up_tr_frag(const unsigned char* p_lo, const unsigned char* p_hi) {
  const up_short4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) up_short4*)p_lo);
  const up_short4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) up_short4*)p_hi);
  up_short8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(up_bf16x8, v);
}

template <int K>
struct UpMfma {
  static constexpr int NRC = K == 2 ? 4 : 3;            // source rows / columns four consecutive outputs of one tap touch
  static constexpr int KIDX = 3 * NRC * NRC;            // contraction entries per filter row: (tx, source row, source column)
  static constexpr int KS = (KIDX + 31) / 32;           // 16x16x32 steps
  static constexpr int OUT_RS = 144;                    // bytes per pixel of the epilogue tile (128 + 16 pad)
  static constexpr int OUT_BYTES = SEP_TH * SEP_TW * OUT_RS;
  static constexpr int STAGE_BYTES = up_lds_bytes<K>();              // one filter row's footprint
  static constexpr int BUF_BYTES = 3 * STAGE_BYTES;                   // all three filter rows of a strip
  // two buffers (the next strip's copy in flight during this strip's work) where two workgroups per CU still fit;
  // otherwise one buffer, which the epilogue tile then reuses
  static constexpr int NBUF = 2 * BUF_BYTES + OUT_BYTES + 2048 <= 80 * 1024 ? 2 : 1;
  static constexpr int OUT_OFF = NBUF == 2 ? 2 * BUF_BYTES : 0;
  static constexpr int MAIN_BYTES = NBUF == 2 ? 2 * BUF_BYTES + OUT_BYTES : (BUF_BYTES > OUT_BYTES ? BUF_BYTES : OUT_BYTES);
  static constexpr int LDS_BYTES = MAIN_BYTES + 4 * 64 * 2 * 4;
};

// Epilogue shared by the matrix-core forwards: accumulators (+ bias) -> bf16 tile in LDS (lane holds pixel m of block
// (i, j) and channels 16 cb + 4 g + (0..3)), then 8 lanes x 16 B = one pixel's 64 channels per store and the optional
// per-channel partial moments of the rounded outputs minus the bias.  The caller has made sure nobody still reads the
// tile's LDS; returns with every wave past its last read of st_lds' previous contents (one barrier inside).
__device__ __forceinline__ void up_mfma_epilogue(const up_f32x4 (&acc)[2][2][4], const float (&b4)[4][4], const float (&sh)[8],
                                                 unsigned char* out_lds, float* st_lds, bf16_t* __restrict__ y, int64_t ldy,
                                                 float* __restrict__ stat, int n, int H, int W, int py0, int px0, int Co,
                                                 int cchunk, int tilesX, int tx_) {
  constexpr int OUT_RS = 144;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m = lane & 15, g = lane >> 4;
  const int c8 = tid & 7, pxl = tid >> 3;
  const int ch = cchunk * 64 + c8 * 8;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int qp = (4 * i + (m >> 2)) * SEP_TW + 4 * (wave + 4 * jj) + (m & 3);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const up_f32x4 v = acc[i][jj][cb];
        *reinterpret_cast<uint2*>(out_lds + qp * OUT_RS + (16 * cb + 4 * g) * 2) =
            make_uint2(pack_bf16x2(v[0] + b4[cb][0], v[1] + b4[cb][1]), pack_bf16x2(v[2] + b4[cb][2], v[3] + b4[cb][3]));
      }
    }
  __syncthreads();
  const int px = px0 + pxl;
  const bool col_ok = px < W;
  float S[8], Q[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) S[v] = Q[v] = 0.f;
#pragma unroll
  for (int py = 0; py < SEP_TH; ++py) {
    if (col_ok && py0 + py < H) {
      const uint4 v = *reinterpret_cast<const uint4*>(out_lds + (py * SEP_TW + pxl) * OUT_RS + c8 * 16);
      *reinterpret_cast<uint4*>(y + (((int64_t)n * H + py0 + py) * W + px) * ldy + ch) = v;
      if (stat) {
        const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float f0 = __uint_as_float(u[e] << 16) - sh[2 * e];
          const float f1 = __uint_as_float(u[e] & 0xffff0000u) - sh[2 * e + 1];
          S[2 * e] += f0; Q[2 * e] += f0 * f0;
          S[2 * e + 1] += f1; Q[2 * e + 1] += f1 * f1;
        }
      }
    }
  }
  if (stat) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1) {
        S[e] += __shfl_xor(S[e], o, 64);
        Q[e] += __shfl_xor(Q[e], o, 64);
      }
    }
    if (lane < 8) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        st_lds[(wave * 64 + lane * 8 + e) * 2] = S[e];
        st_lds[(wave * 64 + lane * 8 + e) * 2 + 1] = Q[e];
      }
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) {
        s += st_lds[(wv * 64 + tid) * 2];
        q += st_lds[(wv * 64 + tid) * 2 + 1];
      }
      const int64_t tile = (int64_t)blockIdx.y * tilesX + tx_;
      const int64_t T = (int64_t)gridDim.y * tilesX;
      float* o = stat + (((int64_t)n * T + tile) * Co + cchunk * 64 + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
}

template <int K, int PH>
__device__ __forceinline__ void upconv_mfma_body(const bf16_t* __restrict__ Z, int64_t ldz, bf16_t* __restrict__ y,
                                                 int64_t ldy, const float* __restrict__ bias, float* __restrict__ stat,
                                                 int h, int w, int Co, int tilesX, int tilesPerWg, unsigned char* smem) {
  using G = UpStage<K, PH>;
  using M = UpMfma<K>;
  constexpr int LOG2K2 = K == 2 ? 2 : K == 4 ? 3 : K == 8 ? 4 : 5;      // log2(2K)
  constexpr int NRC = M::NRC, KS = M::KS;
  constexpr int PIXB = G::QS * 16;                                      // bytes per staged pixel
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunk = Co / 64;
  const int cchunk = blockIdx.x % nchunk, xs = blockIdx.x / nchunk;
  const int py0 = blockIdx.y * SEP_TH, n = blockIdx.z;
  const int H = h * K, W = w * K;
  const int base_lo = (py0 - PH * SEP_TH) / K + G::FLMIN;      // first staged source row (before clamping)
  const unsigned char* zimg = reinterpret_cast<const unsigned char*>(Z + (int64_t)n * h * w * ldz + cchunk * 64);

  // ---- lane geometry.  Matrix part: m = output pixel of a 4x4 block (row m >> 2, column m & 3), g = k octet.
  // A wave owns the blocks (i, j): output rows 4i.. (i = 0, 1), output columns 4j.. with j = wave + 4 jj (jj = 0, 1);
  // 16 columns apart the column phase repeats for every k, so one weight set serves both jj.
  const int m = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  auto decode = [&](int kidx, int& tx, int& r, int& c) -> bool {
    const bool valid = kidx < M::KIDX;
    tx = kidx / (NRC * NRC);
    const int rem = kidx - tx * (NRC * NRC);
    r = rem / NRC;
    c = rem - r * NRC;
    if (!valid) tx = r = c = 0;      // a real (finite) source value times a zero weight
    return valid;
  };
  // column interpolation of this lane's output column for the three taps, relative to the block's first source column
  int c0[3];           // first source column (relative to the 16-column group's origin) a block touches for tap tx
  float wxc[3][NRC];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int q0 = 4 * wave + tx - 1;
    c0[tx] = (2 * q0 + 1 - K) >> LOG2K2;
    const int t2 = 2 * (q0 + (m & 3)) + 1 - K;
    const int d = (t2 >> LOG2K2) - c0[tx];
    const float lam = (float)(t2 & (2 * K - 1)) * (1.f / (float)(2 * K));
#pragma unroll
    for (int c = 0; c < NRC; ++c) wxc[tx][c] = c == d ? 1.f - lam : (c == d + 1 ? lam : 0.f);
  }
  // weights of this lane's entries (kidx = 32 ks + 8 g + e) per (filter row, block row), split hi + lo, packed in pairs
  unsigned whi[3][2][KS][4], wlo[3][2][KS][4];
  int etx[KS][8];      // tap column of each entry (for the border masks)
  {
    float wxe[KS][8];
    int er[KS][8];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        int tx, r, c;
        const bool valid = decode(ks * 32 + 8 * g + e, tx, r, c);
        float wv = 0.f;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3)
#pragma unroll
          for (int cc = 0; cc < NRC; ++cc) wv = (tx == t3 && c == cc) ? wxc[t3][cc] : wv;
        wxe[ks][e] = valid ? wv : 0.f;
        er[ks][e] = r;
        etx[ks][e] = tx;
      }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int q0 = PH * SEP_TH + 4 * i + ty - 1;                 // block's first output row + tap shift (strip-relative)
        const int r0 = (2 * q0 + 1 - K) >> LOG2K2;
        const int t2 = 2 * (q0 + (m >> 2)) + 1 - K;
        const int d = (t2 >> LOG2K2) - r0;
        const float lam = (float)(t2 & (2 * K - 1)) * (1.f / (float)(2 * K));
        const int qy = py0 + 4 * i + (m >> 2) + ty - 1;
        const bool rowok = qy >= 0 && qy < H;                         // the conv's zero padding
        float wyr[NRC];
#pragma unroll
        for (int r = 0; r < NRC; ++r) wyr[r] = !rowok ? 0.f : (r == d ? 1.f - lam : (r == d + 1 ? lam : 0.f));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            unsigned hi2[2], lo2[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int e = 2 * e2 + u;
              float wy = 0.f;
#pragma unroll
              for (int r = 0; r < NRC; ++r) wy = er[ks][e] == r ? wyr[r] : wy;
              const float wv = wy * wxe[ks][e];                      // a multiple of 1/(4 K^2) below 1: at most 10 significant bits
              const unsigned hb = __float_as_uint(wv) & 0xffff0000u;  // its top 8 significant bits ...
              const float rest = wv - __uint_as_float(hb);            // ... and the rest (exact, at most 2 significant bits)
              hi2[u] = hb >> 16;
              lo2[u] = __float_as_uint(rest) >> 16;
            }
            whi[ty][i][ks][e2] = hi2[0] | (hi2[1] << 16);
            wlo[ty][i][ks][e2] = lo2[0] | (lo2[1] << 16);
          }
      }
  }
  // transposing-read addresses of the Z operand: lane -> contraction row 32 ks + 8 g + 4 hq + tq, channels 4 tp .. 4 tp + 3
  int ar[KS][2];                 // source row of that entry inside the block's footprint
  int acol[2][KS][2];            // byte offset of (staged column, tap, channel quad) for jj = 0, 1 (set per strip)
  int atx[KS][2], ac[KS][2];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks)
#pragma unroll
    for (int hq = 0; hq < 2; ++hq) {
      int tx, r, c;
      decode(ks * 32 + 8 * g + 4 * hq + tq, tx, r, c);
      ar[ks][hq] = r; atx[ks][hq] = tx; ac[ks][hq] = c;
    }
#pragma unroll
  for (int jj = 0; jj < 2; ++jj)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int hq = 0; hq < 2; ++hq) {
        int cc0 = 0;
#pragma unroll
        for (int t3 = 0; t3 < 3; ++t3) cc0 = atx[ks][hq] == t3 ? c0[t3] : cc0;
        int col = cc0 + (16 / K) * jj + 1 + ac[ks][hq];            // staged column 0 = source column (strip origin)/K - 1
        col = col > G::CW - 1 ? G::CW - 1 : col;                     // (only entries with a zero weight get clamped)
        acol[jj][ks][hq] = col * PIXB + atx[ks][hq] * 128 + tp * 8;
      }

  // epilogue mapping (as the kernel above): 8 lanes x 16 B = one pixel's 64 channels, a thread walks down the 8 rows
  const int ch = cchunk * 64 + (tid & 7) * 8;
  unsigned char* out_lds = smem + M::OUT_OFF;
  float* st_lds = reinterpret_cast<float*>(smem + M::MAIN_BYTES);
  float b4[4][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[cb][e] = bias ? bias[cchunk * 64 + 16 * cb + 4 * g + e] : 0.f;
  float sh[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) sh[v] = (stat && bias) ? bias[ch + v] : 0.f;

  // ---- global -> LDS copy of a strip's footprint (all three filter rows) into buffer b
  auto stage = [&](int tx_s, int b) {
    const int col_lo = tx_s * SEP_TW / K - 1;
    unsigned char* dst = smem + b * M::BUF_BYTES;
#pragma unroll
    for (int j = 0; j < G::JN; ++j) {
      if (wave + 4 * j < G::NI) {
        const int s = (wave + 4 * j) * 64 + lane;
        const int q = s / G::QS, c = s - q * G::QS;
        const int qr = q / G::CW, qc = q - qr * G::CW;
        int row = base_lo + qr, col = col_lo + qc;
        row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
        col = col < 0 ? 0 : (col > w - 1 ? w - 1 : col);
        const int tx = c >> 3, jc = c & 7;
        const bool real = s < G::NS && c < 24;
        const unsigned char* sp = zimg + (real ? (((int64_t)row * w + col) * ldz + (int64_t)tx * Co + jc * 8) * 2 : 0);
#pragma unroll
        for (int ty = 0; ty < 3; ++ty)
          up_glds16(sp + (int64_t)ty * 3 * Co * 2, dst + ty * M::STAGE_BYTES + (wave + 4 * j) * 1024);
      }
    }
  };

  const int t_begin = xs * tilesPerWg;
  const int t_end = t_begin + tilesPerWg < tilesX ? t_begin + tilesPerWg : tilesX;
  if (M::NBUF == 2) stage(t_begin, 0);
  for (int tx_ = t_begin; tx_ < t_end; ++tx_) {
    const int px0 = tx_ * SEP_TW;
    const int buf = M::NBUF == 2 ? ((tx_ - t_begin) & 1) : 0;
    if (M::NBUF == 1) {
      __syncthreads();                                            // every wave is done reading the previous strip
      stage(tx_, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                              // this strip's footprint has landed for every wave
    if (M::NBUF == 2 && tx_ + 1 < t_end) stage(tx_ + 1, buf ^ 1);  // in flight during the matrix work and the epilogue
    const unsigned char* zb = smem + buf * M::BUF_BYTES;
    // taps that fall outside the image columns (the strip touches the left / right border): zero their weights
    const bool edge = px0 == 0 || px0 + SEP_TW >= W;            // block-uniform
    unsigned cmask[2][KS][4];
    if (edge) {
#pragma unroll
      for (int jj = 0; jj < 2; ++jj) {
        bool ok[3];
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          const int qx = px0 + 16 * jj + 4 * wave + (m & 3) + tx - 1;
          ok[tx] = qx >= 0 && qx < W;
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            bool o0 = false, o1 = false;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
              o0 = etx[ks][2 * e2] == t3 ? ok[t3] : o0;
              o1 = etx[ks][2 * e2 + 1] == t3 ? ok[t3] : o1;
            }
            cmask[jj][ks][e2] = (o0 ? 0xffffu : 0u) | (o1 ? 0xffff0000u : 0u);
          }
      }
    }

    up_f32x4 acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[i][jj][cb] = up_f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      const unsigned char* zs = zb + ty * M::STAGE_BYTES;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        constexpr int dummy = 0; (void)dummy;
        const int rr0 = up_fl<K>(PH * SEP_TH + 4 * i + ty - 1) - G::FLMIN;      // staged row of the block's first source row
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          int rowb[2];
#pragma unroll
          for (int hq = 0; hq < 2; ++hq) {
            int rr = rr0 + ar[ks][hq];
            rr = rr > G::R - 1 ? G::R - 1 : rr;                   // (only entries with a zero weight get clamped)
            rowb[hq] = rr * (G::CW * PIXB);
          }
#pragma unroll
          for (int jj = 0; jj < 2; ++jj) {
            unsigned h4[4], l4[4];
#pragma unroll
            for (int e2 = 0; e2 < 4; ++e2) {
              h4[e2] = whi[ty][i][ks][e2];
              l4[e2] = wlo[ty][i][ks][e2];
              if (edge) { h4[e2] &= cmask[jj][ks][e2]; l4[e2] &= cmask[jj][ks][e2]; }
            }
            const up_bf16x8 wh = __builtin_bit_cast(up_bf16x8, make_uint4(h4[0], h4[1], h4[2], h4[3]));
            const up_bf16x8 wl = __builtin_bit_cast(up_bf16x8, make_uint4(l4[0], l4[1], l4[2], l4[3]));
            const unsigned char* p0 = zs + rowb[0] + acol[jj][ks][0];
            const unsigned char* p1 = zs + rowb[1] + acol[jj][ks][1];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
              const up_bf16x8 zf = up_tr_frag(p0 + cb * 32, p1 + cb * 32);
              acc[i][jj][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, wh, acc[i][jj][cb], 0, 0, 0);
              acc[i][jj][cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, wl, acc[i][jj][cb], 0, 0, 0);
            }
          }
        }
      }
    }

    if (M::NBUF == 1) __syncthreads();                            // one buffer: the tile reuses it once every wave is done reading
    up_mfma_epilogue(acc, b4, sh, out_lds, st_lds, y, ldy, stat, n, H, W, py0, px0, Co, cchunk, tilesX, tx_);
  }
}

// ---- k = 2: four consecutive outputs of one tap touch up to 4 source pixels per axis, so the contraction is done per
// tap column on v_mfma_f32_16x16x16_bf16: entry (source row r, source column c) = 4 r + c fills the 16 exactly, a lane
// (pixel m, k quad g) holds the weights wy(r = g) * wx(tx; c = 0..3), and one transposing read (k rows = the four source
// columns of row g) feeds the hi and the lo product.  Block rows 0..3 and 4..7 have the same phase, so one weight set
// serves all four blocks of a wave; taps outside the image are zeroed per lane.  The three filter rows are staged one at
// a time (43 KB each) and the epilogue tile reuses that buffer.
__device__ __forceinline__ void upconv_mfma2_body(const bf16_t* __restrict__ Z, int64_t ldz, bf16_t* __restrict__ y,
                                                  int64_t ldy, const float* __restrict__ bias, float* __restrict__ stat,
                                                  int h, int w, int Co, int tilesX, int tilesPerWg, unsigned char* smem) {
  constexpr int K = 2;
  using G = UpStage<2, 0>;
  constexpr int PIXB = G::QS * 16;
  constexpr int OUT_BYTES = SEP_TH * SEP_TW * 144;
  constexpr int MAIN = G::LDS_BYTES > OUT_BYTES ? G::LDS_BYTES : OUT_BYTES;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunk = Co / 64;
  const int cchunk = blockIdx.x % nchunk, xs = blockIdx.x / nchunk;
  const int py0 = blockIdx.y * SEP_TH, n = blockIdx.z;
  const int H = h * K, W = w * K;
  const int base_lo = py0 / K + G::FLMIN;
  const unsigned char* zimg = reinterpret_cast<const unsigned char*>(Z + (int64_t)n * h * w * ldz + cchunk * 64);
  const int m = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;

  // weights: whi / wlo[ty][tx] = the four entries c = 0..3 of this lane's source row g, packed in pairs
  int c0[3];
  float wxc[3][4];
#pragma unroll
  for (int tx = 0; tx < 3; ++tx) {
    const int q0 = 4 * wave + tx - 1;
    c0[tx] = (2 * q0 - 1) >> 2;                                  // floor((q + 0.5) / 2 - 0.5) = (2 q - 1) >> 2
    const int t2 = 2 * (q0 + (m & 3)) - 1;
    const int d = (t2 >> 2) - c0[tx];
    const float lam = (float)(t2 & 3) * 0.25f;
#pragma unroll
    for (int c = 0; c < 4; ++c) wxc[tx][c] = c == d ? 1.f - lam : (c == d + 1 ? lam : 0.f);
  }
  unsigned whi[3][3][2], wlo[3][3][2];
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    const int q0 = ty - 1;                                        // (block row 4 i has the same phase)
    const int t2 = 2 * (q0 + (m >> 2)) - 1;
    const int d = (t2 >> 2) - ((2 * q0 - 1) >> 2);
    const float lam = (float)(t2 & 3) * 0.25f;
    const float wy = g == d ? 1.f - lam : (g == d + 1 ? lam : 0.f);
#pragma unroll
    for (int tx = 0; tx < 3; ++tx)
#pragma unroll
      for (int e2 = 0; e2 < 2; ++e2) {
        unsigned hi2[2], lo2[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float wv = wy * wxc[tx][2 * e2 + u];            // a multiple of 1/16: exact in bf16 (lo is zero; kept for symmetry)
          const unsigned hb = __float_as_uint(wv) & 0xffff0000u;
          hi2[u] = hb >> 16;
          lo2[u] = __float_as_uint(wv - __uint_as_float(hb)) >> 16;
        }
        whi[ty][tx][e2] = hi2[0] | (hi2[1] << 16);
        wlo[ty][tx][e2] = lo2[0] | (lo2[1] << 16);
      }
  }
  (void)wlo;

  unsigned char* out_lds = smem;
  float* st_lds = reinterpret_cast<float*>(smem + MAIN);
  float b4[4][4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int e = 0; e < 4; ++e) b4[cb][e] = bias ? bias[cchunk * 64 + 16 * cb + 4 * g + e] : 0.f;
  const int ch = cchunk * 64 + (tid & 7) * 8;
  float sh[8];
#pragma unroll
  for (int v = 0; v < 8; ++v) sh[v] = (stat && bias) ? bias[ch + v] : 0.f;
  const bool yedge = py0 == 0 || py0 + SEP_TH >= H;               // block-uniform

  const int t_begin = xs * tilesPerWg;
  const int t_end = t_begin + tilesPerWg < tilesX ? t_begin + tilesPerWg : tilesX;
  for (int tx_ = t_begin; tx_ < t_end; ++tx_) {
    const int px0 = tx_ * SEP_TW;
    const int col_lo = px0 / K - 1;
    unsigned src[G::JN];
#pragma unroll
    for (int j = 0; j < G::JN; ++j) {
      const int s = (wave + 4 * j) * 64 + lane;
      const int q = s / G::QS, c = s - q * G::QS;
      const int qr = q / G::CW, qc = q - qr * G::CW;
      int row = base_lo + qr, col = col_lo + qc;
      row = row < 0 ? 0 : (row > h - 1 ? h - 1 : row);
      col = col < 0 ? 0 : (col > w - 1 ? w - 1 : col);
      const int tx = c >> 3, jc = c & 7;
      const bool real = s < G::NS && c < 24;
      src[j] = real ? (unsigned)((((int64_t)row * w + col) * ldz + (int64_t)tx * Co + jc * 8) * 2) : 0u;
    }
    const bool edge = yedge || px0 == 0 || px0 + SEP_TW >= W;     // block-uniform: some taps fall outside the image
    // staged column of this lane's transposing-read row (source column c = tq) per (jj, tx), as a byte offset
    int acol[2][3];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        int col = c0[tx] + 8 * jj + 1 + tq;
        col = col > G::CW - 1 ? G::CW - 1 : col;                  // (only entries with a zero weight get clamped)
        acol[jj][tx] = col * PIXB + tx * 128 + tp * 8;
      }

    up_f32x4 acc[2][2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) acc[i][jj][cb] = up_f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int ty = 0; ty < 3; ++ty) {
      __syncthreads();                                            // every wave is done with the previous stage / strip
      const unsigned char* zt = zimg + (int64_t)ty * 3 * Co * 2;
#pragma unroll
      for (int j = 0; j < G::JN; ++j)
        if (wave + 4 * j < G::NI) up_glds16(zt + src[j], smem + (wave + 4 * j) * 1024);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        constexpr int dummy = 0; (void)dummy;
        int rr = up_fl<2>(4 * i + ty - 1) - G::FLMIN + g;         // staged row of this lane's contraction row
        rr = rr > G::R - 1 ? G::R - 1 : rr;                       // (only entries with a zero weight get clamped)
        const int rowb = rr * (G::CW * PIXB);
        const int qy = py0 + 4 * i + (m >> 2) + ty - 1;
        const bool rowok = qy >= 0 && qy < H;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
          for (int tx = 0; tx < 3; ++tx) {
            unsigned h2[2] = {whi[ty][tx][0], whi[ty][tx][1]};
            if (edge) {
              const int qx = px0 + 16 * jj + 4 * wave + (m & 3) + tx - 1;
              const bool ok = rowok && qx >= 0 && qx < W;
              h2[0] = ok ? h2[0] : 0u;
              h2[1] = ok ? h2[1] : 0u;
            }
            up_short4 wh;
            wh[0] = (short)(h2[0] & 0xffff); wh[1] = (short)(h2[0] >> 16); wh[2] = (short)(h2[1] & 0xffff); wh[3] = (short)(h2[1] >> 16);
            const unsigned char* p0 = smem + rowb + acol[jj][tx];
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) {
              const up_short4 zf = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) up_short4*)(p0 + cb * 32));
              acc[i][jj][cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(zf, wh, acc[i][jj][cb], 0, 0, 0);
            }
          }
      }
    }
    __syncthreads();                                              // the tile reuses the stage buffer
    up_mfma_epilogue(acc, b4, sh, out_lds, st_lds, y, ldy, stat, n, H, W, py0, px0, Co, cchunk, tilesX, tx_);
  }
}

__global__ __launch_bounds__(256, 2) void upconv_fwd_mfma2_kernel(const bf16_t* __restrict__ Z, int64_t ldz,
                                                                 bf16_t* __restrict__ y, int64_t ldy,
                                                                 const float* __restrict__ bias, float* __restrict__ stat,
                                                                 int h, int w, int Co, int tilesX, int tilesPerWg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char up_smem[];
  upconv_mfma2_body(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, tilesPerWg, up_smem);
}

template <int K>
__global__ __launch_bounds__(256, 2) void upconv_fwd_mfma_kernel(const bf16_t* __restrict__ Z, int64_t ldz,
                                                                bf16_t* __restrict__ y, int64_t ldy,
                                                                const float* __restrict__ bias, float* __restrict__ stat,
                                                                int h, int w, int Co, int tilesX, int tilesPerWg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char up_smem[];
  if constexpr (K == 16) {
    if (blockIdx.y & 1) upconv_mfma_body<K, 1>(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, tilesPerWg, up_smem);
    else upconv_mfma_body<K, 0>(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, tilesPerWg, up_smem);
  } else {
    upconv_mfma_body<K, 0>(Z, ldz, y, ldy, bias, stat, h, w, Co, tilesX, tilesPerWg, up_smem);
  }
}

template <int K>
int launch_fwd_mfma_k(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, float* stat, int N, int h, int w,
                      int Co, hipStream_t st) {
  constexpr int OUTB = SEP_TH * SEP_TW * 144;
  constexpr int LDS2 = (UpStage<2, 0>::LDS_BYTES > OUTB ? UpStage<2, 0>::LDS_BYTES : OUTB) + 4 * 64 * 2 * 4;
  constexpr int LDS = K == 2 ? LDS2 : UpMfma<K>::LDS_BYTES;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(K == 2 ? reinterpret_cast<const void*>(&upconv_fwd_mfma2_kernel)
                                              : reinterpret_cast<const void*>(&upconv_fwd_mfma_kernel<K>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("upconv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  const int H = h * K, W = w * K;
  const int tilesX = cdiv(W, SEP_TW), tilesY = cdiv(H, SEP_TH);
  // a workgroup walks a run of strips along x (its weights stay in registers); short runs when the map is small
  const int64_t rows = (int64_t)(Co / 64) * tilesY * N;
  int xsplits = (int)cdiv64(2048, rows);
  if (xsplits > tilesX) xsplits = tilesX;
  if (xsplits < 1) xsplits = 1;
  const int tilesPerWg = cdiv(tilesX, xsplits);
  xsplits = cdiv(tilesX, tilesPerWg);
  dim3 grid((unsigned)((Co / 64) * xsplits), (unsigned)tilesY, (unsigned)N);
  if (K == 2)
    upconv_fwd_mfma2_kernel<<<grid, 256, LDS, st>>>((const bf16_t*)Z, ldz, (bf16_t*)y, ldy, bias, stat, h, w, Co, tilesX, tilesPerWg);
  else
    upconv_fwd_mfma_kernel<K><<<grid, 256, LDS, st>>>((const bf16_t*)Z, ldz, (bf16_t*)y, ldy, bias, stat, h, w, Co, tilesX, tilesPerWg);
  return SNWS_OK;
}

int launch_fwd_mfma(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, float* stat, int N, int h, int w,
                    int Co, int k, hipStream_t st) {
  switch (k) {
    case 2: return launch_fwd_mfma_k<2>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
    case 4: return launch_fwd_mfma_k<4>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
    case 8: return launch_fwd_mfma_k<8>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
    default: return launch_fwd_mfma_k<16>(Z, ldz, y, ldy, bias, stat, N, h, w, Co, st);
  }
}

// The matrix forms win for every k on the decoder's shapes: 155 / 134 / 163 / 197 us against 197 / 194 / 238 / 344 us per
// launch for k = 2 / 4 / 8 / 16.  SNWS_UPCONV_MFMA=0 (or snws_upconv_fwd_mode(0)) selects the vector-ALU form.
std::atomic<int> g_upconv_mode{-1};
bool upconv_mfma_on(int k) {
  int mode = g_upconv_mode.load(std::memory_order_relaxed);
  if (mode < 0) {
    mode = getenv("SNWS_UPCONV_MFMA") ? atoi(getenv("SNWS_UPCONV_MFMA")) : 1;
    g_upconv_mode.store(mode, std::memory_order_relaxed);
  }
  (void)k;
  return mode != 0;
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

// ---- separable backward (bf16, 3x3, k in {2,4,8,16}, channel counts that are multiples of 8).
// dZ_{ty,tx}(sy,sx) = sum_Y wy(Y+ty-1 -> sy) * Hrow_tx(Y),   Hrow_tx(Y) = sum_X wx(X+tx-1 -> sx) * dy(Y, X):
// the column weights are the same for every row of the window, so a thread keeps them in registers, reduces each
// of the 2k+2 window rows to three 8-channel vectors (24 FMAs per loaded value instead of 72) and folds those into
// the nine tap accumulators.  For k >= 4 the 2k+2 window columns are split over PARTS adjacent lanes; their partial
// row sums are combined with a butterfly reduce-scatter that leaves lane p with 8/PARTS of the channels, for which
// it then keeps all nine taps - so no work is duplicated and a thread never walks more than 6 columns per row.
template <int K, int PARTS>
__global__ __launch_bounds__(256) void upconv_bwd_sep_kernel(const bf16_t* __restrict__ dy, int64_t lddy,
                                                             bf16_t* __restrict__ dZ, int64_t lddz, int N, int h, int w,
                                                             int Co, int64_t total) {
  constexpr int WIN = 2 * K + 2, CH = (WIN + PARTS - 1) / PARTS, CPP = 8 / PARTS;   // channels per part at the end
  const int H = h * K, W = w * K, CG = Co / 8;
  const float inv_k = 1.f / (float)K;
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool live = idx < total;
  if (!live) idx = total - 1;                       // keep every lane in the shuffles; it just does not store
  const int part = (int)(idx % PARTS);
  int64_t r = idx / PARTS;
  const int cg = (int)(r % CG); r /= CG;
  const int sx = (int)(r % w); r /= w;
  const int sy = (int)(r % h);
  const int n = (int)(r / h);
  const int X0 = K * sx - K / 2 - 1 + part * CH, Y0 = K * sy - K / 2 - 1;
  // column weights of two neighbouring window columns, packed as bf16 pairs (multiples of 1/(2K): exact)
  constexpr int CP = (CH + 1) / 2;
  unsigned wxp[3][CP];
#pragma unroll
  for (int jp = 0; jp < CP; ++jp) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      float wv[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = 2 * jp + u, X = X0 + j;
        const bool in = j < CH && X >= 0 && X < W && part * CH + j < WIN;
        wv[u] = in ? tap_weight(X + tx - 1, W, inv_k, w, sx) : 0.f;
      }
      wxp[tx][jp] = pack_bf16x2(wv[0], wv[1]);
    }
  }
  float acc[9][CPP];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int v = 0; v < CPP; ++v) acc[t][v] = 0.f;
  const bf16_t* gn = dy + (int64_t)n * H * W * lddy + (int64_t)cg * 8;
  // the window rows inside the image, software-pipelined: row y+1 is in flight while row y is reduced
  const int ya = Y0 < 0 ? 0 : Y0, yb = Y0 + WIN - 1 > H - 1 ? H - 1 : Y0 + WIN - 1;
  bool cok[2 * CP];
#pragma unroll
  for (int j = 0; j < 2 * CP; ++j) cok[j] = j < CH && X0 + j >= 0 && X0 + j < W && part * CH + j < WIN;
  uint4 cur[2 * CP], nxt[2 * CP];
  auto load_row = [&](int Y, uint4 (&dst)[2 * CP]) {
    const bf16_t* row = gn + (int64_t)Y * W * lddy;
#pragma unroll
    for (int j = 0; j < 2 * CP; ++j) {
      dst[j] = make_uint4(0u, 0u, 0u, 0u);       // a column outside the window or the image counts as zeros
      if (cok[j]) dst[j] = *reinterpret_cast<const uint4*>(row + (int64_t)(X0 + j) * lddy);
    }
  };
  if (ya <= yb) load_row(ya, cur);
  for (int Y = ya; Y <= yb; ++Y) {
    if (Y < yb) load_row(Y + 1, nxt);
    float wy[3];
#pragma unroll
    for (int ty = 0; ty < 3; ++ty) wy[ty] = tap_weight(Y + ty - 1, H, inv_k, h, sy);
    float hv[24];                                   // index v*3 + tx
#pragma unroll
    for (int i = 0; i < 24; ++i) hv[i] = 0.f;
#pragma unroll
    for (int jp = 0; jp < CP; ++jp) {
      const unsigned ua[4] = {cur[2 * jp].x, cur[2 * jp].y, cur[2 * jp].z, cur[2 * jp].w};
      const unsigned ub[4] = {cur[2 * jp + 1].x, cur[2 * jp + 1].y, cur[2 * jp + 1].z, cur[2 * jp + 1].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned lo = pair_lo(ua[e], ub[e]), hi = pair_hi(ua[e], ub[e]);   // one channel of both columns
#pragma unroll
        for (int tx = 0; tx < 3; ++tx) {
          hv[(2 * e) * 3 + tx] = dot2_bf16(lo, wxp[tx][jp], hv[(2 * e) * 3 + tx]);
          hv[(2 * e + 1) * 3 + tx] = dot2_bf16(hi, wxp[tx][jp], hv[(2 * e + 1) * 3 + tx]);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 2 * CP; ++j) cur[j] = nxt[j];
    // butterfly reduce-scatter over the PARTS lanes of this source pixel: lane p ends with channels p*CPP ...
#pragma unroll
    for (int o = PARTS / 2, nv = 24; o >= 1; o >>= 1, nv >>= 1) {
      const bool up = (part & o) != 0;
#pragma unroll
      for (int i = 0; i < nv / 2; ++i) {
        const float lo = hv[i], hi = hv[i + nv / 2];
        const float recv = __shfl_xor(up ? lo : hi, o, 64);
        hv[i] = (up ? hi : lo) + recv;
      }
    }
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx)
#pragma unroll
        for (int v = 0; v < CPP; ++v) acc[ty * 3 + tx][v] = fmaf(wy[ty], hv[v * 3 + tx], acc[ty * 3 + tx][v]);
  }
  if (!live) return;
  bf16_t* o = dZ + (((int64_t)n * h + sy) * w + sx) * lddz + (int64_t)cg * 8 + part * CPP;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    if constexpr (CPP == 8) {
      Vec<bf16_t, 8>::store(o + (int64_t)t * Co, acc[t]);
    } else if constexpr (CPP == 1) {
      reinterpret_cast<unsigned short*>(o + (int64_t)t * Co)[0] = f32_to_bf16_bits(acc[t][0]);
    } else {
#pragma unroll
      for (int v = 0; v < CPP; v += 2)
        *reinterpret_cast<unsigned*>(o + (int64_t)t * Co + v) = pack_bf16x2(acc[t][v], acc[t][v + 1]);
    }
  }
}

// ---- backward for k = 2 on the matrix cores (bf16, 3x3, 64-channel chunks).
// dZ_t(s) = sum_q w(q -> s) dy(q - t + 1): for a 4x4 block of source pixels the positions q with a non-zero weight form a
// 10x10 window, the same for every tap - only the dy pixel read for a position shifts with the tap.  Per tap that is
//   D[64 channels][16 source pixels] = dyrows[64][160] * Wt[160][16],   contraction entry = (window row, window column
//   padded to 16), read with the transposing LDS read at (row - ty, column - tx) of the staged dy window,
// with ONE weight matrix (products of two interpolation weights: multiples of 1/16, exact in bf16) held in registers.
// The weights come from tap_weight(), i.e. with the interpolation's border clamp and the convolution's zero padding, so
// border blocks need no special path.  A workgroup stages the 12x36 dy window of a 4x16 source tile (8x32 outputs + 2),
// wave w owns source columns 4w..4w+3; 144-byte pixel stride keeps the transposing reads conflict-free; a tap's result
// leaves through a wave-private 16x64 scratch as whole 128-byte lines.  45 matrix instructions per tap and wave replace
// ~66 VALU operations per output element of the separable kernel below.
struct UpB2 {
  static constexpr int TR = 12, TC = 36;
  static constexpr int SL = 9, PIXB = SL * 16;
  static constexpr int NS = TR * TC * SL, NI = (NS + 63) / 64, JN = (NI + 3) / 4;
  static constexpr int STAGE_BYTES = NI * 1024;
  static constexpr int SCR_RS = 144, SCR = 16 * SCR_RS;
  static constexpr int LDS_BYTES = STAGE_BYTES + 4 * SCR;
  static constexpr int KS = 5;
};

__global__ __launch_bounds__(256, 2) void upconv_bwd_mfma2_kernel(const bf16_t* __restrict__ dy, int64_t lddy,
                                                                 bf16_t* __restrict__ dZ, int64_t lddz, int h, int w,
                                                                 int Co, int tilesX, int tilesPerWg, int accumulate) {
  // accumulate != 0: dZ += result (a tap map consumed at several decoder scales: the later consumers add in place)
  using G = UpB2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunk = Co / 64;
  const int cchunk = blockIdx.x % nchunk, xs = blockIdx.x / nchunk;
  const int sy0 = blockIdx.y * 4, n = blockIdx.z;
  const int H = 2 * h, W = 2 * w;
  const int m = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int sr = m >> 2, sc = m & 3;
  unsigned char* scr = smem + G::STAGE_BYTES + wave * G::SCR;
  const unsigned char* dimg = reinterpret_cast<const unsigned char*>(dy + (int64_t)n * H * W * lddy + cchunk * 64);

  // row weights of this lane's source row for the window rows 2 ks + (g >> 1)
  float wyv[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) wyv[ks] = tap_weight(2 * sy0 - 1 + 2 * ks + (g >> 1), H, 0.5f, h, sy0 + sr);
  // transposing-read rows of the dy operand: contraction entry 32 ks + 8 g + 4 hq + tq = (window row 2 ks + (g >> 1),
  // window column 8 (g & 1) + 4 hq + tq; padding columns >= 10 carry zero weights and read a real pixel)
  int abase[2];
#pragma unroll
  for (int hq = 0; hq < 2; ++hq) {
    int qc = 8 * (g & 1) + 4 * hq + tq;
    qc = qc > 9 ? 9 : qc;
    abase[hq] = (((g >> 1) + 2) * G::TC + 8 * wave + qc + 2) * G::PIXB + tp * 8;
  }

  const int t_begin = xs * tilesPerWg;
  const int t_end = t_begin + tilesPerWg < tilesX ? t_begin + tilesPerWg : tilesX;
  for (int tx_ = t_begin; tx_ < t_end; ++tx_) {
    const int sx0 = tx_ * 16, sxb = sx0 + 4 * wave;
    __syncthreads();                                              // every wave is done reading the previous window
#pragma unroll
    for (int j = 0; j < G::JN; ++j) {
      const int piece = wave + 4 * j;
      if (piece < G::NI) {
        const int s = piece * 64 + lane;
        const int pix = s / G::SL, slot = s - pix * G::SL;
        const int r = pix / G::TC, c = pix - r * G::TC;
        const int py = 2 * sy0 - 2 + r, px = 2 * sx0 - 2 + c;
        const bool ok = s < G::NS && slot < 8 && py >= 0 && py < H && px >= 0 && px < W;
        const void* src = ok ? (const void*)(dimg + (((int64_t)py * W + px) * lddy + slot * 8) * 2) : (const void*)up_zero_page;
        up_glds16(src, smem + piece * 1024);
      }
    }
    // column weights of this lane's source column for the window columns 8 (g & 1) + e, and the packed products
    unsigned wf[G::KS][4];
    {
      float wxv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int qc = 8 * (g & 1) + e;
        wxv[e] = qc < 10 ? tap_weight(2 * sxb - 1 + qc, W, 0.5f, w, sxb + sc) : 0.f;
      }
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks)
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)      // multiples of 1/16 (sums of them at a clamped border): exact in bf16
          wf[ks][e2] = (__float_as_uint(wyv[ks] * wxv[2 * e2]) >> 16) | (__float_as_uint(wyv[ks] * wxv[2 * e2 + 1]) & 0xffff0000u);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ty = tap / 3, tx = tap % 3;
      const int toff = -(ty * G::TC + tx) * G::PIXB;
      up_f32x4 acc[4];
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) acc[cb] = up_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const up_bf16x8 wv = __builtin_bit_cast(up_bf16x8, make_uint4(wf[ks][0], wf[ks][1], wf[ks][2], wf[ks][3]));
        const unsigned char* p0 = smem + abase[0] + ks * (2 * G::TC * G::PIXB) + toff;
        const unsigned char* p1 = smem + abase[1] + ks * (2 * G::TC * G::PIXB) + toff;
#pragma unroll
        for (int cb = 0; cb < 4; ++cb)
          acc[cb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(up_tr_frag(p0 + cb * 32, p1 + cb * 32), wv, acc[cb], 0, 0, 0);
      }
      // D[channel 16 cb + 4 g + j][source pixel m] -> scratch -> one 128-byte line per (source pixel, tap)
#pragma unroll
      for (int cb = 0; cb < 4; ++cb)
        *reinterpret_cast<uint2*>(scr + m * G::SCR_RS + (16 * cb + 4 * g) * 2) =
            make_uint2(pack_bf16x2(acc[cb][0], acc[cb][1]), pack_bf16x2(acc[cb][2], acc[cb][3]));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        const int q = it * 64 + lane, p16 = q >> 3, c8 = q & 7;
        const int sy = sy0 + (p16 >> 2), sx = sxb + (p16 & 3);
        if (sy < h && sx < w) {
          uint4* dst = reinterpret_cast<uint4*>(dZ + (((int64_t)n * h + sy) * w + sx) * lddz + (int64_t)tap * Co + cchunk * 64 + c8 * 8);
          uint4 v = *reinterpret_cast<const uint4*>(scr + p16 * G::SCR_RS + c8 * 16);
          if (accumulate) v = add_bf16x8(v, *dst);      // kernel-uniform
          *dst = v;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
}

// ---- the same for k = 4: a 4x4 source block sees a 20x20 window of positions (contraction = 20 window rows x 32
// columns, 20 used), its dy window is 22x22 pixels (70 KB staged), so a workgroup owns ONE block and its four waves split
// the 64 channels; weights are multiples of 1/64 (exact in bf16).  A wave stores its 16 channels of a (pixel, tap) line
// as 32-byte segments; the four waves of the workgroup complete the 128-byte lines within the same tap.
struct UpB4 {
  static constexpr int TR = 22, TC = 22;
  static constexpr int SL = 9, PIXB = SL * 16;
  static constexpr int NS = TR * TC * SL, NI = (NS + 63) / 64, JN = (NI + 3) / 4;
  static constexpr int LDS_BYTES = NI * 1024;
  static constexpr int KS = 20;
};

__global__ __launch_bounds__(256, 2) void upconv_bwd_mfma4_kernel(const bf16_t* __restrict__ dy, int64_t lddy,
                                                                 bf16_t* __restrict__ dZ, int64_t lddz, int h, int w,
                                                                 int Co, int tilesX, int tilesPerWg) {
  using G = UpB4;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nchunk = Co / 64;
  const int cchunk = blockIdx.x % nchunk, xs = blockIdx.x / nchunk;
  const int sy0 = blockIdx.y * 4, n = blockIdx.z;
  const int H = 4 * h, W = 4 * w;
  const int m = lane & 15, g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const int sr = m >> 2, sc = m & 3;
  const unsigned char* dimg = reinterpret_cast<const unsigned char*>(dy + (int64_t)n * H * W * lddy + cchunk * 64);

  // row weights of this lane's source row for the 20 window rows (k-step ks = window row ks)
  float wyv[G::KS];
#pragma unroll
  for (int ks = 0; ks < G::KS; ++ks) wyv[ks] = tap_weight(4 * sy0 - 2 + ks, H, 0.25f, h, sy0 + sr);
  // transposing-read rows of the dy operand: entry 32 ks + 8 g + 4 hq + tq = (window row ks, window column 8 g + 4 hq + tq)
  int abase[2];
#pragma unroll
  for (int hq = 0; hq < 2; ++hq) {
    int qc = 8 * g + 4 * hq + tq;
    qc = qc > 19 ? 19 : qc;                   // padding columns carry zero weights and read a real pixel
    abase[hq] = (2 * G::TC + qc + 2) * G::PIXB + wave * 32 + tp * 8;
  }

  const int t_begin = xs * tilesPerWg;
  const int t_end = t_begin + tilesPerWg < tilesX ? t_begin + tilesPerWg : tilesX;
  for (int tx_ = t_begin; tx_ < t_end; ++tx_) {
    const int sx0 = tx_ * 4;
    __syncthreads();                                              // every wave is done reading the previous window
#pragma unroll
    for (int j = 0; j < G::JN; ++j) {
      const int piece = wave + 4 * j;
      if (piece < G::NI) {
        const int s = piece * 64 + lane;
        const int pix = s / G::SL, slot = s - pix * G::SL;
        const int r = pix / G::TC, c = pix - r * G::TC;
        const int py = 4 * sy0 - 3 + r, px = 4 * sx0 - 3 + c;
        const bool ok = s < G::NS && slot < 8 && py >= 0 && py < H && px >= 0 && px < W;
        const void* src = ok ? (const void*)(dimg + (((int64_t)py * W + px) * lddy + slot * 8) * 2) : (const void*)up_zero_page;
        up_glds16(src, smem + piece * 1024);
      }
    }
    float wxv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int qc = 8 * g + e;
      wxv[e] = qc < 20 ? tap_weight(4 * sx0 - 2 + qc, W, 0.25f, w, sx0 + sc) : 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int sy = sy0 + sr, sx = sx0 + sc;
    bf16_t* op = dZ + (((int64_t)n * h + sy) * w + sx) * lddz + cchunk * 64 + 16 * wave + 4 * g;
    const bool pix_ok = sy < h && sx < w;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int ty = tap / 3, tx = tap % 3;
      const int toff = -(ty * G::TC + tx) * G::PIXB;
      up_f32x4 acc = up_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        unsigned wf[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2)        // multiples of 1/64 (sums of them at a clamped border): exact in bf16
          wf[e2] = (__float_as_uint(wyv[ks] * wxv[2 * e2]) >> 16) | (__float_as_uint(wyv[ks] * wxv[2 * e2 + 1]) & 0xffff0000u);
        const up_bf16x8 wv = __builtin_bit_cast(up_bf16x8, make_uint4(wf[0], wf[1], wf[2], wf[3]));
        const unsigned char* p0 = smem + abase[0] + ks * (G::TC * G::PIXB) + toff;
        const unsigned char* p1 = smem + abase[1] + ks * (G::TC * G::PIXB) + toff;
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(up_tr_frag(p0, p1), wv, acc, 0, 0, 0);
      }
      // D[channel 16 wave + 4 g + j][source pixel m]
      if (pix_ok)
        *reinterpret_cast<uint2*>(op + (int64_t)tap * Co) = make_uint2(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]));
    }
  }
}

int launch_bwd_mfma4(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int N, int h, int w, int Co, hipStream_t st) {
  using G = UpB4;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_bwd_mfma4_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      snws_set_error("upconv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  const int tilesX = cdiv(w, 4), tilesY = cdiv(h, 4);
  const int64_t rows = (int64_t)(Co / 64) * tilesY * N;
  int xsplits = (int)cdiv64(2048, rows);
  if (xsplits > tilesX) xsplits = tilesX;
  if (xsplits < 1) xsplits = 1;
  const int tilesPerWg = cdiv(tilesX, xsplits);
  xsplits = cdiv(tilesX, tilesPerWg);
  dim3 grid((unsigned)((Co / 64) * xsplits), (unsigned)tilesY, (unsigned)N);
  upconv_bwd_mfma4_kernel<<<grid, 256, G::LDS_BYTES, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dZ, lddz, h, w, Co, tilesX, tilesPerWg);
  return SNWS_OK;
}

int launch_bwd_mfma2(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int N, int h, int w, int Co, hipStream_t st,
                     int accumulate = 0) {
  using G = UpB2;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&upconv_bwd_mfma2_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      snws_set_error("upconv: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  const int tilesX = cdiv(w, 16), tilesY = cdiv(h, 4);
  const int64_t rows = (int64_t)(Co / 64) * tilesY * N;
  int xsplits = (int)cdiv64(2048, rows);
  if (xsplits > tilesX) xsplits = tilesX;
  if (xsplits < 1) xsplits = 1;
  const int tilesPerWg = cdiv(tilesX, xsplits);
  xsplits = cdiv(tilesX, tilesPerWg);
  dim3 grid((unsigned)((Co / 64) * xsplits), (unsigned)tilesY, (unsigned)N);
  upconv_bwd_mfma2_kernel<<<grid, 256, G::LDS_BYTES, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dZ, lddz, h, w, Co, tilesX, tilesPerWg,
                                                           accumulate);
  return SNWS_OK;
}

std::atomic<int> g_upconv_bwd_mode{-1};
bool upconv_bwd_mfma_on() {
  int mode = g_upconv_bwd_mode.load(std::memory_order_relaxed);
  if (mode < 0) {
    mode = getenv("SNWS_UPCONV_BWD_MFMA") ? atoi(getenv("SNWS_UPCONV_BWD_MFMA")) : 1;
    g_upconv_bwd_mode.store(mode, std::memory_order_relaxed);
  }
  return mode != 0;
}

template <int K, int PARTS>
void launch_bwd_sep(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int N, int h, int w, int Co, hipStream_t st) {
  const int64_t total = (int64_t)N * h * w * (Co / 8) * PARTS;
  upconv_bwd_sep_kernel<K, PARTS><<<(unsigned)cdiv64(total, 256), 256, 0, st>>>((const bf16_t*)dy, lddy, (bf16_t*)dZ, lddz,
                                                                                N, h, w, Co, total);
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
  static const bool no_lds = getenv("SNWS_UPCONV_NOLDS") != nullptr;    // A/B switch: the kernels without LDS staging
  if (!no_sep && !no_lds && lds_fwd_ok(dtype, Z, ldz, y, ldy, N, h, w, C_out, k, ksize)) {
    if (int rc = upconv_mfma_on(k) ? launch_fwd_mfma(Z, ldz, y, ldy, bias, nullptr, N, h, w, C_out, k, st)
                                  : launch_fwd_lds(Z, ldz, y, ldy, bias, nullptr, N, h, w, C_out, k, st)) return rc;
    SNWS_LAUNCH_CHECK("snws_upconv_fwd");
    return SNWS_OK;
  }
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

extern "C" int snws_upconv_bwd_mode(int mode) {
  upconv_bwd_mfma_on();                 // (reads the environment on first use)
  const int prev = g_upconv_bwd_mode.load(std::memory_order_relaxed);
  if (mode >= 0 && mode <= 1) g_upconv_bwd_mode.store(mode, std::memory_order_relaxed);
  return prev;
}

extern "C" int snws_upconv_fwd_mode(int mode) {
  upconv_mfma_on(2);                    // (reads the environment on first use)
  const int prev = g_upconv_mode.load(std::memory_order_relaxed);
  if (mode >= 0 && mode <= 2) g_upconv_mode.store(mode, std::memory_order_relaxed);
  return prev;
}

extern "C" int snws_upconv_stat_tiles(int H, int W) { return cdiv(H, SEP_TH) * cdiv(W, SEP_TW); }

extern "C" int snws_upconv_fwd_stats(const void* Z, int64_t ldz, void* y, int64_t ldy, const float* bias, int dtype,
                                     int N, int h, int w, int C_out, int k, int ksize, float* stat_part, void* stream) {
  if (int rc = check("snws_upconv_fwd_stats", dtype, N, h, w, C_out, k, ksize)) return rc;
  SNWS_CHECK_ARG(Z && y && stat_part && ldz >= (int64_t)ksize * ksize * C_out && ldy >= C_out,
                 "snws_upconv_fwd_stats: NULL pointer or bad stride");
  if (!lds_fwd_ok(dtype, Z, ldz, y, ldy, N, h, w, C_out, k, ksize))
    SNWS_UNSUPPORTED("snws_upconv_fwd_stats: needs bf16, a 3x3 filter, k in {2,4,8,16}, C_out a multiple of 64 and "
                     "16-byte aligned tensors (got dtype %d, ksize %d, k %d, C_out %d)", dtype, ksize, k, C_out);
  if (int rc = upconv_mfma_on(k) ? launch_fwd_mfma(Z, ldz, y, ldy, bias, stat_part, N, h, w, C_out, k, (hipStream_t)stream)
                                : launch_fwd_lds(Z, ldz, y, ldy, bias, stat_part, N, h, w, C_out, k, (hipStream_t)stream)) return rc;
  SNWS_LAUNCH_CHECK("snws_upconv_fwd_stats");
  return SNWS_OK;
}

extern "C" int snws_upconv_bwd_acc(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int dtype, int N, int h, int w,
                                   int C_out, int k, int ksize, void* stream) {
  if (int rc = check("snws_upconv_bwd_acc", dtype, N, h, w, C_out, k, ksize)) return rc;
  SNWS_CHECK_ARG(dy && dZ && lddz >= (int64_t)ksize * ksize * C_out && lddy >= C_out, "snws_upconv_bwd_acc: NULL pointer or bad stride");
  if (!(dtype == SNWS_BF16 && ksize == 3 && k == 2 && C_out % 64 == 0 && lddz % 8 == 0 && lddy % 8 == 0 && al16(dZ) && al16(dy) &&
        upconv_bwd_mfma_on() && N <= 65535 && cdiv(h, 4) <= 65535))
    SNWS_UNSUPPORTED("snws_upconv_bwd_acc: only the k = 2 matrix-core kernel adds in place (bf16, 3x3, C_out a multiple of 64)");
  if (int rc = launch_bwd_mfma2(dy, lddy, dZ, lddz, N, h, w, C_out, (hipStream_t)stream, 1)) return rc;
  SNWS_LAUNCH_CHECK("snws_upconv_bwd_acc");
  return SNWS_OK;
}

extern "C" int snws_upconv_bwd(const void* dy, int64_t lddy, void* dZ, int64_t lddz, int dtype, int N, int h, int w,
                               int C_out, int k, int ksize, void* stream) {
  if (int rc = check("snws_upconv_bwd", dtype, N, h, w, C_out, k, ksize)) return rc;
  SNWS_CHECK_ARG(dy && dZ && lddz >= (int64_t)ksize * ksize * C_out && lddy >= C_out, "snws_upconv_bwd: NULL pointer or bad stride");
  hipStream_t st = (hipStream_t)stream;
  const int64_t items = (int64_t)N * h * w;
  static const bool no_sep_bwd = getenv("SNWS_UPCONV_BWD_GATHER") != nullptr;   // A/B switch: the non-separable kernel
  if (!no_sep_bwd && dtype == SNWS_BF16 && ksize == 3 && C_out % 8 == 0 && lddz % 8 == 0 && lddy % 8 == 0 && al16(dZ) &&
      al16(dy) && (k == 2 || k == 4 || k == 8 || k == 16) && cdiv64(items * (C_out / 8) * 8, 256) < (1ll << 31)) {
    if ((k == 2 || k == 4) && C_out % 64 == 0 && upconv_bwd_mfma_on() && N <= 65535 && cdiv(h, 4) <= 65535) {
      if (int rc = k == 2 ? launch_bwd_mfma2(dy, lddy, dZ, lddz, N, h, w, C_out, st)
                          : launch_bwd_mfma4(dy, lddy, dZ, lddz, N, h, w, C_out, st)) return rc;
      SNWS_LAUNCH_CHECK("snws_upconv_bwd");
      return SNWS_OK;
    }
    switch (k) {
      case 2: launch_bwd_sep<2, 1>(dy, lddy, dZ, lddz, N, h, w, C_out, st); break;
      case 4: launch_bwd_sep<4, 2>(dy, lddy, dZ, lddz, N, h, w, C_out, st); break;
      case 8: launch_bwd_sep<8, 4>(dy, lddy, dZ, lddz, N, h, w, C_out, st); break;
      default: launch_bwd_sep<16, 8>(dy, lddy, dZ, lddz, N, h, w, C_out, st); break;
    }
    SNWS_LAUNCH_CHECK("snws_upconv_bwd");
    return SNWS_OK;
  }
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
