// float32 implicit-GEMM 3x3 / 1x1 convolution on the gfx950 f32 matrix instruction (v_mfma_f32_32x32x2_f32: f32 in,
// f32 accumulate, exactly an fmaf chain) - the float32 tier of F.conv2d at modeling/weightstd.py:18 of the reference
// and its autograd, for stride-1 "same" convolutions.  NHWC f32 activations; the standardized filter is read in the
// (O, kh, kw, I) f32 layout snws_ws_fwd writes (forward), its flipped transpose (I, kh, kw, O) (dgrad).
//
// The f32 matrix rate is 1/16 of the bf16 one (64 FLOP/clk/SIMD, 157 TFLOP/s), so operand delivery is cheap next to
// the matrix work and the kernels stay simple: operands staged global -> registers -> LDS (padded rows of 20 floats:
// conflict-free 16-byte reads without a swizzle), one 16-byte LDS read per operand feeds four MFMAs.
//   forward / dgrad: workgroup = 8x32 pixels x 64 output channels, wave w = tile rows 2w, 2w+1; D[out ch][pixel].
//     The contraction index is permuted so that a lane's 16-byte fragment (4 consecutive channels) is used by four
//     consecutive MFMAs: MFMA s of a k-group takes element s from both operands (lane half h supplies channels
//     8g + 4h + s of the 16-channel chunk).  Epilogue: bias, direct stores, optional per-tile sum / sum of squares
//     per channel for SwitchNorm (from the accumulators: nothing is rounded in this tier).
//   wgrad: D[out ch][in ch] per tap over pixels; workgroup = a 64x64 (out, in) channel tile over a slice of 2x32-pixel
//     tiles, wave = one 32x32 quadrant, all taps; per-workgroup partial tiles + the fixed-order reduction of
//     wgrad_reduce.h (bit-reproducible).
#include "common.h"
#include "wgrad_reduce.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int F_TH = 8, F_TW = 32, F_KC = 16, F_ROW = 20;   // F_ROW: floats per LDS row (16 + 4 pad)

struct F32FwdArgs {
  const float* x;
  const float* w;        // (C_out, taps, C_in)
  const float* bias;
  float* y;
  float* stat;
  int N, H, W, Cin, Cout;
  int64_t ldx, ldy;
  int tilesX, tilesY, coTiles;
};

// 4 consecutive floats starting at p, of which `left` are inside the tensor (left >= 4: one 16-byte load if aligned)
__device__ __forceinline__ float4 load4_guard(const float* p, int left, bool vec_ok) {
  if (left >= 4 && vec_ok) return *reinterpret_cast<const float4*>(p);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (left > 0) v.x = p[0];
  if (left > 1) v.y = p[1];
  if (left > 2) v.z = p[2];
  if (left > 3) v.w = p[3];
  return v;
}

template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_f32_fwd_kernel(F32FwdArgs a) {
  constexpr int PADK = TAPS == 9 ? 1 : 0, HWd = F_TW + 2 * PADK, HHt = F_TH + 2 * PADK, NPIX = HWd * HHt;
  constexpr int KWN = TAPS == 9 ? 3 : 1;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* in_lds = smem_f;                       // [NPIX][F_ROW]
  float* w_lds = smem_f + NPIX * F_ROW;         // [TAPS * 64][F_ROW]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned L = blockIdx.x;
  const int ct = L % a.coTiles; L /= a.coTiles;
  const int tx = L % a.tilesX; L /= a.tilesX;
  const int ty = L % a.tilesY;
  const int n = L / a.tilesY;
  const int y0 = ty * F_TH, x0 = tx * F_TW;
  const int r = lane & 31, h = lane >> 5;
  const bool vx = (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  const bool vw = (a.Cin & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[c2][p][e] = 0.f;

  const float* img = a.x + (int64_t)n * a.H * a.W * a.ldx;
  const int nchunks = (a.Cin + F_KC - 1) / F_KC;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int c0 = ch * F_KC;
    __syncthreads();     // the previous chunk's fragment reads are done
    for (int idx = tid; idx < NPIX * 4; idx += 256) {
      const int pix = idx >> 2, q = idx & 3;
      const int py = pix / HWd, px = pix - py * HWd;
      const int gy = y0 - PADK + py, gx = x0 - PADK + px;
      const int c = c0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.Cin)
        v = load4_guard(img + ((int64_t)gy * a.W + gx) * a.ldx + c, a.Cin - c, vx);
      *reinterpret_cast<float4*>(in_lds + pix * F_ROW + q * 4) = v;
    }
    for (int idx = tid; idx < TAPS * 64 * 4; idx += 256) {
      const int row = idx >> 2, q = idx & 3;
      const int tap = row >> 6, oc = ct * 64 + (row & 63);
      const int c = c0 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oc < a.Cout && c < a.Cin) v = load4_guard(a.w + ((int64_t)oc * TAPS + tap) * a.Cin + c, a.Cin - c, vw);
      *reinterpret_cast<float4*>(w_lds + row * F_ROW + q * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int kh = 0; kh < KWN; ++kh) {
#pragma unroll
      for (int kw = 0; kw < KWN; ++kw) {
        const int tap = kh * KWN + kw;
#pragma unroll
        for (int g = 0; g < 2; ++g) {          // k-group: channels 8g .. 8g+7 of the chunk, lane half h takes 8g + 4h ..
          float4 A[2], B[2];
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
            A[c2] = *reinterpret_cast<const float4*>(w_lds + (tap * 64 + c2 * 32 + r) * F_ROW + (2 * g + h) * 4);
#pragma unroll
          for (int p = 0; p < 2; ++p)
            B[p] = *reinterpret_cast<const float4*>(in_lds + ((2 * wave + p + kh) * HWd + r + kw) * F_ROW + (2 * g + h) * 4);
#pragma unroll
          for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              acc[c2][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c2].x, B[p].x, acc[c2][p], 0, 0, 0);
              acc[c2][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c2].y, B[p].y, acc[c2][p], 0, 0, 0);
              acc[c2][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c2].z, B[p].z, acc[c2][p], 0, 0, 0);
              acc[c2][p] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c2].w, B[p].w, acc[c2][p], 0, 0, 0);
            }
        }
      }
    }
  }

  // ---- epilogue.  D[row = out channel][col = pixel]: lane holds pixel column r and out channels
  //      c2*32 + 8*gq + 4*h + e (register 4*gq + e)
  const int co0 = ct * 64;
  const int gx = x0 + r;
  const bool vy = (a.ldy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.y) & 15) == 0;
  if (a.stat) {
    // per-tile sum / sum of squares of the bias-free outputs per channel: over this lane's valid pixels, then over
    // the 32 pixel columns of the lane half (xor-shuffles below 32 stay inside the half), then over the four waves
    __syncthreads();                                        // in_lds / w_lds are free now
    float* st_lds = smem_f;                                 // [4 waves][64 channels][2]
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int e16 = 0; e16 < 16; ++e16) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int gy = y0 + 2 * wave + p;
          const float v = (gy < a.H && gx < a.W) ? acc[c2][p][e16] : 0.f;
          s += v;
          q += v * v;
        }
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
          s += __shfl_xor(s, o, 64);
          q += __shfl_xor(q, o, 64);
        }
        if (r == 0) {
          const int cl = c2 * 32 + (e16 & 3) + 8 * (e16 >> 2) + 4 * h;
          st_lds[(wave * 64 + cl) * 2] = s;
          st_lds[(wave * 64 + cl) * 2 + 1] = q;
        }
      }
    __syncthreads();
    if (tid < 64 && co0 + tid < a.Cout) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += st_lds[(w * 64 + tid) * 2];
        q += st_lds[(w * 64 + tid) * 2 + 1];
      }
      const int64_t T = (int64_t)a.tilesX * a.tilesY, tile = (int64_t)ty * a.tilesX + tx;
      float* o = a.stat + (((int64_t)n * T + tile) * a.Cout + co0 + tid) * 2;
      o[0] = s;
      o[1] = q;
    }
  }
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const int oc = co0 + c2 * 32 + 8 * gq + 4 * h;
      if (oc >= a.Cout) continue;
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = oc + e < a.Cout ? a.bias[oc + e] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int gy = y0 + 2 * wave + p;
        if (gy >= a.H || gx >= a.W) continue;
        float* yp = a.y + (((int64_t)n * a.H + gy) * a.W + gx) * a.ldy + oc;
        const float v0 = acc[c2][p][4 * gq] + b4[0], v1 = acc[c2][p][4 * gq + 1] + b4[1];
        const float v2 = acc[c2][p][4 * gq + 2] + b4[2], v3 = acc[c2][p][4 * gq + 3] + b4[3];
        if (vy && oc + 3 < a.Cout) {
          *reinterpret_cast<float4*>(yp) = make_float4(v0, v1, v2, v3);
        } else {
          yp[0] = v0;
          if (oc + 1 < a.Cout) yp[1] = v1;
          if (oc + 2 < a.Cout) yp[2] = v2;
          if (oc + 3 < a.Cout) yp[3] = v3;
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------- wgrad
constexpr int FW_TH = 2, FW_TW = 32;

struct F32WgradArgs {
  const float* x;
  const float* dy;
  float* part;
  float* pbias;
  int N, H, W, Cin, Cout, want_bias;
  int64_t ldx, lddy;
  int tilesX, tilesY, coTiles, ciTiles, tilesPerSplit;
};

template <int TAPS>
__global__ __launch_bounds__(256, 2) void conv_f32_wgrad_kernel(F32WgradArgs a) {
  constexpr int PADK = TAPS == 9 ? 1 : 0, HWd = FW_TW + 2 * PADK, HHt = FW_TH + 2 * PADK, NPIX = HWd * HHt;
  constexpr int KN = TAPS == 9 ? 3 : 1, TPIX = FW_TH * FW_TW;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  float* dy_lds = smem_f;                    // [TPIX][64]
  float* x_lds = smem_f + TPIX * 64;         // [NPIX][64]
  float* b_lds = x_lds + NPIX * 64;          // [256]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = a.coTiles * a.ciTiles;
  const int pair = blockIdx.x % npairs, split = blockIdx.x / npairs;
  const int cot = pair % a.coTiles, cit = pair / a.coTiles;
  const int ntiles = a.N * a.tilesY * a.tilesX;
  const int t_begin = split * a.tilesPerSplit;
  const int t_end = t_begin + a.tilesPerSplit < ntiles ? t_begin + a.tilesPerSplit : ntiles;
  const int co_half = wave & 1, ci_half = wave >> 1;
  const int r = lane & 31, h = lane >> 5;
  const bool vx = (a.ldx & 3) == 0 && (reinterpret_cast<uintptr_t>(a.x) & 15) == 0;
  const bool vd = (a.lddy & 3) == 0 && (reinterpret_cast<uintptr_t>(a.dy) & 15) == 0;
  const bool bias = a.want_bias && cit == 0;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float bsum = 0.f;

  for (int t = t_begin; t < t_end; ++t) {
    int tt = t;
    const int tx = tt % a.tilesX; tt /= a.tilesX;
    const int ty = tt % a.tilesY;
    const int n = tt / a.tilesY;
    const int y0 = ty * FW_TH, x0 = tx * FW_TW;
    __syncthreads();
    for (int idx = tid; idx < TPIX * 16; idx += 256) {      // dy tile: 16 float4 per pixel
      const int pix = idx >> 4, q = idx & 15;
      const int gy = y0 + pix / FW_TW, gx = x0 + pix % FW_TW;
      const int c = cot * 64 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy < a.H && gx < a.W && c < a.Cout)
        v = load4_guard(a.dy + (((int64_t)n * a.H + gy) * a.W + gx) * a.lddy + c, a.Cout - c, vd);
      *reinterpret_cast<float4*>(dy_lds + pix * 64 + q * 4) = v;
    }
    for (int idx = tid; idx < NPIX * 16; idx += 256) {      // x halo tile
      const int pix = idx >> 4, q = idx & 15;
      const int py = pix / HWd, px = pix - py * HWd;
      const int gy = y0 - PADK + py, gx = x0 - PADK + px;
      const int c = cit * 64 + q * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && c < a.Cin)
        v = load4_guard(a.x + (((int64_t)n * a.H + gy) * a.W + gx) * a.ldx + c, a.Cin - c, vx);
      *reinterpret_cast<float4*>(x_lds + pix * 64 + q * 4) = v;
    }
    __syncthreads();
    if (bias) {      // column sums of the dy tile: thread = (channel, quarter of the pixels), fixed order
      const int co = tid & 63, part = tid >> 6;
      float s = 0.f;
      for (int i = 0; i < TPIX / 4; ++i) s += dy_lds[(part * (TPIX / 4) + i) * 64 + co];
      bsum += s;
    }
    // k-step = two neighbouring pixels of a tile row (lane half h takes pixel 2s + h)
#pragma unroll 2
    for (int row = 0; row < FW_TH; ++row) {
#pragma unroll 4
      for (int s2 = 0; s2 < FW_TW / 2; ++s2) {
        const int col = 2 * s2 + h;
        const float af = dy_lds[(row * FW_TW + col) * 64 + co_half * 32 + r];
#pragma unroll
        for (int kh = 0; kh < KN; ++kh)
#pragma unroll
          for (int kw = 0; kw < KN; ++kw) {
            const float bf = x_lds[((row + kh) * HWd + col + kw) * 64 + ci_half * 32 + r];
            acc[kh * KN + kw] = __builtin_amdgcn_mfma_f32_32x32x2f32(af, bf, acc[kh * KN + kw], 0, 0, 0);
          }
      }
    }
  }

  // ---- partial tile in the generic kernel's register order [tap][reg][wave][lane] (wgrad_reduce.h, kind 0)
  float* o = a.part + ((int64_t)split * npairs + pair) * (TAPS * 16 * 256) + tid;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[(t * 16 + e) * 256] = acc[t][e];
  if (bias) {
    __syncthreads();
    b_lds[tid] = bsum;
    __syncthreads();
    if (tid < 64)
      a.pbias[((int64_t)split * a.coTiles + cot) * 64 + tid] = (b_lds[tid] + b_lds[tid + 64]) + (b_lds[tid + 128] + b_lds[tid + 192]);
  }
}

// (O, kh, kw, I) -> (I, kh, kw, O) with both spatial axes reversed: the filter of the dgrad convolution
__global__ __launch_bounds__(256) void conv_f32_flip_kernel(const float* __restrict__ w, float* __restrict__ wd, int O, int I,
                                                            int taps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)O * I * taps) return;
  const int o = (int)(i % O);
  const int tap = (int)((i / O) % taps);
  const int ci = (int)(i / ((int64_t)O * taps));
  wd[i] = w[((int64_t)o * taps + (taps - 1 - tap)) * I + ci];
}

int check_f32_desc(const snws_conv_desc* d, const char* fn) {
  SNWS_CHECK_ARG(d, "%s: desc is NULL", fn);
  SNWS_CHECK_ARG(d->N >= 1 && d->H >= 1 && d->W >= 1 && d->C_in >= 1 && d->C_out >= 1, "%s: non-positive size", fn);
  const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1, k1 = d->kh == 1 && d->kw == 1 && d->pad == 0;
  if (!(k3 || k1) || d->stride != 1 || d->dil != 1 || d->groups != 1)
    SNWS_UNSUPPORTED("%s: only 3x3/pad1 or 1x1/pad0, stride 1, dilation 1, groups 1 (got k=%dx%d s=%d p=%d d=%d g=%d)", fn,
                     d->kh, d->kw, d->stride, d->pad, d->dil, d->groups);
  SNWS_CHECK_ARG(d->ldx >= d->C_in && d->ldy >= d->C_out, "%s: pixel stride smaller than channel count", fn);
  return SNWS_OK;
}

template <int TAPS>
int launch_f32_fwd(const F32FwdArgs& a, unsigned grid, hipStream_t st, const char* fn) {
  constexpr int PADK = TAPS == 9 ? 1 : 0;
  constexpr int LDS = ((F_TW + 2 * PADK) * (F_TH + 2 * PADK) + TAPS * 64) * F_ROW * 4;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_fwd_kernel<TAPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("%s: hipFuncSetAttribute failed: %s", fn, hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_f32_fwd_kernel<TAPS><<<grid, 256, LDS, st>>>(a);
  SNWS_LAUNCH_CHECK(fn);
  return SNWS_OK;
}

int run_f32_fwd(const char* fn, int N, int H, int W, int c_in, int c_out, int taps, const float* x, int64_t ldx,
                const float* w, const float* bias, float* y, int64_t ldy, float* stat, hipStream_t st) {
  SNWS_CHECK_ARG(x && w && y, "%s: NULL pointer", fn);
  F32FwdArgs a;
  a.x = x; a.w = w; a.bias = bias; a.y = y; a.stat = stat;
  a.N = N; a.H = H; a.W = W; a.Cin = c_in; a.Cout = c_out; a.ldx = ldx; a.ldy = ldy;
  a.tilesX = cdiv(W, F_TW); a.tilesY = cdiv(H, F_TH); a.coTiles = cdiv(c_out, 64);
  const int64_t nblk = (int64_t)N * a.tilesX * a.tilesY * a.coTiles;
  SNWS_CHECK_ARG(nblk < (1ll << 31), "%s: grid too large", fn);
  return taps == 9 ? launch_f32_fwd<9>(a, (unsigned)nblk, st, fn) : launch_f32_fwd<1>(a, (unsigned)nblk, st, fn);
}

struct F32WgradPlan {
  int npairs, coTiles, ciTiles, splits, tilesPerSplit, ntiles, E;
};

F32WgradPlan f32_wgrad_plan(const snws_conv_desc* d) {
  F32WgradPlan p;
  const int taps = d->kh * d->kw;
  p.coTiles = cdiv(d->C_out, 64);
  p.ciTiles = cdiv(d->C_in, 64);
  p.npairs = p.coTiles * p.ciTiles;
  p.ntiles = d->N * cdiv(d->W, FW_TW) * cdiv(d->H, FW_TH);
  p.E = taps * 4096;
  int splits = 1024 / p.npairs;          // about two resident rounds of workgroups
  if (splits < 1) splits = 1;
  if (splits > p.ntiles) splits = p.ntiles;
  p.tilesPerSplit = cdiv(p.ntiles, splits);
  p.splits = cdiv(p.ntiles, p.tilesPerSplit);
  return p;
}

template <int TAPS>
int launch_f32_wgrad(const F32WgradArgs& a, unsigned grid, hipStream_t st) {
  constexpr int PADK = TAPS == 9 ? 1 : 0;
  constexpr int LDS = (FW_TH * FW_TW * 64 + (FW_TW + 2 * PADK) * (FW_TH + 2 * PADK) * 64 + 256) * 4;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_f32_wgrad_kernel<TAPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("snws_conv_f32_wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_f32_wgrad_kernel<TAPS><<<grid, 256, LDS, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_f32_wgrad");
  return SNWS_OK;
}

}  // namespace

extern "C" int snws_conv_f32_tiles(int H, int W) { return cdiv(H, F_TH) * cdiv(W, F_TW); }

extern "C" int snws_conv_f32_fwd(const snws_conv_desc* d, const float* x, const float* w_ohwi, const float* bias, float* y,
                                 float* stat_part, void* stream) {
  if (int rc = check_f32_desc(d, "snws_conv_f32_fwd")) return rc;
  return run_f32_fwd("snws_conv_f32_fwd", d->N, d->H, d->W, d->C_in, d->C_out, d->kh * d->kw, x, d->ldx, w_ohwi, bias, y,
                     d->ldy, stat_part, (hipStream_t)stream);
}

extern "C" int snws_conv_f32_dgrad(const snws_conv_desc* d, const float* dy, const float* w_ihwo_flipped, float* dx,
                                   void* stream) {
  if (int rc = check_f32_desc(d, "snws_conv_f32_dgrad")) return rc;
  return run_f32_fwd("snws_conv_f32_dgrad", d->N, d->H, d->W, d->C_out, d->C_in, d->kh * d->kw, dy, d->ldy, w_ihwo_flipped,
                     nullptr, dx, d->ldx, nullptr, (hipStream_t)stream);
}

extern "C" int snws_conv_f32_flip(const float* w_ohwi, float* w_ihwo_flipped, int O, int I, int kh, int kw, void* stream) {
  SNWS_CHECK_ARG(w_ohwi && w_ihwo_flipped && O >= 1 && I >= 1 && kh >= 1 && kw >= 1, "snws_conv_f32_flip: bad arguments");
  const int64_t n = (int64_t)O * I * kh * kw;
  conv_f32_flip_kernel<<<(unsigned)cdiv64(n, 256), 256, 0, (hipStream_t)stream>>>(w_ohwi, w_ihwo_flipped, O, I, kh * kw);
  SNWS_LAUNCH_CHECK("snws_conv_f32_flip");
  return SNWS_OK;
}

extern "C" int64_t snws_conv_f32_wgrad_workspace_bytes(const snws_conv_desc* d) {
  if (!d || check_f32_desc(d, "snws_conv_f32_wgrad_workspace_bytes")) return -1;
  const F32WgradPlan p = f32_wgrad_plan(d);
  return ((int64_t)p.splits * p.npairs * p.E + (int64_t)p.splits * p.coTiles * 64) * 4;
}

extern "C" int snws_conv_f32_wgrad(const snws_conv_desc* d, const float* x, const float* dy, float* dw_ohwi, float* dbias,
                                   void* workspace, int64_t workspace_bytes, void* stream) {
  if (int rc = check_f32_desc(d, "snws_conv_f32_wgrad")) return rc;
  SNWS_CHECK_ARG(x && dy && dw_ohwi && workspace, "snws_conv_f32_wgrad: NULL pointer");
  SNWS_CHECK_ARG(((uintptr_t)workspace & 15) == 0, "snws_conv_f32_wgrad: workspace must be 16-byte aligned");
  const F32WgradPlan p = f32_wgrad_plan(d);
  const int64_t need = ((int64_t)p.splits * p.npairs * p.E + (int64_t)p.splits * p.coTiles * 64) * 4;
  SNWS_CHECK_ARG(workspace_bytes >= need, "snws_conv_f32_wgrad: workspace of %lld bytes, %lld needed",
                 (long long)workspace_bytes, (long long)need);
  hipStream_t st = (hipStream_t)stream;
  const int taps = d->kh * d->kw;
  F32WgradArgs a;
  a.x = x; a.dy = dy; a.part = (float*)workspace; a.pbias = a.part + (int64_t)p.splits * p.npairs * p.E;
  a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->C_in; a.Cout = d->C_out; a.want_bias = dbias != nullptr;
  a.ldx = d->ldx; a.lddy = d->ldy;
  a.tilesX = cdiv(d->W, FW_TW); a.tilesY = cdiv(d->H, FW_TH);
  a.coTiles = p.coTiles; a.ciTiles = p.ciTiles; a.tilesPerSplit = p.tilesPerSplit;
  const unsigned grid = (unsigned)(p.npairs * p.splits);
  if (int rc = taps == 9 ? launch_f32_wgrad<9>(a, grid, st) : launch_f32_wgrad<1>(a, grid, st)) return rc;
  WgradReduceArgs r;
  r.part = a.part; r.pbias = a.pbias; r.dw = dw_ohwi; r.dbias = dbias;
  r.kind = 0; r.taps = taps; r.E = p.E; r.TC = 64;
  r.npairs = p.npairs; r.splits = p.splits; r.coTiles = p.coTiles; r.Cin = d->C_in; r.Cout = d->C_out;
  r.main_blocks = (int)((int64_t)p.npairs * p.E / 256);
  wgrad_reduce_kernel<<<r.main_blocks + (dbias ? p.coTiles : 0), 256, 0, st>>>(r);
  SNWS_LAUNCH_CHECK("snws_conv_f32_wgrad(reduce)");
  return SNWS_OK;
}
