// The last block of the network fused with the classification head: SwitchNorm + ReLU of the final fusion convolution
// (modeling/snorm.py:41-81 + nn.ReLU, modeling/UNet3p_SNws.py:29-33) followed by the 1x1 nn.Conv2d head
// (modeling/UNet3p_SNws.py:93,158; modeling/UNet_SNws.py:93,151), forward and backward, without ever storing the
// normalised activation or its gradient: at 512x512 those two tensors are 2.7 GB each per 16 images, each written once
// and read once or twice by HBM-bound passes.
//   forward : logits = W_head * relu((x - mean) * scale + beta) + b_head   (x = the fusion convolution's output)
//   backward: the gradient of the normalised activation, W_head^T * dlogits (K = 16 padded classes), is rebuilt on the
//             fly by two small MFMAs per 16 pixels x 32 channels wherever the SwitchNorm backward passes need it;
//             pass 1 also accumulates the head's weight / bias gradient (per-workgroup partials, summed in a fixed order).
// bf16 NHWC, C a multiple of 32, at most 16 classes (the logits tensor is 16 channels wide, the padding channels hold
// what zero filter rows give: the bias padding, i.e. 0).  The small contractions run on v_mfma_f32_16x16x16_bf16.
//
// Lane layout of every kernel: a wave works on 16 consecutive pixels x 32 channels at a time; lane = (pixel r = lane &
// 15, channel octet h = lane >> 4) holds channels 8h .. 8h+7 of the 32 - one 16-byte load, and exactly the operand /
// result layout of two 16x16x16 MFMAs whose contraction (forward) or output rows (backward) are permuted accordingly.
#include "common.h"
#include <stdlib.h>
#include "sn_ctx.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ float bf16_round_f(float v) { return bf16_bits_to_f32(f32_to_bf16_bits(v)); }

__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(u[i] << 16);
    f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ s16x4 pack4(float a, float b, float c, float d) {
  const unsigned lo = pack_bf16x2(a, b), hi = pack_bf16x2(c, d);
  s16x4 r;
  r[0] = (short)(lo & 0xffff); r[1] = (short)(lo >> 16); r[2] = (short)(hi & 0xffff); r[3] = (short)(hi >> 16);
  return r;
}

// ------------------------------------------------------------------------------------------------- forward
template <int NB>
__global__ __launch_bounds__(256) void sn_apply_head_kernel(const bf16_t* __restrict__ x, int64_t ldx,
                                                            const float* __restrict__ ctx, const float* __restrict__ beta,
                                                            const bf16_t* __restrict__ hw, const float* __restrict__ hb,
                                                            bf16_t* __restrict__ logits, int N, int64_t HW, int PT, int relu) {
  constexpr int C = 32 * NB;
  __shared__ __attribute__((aligned(16))) float coef[3][C];      // mean, scale, beta of this image's channels
  const int t = blockIdx.x, n = blockIdx.y, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, h = lane >> 4;
  const CtxOff o = ctx_off(N, C);
  for (int c = tid; c < C; c += 256) {
    coef[0][c] = ctx[o.mean + (int64_t)n * C + c];
    coef[1][c] = ctx[o.scale + (int64_t)n * C + c];
    coef[2][c] = beta[c];
  }
  // head filter fragments: lane (class m = r, octet h) holds W[m][32j + 8h .. +7]; first half -> MFMA a, second -> MFMA b
  s16x4 wa[NB], wb[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const uint4 v = *reinterpret_cast<const uint4*>(hw + (int64_t)r * C + 32 * j + 8 * h);
    wa[j][0] = (short)(v.x & 0xffff); wa[j][1] = (short)(v.x >> 16); wa[j][2] = (short)(v.y & 0xffff); wa[j][3] = (short)(v.y >> 16);
    wb[j][0] = (short)(v.z & 0xffff); wb[j][1] = (short)(v.z >> 16); wb[j][2] = (short)(v.w & 0xffff); wb[j][3] = (short)(v.w >> 16);
  }
  float hbv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) hbv[i] = hb ? hb[4 * h + i] : 0.f;
  __syncthreads();

  // PT > 0: the workgroup owns the contiguous pixels [t PT, (t+1) PT); PT == 0: the workgroups of an image interleave
  // their 64-pixel steps, so the resident ones sweep one compact window of the tensor (DRAM page locality)
  const int64_t p0 = PT > 0 ? (int64_t)t * PT : (int64_t)t * 64;
  const int64_t p1 = PT > 0 ? (p0 + PT < HW ? p0 + PT : HW) : HW;
  const int64_t gstep = PT > 0 ? 64 : (int64_t)gridDim.x * 64;
  const bf16_t* xn = x + (int64_t)n * HW * ldx;
  bf16_t* ln_ = logits + (int64_t)n * HW * 16;
  for (int64_t g0 = p0 + 16 * wave; g0 < p1; g0 += gstep) {
    const int64_t px = g0 + r;
    const bool ok = px < p1;
    const bf16_t* xp = xn + (ok ? px : p1 - 1) * ldx + 8 * h;
    uint4 xv[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) xv[j] = *reinterpret_cast<const uint4*>(xp + 32 * j);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};     // two chains: an MFMA does not wait for the one before it
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      float f[8], y[8];
      unpack8(xv[j], f);
      const int c0 = 32 * j + 8 * h;
      const float4 m0 = *reinterpret_cast<const float4*>(&coef[0][c0]), m1 = *reinterpret_cast<const float4*>(&coef[0][c0 + 4]);
      const float4 s0 = *reinterpret_cast<const float4*>(&coef[1][c0]), s1 = *reinterpret_cast<const float4*>(&coef[1][c0 + 4]);
      const float4 b0 = *reinterpret_cast<const float4*>(&coef[2][c0]), b1 = *reinterpret_cast<const float4*>(&coef[2][c0 + 4]);
      const float mu[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
      const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
      const float be[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float z = fmaf(f[e] - mu[e], sc[e], be[e]);
        y[e] = relu ? fmaxf(z, 0.f) : z;
      }
      acc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa[j], pack4(y[0], y[1], y[2], y[3]), acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wb[j], pack4(y[4], y[5], y[6], y[7]), acc2, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] += acc2[i];
    // D[class][pixel]: lane holds pixel r, classes 4h .. 4h+3
    if (ok) {
      uint2 v;
      v.x = pack_bf16x2(acc[0] + hbv[0], acc[1] + hbv[1]);
      v.y = pack_bf16x2(acc[2] + hbv[2], acc[3] + hbv[3]);
      *reinterpret_cast<uint2*>(ln_ + px * 16 + 4 * h) = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------- backward
// Streaming layout of the two backward passes: a wave owns one 64-channel slab (128 bytes per pixel, one cache line)
// and walks over pixels 16 at a time; lane = (pixel row prow = lane >> 3, channel octet oct = lane & 7) reads /
// writes 16 bytes, so a wave instruction moves eight whole lines and a lane's eight channels - hence its SwitchNorm
// coefficients - never change.  dY[channel][pixel] = sum_k W[k][channel] * dlogits[pixel][k] comes from four 16x16x16
// MFMAs per step (lane = pixel lane & 15, channels 4 * (lane >> 4) ..) and reaches the streaming layout through a
// wave-private LDS tile [16 px][64 ch] bf16; pass 1 then overwrites that tile with the activation the head saw and
// reads it (and the dlogits tile) back with the transposing LDS read, pixel-major, for the head's weight gradient
// dW[k][c] = sum_p dlogits[p][k] * y[p][c] - four more MFMAs.  No workgroup barrier anywhere: waves only share L1 hits.
typedef __attribute__((ext_vector_type(4))) short short4v;

template <int C>
struct HeadGeom {
  static constexpr int NS = C / 64;               // channel slabs = waves side by side on the same pixels
  static constexpr int PG = C == 64 ? 4 : 1;      // pixel groups: waves of a block on different pixels
  static constexpr int WAVES = NS * PG, THREADS = 64 * WAVES;
};
constexpr int HB_RS = 144;                         // LDS bytes per pixel row of the exchange tile: 128 + 16 padding
constexpr int HB_TILE = 16 * HB_RS + 512;          // + [16 px][16 classes] bf16 of dlogits

__device__ __forceinline__ void wave_lds_sync() {  // LDS traffic of one wave is in order; this keeps the compiler in order too
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ short4v lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
}

struct HeadBwdArgs {
  const bf16_t* dl;
  const bf16_t* x;
  const float* ctx;
  const float* ctxb;
  const float* beta;
  const bf16_t* hw;
  float* part;      // pass 1: [N][Tn][C][2] = {sum g, sum g * xhat}, g = relu mask * dY
  float* hwp;       // pass 1: [N * Tn][16][C] head weight gradient partials
  float* hbp;       // pass 1: [N * Tn][16] head bias gradient partials
  bf16_t* dx;       // pass 2
  int64_t ldx, lddx, HW;
  int N, PT, Tn, relu, il;
};

template <int C, bool APPLY>
__global__ __launch_bounds__(HeadGeom<C>::THREADS, APPLY ? 4 : 3) void sn_head_bwd_kernel(HeadBwdArgs a) {
  using G = HeadGeom<C>;
  __shared__ __attribute__((aligned(16))) unsigned char lds_all[G::WAVES][HB_TILE];
  const int t = blockIdx.x, n = blockIdx.y, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int s = wave % G::NS, pg = wave / G::NS;
  unsigned char* tile = lds_all[wave];
  unsigned char* dlt = tile + 16 * HB_RS;
  const int r = lane & 15, g = lane >> 4;            // MFMA geometry
  const int oct = lane & 7, prow = lane >> 3;        // streaming geometry
  const int tq = (lane >> 2) & 3, tp = lane & 3;     // transposing read: lane addresses pixel 4g + tq, elements 4tp ..
  const int c0 = 64 * s + 8 * oct;

  // this wave's pixels.  il == 0: the block's tile cut into PG runs of whole 16-pixel steps.  il != 0: the 16-pixel
  // steps of the image are dealt round-robin to the (block, pixel group) pairs, so the resident blocks sweep one
  // compact window of each tensor (DRAM page locality) instead of far-apart runs
  int64_t p0, p1, qstep;
  if (a.il) {
    p0 = ((int64_t)t * G::PG + pg) * 16;
    p1 = a.HW;
    qstep = (int64_t)gridDim.x * G::PG * 16;
  } else {
    const int64_t b0 = (int64_t)t * a.PT;
    const int64_t b1 = b0 + a.PT < a.HW ? b0 + a.PT : a.HW;
    const int64_t per = ((b1 - b0 + G::PG * 16 - 1) / (G::PG * 16)) * 16;
    p0 = b0 + pg * per;
    p1 = p0 + per < b1 ? p0 + per : b1;
    qstep = 16;
  }

  const CtxOff o = ctx_off(a.N, C);
  float mu[8], sc[8], be[8], Bc[8], Kc[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t i = (int64_t)n * C + c0 + e;
    mu[e] = a.ctx[o.mean + i];
    sc[e] = a.ctx[o.scale + i];
    be[e] = a.beta[c0 + e];
    if (APPLY) {
      const CtxbOff ob = ctxb_off(a.N, C);
      Bc[e] = a.ctxb[ob.B + i];
      Kc[e] = a.ctxb[ob.K + i];
    }
  }
  // A fragments of the dY MFMAs: row m = r -> channel 64s + 16cb + r, k = 4g + i -> class
  short4v wa[4];
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      wa[cb][i] = (short)__builtin_bit_cast(unsigned short, a.hw[(int64_t)(4 * g + i) * C + 64 * s + 16 * cb + r]);

  float P[8], Q[8], Bk[4];
  f32x4 wg[4];
#pragma unroll
  for (int e = 0; e < 8; ++e) P[e] = Q[e] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    Bk[i] = 0.f;
    wg[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const bf16_t* xn = a.x + (int64_t)n * a.HW * a.ldx + c0;
  const bf16_t* dn = a.dl + (int64_t)n * a.HW * 16;
  bf16_t* on = APPLY ? a.dx + (int64_t)n * a.HW * a.lddx + c0 : nullptr;
  auto load_step = [&](int64_t base, uint4 (&xv)[2], uint2& dq) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int64_t px = base + prow + 8 * jj;
      xv[jj] = *reinterpret_cast<const uint4*>(xn + (px < p1 ? px : p1 - 1) * a.ldx);
    }
    const int64_t pr = base + r;
    dq = *reinterpret_cast<const uint2*>(dn + (pr < p1 ? pr : p1 - 1) * 16 + 4 * g);
    if (pr >= p1) dq = make_uint2(0u, 0u);            // past the end: a zero gradient row
  };
  auto compute_step = [&](int64_t base, const uint4 (&xv)[2], const uint2 dq) {
    short4v dlf;
    dlf[0] = (short)(dq.x & 0xffff); dlf[1] = (short)(dq.x >> 16);
    dlf[2] = (short)(dq.y & 0xffff); dlf[3] = (short)(dq.y >> 16);
    if (!APPLY) *reinterpret_cast<uint2*>(dlt + r * 32 + 8 * g) = dq;
#pragma unroll
    for (int cb = 0; cb < 4; ++cb) {
      const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(wa[cb], dlf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      // D[channel 16cb + 4g + i][pixel r]
      *reinterpret_cast<uint2*>(tile + r * HB_RS + (16 * cb + 4 * g) * 2) =
          make_uint2(pack_bf16x2(d[0], d[1]), pack_bf16x2(d[2], d[3]));
    }
    wave_lds_sync();
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      unsigned char* mine = tile + (prow + 8 * jj) * HB_RS + oct * 16;
      float f[8], dy[8];
      unpack8(xv[jj], f);
      unpack8(*reinterpret_cast<const uint4*>(mine), dy);
      float out[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float ee = f[e] - mu[e];
        const float z = fmaf(ee, sc[e], be[e]);
        const float gg = (!a.relu || z > 0.f) ? dy[e] : 0.f;
        if (APPLY) {
          out[e] = fmaf(sc[e], gg, fmaf(Bc[e], f[e], Kc[e]));
        } else {
          P[e] += gg;
          Q[e] = fmaf(gg, ee, Q[e]);                  // times rstd at the end
          out[e] = a.relu ? fmaxf(z, 0.f) : z;        // the activation the head multiplied
        }
      }
      const uint4 ov = make_uint4(pack_bf16x2(out[0], out[1]), pack_bf16x2(out[2], out[3]), pack_bf16x2(out[4], out[5]),
                                  pack_bf16x2(out[6], out[7]));
      if (APPLY) {
        const int64_t px = base + prow + 8 * jj;
        if (px < p1) *reinterpret_cast<uint4*>(on + px * a.lddx) = ov;
      } else {
        *reinterpret_cast<uint4*>(mine) = ov;         // over the very bytes this lane has just read
      }
    }
    if (!APPLY) {
      wave_lds_sync();
      // dW[class][channel] += sum over the 16 pixels: A = dlogits^T (row = class r, k = pixel 4g + i), B = y
      const short4v dlT = lds_tr16(dlt + (4 * g + tq) * 32 + 8 * tp);
#pragma unroll
      for (int cb = 0; cb < 4; ++cb) {
        const short4v yB = lds_tr16(tile + (4 * g + tq) * HB_RS + (16 * cb + 4 * tp) * 2);
        wg[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dlT, yB, wg[cb], 0, 0, 0);
      }
      Bk[0] += __uint_as_float(dq.x << 16); Bk[1] += __uint_as_float(dq.x & 0xffff0000u);
      Bk[2] += __uint_as_float(dq.y << 16); Bk[3] += __uint_as_float(dq.y & 0xffff0000u);
    }
    wave_lds_sync();                                  // the next step rewrites the tiles
  };

  // software pipeline: the loads of step i + 1 are in flight while step i computes (addresses past the run are
  // clamped, so the look-ahead load is always legal)
  if (p0 < p1) {
    uint4 xa[2], xb[2];
    uint2 da, db;
    load_step(p0, xa, da);
    for (int64_t q0 = p0; q0 < p1; q0 += 2 * qstep) {
      load_step(q0 + qstep, xb, db);
      compute_step(q0, xa, da);
      if (q0 + qstep >= p1) break;                    // wave-uniform
      load_step(q0 + 2 * qstep, xa, da);
      compute_step(q0 + qstep, xb, db);
    }
  }
  if (APPLY) return;

  const int64_t blk = (int64_t)n * a.Tn + (int64_t)t * G::PG + pg;
  // P, Q: sum over the eight pixel rows (lane bits 3..5); lanes 0..7 hold their octet's totals
#pragma unroll
  for (int e = 0; e < 8; ++e) {
#pragma unroll
    for (int sft = 8; sft < 64; sft <<= 1) {
      P[e] += __shfl_xor(P[e], sft, 64);
      Q[e] += __shfl_xor(Q[e], sft, 64);
    }
  }
  if (lane < 8) {
    float* po = a.part + (blk * C + c0) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      po[2 * e] = P[e];
      po[2 * e + 1] = Q[e] * a.ctx[o.rstd + (int64_t)n * C + c0 + e];
    }
  }
  // dW: lane holds classes 4g + i of channel 64s + 16cb + r
#pragma unroll
  for (int cb = 0; cb < 4; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i) a.hwp[(blk * 16 + 4 * g + i) * C + 64 * s + 16 * cb + r] = wg[cb][i];
  // bias: lane holds classes 4g + i of pixel lane r; sum over r
  if (s == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
      for (int sft = 1; sft < 16; sft <<= 1) Bk[i] += __shfl_xor(Bk[i], sft, 64);
      if (r == 0) a.hbp[blk * 16 + 4 * g + i] = Bk[i];
    }
  }
}

// out[i] (+)= sum over parts of part[p * len + i], parts taken in a fixed order: block = 64 columns x 8 interleaved
// classes of parts (one per wave), four loads in flight per lane
__global__ __launch_bounds__(512) void head_partial_sum_kernel(const float* __restrict__ part, int nparts, int len,
                                                               float* __restrict__ out, int accumulate) {
  __shared__ float red[7][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + lane;
  part += (int64_t)blockIdx.y * nparts * len;       // batch entry
  out += (int64_t)blockIdx.y * len;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < len) {
    const float* p = part + i;
    int q = w;
    for (; q + 24 < nparts; q += 32) {
      s0 += p[(int64_t)q * len]; s1 += p[(int64_t)(q + 8) * len];
      s2 += p[(int64_t)(q + 16) * len]; s3 += p[(int64_t)(q + 24) * len];
    }
    for (; q < nparts; q += 8) s0 += p[(int64_t)q * len];
  }
  s0 = (s0 + s1) + (s2 + s3);
  if (w > 0) red[w - 1][lane] = s0;
  __syncthreads();
  if (w == 0 && i < len) {
#pragma unroll
    for (int k = 0; k < 7; ++k) s0 += red[k][lane];
    out[i] = (accumulate ? out[i] : 0.f) + s0;
  }
}

int head_geometry_ok(int C, int ncls, const char* fn) {
  if (C != 64 && C != 320) SNWS_UNSUPPORTED("%s: %d channels (built for 64 and 320)", fn, C);
  if (ncls < 1 || ncls > 16) SNWS_UNSUPPORTED("%s: %d classes (1..16)", fn, ncls);
  return SNWS_OK;
}

}  // namespace

// Pixels per workgroup of the backward passes (measured at 16 x 512 x 512 x 320: many short workgroups stream better
// than one round of long ones; pass 1 pays for its partial-sum rows, so its runs are longer).
static int head_bwd_pt(int apply) { return apply ? 512 : 2048; }
// SNWS_SN_INTERLEAVE = 0: contiguous pixel runs per workgroup instead of interleaved steps (A/B measurements)
static bool head_interleave() {
  static const bool on = !(getenv("SNWS_SN_INTERLEAVE") && atoi(getenv("SNWS_SN_INTERLEAVE")) == 0);
  return on;
}

// pixel tiles per image: forward workgroups, or (backward = 1) the partial-sum rows per image that
// snws_sn_head_bwd_reduce writes for C channels
extern "C" int snws_sn_head_tiles(int64_t HW, int C, int backward) {
  if (!backward) return (int)cdiv64(HW, 2048);
  return (int)cdiv64(HW, head_bwd_pt(0)) * (C == 64 ? HeadGeom<64>::PG : HeadGeom<320>::PG);
}

extern "C" int snws_sn_apply_head(const void* x, int64_t ldx, const float* ctx, const float* beta, const void* head_w,
                                  const float* head_b, void* logits, int N, int64_t HW, int C, int relu, void* stream) {
  if (int rc = head_geometry_ok(C, 1, "snws_sn_apply_head")) return rc;
  SNWS_CHECK_ARG(x && ctx && beta && head_w && logits, "snws_sn_apply_head: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && HW >= 1 && ldx >= C && ldx % 8 == 0, "snws_sn_apply_head: bad sizes");
  SNWS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)head_w & 15) == 0 && ((uintptr_t)logits & 7) == 0,
                 "snws_sn_apply_head: pointers must be 16-byte aligned");
  const int T = (int)cdiv64(HW, 2048), PT = head_interleave() ? 0 : 2048;
  hipStream_t st = (hipStream_t)stream;
  if (C == 320)
    sn_apply_head_kernel<10><<<dim3(T, N), 256, 0, st>>>((const bf16_t*)x, ldx, ctx, beta, (const bf16_t*)head_w, head_b,
                                                         (bf16_t*)logits, N, HW, PT, relu);
  else
    sn_apply_head_kernel<2><<<dim3(T, N), 256, 0, st>>>((const bf16_t*)x, ldx, ctx, beta, (const bf16_t*)head_w, head_b,
                                                        (bf16_t*)logits, N, HW, PT, relu);
  SNWS_LAUNCH_CHECK("snws_sn_apply_head");
  return SNWS_OK;
}

extern "C" int snws_sn_head_bwd_reduce(const void* dlogits, const void* x, int64_t ldx, const float* ctx, const float* beta,
                                       const void* head_w, float* part, float* hw_part, float* hb_part, int N, int64_t HW,
                                       int C, int relu, void* stream) {
  if (int rc = head_geometry_ok(C, 1, "snws_sn_head_bwd_reduce")) return rc;
  SNWS_CHECK_ARG(dlogits && x && ctx && beta && head_w && part && hw_part && hb_part, "snws_sn_head_bwd_reduce: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && N <= 65535 && HW >= 1 && ldx >= C && ldx % 8 == 0, "snws_sn_head_bwd_reduce: bad sizes");
  SNWS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dlogits & 7) == 0, "snws_sn_head_bwd_reduce: unaligned pointers");
  HeadBwdArgs a{};
  a.dl = (const bf16_t*)dlogits; a.x = (const bf16_t*)x; a.ctx = ctx; a.beta = beta; a.hw = (const bf16_t*)head_w;
  a.part = part; a.hwp = hw_part; a.hbp = hb_part;
  a.ldx = ldx; a.HW = HW; a.N = N; a.PT = head_bwd_pt(0); a.relu = relu; a.il = head_interleave();
  const int T = (int)cdiv64(HW, a.PT);
  a.Tn = snws_sn_head_tiles(HW, C, 1);
  hipStream_t st = (hipStream_t)stream;
  if (C == 320) sn_head_bwd_kernel<320, false><<<dim3(T, N), HeadGeom<320>::THREADS, 0, st>>>(a);
  else sn_head_bwd_kernel<64, false><<<dim3(T, N), HeadGeom<64>::THREADS, 0, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_sn_head_bwd_reduce");
  return SNWS_OK;
}

extern "C" int snws_sn_head_bwd_apply(const void* dlogits, const void* x, int64_t ldx, const float* ctx, const float* ctxb,
                                      const float* beta, const void* head_w, void* dx, int64_t lddx, int N, int64_t HW, int C,
                                      int relu, void* stream) {
  if (int rc = head_geometry_ok(C, 1, "snws_sn_head_bwd_apply")) return rc;
  SNWS_CHECK_ARG(dlogits && x && ctx && ctxb && beta && head_w && dx, "snws_sn_head_bwd_apply: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && N <= 65535 && HW >= 1 && ldx >= C && ldx % 8 == 0 && lddx >= C && lddx % 8 == 0,
                 "snws_sn_head_bwd_apply: bad sizes");
  SNWS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dlogits & 7) == 0 && ((uintptr_t)dx & 15) == 0,
                 "snws_sn_head_bwd_apply: unaligned pointers");
  HeadBwdArgs a{};
  a.dl = (const bf16_t*)dlogits; a.x = (const bf16_t*)x; a.ctx = ctx; a.ctxb = ctxb; a.beta = beta; a.hw = (const bf16_t*)head_w;
  a.dx = (bf16_t*)dx; a.ldx = ldx; a.lddx = lddx; a.HW = HW; a.N = N; a.PT = head_bwd_pt(1); a.relu = relu; a.il = head_interleave();
  const int T = (int)cdiv64(HW, a.PT);
  hipStream_t st = (hipStream_t)stream;
  if (C == 320) sn_head_bwd_kernel<320, true><<<dim3(T, N), HeadGeom<320>::THREADS, 0, st>>>(a);
  else sn_head_bwd_kernel<64, true><<<dim3(T, N), HeadGeom<64>::THREADS, 0, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_sn_head_bwd_apply");
  return SNWS_OK;
}

extern "C" int snws_partial_sum(const float* part, int n_parts, int len, float* out, int accumulate, void* stream) {
  return snws_partial_sum_batched(part, 1, n_parts, len, out, accumulate, stream);
}

extern "C" int snws_partial_sum_batched(const float* part, int batch, int n_parts, int len, float* out, int accumulate,
                                        void* stream) {
  SNWS_CHECK_ARG(part && out && batch >= 1 && batch <= 65535 && n_parts >= 1 && len >= 1, "snws_partial_sum: bad arguments");
  head_partial_sum_kernel<<<dim3(cdiv(len, 64), batch), 512, 0, (hipStream_t)stream>>>(part, n_parts, len, out, accumulate);
  SNWS_LAUNCH_CHECK("snws_partial_sum");
  return SNWS_OK;
}
