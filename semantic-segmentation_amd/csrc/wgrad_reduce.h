// Fixed-order reduction of the per-workgroup partial filter tiles of the wgrad kernels (bf16 MFMA kernels of
// conv_mfma.hip, f32 MFMA kernel of conv_f32.hip).  Internal header: every translation unit that includes it gets its
// own copy of the kernel.
#pragma once
#include "common.h"

namespace {

// Sums the per-workgroup partial tiles of a wgrad launch in a fixed order and writes (OVERWRITES) the f32 gradient
// dw (O, taps, Cin); the bias partials are summed the same way and ADDED to dbias by one thread per channel (no
// atomics: kernels on a stream run one after the other, so the result is bit-reproducible).
// kind: 0 generic 64-wide, 1 generic C_in = 16 (its two pixel-row halves hold partial sums of the same outputs),
//       2 wgrad1, 3 wgrad1w, 5 wgrad9 (conv_wgrad9q_kernel and conv_wgrad9p_kernel), 6 wgrad1t - the register orders the kernels above store
//       their tiles in.
struct WgradReduceArgs {
  const float* part;
  const float* pbias;
  float* dw;
  float* dbias;
  int kind, taps, E, TC;           // E = floats per partial tile, TC = channels per out-channel tile
  int npairs, splits, coTiles, Cin, Cout;
  int main_blocks;                 // blocks [0, main_blocks) reduce tiles, the rest bias partials
};

__device__ __forceinline__ bool wgrad_decode(int kind, int e, int cot, int cit, int& co, int& tap, int& ci) {
  if (kind == 6) {          // wgrad1t (192 x 160 tile): [out block][in block][j][wave = (out half, in half)][lane]
    const int lane = e & 63, wave = (e >> 6) & 3, q = e >> 8;
    const int j = q & 3, blk = q >> 2, ob = blk / 5, ib = blk - ob * 5;
    tap = 0;
    co = cot * 192 + (wave & 1) * 96 + ob * 16 + 4 * (lane >> 4) + j;
    ci = cit * 160 + (wave >> 1) * 80 + ib * 16 + (lane & 15);
    return true;
  }
  if (kind == 5) {          // wgrad9, 16x16x32 form: [tap][out block][in block][j][in-channel half][lane]
    const int lane = e & 63, cih = (e >> 6) & 1, q = e >> 7;
    const int j = q & 3, cbk = (q >> 2) & 1, ob = (q >> 3) & 3;
    tap = q >> 5;
    co = cot * 64 + ob * 16 + 4 * (lane >> 4) + j;
    ci = cit * 64 + cih * 32 + cbk * 16 + (lane & 15);
    return true;
  }
  if (kind == 3) {          // [i][h][r][wave][lane], wave = (out quadrant, in quadrant)
    const int lane = e & 63, wave = (e >> 6) & 3, idx = e >> 8;
    const int r = idx & 15, h = (idx >> 4) & 1, i = idx >> 5;
    tap = 0;
    co = cot * 128 + (wave & 1) * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    ci = cit * 128 + (wave >> 1) * 64 + h * 32 + (lane & 31);
    return true;
  }
  if (kind == 2) {          // [r][wave = quadrant (i, h)][lane]
    const int lane = e & 63, wave = (e >> 6) & 3, r = e >> 8;
    tap = 0;
    co = cot * 64 + (wave >> 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    ci = cit * 64 + (wave & 1) * 32 + (lane & 31);
    return true;
  }
  // generic kernel: [tap][r][wave][lane], wave = (out half, in half | pixel-row half)
  const int lane = e & 63, wave = (e >> 6) & 3, q = e >> 8;
  const int r = q & 15;
  tap = q >> 4;
  co = cot * 64 + (wave & 1) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  if (kind == 1) {          // waves 2, 3 hold the second pixel-row half of the same outputs: summed into waves 0, 1
    ci = lane & 31;
    return (wave >> 1) == 0 && (lane & 31) < 16;
  }
  ci = cit * 64 + (wave >> 1) * 32 + (lane & 31);
  return true;
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(WgradReduceArgs a) {
  __shared__ float4 red[3][64];
  __shared__ float bred[3][64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if ((int)blockIdx.x < a.main_blocks) {
    // block = 256 consecutive tile elements (float4 per lane) x 4 interleaved split classes (one per wave)
    const int64_t g = (int64_t)blockIdx.x * 256 + lane * 4;
    const int pair = (int)(g / a.E), e = (int)(g - (int64_t)pair * a.E);
    const int64_t stride = (int64_t)a.npairs * a.E;
    const float* p = a.part + (int64_t)pair * a.E + e;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    auto add = [](float4& acc, const float4 v) { acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; };
    int s = w;
    for (; s + 12 < a.splits; s += 16) {
      const float4 v0 = *reinterpret_cast<const float4*>(p + (int64_t)s * stride);
      const float4 v1 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 4) * stride);
      const float4 v2 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 8) * stride);
      const float4 v3 = *reinterpret_cast<const float4*>(p + (int64_t)(s + 12) * stride);
      add(s0, v0); add(s1, v1); add(s2, v2); add(s3, v3);
    }
    for (; s < a.splits; s += 4) add(s0, *reinterpret_cast<const float4*>(p + (int64_t)s * stride));
    add(s0, s1); add(s2, s3); add(s0, s2);
    if (a.kind == 1) {      // fold the second pixel-row half (tile element e + 128) in
      float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (((e >> 6) & 3) < 2)
        for (int s_ = w; s_ < a.splits; s_ += 4) add(t0, *reinterpret_cast<const float4*>(p + 128 + (int64_t)s_ * stride));
      add(s0, t0);
    }
    if (w > 0) red[w - 1][lane] = s0;
    __syncthreads();
    if (w == 0) {
      add(s0, red[0][lane]); add(s0, red[1][lane]); add(s0, red[2][lane]);
      const int cot = pair % a.coTiles, cit = pair / a.coTiles;
      const float v[4] = {s0.x, s0.y, s0.z, s0.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int co, tap, ci;
        if (wgrad_decode(a.kind, e + j, cot, cit, co, tap, ci) && co < a.Cout && ci < a.Cin)
          a.dw[((int64_t)co * a.taps + tap) * a.Cin + ci] = v[j];
      }
    }
  } else {
    // bias: block = 64 channels of one out-channel tile x 4 split classes
    const int b = (int)blockIdx.x - a.main_blocks;          // b = cot * (TC / 64) + 64-channel group
    const int per = a.TC / 64, cot = b / per, c = (b % per) * 64 + lane;
    const float* p = a.pbias + (int64_t)cot * a.TC + c;
    const int64_t stride = (int64_t)a.coTiles * a.TC;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int s = w;
    for (; s + 12 < a.splits; s += 16) {
      s0 += p[(int64_t)s * stride]; s1 += p[(int64_t)(s + 4) * stride];
      s2 += p[(int64_t)(s + 8) * stride]; s3 += p[(int64_t)(s + 12) * stride];
    }
    for (; s < a.splits; s += 4) s0 += p[(int64_t)s * stride];
    s0 = (s0 + s1) + (s2 + s3);
    if (w > 0) bred[w - 1][lane] = s0;
    __syncthreads();
    if (w == 0) {
      s0 = ((s0 + bred[0][lane]) + bred[1][lane]) + bred[2][lane];
      const int co = cot * a.TC + c;
      if (co < a.Cout) a.dbias[co] += s0;
    }
  }
}


}  // namespace
