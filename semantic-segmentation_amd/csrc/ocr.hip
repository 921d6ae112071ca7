// The two attention steps of the OCR head (modeling/UNet3p_res_ocr_SNws.py of the reference) as fused kernels:
//   SpatialGather_Module.forward (:21-28)  ctx[k][c] = sum_p softmax_p(scale * probs[p][k]) * feats[p][c]
//   _ObjectAttentionBlock.forward (:87-100) out[p][c] = sum_k softmax_k(key_channels^-0.5 * q[p] . key[k]) * value[k][c]
// and their gradients.  Both contract a handful of classes (K <= 16) against millions of pixels, so they are streaming
// passes over the pixel tensors (bf16 NHWC) with every small matrix product on the matrix cores:
//   * a wave owns 16 consecutive pixels x all C channels per step; the 16-byte loads are lane = (pixel lane >> 3,
//     channel octet lane & 7) per 64-channel slab - whole 128-byte lines - and the step's pixels are parked in a
//     wave-private LDS tile [16 px][C] (rows padded by 16 bytes);
//   * products that contract channels (q . key, dO . value, feats . dctx) read the tile as v_mfma_f32_16x16x32_bf16 B
//     fragments (pixel lane & 15, channel octet lane >> 4); products that contract classes (P . value, dS . key,
//     P . dctx) take the soft-max output straight from the accumulator layout (pixel lane & 15, classes 4 * (lane >>
//     4) ..) as the B operand of v_mfma_f32_16x16x16_bf16; products that contract pixels (P^T . dO, dS^T . q, P^T .
//     feats) read both operands pixel-major with the transposing LDS read;
//   * per-pixel results return through the same tile to whole-line stores; per-class results are per-wave partial sums
//     (summed afterwards in a fixed order: bit-reproducible).
// The soft-max runs in float32 on the accumulators; probabilities enter the second product rounded to bf16.
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4v;

__device__ __forceinline__ void wave_lds_sync() {  // LDS traffic of one wave is in order; this keeps the compiler in order too
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ short4v lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
}

struct LaneGeo {
  int r, g;        // MFMA geometry: row / column lane & 15, k group lane >> 4
  int oct, prow;   // streaming geometry: channel octet lane & 7 of a 64-channel slab, pixel row lane >> 3 (+ 8)
  int tq, tp;      // transposing read: lane addresses k-row 4g + tq, elements 4tp ..
};
__device__ __forceinline__ LaneGeo lane_geo() {
  const int lane = threadIdx.x & 63;
  LaneGeo L;
  L.r = lane & 15; L.g = lane >> 4; L.oct = lane & 7; L.prow = lane >> 3; L.tq = (lane >> 2) & 3; L.tp = lane & 3;
  return L;
}

__device__ __forceinline__ uint2 pack4(float a, float b, float c, float d) {
  return make_uint2(pack_bf16x2(a, b), pack_bf16x2(c, d));
}
__device__ __forceinline__ short4v as_frag(uint2 v) {
  short4v f;
  f[0] = (short)(v.x & 0xffff); f[1] = (short)(v.x >> 16); f[2] = (short)(v.y & 0xffff); f[3] = (short)(v.y >> 16);
  return f;
}

// [16 rows][C] bf16 in LDS, rows RS bytes apart (16 bytes of padding: every access pattern below is conflict-free or 2-way)
template <int C>
struct Tile {
  static constexpr int NSL = C / 64, RS = 2 * C + 16, BYTES = 16 * RS, NV = 2 * NSL;
  // 16 pixels q0 .. q0+15 of an image (rows past p1 repeat the last valid pixel)
  static __device__ __forceinline__ void load(const bf16_t* img, int64_t ld, int64_t q0, int64_t p1, const LaneGeo& L, uint4 (&v)[NV]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int64_t px = q0 + L.prow + 8 * jj;
      const bf16_t* p = img + (px < p1 ? px : p1 - 1) * ld + 8 * L.oct;
#pragma unroll
      for (int j = 0; j < NSL; ++j) v[2 * j + jj] = *reinterpret_cast<const uint4*>(p + 64 * j);
    }
  }
  static __device__ __forceinline__ void stash(unsigned char* t, const LaneGeo& L, const uint4 (&v)[NV]) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int j = 0; j < NSL; ++j) *reinterpret_cast<uint4*>(t + (L.prow + 8 * jj) * RS + j * 128 + L.oct * 16) = v[2 * j + jj];
  }
  // the same through ONE vector value (4 * NV dwords) instead of an array: a loop-carried array of loads does not
  // get promoted to registers by hipcc, a vector value is never in memory
  typedef unsigned vec_t __attribute__((ext_vector_type(4 * NV)));
  static __device__ __forceinline__ vec_t load_vec(const bf16_t* img, int64_t ld, int64_t q0, int64_t p1, const LaneGeo& L) {
    vec_t v;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int64_t px = q0 + L.prow + 8 * jj;
      const bf16_t* p = img + (px < p1 ? px : p1 - 1) * ld + 8 * L.oct;
#pragma unroll
      for (int j = 0; j < NSL; ++j) {
        const uint4 t = *reinterpret_cast<const uint4*>(p + 64 * j);
        v[4 * (2 * j + jj)] = t.x; v[4 * (2 * j + jj) + 1] = t.y; v[4 * (2 * j + jj) + 2] = t.z; v[4 * (2 * j + jj) + 3] = t.w;
      }
    }
    return v;
  }
  static __device__ __forceinline__ void stash_vec(unsigned char* t, const LaneGeo& L, const vec_t v) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
      for (int j = 0; j < NSL; ++j)
        *reinterpret_cast<uint4*>(t + (L.prow + 8 * jj) * RS + j * 128 + L.oct * 16) =
            make_uint4(v[4 * (2 * j + jj)], v[4 * (2 * j + jj) + 1], v[4 * (2 * j + jj) + 2], v[4 * (2 * j + jj) + 3]);
  }
  // 16x16x32 operand: row lane & 15, channels 32 ks + 8 (lane >> 4) ..
  static __device__ __forceinline__ bf16x8 frag(const unsigned char* t, const LaneGeo& L, int ks) {
    return *reinterpret_cast<const bf16x8*>(t + L.r * RS + ks * 64 + L.g * 16);
  }
  // 16x16x16 operand, transposed: channel 16 cb + (lane & 15), rows 4 (lane >> 4) ..
  static __device__ __forceinline__ short4v tr(const unsigned char* t, const LaneGeo& L, int cb) {
    return lds_tr16(t + (4 * L.g + L.tq) * RS + (16 * cb + 4 * L.tp) * 2);
  }
  // accumulator D[channel 16 cb + 4 (lane >> 4) + i][row lane & 15] -> the tile
  static __device__ __forceinline__ void put4(unsigned char* t, const LaneGeo& L, int cb, const f32x4 d) {
    *reinterpret_cast<uint2*>(t + L.r * RS + (16 * cb + 4 * L.g) * 2) = pack4(d[0], d[1], d[2], d[3]);
  }
  static __device__ __forceinline__ void store(bf16_t* img, int64_t ld, int64_t q0, int64_t p1, const LaneGeo& L, const unsigned char* t) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int64_t px = q0 + L.prow + 8 * jj;
      if (px < p1) {
        bf16_t* p = img + px * ld + 8 * L.oct;
#pragma unroll
        for (int j = 0; j < NSL; ++j)
          *reinterpret_cast<uint4*>(p + 64 * j) = *reinterpret_cast<const uint4*>(t + (L.prow + 8 * jj) * RS + j * 128 + L.oct * 16);
      }
    }
  }
  static __device__ __forceinline__ void store_add(bf16_t* img, int64_t ld, int64_t q0, int64_t p1, const LaneGeo& L, const unsigned char* t) {
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
      const int64_t px = q0 + L.prow + 8 * jj;
      if (px < p1) {
        bf16_t* p = img + px * ld + 8 * L.oct;
#pragma unroll
        for (int j = 0; j < NSL; ++j)
          *reinterpret_cast<uint4*>(p + 64 * j) = add_bf16x8(*reinterpret_cast<const uint4*>(t + (L.prow + 8 * jj) * RS + j * 128 + L.oct * 16),
                                                             *reinterpret_cast<const uint4*>(p + 64 * j));
      }
    }
  }
  // a dense [16][C] bf16 matrix (class rows) into a tile, by the whole workgroup
  static __device__ __forceinline__ void fill(unsigned char* t, const bf16_t* src) {
    for (int i = threadIdx.x; i < 16 * (C / 8); i += blockDim.x) {
      const int row = i / (C / 8), ch = i - row * (C / 8);
      *reinterpret_cast<uint4*>(t + row * RS + ch * 16) = *reinterpret_cast<const uint4*>(src + row * C + ch * 8);
    }
  }
};

// [16 px][16 classes] bf16: written from the accumulator layout, read back transposed (class lane & 15, pixels 4g ..)
__device__ __forceinline__ void small_put(unsigned char* t, const LaneGeo& L, uint2 v) {
  *reinterpret_cast<uint2*>(t + L.r * 32 + 8 * L.g) = v;
}
__device__ __forceinline__ short4v small_tr(const unsigned char* t, const LaneGeo& L) {
  return lds_tr16(t + (4 * L.g + L.tq) * 32 + 8 * L.tp);
}

__device__ __forceinline__ float xg_max(float v) {   // over the four k groups that share a pixel
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xg_sum(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

constexpr int OCR_WAVES = 4, OCR_THREADS = 64 * OCR_WAVES;

// this wave's 16-pixel steps: first pixel p0, end p1, distance qs between consecutive steps.  il == 0: a contiguous run
// inside the workgroup's pixel tile.  il != 0: the 16-pixel steps of the image are dealt round-robin to the (workgroup,
// wave) pairs, so the resident workgroups sweep one compact window of each tensor instead of far-apart runs (DRAM
// page locality); per-workgroup partial results then belong to a strided pixel set - their consumers only sum them.
__device__ __forceinline__ void wave_run(int t, int PT, int64_t HW, int wave, int il, int64_t& p0, int64_t& p1, int64_t& qs) {
  if (il) {
    p0 = ((int64_t)t * OCR_WAVES + wave) * 16;
    p1 = HW;
    qs = (int64_t)gridDim.x * OCR_WAVES * 16;
    return;
  }
  const int64_t b0 = (int64_t)t * PT;
  const int64_t b1 = b0 + PT < HW ? b0 + PT : HW;
  const int64_t per = ((b1 - b0 + OCR_WAVES * 16 - 1) / (OCR_WAVES * 16)) * 16;
  p0 = b0 + wave * per;
  p1 = p0 + per < b1 ? p0 + per : b1;
  qs = 16;
}
// SNWS_SN_INTERLEAVE = 0: contiguous runs (A/B measurements)
inline int ocr_interleave() {
  static const int on = !(getenv("SNWS_SN_INTERLEAVE") && atoi(getenv("SNWS_SN_INTERLEAVE")) == 0);
  return on;
}

// ------------------------------------------------------------------------------------------------ object attention
struct AttnArgs {
  const bf16_t* q;      // [B][HW][ldq]
  const bf16_t* dout;   // backward: [B][HW][lddo]
  const bf16_t* key;    // [B][16][C], rows >= K zero
  const bf16_t* val;    // [B][16][C]
  bf16_t* out;          // forward: context [B][HW][ldo]; backward: dq [B][HW][ldo]
  float* dkey_part;     // backward: [B][Tn][16][C]
  float* dval_part;
  int64_t ldq, lddo, ldo, HW;
  int K, PT, Tn, il;
  float scale;
};

// soft-max over the classes of each pixel from S[class 4g + i][pixel r]; classes >= K get probability 0
__device__ __forceinline__ void class_softmax(const f32x4 S, float scale, int K, const LaneGeo& L, float (&p)[4]) {
  float s[4], m = -INFINITY;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    s[i] = 4 * L.g + i < K ? S[i] * scale : -INFINITY;
    m = fmaxf(m, s[i]);
  }
  m = xg_max(m);
  float z = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    p[i] = 4 * L.g + i < K ? __expf(s[i] - m) : 0.f;
    z += p[i];
  }
  z = 1.f / xg_sum(z);
#pragma unroll
  for (int i = 0; i < 4; ++i) p[i] *= z;
}

template <int C>
__global__ __launch_bounds__(OCR_THREADS) void ocr_attend_fwd_kernel(AttnArgs a) {
  using T = Tile<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* key_t = smem;
  unsigned char* val_t = smem + T::BYTES;
  const int t = blockIdx.x, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned char* tile = smem + (2 + wave) * T::BYTES;
  const LaneGeo L = lane_geo();
  T::fill(key_t, a.key + (int64_t)b * 16 * C);
  T::fill(val_t, a.val + (int64_t)b * 16 * C);
  __syncthreads();
  bf16x8 kf[C / 32];        // A of S = key . q: class row lane & 15
  short4v vT[C / 16];       // A of out = value^T . P: channel row, classes 4g ..
#pragma unroll
  for (int ks = 0; ks < C / 32; ++ks) kf[ks] = T::frag(key_t, L, ks);
#pragma unroll
  for (int cb = 0; cb < C / 16; ++cb) vT[cb] = T::tr(val_t, L, cb);

  int64_t p0, p1, qs;
  wave_run(t, a.PT, a.HW, wave, a.il, p0, p1, qs);
  const bf16_t* qi = a.q + (int64_t)b * a.HW * a.ldq;
  bf16_t* oi = a.out + (int64_t)b * a.HW * a.ldo;
  if (p0 >= p1) return;
  typename T::vec_t v = T::load_vec(qi, a.ldq, p0, p1, L);
  for (int64_t q0 = p0; q0 < p1; q0 += qs) {
    T::stash_vec(tile, L, v);
    v = T::load_vec(qi, a.ldq, q0 + qs, p1, L);  // the next step's pixels, in flight while this step computes (clamped past the run)
    wave_lds_sync();
    f32x4 S = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[ks], T::frag(tile, L, ks), S, 0, 0, 0);
    float p[4];
    class_softmax(S, a.scale, a.K, L, p);
    const short4v pf = as_frag(pack4(p[0], p[1], p[2], p[3]));
    wave_lds_sync();          // every fragment of q is read: the tile turns into the output tile
#pragma unroll
    for (int cb = 0; cb < C / 16; ++cb)
      T::put4(tile, L, cb, __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(vT[cb], pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0));
    wave_lds_sync();
    T::store(oi, a.ldo, q0, p1, L, tile);
    wave_lds_sync();
  }
}

template <int C>
__global__ __launch_bounds__(OCR_THREADS) void ocr_attend_bwd_kernel(AttnArgs a) {
  using T = Tile<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* key_t = smem;
  unsigned char* val_t = smem + T::BYTES;
  const int t = blockIdx.x, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  constexpr int PER_WAVE = 2 * T::BYTES + 1024;
  unsigned char* q_t = smem + 2 * T::BYTES + wave * PER_WAVE;
  unsigned char* do_t = q_t + T::BYTES;
  unsigned char* p_t = do_t + T::BYTES;        // P, then dS behind it: [16 px][16 classes] bf16 each
  unsigned char* ds_t = p_t + 512;
  const LaneGeo L = lane_geo();
  T::fill(key_t, a.key + (int64_t)b * 16 * C);
  T::fill(val_t, a.val + (int64_t)b * 16 * C);
  __syncthreads();

  f32x4 dV[C / 16], dK[C / 16];
#pragma unroll
  for (int cb = 0; cb < C / 16; ++cb) dV[cb] = dK[cb] = f32x4{0.f, 0.f, 0.f, 0.f};

  int64_t p0, p1, qs;
  wave_run(t, a.PT, a.HW, wave, a.il, p0, p1, qs);
  const bf16_t* qi = a.q + (int64_t)b * a.HW * a.ldq;
  const bf16_t* di = a.dout + (int64_t)b * a.HW * a.lddo;
  bf16_t* oi = a.out + (int64_t)b * a.HW * a.ldo;
  for (int64_t q0 = p0; q0 < p1; q0 += qs) {
    {
      uint4 v[T::NV], w[T::NV];
      T::load(qi, a.ldq, q0, p1, L, v);
      T::load(di, a.lddo, q0, p1, L, w);
      T::stash(q_t, L, v);
      T::stash(do_t, L, w);
    }
    wave_lds_sync();
    f32x4 S = {0.f, 0.f, 0.f, 0.f}, G = {0.f, 0.f, 0.f, 0.f};     // S = key . q, G = value . dO, both [class][pixel]
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) {
      S = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::frag(key_t, L, ks), T::frag(q_t, L, ks), S, 0, 0, 0);
      G = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::frag(val_t, L, ks), T::frag(do_t, L, ks), G, 0, 0, 0);
    }
    float p[4], ds[4];
    class_softmax(S, a.scale, a.K, L, p);
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) dot = fmaf(p[i], G[i], dot);
    dot = xg_sum(dot);
    const bool live = q0 + L.r < p1;            // pixels past the run contribute nothing to the per-class sums
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      ds[i] = live ? p[i] * (G[i] - dot) * a.scale : 0.f;
      p[i] = live ? p[i] : 0.f;
    }
    const uint2 dsp = pack4(ds[0], ds[1], ds[2], ds[3]);
    small_put(p_t, L, pack4(p[0], p[1], p[2], p[3]));
    small_put(ds_t, L, dsp);
    wave_lds_sync();
    const short4v pT = small_tr(p_t, L), dsT = small_tr(ds_t, L);   // class row lane & 15, pixels 4g ..
#pragma unroll
    for (int cb = 0; cb < C / 16; ++cb) {
      dV[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pT, T::tr(do_t, L, cb), dV[cb], 0, 0, 0);
      dK[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(dsT, T::tr(q_t, L, cb), dK[cb], 0, 0, 0);
    }
    wave_lds_sync();          // q is spent: its tile carries dq = key^T . dS out
    const short4v dsf = as_frag(dsp);
#pragma unroll
    for (int cb = 0; cb < C / 16; ++cb)
      T::put4(q_t, L, cb, __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(T::tr(key_t, L, cb), dsf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0));
    wave_lds_sync();
    T::store(oi, a.ldo, q0, p1, L, q_t);
    wave_lds_sync();
  }
  // D[class 4g + i][channel 16 cb + r]
  const int64_t row0 = ((int64_t)b * a.Tn + (int64_t)t * OCR_WAVES + wave) * 16;
#pragma unroll
  for (int cb = 0; cb < C / 16; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a.dval_part[(row0 + 4 * L.g + i) * C + 16 * cb + L.r] = dV[cb][i];
      a.dkey_part[(row0 + 4 * L.g + i) * C + 16 * cb + L.r] = dK[cb][i];
    }
}

// ------------------------------------------------------------------------------------------------ spatial gather
// per (image, class): max and sum of exp over a run of pixels, merged on line (m, s) -> (max(m, v), s e^(m-M) + e^(v-M))
__global__ __launch_bounds__(256) void ocr_gather_stats_kernel(const bf16_t* __restrict__ probs, int64_t ldp, int64_t HW, int K,
                                                               float scale, int PT, int Tn, float* __restrict__ part) {
  __shared__ float red[4][16][2];
  const int t = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b0 = (int64_t)t * PT;
  const int64_t b1 = b0 + PT < HW ? b0 + PT : HW;
  float m[16], s[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) { m[k] = -INFINITY; s[k] = 0.f; }
  const bf16_t* pi = probs + (int64_t)b * HW * ldp;
  for (int64_t px = b0 + tid; px < b1; px += 256) {
    const uint4 v0 = *reinterpret_cast<const uint4*>(pi + px * ldp), v1 = *reinterpret_cast<const uint4*>(pi + px * ldp + 8);
    const unsigned u[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const float v = scale * ((k & 1) ? __uint_as_float(u[k >> 1] & 0xffff0000u) : __uint_as_float(u[k >> 1] << 16));
      const float mn = fmaxf(m[k], v);
      s[k] = s[k] * __expf(m[k] - mn) + __expf(v - mn);
      m[k] = mn;
    }
  }
#pragma unroll
  for (int k = 0; k < 16; ++k) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float mo = __shfl_xor(m[k], o, 64), so = __shfl_xor(s[k], o, 64);
      const float mn = fmaxf(m[k], mo);
      // (-inf) - (-inf) would be NaN: an empty side has s = 0 and is skipped by the guard
      const float a_ = m[k] == mn ? s[k] : s[k] * __expf(m[k] - mn), b_ = mo == mn ? so : so * __expf(mo - mn);
      s[k] = a_ + b_;
      m[k] = mn;
    }
    if (lane == 0) { red[wave][k][0] = m[k]; red[wave][k][1] = s[k]; }
  }
  __syncthreads();
  if (tid < 16) {
    float M = -INFINITY;
    for (int w = 0; w < 4; ++w) M = fmaxf(M, red[w][tid][0]);
    float S = 0.f;
    for (int w = 0; w < 4; ++w) S += red[w][tid][0] == M ? red[w][tid][1] : red[w][tid][1] * __expf(red[w][tid][0] - M);
    float* o = part + (((int64_t)b * Tn + t) * 16 + tid) * 2;
    o[0] = tid < K ? M : 0.f;
    o[1] = tid < K ? S : 1.f;
  }
}

// merge an image's Tn partial (max, sum) pairs: stat[k] = (max, 1 / sum); by threads 0 .. 15 of a workgroup
__device__ __forceinline__ void merge_stats(const float* part_b, int Tn, float (*stat)[2]) {
  const int k = threadIdx.x;
  if (k < 16) {
    float M = -INFINITY;
    for (int i = 0; i < Tn; ++i) M = fmaxf(M, part_b[(i * 16 + k) * 2]);
    float S = 0.f;
    for (int i = 0; i < Tn; ++i) S += part_b[(i * 16 + k) * 2 + 1] * __expf(part_b[(i * 16 + k) * 2] - M);
    stat[k][0] = M;
    stat[k][1] = 1.f / S;
  }
}

struct GatherArgs {
  const bf16_t* feats;    // [B][HW][ldf]
  const bf16_t* probs;    // [B][HW][ldp], classes >= K ignored
  const float* stat_part; // [B][Ts][16][2] from ocr_gather_stats_kernel
  float* ctx_part;        // forward: [B][Tn][16][C]
  const bf16_t* dctx;     // backward: [B][16][C] bf16, rows >= K zero
  const float* dot;       // backward: [B][16] = <ctx[k], dctx[k]>
  bf16_t* dfeats;         // backward: [B][HW][lddf]
  bf16_t* dprobs;         // backward: [B][HW][16]
  int64_t ldf, ldp, lddf, HW;
  int K, PT, Tn, Ts, il;
  float scale;
  int accumulate;         // backward: dfeats += (it holds the other consumers' shares of the gradient)
};

// probabilities of this lane's four classes at pixel px (0 past the run or past K)
__device__ __forceinline__ void pixel_probs(const GatherArgs& a, const bf16_t* pi, int64_t px, int64_t p1, const LaneGeo& L,
                                            const float (*stat)[2], float (&p)[4]) {
  const uint2 v = *reinterpret_cast<const uint2*>(pi + (px < p1 ? px : p1 - 1) * a.ldp + 4 * L.g);
  const float f[4] = {__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                      __uint_as_float(v.y & 0xffff0000u)};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int k = 4 * L.g + i;
    p[i] = (k < a.K && px < p1) ? __expf(a.scale * f[i] - stat[k][0]) * stat[k][1] : 0.f;
  }
}

template <int C>
__global__ __launch_bounds__(OCR_THREADS) void ocr_gather_fwd_kernel(GatherArgs a) {
  using T = Tile<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = blockIdx.x, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float (*stat)[2] = reinterpret_cast<float (*)[2]>(smem);
  unsigned char* tile = smem + 128 + wave * (T::BYTES + 512);
  unsigned char* p_t = tile + T::BYTES;
  const LaneGeo L = lane_geo();
  merge_stats(a.stat_part + (int64_t)b * a.Ts * 32, a.Ts, stat);
  __syncthreads();
  f32x4 acc[C / 16];
#pragma unroll
  for (int cb = 0; cb < C / 16; ++cb) acc[cb] = f32x4{0.f, 0.f, 0.f, 0.f};
  int64_t p0, p1, qs;
  wave_run(t, a.PT, a.HW, wave, a.il, p0, p1, qs);
  const bf16_t* fi = a.feats + (int64_t)b * a.HW * a.ldf;
  const bf16_t* pi = a.probs + (int64_t)b * a.HW * a.ldp;
  for (int64_t q0 = p0; q0 < p1; q0 += qs) {
    {
      uint4 v[T::NV];
      T::load(fi, a.ldf, q0, p1, L, v);
      T::stash(tile, L, v);
    }
    float p[4];
    pixel_probs(a, pi, q0 + L.r, p1, L, stat, p);
    small_put(p_t, L, pack4(p[0], p[1], p[2], p[3]));
    wave_lds_sync();
    const short4v pT = small_tr(p_t, L);
#pragma unroll
    for (int cb = 0; cb < C / 16; ++cb) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(pT, T::tr(tile, L, cb), acc[cb], 0, 0, 0);
    wave_lds_sync();
  }
  const int64_t row0 = ((int64_t)b * a.Tn + (int64_t)t * OCR_WAVES + wave) * 16;
#pragma unroll
  for (int cb = 0; cb < C / 16; ++cb)
#pragma unroll
    for (int i = 0; i < 4; ++i) a.ctx_part[(row0 + 4 * L.g + i) * C + 16 * cb + L.r] = acc[cb][i];
}

template <int C>
__global__ __launch_bounds__(OCR_THREADS) void ocr_gather_bwd_kernel(GatherArgs a) {
  using T = Tile<C>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int t = blockIdx.x, b = blockIdx.y;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float (*stat)[2] = reinterpret_cast<float (*)[2]>(smem);
  float* dots = reinterpret_cast<float*>(smem + 128);
  unsigned char* dctx_t = smem + 256;
  unsigned char* tile = smem + 256 + (1 + wave) * T::BYTES;
  const LaneGeo L = lane_geo();
  merge_stats(a.stat_part + (int64_t)b * a.Ts * 32, a.Ts, stat);
  if (threadIdx.x < 16) dots[threadIdx.x] = a.dot[b * 16 + threadIdx.x];
  T::fill(dctx_t, a.dctx + (int64_t)b * 16 * C);
  __syncthreads();
  int64_t p0, p1, qs;
  wave_run(t, a.PT, a.HW, wave, a.il, p0, p1, qs);
  const bf16_t* fi = a.feats + (int64_t)b * a.HW * a.ldf;
  const bf16_t* pi = a.probs + (int64_t)b * a.HW * a.ldp;
  bf16_t* dfi = a.dfeats + (int64_t)b * a.HW * a.lddf;
  bf16_t* dpi = a.dprobs + (int64_t)b * a.HW * 16;
  for (int64_t q0 = p0; q0 < p1; q0 += qs) {
    {
      uint4 v[T::NV];
      T::load(fi, a.ldf, q0, p1, L, v);
      T::stash(tile, L, v);
    }
    float p[4];
    pixel_probs(a, pi, q0 + L.r, p1, L, stat, p);
    wave_lds_sync();
    f32x4 D = {0.f, 0.f, 0.f, 0.f};       // D[class][pixel] = dctx[class] . feats[pixel]
#pragma unroll
    for (int ks = 0; ks < C / 32; ++ks) D = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::frag(dctx_t, L, ks), T::frag(tile, L, ks), D, 0, 0, 0);
    // d probs = scale * P * (feats . dctx - <ctx, dctx>)
    if (q0 + L.r < p1) {
      float d[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = a.scale * p[i] * (D[i] - dots[4 * L.g + i]);
      *reinterpret_cast<uint2*>(dpi + (q0 + L.r) * 16 + 4 * L.g) = pack4(d[0], d[1], d[2], d[3]);
    }
    const short4v pf = as_frag(pack4(p[0], p[1], p[2], p[3]));
    wave_lds_sync();          // feats is spent: its tile carries d feats = dctx^T . P out
#pragma unroll
    for (int cb = 0; cb < C / 16; ++cb)
      T::put4(tile, L, cb, __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(T::tr(dctx_t, L, cb), pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0));
    wave_lds_sync();
    if (a.accumulate) T::store_add(dfi, a.lddf, q0, p1, L, tile);   // kernel-uniform
    else T::store(dfi, a.lddf, q0, p1, L, tile);
    wave_lds_sync();
  }
}

template <typename K>
int set_lds(K kernel, int bytes, const char* fn, PerDeviceOnce& once) {
  std::atomic<bool>* const flag = once.flag();
  if (!flag || !flag->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) {
      snws_set_error("%s: hipFuncSetAttribute failed: %s", fn, hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (flag) flag->store(true, std::memory_order_release);
  }
  return SNWS_OK;
}

constexpr int ATTN_C = 256, GATHER_C = 512;
constexpr int ATTN_PT = 1024, GATHER_PT = 1024, STATS_PT = 8192;

}  // namespace

extern "C" int snws_ocr_tiles(int64_t HW, int kind) {
  if (kind == 0) return (int)cdiv64(HW, STATS_PT);                      // rows of the gather statistics per image
  if (kind == 1) return (int)cdiv64(HW, GATHER_PT) * OCR_WAVES;         // partial rows of snws_ocr_gather_fwd per image
  return (int)cdiv64(HW, ATTN_PT) * OCR_WAVES;                          // partial rows of snws_ocr_attend_bwd per image
}

extern "C" int snws_ocr_attend_fwd(const void* q, int64_t ldq, const void* key, const void* val, void* out, int64_t ldo, int B,
                                   int64_t HW, int C, int K, float scale, void* stream) {
  if (C != ATTN_C) SNWS_UNSUPPORTED("snws_ocr_attend_fwd: %d key channels (built for %d)", C, ATTN_C);
  SNWS_CHECK_ARG(q && key && val && out, "snws_ocr_attend_fwd: NULL pointer");
  SNWS_CHECK_ARG(B >= 1 && B <= 65535 && HW >= 1 && K >= 1 && K <= 16 && ldq >= C && ldq % 8 == 0 && ldo >= C && ldo % 8 == 0,
                 "snws_ocr_attend_fwd: bad sizes");
  SNWS_CHECK_ARG((((uintptr_t)q | (uintptr_t)key | (uintptr_t)val | (uintptr_t)out) & 15) == 0, "snws_ocr_attend_fwd: unaligned pointers");
  using T = Tile<ATTN_C>;
  constexpr int LDS = (2 + OCR_WAVES) * T::BYTES;
  static PerDeviceOnce once;
  if (int rc = set_lds(&ocr_attend_fwd_kernel<ATTN_C>, LDS, "snws_ocr_attend_fwd", once)) return rc;
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.key = (const bf16_t*)key; a.val = (const bf16_t*)val; a.out = (bf16_t*)out;
  a.ldq = ldq; a.ldo = ldo; a.HW = HW; a.K = K; a.PT = ATTN_PT; a.scale = scale; a.il = ocr_interleave();
  ocr_attend_fwd_kernel<ATTN_C><<<dim3((unsigned)cdiv64(HW, ATTN_PT), B), OCR_THREADS, LDS, (hipStream_t)stream>>>(a);
  SNWS_LAUNCH_CHECK("snws_ocr_attend_fwd");
  return SNWS_OK;
}

extern "C" int snws_ocr_attend_bwd(const void* q, int64_t ldq, const void* dout, int64_t lddo, const void* key, const void* val,
                                   void* dq, int64_t lddq, float* dkey_part, float* dval_part, int B, int64_t HW, int C, int K,
                                   float scale, void* stream) {
  if (C != ATTN_C) SNWS_UNSUPPORTED("snws_ocr_attend_bwd: %d key channels (built for %d)", C, ATTN_C);
  SNWS_CHECK_ARG(q && dout && key && val && dq && dkey_part && dval_part, "snws_ocr_attend_bwd: NULL pointer");
  SNWS_CHECK_ARG(B >= 1 && B <= 65535 && HW >= 1 && K >= 1 && K <= 16 && ldq >= C && ldq % 8 == 0 && lddo >= C && lddo % 8 == 0 &&
                     lddq >= C && lddq % 8 == 0, "snws_ocr_attend_bwd: bad sizes");
  SNWS_CHECK_ARG((((uintptr_t)q | (uintptr_t)dout | (uintptr_t)key | (uintptr_t)val | (uintptr_t)dq) & 15) == 0,
                 "snws_ocr_attend_bwd: unaligned pointers");
  using T = Tile<ATTN_C>;
  constexpr int LDS = 2 * T::BYTES + OCR_WAVES * (2 * T::BYTES + 1024);
  static PerDeviceOnce once;
  if (int rc = set_lds(&ocr_attend_bwd_kernel<ATTN_C>, LDS, "snws_ocr_attend_bwd", once)) return rc;
  AttnArgs a{};
  a.q = (const bf16_t*)q; a.dout = (const bf16_t*)dout; a.key = (const bf16_t*)key; a.val = (const bf16_t*)val; a.out = (bf16_t*)dq;
  a.dkey_part = dkey_part; a.dval_part = dval_part;
  a.ldq = ldq; a.lddo = lddo; a.ldo = lddq; a.HW = HW; a.K = K; a.PT = ATTN_PT; a.Tn = snws_ocr_tiles(HW, 2); a.scale = scale; a.il = ocr_interleave();
  ocr_attend_bwd_kernel<ATTN_C><<<dim3((unsigned)cdiv64(HW, ATTN_PT), B), OCR_THREADS, LDS, (hipStream_t)stream>>>(a);
  SNWS_LAUNCH_CHECK("snws_ocr_attend_bwd");
  return SNWS_OK;
}

extern "C" int snws_ocr_gather_stats(const void* probs, int64_t ldp, float* stat_part, int B, int64_t HW, int K, float scale,
                                     void* stream) {
  SNWS_CHECK_ARG(probs && stat_part, "snws_ocr_gather_stats: NULL pointer");
  SNWS_CHECK_ARG(B >= 1 && B <= 65535 && HW >= 1 && K >= 1 && K <= 16 && ldp >= 16 && ldp % 8 == 0 && ((uintptr_t)probs & 15) == 0,
                 "snws_ocr_gather_stats: bad sizes (class rows of 16 bf16, 16-byte aligned)");
  const int Ts = snws_ocr_tiles(HW, 0);
  ocr_gather_stats_kernel<<<dim3(Ts, B), 256, 0, (hipStream_t)stream>>>((const bf16_t*)probs, ldp, HW, K, scale, STATS_PT, Ts, stat_part);
  SNWS_LAUNCH_CHECK("snws_ocr_gather_stats");
  return SNWS_OK;
}

extern "C" int snws_ocr_gather_fwd(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part,
                                   float* ctx_part, int B, int64_t HW, int C, int K, float scale, void* stream) {
  if (C != GATHER_C) SNWS_UNSUPPORTED("snws_ocr_gather_fwd: %d feature channels (built for %d)", C, GATHER_C);
  SNWS_CHECK_ARG(feats && probs && stat_part && ctx_part, "snws_ocr_gather_fwd: NULL pointer");
  SNWS_CHECK_ARG(B >= 1 && B <= 65535 && HW >= 1 && K >= 1 && K <= 16 && ldf >= C && ldf % 8 == 0 && ldp >= 16 && ldp % 4 == 0,
                 "snws_ocr_gather_fwd: bad sizes");
  SNWS_CHECK_ARG(((uintptr_t)feats & 15) == 0 && ((uintptr_t)probs & 7) == 0, "snws_ocr_gather_fwd: unaligned pointers");
  using T = Tile<GATHER_C>;
  constexpr int LDS = 128 + OCR_WAVES * (T::BYTES + 512);
  static PerDeviceOnce once;
  if (int rc = set_lds(&ocr_gather_fwd_kernel<GATHER_C>, LDS, "snws_ocr_gather_fwd", once)) return rc;
  GatherArgs a{};
  a.feats = (const bf16_t*)feats; a.probs = (const bf16_t*)probs; a.stat_part = stat_part; a.ctx_part = ctx_part;
  a.ldf = ldf; a.ldp = ldp; a.HW = HW; a.K = K; a.PT = GATHER_PT; a.Tn = snws_ocr_tiles(HW, 1); a.Ts = snws_ocr_tiles(HW, 0); a.il = ocr_interleave();
  a.scale = scale;
  ocr_gather_fwd_kernel<GATHER_C><<<dim3((unsigned)cdiv64(HW, GATHER_PT), B), OCR_THREADS, LDS, (hipStream_t)stream>>>(a);
  SNWS_LAUNCH_CHECK("snws_ocr_gather_fwd");
  return SNWS_OK;
}

namespace {
int run_ocr_gather_bwd(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part,
                       const void* dctx, const float* dot, void* dfeats, int64_t lddf, void* dprobs, int B, int64_t HW,
                       int C, int K, float scale, int accumulate, void* stream) {
  if (C != GATHER_C) SNWS_UNSUPPORTED("snws_ocr_gather_bwd: %d feature channels (built for %d)", C, GATHER_C);
  SNWS_CHECK_ARG(feats && probs && stat_part && dctx && dot && dfeats && dprobs, "snws_ocr_gather_bwd: NULL pointer");
  SNWS_CHECK_ARG(B >= 1 && B <= 65535 && HW >= 1 && K >= 1 && K <= 16 && ldf >= C && ldf % 8 == 0 && lddf >= C && lddf % 8 == 0 &&
                     ldp >= 16 && ldp % 4 == 0, "snws_ocr_gather_bwd: bad sizes");
  SNWS_CHECK_ARG((((uintptr_t)feats | (uintptr_t)dctx | (uintptr_t)dfeats) & 15) == 0 && (((uintptr_t)probs | (uintptr_t)dprobs) & 7) == 0,
                 "snws_ocr_gather_bwd: unaligned pointers");
  using T = Tile<GATHER_C>;
  constexpr int LDS = 256 + (1 + OCR_WAVES) * T::BYTES;
  static PerDeviceOnce once;
  if (int rc = set_lds(&ocr_gather_bwd_kernel<GATHER_C>, LDS, "snws_ocr_gather_bwd", once)) return rc;
  GatherArgs a{};
  a.feats = (const bf16_t*)feats; a.probs = (const bf16_t*)probs; a.stat_part = stat_part; a.dctx = (const bf16_t*)dctx; a.dot = dot;
  a.dfeats = (bf16_t*)dfeats; a.dprobs = (bf16_t*)dprobs;
  a.ldf = ldf; a.ldp = ldp; a.lddf = lddf; a.HW = HW; a.K = K; a.PT = GATHER_PT; a.Ts = snws_ocr_tiles(HW, 0); a.scale = scale; a.il = ocr_interleave();
  a.accumulate = accumulate;
  ocr_gather_bwd_kernel<GATHER_C><<<dim3((unsigned)cdiv64(HW, GATHER_PT), B), OCR_THREADS, LDS, (hipStream_t)stream>>>(a);
  SNWS_LAUNCH_CHECK("snws_ocr_gather_bwd");
  return SNWS_OK;
}
}  // namespace

extern "C" int snws_ocr_gather_bwd(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part,
                                   const void* dctx, const float* dot, void* dfeats, int64_t lddf, void* dprobs, int B, int64_t HW,
                                   int C, int K, float scale, void* stream) {
  return run_ocr_gather_bwd(feats, ldf, probs, ldp, stat_part, dctx, dot, dfeats, lddf, dprobs, B, HW, C, K, scale, 0, stream);
}

extern "C" int snws_ocr_gather_bwd_acc(const void* feats, int64_t ldf, const void* probs, int64_t ldp, const float* stat_part,
                                       const void* dctx, const float* dot, void* dfeats, int64_t lddf, void* dprobs, int B,
                                       int64_t HW, int C, int K, float scale, void* stream) {
  return run_ocr_gather_bwd(feats, ldf, probs, ldp, stat_part, dctx, dot, dfeats, lddf, dprobs, B, HW, C, K, scale, 1, stream);
}
