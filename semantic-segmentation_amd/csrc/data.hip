// Input-side tile preparation on the device (replaces the numpy lines of CLOUDdataset.__getitem__,
// dataloaders/datasets/cloud_dataset.py:127-134 of the reference): per-tile min-max scaling to [0, 255] with NaN -> 0,
// and the label remap NaN -> 255, 10 -> 0, 255 -> 0.  HBM-bound streaming kernels: 2 reads + 1 write of the tile.
// The arithmetic is the reference's, operation by operation in float32 ((x - min) / (max - min), then * 255), so the
// result is bit-identical to numpy's; min and max do not depend on the reduction order.
#include "common.h"

namespace {

constexpr int DT_THREADS = 256;

__device__ __forceinline__ float nan_to_zero(float v) { return v != v ? 0.f : v; }

// partial min / max of one tile: grid (blocks_per_tile, n_tiles); part[(tile * B + b) * 2 + {0, 1}]
__global__ __launch_bounds__(DT_THREADS) void tile_minmax_kernel(const float* __restrict__ x, int64_t elems, int B,
                                                                 float* __restrict__ part) {
  __shared__ float smin[DT_THREADS / 64], smax[DT_THREADS / 64];
  const int tile = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  const float* xt = x + (int64_t)tile * elems;
  float mn = INFINITY, mx = -INFINITY;
  const int64_t nvec = ((reinterpret_cast<uintptr_t>(xt) & 15) == 0) ? elems / 4 : 0;
  for (int64_t i = (int64_t)b * DT_THREADS + tid; i < nvec; i += (int64_t)B * DT_THREADS) {
    const float4 v = *reinterpret_cast<const float4*>(xt + 4 * i);
    const float a0 = nan_to_zero(v.x), a1 = nan_to_zero(v.y), a2 = nan_to_zero(v.z), a3 = nan_to_zero(v.w);
    mn = fminf(fminf(mn, a0), fminf(a1, fminf(a2, a3)));
    mx = fmaxf(fmaxf(mx, a0), fmaxf(a1, fmaxf(a2, a3)));
  }
  for (int64_t i = nvec * 4 + (int64_t)b * DT_THREADS + tid; i < elems; i += (int64_t)B * DT_THREADS) {
    const float a = nan_to_zero(xt[i]);
    mn = fminf(mn, a);
    mx = fmaxf(mx, a);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, o, 64));
    mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  }
  if ((tid & 63) == 0) { smin[tid >> 6] = mn; smax[tid >> 6] = mx; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < DT_THREADS / 64; ++w) { mn = fminf(mn, smin[w]); mx = fmaxf(mx, smax[w]); }
    part[((int64_t)tile * B + b) * 2] = mn;
    part[((int64_t)tile * B + b) * 2 + 1] = mx;
  }
}

template <typename T>
__device__ __forceinline__ void store_out(T* p, float v);
template <>
__device__ __forceinline__ void store_out<float>(float* p, float v) { *p = v; }
template <>
__device__ __forceinline__ void store_out<bf16_t>(bf16_t* p, float v) { *p = (bf16_t)v; }

template <typename T>
__global__ __launch_bounds__(DT_THREADS) void tile_scale_kernel(const float* __restrict__ x, T* __restrict__ y,
                                                                int64_t elems, int B, const float* __restrict__ part,
                                                                float* __restrict__ minmax_out) {
  __shared__ float s_mn, s_mx;
  const int tile = blockIdx.y, b = blockIdx.x, tid = threadIdx.x;
  if (tid < 64) {   // every block folds the tile's B partials itself (B <= 64; cheaper than a third launch)
    float mn = tid < B ? part[((int64_t)tile * B + tid) * 2] : INFINITY;
    float mx = tid < B ? part[((int64_t)tile * B + tid) * 2 + 1] : -INFINITY;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      mn = fminf(mn, __shfl_xor(mn, o, 64));
      mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    }
    if (tid == 0) {
      s_mn = mn;
      s_mx = mx;
      if (b == 0 && minmax_out) { minmax_out[2 * tile] = mn; minmax_out[2 * tile + 1] = mx; }
    }
  }
  __syncthreads();
  const float mn = s_mn, range = s_mx - s_mn;
  const float* xt = x + (int64_t)tile * elems;
  T* yt = y + (int64_t)tile * elems;
  for (int64_t i = (int64_t)b * DT_THREADS + tid; i < elems; i += (int64_t)B * DT_THREADS) {
    // cloud_dataset.py:130-131, one rounding per operation as numpy does: subtract, divide, multiply
    const float d = __fsub_rn(nan_to_zero(xt[i]), mn);
    const float q = __fdiv_rn(d, range);
    store_out<T>(yt + i, __fmul_rn(q, 255.0f));
  }
}

template <typename S>
__global__ __launch_bounds__(DT_THREADS) void label_remap_kernel(const S* __restrict__ src, float* __restrict__ dst,
                                                                 int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * DT_THREADS + threadIdx.x;
  if (i >= n) return;
  float v = (float)src[i];
  if (v != v) v = 255.f;                  // mask[np.isnan(mask)] = 255
  if (v == 10.f) v = 0.f;                 // mask[mask == 10] = 0
  if (v == 255.f) v = 0.f;                // mask[mask == 255] = 0
  dst[i] = v;
}


// Planar (N, C, H, W) images -> pixel-major (N, H, W, c_pad) in the compute dtype, channels C .. c_pad-1 zero: the layout
// the convolution kernels read, in one pass (the stock route is a dtype cast, a layout copy, a zero fill and a padded
// copy).  A workgroup takes 256 consecutive pixels of one image: a wave reads a channel's 256 pixels as one coalesced
// row (four pixels per lane - one 8- or 16-byte load where the alignment allows, VEC), the tile goes through LDS (row
// stride 257 floats: conflict-free both ways), and thread p writes pixel p's padded channels, so neighbouring threads
// write neighbouring bytes.
// 48 channels x 257 floats = 49 344 B: inside the 64 KB of dynamic LDS a kernel gets without an attribute call
constexpr int LAY_PIX = 256, LAY_MAXC = 48;
template <typename TI, typename TO, bool VEC>
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const TI* __restrict__ x, TO* __restrict__ y, int C, int64_t HW,
                                                               int c_pad) {
  extern __shared__ __attribute__((aligned(16))) float lay_tile[];     // [c_pad][LAY_PIX + 1]
  const int n = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t p0 = (int64_t)blockIdx.x * LAY_PIX;
  const TI* xn = x + (int64_t)n * C * HW;
  for (int c = wave; c < c_pad; c += 4) {
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
      const TI* src = xn + (int64_t)c * HW + p0 + 4 * lane;
      if (VEC && p0 + 4 * lane + 3 < HW) {
        if constexpr (sizeof(TI) == 2) {
          const uint2 q = *reinterpret_cast<const uint2*>(src);
          v[0] = __uint_as_float(q.x << 16); v[1] = __uint_as_float(q.x & 0xffff0000u);
          v[2] = __uint_as_float(q.y << 16); v[3] = __uint_as_float(q.y & 0xffff0000u);
        } else {
          const float4 q = *reinterpret_cast<const float4*>(src);
          v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (p0 + 4 * lane + e < HW) v[e] = (float)src[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) lay_tile[c * (LAY_PIX + 1) + 4 * lane + e] = v[e];
  }
  __syncthreads();
  if (p0 + tid < HW) {
    TO* o = y + ((int64_t)n * HW + p0 + tid) * c_pad;
    const float* t = lay_tile + tid;
    if constexpr (sizeof(TO) == 2) {
      for (int c = 0; c + 8 <= c_pad; c += 8) {       // c_pad is a multiple of 8 for 2-byte outputs (host check)
        uint4 q;
        q.x = pack_bf16x2(t[(c + 0) * (LAY_PIX + 1)], t[(c + 1) * (LAY_PIX + 1)]);
        q.y = pack_bf16x2(t[(c + 2) * (LAY_PIX + 1)], t[(c + 3) * (LAY_PIX + 1)]);
        q.z = pack_bf16x2(t[(c + 4) * (LAY_PIX + 1)], t[(c + 5) * (LAY_PIX + 1)]);
        q.w = pack_bf16x2(t[(c + 6) * (LAY_PIX + 1)], t[(c + 7) * (LAY_PIX + 1)]);
        *reinterpret_cast<uint4*>(o + c) = q;
      }
    } else {
      for (int c = 0; c + 4 <= c_pad; c += 4)
        *reinterpret_cast<float4*>(o + c) = make_float4(t[c * (LAY_PIX + 1)], t[(c + 1) * (LAY_PIX + 1)],
                                                        t[(c + 2) * (LAY_PIX + 1)], t[(c + 3) * (LAY_PIX + 1)]);
    }
  }
}

}  // namespace

extern "C" int snws_tile_minmax_blocks(int64_t elems_per_tile) {
  int64_t b = (elems_per_tile + 65535) / 65536;
  if (b < 1) b = 1;
  if (b > 64) b = 64;
  return (int)b;
}

extern "C" int snws_tile_minmax_scale(const float* x, void* y, int out_dtype, int n_tiles, int64_t elems_per_tile,
                                      float* part, float* minmax_out, void* stream) {
  SNWS_CHECK_ARG(x && y && part, "snws_tile_minmax_scale: NULL pointer");
  SNWS_CHECK_ARG(n_tiles >= 1 && n_tiles <= 65535 && elems_per_tile >= 1,
                 "snws_tile_minmax_scale: n_tiles %d (1..65535), elems_per_tile %lld", n_tiles, (long long)elems_per_tile);
  SNWS_CHECK_ARG(out_dtype == SNWS_F32 || out_dtype == SNWS_BF16, "snws_tile_minmax_scale: out_dtype %d is not F32/BF16",
                 out_dtype);
  const int B = snws_tile_minmax_blocks(elems_per_tile);
  hipStream_t st = (hipStream_t)stream;
  tile_minmax_kernel<<<dim3(B, n_tiles), DT_THREADS, 0, st>>>(x, elems_per_tile, B, part);
  SNWS_LAUNCH_CHECK("snws_tile_minmax_scale(1)");
  if (out_dtype == SNWS_F32)
    tile_scale_kernel<float><<<dim3(B, n_tiles), DT_THREADS, 0, st>>>(x, (float*)y, elems_per_tile, B, part, minmax_out);
  else
    tile_scale_kernel<bf16_t><<<dim3(B, n_tiles), DT_THREADS, 0, st>>>(x, (bf16_t*)y, elems_per_tile, B, part, minmax_out);
  SNWS_LAUNCH_CHECK("snws_tile_minmax_scale(2)");
  return SNWS_OK;
}

extern "C" int snws_label_remap(const void* src, int src_dtype, float* dst, int64_t n, void* stream) {
  SNWS_CHECK_ARG(src && dst, "snws_label_remap: NULL pointer");
  SNWS_CHECK_ARG(n >= 0, "snws_label_remap: negative size");
  if (n == 0) return SNWS_OK;
  const unsigned grid = (unsigned)cdiv64(n, DT_THREADS);
  hipStream_t st = (hipStream_t)stream;
  switch (src_dtype) {
    case SNWS_F32: label_remap_kernel<float><<<grid, DT_THREADS, 0, st>>>((const float*)src, dst, n); break;
    case SNWS_F64: label_remap_kernel<double><<<grid, DT_THREADS, 0, st>>>((const double*)src, dst, n); break;
    case SNWS_I64: label_remap_kernel<int64_t><<<grid, DT_THREADS, 0, st>>>((const int64_t*)src, dst, n); break;
    case SNWS_I32: label_remap_kernel<int><<<grid, DT_THREADS, 0, st>>>((const int*)src, dst, n); break;
    case SNWS_U8: label_remap_kernel<unsigned char><<<grid, DT_THREADS, 0, st>>>((const unsigned char*)src, dst, n); break;
    default: SNWS_CHECK_ARG(false, "snws_label_remap: unsupported src_dtype %d", src_dtype);
  }
  SNWS_LAUNCH_CHECK("snws_label_remap");
  return SNWS_OK;
}

extern "C" int snws_nchw_to_nhwc_pad(const void* x, int in_dtype, void* y, int out_dtype, int N, int C, int H, int W, int c_pad,
                                     void* stream) {
  SNWS_CHECK_ARG(x && y, "snws_nchw_to_nhwc_pad: NULL pointer");
  SNWS_CHECK_ARG(N >= 1 && N <= 65535 && C >= 1 && H >= 1 && W >= 1, "snws_nchw_to_nhwc_pad: bad sizes");
  SNWS_CHECK_ARG((in_dtype == SNWS_F32 || in_dtype == SNWS_BF16) && (out_dtype == SNWS_F32 || out_dtype == SNWS_BF16),
                 "snws_nchw_to_nhwc_pad: dtypes must be F32 or BF16");
  const int mult = out_dtype == SNWS_BF16 ? 8 : 4;      // whole 16-byte stores per pixel
  SNWS_CHECK_ARG(c_pad >= C && c_pad % mult == 0 && c_pad <= LAY_MAXC,
                 "snws_nchw_to_nhwc_pad: c_pad %d must be a multiple of %d in [C = %d, %d]", c_pad, mult, C, LAY_MAXC);
  SNWS_CHECK_ARG(((uintptr_t)y & 15) == 0, "snws_nchw_to_nhwc_pad: y must be 16-byte aligned");
  const int64_t HW = (int64_t)H * W;
  const dim3 grid((unsigned)cdiv64(HW, LAY_PIX), N);
  const size_t lds = (size_t)c_pad * (LAY_PIX + 1) * sizeof(float);
  const int isz = in_dtype == SNWS_F32 ? 4 : 2;
  const bool vec = HW % 4 == 0 && ((uintptr_t)x % (4 * isz)) == 0;     // every (channel, 4-pixel group) is an aligned vector
  hipStream_t st = (hipStream_t)stream;
#define LAY(TI, TO)                                                                                                   \
  do {                                                                                                                \
    if (vec) nchw_to_nhwc_pad_kernel<TI, TO, true><<<grid, 256, lds, st>>>((const TI*)x, (TO*)y, C, HW, c_pad);        \
    else nchw_to_nhwc_pad_kernel<TI, TO, false><<<grid, 256, lds, st>>>((const TI*)x, (TO*)y, C, HW, c_pad);           \
  } while (0)
  if (in_dtype == SNWS_F32 && out_dtype == SNWS_F32) LAY(float, float);
  else if (in_dtype == SNWS_F32) LAY(float, bf16_t);
  else if (out_dtype == SNWS_F32) LAY(bf16_t, float);
  else LAY(bf16_t, bf16_t);
#undef LAY
  SNWS_LAUNCH_CHECK("snws_nchw_to_nhwc_pad");
  return SNWS_OK;
}
