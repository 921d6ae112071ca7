// Shared helpers for the gfx950 kernels of libsnws (error plumbing, bf16 vector I/O, reductions).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/snws.h"

// ------------------------------------------------------------------ errors
void snws_set_error(const char* fmt, ...);

#define SNWS_CHECK_ARG(cond, ...)                          \
  do {                                                     \
    if (!(cond)) {                                         \
      snws_set_error(__VA_ARGS__);                         \
      return SNWS_ERR_INVALID;                             \
    }                                                      \
  } while (0)

#define SNWS_UNSUPPORTED(...)                              \
  do {                                                     \
    snws_set_error(__VA_ARGS__);                           \
    return SNWS_ERR_UNSUPPORTED;                           \
  } while (0)

// every launch is followed by this: reports launch-configuration errors without synchronising
#define SNWS_LAUNCH_CHECK(name)                                                       \
  do {                                                                                \
    hipError_t e__ = hipGetLastError();                                               \
    if (e__ != hipSuccess) {                                                          \
      snws_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));          \
      return SNWS_ERR_HIP;                                                            \
    }                                                                                 \
  } while (0)

// hipFuncSetAttribute applies to the current device only: one flag per device and kernel, so a process that drives
// several GPUs configures each of them (std::atomic: the first calls may come from different host threads)
#include <atomic>
struct PerDeviceOnce {
  std::atomic<bool> done[64];
  // the flag of the current device, or nullptr when the device cannot be determined / is out of range (then: always configure)
  std::atomic<bool>* flag() {
    int d = -1;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
    return &done[d];
  }
};

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ------------------------------------------------------------------ bf16 <-> f32
typedef __bf16 bf16_t;

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  bf16_t h = (bf16_t)f;  // v_cvt_pk_bf16_f32: round-to-nearest-even, NaN stays NaN
  return __builtin_bit_cast(unsigned short, h);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  return (unsigned)f32_to_bf16_bits(lo) | ((unsigned)f32_to_bf16_bits(hi) << 16);
}

// eight bf16 values + eight bf16 values, summed in f32 and rounded once
__device__ __forceinline__ uint4 add_bf16x8(uint4 a, uint4 b) {
  const unsigned x[4] = {a.x, a.y, a.z, a.w}, y[4] = {b.x, b.y, b.z, b.w};
  unsigned r[4];
#pragma unroll
  for (int e = 0; e < 4; ++e)
    r[e] = pack_bf16x2(__uint_as_float(x[e] << 16) + __uint_as_float(y[e] << 16),
                       __uint_as_float(x[e] & 0xffff0000u) + __uint_as_float(y[e] & 0xffff0000u));
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// Vector access: V consecutive elements of type T as floats.  V*sizeof(T) is 16 bytes on the fast path, V=1 otherwise.
template <typename T, int V>
struct Vec;

template <>
struct Vec<float, 4> {
  static __device__ __forceinline__ void load(const float* p, float (&f)[4]) {
    float4 v = *reinterpret_cast<const float4*>(p);
    f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float (&f)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(f[0], f[1], f[2], f[3]);
  }
};
template <>
struct Vec<float, 1> {
  static __device__ __forceinline__ void load(const float* p, float (&f)[1]) { f[0] = *p; }
  static __device__ __forceinline__ void store(float* p, const float (&f)[1]) { *p = f[0]; }
};
template <>
struct Vec<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
    uint4 v = *reinterpret_cast<const uint4*>(p);
    unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(u[i] << 16);
      f[2 * i + 1] = __uint_as_float(u[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
    uint4 v;
    v.x = pack_bf16x2(f[0], f[1]);
    v.y = pack_bf16x2(f[2], f[3]);
    v.z = pack_bf16x2(f[4], f[5]);
    v.w = pack_bf16x2(f[6], f[7]);
    *reinterpret_cast<uint4*>(p) = v;
  }
};
// streaming variants: non-temporal 16-byte accesses for tensors that are read or written exactly once per kernel
template <typename T, int V>
struct VecNT {
  static __device__ __forceinline__ void load(const T* p, float (&f)[V]) { Vec<T, V>::load(p, f); }
  static __device__ __forceinline__ void store(T* p, const float (&f)[V]) { Vec<T, V>::store(p, f); }
};
typedef unsigned int snws_u32x4 __attribute__((ext_vector_type(4)));
template <>
struct VecNT<bf16_t, 8> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[8]) {
    const snws_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const snws_u32x4*>(p));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f[2 * i] = __uint_as_float(v[i] << 16);
      f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[8]) {
    snws_u32x4 v;
    v[0] = pack_bf16x2(f[0], f[1]);
    v[1] = pack_bf16x2(f[2], f[3]);
    v[2] = pack_bf16x2(f[4], f[5]);
    v[3] = pack_bf16x2(f[6], f[7]);
    __builtin_nontemporal_store(v, reinterpret_cast<snws_u32x4*>(p));
  }
};

template <>
struct Vec<bf16_t, 1> {
  static __device__ __forceinline__ void load(const bf16_t* p, float (&f)[1]) { f[0] = (float)(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, const float (&f)[1]) { *p = (bf16_t)f[0]; }
};

__device__ __forceinline__ float round_through(float v, float*) { return v; }
__device__ __forceinline__ float round_through(float v, bf16_t*) { return (float)((bf16_t)v); }

// ------------------------------------------------------------------ reductions (fixed order => bit-reproducible)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Sum over all threads of a block (blockDim.x multiple of 64, <= 1024).  Result valid in every thread.
template <typename F>
__device__ __forceinline__ F block_sum(F v, F* scratch /* >= 16 entries */) {
  v = wave_sum(v);
  const int wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) scratch[wid] = v;
  __syncthreads();
  F t = 0;
  for (int i = 0; i < nw; ++i) t += scratch[i];
  return t;
}
