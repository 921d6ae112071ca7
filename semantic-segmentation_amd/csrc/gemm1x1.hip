// Wide 1x1 convolution as a bf16 GEMM on the gfx950 matrix cores: y[m][n] = sum_k x[m][k] * w[n][k] (+ bias[n]) over
// all pixels m of an NHWC tensor.  Serves the plain nn.Conv2d 1x1 layers of the OCR head
// (modeling/UNet3p_res_ocr_SNws.py:39-73,118-123,269-274 of the reference), their input gradients (the same product
// with the transposed filter), and the tap form of conv(upsample(x)) (1x1 with taps * C_out output channels).
//
// These products are tall and thin - millions of pixels, a few hundred channels on either side - so the pixel operand
// is streamed from HBM once and the filter lives in L2.  A persistent workgroup (8 waves, one per CU) owns 256-pixel
// x BN-channel output tiles; the contraction walks 64 channels per step.  Both operands are K-contiguous in memory,
// so a stage is 128-byte rows copied global -> LDS by 16-byte LDS-DMA loads (1 KB = 8 rows per wave-instruction) into
// a two-stage ring: the copy of step s + 1 - of the next tile's first step at a tile's end - is in flight while step s
// multiplies, retired by a counted s_waitcnt and a raw s_barrier.  Rows are XOR-swizzled by 16-byte chunk (chunk ^=
// (row >> 1) & 7, applied to the SOURCE address: LDS-DMA writes lane-linearly) so the 16 lanes of a ds_read_b128 group
// hit 16 distinct bank slots.  Matrix orientation D[channel][pixel] (A = filter rows, B = pixels): a lane then holds
// four consecutive channels of one pixel, and the epilogue turns a wave's 16-pixel slab through a wave-private LDS
// scratch into whole-line (>= 96-byte) stores without any workgroup barrier.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// XCD-aware block id: blocks dealt to the same XCD (bid % 8) get consecutive logical ids.  Bijective for any grid.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct GemmArgs {
  const bf16_t* x;     // [M][ldx]
  const bf16_t* w;     // [N][K]
  const float* bias;   // [N] or nullptr
  bf16_t* y;           // [M][ldy]
  int64_t ldx, ldy, M;
  int N, K, ntN;       // ntN: channel tiles per pixel tile
  unsigned ntiles;
  int accumulate;      // != 0: y += product (y holds a bf16 partial result or a gradient with several contributors)
  // STATS instantiations (NR == 4, M and HW multiples of 256): per-tile per-channel sums of the STORED values (after
  // rounding to bf16 and, when accumulating, after the add) for a normalisation layer that follows:
  // stat[((img * Tn + tile) * N + c) * 2 + {0, 1}] = sum, sum of squares of y (no shift: a finalize pass takes shift 0)
  float* stat;
  int64_t HW;
  int Tn;              // 256-pixel tiles per image
};

// WM x WN waves, each MR x NR MFMA tiles of 16 x 16: workgroup tile = (WM*MR*16 = 256 pixels) x (WN*NR*16 channels)
template <int WM, int WN, int MR, int NR>
struct GemmGeom {
  static_assert(WM * WN == 8 && WM * MR * 16 == 256, "8 waves, 256 pixels");
  static constexpr int BM = 256, BN = WN * NR * 16;
  static constexpr int AW = BM / 8 / 8;                       // pixel pieces (8 rows x 128 B) per wave and step
  static constexpr int BW = (BN / 8 + 7) / 8;                 // filter pieces per wave and step (region rounded up)
  static constexpr int A_BYTES = BM * 128, B_BYTES = BW * 8 * 1024, STAGE = A_BYTES + B_BYTES;
  static constexpr int NLOAD = AW + BW;                       // LDS-DMA loads per wave and step
  static constexpr int RS = NR * 32 + 16;                     // scratch bytes per pixel row (padded)
  static constexpr int SCR = 16 * RS;
  static constexpr int RED_BYTES = 8 * 64 * 2 * 4;            // per-wave channel partials of the statistics epilogue
  static constexpr int LDS_BYTES = 2 * STAGE + 8 * SCR + RED_BYTES;
};

template <int WM, int WN, int MR, int NR, bool STATS = false>
__global__ __launch_bounds__(512) void gemm1x1_kernel(GemmArgs a) {
  static_assert(!STATS || NR == 4, "the statistics epilogue needs 8 chunks per row: a lane keeps its channel octet");
  using G = GemmGeom<WM, WN, MR, NR>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nk = a.K >> 6;
  unsigned char* scr = smem + 2 * G::STAGE + wave * G::SCR;

  // LDS-DMA lane geometry: piece p = wave + 8j covers rows 8p .. 8p+7; lane -> row lane >> 3, LDS slot lane & 7, and
  // fetches the global chunk slot ^ ((row >> 1) & 7) = slot ^ ((4 * (p & 1) + (lane >> 4)) & 7); p & 1 == wave & 1
  const int drow = lane >> 3;
  const int dchunk = (lane & 7) ^ ((4 * (wave & 1) + (lane >> 4)) & 7);
  // fragment reads: lane -> row lane & 15 of a 16-row block, k octet lane >> 4 of the 32-wide k-step ks
  int fro[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) fro[ks] = (lane & 15) * 128 + (((ks * 4 + (lane >> 4)) ^ ((lane >> 1) & 7)) * 16);

  const unsigned char* asrc[G::AW];
  const unsigned char* bsrc[G::BW];
  auto set_src = [&](unsigned t) {
    const int64_t m0 = (int64_t)(t / a.ntN) * G::BM;
    const int n0 = (int)(t % a.ntN) * G::BN;
#pragma unroll
    for (int j = 0; j < G::AW; ++j) {
      int64_t row = m0 + 8 * (wave + 8 * j) + drow;
      if (row > a.M - 1) row = a.M - 1;                     // past the end: a valid row whose results are not stored
      asrc[j] = reinterpret_cast<const unsigned char*>(a.x + row * a.ldx + dchunk * 8);
    }
#pragma unroll
    for (int j = 0; j < G::BW; ++j) {
      int row = n0 + 8 * (wave + 8 * j) + drow;
      if (row > a.N - 1) row = a.N - 1;
      bsrc[j] = reinterpret_cast<const unsigned char*>(a.w + (int64_t)row * a.K + dchunk * 8);
    }
  };
  auto stage = [&](int kt, int buf) {
    unsigned char* base = smem + buf * G::STAGE;
#pragma unroll
    for (int j = 0; j < G::AW; ++j) glds16(asrc[j] + kt * 128, base + (wave + 8 * j) * 1024);
#pragma unroll
    for (int j = 0; j < G::BW; ++j) glds16(bsrc[j] + kt * 128, base + G::A_BYTES + (wave + 8 * j) * 1024);
  };

  f32x4 acc[MR][NR];
  auto compute = [&](int buf) {
    const unsigned char* xb = smem + buf * G::STAGE + (wm * MR * 16) * 128;
    const unsigned char* wb = smem + buf * G::STAGE + G::A_BYTES + (wn * NR * 16) * 128;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 wf[NR], xf[MR];
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) wf[nr] = *reinterpret_cast<const bf16x8*>(wb + nr * 2048 + fro[ks]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr) xf[mr] = *reinterpret_cast<const bf16x8*>(xb + mr * 2048 + fro[ks]);
#pragma unroll
      for (int mr = 0; mr < MR; ++mr)
#pragma unroll
        for (int nr = 0; nr < NR; ++nr)
          acc[mr][nr] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nr], xf[mr], acc[mr][nr], 0, 0, 0);
    }
  };

  // D[channel][pixel]: lane holds pixel lane & 15 of the 16-pixel slab mr and channels nr*16 + 4*(lane >> 4) + (0..3)
  auto epilogue = [&](unsigned t) {
    const int64_t m0 = (int64_t)(t / a.ntN) * G::BM + wm * MR * 16;
    const int n0 = (int)(t % a.ntN) * G::BN + wn * NR * 16;
    float b4[NR][4];
#pragma unroll
    for (int nr = 0; nr < NR; ++nr)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int n = n0 + nr * 16 + 4 * (lane >> 4) + j;
        b4[nr][j] = (a.bias && n < a.N) ? a.bias[n] : 0.f;
      }
    constexpr int CPR = NR * 2, TOTAL = 16 * CPR, ITERS = (TOTAL + 63) / 64;   // 16-byte chunks per row / per slab
    // statistics epilogue (CPR == 8: a lane keeps its channel octet lane & 7 through every store iteration)
    constexpr bool stats = STATS;
    float S[8], Q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) S[e] = Q[e] = 0.f;
#pragma unroll
    for (int mr = 0; mr < MR; ++mr) {
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const f32x4 v = acc[mr][nr];
        *reinterpret_cast<uint2*>(scr + (lane & 15) * G::RS + (nr * 16 + 4 * (lane >> 4)) * 2) =
            make_uint2(pack_bf16x2(v[0] + b4[nr][0], v[1] + b4[nr][1]), pack_bf16x2(v[2] + b4[nr][2], v[3] + b4[nr][3]));
      }
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < ITERS; ++it) {
        const int q = it * 64 + lane;
        const int row = q / CPR, c = q - row * CPR;
        const int64_t m = m0 + mr * 16 + row;
        const int n = n0 + c * 8;
        if (q < TOTAL && m < a.M && n < a.N) {
          uint4 v = *reinterpret_cast<const uint4*>(scr + row * G::RS + c * 16);
          if (a.accumulate) v = add_bf16x8(v, *reinterpret_cast<const uint4*>(a.y + m * a.ldy + n));   // kernel-uniform
          *reinterpret_cast<uint4*>(a.y + m * a.ldy + n) = v;
          if (stats) {
            const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float f0 = __uint_as_float(u[e] << 16);
              const float f1 = __uint_as_float(u[e] & 0xffff0000u);
              S[2 * e] += f0; Q[2 * e] = fmaf(f0, f0, Q[2 * e]);
              S[2 * e + 1] += f1; Q[2 * e + 1] = fmaf(f1, f1, Q[2 * e + 1]);
            }
          }
        }
      }
      wave_lds_sync();
    }
    if (stats) {
      // rows of a store iteration sit in lane bits 3..5: fold them, then the WM waves that share these channels (fixed
      // order: bit-reproducible)
      float* red = reinterpret_cast<float*>(smem + 2 * G::STAGE + 8 * G::SCR);
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
          red[(wave * 64 + lane * 8 + e) * 2] = S[e];
          red[(wave * 64 + lane * 8 + e) * 2 + 1] = Q[e];
        }
      }
      // (raw barriers: __syncthreads would also wait for the tile's global stores)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (wm == 0) {
        const int n = n0 + lane;
        float s_ = 0.f, q_ = 0.f;
#pragma unroll
        for (int w_ = 0; w_ < WM; ++w_) {
          s_ += red[((w_ * WN + wn) * 64 + lane) * 2];
          q_ += red[((w_ * WN + wn) * 64 + lane) * 2 + 1];
        }
        const int64_t mt = (int64_t)(t / a.ntN) * G::BM;        // first pixel of the tile: inside one image
        const int64_t img = mt / a.HW, tile = (mt - img * a.HW) / G::BM;
        if (n < a.N) {
          float* o = a.stat + ((img * a.Tn + tile) * a.N + n) * 2;
          o[0] = s_;
          o[1] = q_;
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // red is free again before any wave's next epilogue
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const unsigned first = xcd_remap(blockIdx.x, gridDim.x);
  if (first >= a.ntiles) return;
  set_src(first);
  stage(0, 0);
  int buf = 0;
  bool landed = false;          // true when the stage about to be multiplied was retired before the last epilogue
  for (unsigned t = first; t < a.ntiles; t += gridDim.x) {
#pragma unroll
    for (int mr = 0; mr < MR; ++mr)
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) acc[mr][nr] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned tn = t + gridDim.x;
    for (int kt = 0; kt < nk; ++kt) {
      bool more = true;
      if (kt + 1 < nk) {
        stage(kt + 1, buf ^ 1);
      } else if (tn < a.ntiles) {
        set_src(tn);
        stage(0, buf ^ 1);
      } else {
        more = false;
      }
      if (!landed) {
        if (more) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NLOAD) : "memory");   // this step landed, the next in flight
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      landed = false;
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      compute(buf);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // every wave is done reading this stage before it is refilled
      buf ^= 1;
    }
    // The next tile's first stage has had a whole step to land: retire it now, so that the counted wait of the step
    // after next does not have to drain the stores below as well (loads and stores share the counter).
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    landed = true;
    epilogue(t);
  }
}

template <int WM, int WN, int MR, int NR, bool STATS = false>
int launch_gemm(GemmArgs a, hipStream_t st) {
  using G = GemmGeom<WM, WN, MR, NR>;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm1x1_kernel<WM, WN, MR, NR, STATS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      snws_set_error("snws_gemm1x1: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  a.ntN = cdiv(a.N, G::BN);
  const int64_t tiles = cdiv64(a.M, G::BM) * a.ntN;
  if (tiles > 0x7fffffff) {
    snws_set_error("snws_gemm1x1: too many tiles");
    return SNWS_ERR_INVALID;
  }
  a.ntiles = (unsigned)tiles;
  const unsigned grid = a.ntiles < 256u ? a.ntiles : 256u;    // persistent: one workgroup per CU
  gemm1x1_kernel<WM, WN, MR, NR, STATS><<<grid, 512, G::LDS_BYTES, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_gemm1x1");
  return SNWS_OK;
}

// channel-tile width with the least padded work for N output channels (ties: the wider tile)
int pick_bn(int N) {
  const int cand[4] = {256, 192, 160, 128};
  int best = 256;
  int64_t best_cost = -1;
  for (int i = 0; i < 4; ++i) {
    const int64_t cost = (int64_t)cdiv(N, cand[i]) * cand[i];
    if (best_cost < 0 || cost < best_cost) { best = cand[i]; best_cost = cost; }
  }
  return best;
}

}  // namespace

extern "C" int snws_gemm1x1_supported(int N, int K) { return N >= 64 && N % 8 == 0 && K >= 64 && K % 64 == 0; }

namespace {
int run_gemm1x1(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t M, int N, int K,
                int accumulate, void* stream, float* stat = nullptr, int64_t HW = 0) {
  SNWS_CHECK_ARG(x && w && y, "snws_gemm1x1: NULL pointer");
  if (!snws_gemm1x1_supported(N, K)) SNWS_UNSUPPORTED("snws_gemm1x1: N = %d (>= 64, multiple of 8), K = %d (multiple of 64)", N, K);
  SNWS_CHECK_ARG(M >= 1 && ldx >= K && ldx % 8 == 0 && ldy >= N && ldy % 8 == 0, "snws_gemm1x1: bad sizes");
  SNWS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0,
                 "snws_gemm1x1: pointers must be 16-byte aligned");
  GemmArgs a{};
  a.x = (const bf16_t*)x; a.w = (const bf16_t*)w; a.bias = bias; a.y = (bf16_t*)y;
  a.ldx = ldx; a.ldy = ldy; a.M = M; a.N = N; a.K = K; a.accumulate = accumulate;
  hipStream_t st = (hipStream_t)stream;
  if (stat) {
    // the statistics epilogue exists for the 256- and 128-wide channel tiles: the one with less padded work
    const int bn = cdiv(N, 256) * 256 <= cdiv(N, 128) * 128 ? 256 : 128;
    if (HW < 256 || HW % 256 || M % HW)
      SNWS_UNSUPPORTED("snws_gemm1x1_stats: needs whole 256-pixel tiles per image (HW = %lld, M = %lld)", (long long)HW,
                       (long long)M);
    a.stat = stat; a.HW = HW; a.Tn = (int)(HW / 256);
    return bn == 256 ? launch_gemm<2, 4, 8, 4, true>(a, st) : launch_gemm<4, 2, 4, 4, true>(a, st);
  }
  switch (pick_bn(N)) {
    case 256: return launch_gemm<2, 4, 8, 4>(a, st);
    case 192: return launch_gemm<2, 4, 8, 3>(a, st);
    case 160: return launch_gemm<4, 2, 4, 5>(a, st);
    default: return launch_gemm<4, 2, 4, 4>(a, st);
  }
}
}  // namespace

extern "C" int snws_gemm1x1(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t M,
                            int N, int K, void* stream) {
  return run_gemm1x1(x, ldx, w, bias, y, ldy, M, N, K, 0, stream);
}

extern "C" int snws_gemm1x1_acc(const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int64_t M, int N, int K,
                                void* stream) {
  return run_gemm1x1(x, ldx, w, nullptr, y, ldy, M, N, K, 1, stream);
}

extern "C" int snws_gemm1x1_stat_tiles(int64_t HW) { return (HW >= 256 && HW % 256 == 0) ? (int)(HW / 256) : -1; }

extern "C" int snws_gemm1x1_stats(const void* x, int64_t ldx, const void* w, const float* bias, void* y, int64_t ldy, int64_t M,
                                  int N, int K, int accumulate, int64_t HW, float* stat_part, void* stream) {
  SNWS_CHECK_ARG(stat_part, "snws_gemm1x1_stats: NULL stat_part");
  SNWS_CHECK_ARG(!(accumulate && bias), "snws_gemm1x1_stats: no bias in accumulate mode");
  return run_gemm1x1(x, ldx, w, bias, y, ldy, M, N, K, accumulate, stream, stat_part, HW);
}
