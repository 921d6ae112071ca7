// Implicit-GEMM 3x3 / 1x1 convolution on the gfx950 bf16 matrix cores (replaces F.conv2d at
// modeling/weightstd.py:18 of the reference and its autograd for the stride-1 "same" convolutions of the
// UNet3+ stack).  NHWC bf16 activations, f32 accumulation, v_mfma_f32_16x16x32_bf16 (v_mfma_f32_32x32x16_bf16 forms
// are kept behind switches for A/B runs).
//
// Forward / dgrad (one kernel; dgrad = forward over dy with the flipped, transposed filter image):
//   workgroup = 4 waves = a 16x32 pixel patch of one image x 64 output channels; wave w owns rows 4w..4w+3.
//   GEMM orientation: D[out channel][pixel] = sum_k A[out channel][k] * B[k][pixel], k = (tap, in channel).
//   The contraction walks KC input channels at a time: the (16+2)x(32+2) input halo patch and the filter slab of
//   that chunk are copied global->LDS with 16-byte LDS-DMA loads (out-of-image pixels read a zero page), then
//   every tap reads its shifted B fragments from the SAME halo patch.  Pixels are 16-byte-chunk XOR-swizzled so
//   the 16 lanes of a ds_read_b128 group hit 16 distinct bank slots; the swizzle is applied to the per-lane
//   SOURCE address because LDS-DMA writes lane-linearly.  Two workgroups per CU (76.8 KB LDS each) overlap one
//   block's staging with the other's MFMAs; inside the matrix phase a wave's priority rises with its progress, so
//   the co-resident pair drifts into alternation instead of multiplying - and then waiting - together.
//   Epilogue: accumulators (+bias) -> bf16 -> LDS -> 128-byte-per-pixel coalesced stores (issued last: nothing in
//   the workgroup waits for them), plus optional per-tile sum / sum-of-squares of the ROUNDED outputs per channel,
//   which SwitchNorm consumes instead of re-reading the tensor.  The BWR instantiation (dgrad only) instead reduces,
//   per tile, what pass 1 of the backward of the SwitchNorm(+ReLU) layers behind the gradient needs (sum g,
//   sum g * xhat with g masked by the recomputed ReLU), reading those layers' inputs for the tile it is storing.
//
// Wgrad: D[out channel][in channel] = sum over pixels; both operands are needed pixel-major per lane, so the
//   NHWC tiles are read from LDS with the transposing ds_read_b64_tr_b16.  A workgroup owns a 64x64 (out,in)
//   filter tile for all taps (144 accumulator registers per lane) and sweeps a slice of the pixel tiles; in the 3x3
//   kernel a wave owns all 64 out channels x a 16-wide in-channel quarter x all nine taps and walks the halo rows, so
//   one B fragment serves three taps.  The result leaves either as per-workgroup partial tiles that a fixed-order
//   reduction sums (bit-reproducible; what the host modules use) or by float atomics into the f32 gradient.  3x3
//   and 1x1 have their own kernels (conv_wgrad9p_kernel / conv_wgrad9q_kernel, conv_wgrad1*_kernel); the generic one
//   serves C_in = 16.
#include "common.h"
#include "pack.h"
#include "wgrad_reduce.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(8))) short short8v;

constexpr int ZERO_PAGE_BYTES = 8192;
__device__ __attribute__((aligned(256))) unsigned int snws_zero_page[ZERO_PAGE_BYTES / 4];  // never written: all zero

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

#ifdef SNWS_TIMING
// Diagnostics build (-DSNWS_TIMING): shader-clock cycles the forward kernel's workgroups spend waiting for staged data,
// in the matrix phase and in the epilogue ([0..3]), and the 3x3 wgrad's per pixel tile waiting / multiplying ([4..7]),
// summed over workgroups (wave 0 only); snws_dbg_read copies and clears them.
__device__ unsigned long long snws_dbg[16];   // [8..15]: the ping-pong 3x3 wgrad (see conv_wgrad9p_kernel)
#define SNWS_T(var) const long long var = clock64()
#else
#define SNWS_T(var)
#endif

// Progressive wave priority.  Two workgroups share a CU so that one stages while the other multiplies; with equal
// priorities two waves that multiply at the same time share the matrix pipe evenly, finish together and then stage
// together - the pipe idles for a whole staging latency.  A wave that is further into its matrix phase outranks one
// that has just begun, so the leader finishes at full speed (first come, first served) and the pair falls into
// alternation.  level: 0 outside the matrix phase, 1..3 by progress.  The levels are set at compile-time positions of
// the unrolled matrix phase (template parameter PRIO; a run-time test there costs more than the priority gains).
// XCD-aware block id: blocks that are dealt to the same XCD (bid % 8) get consecutive logical ids, so tiles that
// share an input patch or a filter slab also share an L2.  Bijective for any grid size.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
}

// One 64-output-channel tile of a dgrad launch whose output is also the incoming gradient of a SwitchNorm(+ReLU)
// layer: the epilogue then produces that layer's backward partial sums (what sn_bwd_reduce would compute).
struct SnBwdTile {
  const bf16_t* x;       // the layer's input, already offset to this tile's first channel
  const float* mean;     // per (image, channel) coefficients of the layer, offset likewise; image stride = cseg
  const float* rstd;
  const float* scale;
  const float* beta;     // per channel, offset likewise
  float* part;           // [N][T][cseg][2], offset to this tile's first channel; nullptr = no layer on this tile
  int ldx, cseg, cvalid, relu;   // cvalid = channels of the layer inside this tile
};
constexpr int MAX_BWR_TILES = 16;

struct FwdArgs {
  const bf16_t* x;
  const bf16_t* pack;
  const float* bias;
  bf16_t* y;
  float* stat;
  int N, H, W, Cout;
  int64_t ldx, ldy;
  int tilesX, tilesY, coTiles, nchunks;
  int accumulate;     // != 0: y += result (a gradient with several contributors: a later contributor adds in place)
  int prio;           // host side only: which PRIO instantiation to launch
};
// arguments of the dgrad variant whose epilogue also reduces for the SwitchNorm layers behind its output
struct FwdArgsBwr : FwdArgs {
  SnBwdTile bwr[MAX_BWR_TILES];   // [0 .. coTiles) is valid
};

constexpr int FWD_TH = 16, FWD_TW = 32;
constexpr int OUT_STRIDE = 144;                       // bytes per pixel row of the epilogue tile (128 + 16 pad)
constexpr int OUT_BYTES = FWD_TH * FWD_TW * OUT_STRIDE;  // 73,728
constexpr int STAT_BYTES = 4 * 64 * 2 * 4;            // per-wave channel partials

template <int KC, int TAPS>
struct FwdGeom {
  static constexpr int PADK = TAPS == 9 ? 1 : 0;
  static constexpr int HWd = FWD_TW + 2 * PADK, HHt = FWD_TH + 2 * PADK, NPIX = HWd * HHt;
  static constexpr int PIXB = KC * 2, VPP = KC / 8, PPI = 64 / VPP;
  static constexpr int NI = (NPIX + PPI - 1) / PPI, IN_BYTES = NI * 1024;
  static constexpr int KS = KC / 16, W_BYTES = TAPS * KS * 2048, NIW = W_BYTES / 1024;
  static constexpr int JIN = (NI + 3) / 4, JW = (NIW + 3) / 4;
  static constexpr int MAIN_BYTES = IN_BYTES + W_BYTES;
  static constexpr int LDS_BYTES = (MAIN_BYTES > OUT_BYTES + STAT_BYTES) ? MAIN_BYTES : OUT_BYTES + STAT_BYTES;
  // double-buffered variant: every wave issues the same number of LDS-DMA loads per chunk (so one counted
  // s_waitcnt retires exactly the previous chunk), hence the input region is rounded up to 4*JIN pieces
  // (filter pieces that do not divide by 4 are topped up with repeats of a wave's own previous piece)
  static constexpr int IN_BYTES_DB = JIN * 4 * 1024, STAGE_DB = IN_BYTES_DB + W_BYTES, NLOAD_DB = JIN + JW;
  static constexpr int LDS_BYTES_DB = (2 * STAGE_DB > OUT_BYTES + STAT_BYTES) ? 2 * STAGE_DB : OUT_BYTES + STAT_BYTES;
};

// 16-byte-chunk XOR swizzle of a halo pixel, a function of its COLUMN only: a ds_read_b128 lane group reads 16
// consecutive columns of one row, and the row then enters the address as a compile-time immediate.
// (the 16x16x32 fragment shape puts 16 columns x one chunk in a lane group and needs a different XOR pattern)
template <int VPP, bool M16>
__device__ __forceinline__ int col_swz(int px) {
  return M16 ? ((px >> 2) & 1) * 2 : VPP == 4 ? ((px >> 2) & 3) : ((px >> 3) & 1);
}

// DB = false: one LDS stage, two workgroups per CU hide each other's staging.
// DB = true : two LDS stages in one workgroup per CU; the next chunk's LDS-DMA is issued before the current chunk's
//             MFMAs and retired by a counted s_waitcnt + raw s_barrier, so staging overlaps the matrix work.
// M16      : v_mfma_f32_16x16x32_bf16 instead of 32x32x16 (KC = 32 only).  Same FLOPs per cycle and the same LDS
//             bytes per FLOP, but the chip holds a higher clock under this shape.
// BWR      : the dgrad + SwitchNorm-backward-reduction variant (its own kernel, so a profile lists it separately).
template <int KC, int TAPS, bool DB, bool M16, bool BWR = false, int PRIO = 0>
__global__ __launch_bounds__(256, (DB && TAPS == 9) ? 1 : 2) void conv_fwd_kernel(
    std::conditional_t<BWR, FwdArgsBwr, FwdArgs> a) {
  static_assert(!M16 || KC == 32, "the 16x16x32 shape contracts 32 channels per instruction");
  using G = FwdGeom<KC, TAPS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int IN_B = DB ? G::IN_BYTES_DB : G::IN_BYTES;
  constexpr int STAGE = DB ? G::STAGE_DB : 0;
  unsigned char* in_lds = smem;
  unsigned char* w_lds = smem + IN_B;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned L = xcd_remap(blockIdx.x, gridDim.x);
  const int ct = L % a.coTiles;
  unsigned pt_ = L / a.coTiles;
  const int tx = pt_ % a.tilesX; pt_ /= a.tilesX;
  const int ty = pt_ % a.tilesY;
  const int n = pt_ / a.tilesY;
  const int y0 = ty * FWD_TH, x0 = tx * FWD_TW;

  // per-lane source of every input LDS-DMA this wave issues, as a byte offset from the image base
  // (constant across chunks except the channel offset); ~0u marks a pixel outside the image
  const unsigned char* img = reinterpret_cast<const unsigned char*>(a.x + (int64_t)n * a.H * a.W * a.ldx);
  unsigned src[G::JIN];
#pragma unroll
  for (int j = 0; j < G::JIN; ++j) {
    const int i = wave + 4 * j;
    const int pix = i * G::PPI + lane / G::VPP;
    const int py = pix / G::HWd, px = pix - py * G::HWd;
    const int gy = y0 - G::PADK + py, gx = x0 - G::PADK + px;
    const bool ok = (pix < G::NPIX) && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
    const int chunk = (lane % G::VPP) ^ col_swz<G::VPP, M16>(px);
    src[j] = ok ? (unsigned)((((int64_t)gy * a.W + gx) * a.ldx + chunk * 8) * 2) : ~0u;
  }

  // accumulators: 32x32x16: acc[c2][p] = out channels c2*32.. x tile row p (32 pixels), 16 floats;
  //               16x16x32: acc16[c4][2p+hf] = out channels c4*16.. x columns hf*16.. of tile row p, 4 floats
  f32x16 acc[M16 ? 1 : 2][M16 ? 1 : 4];
  f32x4 acc16[M16 ? 4 : 1][M16 ? 8 : 1];
  if constexpr (M16) {
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
      for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[c4][p][r] = 0.f;
  } else {
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c2][p][r] = 0.f;
  }

  const int r = M16 ? (lane & 15) : (lane & 31), h = M16 ? (lane >> 4) : (lane >> 5);
  const unsigned char* wbase = reinterpret_cast<const unsigned char*>(a.pack) + (int64_t)ct * a.nchunks * G::W_BYTES;
  // LDS read bases: one per (kw, k-step) for the pixel operand (tile rows and kh are immediates), one for the filter.
  // 16x16x32: lane = (column r of 16, 8-channel quarter h of the 32-channel chunk); one k-step per chunk.
  constexpr int KWN = TAPS == 9 ? 3 : 1;
  constexpr int KSB = M16 ? 1 : G::KS;
  const unsigned char* bbase[KWN][KSB];
#pragma unroll
  for (int kw = 0; kw < KWN; ++kw)
#pragma unroll
    for (int ks = 0; ks < KSB; ++ks)
      bbase[kw][ks] = in_lds + (wave * 4 * G::HWd + r + kw) * G::PIXB +
                      (((M16 ? h : ks * 2 + h) ^ col_swz<G::VPP, M16>(r + kw)) * 16);
  const unsigned char* abase = M16 ? w_lds + (h >> 1) * 2048 + (h & 1) * 1024 + r * 16 : w_lds + h * 1024 + r * 16;

  // ---- stage chunk ch into LDS stage `so` (byte offset): halo patch + filter slab
  auto stage = [&](int ch, int so) {
#pragma unroll
    for (int j = 0; j < G::JIN; ++j) {
      const int i = wave + 4 * j;
      if (DB || i < G::NI) {
        const unsigned char* p = src[j] == ~0u ? reinterpret_cast<const unsigned char*>(snws_zero_page)
                                               : img + src[j] + ch * (KC * 2);
        glds16(p, in_lds + so + i * 1024);
      }
    }
    const unsigned char* wslab = wbase + (int64_t)ch * G::W_BYTES + lane * 16;
#pragma unroll
    for (int j = 0; j < G::JW; ++j) {
      int i = wave + 4 * j;
      if (DB && i >= G::NIW) i -= 4;   // keep the per-wave load count uniform: repeat this wave's previous piece
      if (i < G::NIW) glds16(wslab + i * 1024, w_lds + so + i * 1024);
    }
  };
  // ---- MFMAs of one chunk: every tap reads its shifted fragments from the same halo patch.
  // Work is cut into groups g = (kw, k-step, kh) of 8 MFMAs.  The 4 + (KHN-1) pixel-row fragments of a (kw, k-step)
  // set serve all its kh groups; a software pipeline reads the next group's two filter fragments and a share of
  // the next set's pixel fragments while the current group's MFMAs run (two register sets each).
  auto compute32 = [&](int so) {
    constexpr int KHN = TAPS == 9 ? 3 : 1, NB = 4 + KHN - 1, NS = KWN * G::KS, NG = NS * KHN, BPG = (NB + KHN - 1) / KHN;
    bf16x8 A[2][2], B[2][NB];
    auto load_a = [&](int g) {
      const int s_ = g / KHN, kh = g % KHN, kw = s_ / G::KS, ks = s_ % G::KS;
      const int tap = kh * 3 + kw;
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
        A[g & 1][c2] = *reinterpret_cast<const bf16x8*>(abase + so + (tap * G::KS + ks) * 2048 + c2 * 512);
    };
    auto load_b = [&](int s_, int part) {
      const int kw = s_ / G::KS, ks = s_ % G::KS;
#pragma unroll
      for (int q = part * BPG; q < (part + 1) * BPG && q < NB; ++q)
        B[s_ & 1][q] = *reinterpret_cast<const bf16x8*>(bbase[kw][ks] + so + q * (G::HWd * G::PIXB));
    };
#pragma unroll
    for (int part = 0; part < KHN; ++part) load_b(0, part);
    load_a(0);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int s_ = g / KHN, kh = g % KHN;
      if (g + 1 < NG) load_a(g + 1);
      if (s_ + 1 < NS) load_b(s_ + 1, kh);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2)
#pragma unroll
        for (int p = 0; p < 4; ++p)
          acc[c2][p] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[g & 1][c2], B[s_ & 1][p + kh], acc[c2][p], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // 16x16x32: steps (kw, input row rho of the wave's 4 + KHN - 1 halo rows).  A row's two 16-column fragments feed
  // every (kh, tile row p) with p + kh = rho; the filter fragments of the three kh of a kw stay in registers and
  // are replaced one kh at a time as soon as their last use is behind (single register set).
  auto compute16 = [&](int so) {
    constexpr int KHN = TAPS == 9 ? 3 : 1, NR = 4 + KHN - 1, NSTEP = KWN * NR;
    bf16x8 A[KHN][4], B[2][2];
    auto load_a = [&](int kw, int kh) {
      const int tap = kh * 3 + kw;
#pragma unroll
      for (int c4 = 0; c4 < 4; ++c4)
        A[kh][c4] = *reinterpret_cast<const bf16x8*>(abase + so + tap * 4096 + c4 * 256);
    };
    auto load_b = [&](int st) {
      const int kw = st / NR, rho = st % NR;
#pragma unroll
      for (int hf = 0; hf < 2; ++hf)
        B[st & 1][hf] = *reinterpret_cast<const bf16x8*>(bbase[kw][0] + so + rho * (G::HWd * G::PIXB) + hf * (16 * G::PIXB));
    };
    load_a(0, 0);
    load_b(0);
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
      const int kw = st / NR, rho = st % NR;
      // (compile-time: a branch here would cut the unrolled walk into blocks)
      if constexpr (PRIO == 1) {
        if (st == 0) __builtin_amdgcn_s_setprio(1);
        if (st == NSTEP / 3) __builtin_amdgcn_s_setprio(2);
        if (st == 2 * NSTEP / 3) __builtin_amdgcn_s_setprio(3);
      }
      if (st + 1 < NSTEP) load_b(st + 1);
      if (rho + 1 < KHN) load_a(kw, rho + 1);                 // first used by the next row
      if (rho == NR - 1 && kw + 1 < KWN) load_a(kw + 1, 0);   // kh = 0 of this kw was last used at row 3
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kh = 0; kh < KHN; ++kh) {
        const int p = rho - kh;
        if (p < 0 || p > 3) continue;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
          for (int c4 = 0; c4 < 4; ++c4)
            acc16[c4][2 * p + hf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(A[kh][c4], B[st & 1][hf], acc16[c4][2 * p + hf], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (PRIO != 0) __builtin_amdgcn_s_setprio(0);
  };
  auto compute = [&](int so) {
    if constexpr (M16) compute16(so); else compute32(so);
  };

#ifdef SNWS_TIMING
  long long dbg_wait = 0, dbg_comp = 0;
  const long long dbg_t0 = clock64();
#endif
  if (!DB) {
    for (int ch = 0; ch < a.nchunks; ++ch) {
      SNWS_T(ta);
      stage(ch, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      SNWS_T(tb);
      compute(0);
      __syncthreads();
#ifdef SNWS_TIMING
      const long long tc = clock64();
      dbg_wait += tb - ta;
      dbg_comp += tc - tb;
#endif
    }
  } else {
    stage(0, 0);
    for (int ch = 0; ch < a.nchunks; ++ch) {
      const int so = (ch & 1) * STAGE;
      if (ch + 1 < a.nchunks) {
        stage(ch + 1, STAGE - so);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::NLOAD_DB) : "memory");   // chunk ch has landed, ch+1 in flight
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      compute(so);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // every wave is done reading this stage before it is refilled
    }
  }

  // ---- epilogue: accumulators (+bias) -> bf16 tile in LDS.  D[row = out channel][col = pixel]:
  //      lane holds pixel r of its 32-pixel row and out channels (reg&3) + 8*(reg>>2) + 4*h.
  const int co0 = ct * 64;
  unsigned char* out_lds = smem;
  // store phase mapping: 8 lanes x 16 B = one pixel's 64 channels; a pass covers one 32-pixel tile row, so a thread
  // keeps its column (tid >> 3) and channel group (tid & 7) and walks down the 16 rows with a constant stride
  const int j8 = tid & 7, q0 = tid >> 3;
  const int gxs = x0 + q0;
  const int rows = a.H - y0 < FWD_TH ? a.H - y0 : FWD_TH;      // block-uniform
  // A dgrad launch that also reduces for a SwitchNorm layer needs that layer's input at the pixels this thread will
  // store: all 16 loads are issued here, so their latency overlaps the accumulator write-out and the barrier.
  const SnBwdTile* gt = nullptr;
  if constexpr (BWR) gt = &a.bwr[ct];
  const bool bw = BWR && gt->part != nullptr;     // block-uniform
  const bool bwc = bw && j8 * 8 < gt->cvalid && gxs < a.W;
  uint4 xin[16];
  if (bw) {
    const SnBwdTile& g = *gt;
    if (bwc) {
      const bf16_t* bx = g.x + (((int64_t)n * a.H + y0) * a.W + gxs) * g.ldx + j8 * 8;
      const int64_t bstep = (int64_t)a.W * g.ldx;
      if (rows == FWD_TH) {
#pragma unroll
        for (int i = 0; i < 16; ++i) xin[i] = *reinterpret_cast<const uint4*>(bx + i * bstep);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < rows) xin[i] = *reinterpret_cast<const uint4*>(bx + i * bstep);
      }
    }
  }
  // (two inlined copies: a reducing dgrad launch has no bias, and a bias load here would make the compiler wait for
  //  every load in flight, the xin prefetch included)
  auto write_acc = [&](auto with_bias) {
  constexpr bool BIAS = decltype(with_bias)::value;
  const float* bias = a.bias;
  if constexpr (M16) {   // lane holds column r of a 16-pixel half row and out channels c4*16 + 4*h + (0..3)
#pragma unroll
    for (int c4 = 0; c4 < 4; ++c4) {
      const int cl = c4 * 16 + 4 * h;
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = (co0 + cl + e < a.Cout) ? bias[co0 + cl + e] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        const int q = (wave * 4 + (p >> 1)) * 32 + (p & 1) * 16 + r;
        uint2 v;
        v.x = pack_bf16x2(acc16[c4][p][0] + b4[0], acc16[c4][p][1] + b4[1]);
        v.y = pack_bf16x2(acc16[c4][p][2] + b4[2], acc16[c4][p][3] + b4[3]);
        *reinterpret_cast<uint2*>(out_lds + q * OUT_STRIDE + cl * 2) = v;
      }
    }
  } else {
#pragma unroll
  for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int cl = c2 * 32 + 8 * g + 4 * h;  // first of 4 consecutive channels
      float b4[4] = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BIAS) {
#pragma unroll
        for (int e = 0; e < 4; ++e) b4[e] = (co0 + cl + e < a.Cout) ? bias[co0 + cl + e] : 0.f;
      }
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int q = (wave * 4 + p) * 32 + r;
        uint2 v;
        v.x = pack_bf16x2(acc[c2][p][4 * g + 0] + b4[0], acc[c2][p][4 * g + 1] + b4[1]);
        v.y = pack_bf16x2(acc[c2][p][4 * g + 2] + b4[2], acc[c2][p][4 * g + 3] + b4[3]);
        *reinterpret_cast<uint2*>(out_lds + q * OUT_STRIDE + cl * 2) = v;
      }
    }
  }
  }
  };
  if (bw || !a.bias) write_acc(std::false_type{}); else write_acc(std::true_type{});
  __syncthreads();

  // ---- coalesced stores (8 lanes x 16 B = one pixel's 64 channels) + per-channel partial sums: either the moments
  //      of the output (forward: SwitchNorm statistics) or, for a dgrad launch, sum(g) and sum(g * xhat) of the
  //      SwitchNorm layer this gradient belongs to (g = dy masked by the layer's ReLU, recomputed from its input)
  const int cbase = co0 + j8 * 8;
  const bool cok = cbase < a.Cout;
  float sh[8], S[8], Q[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sh[e] = (a.stat && a.bias && cok) ? a.bias[cbase + e] : 0.f;
    S[e] = Q[e] = 0.f;
  }
  const bool lane_ok = cok && gxs < a.W;
  const unsigned char* lp = out_lds + q0 * OUT_STRIDE + j8 * 16;
  bf16_t* yp = a.y + (((int64_t)n * a.H + y0) * a.W + gxs) * a.ldy + cbase;
  const int64_t ystep = (int64_t)a.W * a.ldy;
  // (every loop below exists twice: straight-line for a full 16-row tile, row-checked for the bottom image border)
  if (bw) {   // block-uniform
    // (the output tile itself is stored at the very end: loads and stores share one in-flight counter and may
    // complete out of order with respect to each other, so a wait for xin[i] issued after a store would also drain
    // the stores - and so would the barrier of the cross-wave reduction below)
    const SnBwdTile& g = *gt;
    float mu[8], sc[8], be[8], rs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      mu[e] = rs[e] = sc[e] = 0.f;
      be[e] = g.relu ? 0.f : 1.f;
    }
    if (bwc) {
      // the layer's per-(image, channel) coefficients for this thread's 8 channels (32-byte aligned groups)
      const int64_t ci = (int64_t)n * g.cseg + j8 * 8;
      const float4 m0 = *reinterpret_cast<const float4*>(g.mean + ci), m1 = *reinterpret_cast<const float4*>(g.mean + ci + 4);
      const float4 r0 = *reinterpret_cast<const float4*>(g.rstd + ci), r1 = *reinterpret_cast<const float4*>(g.rstd + ci + 4);
      mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w; mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
      rs[0] = r0.x; rs[1] = r0.y; rs[2] = r0.z; rs[3] = r0.w; rs[4] = r1.x; rs[5] = r1.y; rs[6] = r1.z; rs[7] = r1.w;
      if (g.relu) {   // no ReLU after the layer: scale 0, bias 1 make the mask expression true for every finite input
        const float4 s0 = *reinterpret_cast<const float4*>(g.scale + ci), s1 = *reinterpret_cast<const float4*>(g.scale + ci + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(g.beta + j8 * 8), b1 = *reinterpret_cast<const float4*>(g.beta + j8 * 8 + 4);
        sc[0] = s0.x; sc[1] = s0.y; sc[2] = s0.z; sc[3] = s0.w; sc[4] = s1.x; sc[5] = s1.y; sc[6] = s1.z; sc[7] = s1.w;
        be[0] = b0.x; be[1] = b0.y; be[2] = b0.z; be[3] = b0.w; be[4] = b1.x; be[5] = b1.y; be[6] = b1.z; be[7] = b1.w;
      }
    }
    auto reduce_row = [&](int i) {
      const uint4 v = *reinterpret_cast<const uint4*>(lp + i * 32 * OUT_STRIDE);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
      const unsigned w[4] = {xin[i].x, xin[i].y, xin[i].z, xin[i].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float g0 = __uint_as_float(u[e] << 16), g1 = __uint_as_float(u[e] & 0xffff0000u);
        const float e0 = __uint_as_float(w[e] << 16) - mu[2 * e];
        const float e1 = __uint_as_float(w[e] & 0xffff0000u) - mu[2 * e + 1];
        g0 = fmaf(e0, sc[2 * e], be[2 * e]) > 0.f ? g0 : 0.f;
        g1 = fmaf(e1, sc[2 * e + 1], be[2 * e + 1]) > 0.f ? g1 : 0.f;
        S[2 * e] += g0; Q[2 * e] = fmaf(g0, e0, Q[2 * e]);
        S[2 * e + 1] += g1; Q[2 * e + 1] = fmaf(g1, e1, Q[2 * e + 1]);
      }
    };
    if (lane_ok && bwc) {
      if (rows == FWD_TH) {
#pragma unroll
        for (int i = 0; i < 16; ++i) reduce_row(i);
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (i < rows) reduce_row(i);
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) Q[e] *= rs[e];     // sum(g * (x - mean)) * rstd = sum(g * xhat)
  } else if (a.stat && lane_ok) {
    auto stat_row = [&](int i) {
      const uint4 v = *reinterpret_cast<const uint4*>(lp + i * 32 * OUT_STRIDE);
      const unsigned u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float f0 = __uint_as_float(u[e] << 16) - sh[2 * e];
        const float f1 = __uint_as_float(u[e] & 0xffff0000u) - sh[2 * e + 1];
        S[2 * e] += f0; Q[2 * e] += f0 * f0;
        S[2 * e + 1] += f1; Q[2 * e + 1] += f1 * f1;
      }
    };
    if (rows == FWD_TH) {
#pragma unroll
      for (int i = 0; i < 16; ++i) stat_row(i);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < rows) stat_row(i);
    }
  }
  if (a.stat || bw) {
    float* st_lds = reinterpret_cast<float*>(smem + OUT_BYTES);
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
    const int64_t tile = (int64_t)ty * a.tilesX + tx;
    const int64_t T = (int64_t)a.tilesX * a.tilesY;
    float* o = nullptr;
    if (bw) {
      if (tid < gt->cvalid) o = gt->part + (((int64_t)n * T + tile) * gt->cseg + tid) * 2;
    } else if (tid < 64 && co0 + tid < a.Cout) {
      o = a.stat + (((int64_t)n * T + tile) * a.Cout + co0 + tid) * 2;
    }
    if (o) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        s += st_lds[(w * 64 + tid) * 2];
        q += st_lds[(w * 64 + tid) * 2 + 1];
      }
      o[0] = s;
      o[1] = q;
    }
  }
#ifdef SNWS_TIMING
  if (tid == 0) {
    const long long te = clock64();
    atomicAdd(&snws_dbg[0], (unsigned long long)dbg_wait);
    atomicAdd(&snws_dbg[1], (unsigned long long)dbg_comp);
    atomicAdd(&snws_dbg[2], (unsigned long long)(te - dbg_t0 - dbg_wait - dbg_comp));
    atomicAdd(&snws_dbg[3], 1ull);
  }
#endif
  // ---- the output tile, last: nothing in this workgroup waits for these stores
  if (lane_ok) {
    if (a.accumulate) {     // kernel-uniform: add to what the destination holds (f32 sum of the two bf16 values, rounded once)
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < rows) {
          const uint4 nv = *reinterpret_cast<const uint4*>(lp + i * 32 * OUT_STRIDE);
          const uint4 ov = *reinterpret_cast<const uint4*>(yp + i * ystep);
          const unsigned n4[4] = {nv.x, nv.y, nv.z, nv.w}, o4[4] = {ov.x, ov.y, ov.z, ov.w};
          unsigned r4[4];
#pragma unroll
          for (int e = 0; e < 4; ++e)
            r4[e] = pack_bf16x2(__uint_as_float(n4[e] << 16) + __uint_as_float(o4[e] << 16),
                                __uint_as_float(n4[e] & 0xffff0000u) + __uint_as_float(o4[e] & 0xffff0000u));
          *reinterpret_cast<uint4*>(yp + i * ystep) = make_uint4(r4[0], r4[1], r4[2], r4[3]);
        }
    } else if (rows == FWD_TH) {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        *reinterpret_cast<uint4*>(yp + i * ystep) = *reinterpret_cast<const uint4*>(lp + i * 32 * OUT_STRIDE);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i)
        if (i < rows) *reinterpret_cast<uint4*>(yp + i * ystep) = *reinterpret_cast<const uint4*>(lp + i * 32 * OUT_STRIDE);
    }
  }
}

// ------------------------------------------------------------------------------------------------- wgrad
// Flush modes of the wgrad kernels.  part == nullptr: float atomics into dw / dbias (the caller zeroed them; the
// order of the adds, hence the last bits of the result, varies from run to run).  part != nullptr: every workgroup
// stores its partial tile with plain coalesced stores to part[(split * npairs + pair) * E ...] (E = elements of a
// channel tile, in the kernel's own register order) and its bias partial to pbias[(split * coTiles + cot) * TC ...];
// wgrad_reduce_kernel then sums the splits in a fixed order - bit-reproducible, and cheaper than the atomics.
struct WgradArgs {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  float* dbias;
  float* part;
  float* pbias;
  int N, H, W, Cin, Cout;   // Cin: channels of x that take part (multiple of 16); dw rows have Cin entries
  int64_t ldx, lddy;
  int tilesX, tilesY, coTiles, ciTiles, tilesPerSplit;
  int prio;                 // host side only: which PRIO instantiation to launch
};

constexpr int WG_TH = 8, WG_TW = 32;

template <int CIC, int TAPS>
struct WgGeom {
  static constexpr int PADK = TAPS == 9 ? 1 : 0;
  static constexpr int HWd = WG_TW + 2 * PADK, HHt = WG_TH + 2 * PADK, NPIX = HWd * HHt;
  static constexpr int XPIXB = CIC * 2, XVPP = CIC / 8, XPPI = 64 / XVPP;
  static constexpr int XNI = (NPIX + XPPI - 1) / XPPI, X_BYTES = XNI * 1024 + 64;  // +64: harmless over-read of the last pixel (CIC=16)
  static constexpr int DY_NI = WG_TH * WG_TW / 8, DY_BYTES = DY_NI * 1024;         // [256 px][64 ch] bf16
  static constexpr int JX = (XNI + 3) / 4, JD = DY_NI / 4;
  static constexpr int LDS_BYTES = DY_BYTES + ((X_BYTES + 15) / 16) * 16 + 4 * 64 * 4;
};

__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* p_lo, const unsigned char* p_hi) {
  const short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p_lo);
  const short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p_hi);
  short8v v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
  v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

template <int CIC, int TAPS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs a) {
  using G = WgGeom<CIC, TAPS>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* dy_lds = smem;
  unsigned char* x_lds = smem + G::DY_BYTES;
  float* b_lds = reinterpret_cast<float*>(smem + G::DY_BYTES + ((G::X_BYTES + 15) / 16) * 16);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = a.coTiles * a.ciTiles;
  // XCD-aware order: the (out, in) tile pairs of one pixel slice run on the same XCD at about the same time, so the
  // slice's x / dy tiles are fetched into that L2 once instead of once per pair
  const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % npairs, split = bid / npairs;
  const int cot = pair % a.coTiles, cit = pair / a.coTiles;
  const int ntiles = a.N * a.tilesY * a.tilesX;
  const int t_begin = split * a.tilesPerSplit;
  const int t_end = t_begin + a.tilesPerSplit < ntiles ? t_begin + a.tilesPerSplit : ntiles;

  // wave roles: out-channel half (32 of the tile's 64) x {in-channel half (CIC=64) | pixel-row half (CIC=16)}
  const int co_half = wave & 1, other = wave >> 1;
  const int ci_half = CIC == 64 ? other : 0;
  constexpr int NKS = CIC == 64 ? 16 : 8;
  const int ks_begin = CIC == 64 ? 0 : other * 8;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  // transposing-read lane geometry: 16-lane group g: rows (channels) rb = 16*(g&1).., k half kb = 8*(g>>1);
  // lane 4q+p of the group addresses pixel (kb + q), channels 4p..4p+3 of the group's 16.
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int rb = (grp & 1) * 16, kb = (grp >> 1) * 8;

  for (int t = t_begin; t < t_end; ++t) {
    int tt = t;
    const int tx = tt % a.tilesX; tt /= a.tilesX;
    const int ty = tt % a.tilesY;
    const int n = tt / a.tilesY;
    const int y0 = ty * WG_TH, x0 = tx * WG_TW;
    // ---- stage dy tile [256 px][64 ch] (8 lanes per pixel) and the x halo tile
#pragma unroll
    for (int j = 0; j < G::JD; ++j) {
      const int i = wave + 4 * j;
      const int pix = i * 8 + (lane >> 3);
      const int gy = y0 + (pix >> 5), gx = x0 + (pix & 31);
      const int chunk = (lane & 7) ^ (((pix >> 1) & 1) << 2);
      const bool ok = gy < a.H && gx < a.W && cot * 64 + chunk * 8 < a.Cout;
      const void* s = ok ? (const void*)(a.dy + (((int64_t)n * a.H + gy) * a.W + gx) * a.lddy + cot * 64 + chunk * 8)
                         : (const void*)snws_zero_page;
      glds16(s, dy_lds + i * 1024);
    }
#pragma unroll
    for (int j = 0; j < G::JX; ++j) {
      const int i = wave + 4 * j;
      if (i < G::XNI) {
        const int pix = i * G::XPPI + lane / G::XVPP;
        const int py = pix / G::HWd, px = pix - py * G::HWd;
        const int gy = y0 - G::PADK + py, gx = x0 - G::PADK + px;
        const int cl = lane % G::XVPP;
        const int chunk = CIC == 64 ? (cl ^ (((pix >> 1) & 1) << 2)) : cl;
        const bool ok = pix < G::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && cit * CIC + chunk * 8 < a.Cin;
        const void* s = ok ? (const void*)(a.x + (((int64_t)n * a.H + gy) * a.W + gx) * a.ldx + cit * CIC + chunk * 8)
                           : (const void*)snws_zero_page;
        glds16(s, x_lds + i * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (a.dbias && cit == 0) {  // bias gradient: column sums of the dy tile (block-uniform branch)
      const int co = tid & 63, part = tid >> 6;
      const int c8 = co >> 3, e = co & 7;
      float s = 0.f;
      for (int i = 0; i < 64; ++i) {
        const int pix = part * 64 + i;
        const int chunk = c8 ^ (((pix >> 1) & 1) << 2);
        s += bf16_bits_to_f32(*reinterpret_cast<const unsigned short*>(dy_lds + pix * 128 + chunk * 16 + e * 2));
      }
      bsum += s;
    }

#pragma unroll 2
    for (int kk = 0; kk < NKS; ++kk) {
      const int ks = ks_begin + kk;
      const int ty_ = ks >> 1, cb = (ks & 1) * 16 + kb + gq;  // tile row, first of this lane's two pixel columns
      // A: dy, rows = out channels co_half*32 + rb + 4*gp.., k = pixels
      const int ach = co_half * 32 + rb + 4 * gp;             // channel within the 64-wide tile
      const int p_lo = ty_ * 32 + cb, p_hi = p_lo + 4;
      const unsigned char* a_lo = dy_lds + p_lo * 128 + (((ach >> 3) ^ (((p_lo >> 1) & 1) << 2)) * 16) + (ach & 7) * 2;
      const unsigned char* a_hi = dy_lds + p_hi * 128 + (((ach >> 3) ^ (((p_hi >> 1) & 1) << 2)) * 16) + (ach & 7) * 2;
      const bf16x8 af = tr_frag(a_lo, a_hi);
      const int bch = ci_half * 32 + rb + 4 * gp;
#pragma unroll
      for (int kh = 0; kh < (TAPS == 9 ? 3 : 1); ++kh) {
#pragma unroll
        for (int kw = 0; kw < (TAPS == 9 ? 3 : 1); ++kw) {
          const int h_lo = (ty_ + kh) * G::HWd + cb + kw, h_hi = h_lo + 4;
          const unsigned char *b_lo, *b_hi;
          if (CIC == 64) {
            b_lo = x_lds + h_lo * 128 + (((bch >> 3) ^ (((h_lo >> 1) & 1) << 2)) * 16) + (bch & 7) * 2;
            b_hi = x_lds + h_hi * 128 + (((bch >> 3) ^ (((h_hi >> 1) & 1) << 2)) * 16) + (bch & 7) * 2;
          } else {
            b_lo = x_lds + h_lo * G::XPIXB + bch * 2;
            b_hi = x_lds + h_hi * G::XPIXB + bch * 2;
          }
          const bf16x8 bfr = tr_frag(b_lo, b_hi);
          acc[kh * 3 + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[kh * 3 + kw], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- accumulate into the f32 gradient: D[row = out channel][col = in channel]
  const int ci = cit * CIC + ci_half * 32 + (lane & 31);
  const bool ci_ok = (CIC == 64 || (lane & 31) < 16) && ci < a.Cin;
  if (a.part) {   // partial tile [tap][r][wave][lane] (the CIC = 16 variant's two pixel-row halves are summed by the reducer)
    float* o = a.part + ((int64_t)split * npairs + pair) * (TAPS * 16 * 256) + tid;
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) o[(t * 16 + r) * 256] = acc[t][r];
  } else {
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * 64 + co_half * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (ci_ok && co < a.Cout) atomicAdd(a.dw + ((int64_t)co * TAPS + t) * a.Cin + ci, acc[t][r]);
      }
    }
  }
  if (a.dbias && cit == 0) {
    b_lds[tid] = bsum;
    __syncthreads();
    if (tid < 64) {
      const float v = b_lds[tid] + b_lds[tid + 64] + b_lds[tid + 128] + b_lds[tid + 192];
      if (a.part) a.pbias[((int64_t)split * a.coTiles + cot) * 64 + tid] = v;
      else if (cot * 64 + tid < a.Cout) atomicAdd(a.dbias + cot * 64 + tid, v);
    }
  }
}

// 1x1 wgrad over 64-wide channel tiles.  In the generic kernel a wave owns a 32x32 corner of the (out, in) tile and
// reads two LDS fragments per MFMA, which asks the LDS for more than it delivers.  Here the four waves split the
// PIXELS of a tile (four of its sixteen 16-pixel k-steps each) and every wave accumulates the whole 64x64 tile: two
// dy and two x fragments feed four MFMAs.  The waves' partial tiles are summed through LDS before the atomic flush.
template <bool BIAS>
__global__ __launch_bounds__(256, 2) void conv_wgrad1_kernel(WgradArgs a) {
  using G = WgGeom<64, 1>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* dy_lds = smem;
  unsigned char* x_lds = smem + G::DY_BYTES;
  float* b_lds = reinterpret_cast<float*>(smem + G::DY_BYTES + ((G::X_BYTES + 15) / 16) * 16);   // [4 waves][64 channels]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = a.coTiles * a.ciTiles;
  const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % npairs, split = bid / npairs;
  const int cot = pair % a.coTiles, cit = pair / a.coTiles;
  const int ntiles = a.N * a.tilesY * a.tilesX;
  const int t_begin = split * a.tilesPerSplit;
  const int t_end = t_begin + a.tilesPerSplit < ntiles ? t_begin + a.tilesPerSplit : ntiles;
  const bool want_bias = BIAS && cit == 0;

  f32x16 acc[2][2];   // [out-channel half][in-channel half]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;
  if (want_bias) b_lds[tid] = 0.f;   // ordered before its first use by the barrier that follows the first staging

  // transposing-read lane geometry (see the generic kernel).  Pixel p of k-step ks: p = 16*ks + kb + gq (+4 for the
  // second half); its chunk swizzle bit (p >> 1) & 1 only depends on kb + gq.
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int rb = (grp & 1) * 16, kb = (grp >> 1) * 8;
  const int cb0 = kb + gq;
  const unsigned char *abase[2], *bbase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ch = h * 32 + rb + 4 * gp;
    const int off = (wave * 64 + cb0) * 128 + (((ch >> 3) ^ (((cb0 >> 1) & 1) << 2)) * 16) + (ch & 7) * 2;
    abase[h] = dy_lds + off;
    bbase[h] = x_lds + off;
  }
  // staging: lane = (pixel q of a piece's 8, 16-byte chunk c); piece i holds pixels 8i .. 8i+7
  const int st_q = lane >> 3, st_chunk = (lane & 7) ^ (((st_q >> 1) & 1) << 2);
  const bool st_dy_ok = cot * 64 + st_chunk * 8 < a.Cout, st_x_ok = cit * 64 + st_chunk * 8 < a.Cin;

  for (int t = t_begin; t < t_end; ++t) {
    int tt = t;
    const int tx = tt % a.tilesX; tt /= a.tilesX;
    const int ty = tt % a.tilesY;
    const int n = tt / a.tilesY;
    const int y0 = ty * WG_TH, x0 = tx * WG_TW;
#pragma unroll
    for (int j = 0; j < G::JD; ++j) {       // piece wave + 4j = tile row j, columns 8*wave .. 8*wave+7
      const int gy = y0 + j, gx = x0 + wave * 8 + st_q;
      const bool in = gy < a.H && gx < a.W;
      const int64_t pix = ((int64_t)n * a.H + gy) * a.W + gx;
      const void* sd = (in && st_dy_ok) ? (const void*)(a.dy + pix * a.lddy + cot * 64 + st_chunk * 8) : (const void*)snws_zero_page;
      const void* sx = (in && st_x_ok) ? (const void*)(a.x + pix * a.ldx + cit * 64 + st_chunk * 8) : (const void*)snws_zero_page;
      glds16(sd, dy_lds + (wave + 4 * j) * 1024);
      glds16(sx, x_lds + (wave + 4 * j) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (want_bias) {  // bias gradient: column sums of the dy tile, kept per wave in LDS (block-uniform branch)
      const int c8 = tid & 7;
      float bs[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bs[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pix = (tid >> 3) + 32 * j;
        float f[8];
        Vec<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(dy_lds + pix * 128 + ((c8 ^ (((pix >> 1) & 1) << 2)) * 16)), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] += f[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bs[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8) b_lds[wave * 64 + lane * 8 + e] += v;
      }
    }

    // this wave's four k-steps (pixels 64*wave .. 64*wave+63), fragments of step s+1 read while step s multiplies
    bf16x8 af[2][2], bf[2][2];
    auto load_step = [&](int s) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        af[s & 1][h] = tr_frag(abase[h] + s * 2048, abase[h] + s * 2048 + 512);
        bf[s & 1][h] = tr_frag(bbase[h] + s * 2048, bbase[h] + s * 2048 + 512);
      }
    };
    load_step(0);
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_) {
      if (s_ + 1 < 4) load_step(s_ + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s_ & 1][i], bf[s_ & 1][h], acc[i][h], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- sum the four waves' partial tiles through LDS (the staging buffers are free now: the loop's last barrier
  //      ordered every read), then one atomic flush: D[row = out channel][col = in channel]
  float* red = reinterpret_cast<float*>(smem);            // [4 waves][4 quadrants][16 regs][64 lanes] = 64 KB
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * 4 + i * 2 + h) * 16 + r) * 64 + lane] = acc[i][h][r];
  float bias_sum = 0.f;
  if (want_bias && tid < 64) bias_sum = b_lds[tid] + b_lds[tid + 64] + b_lds[tid + 128] + b_lds[tid + 192];
  __syncthreads();
  // wave w flushes quadrant w (i = w >> 1, h = w & 1)
  {
    const int i = wave >> 1, h = wave & 1;
    const int ci = cit * 64 + h * 32 + (lane & 31);
    const bool ci_ok = ci < a.Cin;
    float* po = a.part ? a.part + ((int64_t)split * npairs + pair) * 4096 + tid : nullptr;   // [r][wave = quadrant][lane]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) v += red[((w * 4 + wave) * 16 + r) * 64 + lane];
      const int co = cot * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (po) po[r * 256] = v;
      else if (ci_ok && co < a.Cout) atomicAdd(a.dw + (int64_t)co * a.Cin + ci, v);
    }
  }
  if (want_bias && tid < 64) {
    if (a.part) a.pbias[((int64_t)split * a.coTiles + cot) * 64 + tid] = bias_sum;
    else if (cot * 64 + tid < a.Cout) atomicAdd(a.dbias + cot * 64 + tid, bias_sum);
  }
}

// 1x1 wgrad for wide layers (both channel counts above 64): a 128x128 (out, in) channel tile per workgroup, one 64x64
// quadrant per wave, over tiles of 128 CONSECUTIVE pixels (a 1x1 convolution has no spatial structure, so the N*H*W
// pixels are one linear axis).  Against the 64-wide kernel every staged byte feeds twice the matrix work and the
// operands are re-read half as often.  LDS image per operand: [128 px][128 ch] bf16 (256 B per pixel, 16 chunks of
// 16 B); chunk c of pixel p sits in slot c ^ ((p & 3) << 2), so the four pixel rows of a transposing read land in
// four different 64-byte bank windows.
struct Wgrad1wArgs {
  const bf16_t* x;
  const bf16_t* dy;
  float* dw;
  float* dbias;
  float* part;
  float* pbias;
  int64_t P;                 // N*H*W pixels
  int Cin, Cout;
  int64_t ldx, lddy;
  int coTiles, ciTiles, tilesPerSplit, ntiles;
};

template <bool BIAS>
__global__ __launch_bounds__(256, 2) void conv_wgrad1w_kernel(Wgrad1wArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* dy_lds = smem;              // 32 KB
  unsigned char* x_lds = smem + 32768;       // 32 KB
  float* b_lds = reinterpret_cast<float*>(smem + 65536);   // [4 waves][128 channels]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = a.coTiles * a.ciTiles;
  const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % npairs, split = bid / npairs;
  const int cot = pair % a.coTiles, cit = pair / a.coTiles;
  const int t_begin = split * a.tilesPerSplit;
  const int t_end = t_begin + a.tilesPerSplit < a.ntiles ? t_begin + a.tilesPerSplit : a.ntiles;
  const bool want_bias = BIAS && cit == 0;
  const int co_q = wave & 1, ci_q = wave >> 1;      // this wave's 64x64 quadrant

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][h][r] = 0.f;
  if (want_bias) {
    b_lds[tid] = 0.f;
    b_lds[tid + 256] = 0.f;
  }

  // transposing-read geometry: lane -> pixel kb + gq (+4 for the second half) of a 16-pixel k-step, channels rb + 4gp ..
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int rb = (grp & 1) * 16, kb = (grp >> 1) * 8;
  const int cb0 = kb + gq;                          // (cb0 & 3) == gq, also for cb0 + 4
  const unsigned char *abase[2], *bbase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int cha = co_q * 64 + h * 32 + rb + 4 * gp, chb = ci_q * 64 + h * 32 + rb + 4 * gp;
    abase[h] = dy_lds + cb0 * 256 + (((cha >> 3) ^ (gq << 2)) * 16) + (cha & 7) * 2;
    bbase[h] = x_lds + cb0 * 256 + (((chb >> 3) ^ (gq << 2)) * 16) + (chb & 7) * 2;
  }
  // staging: a 1 KiB piece = 4 pixels x 16 chunks; lane = (pixel q, slot); slot holds chunk slot ^ ((q & 3) << 2)
  const int st_q = lane >> 4, st_chunk = (lane & 15) ^ (st_q << 2);
  const bool st_dy_ok = cot * 128 + st_chunk * 8 < a.Cout, st_x_ok = cit * 128 + st_chunk * 8 < a.Cin;
  const bf16_t* dyc = a.dy + cot * 128 + st_chunk * 8;
  const bf16_t* xc = a.x + cit * 128 + st_chunk * 8;

  for (int t = t_begin; t < t_end; ++t) {
    const int64_t p0 = (int64_t)t * 128;
#pragma unroll
    for (int j = 0; j < 8; ++j) {            // piece wave + 4j: pixels 4*(wave + 4j) .. +3
      const int64_t p = p0 + 4 * (wave + 4 * j) + st_q;
      const bool in = p < a.P;
      const void* sd = (in && st_dy_ok) ? (const void*)(dyc + p * a.lddy) : (const void*)snws_zero_page;
      const void* sx = (in && st_x_ok) ? (const void*)(xc + p * a.ldx) : (const void*)snws_zero_page;
      glds16(sd, dy_lds + (wave + 4 * j) * 1024);
      glds16(sx, x_lds + (wave + 4 * j) * 1024);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (want_bias) {  // bias gradient: column sums of the dy tile; thread = (16-byte slot, group of 8 pixels)
      const int slot = tid & 15, pg = tid >> 4;
      float bs[4][8];                      // by pixel class pix & 3: slot holds chunk slot ^ (class << 2)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[c][e] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float f[8];
        Vec<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(dy_lds + (pg * 8 + j) * 256 + slot * 16), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[j & 3][e] += f[e];
      }
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float v = bs[c][e];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          if (lane < 16) b_lds[wave * 128 + (slot ^ (c << 2)) * 8 + e] += v;
        }
    }

    bf16x8 af[2][2], bf[2][2];
    auto load_step = [&](int s_) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        af[s_ & 1][h] = tr_frag(abase[h] + s_ * 4096, abase[h] + s_ * 4096 + 1024);
        bf[s_ & 1][h] = tr_frag(bbase[h] + s_ * 4096, bbase[h] + s_ * 4096 + 1024);
      }
    };
    load_step(0);
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      if (s_ + 1 < 8) load_step(s_ + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          acc[i][h] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s_ & 1][i], bf[s_ & 1][h], acc[i][h], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- accumulate into the f32 gradient: D[row = out channel][col = in channel]
  float* po = a.part ? a.part + ((int64_t)split * npairs + pair) * 16384 + tid : nullptr;   // [i][h][r][wave][lane]
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ci = cit * 128 + ci_q * 64 + h * 32 + (lane & 31);
      const bool ci_ok = ci < a.Cin;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = cot * 128 + co_q * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (po) po[((i * 2 + h) * 16 + r) * 256] = acc[i][h][r];
        else if (ci_ok && co < a.Cout) atomicAdd(a.dw + (int64_t)co * a.Cin + ci, acc[i][h][r]);
      }
    }
  }
  if (want_bias && tid < 128) {   // the loop's last barrier ordered every wave's partials
    const float v = b_lds[tid] + b_lds[tid + 128] + b_lds[tid + 256] + b_lds[tid + 384];
    if (a.part) a.pbias[((int64_t)split * a.coTiles + cot) * 128 + tid] = v;
    else if (cot * 128 + tid < a.Cout) atomicAdd(a.dbias + cot * 128 + tid, v);
  }
}

// 1x1 wgrad for filters with a multiple of 192 output channels - the tap form of conv(upsample(x)), 9 x 64 = 576 outputs
// (modeling/UNet3p_SNws.py:131-155 of the reference) from 320 or 1024 inputs: a 192 x 160 (out, in) channel tile per
// workgroup divides 576 x 320 without padding (the 128 x 128 tile above wastes a quarter of its work there, the 64 x 64
// one re-reads the pixels for 45 tile pairs instead of 6).  Wave (wo, wi) owns 96 x 80 = 6 x 5 blocks of 16 x 16 on
// v_mfma_f32_16x16x32_bf16: 11 fragment reads per 30 matrix instructions.  Pixels are one linear axis, 96 per tile
// (three 32-pixel k-steps); LDS image per operand [96 px][channels] bf16, dy rows padded from 384 to 400 bytes so the
// four pixel rows of a transposing read fall into different bank windows (x rows, 320 bytes, already do).
struct Wgrad1tGeom {
  static constexpr int TM = 192, TN = 160, TP = 96;
  static constexpr int DY_SL = 25, X_SL = 20;                            // 16-byte slots per pixel (dy: 24 + 1 pad)
  static constexpr int DY_PIXB = DY_SL * 16, X_PIXB = X_SL * 16;
  static constexpr int DY_NS = TP * DY_SL, X_NS = TP * X_SL;
  static constexpr int DY_NI = (DY_NS + 63) / 64, X_NI = (X_NS + 63) / 64;      // 38, 30
  static constexpr int DY_BYTES = DY_NI * 1024, X_BYTES = X_NI * 1024;
  static constexpr int LDS_BYTES = DY_BYTES + X_BYTES;
  static constexpr int E = TM * TN;
};

template <bool BIAS>
__global__ __launch_bounds__(256, 2) void conv_wgrad1t_kernel(Wgrad1wArgs a) {
  using G = Wgrad1tGeom;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* dy_lds = smem;
  unsigned char* x_lds = smem + G::DY_BYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = a.coTiles * a.ciTiles;
  const int bid = (int)xcd_remap(blockIdx.x, gridDim.x);
  const int pair = bid % npairs, split = bid / npairs;
  const int cot = pair % a.coTiles, cit = pair / a.coTiles;
  const int t_begin = split * a.tilesPerSplit;
  const int t_end = t_begin + a.tilesPerSplit < a.ntiles ? t_begin + a.tilesPerSplit : a.ntiles;
  const bool want_bias = BIAS && cit == 0;
  const int wo = wave & 1, wi = wave >> 1;

  f32x4 acc[6][5];
#pragma unroll
  for (int ob = 0; ob < 6; ++ob)
#pragma unroll
    for (int ib = 0; ib < 5; ++ib) acc[ob][ib] = f32x4{0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;       // BIAS: thread tid < 192 sums out channel tid of the dy tiles

  // transposing-read geometry: lane -> pixel 8 g + tq (+4 for the second half) of a 32-pixel k-step, channels 4 tp ..
  const int g = lane >> 4, tq = (lane >> 2) & 3, tp = lane & 3;
  const unsigned char* abase = dy_lds + (8 * g + tq) * G::DY_PIXB + (wo * 96 + 4 * tp) * 2;
  const unsigned char* bbase = x_lds + (8 * g + tq) * G::X_PIXB + (wi * 80 + 4 * tp) * 2;

  for (int t = t_begin; t < t_end; ++t) {
    const int64_t p0 = (int64_t)t * G::TP;
#pragma unroll
    for (int j = 0; j < (G::DY_NI + 3) / 4; ++j) {
      const int piece = wave + 4 * j;
      if (piece < G::DY_NI) {
        const int s = piece * 64 + lane;
        const int px = s / G::DY_SL, c = s - px * G::DY_SL;
        const bool ok = s < G::DY_NS && c < 24 && p0 + px < a.P;
        const void* src = ok ? (const void*)(a.dy + (p0 + px) * a.lddy + cot * G::TM + c * 8) : (const void*)snws_zero_page;
        glds16(src, dy_lds + piece * 1024);
      }
    }
#pragma unroll
    for (int j = 0; j < (G::X_NI + 3) / 4; ++j) {
      const int piece = wave + 4 * j;
      if (piece < G::X_NI) {
        const int s = piece * 64 + lane;
        const int px = s / G::X_SL, c = s - px * G::X_SL;
        const bool ok = p0 + px < a.P && cit * G::TN + c * 8 < a.Cin;
        const void* src = ok ? (const void*)(a.x + (p0 + px) * a.ldx + cit * G::TN + c * 8) : (const void*)snws_zero_page;
        glds16(src, x_lds + piece * 1024);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    if (want_bias && tid < G::TM) {   // block-uniform branch; rare (the tap convolutions have no bias)
      float s = 0.f;
      for (int px = 0; px < G::TP; ++px)
        s += (float)*reinterpret_cast<const bf16_t*>(dy_lds + px * G::DY_PIXB + tid * 2);
      bsum += s;
    }

    // A fragments (dy, 6 per k-step) in two register sets, loaded a k-step ahead; B fragments (x) roll one ahead
    bf16x8 af[2][6], bq[2];
    auto load_a = [&](int s_) {
#pragma unroll
      for (int ob = 0; ob < 6; ++ob)
        af[s_ & 1][ob] = tr_frag(abase + s_ * 32 * G::DY_PIXB + ob * 32, abase + s_ * 32 * G::DY_PIXB + ob * 32 + 4 * G::DY_PIXB);
    };
    auto load_b = [&](int f) {      // f = 5 * k-step + in block
      const int s_ = f / 5, ib = f % 5;
      bq[f & 1] = tr_frag(bbase + s_ * 32 * G::X_PIXB + ib * 32, bbase + s_ * 32 * G::X_PIXB + ib * 32 + 4 * G::X_PIXB);
    };
    load_a(0);
    load_b(0);
#pragma unroll
    for (int f = 0; f < 15; ++f) {
      const int s_ = f / 5, ib = f % 5;
      if (f + 1 < 15) load_b(f + 1);
      if (ib == 0 && s_ + 1 < 3) load_a(s_ + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ob = 0; ob < 6; ++ob)
        acc[ob][ib] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[s_ & 1][ob], bq[f & 1], acc[ob][ib], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
  }

  // ---- flush: D[out channel 16 ob + 4 (lane >> 4) + j][in channel 16 ib + (lane & 15)] of this wave's quadrant
  if (a.part) {   // partial tile [ob][ib][j][wave][lane]
    float* po = a.part + ((int64_t)split * npairs + pair) * G::E + tid;
#pragma unroll
    for (int ob = 0; ob < 6; ++ob)
#pragma unroll
      for (int ib = 0; ib < 5; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j) po[((ob * 5 + ib) * 4 + j) * 256] = acc[ob][ib][j];
  } else {
#pragma unroll
    for (int ob = 0; ob < 6; ++ob)
#pragma unroll
      for (int ib = 0; ib < 5; ++ib)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int co = cot * G::TM + wo * 96 + ob * 16 + 4 * (lane >> 4) + j;
          const int ci = cit * G::TN + wi * 80 + ib * 16 + (lane & 15);
          if (ci < a.Cin && co < a.Cout) atomicAdd(a.dw + (int64_t)co * a.Cin + ci, acc[ob][ib][j]);
        }
  }
  if (want_bias && tid < G::TM) {
    if (a.part) a.pbias[((int64_t)split * a.coTiles + cot) * G::TM + tid] = bsum;
    else if (cot * G::TM + tid < a.Cout) atomicAdd(a.dbias + cot * G::TM + tid, bsum);
  }
}

// 3x3 wgrad over 64-wide in-channel tiles.  The generic kernel above reads one B fragment from LDS per MFMA, which
// asks the LDS for its full 128 B/clk at the matrix pipe's peak rate.  Here a wave owns all 64 out channels x one 16-wide
// in-channel QUARTER x all nine taps (v_mfma_f32_16x16x32_bf16: a k-step is one 32-pixel tile row, 4 output blocks of
// 16 x 16 per tap).  The B fragment of (halo row rho, kw) serves the three taps (kh, kw) at the k-steps rho - kh, so a
// tile needs 30 B fragments and 32 A fragments for its 288 matrix instructions, and the four waves carry equal work.
// The walk is over halo rows: A fragments of the three live k-steps sit in a rolling three-slot set, the next halo
// row's B fragments and the next k-step's A fragments are read while the current row multiplies.  All LDS addresses are
// a per-lane base plus an immediate.  (Retired after same-box measurements: a split by in-channel half x 5 / 4 taps,
// which read 100 fragments per tile and loaded the waves unequally, and the 32x32x16 instruction shape, under which the
// chip holds a lower clock.)
template <bool BIAS, bool PRIO>
__device__ __forceinline__ void wgrad9_body(const WgradArgs& a, unsigned char* smem, int bid) {
  using G = WgGeom<64, 9>;
  unsigned char* dy_lds = smem;
  unsigned char* x_lds = smem + G::DY_BYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npairs = a.coTiles * a.ciTiles;
  const int pair = bid % npairs, split = bid / npairs;
  const int cot = pair % a.coTiles, cit = pair / a.coTiles;
  const int ntiles = a.N * a.tilesY * a.tilesX;
  const int t_begin = split * a.tilesPerSplit;
  const int t_end = t_begin + a.tilesPerSplit < ntiles ? t_begin + a.tilesPerSplit : ntiles;
  const bool want_bias = BIAS && cit == 0;

  f32x4 acc16[9][4];    // [tap][out block of 16]
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc16[i][ob] = f32x4{0.f, 0.f, 0.f, 0.f};
  float* b_lds = reinterpret_cast<float*>(smem + G::DY_BYTES + ((G::X_BYTES + 15) / 16) * 16);   // [4 waves][64 channels]
  if (want_bias) b_lds[tid] = 0.f;   // ordered before its first use by the barrier that follows the first staging

  // transposing-read lane geometry (see the generic kernel): lane -> (channel group of 4, pixel within the k-step)
  // (a 16-lane group reads the SAME 16 channels at the k octet 8 * group of a 32-pixel k-step)
  const int grp = lane >> 4, gq = (lane >> 2) & 3, gp = lane & 3;
  const int cb0 = grp * 8 + gq;   // this lane's first pixel column inside a k-step; the second is cb0 + 4
  // dy tile: pixel p = row*32 + col; its chunk swizzle bit (p >> 1) & 1 only depends on cb0
  // (abase[h] serves out blocks 2h and, 32 bytes further, 2h + 1)
  const unsigned char* abase[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int ach = h * 32 + 4 * gp;
    abase[h] = dy_lds + cb0 * 128 + (((ach >> 3) ^ (((cb0 >> 1) & 1) << 2)) * 16) + (ach & 7) * 2;
  }
  // x halo: pixel q = (row + kh)*34 + col + kw; swizzle bit (q >> 1) & 1 = (((cb0 + kw) >> 1) + row + kh) & 1, so an
  // odd tile row flips chunk bit 2 (byte-offset bit 6) relative to an even one
  // (one base per kw and halo-row parity, the halo row enters as an immediate)
  const unsigned char* bbase[3][2];
  {
    const int bch = wave * 16 + 4 * gp;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int sw = (((cb0 + kw) >> 1) & 1) << 2;
      const int off = (cb0 + kw) * 128 + (((bch >> 3) ^ sw) * 16) + (bch & 7) * 2;
      bbase[kw][0] = x_lds + off;
      bbase[kw][1] = x_lds + (off ^ 64);
    }
  }

  // per-lane staging constants: lane = (pixel q of a piece's 8, 16-byte channel chunk c); the chunk swizzle bit
  // (pix >> 1) & 1 of pixel 8*i + q only depends on q
  const int st_q = lane >> 3, st_chunk = (lane & 7) ^ (((st_q >> 1) & 1) << 2);
  const bool st_dy_ok = cot * 64 + st_chunk * 8 < a.Cout, st_x_ok = cit * 64 + st_chunk * 8 < a.Cin;
  const unsigned st_dy_off = (unsigned)(((wave * 8 + st_q) * a.lddy + st_chunk * 8) * 2);
  const int st_px0 = wave * 8 + st_q;                                  // halo pixel 8*wave + q lies in halo row 0
  const unsigned st_x_off = (unsigned)((st_px0 * a.ldx + st_chunk * 8) * 2);
#ifdef SNWS_TIMING
  long long dbg_wait = 0, dbg_comp = 0;
#endif
  for (int t = t_begin; t < t_end; ++t) {
#ifdef SNWS_TIMING
    const long long dbg_ta = clock64();
#endif
    int tt = t;
    const int tx = tt % a.tilesX; tt /= a.tilesX;
    const int ty = tt % a.tilesY;
    const int n = tt / a.tilesY;
    const int y0 = ty * WG_TH, x0 = tx * WG_TW;
    if (y0 >= 1 && y0 + WG_TH < a.H && x0 >= 1 && x0 + WG_TW < a.W) {
      // interior tile (the common case): no per-pixel bounds tests; addresses = tile base + a per-lane offset that
      // advances by a two-way recurrence from piece to piece (a piece is 8 consecutive halo pixels x 64 channels)
      const unsigned char* dyb = reinterpret_cast<const unsigned char*>(a.dy + (((int64_t)n * a.H + y0) * a.W + x0) * a.lddy + cot * 64);
      const int64_t dyrow = (int64_t)a.W * a.lddy * 2;
#pragma unroll
      for (int j = 0; j < G::JD; ++j) {       // piece wave + 4j = tile row j, columns 8*wave..8*wave+7
        const void* s_ = st_dy_ok ? (const void*)(dyb + j * dyrow + st_dy_off) : (const void*)snws_zero_page;
        glds16(s_, dy_lds + (wave + 4 * j) * 1024);
      }
      const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x + (((int64_t)n * a.H + y0 - 1) * a.W + x0 - 1) * a.ldx + cit * 64);
      const unsigned rowb = (unsigned)a.W * (unsigned)a.ldx * 2u, pixb = (unsigned)a.ldx * 2u;
      int px_ = st_px0;
      unsigned xo = st_x_off;
#pragma unroll
      for (int j = 0; j < G::JX; ++j) {
        const int i = wave + 4 * j;
        if (i < G::XNI) {
          const bool ok = st_x_ok && (j < G::JX - 1 || i * 8 + (lane >> 3) < G::NPIX);
          const void* s_ = ok ? (const void*)(xb + xo) : (const void*)snws_zero_page;
          glds16(s_, x_lds + i * 1024);
          // next piece of this wave: 32 halo pixels further in the 34-wide halo rows
          const bool wrap = px_ >= 2;
          xo += wrap ? rowb - 2u * pixb : 32u * pixb;
          px_ += wrap ? -2 : 32;
        }
      }
    } else {
      // border tile: per-pixel bounds tests, recomputed from an opaque copy of the lane id so that their per-lane
      // invariants are not hoisted into registers the accumulator-heavy main loop needs
      int ln = lane;
      asm volatile("" : "+v"(ln));
  #pragma unroll
      for (int j = 0; j < G::JD; ++j) {
        const int i = wave + 4 * j;
        const int pix = i * 8 + (ln >> 3);
        const int gy = y0 + (pix >> 5), gx = x0 + (pix & 31);
        const int chunk = (ln & 7) ^ (((pix >> 1) & 1) << 2);
        const bool ok = gy < a.H && gx < a.W && cot * 64 + chunk * 8 < a.Cout;
        const void* s = ok ? (const void*)(a.dy + (((int64_t)n * a.H + gy) * a.W + gx) * a.lddy + cot * 64 + chunk * 8)
                           : (const void*)snws_zero_page;
        glds16(s, dy_lds + i * 1024);
      }
  #pragma unroll
      for (int j = 0; j < G::JX; ++j) {
        const int i = wave + 4 * j;
        if (i < G::XNI) {
          const int pix = i * 8 + (ln >> 3);
          const int py = pix / G::HWd, px = pix - py * G::HWd;
          const int gy = y0 - 1 + py, gx = x0 - 1 + px;
          const int chunk = (ln & 7) ^ (((pix >> 1) & 1) << 2);
          const bool ok = pix < G::NPIX && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W && cit * 64 + chunk * 8 < a.Cin;
          const void* s = ok ? (const void*)(a.x + (((int64_t)n * a.H + gy) * a.W + gx) * a.ldx + cit * 64 + chunk * 8)
                             : (const void*)snws_zero_page;
          glds16(s, x_lds + i * 1024);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SNWS_TIMING
    const long long dbg_tb = clock64();
    dbg_wait += dbg_tb - dbg_ta;
#endif

    if (want_bias) {  // bias gradient: column sums of the dy tile, kept per wave in LDS (block-uniform branch)
      const int c8 = tid & 7;
      float bs[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) bs[e] = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int pix = (tid >> 3) + 32 * j;
        float f[8];
        Vec<bf16_t, 8>::load(reinterpret_cast<const bf16_t*>(dy_lds + pix * 128 + ((c8 ^ (((pix >> 1) & 1) << 2)) * 16)), f);
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] += f[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float v = bs[e];
        v += __shfl_xor(v, 8, 64);
        v += __shfl_xor(v, 16, 64);
        v += __shfl_xor(v, 32, 64);
        if (lane < 8) b_lds[wave * 64 + lane * 8 + e] += v;
      }
    }

    {
    bf16x8 af[3][4], bq[2][3];
    auto load_a1 = [&](int ks, int ob) {
      const int a_imm = ks * 32 * 128 + (ob & 1) * 32;
      af[ks % 3][ob] = tr_frag(abase[ob >> 1] + a_imm, abase[ob >> 1] + a_imm + 512);
    };
    auto load_b1 = [&](int rho, int kw) {
      const int b_imm = rho * G::HWd * 128;
      bq[rho & 1][kw] = tr_frag(bbase[kw][rho & 1] + b_imm, bbase[kw][rho & 1] + b_imm + 512);
    };
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) load_a1(0, ob);
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) load_b1(0, kw);
#pragma unroll
    for (int rho = 0; rho < 10; ++rho) {
      // groups of this halo row in the order kh = 2, 1, 0 (the oldest k-step first: its A slot is refilled next)
      const int nkh2 = (rho >= 2 && rho - 2 < 8) ? 3 : 0;
      const int ngroups = ((rho >= 2 && rho - 2 < 8) ? 3 : 0) + ((rho >= 1 && rho - 1 < 8) ? 3 : 0) + (rho < 8 ? 3 : 0);
      const bool want_a = rho + 1 < 8, want_b = rho + 1 < 10;
      if constexpr (PRIO) {     // (compile-time: a branch here would cut the unrolled walk into blocks and spill)
        if (rho == 0) __builtin_amdgcn_s_setprio(1);
        if (rho == 4) __builtin_amdgcn_s_setprio(2);
        if (rho == 7) __builtin_amdgcn_s_setprio(3);
      }
      const int a_per = (4 + (ngroups - nkh2) - 1) / (ngroups - nkh2 > 0 ? ngroups - nkh2 : 1);
      int gi = 0, a_next = 0;
#pragma unroll
      for (int kh = 2; kh >= 0; --kh) {
        const int ks = rho - kh;
        if (ks < 0 || ks >= 8) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          if (want_b && gi < 3) load_b1(rho + 1, gi);
          if (want_a && gi >= nkh2) {
#pragma unroll
            for (int u = 0; u < a_per; ++u)
              if (a_next < 4) { load_a1(rho + 1, a_next); ++a_next; }
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ob = 0; ob < 4; ++ob)
            acc16[kh * 3 + kw][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks % 3][ob], bq[rho & 1][kw], acc16[kh * 3 + kw][ob], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          ++gi;
        }
      }
    }
    if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
    }
    __syncthreads();
#ifdef SNWS_TIMING
    dbg_comp += clock64() - dbg_tb;
#endif
  }
#ifdef SNWS_TIMING
  if (tid == 0) {
    atomicAdd(&snws_dbg[4], (unsigned long long)dbg_wait);
    atomicAdd(&snws_dbg[5], (unsigned long long)dbg_comp);
    atomicAdd(&snws_dbg[6], (unsigned long long)(t_end - t_begin));
    atomicAdd(&snws_dbg[7], 1ull);
  }
#endif

  // ---- accumulate into the f32 gradient: D[row = out channel][col = in channel]
  // D[out channel 16 ob + 4 (lane >> 4) + j][in channel 16 wave + (lane & 15)]; the wave's in-channel quarter is block
  // (wave & 1) of half (wave >> 1) in the partial-tile layout [tap][ob][in block][j][in-channel half][lane]
  {
    const int fl_half = wave >> 1, fl_cb0 = wave & 1;
    if (a.part) {
      float* o = a.part + ((int64_t)split * npairs + pair) * 36864 + fl_half * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[(((i * 4 + ob) * 2 + fl_cb0) * 4 + j) * 128] = acc16[i][ob][j];
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = cot * 64 + ob * 16 + 4 * (lane >> 4) + j;
            const int ci16 = cit * 64 + fl_half * 32 + fl_cb0 * 16 + (lane & 15);
            if (ci16 < a.Cin && co < a.Cout) atomicAdd(a.dw + ((int64_t)co * 9 + i) * a.Cin + ci16, acc16[i][ob][j]);
          }
    }
  }
  if (want_bias && tid < 64) {   // the loop's last barrier ordered every wave's partials
    const float v = b_lds[tid] + b_lds[tid + 64] + b_lds[tid + 128] + b_lds[tid + 192];
    if (a.part) a.pbias[((int64_t)split * a.coTiles + cot) * 64 + tid] = v;
    else if (cot * 64 + tid < a.Cout) atomicAdd(a.dbias + cot * 64 + tid, v);
  }
}

template <bool BIAS, bool PRIO>
__global__ __launch_bounds__(256, 2) void conv_wgrad9q_kernel(WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  wgrad9_body<BIAS, PRIO>(a, smem, (int)xcd_remap(blockIdx.x, gridDim.x));
}

// ------------------------------------------------------------------------------------------------- 3x3 wgrad, ping-pong
// conv_wgrad9q_kernel relies on two independent workgroups per CU to cover each other's staging; left to themselves
// the two drift into lock-step (multiply together at half speed each, then stage together with the matrix pipe idle).
// Here ONE 8-wave workgroup per CU holds both roles and alternates them by construction: waves 0-3 (group A) and
// waves 4-7 (group B) each own an (out tile, in tile) pair - one wave of each group per SIMD - and every phase ends in
// a workgroup barrier: while one group multiplies its pixel tile, the other issues the LDS-DMA copies for its next
// one and waits for them.  The matrix pipe never sees two waves at once and never waits for a copy that fits into
// one matrix phase.  When the two pairs share the in-channel tile and the pixel range (neighbouring out tiles of one
// launch: the common case) they also share the x halo tile: it is staged ONCE per pixel tile, half by each group,
// into one of two x buffers (the next tile's copy lands while this tile is still being read), so a phase stages 54 KB
// instead of 76 KB.  Otherwise each group stages its own x tile into "its" buffer.  Partial tiles and their layout are
// those of conv_wgrad9q_kernel (same reduction kernel, bit-identical results: the per-wave accumulation order over
// pixel tiles is the same).  No bias gradient: inside a Conv2dws -> SwitchNorm block it comes in closed form from the
// SwitchNorm backward, and a launch that does want it takes conv_wgrad9q_kernel.
struct W9pItem {
  int cot, cit, split, pair, t_begin, n;   // n = pixel tiles of this item (0: no such item)
};

__device__ __forceinline__ W9pItem w9p_item(const WgradArgs& a, int item, int nitems, int npairs, int ntiles) {
  W9pItem it;
  it.pair = item % npairs;
  it.split = item / npairs;
  it.cot = it.pair % a.coTiles;
  it.cit = it.pair / a.coTiles;
  it.t_begin = it.split * a.tilesPerSplit;
  int t_end = it.t_begin + a.tilesPerSplit;
  if (t_end > ntiles) t_end = ntiles;
  it.n = item < nitems ? t_end - it.t_begin : 0;
  return it;
}

// rounds [J0, J1) of the x halo tile of pixel tile t (piece lw + 4j: 8 halo pixels x 64 channels = 1 KB).  The tile
// base is wave-uniform; a lane's offset inside the halo walks from piece to piece by a two-way recurrence (32 halo
// pixels further in the 34-wide halo rows), so a piece costs a handful of vector instructions and the copies go out
// back to back - the group that stages has one matrix phase of its partner to get them issued AND landed.
template <int J0, int J1>
__device__ __forceinline__ void w9p_stage_x(const WgradArgs& a, unsigned char* x_lds, int lw, int lane, int t, int cit) {
  using G = WgGeom<64, 9>;
  // per-lane constants are recomputed from an opaque copy of the lane id on every call: hoisted out of the tile loop
  // they would sit in registers across the accumulator-heavy matrix phase and spill - and a spill reload in HERE
  // waits for vmcnt(0), i.e. for every copy issued so far to land
  asm volatile("" : "+v"(lane));
  int tt = t;
  const int tx = tt % a.tilesX; tt /= a.tilesX;
  const int ty = tt % a.tilesY;
  const int n = tt / a.tilesY;
  const int y0 = ty * WG_TH, x0 = tx * WG_TW;
  const int q = lane >> 3;
  const int chunk = (lane & 7) ^ (((q >> 1) & 1) << 2);
  const bool ch_ok = cit * 64 + chunk * 8 < a.Cin;
  const bool interior = y0 >= 1 && y0 + WG_TH < a.H && x0 >= 1 && x0 + WG_TW < a.W;
  // (for a border tile the base may point in front of the image: it is only dereferenced for pixels inside)
  const unsigned char* xb = reinterpret_cast<const unsigned char*>(a.x + (((int64_t)n * a.H + y0 - 1) * a.W + x0 - 1) * a.ldx + cit * 64);
  const int rowb = a.W * (int)a.ldx * 2, pixb = (int)a.ldx * 2;
  const int pix0 = 8 * lw + q + 32 * J0;
  int py = pix0 / G::HWd, px = pix0 - py * G::HWd;
  int xo = py * rowb + px * pixb + chunk * 16;
  if (interior) {
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      const int i = lw + 4 * j;
      if (i < G::XNI) {
        const bool ok = ch_ok && (j < G::JX - 1 || py < G::HHt);
        const void* s_ = ok ? (const void*)(xb + xo) : (const void*)snws_zero_page;
        glds16(s_, x_lds + i * 1024);
        const bool wrap = px >= 2;
        xo += wrap ? rowb - 2 * pixb : 32 * pixb;
        px += wrap ? -2 : 32;
        py += wrap ? 1 : 0;
      }
    }
  } else {
#pragma unroll
    for (int j = J0; j < J1; ++j) {
      const int i = lw + 4 * j;
      if (i < G::XNI) {
        const int gy = y0 - 1 + py, gx = x0 - 1 + px;
        const bool ok = ch_ok && py < G::HHt && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W;
        const void* s_ = ok ? (const void*)(xb + xo) : (const void*)snws_zero_page;
        glds16(s_, x_lds + i * 1024);
        const bool wrap = px >= 2;
        xo += wrap ? rowb - 2 * pixb : 32 * pixb;
        px += wrap ? -2 : 32;
        py += wrap ? 1 : 0;
      }
    }
  }
}

__device__ __forceinline__ void w9p_stage_dy(const WgradArgs& a, unsigned char* dy_lds, int lw, int lane, int t, int cot) {
  using G = WgGeom<64, 9>;
  asm volatile("" : "+v"(lane));
  int tt = t;
  const int tx = tt % a.tilesX; tt /= a.tilesX;
  const int ty = tt % a.tilesY;
  const int n = tt / a.tilesY;
  const int y0 = ty * WG_TH, x0 = tx * WG_TW;
  const int q = lane >> 3;
  const int chunk = (lane & 7) ^ (((q >> 1) & 1) << 2);
  const int gx = x0 + 8 * lw + q;
  const bool col_ok = cot * 64 + chunk * 8 < a.Cout && gx < a.W;
  const unsigned char* dyb = reinterpret_cast<const unsigned char*>(a.dy + (((int64_t)n * a.H + y0) * a.W + x0) * a.lddy + cot * 64);
  const int rowb = a.W * (int)a.lddy * 2;
  int off = (8 * lw + q) * (int)a.lddy * 2 + chunk * 16;
#pragma unroll
  for (int j = 0; j < G::JD; ++j) {      // piece lw + 4j = tile row j, columns 8*lw .. 8*lw + 7
    const bool ok = col_ok && y0 + j < a.H;
    const void* s_ = ok ? (const void*)(dyb + off) : (const void*)snws_zero_page;
    glds16(s_, dy_lds + (lw + 4 * j) * 1024);
    off += rowb;
  }
}

#ifndef W9P_PRIO
#define W9P_PRIO 2
#endif
#ifndef W9P_NOSTAGE
#define W9P_NOSTAGE 0      // 1 (diagnostics only, wrong results): no copies after the first tile - the matrix phases alone
#endif
__global__ __launch_bounds__(512) void conv_wgrad9p_kernel(WgradArgs a) {
  using G = WgGeom<64, 9>;
  constexpr int XB = ((G::X_BYTES + 15) / 16) * 16;
  constexpr int JX = G::JX, JXH = (G::JX + 1) / 2;        // x rounds: all / the first group's share of a shared tile
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const xbuf = smem;                        // two x halo buffers
  unsigned char* const dybuf = smem + 2 * XB;              // one dy tile per group

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, lw = wave & 3;
  const int npairs = a.coTiles * a.ciTiles;
  const int ntiles = a.N * a.tilesY * a.tilesX;
  const int nitems = npairs * ((ntiles + a.tilesPerSplit - 1) / a.tilesPerSplit);
  const int wg = (int)xcd_remap(blockIdx.x, gridDim.x);
  const W9pItem itA = w9p_item(a, 2 * wg, nitems, npairs, ntiles), itB = w9p_item(a, 2 * wg + 1, nitems, npairs, ntiles);
  const W9pItem me = grp ? itB : itA;
  const bool shared = itB.n > 0 && itA.split == itB.split && itA.cit == itB.cit;      // then itA.n == itB.n
  const int nloop = itA.n > itB.n ? itA.n : itB.n;
  unsigned char* const dy_lds = dybuf + grp * G::DY_BYTES;

  // v_mfma_f32_16x16x32_bf16; a wave owns all 64 out channels x in-channel quarter lw (the split of conv_wgrad9q_kernel).
  // (A 32x32x16 form of this walk - out half x in half per wave, the transposing reads hidden in the 24 issue cycles
  // that shape leaves free - needed 5 % fewer cycles per tile and ran 6 % slower: the chip clocks lower under that shape.)
  f32x4 acc16[9][4];
#pragma unroll
  for (int i = 0; i < 9; ++i)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) acc16[i][ob] = f32x4{0.f, 0.f, 0.f, 0.f};

  // one pixel tile: the matrix walk of conv_wgrad9q_kernel over halo rows
  auto multiply = [&](const unsigned char* x_lds) __attribute__((always_inline)) {
    // the multiplying wave outranks its SIMD partner, which is issuing copies: an MFMA leaves half of its issue
    // cycles to other vector instructions, enough for the partner's address arithmetic
    __builtin_amdgcn_s_setprio(W9P_PRIO);
    // transposing-read lane geometry of conv_wgrad9q_kernel (16x16x32, quarter-wave split), rebuilt per tile from an
    // opaque copy of the lane id: eight pointers that would otherwise stay live (and spill) across the staging phases
    int ln = lane;
    asm volatile("" : "+v"(ln));
    const int grp16 = ln >> 4, gq = (ln >> 2) & 3, gp = ln & 3;
    const int cb0 = grp16 * 8 + gq;
    const unsigned char* abase[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int ach = h * 32 + 4 * gp;
      abase[h] = dy_lds + cb0 * 128 + (((ach >> 3) ^ (((cb0 >> 1) & 1) << 2)) * 16) + (ach & 7) * 2;
    }
    const unsigned char* bb[3][2];      // [kw][halo-row parity]
    {
      const int bch = lw * 16 + 4 * gp;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int sw = (((cb0 + kw) >> 1) & 1) << 2;
        const unsigned boff = (unsigned)((cb0 + kw) * 128 + (((bch >> 3) ^ sw) * 16) + (bch & 7) * 2);
        bb[kw][0] = x_lds + boff;
        bb[kw][1] = x_lds + (boff ^ 64u);
      }
    }
    // Fragments are kept as their two 8-byte transposing reads, so that ONE read can be placed behind each matrix
    // instruction: this wave is alone on its SIMD's matrix pipe during its phase (its partner stages), and a matrix
    // instruction leaves the SIMD only half of its 16 issue cycles - a burst of reads in front of a group of four would
    // stall the pipe for its length (two co-resident multiplying waves, as in conv_wgrad9q_kernel, cover each other's
    // bursts; here nobody does).
    short4v alo[3][4], ahi[3][4], blo[2][3], bhi[2][3];
    auto rd = [&](const unsigned char* p_) {
      return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p_);
    };
    auto load_a_half = [&](int ks, int ob, int half) {
      const int a_imm = ks * 32 * 128 + (ob & 1) * 32 + half * 512;
      if (half == 0) alo[ks % 3][ob] = rd(abase[ob >> 1] + a_imm);
      else ahi[ks % 3][ob] = rd(abase[ob >> 1] + a_imm);
    };
    auto load_b_half = [&](int rho, int kw, int half) {
      const int b_imm = rho * G::HWd * 128 + half * 512;
      if (half == 0) blo[rho & 1][kw] = rd(bb[kw][rho & 1] + b_imm);
      else bhi[rho & 1][kw] = rd(bb[kw][rho & 1] + b_imm);
    };
    auto frag = [&](const short4v& lo, const short4v& hi) {
      short8v v;
      v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3];
      v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
      return __builtin_bit_cast(bf16x8, v);
    };
#pragma unroll
    for (int h = 0; h < 8; ++h) load_a_half(0, h >> 1, h & 1);
#pragma unroll
    for (int h = 0; h < 6; ++h) load_b_half(0, h >> 1, h & 1);
#pragma unroll
    for (int rho = 0; rho < 10; ++rho) {
      // this halo row's matrix instructions ("slots"), in the order kh = 2, 1, 0 (the oldest k-step first: its A slot is
      // the one the next k-step's fragments go to, so those reads start behind the kh = 2 groups)
      const int nkh2 = (rho >= 2 && rho - 2 < 8) ? 12 : 0;
      const int nkh1 = (rho >= 1 && rho - 1 < 8) ? 12 : 0;
      const int nslots = nkh2 + nkh1 + (rho < 8 ? 12 : 0);
      const int nb = rho + 1 < 10 ? 6 : 0, na = rho + 1 < 8 ? 8 : 0;       // 8-byte reads to place in this row
      // B read h goes behind slot h * sb, A read j behind slot nkh2 + j * sa (all inside the row for every rho)
      const int sb = nb ? nslots / nb : 1, sa = na ? (nslots - nkh2) / na : 1;
#pragma unroll
      for (int kh = 2; kh >= 0; --kh) {
        const int ks = rho - kh;
        if (ks >= 0 && ks < 8) {
          const int before = (kh < 2 ? nkh2 : 0) + (kh < 1 ? nkh1 : 0);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
#pragma unroll
            for (int ob = 0; ob < 4; ++ob) {
              const int slot = before + kw * 4 + ob;
              acc16[kh * 3 + kw][ob] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag(alo[ks % 3][ob], ahi[ks % 3][ob]),
                                                                               frag(blo[rho & 1][kw], bhi[rho & 1][kw]),
                                                                               acc16[kh * 3 + kw][ob], 0, 0, 0);
              if (nb && slot % sb == 0 && slot / sb < nb) load_b_half(rho + 1, (slot / sb) >> 1, (slot / sb) & 1);
              if (na && slot >= nkh2 && (slot - nkh2) % sa == 0 && (slot - nkh2) / sa < na)
                load_a_half(rho + 1, ((slot - nkh2) / sa) >> 1, ((slot - nkh2) / sa) & 1);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // ---- prologue: the first x tile (group A's; both groups' when shared) and group A's first dy tile
  if (shared) {
    if (grp == 0) { w9p_stage_x<0, JXH>(a, xbuf, lw, lane, itA.t_begin, itA.cit); w9p_stage_dy(a, dy_lds, lw, lane, itA.t_begin, itA.cot); }
    else w9p_stage_x<JXH, JX>(a, xbuf, lw, lane, itA.t_begin, itA.cit);
  } else if (grp == 0 && itA.n > 0) {
    w9p_stage_x<0, JX>(a, xbuf, lw, lane, itA.t_begin, itA.cit);
    w9p_stage_dy(a, dy_lds, lw, lane, itA.t_begin, itA.cot);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

#ifdef SNWS_TIMING
  long long dbg[6] = {0, 0, 0, 0, 0, 0};
#define W9P_T(v) const long long v = clock64()
#else
#define W9P_T(v)
#endif
  for (int k = 0; k < nloop; ++k) {
    // ---- phase 2k: group A multiplies tile k; group B stages its tile k (and half of the shared x tile k + 1)
    W9P_T(t0);
    if (grp == 0) {
      if (k < itA.n) multiply(xbuf + (shared ? (k & 1) * XB : 0));
    } else if (!W9P_NOSTAGE) {
      if (shared) {
        if (k + 1 < nloop) w9p_stage_x<0, JXH>(a, xbuf + ((k + 1) & 1) * XB, lw, lane, itA.t_begin + k + 1, itA.cit);
      } else if (k < itB.n) {
        w9p_stage_x<0, JX>(a, xbuf + XB, lw, lane, itB.t_begin + k, itB.cit);
      }
      if (k < itB.n) w9p_stage_dy(a, dy_lds, lw, lane, itB.t_begin + k, itB.cot);
      W9P_T(t1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SNWS_TIMING
      dbg[2] += t1 - t0; dbg[3] += clock64() - t1;
#endif
    }
    W9P_T(t2);
    __syncthreads();
    W9P_T(t3);
#ifdef SNWS_TIMING
    if (grp == 0) { dbg[0] += t2 - t0; dbg[1] += t3 - t2; } else dbg[4] += t3 - t2;
#endif
    // ---- phase 2k + 1: group B multiplies tile k; group A stages its tile k + 1
    if (grp == 1) {
      if (k < itB.n) multiply(xbuf + (shared ? (k & 1) * XB : XB));
    } else if (!W9P_NOSTAGE) {
      if (k + 1 < itA.n) {
        if (shared) w9p_stage_x<JXH, JX>(a, xbuf + ((k + 1) & 1) * XB, lw, lane, itA.t_begin + k + 1, itA.cit);
        else w9p_stage_x<0, JX>(a, xbuf, lw, lane, itA.t_begin + k + 1, itA.cit);
        w9p_stage_dy(a, dy_lds, lw, lane, itA.t_begin + k + 1, itA.cot);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
#ifdef SNWS_TIMING
    dbg[5] += clock64() - t3;
#endif
  }
#ifdef SNWS_TIMING
  if (tid == 0) {      // group A, wave 0: cycles multiplying, at the barrier behind it, whole second phase; tiles
    atomicAdd(&snws_dbg[8], (unsigned long long)dbg[0]);
    atomicAdd(&snws_dbg[9], (unsigned long long)dbg[1]);
    atomicAdd(&snws_dbg[14], (unsigned long long)dbg[5]);
    atomicAdd(&snws_dbg[13], (unsigned long long)nloop);
  }
  if (tid == 256) {    // group B, wave 0: cycles issuing the copies, waiting for them, at the barrier
    atomicAdd(&snws_dbg[10], (unsigned long long)dbg[2]);
    atomicAdd(&snws_dbg[11], (unsigned long long)dbg[3]);
    atomicAdd(&snws_dbg[12], (unsigned long long)dbg[4]);
  }
#endif

  // ---- flush: the partial-tile layout [tap][ob][in block][j][in-channel half][lane] of conv_wgrad9q_kernel
  if (me.n > 0) {
    const int fl_half = lw >> 1, fl_cb0 = lw & 1;
    if (a.part) {
      float* o = a.part + ((int64_t)me.split * npairs + me.pair) * 36864 + fl_half * 64 + lane;
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int j = 0; j < 4; ++j) o[(((i * 4 + ob) * 2 + fl_cb0) * 4 + j) * 128] = acc16[i][ob][j];
    } else {
#pragma unroll
      for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int ob = 0; ob < 4; ++ob)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int co = me.cot * 64 + ob * 16 + 4 * (lane >> 4) + j;
            const int ci16 = me.cit * 64 + fl_half * 32 + fl_cb0 * 16 + (lane & 15);
            if (ci16 < a.Cin && co < a.Cout) atomicAdd(a.dw + ((int64_t)co * 9 + i) * a.Cin + ci16, acc16[i][ob][j]);
          }
    }
  }
}

// ------------------------------------------------------------------------------------------------- host
int check_mfma_desc(const snws_conv_desc* d, const char* fn) {
  SNWS_CHECK_ARG(d, "%s: desc is NULL", fn);
  SNWS_CHECK_ARG(d->N >= 1 && d->H >= 1 && d->W >= 1 && d->C_in >= 1 && d->C_out >= 1, "%s: non-positive size", fn);
  const bool k3 = d->kh == 3 && d->kw == 3 && d->pad == 1, k1 = d->kh == 1 && d->kw == 1 && d->pad == 0;
  if (!(k3 || k1) || d->stride != 1 || d->dil != 1 || d->groups != 1)
    SNWS_UNSUPPORTED("%s: only 3x3/pad1 or 1x1/pad0, stride 1, dilation 1, groups 1 (got k=%dx%d s=%d p=%d d=%d g=%d)", fn,
                     d->kh, d->kw, d->stride, d->pad, d->dil, d->groups);
  if (d->C_in % 16 || d->C_out % 8) SNWS_UNSUPPORTED("%s: C_in %d must be a multiple of 16 and C_out %d of 8", fn, d->C_in, d->C_out);
  if (d->ldx % 8 || d->ldy % 8) SNWS_UNSUPPORTED("%s: pixel strides must be multiples of 8 elements", fn);
  if (d->C_in > (ZERO_PAGE_BYTES - 64) / 2 || d->C_out > (ZERO_PAGE_BYTES - 64) / 2)
    SNWS_UNSUPPORTED("%s: more than %d channels", fn, (ZERO_PAGE_BYTES - 64) / 2);
  SNWS_CHECK_ARG(d->ldx >= d->C_in && d->ldy >= d->C_out, "%s: pixel stride smaller than channel count", fn);
  if ((int64_t)d->H * d->W * d->ldx * 2 >= 0xFFFFFFFFll || (int64_t)d->H * d->W * d->ldy * 2 >= 0xFFFFFFFFll)
    SNWS_UNSUPPORTED("%s: one image exceeds 4 GiB", fn);
  return SNWS_OK;
}

// SNWS_PRIO = 0 switches the progressive wave priority of the matrix phases off (A/B measurements)
int conv_prio() {
  static const int on = getenv("SNWS_PRIO") ? atoi(getenv("SNWS_PRIO")) != 0 : 1;
  return on;
}

// SNWS_CONV_DB=0/1 overrides the default choice between the single- and double-buffered forward kernels
int conv_db_mode() {
  static int mode = -2;
  if (mode == -2) {
    const char* e = getenv("SNWS_CONV_DB");
    mode = e ? atoi(e) : -1;
  }
  return mode;
}

template <int KC, int TAPS, bool DB, bool M16, bool BWR, int PRIO = 0>
int launch_fwd_v(const std::conditional_t<BWR, FwdArgsBwr, FwdArgs>& a, unsigned grid, hipStream_t st) {
  using G = FwdGeom<KC, TAPS>;
  constexpr int LDS = DB ? G::LDS_BYTES_DB : G::LDS_BYTES;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_fwd_kernel<KC, TAPS, DB, M16, BWR, PRIO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_fwd_kernel<KC, TAPS, DB, M16, BWR, PRIO><<<grid, 256, LDS, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_fwd");
  return SNWS_OK;
}

// SNWS_CONV_DB = 0/1 forces the single- / double-buffered kernel, SNWS_CONV_M16 = 0/1 the MFMA shape; unset = the
// measured best per shape
template <int KC, int TAPS, bool BWR>
int launch_fwd(const std::conditional_t<BWR, FwdArgsBwr, FwdArgs>& a, unsigned grid, hipStream_t st) {
  const int mode = conv_db_mode();
  const bool db = (mode >= 0 ? mode == 1 : TAPS == 1) && a.nchunks >= 2;   // pays for 1x1 (short chunks), loses for 3x3
  if constexpr (KC == 32) {
    static const int m16_env = getenv("SNWS_CONV_M16") ? atoi(getenv("SNWS_CONV_M16")) : -1;
    const bool m16 = m16_env >= 0 ? m16_env != 0 : true;   // +13% over 32x32x16 at equal cycles (higher sustained clock)
    if (db) return m16 ? launch_fwd_v<32, TAPS, true, true, BWR>(a, grid, st) : launch_fwd_v<32, TAPS, true, false, BWR>(a, grid, st);
    if (m16 && TAPS == 9 && a.prio == 1) return launch_fwd_v<32, TAPS, false, true, BWR, (TAPS == 9 ? 1 : 0)>(a, grid, st);
    return m16 ? launch_fwd_v<32, TAPS, false, true, BWR>(a, grid, st) : launch_fwd_v<32, TAPS, false, false, BWR>(a, grid, st);
  } else {
    return launch_fwd_v<KC, TAPS, false, false, BWR>(a, grid, st);
  }
}

template <bool BWR>
int launch_fwd_any(const std::conditional_t<BWR, FwdArgsBwr, FwdArgs>& a, int kc, int taps, unsigned grid, hipStream_t st) {
  if (kc == 32) return taps == 9 ? launch_fwd<32, 9, BWR>(a, grid, st) : launch_fwd<32, 1, BWR>(a, grid, st);
  return taps == 9 ? launch_fwd<16, 9, BWR>(a, grid, st) : launch_fwd<16, 1, BWR>(a, grid, st);
}

// runs y(C_out_run channels) = conv(x(C_in_run channels), pack)
int run_fwd_like(const char* fn, int N, int H, int W, int c_in_run, int c_out_run, int taps, const void* x, int64_t ldx,
                 const void* pack, const float* bias, void* y, int64_t ldy, float* stat, hipStream_t st,
                 const snws_sn_bwd_seg* segs = nullptr, int n_segs = 0, int accumulate = 0) {
  SNWS_CHECK_ARG(x && pack && y, "%s: NULL pointer", fn);
  SNWS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)pack & 15) == 0,
                 "%s: pointers must be 16-byte aligned", fn);
  const PackGeom g = pack_geom(c_in_run, c_out_run, taps);
  FwdArgsBwr a;
  a.x = (const bf16_t*)x; a.pack = (const bf16_t*)pack; a.bias = bias; a.y = (bf16_t*)y; a.stat = stat;
  a.N = N; a.H = H; a.W = W; a.Cout = c_out_run; a.ldx = ldx; a.ldy = ldy;
  a.tilesX = cdiv(W, FWD_TW); a.tilesY = cdiv(H, FWD_TH); a.coTiles = g.c_out_p / 64; a.nchunks = g.nchunks;
  a.accumulate = accumulate;
  a.prio = conv_prio();
  bool bwr = false;
  if (n_segs > 0) {
    if (a.coTiles > MAX_BWR_TILES) SNWS_UNSUPPORTED("%s: more than %d output-channel tiles", fn, MAX_BWR_TILES * 64);
    for (int t = 0; t < MAX_BWR_TILES; ++t) a.bwr[t] = SnBwdTile{};
    for (int i = 0; i < n_segs; ++i) {
      const snws_sn_bwd_seg& sg = segs[i];
      SNWS_CHECK_ARG(sg.x && sg.ctx && sg.beta && sg.part, "%s: NULL pointer in segment %d", fn, i);
      SNWS_CHECK_ARG(sg.c_begin >= 0 && sg.c_count > 0 && sg.c_begin % 64 == 0 && sg.c_count % 8 == 0 &&
                         sg.c_begin + sg.c_count <= c_out_run,
                     "%s: segment %d covers channels [%d, %d): begin must be a multiple of 64, count of 8, inside %d",
                     fn, i, sg.c_begin, sg.c_begin + sg.c_count, c_out_run);
      SNWS_CHECK_ARG(((uintptr_t)sg.x & 15) == 0 && sg.ldx % 8 == 0 && sg.ldx >= sg.c_count,
                     "%s: segment %d input must be 16-byte aligned with a pixel stride that is a multiple of 8", fn, i);
      const int64_t nc = (int64_t)N * sg.c_count;     // ctx layout: [mean_in][var_in][mean][rstd][scale], N*C each
      for (int c = 0; c < sg.c_count; c += 64) {
        SnBwdTile& tl = a.bwr[(sg.c_begin + c) / 64];
        SNWS_CHECK_ARG(tl.part == nullptr, "%s: segments overlap at channel %d", fn, sg.c_begin + c);
        tl.x = (const bf16_t*)sg.x + c;
        tl.mean = sg.ctx + 2 * nc + c;
        tl.rstd = sg.ctx + 3 * nc + c;
        tl.scale = sg.ctx + 4 * nc + c;
        tl.beta = sg.beta + c;
        tl.part = sg.part + 2 * c;
        tl.ldx = (int)sg.ldx;
        tl.cseg = sg.c_count;
        tl.cvalid = sg.c_count - c < 64 ? sg.c_count - c : 64;
        tl.relu = sg.relu;
      }
    }
    bwr = true;
  }
  const int64_t nblk = (int64_t)N * a.tilesX * a.tilesY * a.coTiles;
  SNWS_CHECK_ARG(nblk < (1ll << 31), "%s: grid too large", fn);
  if (bwr) return launch_fwd_any<true>(a, g.KC, taps, (unsigned)nblk, st);
  return launch_fwd_any<false>(static_cast<const FwdArgs&>(a), g.KC, taps, (unsigned)nblk, st);
}

template <int CIC, int TAPS>
int launch_wgrad(const WgradArgs& a, unsigned grid, hipStream_t st) {
  using G = WgGeom<CIC, TAPS>;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_kernel<CIC, TAPS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_wgrad_kernel<CIC, TAPS><<<grid, 256, G::LDS_BYTES, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_wgrad");
  return SNWS_OK;
}

template <bool BIAS>
int launch_wgrad1(const WgradArgs& a, unsigned grid, hipStream_t st) {
  constexpr int LDS = WgGeom<64, 1>::LDS_BYTES > 65536 + 1024 ? WgGeom<64, 1>::LDS_BYTES : 65536 + 1024;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1_kernel<BIAS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_wgrad1_kernel<BIAS><<<grid, 256, LDS, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_wgrad");
  return SNWS_OK;
}

template <bool BIAS>
int launch_wgrad1w(const Wgrad1wArgs& a, unsigned grid, hipStream_t st) {
  constexpr int LDS = 65536 + 4 * 128 * 4;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1w_kernel<BIAS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_wgrad1w_kernel<BIAS><<<grid, 256, LDS, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_wgrad");
  return SNWS_OK;
}

template <bool BIAS>
int launch_wgrad1t(const Wgrad1wArgs& a, unsigned grid, hipStream_t st) {
  using G = Wgrad1tGeom;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad1t_kernel<BIAS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_wgrad1t_kernel<BIAS><<<grid, 256, G::LDS_BYTES, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_wgrad");
  return SNWS_OK;
}

template <bool BIAS, bool PRIO>
int launch_wgrad9q(const WgradArgs& a, unsigned grid, hipStream_t st) {
  using G = WgGeom<64, 9>;
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad9q_kernel<BIAS, PRIO>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_wgrad9q_kernel<BIAS, PRIO><<<grid, 256, G::LDS_BYTES, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_wgrad");
  return SNWS_OK;
}

// SNWS_WGRAD9_PP = 0 selects conv_wgrad9q_kernel (two independent workgroups per CU) instead of the ping-pong kernel
bool wgrad9_pp() {
  static const bool on = !(getenv("SNWS_WGRAD9_PP") && atoi(getenv("SNWS_WGRAD9_PP")) == 0);
  return on;
}

int wgrad9_pp_min_pairs() {
  static const int n = getenv("SNWS_WGRAD9_PP_MIN_PAIRS") ? atoi(getenv("SNWS_WGRAD9_PP_MIN_PAIRS")) : 2;
  return n;
}

int launch_wgrad9p(const WgradArgs& a, unsigned items, hipStream_t st) {
  using G = WgGeom<64, 9>;
  constexpr int LDS = 2 * (((G::X_BYTES + 15) / 16) * 16) + 2 * G::DY_BYTES;
  static_assert(LDS <= 160 * 1024, "the ping-pong wgrad needs both x buffers and both dy tiles in one CU's LDS");
  static PerDeviceOnce once;
  std::atomic<bool>* const attr_set = once.flag();
  if (!attr_set || !attr_set->load(std::memory_order_acquire)) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad9p_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) {
      snws_set_error("conv_mfma: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SNWS_ERR_HIP;
    }
    if (attr_set) attr_set->store(true, std::memory_order_release);
  }
  conv_wgrad9p_kernel<<<(items + 1) / 2, 512, LDS, st>>>(a);
  SNWS_LAUNCH_CHECK("snws_conv_mfma_wgrad");
  return SNWS_OK;
}

}  // namespace

#ifdef SNWS_TIMING
extern "C" int snws_dbg_read(unsigned long long* out8) {
  hipDeviceSynchronize();
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(snws_dbg), sizeof(snws_dbg)) != hipSuccess) return -1;
  unsigned long long z[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return hipMemcpyToSymbol(HIP_SYMBOL(snws_dbg), z, sizeof(z)) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int snws_conv_mfma_tiles(int H, int W) { return cdiv(H, FWD_TH) * cdiv(W, FWD_TW); }

extern "C" int snws_conv_mfma_fwd(const snws_conv_desc* d, const void* x, const void* pack_fwd, const float* bias,
                                  void* y, float* stat_part, void* stream) {
  if (int rc = check_mfma_desc(d, "snws_conv_mfma_fwd")) return rc;
  return run_fwd_like("snws_conv_mfma_fwd", d->N, d->H, d->W, d->C_in, d->C_out, d->kh * d->kw, x, d->ldx, pack_fwd, bias,
                      y, d->ldy, stat_part, (hipStream_t)stream);
}

extern "C" int snws_conv_mfma_dgrad(const snws_conv_desc* d, const void* dy, const void* pack_dgrad, void* dx,
                                    void* stream) {
  if (int rc = check_mfma_desc(d, "snws_conv_mfma_dgrad")) return rc;
  if (d->C_out % 16 || d->C_in % 8) SNWS_UNSUPPORTED("snws_conv_mfma_dgrad: C_out %d must be a multiple of 16 and C_in %d of 8", d->C_out, d->C_in);
  return run_fwd_like("snws_conv_mfma_dgrad", d->N, d->H, d->W, d->C_out, d->C_in, d->kh * d->kw, dy, d->ldy, pack_dgrad,
                      nullptr, dx, d->ldx, nullptr, (hipStream_t)stream);
}

extern "C" int snws_conv_mfma_dgrad_acc(const snws_conv_desc* d, const void* dy, const void* pack_dgrad, void* dx,
                                        void* stream) {
  if (int rc = check_mfma_desc(d, "snws_conv_mfma_dgrad_acc")) return rc;
  if (d->C_out % 16 || d->C_in % 8) SNWS_UNSUPPORTED("snws_conv_mfma_dgrad_acc: C_out %d must be a multiple of 16 and C_in %d of 8", d->C_out, d->C_in);
  return run_fwd_like("snws_conv_mfma_dgrad_acc", d->N, d->H, d->W, d->C_out, d->C_in, d->kh * d->kw, dy, d->ldy, pack_dgrad,
                      nullptr, dx, d->ldx, nullptr, (hipStream_t)stream, nullptr, 0, 1);
}

extern "C" int snws_conv_mfma_dgrad_sn(const snws_conv_desc* d, const void* dy, const void* pack_dgrad, void* dx,
                                       const snws_sn_bwd_seg* segs, int n_segs, void* stream) {
  if (int rc = check_mfma_desc(d, "snws_conv_mfma_dgrad_sn")) return rc;
  if (d->C_out % 16 || d->C_in % 8) SNWS_UNSUPPORTED("snws_conv_mfma_dgrad_sn: C_out %d must be a multiple of 16 and C_in %d of 8", d->C_out, d->C_in);
  SNWS_CHECK_ARG(n_segs >= 0 && (n_segs == 0 || segs), "snws_conv_mfma_dgrad_sn: bad segment list");
  return run_fwd_like("snws_conv_mfma_dgrad_sn", d->N, d->H, d->W, d->C_out, d->C_in, d->kh * d->kw, dy, d->ldy, pack_dgrad,
                      nullptr, dx, d->ldx, nullptr, (hipStream_t)stream, segs, n_segs);
}

namespace {

// how a wgrad launch is cut: which kernel, how many (out, in) tile pairs and pixel splits
struct WgradPlan {
  int kind;          // as WgradReduceArgs::kind
  int npairs, coTiles, ciTiles, splits, tilesPerSplit, ntiles;
  int E, TC;
  bool wide;
};

WgradPlan wgrad_plan(const snws_conv_desc* d, bool det, bool with_bias) {
  WgradPlan p;
  const bool c16 = d->C_in == 16, c64 = !c16;
  const int taps = d->kh * d->kw;
  const int tilesX = cdiv(d->W, WG_TW), tilesY = cdiv(d->H, WG_TH);
  p.wide = false;
  p.coTiles = cdiv(d->C_out, 64);
  p.ciTiles = c64 ? cdiv(d->C_in, 64) : 1;
  p.ntiles = d->N * tilesX * tilesY;
  p.TC = 64;
  static const bool generic = getenv("SNWS_WGRAD_GENERIC") != nullptr;
  static const bool narrow1 = getenv("SNWS_WGRAD1_NARROW") != nullptr;    // A/B switch: the 64-wide 1x1 kernel
  static const int forced_blocks = getenv("SNWS_WGRAD_BLOCKS") ? atoi(getenv("SNWS_WGRAD_BLOCKS")) : 0;
  if (c64 && taps == 1 && !generic && !narrow1 && d->C_in > 64 && d->C_out > 64) {
    // the wide kernel pays when its 128-wide tiles are mostly full (measured: 512->256 @512^2 2x the narrow kernel;
    // 320->576, whose tiles are 25 % padding, 0.85x)
    const int64_t padded = (int64_t)cdiv(d->C_in, 128) * 128 * cdiv(d->C_out, 128) * 128;
    p.wide = padded * 5 <= (int64_t)d->C_in * d->C_out * 6;
  }
  static const bool no_t192 = getenv("SNWS_WGRAD1_T192") && atoi(getenv("SNWS_WGRAD1_T192")) == 0;   // A/B switch
  static const bool force_t192 = getenv("SNWS_WGRAD1_T192") && atoi(getenv("SNWS_WGRAD1_T192")) == 2;    // tests: any size
  if (c64 && taps == 1 && !generic && !narrow1 && !no_t192 && d->C_out % 192 == 0 && d->C_in >= 160 &&
      (force_t192 || (int64_t)d->N * d->H * d->W >= 131072)) {
    // the 192 x 160 tile: no padding for 576 outputs.  Measured on 320 -> 576, batch 16: 0.92 -> 0.49 ms at 256^2 (420 ->
    // 794 TFLOP/s), 0.23 -> 0.16 ms at 128^2; below ~128 K pixels its 123 KB flush per workgroup outweighs the gain
    // (0.054 -> 0.069 ms at 64^2) and the 64-wide kernel stays.
    p.wide = true;
    p.kind = 6; p.E = Wgrad1tGeom::E; p.TC = 192;
    p.coTiles = d->C_out / 192; p.ciTiles = cdiv(d->C_in, 160);
    p.npairs = p.coTiles * p.ciTiles;
    const int64_t nt = ((int64_t)d->N * d->H * d->W + Wgrad1tGeom::TP - 1) / Wgrad1tGeom::TP;
    p.ntiles = nt < (1ll << 31) ? (int)nt : -1;
    int sp = (forced_blocks > 0 ? forced_blocks : 512) / p.npairs;   // one resident round of long workgroups
    if (sp < 1) sp = 1;
    if (sp > p.ntiles) sp = p.ntiles;
    p.tilesPerSplit = cdiv(p.ntiles, sp);
    p.splits = cdiv(p.ntiles, p.tilesPerSplit);
    return p;
  }
  if (p.wide) {
    p.kind = 3; p.E = 16384; p.TC = 128;
    p.coTiles = cdiv(d->C_out, 128); p.ciTiles = cdiv(d->C_in, 128);
    p.npairs = p.coTiles * p.ciTiles;
    const int64_t nt = ((int64_t)d->N * d->H * d->W + 127) / 128;
    p.ntiles = nt < (1ll << 31) ? (int)nt : -1;
    int sp = (forced_blocks > 0 ? forced_blocks : 512) / p.npairs;   // one resident round of long workgroups
    if (sp < 1) sp = 1;
    if (sp > p.ntiles) sp = p.ntiles;
    p.tilesPerSplit = cdiv(p.ntiles, sp);
    p.splits = cdiv(p.ntiles, p.tilesPerSplit);
    return p;
  }
  p.npairs = p.coTiles * p.ciTiles;
  if (c64 && taps == 9 && !generic) { p.kind = 5; p.E = 36864; }
  else if (c64 && taps == 1 && !generic) { p.kind = 2; p.E = 4096; }
  else { p.kind = c64 ? 0 : 1; p.E = taps * 4096; }
  // Atomic flush: every workgroup ends with a 64x64x9 float-atomic flush (147 KB; the chip retires ~1.3 TB/s of
  // atomics), so a launch of few channel-tile pairs wants few, long workgroups: one resident round (256 CUs x 2).
  // (Measured: 64->64 @512^2 650 -> 813 TFLOP/s, 128->128 @256^2 703 -> 924 with one round instead of 2048
  // workgroups.)  With many pairs the workgroups of one pixel slice share its tiles through L2 only while they walk
  // in step, and they drift apart over a long sweep: ~32 pixel tiles per workgroup is the measured optimum then
  // (320->320: @512^2 981 TFLOP/s with 512 workgroups, 1045 with 2048, 1158 with 8192; @256^2 1077 / 1116 / 1078
  // with 2048 / 3072 / 8192).  The partial-tile flush (det) keeps the same cut: its stores are ~5x cheaper than the
  // atomics, but every split is read back once by the reducer.
  // The ping-pong 3x3 kernel (no bias sums, >= wgrad9_pp_min_pairs() channel-tile pairs) shares the x tile inside its
  // workgroup and keeps the sibling pairs of a pixel slice in step by construction, so it wants the other extreme: ONE
  // resident round of 256 workgroups x 2 items, each a long sweep - 320->320 @512^2 5.43 -> 5.10 ms, @256^2 1.46 -> 1.27
  // ms against the ~32-tile items; 1024 items equal, 768 (one and a half rounds) 25 % slower, 256 (half the chip idle
  // in a workgroup's second half) 50 % slower - and its partial tiles shrink from 1.2 GB to 74 MB on that layer.
  (void)det;
  const bool pingpong = p.kind == 5 && wgrad9_pp() && !with_bias && p.npairs >= wgrad9_pp_min_pairs();
  const int64_t sweep = (int64_t)p.ntiles * p.npairs;
  const int target_blocks = forced_blocks > 0 ? forced_blocks
                            : pingpong ? 512
                            : sweep >= 32ll * 2048 ? (int)(sweep / 32 < 8192 ? sweep / 32 : 8192) : 512;
  int splits = target_blocks / p.npairs;   // never more than target_blocks workgroups: two extra ones cost a whole extra round
  if (splits < 1) splits = 1;
  if (splits > p.ntiles) splits = p.ntiles;
  p.tilesPerSplit = cdiv(p.ntiles, splits);
  p.splits = cdiv(p.ntiles, p.tilesPerSplit);
  return p;
}

int64_t wgrad_ws_floats(const WgradPlan& p) {
  return (int64_t)p.splits * p.npairs * p.E + (int64_t)p.splits * p.coTiles * p.TC;
}

int run_wgrad(const char* fn, const snws_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* dbias,
              float* ws, int64_t ws_bytes, hipStream_t st) {
  if (int rc = check_mfma_desc(d, fn)) return rc;
  SNWS_CHECK_ARG(x && dy && dw_ohwi, "%s: NULL pointer", fn);
  SNWS_CHECK_ARG(((uintptr_t)x & 15) == 0 && ((uintptr_t)dy & 15) == 0, "%s: pointers must be 16-byte aligned", fn);
  const bool det = ws != nullptr;
  const WgradPlan p = wgrad_plan(d, det, dbias != nullptr);
  SNWS_CHECK_ARG(p.ntiles > 0, "%s: too many pixels", fn);
  float* part = nullptr;
  float* pbias = nullptr;
  if (det) {
    SNWS_CHECK_ARG(((uintptr_t)ws & 15) == 0, "%s: workspace must be 16-byte aligned", fn);
    SNWS_CHECK_ARG(ws_bytes >= wgrad_ws_floats(p) * 4, "%s: workspace of %lld bytes, %lld needed", fn,
                   (long long)ws_bytes, (long long)wgrad_ws_floats(p) * 4);
    part = ws;
    pbias = ws + (int64_t)p.splits * p.npairs * p.E;
  }
  const int taps = d->kh * d->kw;
  const unsigned grid = (unsigned)(p.npairs * p.splits);
  int rc;
  if (p.wide) {
    Wgrad1wArgs w;
    w.x = (const bf16_t*)x; w.dy = (const bf16_t*)dy; w.dw = dw_ohwi; w.dbias = dbias; w.part = part; w.pbias = pbias;
    w.P = (int64_t)d->N * d->H * d->W; w.Cin = d->C_in; w.Cout = d->C_out; w.ldx = d->ldx; w.lddy = d->ldy;
    w.coTiles = p.coTiles; w.ciTiles = p.ciTiles; w.ntiles = p.ntiles; w.tilesPerSplit = p.tilesPerSplit;
    if (p.kind == 6) rc = dbias ? launch_wgrad1t<true>(w, grid, st) : launch_wgrad1t<false>(w, grid, st);
    else rc = dbias ? launch_wgrad1w<true>(w, grid, st) : launch_wgrad1w<false>(w, grid, st);
  } else {
    WgradArgs a;
    a.x = (const bf16_t*)x; a.dy = (const bf16_t*)dy; a.dw = dw_ohwi; a.dbias = dbias; a.part = part; a.pbias = pbias;
    a.N = d->N; a.H = d->H; a.W = d->W; a.Cin = d->C_in; a.Cout = d->C_out; a.ldx = d->ldx; a.lddy = d->ldy;
    a.tilesX = cdiv(d->W, WG_TW); a.tilesY = cdiv(d->H, WG_TH);
    a.coTiles = p.coTiles; a.ciTiles = p.ciTiles; a.tilesPerSplit = p.tilesPerSplit;
    a.prio = conv_prio();
    if (p.kind == 5) {
      // the ping-pong kernel from 4 channel-tile pairs up (measured on one box against conv_wgrad9q_kernel: +1 % on
      // 320->320 @512^2 at 0.57x its HBM fetch, +2..11 % on 256->256 @128^2 ... 1024->1024 @32^2, -3..-7 % on the one-
      // and two-pair launches, whose two groups cannot share an x tile); a launch that wants the bias gradient (a
      // Conv2dws outside a normalised block) takes conv_wgrad9q_kernel, which sums the staged gradient tile's columns
      if (wgrad9_pp() && !dbias && p.npairs >= wgrad9_pp_min_pairs()) rc = launch_wgrad9p(a, grid, st);
      else if (a.prio) rc = dbias ? launch_wgrad9q<true, true>(a, grid, st) : launch_wgrad9q<false, true>(a, grid, st);
      else rc = dbias ? launch_wgrad9q<true, false>(a, grid, st) : launch_wgrad9q<false, false>(a, grid, st);
    }
    else if (p.kind == 2) rc = dbias ? launch_wgrad1<true>(a, grid, st) : launch_wgrad1<false>(a, grid, st);
    else if (p.kind == 0) rc = taps == 9 ? launch_wgrad<64, 9>(a, grid, st) : launch_wgrad<64, 1>(a, grid, st);
    else rc = taps == 9 ? launch_wgrad<16, 9>(a, grid, st) : launch_wgrad<16, 1>(a, grid, st);
  }
  if (rc || !det) return rc;
  WgradReduceArgs r;
  r.part = part; r.pbias = pbias; r.dw = dw_ohwi; r.dbias = dbias;
  r.kind = p.kind; r.taps = taps; r.E = p.E; r.TC = p.TC;
  r.npairs = p.npairs; r.splits = p.splits; r.coTiles = p.coTiles; r.Cin = d->C_in; r.Cout = d->C_out;
  r.main_blocks = (int)((int64_t)p.npairs * p.E / 256);
  const int bias_blocks = dbias ? p.coTiles * (p.TC / 64) : 0;
  wgrad_reduce_kernel<<<r.main_blocks + bias_blocks, 256, 0, st>>>(r);
  SNWS_LAUNCH_CHECK(fn);
  return SNWS_OK;
}

}  // namespace

extern "C" int snws_conv_mfma_wgrad(const snws_conv_desc* d, const void* x, const void* dy, float* dw_ohwi, float* dbias,
                                    void* stream) {
  return run_wgrad("snws_conv_mfma_wgrad", d, x, dy, dw_ohwi, dbias, nullptr, 0, (hipStream_t)stream);
}

extern "C" int64_t snws_conv_mfma_wgrad_workspace_bytes(const snws_conv_desc* d) {
  if (!d || check_mfma_desc(d, "snws_conv_mfma_wgrad_workspace_bytes")) return -1;
  // the cut of the work depends on whether the bias gradient is asked for: enough for either call
  const WgradPlan p = wgrad_plan(d, true, false), pb = wgrad_plan(d, true, true);
  if (p.ntiles <= 0 || pb.ntiles <= 0) return -1;
  const int64_t f = wgrad_ws_floats(p), fb = wgrad_ws_floats(pb);
  return (f > fb ? f : fb) * 4;
}

extern "C" int snws_conv_mfma_wgrad_det(const snws_conv_desc* d, const void* x, const void* dy, float* dw_ohwi,
                                        float* dbias, void* workspace, int64_t workspace_bytes, void* stream) {
  SNWS_CHECK_ARG(workspace, "snws_conv_mfma_wgrad_det: NULL workspace");
  return run_wgrad("snws_conv_mfma_wgrad_det", d, x, dy, dw_ohwi, dbias, (float*)workspace, workspace_bytes,
                   (hipStream_t)stream);
}
