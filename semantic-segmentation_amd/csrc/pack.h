// Layout of the bf16 filter image consumed by the MFMA convolution kernels.
//
// A convolution contracts over (tap, input channel) and produces `c_out` channels.  The kernel walks the
// contraction in chunks of KC input channels; inside a chunk it reads, for every tap and every 16-channel k-step,
// one A-operand fragment per lane: 8 consecutive channels of one output channel.  The image stores exactly those
// 16-byte fragments so that a (64-output-channel tile, chunk) slab is one contiguous block that is copied to LDS
// linearly, and so that the 16 lanes of a ds_read_b128 group land on 16 distinct 16-byte bank slots:
//
//   [out tile of 64][chunk][tap][k-step][half h = 0/1][out channel within tile (64)][8 channels]
//
// Channels beyond c_in (up to the next multiple of 16) and output channels beyond c_out (up to the next
// multiple of 64) are zero; the buffer is zero-filled once by its owner and the writers never touch the padding.
#pragma once
#include <stdint.h>

struct PackGeom {
  int c_in, c_out, taps;
  int c_in_p;   // c_in rounded up to 16
  int c_out_p;  // c_out rounded up to 64
  int KC;       // channels per chunk: 32 when c_in_p is a multiple of 32, else 16
  int KS;       // 16-channel k-steps per chunk
  int nchunks;
};

__host__ __device__ inline PackGeom pack_geom(int c_in, int c_out, int taps) {
  PackGeom g;
  g.c_in = c_in; g.c_out = c_out; g.taps = taps;
  g.c_in_p = (c_in + 15) / 16 * 16;
  g.c_out_p = (c_out + 63) / 64 * 64;
  g.KC = (g.c_in_p % 32 == 0) ? 32 : 16;
  g.KS = g.KC / 16;
  g.nchunks = g.c_in_p / g.KC;
  return g;
}

__host__ __device__ inline int64_t pack_elems(const PackGeom& g) { return (int64_t)g.c_out_p * g.c_in_p * g.taps; }

// element offset of filter value (output channel co, input channel ci, tap)
__host__ __device__ inline int64_t pack_index(const PackGeom& g, int co, int ci, int tap) {
  const int ct = co >> 6, col = co & 63;
  const int ch = ci / g.KC, cil = ci % g.KC;
  const int ks = cil >> 4, h = (cil >> 3) & 1, j = cil & 7;
  return ((((((int64_t)ct * g.nchunks + ch) * g.taps + tap) * g.KS + ks) * 2 + h) * 64 + col) * 8 + j;
}
