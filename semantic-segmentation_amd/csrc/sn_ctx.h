// Layout of the per-call SwitchNorm coefficient buffers (floats), shared by sn.hip and sn_head.hip.
#pragma once
#include <stdint.h>

namespace {

// ctx layout (floats)
struct CtxOff {
  int64_t mean_in, var_in, mean, rstd, scale, ln, bn, total;
};
__host__ __device__ inline CtxOff ctx_off(int N, int C) {
  CtxOff o;
  int64_t NC = (int64_t)N * C;
  o.mean_in = 0; o.var_in = NC; o.mean = 2 * NC; o.rstd = 3 * NC; o.scale = 4 * NC;
  o.ln = 5 * NC; o.bn = 5 * NC + 2 * N; o.total = 5 * NC + 2 * N + 2 * C;
  return o;
}
struct CtxbOff {
  int64_t P, Q, dm, dv, B, K, lnb, total;
};
__host__ __device__ inline CtxbOff ctxb_off(int N, int C) {
  CtxbOff o;
  int64_t NC = (int64_t)N * C;
  o.P = 0; o.Q = NC; o.dm = 2 * NC; o.dv = 3 * NC; o.B = 4 * NC; o.K = 5 * NC; o.lnb = 6 * NC; o.total = 6 * NC + 2 * N;
  return o;
}

}  // namespace
