// Two-stage, deterministic, segmented column sums:  out[s][n] = sum_m x[s][m][n].
// Stage 1: grid (nparts, S); a 256-thread block covers N/4 float4 columns x R row lanes
//          (coalesced float4 reads along n), LDS-combines the row lanes -> part[s][p][n].
// Stage 2: grid (ceil(N/16), S); 16 columns x 16 part lanes per block, LDS tree -> out.
// Used for bias / time-embedding gradients and GroupNorm dgamma / dbeta.
#pragma once
#include "gad_common.h"

namespace gad_reduce {
namespace {   // internal linkage: the header is included by several translation units

constexpr int NT = 256;

__global__ __launch_bounds__(NT) void colsum_part_vec(const float* __restrict__ x, float* __restrict__ part, long M, int N,
                                                      int rows_per, int nparts) {
  __shared__ float red[NT * 4];
  const int seg = blockIdx.y, C4 = N >> 2;
  long r0 = (long)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
  if (C4 > NT) {                                          // wide rows (LayerNorm / GEGLU widths of the SD blocks): one
    const f32x4* b = reinterpret_cast<const f32x4*>(x + (long)seg * M * N);   // row lane, threads stride the column quads
    for (int cq = threadIdx.x; cq < C4; cq += NT) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      for (long r = r0; r < r1; ++r) s += b[r * C4 + cq];
      *reinterpret_cast<f32x4*>(part + ((long)seg * nparts + blockIdx.x) * N + cq * 4) = s;
    }
    return;
  }
  const int lanes = NT / C4 > 0 ? NT / C4 : 1;           // row lanes
  const int tid = threadIdx.x, lane = tid / C4, cq = tid - lane * C4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (lane < lanes) {
    const f32x4* b = reinterpret_cast<const f32x4*>(x + (long)seg * M * N) + cq;
    for (long r = r0 + lane; r < r1; r += lanes) s += b[r * C4];
    *reinterpret_cast<f32x4*>(red + (lane * C4 + cq) * 4) = s;
  }
  __syncthreads();
  for (int n = tid; n < N; n += NT) {
    float t = 0.f;
    for (int l = 0; l < lanes; ++l) t += red[l * N + n];
    part[((long)seg * nparts + blockIdx.x) * N + n] = t;
  }
}

__global__ __launch_bounds__(NT) void colsum_part_scalar(const float* __restrict__ x, float* __restrict__ part, long M, int N,
                                                         int rows_per, int nparts) {
  const int seg = blockIdx.y;
  long r0 = (long)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
  const float* b = x + (long)seg * M * N;
  for (int n = threadIdx.x; n < N; n += NT) {
    float t = 0.f;
    for (long r = r0; r < r1; ++r) t += b[r * N + n];
    part[((long)seg * nparts + blockIdx.x) * N + n] = t;
  }
}

// out0/out1: plain (out1 == nullptr): out0[s][n].  De-interleaving (out1 != nullptr, S == 1):
// even columns -> out0[n/2], odd columns -> out1[n/2]   (GroupNorm: (dbeta, dgamma) pairs)
// Block = 16 columns x 16 part lanes: the partials are few hundred rows of a small matrix, so the kernel is a
// latency chain - short per-thread loops and many blocks matter, not bandwidth.  Fixed summation order.
constexpr int FC = 16, FL = NT / FC;
__global__ __launch_bounds__(NT) void colsum_final(const float* __restrict__ part, float* __restrict__ out0,
                                                   float* __restrict__ out1, int nparts, int N) {
  __shared__ float red[NT];
  const int seg = blockIdx.y, col = threadIdx.x % FC, lane = threadIdx.x / FC;
  const int n = blockIdx.x * FC + col;
  float t0 = 0.f, t1 = 0.f;
  if (n < N) {
    const float* p = part + (long)seg * nparts * N + n;
    int k = lane;
    for (; k + FL < nparts; k += 2 * FL) {     // two independent chains per thread
      t0 += p[(long)k * N];
      t1 += p[(long)(k + FL) * N];
    }
    if (k < nparts) t0 += p[(long)k * N];
  }
  red[threadIdx.x] = t0 + t1;
  __syncthreads();
#pragma unroll
  for (int st = FL / 2; st >= 1; st >>= 1) {
    if (lane < st) red[threadIdx.x] += red[threadIdx.x + st * FC];
    __syncthreads();
  }
  if (lane == 0 && n < N) {
    float v = red[col];
    if (out1 == nullptr) out0[(long)seg * N + n] = v;
    else if (n & 1) out1[n >> 1] = v;
    else out0[n >> 1] = v;
  }
}

struct Plan {
  int rows_per, nparts;
  bool vec;
};
static inline Plan plan(int S, long M, int N) {
  Plan p;
  long want = 512 / (S > 0 ? S : 1);
  if (want < 1) want = 1;
  long rp = gad_ceil_div(M, want);
  if (rp < 32) rp = 32;
  p.rows_per = (int)rp;
  p.nparts = (int)gad_ceil_div(M, rp);
  p.vec = (N % 4 == 0);
  return p;
}
static inline int64_t ws_bytes(int S, long M, int N) { return (int64_t)S * plan(S, M, N).nparts * N * 4; }

// returns 0 on success (launch errors are checked by the caller)
static inline void launch(const float* x, float* out0, float* out1, int S, long M, int N, float* ws, hipStream_t st) {
  if (M <= 512) {      // few rows (per-image partials, [B, C] time-embedding gradients): the final stage alone, one launch
    hipLaunchKernelGGL(colsum_final, dim3((N + FC - 1) / FC, S), dim3(NT), 0, st, x, out0, out1, (int)M, N);
    return;
  }
  Plan p = plan(S, M, N);
  if (p.vec && gad_aligned16(x))
    hipLaunchKernelGGL(colsum_part_vec, dim3(p.nparts, S), dim3(NT), 0, st, x, ws, M, N, p.rows_per, p.nparts);
  else
    hipLaunchKernelGGL(colsum_part_scalar, dim3(p.nparts, S), dim3(NT), 0, st, x, ws, M, N, p.rows_per, p.nparts);
  hipLaunchKernelGGL(colsum_final, dim3((N + FC - 1) / FC, S), dim3(NT), 0, st, (const float*)ws, out0, out1, p.nparts, N);
}

}  // namespace
}  // namespace gad_reduce
