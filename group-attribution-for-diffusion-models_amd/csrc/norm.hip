// GroupNorm (+SiLU) forward / backward for NHWC fp32 activations, gfx950.
//
// HBM-bound: algorithmic traffic 8 B/elem forward (read x, write y) and 16 B/elem
// backward (read x, dy, write dx + re-read).  Each image is cut into NCH pixel chunks so
// that the launch has >= ~1-2k workgroups (256 CUs x 8 XCDs need far more than 256):
//   pass 1 (stats):  per (image, chunk) partial moments, coalesced float4 reads along C,
//                    wavefront-free LDS combine; written to the caller's workspace;
//   pass 2 (apply):  every workgroup re-combines the (tiny) partials of its image with
//                    Chan's formula - no atomics, deterministic - then normalises its chunk.
// Forward has a one-pass plan (gn_slab_kernel, below) that reads x once when an (image, channel slab) fits
// in one workgroup's registers; the two-pass plan remains for the other shapes.
#include "gad_common.h"
#include "gad_reduce.h"
#include "gemm_dev.h"

namespace {

constexpr int NT = 256;
constexpr int MAXC = 3072;   // SD up-blocks concatenate to 2560 channels

struct Geo {
  int B, HW, C, G, cpg, C4, tpr, rows_par, nch, ppc;  // ppc = pixels per chunk
};

static Geo make_geo(const gad_groupnorm_args* a) {
  Geo g;
  g.B = a->B; g.HW = a->HW; g.C = a->C; g.G = a->G;
  g.cpg = a->C / a->G;
  g.C4 = a->C / 4;
  g.tpr = g.C4 < NT ? g.C4 : NT;       // threads per pixel row; wider rows loop over channel quads
  g.rows_par = NT / g.tpr;
  int nch = 2048 / (a->B > 0 ? a->B : 1);
  int maxch = a->HW / (8 * g.rows_par > 32 ? 8 * g.rows_par : 32);   // >= 8 pixels per thread row, >= 32 px per chunk
  if (nch > maxch) nch = maxch;
  if (nch < 1) nch = 1;
  g.ppc = (a->HW + nch - 1) / nch;
  g.nch = (a->HW + g.ppc - 1) / g.ppc;
  return g;
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + expf(-z)); }

// ---------------------------------------------------------------- forward ----
// Two sources (x2 != NULL): channels [0, C1) of the normalised tensor come from x ([B][HW][C1]) and [C1, C) from x2
// ([B][HW][C - C1]) - UpBlock2D's torch.cat([h, skip], 1) read in place; C1 % 4 == 0, so a channel quad has one source.
__device__ __forceinline__ const float* gn_src(const float* x, const float* x2, int C1, int C, int HW, int b, int c0, int* stride) {
  if (x2 == nullptr || c0 < C1) {
    const int w = x2 ? C1 : C;
    *stride = w;
    return x + ((long)b * HW) * w + c0;
  }
  *stride = C - C1;
  return x2 + ((long)b * HW) * (C - C1) + (c0 - C1);
}

__global__ __launch_bounds__(NT) void gn_stats_kernel(const float* __restrict__ x, float* __restrict__ part, Geo g,
                                                      const float* __restrict__ x2, int C1) {
  __shared__ float red[2 * MAXC];  // [rows_par][C] sums then sumsqs; rows_par*C <= max(1024, C)
  int b = blockIdx.x / g.nch, ch = blockIdx.x - b * g.nch;
  int p0 = ch * g.ppc, p1 = min(g.HW, p0 + g.ppc);
  int tid = threadIdx.x;
  int prow = tid / g.tpr, cfirst = tid - prow * g.tpr;
  if (prow < g.rows_par) {
    for (int cq = cfirst; cq < g.C4; cq += g.tpr) {
      f32x4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
      int ldx;
      const float* xb = gn_src(x, x2, C1, g.C, g.HW, b, cq * 4, &ldx);
      int p = p0 + prow;
      const int R = g.rows_par;
      for (; p + 3 * R < p1; p += 4 * R) {      // 4 independent 16-B loads in flight per thread
        f32x4 v0 = *reinterpret_cast<const f32x4*>(xb + (long)p * ldx);
        f32x4 v1 = *reinterpret_cast<const f32x4*>(xb + (long)(p + R) * ldx);
        f32x4 v2 = *reinterpret_cast<const f32x4*>(xb + (long)(p + 2 * R) * ldx);
        f32x4 v3 = *reinterpret_cast<const f32x4*>(xb + (long)(p + 3 * R) * ldx);
        s += (v0 + v1) + (v2 + v3);
        ss += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
      }
      for (; p < p1; p += R) {
        f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)p * ldx);
        s += v;
        ss += v * v;
      }
      float* r0 = red + prow * g.C + cq * 4;
      float* r1 = red + g.rows_par * g.C + prow * g.C + cq * 4;
#pragma unroll
      for (int e = 0; e < 4; ++e) { r0[e] = s[e]; r1[e] = ss[e]; }
    }
  }
  __syncthreads();
  for (int grp = tid; grp < g.G; grp += NT) {
    float S = 0.f, SS = 0.f;
    for (int r = 0; r < g.rows_par; ++r)
      for (int c = grp * g.cpg; c < (grp + 1) * g.cpg; ++c) {
        S += red[r * g.C + c];
        SS += red[g.rows_par * g.C + r * g.C + c];
      }
    float n = (float)((p1 - p0) * g.cpg);
    float mean = S / n;
    float m2 = fmaxf(SS - S * mean, 0.f);
    float* o = part + (((long)b * g.nch + ch) * g.G + grp) * 2;
    o[0] = mean;
    o[1] = m2;
  }
}

// combine the chunk partials of image b into LDS mean/rstd (Chan et al. parallel variance)
__device__ __forceinline__ void gn_combine(const float* part, Geo g, int b, float eps, float* s_mean, float* s_rstd) {
  for (int grp = threadIdx.x; grp < g.G; grp += NT) {
    float n = 0.f, mean = 0.f, m2 = 0.f;
    for (int ch = 0; ch < g.nch; ++ch) {
      int p0 = ch * g.ppc, p1 = min(g.HW, p0 + g.ppc);
      float nb = (float)((p1 - p0) * g.cpg);
      const float* o = part + (((long)b * g.nch + ch) * g.G + grp) * 2;
      float d = o[0] - mean, nt = n + nb;
      mean += d * (nb / nt);
      m2 += o[1] + d * d * (n * nb / nt);
      n = nt;
    }
    s_mean[grp] = mean;
    s_rstd[grp] = rsqrtf(m2 / n + eps);
  }
}

__global__ __launch_bounds__(NT) void gn_apply_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      const float* __restrict__ part, float* __restrict__ mean_out,
                                                      float* __restrict__ rstd_out, Geo g, float eps, int silu,
                                                      const float* __restrict__ x2, int C1) {
  __shared__ float s_mean[256], s_rstd[256];
  int b = blockIdx.x / g.nch, ch = blockIdx.x - b * g.nch;
  gn_combine(part, g, b, eps, s_mean, s_rstd);
  __syncthreads();
  int tid = threadIdx.x;
  if (ch == 0)
    for (int grp = tid; grp < g.G; grp += NT) {
      mean_out[b * g.G + grp] = s_mean[grp];
      rstd_out[b * g.G + grp] = s_rstd[grp];
    }
  int prow = tid / g.tpr, cfirst = tid - prow * g.tpr;
  if (prow >= g.rows_par) return;
  int p0 = ch * g.ppc, p1 = min(g.HW, p0 + g.ppc);
  for (int cq = cfirst; cq < g.C4; cq += g.tpr) {
    int c0 = cq * 4;
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 mu, rs;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int grp = (c0 + e) / g.cpg;
      mu[e] = s_mean[grp];
      rs[e] = s_rstd[grp];
    }
    f32x4 scale = rs * ga, shift = be - mu * scale;
    long base = ((long)b * g.HW) * g.C + c0;
    int ldx;
    const float* xb = gn_src(x, x2, C1, g.C, g.HW, b, c0, &ldx);
    int p = p0 + prow;
    const int R = g.rows_par;
    for (; p + 3 * R < p1; p += 4 * R) {        // 4 independent 16-B loads in flight per thread
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4*>(xb + (long)(p + u * R) * ldx);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        f32x4 z = v[u] * scale + shift;
        if (silu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = silu_f(z[e]);
        }
        *reinterpret_cast<f32x4*>(y + base + (long)(p + u * R) * g.C) = z;
      }
    }
    for (; p < p1; p += R) {
      f32x4 v = *reinterpret_cast<const f32x4*>(xb + (long)p * ldx);
      f32x4 z = v * scale + shift;
      if (silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = silu_f(z[e]);
      }
      *reinterpret_cast<f32x4*>(y + base + (long)p * g.C) = z;
    }
  }
}


// ---------------------------------------------------------- forward, one pass ----
// When a whole (image, channel slab) fits in the registers of one workgroup the forward reads x ONCE (8 B/elem, the
// algorithmic figure): workgroup = (image b, slab of SC channels made of whole groups), thread = (pixel lane pl,
// channel quad q) holding NV float4 (pixels pl, pl+PL, ...).  Two-pass moments in registers (mean, then centred
// sum of squares), LDS tree reductions in a fixed order (deterministic), then normalise + SiLU from registers.
struct SlabGeo {
  int HW, C, G, cpg, SC, qpr, PL, P2, nslab, gps, qpg;   // qpr quads per slab row, PL pixel lanes, P2 = pow2 >= PL
  int xgrp;   // slab rows are not whole 128-B lines: the slabs of one image go to ONE XCD (block ids 8 apart), so the
              // lines two neighbouring slabs share are fetched from HBM once and found in that XCD's L2 by the second
};
// block id -> (image, slab).  Hardware deals consecutive block ids round-robin to the 8 XCDs.
__device__ __forceinline__ void slab_of_block(const SlabGeo& s, int& b, int& sl) {
  const int bid = blockIdx.x;
  if (s.xgrp) {                        // B % 8 == 0 (host): image b = 8 b' + x owns block ids (b' nslab + j) 8 + x
    const int x = bid & 7, r = bid >> 3;
    const int bp = r / s.nslab;
    sl = r - bp * s.nslab;
    b = bp * 8 + x;
  } else {
    b = bid / s.nslab;
    sl = bid - b * s.nslab;
  }
}

// sum over the pixel lanes of each quad; result for quad q in red[q]
__device__ __forceinline__ void slab_reduce(float* red, float val, const SlabGeo& s, int pl, bool act) {
  red[threadIdx.x] = val;
  for (int st = s.P2 >> 1; st >= 1; st >>= 1) {
    __syncthreads();
    if (act && pl < st && pl + st < s.PL) red[threadIdx.x] += red[threadIdx.x + st * s.qpr];
  }
  __syncthreads();
}

// the same over float4 (per-channel sums); result for quad q in red[q]
template <int NTH>
__device__ __forceinline__ void slab_reduce4(f32x4* red, f32x4 val, const SlabGeo& s, int pl, bool act) {
  red[threadIdx.x] = val;
  for (int st = s.P2 >> 1; st >= 1; st >>= 1) {
    __syncthreads();
    if (act && pl < st && pl + st < s.PL) red[threadIdx.x] += red[threadIdx.x + st * s.qpr];
  }
  __syncthreads();
}

// x2 != nullptr: channels [C1, C) of the (virtual) concatenation live in x2 ([B][HW][C-C1]); the source is chosen per
// channel quad (C1 % 4 == 0), so a slab may straddle the split
// PERCH = true: channels per group not a multiple of 4 (pruned widths 3 / 6 / 9, CelebA 7 ..., SD 10 ...): a thread's
// float4 may straddle two groups, so the pixel-lane reductions run PER CHANNEL (float4 wide) and a group's moments are
// sums over its cpg channel entries; mean / rstd are then per-channel values of the thread's quad.
template <int NV, bool PERCH = false, int NTH = NT>
__global__ __launch_bounds__(NTH) void gn_slab_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1,
                                                     float* __restrict__ y,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     SlabGeo s, float eps, int silu) {
  __shared__ float red[PERCH ? 4 : NTH];
  __shared__ f32x4 red4[PERCH ? NTH : 1];
  __shared__ float s_mean[64], s_rstd[64];
  int b, sl;
  slab_of_block(s, b, sl);
  const int tid = threadIdx.x;
  const int pl = tid / s.qpr, q = tid - pl * s.qpr;
  const bool act = pl < s.PL;
  const int c0 = sl * s.SC + q * 4;
  const long base = ((long)b * s.HW) * s.C + c0;
  const bool second = c0 >= C1;                       // per thread (its channel quad lies in one source)
  const int ldin = second ? s.C - C1 : C1;
  const float* xin = (second ? x2 + (c0 - C1) : x + c0) + ((long)b * s.HW) * ldin;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int p = pl + i * s.PL;
    v[i] = (act && p < s.HW) ? *reinterpret_cast<const f32x4*>(xin + (long)p * ldin) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) acc += v[i];
  const float inv_n = 1.f / (float)((long)s.HW * s.cpg);
  const float* chan = reinterpret_cast<const float*>(red4);     // PERCH: per-channel sums of the slab, channel c at [c]
  int gi[4] = {0, 0, 0, 0};                                        // PERCH: group (within the slab) of each channel of the quad
  if (PERCH) {
#pragma unroll
    for (int e = 0; e < 4; ++e) gi[e] = act ? (q * 4 + e) / s.cpg : 0;
    slab_reduce4<NTH>(red4, acc, s, pl, act);
  } else {
    slab_reduce(red, (acc[0] + acc[1]) + (acc[2] + acc[3]), s, pl, act);
  }
  if (tid < s.gps) {
    float S = 0.f;
    if (PERCH) for (int j = 0; j < s.cpg; ++j) S += chan[tid * s.cpg + j];
    else for (int j = 0; j < s.qpg; ++j) S += red[tid * s.qpg + j];
    s_mean[tid] = S * inv_n;
  }
  __syncthreads();
  f32x4 mu;
  if (PERCH) mu = f32x4{s_mean[gi[0]], s_mean[gi[1]], s_mean[gi[2]], s_mean[gi[3]]};
  else { const float m1 = act ? s_mean[q / s.qpg] : 0.f; mu = f32x4{m1, m1, m1, m1}; }
  acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int p = pl + i * s.PL;
    f32x4 d = v[i] - mu;
    if (act && p < s.HW) acc += d * d;
  }
  if (PERCH) {
    __syncthreads();                       // every reader of the first reduction's sums is done
    slab_reduce4<NTH>(red4, acc, s, pl, act);
  } else {
    slab_reduce(red, (acc[0] + acc[1]) + (acc[2] + acc[3]), s, pl, act);
  }
  if (tid < s.gps) {
    float SS = 0.f;
    if (PERCH) for (int j = 0; j < s.cpg; ++j) SS += chan[tid * s.cpg + j];
    else for (int j = 0; j < s.qpg; ++j) SS += red[tid * s.qpg + j];
    float rs = rsqrtf(SS * inv_n + eps);
    s_rstd[tid] = rs;
    mean_out[b * s.G + sl * s.gps + tid] = s_mean[tid];
    rstd_out[b * s.G + sl * s.gps + tid] = rs;
  }
  __syncthreads();
  if (!act) return;
  f32x4 rs;
  if (PERCH) rs = f32x4{s_rstd[gi[0]], s_rstd[gi[1]], s_rstd[gi[2]], s_rstd[gi[3]]};
  else { const float r1 = s_rstd[q / s.qpg]; rs = f32x4{r1, r1, r1, r1}; }
  f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 scale = ga * rs, shift = be - scale * mu;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int p = pl + i * s.PL;
    if (p < s.HW) {
      f32x4 z = v[i] * scale + shift;
      if (silu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = silu_f(z[e]);
      }
      *reinterpret_cast<f32x4*>(y + base + (long)p * s.C) = z;
    }
  }
}

// pick the channel slab; returns NV (0 = no one-pass plan: odd channels-per-group or too many pixels)
static int make_slab(const gad_groupnorm_args* a, SlabGeo* out, const int nth = NT, const int maxnv = 32) {
  const int cpg = a->C / a->G;
  SlabGeo best{};
  int best_nv = 0;
  long best_score = -1;
  for (int k = 1; k * cpg <= a->C && k <= 64; ++k) {
    int SC = k * cpg;
    if (a->C % SC != 0 || SC % 4 != 0) continue;       // whole groups, whole float4 quads
    // slab rows should be whole 128-B lines, or two workgroups fetch each line; where no such slab fits the registers the
    // slabs of an image are sent to one XCD instead (xgrp: needs B % 8 == 0) and share the line through its L2
    const bool lines = (SC * 4) % 128 == 0 || SC == a->C;
    // (only where the sharing can happen: a few slabs per image, and few enough workgroups that the slabs in flight on an
    //  XCD fit its 4 MB L2 - measured: 4 slabs of 96 B at C = 96, B = 128: 31.9 -> 24.1 us; the same at B = 1024, or 16
    //  slabs at C = 384: no better / slower than the two-pass plan, tools/gn_ab.py)
    if (!lines && (a->B % 8 != 0 || a->C / SC > 8 || (long)a->B * (a->C / SC) > 1024)) continue;
    int qpr = SC / 4;
    if (qpr > nth) break;
    int PL = nth / qpr;
    int nv = (a->HW + PL - 1) / PL;
    if (nv > maxnv) continue;
    long blocks = (long)a->B * (a->C / SC);
    // enough workgroups for 256 CUs first, then the longest contiguous run per pixel
    long score = (blocks >= 1024 ? (1L << 40) : blocks << 16) + SC + (lines ? (1L << 41) : 0);   // whole-line plans first
    if (score > best_score) {
      best_score = score;
      best_nv = nv;
      int P2 = 1;
      while (P2 < PL) P2 <<= 1;
      best = SlabGeo{a->HW, a->C, a->G, cpg, SC, qpr, PL, P2, a->C / SC, k, cpg / 4, lines ? 0 : 1};
    }
  }
  if (!best_nv) return 0;
  *out = best;
  return best_nv <= 4 ? 4 : best_nv <= 8 ? 8 : best_nv <= 16 ? 16 : 32;
}

// ------------------------------------------- forward, one pass, Winograd output ----
// GroupNorm + SiLU whose consumer is a 3x3 / stride-1 convolution on the Winograd F(4x4, 3x3) route (ResnetBlock2D:
// norm -> silu -> conv, reference diffusers ResnetBlock2D.forward): instead of y the kernel writes the route's
// transformed input V [36][tiles][C] = B^T y_patch B directly, so y never exists in HBM and the route's own input-transform
// launch (one read of y, one write of V) disappears.  Same slab plan and the same statistics arithmetic as gn_slab_kernel
// (mean / rstd bit-identical); the normalised slab then goes to LDS ([pixel][SC channels], up to 128 KB: a 32 x 32 map of 32
// channels) and every thread transforms 6 x 6 patches of one channel quad from there - the halo re-reads (2.25x) stay on
// chip.  LDS image: pixel column x is stored at x ^ ((x >> 2) & 1), which puts the same patch position of two neighbouring
// tiles on opposite bank halves, so a 16-lane ds_read_b128 group (8 quads x 2 tiles at SC = 32) is conflict-free.
// V rows are whole 128-B lines (SC % 32 == 0).
struct GnWino {
  float* V;
  int W, TH, TW;
  long pos_stride;           // floats between two Winograd positions: B * TH * TW * C
};

template <int NV>
__global__ __launch_bounds__(512) void gn_wino4_kernel(const float* __restrict__ x, const float* __restrict__ x2, int C1,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                      SlabGeo s, float eps, int silu, GnWino w) {
  constexpr int NTH = 512;
  extern __shared__ __attribute__((aligned(16))) float gn_img[];       // [HW][SC]
  __shared__ float red[NTH];
  __shared__ float s_mean[64], s_rstd[64];
  int b, sl;
  slab_of_block(s, b, sl);
  const int tid = threadIdx.x;
  const int pl = tid / s.qpr, q = tid - pl * s.qpr;
  const bool act = pl < s.PL;
  const int c0 = sl * s.SC + q * 4;
  const bool second = c0 >= C1;
  const int ldin = second ? s.C - C1 : C1;
  const float* xin = (second ? x2 + (c0 - C1) : x + c0) + ((long)b * s.HW) * ldin;
  f32x4 v[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int p = pl + i * s.PL;
    v[i] = (act && p < s.HW) ? *reinterpret_cast<const f32x4*>(xin + (long)p * ldin) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) acc += v[i];
  const float inv_n = 1.f / (float)((long)s.HW * s.cpg);
  slab_reduce(red, (acc[0] + acc[1]) + (acc[2] + acc[3]), s, pl, act);
  if (tid < s.gps) {
    float S = 0.f;
    for (int j = 0; j < s.qpg; ++j) S += red[tid * s.qpg + j];
    s_mean[tid] = S * inv_n;
  }
  __syncthreads();
  const float m1 = act ? s_mean[q / s.qpg] : 0.f;
  const f32x4 mu = f32x4{m1, m1, m1, m1};
  acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    int p = pl + i * s.PL;
    f32x4 d = v[i] - mu;
    if (act && p < s.HW) acc += d * d;
  }
  slab_reduce(red, (acc[0] + acc[1]) + (acc[2] + acc[3]), s, pl, act);
  if (tid < s.gps) {
    float SS = 0.f;
    for (int j = 0; j < s.qpg; ++j) SS += red[tid * s.qpg + j];
    float rs = rsqrtf(SS * inv_n + eps);
    s_rstd[tid] = rs;
    mean_out[b * s.G + sl * s.gps + tid] = s_mean[tid];
    rstd_out[b * s.G + sl * s.gps + tid] = rs;
  }
  __syncthreads();
  f32x4* img4 = reinterpret_cast<f32x4*>(gn_img);
  if (act) {
    const float r1 = s_rstd[q / s.qpg];
    const f32x4 rs = f32x4{r1, r1, r1, r1};
    const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    const f32x4 scale = ga * rs, shift = be - scale * mu;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int p = pl + i * s.PL;
      if (p < s.HW) {
        f32x4 z = v[i] * scale + shift;
        if (silu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) z[e] = silu_f(z[e]);
        }
        const int yy = p / w.W, xx = p - yy * w.W;
        img4[(yy * w.W + (xx ^ ((xx >> 2) & 1))) * s.qpr + q] = z;
      }
    }
  }
  __syncthreads();
  // ---- 6 x 6 patches of one channel quad -> V (the arithmetic of wino4_input_kernel, gemm_f32.hip) ----
  const int H = s.HW / w.W;
  const int items = w.TH * w.TW * s.qpr;
  for (int it = tid; it < items; it += NTH) {
    const int tile = it / s.qpr, qq = it - tile * s.qpr;
    const int ty = tile / w.TW, tx = tile - ty * w.TW;
    f32x4 t[6][6];
#pragma unroll
    for (int dx = 0; dx < 6; ++dx) {
      const int xx = 4 * tx - 1 + dx;
      const bool xok = xx >= 0 && xx < w.W;
      const int xs = xx ^ ((xx >> 2) & 1);
      f32x4 d[6], col[6];
#pragma unroll
      for (int dy = 0; dy < 6; ++dy) {
        const int yy = 4 * ty - 1 + dy;
        const bool ok = xok && yy >= 0 && yy < H;
        d[dy] = ok ? img4[(yy * w.W + xs) * s.qpr + qq] : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      gadk::wino4_bt(d, col);
#pragma unroll
      for (int i = 0; i < 6; ++i) t[i][dx] = col[i];
    }
    float* out = w.V + ((long)b * (w.TH * w.TW) + tile) * s.C + sl * s.SC + qq * 4;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      f32x4 vv[6];
      gadk::wino4_bt(t[i], vv);
#pragma unroll
      for (int j = 0; j < 6; ++j) *reinterpret_cast<f32x4*>(out + (long)(6 * i + j) * w.pos_stride) = vv[j];
    }
  }
}

// slab plan of gn_wino4_kernel: whole groups, whole 128-B lines, the normalised slab within 128 KB of LDS, <= 16 slots per
// thread at 512 threads; among those enough workgroups for the chip first, then the widest slab.  Returns NV (0: no plan)
static int make_slab_wino(const gad_groupnorm_args* a, SlabGeo* out) {
  const int cpg = a->C / a->G, nth = 512;
  if (cpg % 4 != 0) return 0;
  int best_nv = 0;
  long best_score = -1;
  for (int k = 1; k * cpg <= a->C && k <= 64; ++k) {
    const int SC = k * cpg;
    if (a->C % SC != 0 || SC % 32 != 0) continue;
    if ((long)a->HW * SC * 4 > 128 * 1024) break;
    const int qpr = SC / 4;
    if (qpr > nth) break;
    const int PL = nth / qpr, nv = (a->HW + PL - 1) / PL;
    if (nv > 16) continue;
    const long blocks = (long)a->B * (a->C / SC);
    const long score = (blocks >= 512 ? (1L << 40) : blocks << 16) + SC;
    if (score > best_score) {
      best_score = score;
      best_nv = nv;
      int P2 = 1;
      while (P2 < PL) P2 <<= 1;
      *out = SlabGeo{a->HW, a->C, a->G, cpg, SC, qpr, PL, P2, a->C / SC, k, cpg / 4, 0};
    }
  }
  return !best_nv ? 0 : best_nv <= 4 ? 4 : best_nv <= 8 ? 8 : 16;
}

// --------------------------------------------------------------- backward ----
// g_e = dy * silu'(z) (or dy);  per channel partials  A_c = sum g_e,  Bx_c = sum g_e * xhat_e
__device__ __forceinline__ float act_grad(float dy, float z, int silu) {
  if (!silu) return dy;
  float s = 1.f / (1.f + expf(-z));
  return dy * s * (1.f + z * (1.f - s));
}

// ------------------------------------------------------ backward, one pass ----
// The backward reads x and dy ONCE (12 B/elem instead of 20) when an (image, channel slab) fits in the registers of one
// workgroup - the forward's slab plan with two float4 per slot (xhat, g), so 512 threads x 16 slots cover a
// 32 x 32 x 32-channel slab.  Per-channel sums A_c = sum g, Bx_c = sum g xhat are reduced over the pixel lanes by a fixed
// LDS tree (float4 wide), written as the (image, channel) partials of dgamma / dbeta, combined per group with gamma, and
// dx = rstd (g gamma - m1 - xhat m2) comes from the registers.
template <int NV, int NTH>
__global__ __launch_bounds__(NTH) void gn_bwd_slab_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ part,
                                                          SlabGeo s, int silu, const float* __restrict__ dx_add) {
  __shared__ f32x4 red_a[NTH], red_b[NTH];
  __shared__ float s_m1[64], s_m2[64];
  int b, sl;
  slab_of_block(s, b, sl);
  const int tid = threadIdx.x;
  const int pl = tid / s.qpr, q = tid - pl * s.qpr;
  const bool act = pl < s.PL;
  const int c0 = sl * s.SC + (act ? q : 0) * 4;
  const long base = ((long)b * s.HW) * s.C + c0;
  int gi[4];                         // group (within the slab) of each channel of the quad: any channels-per-group
  f32x4 mu, rs;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    gi[e] = ((act ? q : 0) * 4 + e) / s.cpg;
    mu[e] = mean[b * s.G + sl * s.gps + gi[e]];
    rs[e] = rstd[b * s.G + sl * s.gps + gi[e]];
  }
  const f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
  f32x4 xh[NV], ge[NV];
  f32x4 sa = {0.f, 0.f, 0.f, 0.f}, sb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = pl + i * s.PL;
    if (act && p < s.HW) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + base + (long)p * s.C);
      const f32x4 d = *reinterpret_cast<const f32x4*>(dy + base + (long)p * s.C);
      xh[i] = (v - mu) * rs;
#pragma unroll
      for (int e = 0; e < 4; ++e) ge[i][e] = act_grad(d[e], xh[i][e] * ga[e] + be[e], silu);
      sa += ge[i];
      sb += ge[i] * xh[i];
    } else {
      xh[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      ge[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  slab_reduce4<NTH>(red_a, sa, s, pl, act);
  slab_reduce4<NTH>(red_b, sb, s, pl, act);
  if (tid < s.qpr) {                                   // pixel lane 0 holds the sums of its channel quad
    const int c = sl * s.SC + tid * 4;
    float* o = part + ((long)b * s.C + c) * 2;
    const f32x4 A = red_a[tid], Bx = red_b[tid];
#pragma unroll
    for (int e = 0; e < 4; ++e) { o[2 * e] = A[e]; o[2 * e + 1] = Bx[e]; }
  }
  if (tid < s.gps) {
    const float inv_n = 1.f / (float)((long)s.HW * s.cpg);
    const float* ca = reinterpret_cast<const float*>(red_a);       // per-channel sums of the slab, channel c at [c]
    const float* cb = reinterpret_cast<const float*>(red_b);
    float s1 = 0.f, s2 = 0.f;
    for (int j = 0; j < s.cpg; ++j) {
      const int c = tid * s.cpg + j;
      const float gj = gamma[sl * s.SC + c];
      s1 += ca[c] * gj;
      s2 += cb[c] * gj;
    }
    s_m1[tid] = s1 * inv_n;
    s_m2[tid] = s2 * inv_n;
  }
  __syncthreads();
  if (!act) return;
  const f32x4 m1 = {s_m1[gi[0]], s_m1[gi[1]], s_m1[gi[2]], s_m1[gi[3]]}, m2 = {s_m2[gi[0]], s_m2[gi[1]], s_m2[gi[2]], s_m2[gi[3]]};
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int p = pl + i * s.PL;
    if (p < s.HW) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = rs[e] * (ge[i][e] * ga[e] - m1[e] - xh[i][e] * m2[e]);
      if (dx_add) o += *reinterpret_cast<const f32x4*>(dx_add + base + (long)p * s.C);     // the bypass branch's gradient
      *reinterpret_cast<f32x4*>(dx + base + (long)p * s.C) = o;
    }
  }
}

__global__ __launch_bounds__(NT) void gn_bwd_stats_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          float* __restrict__ part, Geo g, int silu) {
  __shared__ float red[2 * MAXC];
  int b = blockIdx.x / g.nch, ch = blockIdx.x - b * g.nch;
  int p0 = ch * g.ppc, p1 = min(g.HW, p0 + g.ppc);
  int tid = threadIdx.x;
  int prow = tid / g.tpr, cfirst = tid - prow * g.tpr;
  if (prow < g.rows_par) {
    for (int cq = cfirst; cq < g.C4; cq += g.tpr) {
      int c0 = cq * 4;
      f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
      f32x4 mu, rs;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        int grp = (c0 + e) / g.cpg;
        mu[e] = mean[b * g.G + grp];
        rs[e] = rstd[b * g.G + grp];
      }
      f32x4 sa = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
      long base = ((long)b * g.HW) * g.C + c0;
      for (int p = p0 + prow; p < p1; p += g.rows_par) {
        f32x4 v = *reinterpret_cast<const f32x4*>(x + base + (long)p * g.C);
        f32x4 d = *reinterpret_cast<const f32x4*>(dy + base + (long)p * g.C);
        f32x4 xh = (v - mu) * rs;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float ge = act_grad(d[e], xh[e] * ga[e] + be[e], silu);
          sa[e] += ge;
          sb[e] += ge * xh[e];
        }
      }
      float* r0 = red + prow * g.C + c0;
      float* r1 = red + g.rows_par * g.C + prow * g.C + c0;
#pragma unroll
      for (int e = 0; e < 4; ++e) { r0[e] = sa[e]; r1[e] = sb[e]; }
    }
  }
  __syncthreads();
  for (int c = tid; c < g.C; c += NT) {
    float A = 0.f, Bx = 0.f;
    for (int r = 0; r < g.rows_par; ++r) {
      A += red[r * g.C + c];
      Bx += red[g.rows_par * g.C + r * g.C + c];
    }
    float* o = part + (((long)b * g.nch + ch) * g.C + c) * 2;
    o[0] = A;
    o[1] = Bx;
  }
}

__global__ __launch_bounds__(NT) void gn_bwd_apply_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          float* __restrict__ dx, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ part,
                                                          Geo g, int silu, const float* __restrict__ dx_add) {
  __shared__ float s_a[MAXC], s_b[MAXC];     // per channel sums (gamma-weighted)
  __shared__ float s_s1[256], s_s2[256];     // per group
  int b = blockIdx.x / g.nch, ch = blockIdx.x - b * g.nch;
  int tid = threadIdx.x;
  for (int c = tid; c < g.C; c += NT) {
    float A = 0.f, Bx = 0.f;
    for (int k = 0; k < g.nch; ++k) {
      const float* o = part + (((long)b * g.nch + k) * g.C + c) * 2;
      A += o[0];
      Bx += o[1];
    }
    float ga = gamma[c];
    s_a[c] = A * ga;
    s_b[c] = Bx * ga;
  }
  __syncthreads();
  for (int grp = tid; grp < g.G; grp += NT) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = grp * g.cpg; c < (grp + 1) * g.cpg; ++c) {
      s1 += s_a[c];
      s2 += s_b[c];
    }
    float inv_n = 1.f / (float)((long)g.HW * g.cpg);
    s_s1[grp] = s1 * inv_n;
    s_s2[grp] = s2 * inv_n;
  }
  __syncthreads();
  int prow = tid / g.tpr, cfirst = tid - prow * g.tpr;
  if (prow >= g.rows_par) return;
  int p0 = ch * g.ppc, p1 = min(g.HW, p0 + g.ppc);
  for (int cq = cfirst; cq < g.C4; cq += g.tpr) {
    int c0 = cq * 4;
    f32x4 ga = *reinterpret_cast<const f32x4*>(gamma + c0), be = *reinterpret_cast<const f32x4*>(beta + c0);
    f32x4 mu, rs, m1, m2;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int grp = (c0 + e) / g.cpg;
      mu[e] = mean[b * g.G + grp];
      rs[e] = rstd[b * g.G + grp];
      m1[e] = s_s1[grp];
      m2[e] = s_s2[grp];
    }
    long base = ((long)b * g.HW) * g.C + c0;
    for (int p = p0 + prow; p < p1; p += g.rows_par) {
      f32x4 v = *reinterpret_cast<const f32x4*>(x + base + (long)p * g.C);
      f32x4 d = *reinterpret_cast<const f32x4*>(dy + base + (long)p * g.C);
      f32x4 xh = (v - mu) * rs, o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float ge = act_grad(d[e], xh[e] * ga[e] + be[e], silu);
        o[e] = rs[e] * (ge * ga[e] - m1[e] - xh[e] * m2[e]);
      }
      if (dx_add) o += *reinterpret_cast<const f32x4*>(dx_add + base + (long)p * g.C);
      *reinterpret_cast<f32x4*>(dx + base + (long)p * g.C) = o;
    }
  }
}

static int check(const gad_groupnorm_args* a, const char* who) {
  GAD_CHECK(a && a->x && a->y && a->gamma && a->beta && a->mean && a->rstd, "%s: null pointer", who);
  GAD_CHECK(a->B > 0 && a->HW > 0 && a->C > 0 && a->G > 0 && a->C % a->G == 0, "%s: bad shape", who);
  GAD_CHECK(a->C % 4 == 0 && a->C <= MAXC && a->G <= 256, "%s: needs C%%4==0, C<=3072, G<=256 (C=%d G=%d)", who, a->C, a->G);
  GAD_CHECK(gad_aligned16(a->x) && gad_aligned16(a->y) && gad_aligned16(a->gamma) && gad_aligned16(a->beta), "%s: pointers must be 16-byte aligned", who);
  int64_t need = gad_groupnorm_workspace_bytes(a);
  GAD_CHECK(a->ws && a->ws_bytes >= need, "%s: workspace too small (%lld < %lld)", who, (long long)a->ws_bytes, (long long)need);
  return 0;
}

}  // namespace

extern "C" int64_t gad_groupnorm_workspace_bytes(const gad_groupnorm_args* a) {
  Geo g = make_geo(a);
  int64_t parts = (int64_t)a->B * g.nch * a->C * 2 * (int64_t)sizeof(float);  // covers fwd (G<=C) and bwd
  const int64_t r1 = gad_reduce::ws_bytes(1, (long)a->B * g.nch, 2 * a->C);   // + dgamma/dbeta reduction (two-pass plan)
  const int64_t r2 = gad_reduce::ws_bytes(1, (long)a->B, 2 * a->C);           //   (one-pass plan: one partial row per image)
  return parts + (r1 > r2 ? r1 : r2);
}

extern "C" int gad_groupnorm_one_pass(const gad_groupnorm_args* a) {
  if (!a || a->B <= 0 || a->HW <= 0 || a->C <= 0 || a->G <= 0 || a->C % a->G != 0 || a->C % 4 != 0) return 0;
  if (a->flags & GAD_GN_TWO_PASS) return 0;
  SlabGeo sg;
  return make_slab(a, &sg) ? 1 : 0;
}

extern "C" int gad_groupnorm_silu_fwd(const gad_groupnorm_args* a, void* stream) {
  if (check(a, "gad_groupnorm_silu_fwd")) return 1;
  hipStream_t st = (hipStream_t)stream;
  SlabGeo sg;
  int nv = (a->flags & GAD_GN_TWO_PASS) ? 0 : make_slab(a, &sg);
  if (a->x2)
    GAD_CHECK(a->C1 > 0 && a->C1 < a->C && a->C1 % 4 == 0 && (a->C - a->C1) % 4 == 0 && gad_aligned16(a->x2), "gad_groupnorm_silu_fwd: bad C1 / x2");
  if (nv) {
    const int c1 = a->x2 ? a->C1 : a->C;
    dim3 sgrid(a->B * sg.nslab), sblock(NT);
    const bool perch = (a->C / a->G) % 4 != 0;
    // 32 slots per thread (185 registers) leave room for two 4-wave workgroups per CU only: where a 512-thread x 16-slot
    // plan covers the same slab (100 registers), more waves overlap their load / reduce / store phases - measured
    // [1024, 32, 32, 128]: 243 -> 228 us (4.7 TB/s), [128, 32, 32, 256]: 54.8 -> 51.5 us (5.2 TB/s)
    SlabGeo sgw;
    if (nv == 32 && !perch && make_slab(a, &sgw, 512, 16) == 16 && sgw.SC == sg.SC) {
      dim3 wgrid(a->B * sgw.nslab);
      hipLaunchKernelGGL((gn_slab_kernel<16, false, 512>), wgrid, dim3(512), 0, st, a->x, a->x2, c1, a->y, a->gamma, a->beta,
                         a->mean, a->rstd, sgw, a->eps, a->silu);
      GAD_LAUNCH_CHECK("gn_slab(512)");
      return 0;
    }
#define GAD_GNF(NV_, P_) hipLaunchKernelGGL((gn_slab_kernel<NV_, P_>), sgrid, sblock, 0, st, a->x, a->x2, c1, a->y, a->gamma, a->beta, \
                                            a->mean, a->rstd, sg, a->eps, a->silu)
    switch (nv) {
      case 4: if (perch) GAD_GNF(4, true); else GAD_GNF(4, false); break;
      case 8: if (perch) GAD_GNF(8, true); else GAD_GNF(8, false); break;
      case 16: if (perch) GAD_GNF(16, true); else GAD_GNF(16, false); break;
      default: if (perch) GAD_GNF(32, true); else GAD_GNF(32, false); break;
    }
#undef GAD_GNF
    GAD_LAUNCH_CHECK("gn_slab");
    return 0;
  }
  Geo g = make_geo(a);
  dim3 grid(a->B * g.nch), block(NT);
  hipLaunchKernelGGL(gn_stats_kernel, grid, block, 0, st, a->x, (float*)a->ws, g, a->x2, a->C1);
  GAD_LAUNCH_CHECK("gn_stats");
  hipLaunchKernelGGL(gn_apply_kernel, grid, block, 0, st, a->x, a->y, a->gamma, a->beta, (const float*)a->ws, a->mean, a->rstd, g, a->eps, a->silu,
                     a->x2, a->C1);
  GAD_LAUNCH_CHECK("gn_apply");
  return 0;
}


// GroupNorm (+SiLU) forward writing the F(4x4, 3x3) Winograd input transform of its result (see gn_wino4_kernel)
static bool gn_wino_shapes_ok(const gad_groupnorm_args* a, int32_t W) {
  if (!a || a->B <= 0 || a->HW <= 0 || a->C <= 0 || a->G <= 0 || a->C % a->G != 0 || a->C % 32 != 0 || a->G > 256) return false;
  if (W <= 0 || a->HW % W != 0 || (W & 3) || ((a->HW / W) & 3)) return false;
  if (a->x2 && !(a->C1 > 0 && a->C1 < a->C && a->C1 % 4 == 0 && (a->C - a->C1) % 4 == 0)) return false;
  return true;
}
extern "C" int gad_groupnorm_wino4_ok(const gad_groupnorm_args* a, int32_t W) {
  SlabGeo sg;
  return gn_wino_shapes_ok(a, W) && make_slab_wino(a, &sg) ? 1 : 0;
}
extern "C" int gad_groupnorm_silu_wino4(const gad_groupnorm_args* a, float* V, int32_t W, void* stream) {
  GAD_CHECK(a && a->x && a->gamma && a->beta && a->mean && a->rstd && V, "gad_groupnorm_silu_wino4: null pointer");
  GAD_CHECK(gn_wino_shapes_ok(a, W), "gad_groupnorm_silu_wino4: shapes outside the plan (C %% 32, W and H multiples of 4; C=%d G=%d HW=%d W=%d)", a->C, a->G, a->HW, W);
  GAD_CHECK(gad_aligned16(a->x) && gad_aligned16(V) && gad_aligned16(a->gamma) && gad_aligned16(a->beta) && (!a->x2 || gad_aligned16(a->x2)),
            "gad_groupnorm_silu_wino4: pointers must be 16-byte aligned");
  SlabGeo sg;
  const int nv = make_slab_wino(a, &sg);
  GAD_CHECK(nv, "gad_groupnorm_silu_wino4: no slab plan (gad_groupnorm_wino4_ok tells)");
  GnWino w;
  w.V = V; w.W = W; w.TH = (a->HW / W) / 4; w.TW = W / 4;
  w.pos_stride = (long)a->B * w.TH * w.TW * a->C;
  const int c1 = a->x2 ? a->C1 : a->C;
  const int bytes = a->HW * sg.SC * (int)sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(a->B * sg.nslab), block(512);
#define GAD_GNW(NV_)                                                                                                        \
  do {                                                                                                                      \
    static int lds_set = 0;                                                                                                 \
    if (bytes > lds_set) {                                                                                                  \
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gn_wino4_kernel<NV_>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes); \
      GAD_CHECK(e == hipSuccess, "gad_groupnorm_silu_wino4: cannot reserve %d bytes of LDS: %s", bytes, hipGetErrorString(e)); \
      lds_set = bytes;                                                                                                      \
    }                                                                                                                       \
    hipLaunchKernelGGL((gn_wino4_kernel<NV_>), grid, block, bytes, st, a->x, a->x2, c1, a->gamma, a->beta, a->mean, a->rstd, sg, a->eps, a->silu, w); \
  } while (0)
  if (nv == 4) GAD_GNW(4);
  else if (nv == 8) GAD_GNW(8);
  else GAD_GNW(16);
#undef GAD_GNW
  GAD_LAUNCH_CHECK("gn_wino4");
  return 0;
}

extern "C" int gad_groupnorm_silu_bwd(const gad_groupnorm_args* a, void* stream) {
  if (check(a, "gad_groupnorm_silu_bwd")) return 1;
  GAD_CHECK(a->dy && gad_aligned16(a->dy) && ((a->dgamma == nullptr) == (a->dbeta == nullptr)), "gad_groupnorm_silu_bwd: null/unaligned grad pointer");
  GAD_CHECK(!a->dx_add || gad_aligned16(a->dx_add), "gad_groupnorm_silu_bwd: dx_add must be 16-byte aligned");
  Geo g = make_geo(a);
  hipStream_t st = (hipStream_t)stream;
  if (!(a->flags & GAD_GN_TWO_PASS)) {
    // one pass: a slab plan whose two float4 per slot fit the registers - 256 threads x <= 16 slots, else 512 x <= 16
    SlabGeo sg;
    int nth = NT, nv = make_slab(a, &sg, NT, 16);
    if (!nv) { nth = 512; nv = make_slab(a, &sg, 512, 16); }
    if (nv) {
      float* part = (float*)a->ws;
      dim3 sgrid(a->B * sg.nslab);
#define GAD_GNB(NV_, NTH_) hipLaunchKernelGGL((gn_bwd_slab_kernel<NV_, NTH_>), sgrid, dim3(NTH_), 0, st, a->x, a->dy, a->y, a->gamma, \
                                              a->beta, a->mean, a->rstd, part, sg, a->silu, a->dx_add)
      if (nth == NT) {              // (the 8-slot instance is not built: hipcc allocates it 256 registers + scratch)
        if (nv <= 4) GAD_GNB(4, NT);
        else GAD_GNB(16, NT);
      } else {
        GAD_GNB(16, 512);
      }
#undef GAD_GNB
      GAD_LAUNCH_CHECK("gn_bwd_slab");
      if (a->dgamma) {
        float* ws2 = part + (long)a->B * a->C * 2;
        gad_reduce::launch(part, a->dbeta, a->dgamma, 1, (long)a->B, 2 * a->C, ws2, st);
        GAD_LAUNCH_CHECK("gn_bwd_param");
      }
      return 0;
    }
  }
  dim3 grid(a->B * g.nch), block(NT);
  hipLaunchKernelGGL(gn_bwd_stats_kernel, grid, block, 0, st, a->x, a->dy, a->gamma, a->beta, a->mean, a->rstd, (float*)a->ws, g, a->silu);
  GAD_LAUNCH_CHECK("gn_bwd_stats");
  hipLaunchKernelGGL(gn_bwd_apply_kernel, grid, block, 0, st, a->x, a->dy, a->y, a->gamma, a->beta, a->mean, a->rstd, (const float*)a->ws, g, a->silu, a->dx_add);
  GAD_LAUNCH_CHECK("gn_bwd_apply");
  // dbeta[c] = sum parts[.][c][0], dgamma[c] = sum parts[.][c][1]: a column sum of the [B*nch][2C] partials -
  // skipped when the caller wants no affine gradients (dgamma == dbeta == NULL: frozen norms of LoRA training)
  if (a->dgamma) {
    float* ws2 = (float*)a->ws + (long)a->B * g.nch * a->C * 2;
    gad_reduce::launch((const float*)a->ws, a->dbeta, a->dgamma, 1, (long)a->B * g.nch, 2 * a->C, ws2, st);
    GAD_LAUNCH_CHECK("gn_bwd_param");
  }
  return 0;
}
