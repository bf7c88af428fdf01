// HBM-bound elementwise / small-reduction kernels of the hot path (gfx950).
// Pattern: grid-stride float4 loops, <= 2048 workgroups of 256 threads (8 per CU),
// scalar tail; reductions are two-stage through a caller-owned workspace so results
// are deterministic run to run (no float atomics).
#include <stdarg.h>

#include "gad_common.h"
#include "gad_reduce.h"

// ------------------------------------------------------------------ error state ----
static thread_local char g_err[512] = "";
void gad_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* gad_last_error(void) { return g_err; }
extern "C" int gad_version(void) { return 100; }

namespace {

constexpr int NT = 256;
inline dim3 ew_grid(int64_t n_vec) {
  int64_t b = gad_ceil_div(n_vec > 0 ? n_vec : 1, NT);
  return dim3((unsigned)(b < 2048 ? b : 2048));
}

__device__ __forceinline__ float silu_f(float z) { return z / (1.f + expf(-z)); }

// Generic elementwise launcher: F(i4 index, vectors) over n elements; requires 16-B aligned
// pointers, handles n % 4 tail with the scalar functor.
#define EW_LOOP_VEC(n)                                                                     \
  const long nv = (n) >> 2;                                                                \
  const long gs = (long)gridDim.x * blockDim.x;                                            \
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gs)
#define EW_LOOP_TAIL(n) for (long i = ((n) & ~3L) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < (n); i += (long)gridDim.x * blockDim.x)

__global__ void silu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  EW_LOOP_VEC(n) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
  EW_LOOP_TAIL(n) y[i] = silu_f(x[i]);
}

__device__ __forceinline__ float silu_grad(float x, float dy) {
  float s = 1.f / (1.f + expf(-x));
  return dy * s * (1.f + x * (1.f - s));
}
__global__ void silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dx, long n) {
  EW_LOOP_VEC(n) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i], d = reinterpret_cast<const f32x4*>(dy)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = silu_grad(v[e], d[e]);
    reinterpret_cast<f32x4*>(dx)[i] = v;
  }
  EW_LOOP_TAIL(n) dx[i] = silu_grad(x[i], dy[i]);
}

// x / xp are NOT restrict-qualified: the samplers launch this in place (xp == x)
__global__ void ddim_step_kernel(const float* x, const float* __restrict__ eps, float* xp,
                                 long n, float sa, float sb, float spa, float spb, float clip) {
  EW_LOOP_VEC(n) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i], e4 = reinterpret_cast<const f32x4*>(eps)[i], o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float x0 = (v[e] - sb * e4[e]) / sa;
      if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
      o[e] = spa * x0 + spb * e4[e];
    }
    reinterpret_cast<f32x4*>(xp)[i] = o;
  }
  EW_LOOP_TAIL(n) {
    float x0 = (x[i] - sb * eps[i]) / sa;
    if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
    xp[i] = spa * x0 + spb * eps[i];
  }
}

__global__ void cfg_ddim_step_kernel(const float* x, const float* __restrict__ eu, const float* __restrict__ ec,
                                     float* xp, long n, float gd, float sa, float sb, float spa, float spb, float clip) {
  EW_LOOP_VEC(n) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i], u = reinterpret_cast<const f32x4*>(eu)[i], c = reinterpret_cast<const f32x4*>(ec)[i], o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float eps = u[e] + gd * (c[e] - u[e]);
      float x0 = (v[e] - sb * eps) / sa;
      if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
      o[e] = spa * x0 + spb * eps;
    }
    reinterpret_cast<f32x4*>(xp)[i] = o;
  }
  EW_LOOP_TAIL(n) {
    float eps = eu[i] + gd * (ec[i] - eu[i]);
    float x0 = (x[i] - sb * eps) / sa;
    if (clip > 0.f) x0 = fminf(fmaxf(x0, -clip), clip);
    xp[i] = spa * x0 + spb * eps;
  }
}

__global__ void to_image01_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
  EW_LOOP_VEC(n) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e] / 2.f + 0.5f, 0.f), 1.f);
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
  EW_LOOP_TAIL(n) y[i] = fminf(fmaxf(x[i] / 2.f + 0.5f, 0.f), 1.f);
}

// one block row per sample: per_sample elements share (sa, sb)
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ eps, const int64_t* __restrict__ t,
                                 const float* __restrict__ ac, float* __restrict__ xt, int B, long per) {
  long total = (long)B * per;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int b = (int)(i / per);
    float a = ac[t[b]];
    xt[i] = sqrtf(a) * x0[i] + sqrtf(1.f - a) * eps[i];
  }
}

__global__ void timestep_embedding_kernel(const int64_t* __restrict__ t, float* __restrict__ out, int B, int dim, int flip,
                                          float shift, float max_period) {
  int half = dim / 2;
  int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= B * half) return;
  int b = idx / half, i = idx - b * half;
  // fp32 arithmetic in the same order as diffusers get_timestep_embedding
  float expo = (-logf(max_period) * (float)i) / ((float)half - shift);
  float arg = (float)t[b] * expf(expo);
  float s = sinf(arg), c = cosf(arg);
  float* o = out + (long)b * dim;
  if (flip) {
    o[i] = c;
    o[half + i] = s;
  } else {
    o[i] = s;
    o[half + i] = c;
  }
  if ((dim & 1) && i == 0) o[dim - 1] = 0.f;
}

__global__ void concat_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long pixels,
                              int C1, int C2) {
  int Ct = (C1 + C2) >> 2, c14 = C1 >> 2, c24 = C2 >> 2;
  long total = pixels * Ct;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long p = i / Ct;
    int c = (int)(i - p * Ct);
    f32x4 v = c < c14 ? reinterpret_cast<const f32x4*>(a)[p * c14 + c] : reinterpret_cast<const f32x4*>(b)[p * c24 + (c - c14)];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}
__global__ void split_kernel(const float* __restrict__ d, float* __restrict__ da, float* __restrict__ db, long pixels, int C1,
                             int C2) {
  int Ct = (C1 + C2) >> 2, c14 = C1 >> 2, c24 = C2 >> 2;
  long total = pixels * Ct;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long p = i / Ct;
    int c = (int)(i - p * Ct);
    f32x4 v = reinterpret_cast<const f32x4*>(d)[i];
    if (c < c14)
      reinterpret_cast<f32x4*>(da)[p * c14 + c] = v;
    else
      reinterpret_cast<f32x4*>(db)[p * c24 + (c - c14)] = v;
  }
}

// layout change through a 32x33 LDS tile: both sides coalesced
__global__ void transpose_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int cols) {
  // x: [batch][rows][cols] -> y: [batch][cols][rows]
  __shared__ float tile[32][33];
  long boff = (long)blockIdx.z * rows * cols;
  int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int j = ty; j < 32; j += 8) {
    int r = r0 + j, c = c0 + tx;
    tile[j][tx] = (r < rows && c < cols) ? x[boff + (long)r * cols + c] : 0.f;
  }
  __syncthreads();
  for (int j = ty; j < 32; j += 8) {
    int c = c0 + j, r = r0 + tx;
    if (r < rows && c < cols) y[boff + (long)c * rows + r] = tile[tx][j];
  }
}

__global__ void upsample2x_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int H, int W, int C4) {
  long total = (long)B * H * W * C4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C4);
    long p = i / C4;
    int w = (int)(p % W);
    long q = p / W;
    int h = (int)(q % H), b = (int)(q / H);
    const f32x4* s = reinterpret_cast<const f32x4*>(dy) + (((long)b * 2 * H + 2 * h) * 2 * W + 2 * w) * C4 + c;
    f32x4 v = s[0] + s[C4] + s[(long)2 * W * C4] + s[(long)2 * W * C4 + C4];
    reinterpret_cast<f32x4*>(dx)[i] = v;
  }
}

// ---- block reduction helper ----
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  int w = threadIdx.x >> 6, l = threadIdx.x & 63;
  if (l == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0)
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) r += sh[k];
  return r;  // valid on thread 0
}

__global__ void mse_part_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ d,
                                float* __restrict__ part, long n, float gsc) {
  __shared__ float sh[8];
  float acc = 0.f;
  EW_LOOP_VEC(n) {
    f32x4 u = reinterpret_cast<const f32x4*>(a)[i] - reinterpret_cast<const f32x4*>(b)[i];
    acc += u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3];
    reinterpret_cast<f32x4*>(d)[i] = u * gsc;
  }
  EW_LOOP_TAIL(n) {
    float u = a[i] - b[i];
    acc += u * u;
    d[i] = u * gsc;
  }
  float r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ void sumsq_part_kernel(const float* __restrict__ g, float* __restrict__ part, long n) {
  __shared__ float sh[8];
  float acc = 0.f;
  EW_LOOP_VEC(n) {
    f32x4 u = reinterpret_cast<const f32x4*>(g)[i];
    acc += u[0] * u[0] + u[1] * u[1] + u[2] * u[2] + u[3] * u[3];
  }
  EW_LOOP_TAIL(n) acc += g[i] * g[i];
  float r = block_sum(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = r;
}
__global__ void final_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int nparts, float scale) {
  __shared__ float sh[8];
  float acc = 0.f;
  for (int i = threadIdx.x; i < nparts; i += blockDim.x) acc += part[i];
  float r = block_sum(acc, sh);
  if (threadIdx.x == 0) out[0] = r * scale;
}

// ---- row softmax, one wave per row ----
// in-place launches (p == s, ds == dp) are part of the contract: no restrict on the aliased pairs
// 3x3 convolution weights of a flat parameter buffer, all at once, into the layout that turns a data gradient into a
// FORWARD convolution: dst[ci][2-r][2-s][co] = src[co][r][s][ci] (180-degree rotation + channel transpose).  One
// workgroup per (weight, 32 co x 32 ci tile): the nine taps go through a padded LDS tile so that both the reads (ci
// contiguous) and the writes (co contiguous) are whole 128-B lines.  table[t] = {offset of the weight in both buffers,
// Cout, Cin, co0, ci0}.
__global__ __launch_bounds__(256) void rotate_conv3x3_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                             const int64_t* __restrict__ table) {
  __shared__ float tile[32][33];
  const int64_t* e = table + 5 * (long)blockIdx.x;
  const long off = e[0];
  const int Cout = (int)e[1], Cin = (int)e[2], co0 = (int)e[3], ci0 = (int)e[4];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
  const float* S = src + off;
  float* D = dst + off;
  for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int co = co0 + ty + 8 * r, ci = ci0 + tx;
      tile[ty + 8 * r][tx] = (co < Cout && ci < Cin) ? S[((long)co * 9 + tap) * Cin + ci] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ci = ci0 + ty + 8 * r, co = co0 + tx;
      if (ci < Cin && co < Cout) D[((long)ci * 9 + (8 - tap)) * Cout + co] = tile[tx][ty + 8 * r];
    }
    __syncthreads();
  }
}

__global__ void softmax_fwd_kernel(const float* s, float* p, long rows, int n, float scale) {
  long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  int lane = threadIdx.x & 63;
  const float* sr = s + row * n;
  float* pr = p + row * n;
  float mx = -INFINITY;
  for (int i = lane; i < n; i += 64) mx = fmaxf(mx, sr[i] * scale);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int i = lane; i < n; i += 64) sum += expf(sr[i] * scale - mx);
  sum = wave_sum(sum);
  float inv = 1.f / sum;
  for (int i = lane; i < n; i += 64) pr[i] = expf(sr[i] * scale - mx) * inv;
}
__global__ void softmax_bwd_kernel(const float* __restrict__ p, const float* dp, float* ds, long rows,
                                   int n, float scale) {
  long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  int lane = threadIdx.x & 63;
  const float* pr = p + row * n;
  const float* dr = dp + row * n;
  float dot = 0.f;
  for (int i = lane; i < n; i += 64) dot += pr[i] * dr[i];
  dot = wave_sum(dot);
  for (int i = lane; i < n; i += 64) ds[row * n + i] = scale * pr[i] * (dr[i] - dot);
}

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int gad_silu_fwd(const float* x, float* y, int64_t n, void* stream) {
  GAD_CHECK(x && y && n >= 0 && gad_aligned16(x) && gad_aligned16(y), "gad_silu_fwd: bad args");
  hipLaunchKernelGGL(silu_fwd_kernel, ew_grid(n / 4), dim3(NT), 0, ST, x, y, (long)n);
  GAD_LAUNCH_CHECK("gad_silu_fwd");
  return 0;
}
extern "C" int gad_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream) {
  GAD_CHECK(x && dy && dx && gad_aligned16(x) && gad_aligned16(dy) && gad_aligned16(dx), "gad_silu_bwd: bad args");
  hipLaunchKernelGGL(silu_bwd_kernel, ew_grid(n / 4), dim3(NT), 0, ST, x, dy, dx, (long)n);
  GAD_LAUNCH_CHECK("gad_silu_bwd");
  return 0;
}
extern "C" int gad_ddim_step(const float* x, const float* eps, float* x_prev, int64_t n, float alpha_t, float alpha_prev,
                             float clip, void* stream) {
  GAD_CHECK(x && eps && x_prev && gad_aligned16(x) && gad_aligned16(eps) && gad_aligned16(x_prev), "gad_ddim_step: bad args");
  GAD_CHECK(alpha_t > 0.f && alpha_t <= 1.f && alpha_prev > 0.f && alpha_prev <= 1.f, "gad_ddim_step: alphas out of (0,1]");
  float sa = sqrtf(alpha_t), sb = sqrtf(1.f - alpha_t), spa = sqrtf(alpha_prev), spb = sqrtf(1.f - alpha_prev);
  hipLaunchKernelGGL(ddim_step_kernel, ew_grid(n / 4), dim3(NT), 0, ST, x, eps, x_prev, (long)n, sa, sb, spa, spb, clip);
  GAD_LAUNCH_CHECK("gad_ddim_step");
  return 0;
}
extern "C" int gad_cfg_ddim_step(const float* x, const float* eps_uc, float* x_prev, int64_t n, float guidance, float alpha_t,
                                 float alpha_prev, float clip, void* stream) {
  GAD_CHECK(x && eps_uc && x_prev && n > 0 && n % 4 == 0 && gad_aligned16(x) && gad_aligned16(eps_uc) && gad_aligned16(x_prev),
            "gad_cfg_ddim_step: bad args (n must be a multiple of 4)");
  GAD_CHECK(alpha_t > 0.f && alpha_t <= 1.f && alpha_prev > 0.f && alpha_prev <= 1.f, "gad_cfg_ddim_step: alphas out of (0,1]");
  hipLaunchKernelGGL(cfg_ddim_step_kernel, ew_grid(n / 4), dim3(NT), 0, ST, x, eps_uc, eps_uc + n, x_prev, (long)n, guidance,
                     sqrtf(alpha_t), sqrtf(1.f - alpha_t), sqrtf(alpha_prev), sqrtf(1.f - alpha_prev), clip);
  GAD_LAUNCH_CHECK("gad_cfg_ddim_step");
  return 0;
}
extern "C" int gad_to_image01(const float* x, float* y, int64_t n, void* stream) {
  GAD_CHECK(x && y && gad_aligned16(x) && gad_aligned16(y), "gad_to_image01: bad args");
  hipLaunchKernelGGL(to_image01_kernel, ew_grid(n / 4), dim3(NT), 0, ST, x, y, (long)n);
  GAD_LAUNCH_CHECK("gad_to_image01");
  return 0;
}
extern "C" int gad_add_noise(const float* x0, const float* eps, const int64_t* t, const float* ac, float* xt, int32_t B,
                             int64_t per, void* stream) {
  GAD_CHECK(x0 && eps && t && ac && xt && B > 0 && per > 0, "gad_add_noise: bad args");
  hipLaunchKernelGGL(add_noise_kernel, ew_grid((int64_t)B * per), dim3(NT), 0, ST, x0, eps, t, ac, xt, B, (long)per);
  GAD_LAUNCH_CHECK("gad_add_noise");
  return 0;
}
extern "C" int gad_timestep_embedding(const int64_t* t, float* out, int32_t B, int32_t dim, int32_t flip, float shift,
                                      float max_period, void* stream) {
  GAD_CHECK(t && out && B > 0 && dim >= 2, "gad_timestep_embedding: bad args");
  int n = B * (dim / 2);
  hipLaunchKernelGGL(timestep_embedding_kernel, dim3((n + NT - 1) / NT), dim3(NT), 0, ST, t, out, B, dim, flip, shift, max_period);
  GAD_LAUNCH_CHECK("gad_timestep_embedding");
  return 0;
}
extern "C" int gad_concat_channels(const float* a, const float* b, float* out, int64_t pixels, int32_t C1, int32_t C2, void* stream) {
  GAD_CHECK(a && b && out && C1 % 4 == 0 && C2 % 4 == 0 && gad_aligned16(a) && gad_aligned16(b) && gad_aligned16(out), "gad_concat_channels: needs C%%4==0 and 16-B alignment");
  hipLaunchKernelGGL(concat_kernel, ew_grid(pixels * ((C1 + C2) / 4)), dim3(NT), 0, ST, a, b, out, (long)pixels, C1, C2);
  GAD_LAUNCH_CHECK("gad_concat_channels");
  return 0;
}
extern "C" int gad_split_channels(const float* d, float* da, float* db, int64_t pixels, int32_t C1, int32_t C2, void* stream) {
  GAD_CHECK(d && da && db && C1 % 4 == 0 && C2 % 4 == 0 && gad_aligned16(d) && gad_aligned16(da) && gad_aligned16(db), "gad_split_channels: needs C%%4==0 and 16-B alignment");
  hipLaunchKernelGGL(split_kernel, ew_grid(pixels * ((C1 + C2) / 4)), dim3(NT), 0, ST, d, da, db, (long)pixels, C1, C2);
  GAD_LAUNCH_CHECK("gad_split_channels");
  return 0;
}
static int transpose_launch(const float* x, float* y, int batch, int rows, int cols, hipStream_t st) {
  dim3 grid((cols + 31) / 32, (rows + 31) / 32, batch);
  hipLaunchKernelGGL(transpose_kernel, grid, dim3(NT), 0, st, x, y, rows, cols);
  return 0;
}
extern "C" int gad_nchw_to_nhwc(const float* x, float* y, int32_t B, int32_t C, int32_t HW, void* stream) {
  GAD_CHECK(x && y && B > 0 && C > 0 && HW > 0 && B < 65536, "gad_nchw_to_nhwc: bad args");
  transpose_launch(x, y, B, C, HW, ST);
  GAD_LAUNCH_CHECK("gad_nchw_to_nhwc");
  return 0;
}
extern "C" int gad_nhwc_to_nchw(const float* x, float* y, int32_t B, int32_t C, int32_t HW, void* stream) {
  GAD_CHECK(x && y && B > 0 && C > 0 && HW > 0 && B < 65536, "gad_nhwc_to_nchw: bad args");
  transpose_launch(x, y, B, HW, C, ST);
  GAD_LAUNCH_CHECK("gad_nhwc_to_nchw");
  return 0;
}
extern "C" int gad_upsample2x_bwd(const float* dy, float* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  GAD_CHECK(dy && dx && C % 4 == 0 && gad_aligned16(dy) && gad_aligned16(dx), "gad_upsample2x_bwd: needs C%%4==0 and 16-B alignment");
  hipLaunchKernelGGL(upsample2x_bwd_kernel, ew_grid((int64_t)B * H * W * (C / 4)), dim3(NT), 0, ST, dy, dx, B, H, W, C / 4);
  GAD_LAUNCH_CHECK("gad_upsample2x_bwd");
  return 0;
}
extern "C" int gad_colsum(const float* dy, float* out, int32_t S, int64_t M, int32_t N, void* ws, int64_t ws_bytes,
                          void* stream) {
  GAD_CHECK(dy && out && S > 0 && S < 65536 && M > 0 && N > 0, "gad_colsum: bad args");
  GAD_CHECK(ws && ws_bytes >= gad_reduce::ws_bytes(S, M, N), "gad_colsum: workspace too small");
  gad_reduce::launch(dy, out, nullptr, S, M, N, (float*)ws, ST);
  GAD_LAUNCH_CHECK("gad_colsum");
  return 0;
}
extern "C" int gad_mse_fwd_bwd(const float* a, const float* b, float* loss, float* d, int64_t n, float gscale, void* ws,
                               int64_t ws_bytes, void* stream) {
  GAD_CHECK(a && b && loss && d && n > 0 && gad_aligned16(a) && gad_aligned16(b) && gad_aligned16(d), "gad_mse_fwd_bwd: bad args");
  dim3 grid = ew_grid(n / 4);
  GAD_CHECK(ws && ws_bytes >= (int64_t)grid.x * 4, "gad_mse_fwd_bwd: workspace too small");
  hipLaunchKernelGGL(mse_part_kernel, grid, dim3(NT), 0, ST, a, b, d, (float*)ws, (long)n, 2.f * gscale / (float)n);
  GAD_LAUNCH_CHECK("gad_mse_fwd_bwd(part)");
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NT), 0, ST, (const float*)ws, loss, (int)grid.x, 1.f / (float)n);
  GAD_LAUNCH_CHECK("gad_mse_fwd_bwd(final)");
  return 0;
}
extern "C" int gad_sumsq(const float* g, float* out, int64_t n, void* ws, int64_t ws_bytes, void* stream) {
  GAD_CHECK(g && out && n > 0 && gad_aligned16(g), "gad_sumsq: bad args");
  dim3 grid = ew_grid(n / 4);
  GAD_CHECK(ws && ws_bytes >= (int64_t)grid.x * 4, "gad_sumsq: workspace too small");
  hipLaunchKernelGGL(sumsq_part_kernel, grid, dim3(NT), 0, ST, g, (float*)ws, (long)n);
  GAD_LAUNCH_CHECK("gad_sumsq(part)");
  hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(NT), 0, ST, (const float*)ws, out, (int)grid.x, 1.f);
  GAD_LAUNCH_CHECK("gad_sumsq(final)");
  return 0;
}
extern "C" int gad_softmax_fwd(const float* s, float* p, int64_t rows, int32_t n, float scale, void* stream) {
  GAD_CHECK(s && p && rows > 0 && n > 0, "gad_softmax_fwd: bad args");
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3((unsigned)gad_ceil_div(rows, 4)), dim3(NT), 0, ST, s, p, (long)rows, n, scale);
  GAD_LAUNCH_CHECK("gad_softmax_fwd");
  return 0;
}
extern "C" int gad_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int32_t n, float scale, void* stream) {
  GAD_CHECK(p && dp && ds && rows > 0 && n > 0, "gad_softmax_bwd: bad args");
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)gad_ceil_div(rows, 4)), dim3(NT), 0, ST, p, dp, ds, (long)rows, n, scale);
  GAD_LAUNCH_CHECK("gad_softmax_bwd");
  return 0;
}

extern "C" int gad_rotate_conv3x3(const float* src, float* dst, const int64_t* table, int32_t n_tiles, void* stream) {
  GAD_CHECK(src && dst && table && n_tiles > 0, "gad_rotate_conv3x3: null pointer or no tiles");
  GAD_CHECK(src != dst, "gad_rotate_conv3x3: in place is not supported");
  hipLaunchKernelGGL(rotate_conv3x3_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, src, dst, table);
  GAD_LAUNCH_CHECK("gad_rotate_conv3x3");
  return 0;
}
