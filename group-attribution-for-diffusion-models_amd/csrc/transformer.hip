// LayerNorm and GEGLU kernels for the Stable-Diffusion transformer blocks (UNet2DConditionModel,
// BasicTransformerBlock: reference text_to_image/train_text_to_image_lora.py:1268-1270 via diffusers).
// HBM-bound: LayerNorm 8 B/elem fwd, 16 B/elem bwd; GEGLU 12 B per output elem fwd.
// One wave per row (C <= 2048 floats held in registers: 8 float4 per lane), wavefront shuffles for the
// row moments, no LDS; parameter gradients are per-workgroup partial sums reduced by gad_reduce.
#include "gad_common.h"
#include "gad_reduce.h"

namespace {

constexpr int NT = 256, WPB = 4, MAXV = 8;   // waves per block; float4 per lane (C <= 64*4*8 = 2048)

__global__ __launch_bounds__(NT) void ln_fwd_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    float* __restrict__ mean, float* __restrict__ rstd, long rows, int C, float eps) {
  const int lane = threadIdx.x & 63, C4 = C >> 2;
  for (long row = (long)blockIdx.x * WPB + (threadIdx.x >> 6); row < rows; row += (long)gridDim.x * WPB) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * C);
    f32x4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      v[i] = c < C4 ? xr[c] : f32x4{0, 0, 0, 0};
      s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
    }
    float mu = wave_sum(s) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (lane + 64 * i < C4) {
        f32x4 d = v[i] - mu;
        q += d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + d[3] * d[3];
      }
    float rs = rsqrtf(wave_sum(q) / (float)C + eps);
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    f32x4* yr = reinterpret_cast<f32x4*>(y + row * C);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      if (c < C4) yr[c] = (v[i] - mu) * rs * reinterpret_cast<const f32x4*>(gamma)[c] + reinterpret_cast<const f32x4*>(beta)[c];
    }
  }
}

// dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma; per-block partial dgamma/dbeta
__global__ __launch_bounds__(NT) void ln_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                    float* __restrict__ dx, const float* __restrict__ gamma,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    float* __restrict__ part, long rows, int C, const float* __restrict__ dx_add) {
  __shared__ float red[2 * 2048];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, C4 = C >> 2;
  f32x4 ag[MAXV], ab[MAXV];
#pragma unroll
  for (int i = 0; i < MAXV; ++i) ag[i] = ab[i] = f32x4{0, 0, 0, 0};
  for (long row = (long)blockIdx.x * WPB + wave; row < rows; row += (long)gridDim.x * WPB) {
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + row * C);
    const f32x4* dr = reinterpret_cast<const f32x4*>(dy + row * C);
    float mu = mean[row], rs = rstd[row];
    f32x4 xh[MAXV], g[MAXV];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      if (c < C4) {
        f32x4 d = dr[c];
        xh[i] = (xr[c] - mu) * rs;
        g[i] = d * reinterpret_cast<const f32x4*>(gamma)[c];
        ab[i] += d;
        ag[i] += d * xh[i];
        f32x4 gx = g[i] * xh[i];
        s1 += g[i][0] + g[i][1] + g[i][2] + g[i][3];
        s2 += gx[0] + gx[1] + gx[2] + gx[3];
      } else {
        xh[i] = g[i] = f32x4{0, 0, 0, 0};
      }
    }
    float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
    f32x4* xo = reinterpret_cast<f32x4*>(dx + row * C);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      int c = lane + 64 * i;
      if (c < C4) {
        f32x4 o = (g[i] - m1 - xh[i] * m2) * rs;
        if (dx_add) o += reinterpret_cast<const f32x4*>(dx_add + row * C)[c];      // the residual branch's gradient
        xo[c] = o;
      }
    }
  }
  // combine the 4 waves' per-channel partials -> part[block][2][C]   (rows: 0 = dgamma, 1 = dbeta)
  for (int w = 0; w < WPB; ++w) {
    if (wave == w) {
#pragma unroll
      for (int i = 0; i < MAXV; ++i) {
        int c = lane + 64 * i;
        if (c < C4) {
          float* rg = red + c * 4;
          float* rb = red + 2048 + c * 4;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            rg[e] = (w == 0 ? 0.f : rg[e]) + ag[i][e];
            rb[e] = (w == 0 ? 0.f : rb[e]) + ab[i][e];
          }
        }
      }
    }
    __syncthreads();
  }
  float* o = part + (long)blockIdx.x * 2 * C;
  for (int c = threadIdx.x; c < C; c += NT) {
    o[c] = red[c];
    o[C + c] = red[2048 + c];
  }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
  float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  float pdf = 0.39894228040143268f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// h [M][2F] -> out [M][F] = h[:, :F] * gelu(h[:, F:])
__global__ void geglu_fwd_kernel(const float* __restrict__ h, float* __restrict__ out, long M, int F4) {
  long total = M * F4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long m = i / F4;
    int c = (int)(i - m * F4);
    const f32x4* hr = reinterpret_cast<const f32x4*>(h) + m * 2 * F4;
    f32x4 a = hr[c], g = hr[F4 + c], o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a[e] * gelu_f(g[e]);
    reinterpret_cast<f32x4*>(out)[i] = o;
  }
}
__global__ void geglu_bwd_kernel(const float* __restrict__ h, const float* __restrict__ dout, float* __restrict__ dh, long M, int F4) {
  long total = M * F4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long m = i / F4;
    int c = (int)(i - m * F4);
    const f32x4* hr = reinterpret_cast<const f32x4*>(h) + m * 2 * F4;
    f32x4* dr = reinterpret_cast<f32x4*>(dh) + m * 2 * F4;
    f32x4 a = hr[c], g = hr[F4 + c], d = reinterpret_cast<const f32x4*>(dout)[i], da, dg;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      da[e] = d[e] * gelu_f(g[e]);
      dg[e] = d[e] * a[e] * gelu_grad(g[e]);
    }
    dr[c] = da;
    dr[F4 + c] = dg;
  }
}

int ln_blocks(long rows) {
  long b = gad_ceil_div(rows, WPB);
  return (int)(b < 1024 ? b : 1024);
}
}  // namespace

extern "C" int64_t gad_layernorm_workspace_bytes(int64_t rows, int32_t C) {
  long nb = ln_blocks(rows);
  return (nb * 2L * C + gad_reduce::ws_bytes(1, nb, 2 * C) / 4) * 4;
}

extern "C" int gad_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                                 int64_t rows, int32_t C, float eps, void* stream) {
  GAD_CHECK(x && y && gamma && beta && mean && rstd && rows > 0, "gad_layernorm_fwd: bad args");
  GAD_CHECK(C % 4 == 0 && C <= 2048 && gad_aligned16(x) && gad_aligned16(y) && gad_aligned16(gamma) && gad_aligned16(beta),
            "gad_layernorm_fwd: needs C%%4==0, C<=2048, 16-B alignment (C=%d)", C);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(ln_blocks(rows)), dim3(NT), 0, (hipStream_t)stream, x, y, gamma, beta, mean, rstd, (long)rows, C, eps);
  GAD_LAUNCH_CHECK("gad_layernorm_fwd");
  return 0;
}

extern "C" int gad_layernorm_bwd(const float* x, const float* dy, float* dx, const float* dx_add, const float* gamma,
                                 const float* mean, const float* rstd, float* dgamma_dbeta, int64_t rows, int32_t C, void* ws,
                                 int64_t ws_bytes, void* stream) {
  GAD_CHECK(x && dy && dx && gamma && mean && rstd && rows > 0, "gad_layernorm_bwd: bad args");
  GAD_CHECK(C % 4 == 0 && C <= 2048 && gad_aligned16(x) && gad_aligned16(dy) && gad_aligned16(dx) && gad_aligned16(gamma),
            "gad_layernorm_bwd: needs C%%4==0, C<=2048, 16-B alignment (C=%d)", C);
  GAD_CHECK(ws && ws_bytes >= gad_layernorm_workspace_bytes(rows, C), "gad_layernorm_bwd: workspace too small");
  GAD_CHECK(!dx_add || gad_aligned16(dx_add), "gad_layernorm_bwd: dx_add must be 16-byte aligned");
  int nb = ln_blocks(rows);
  float* part = (float*)ws;
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(nb), dim3(NT), 0, (hipStream_t)stream, x, dy, dx, gamma, mean, rstd, part, (long)rows, C, dx_add);
  GAD_LAUNCH_CHECK("gad_layernorm_bwd");
  if (dgamma_dbeta) {      // NULL: frozen affine parameters (LoRA training) - no reduction of the partials
    gad_reduce::launch(part, dgamma_dbeta, nullptr, 1, nb, 2 * C, part + (long)nb * 2 * C, (hipStream_t)stream);   // [2C] = dgamma | dbeta
    GAD_LAUNCH_CHECK("gad_layernorm_bwd(reduce)");
  }
  return 0;
}

extern "C" int gad_geglu_fwd(const float* h, float* out, int64_t M, int32_t F, void* stream) {
  GAD_CHECK(h && out && M > 0 && F > 0 && F % 4 == 0 && gad_aligned16(h) && gad_aligned16(out), "gad_geglu_fwd: needs F%%4==0 and 16-B alignment");
  long nv = M * (F / 4);
  long b = gad_ceil_div(nv, 256);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, (hipStream_t)stream, h, out, (long)M, F / 4);
  GAD_LAUNCH_CHECK("gad_geglu_fwd");
  return 0;
}
extern "C" int gad_geglu_bwd(const float* h, const float* dout, float* dh, int64_t M, int32_t F, void* stream) {
  GAD_CHECK(h && dout && dh && M > 0 && F > 0 && F % 4 == 0 && gad_aligned16(h) && gad_aligned16(dout) && gad_aligned16(dh), "gad_geglu_bwd: needs F%%4==0 and 16-B alignment");
  long nv = M * (F / 4);
  long b = gad_ceil_div(nv, 256);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3((unsigned)(b < 2048 ? b : 2048)), dim3(256), 0, (hipStream_t)stream, h, dout, dh, (long)M, F / 4);
  GAD_LAUNCH_CHECK("gad_geglu_bwd");
  return 0;
}
