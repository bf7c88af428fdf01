// Sustained f32-MFMA rate of this GPU with NO memory traffic: the ceiling any fp32 contraction kernel can reach on this
// box (v_mfma_f32_32x32x2_f32: 4096 FLOP per 64 cycles per SIMD = 157.3 TF/s at 2.4 GHz; a lower sustained clock
// lowers it).  usage: mfma_peak [waves_per_simd=2] [iters=20000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  float a = a0 + threadIdx.x * 1e-9f, b = b0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
  }
  float s = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) s += acc[t][e];
  if (s == 123.456f) out[0] = s;
}
int main(int argc, char** argv) {
  int wps = argc > 1 ? atoi(argv[1]) : 2, iters = argc > 2 ? atoi(argv[2]) : 20000;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  float* out;
  hipMalloc(&out, 4);
  dim3 grid(cus * wps), block(256);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, grid, block, 0, 0, out, iters, 1.0f, 1e-6f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)cus * wps * 4 * iters * 16.0 * 4096.0;
    printf("rep %d: %d CUs x %d waves/SIMD, %d iters: %.3f ms  %.1f TFLOP/s  (implied clock %.3f GHz; clockRate attr %.3f GHz)\n", rep, cus, wps,
           iters, ms, flops / ms / 1e9, flops / ms / 1e9 / 157.2864 * 2.4, prop.clockRate / 1e6);
  }
  return 0;
}
