// Where do the MFMA cycles of the patch-convolution inner loop go?  A synthetic loop with the same instruction mix
// (per wave and K step of 32: 64 x v_mfma_f32_32x32x2_f32 on 4 accumulator tiles, 4 x (1 A + 4 B) ds_read_b128 fragments,
// optional workgroup barrier, optional 4 x global_load_lds_dwordx4 per thread) and 2 workgroups of 4 waves per CU.
// usage: mfma_loop_model [ksteps=2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <bool READS, bool BARRIER, bool DMA>
__global__ __launch_bounds__(256) void loop_kernel(float* out, const float* src, int ksteps) {
  __shared__ __attribute__((aligned(16))) float lds[15616];   // 61 KB: two workgroups per CU, as the real kernel
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 15616; i += 256) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const float* pa = lds + (lane & 31) * 36 + (lane >> 5) * 4 + wave * 1152;     // 144-B pixel stride, like the patch
  const float* pb = lds + 7424 + (lane & 31) * 32 + (lane >> 5) * 4;            // weight tile rows
  f32x4 fa = {1.f, 1.f, 1.f, 1.f}, fb[4];
  for (int j = 0; j < 4; ++j) fb[j] = f32x4{1e-6f, 1e-6f, 1e-6f, 1e-6f};
  const float* g = src + (size_t)(blockIdx.x & 63) * 4096 + tid * 4;
  for (int ks = 0; ks < ksteps; ++ks) {
    float* nb = lds + 7424 + ((ks + 1) & 1) * 4096;
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
      if (READS) {
        fa = *reinterpret_cast<const f32x4*>(pa + 8 * grp);
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const f32x4*>(pb + ((ks & 1) * 4096) + j * 1024 + 8 * grp);
      }
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[s], fb[j][s], acc[j], 0, 0, 0);
        if (DMA && s == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + grp * 1024),
                                           (__attribute__((address_space(3))) void*)(nb + wave * 256 + grp * 1024), 16, 0, 0);
      }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BARRIER) __syncthreads();
  }
  float sum = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) sum += acc[t][e];
  if (sum == 123.456f) out[0] = sum;
}

template <bool R, bool B, bool D>
static void run(const char* name, float* out, const float* src, int ksteps, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((loop_kernel<R, B, D>), dim3(cus * 2), dim3(256), 0, 0, out, src, ksteps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  double flops = (double)cus * 2 * 4 * ksteps * 64.0 * 4096.0;
  printf("%-44s %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.2864);
}

// PIPE = 1: fragments of group g+1 are read during group g (as the real kernel does); the first group of a K step is read
//           right after the barrier (its latency is exposed when both workgroups of a CU arrive together).
// PIPE = 2: the first group of the NEXT K step is read before the barrier too (needs a third weight stage in the real kernel).
template <int PIPE, bool BARRIER, bool DMA>
__global__ __launch_bounds__(256) void pipe_kernel(float* out, const float* src, int ksteps) {
  __shared__ __attribute__((aligned(16))) float lds[15616];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 15616; i += 256) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[4];
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
  const float* pa = lds + (lane & 31) * 36 + (lane >> 5) * 4 + wave * 1152;
  const float* pb = lds + 7424 + (lane & 31) * 32 + (lane >> 5) * 4;
  f32x4 fa[2], fb[2][4], stg[4];
  int junk[4] = {lane, lane + 1, lane + 2, lane + 3};
  int ujunk[4] = {ksteps, ksteps + 1, ksteps + 2, ksteps + 3};
  auto rd = [&](int buf, int ks, int grp) {
    fa[buf] = *reinterpret_cast<const f32x4*>(pa + 8 * grp);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[buf][j] = *reinterpret_cast<const f32x4*>(pb + ((ks & 1) * 4096) + j * 1024 + 8 * grp);
  };
  const float* g = src + (size_t)(blockIdx.x & 63) * 4096 + tid * 4;
  rd(0, 0, 0);
  for (int ks = 0; ks < ksteps; ++ks) {
    float* nb = lds + 7424 + ((ks + 1) & 1) * 4096;
    if (PIPE != 2 && ks > 0) rd(0, ks, 0);   // (PIPE >= 10 behaves as PIPE 1)
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[grp & 1][s], fb[grp & 1][j][s], acc[j], 0, 0, 0);
        if (DMA && grp == 0) {    // the real kernel issues its 4 weight slots during the first MFMA group
          if (PIPE == 5) stg[s] = *reinterpret_cast<const f32x4*>(g + s * 1024);
          else
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + s * 1024),
                                             (__attribute__((address_space(3))) void*)(nb + wave * 256 + s * 1024), 16, 0, 0);
        }
        if (DMA && PIPE == 5 && grp == 3)
          *reinterpret_cast<f32x4*>(nb + wave * 256 + s * 1024 + lane * 4) = stg[s];
        if (s == 1) {
          if (grp < 3) rd((grp + 1) & 1, ks, grp + 1);
          else if (PIPE == 2) rd(0, ks + 1, 0);
        }
        if (PIPE >= 40) {           // PIPE - 40 scalar (wave-uniform) integer operations per MFMA quartet
#pragma unroll
          for (int u = 0; u < PIPE - 40; ++u) ujunk[u & 3] = (ujunk[u & 3] ^ ks) + u + grp;
        } else if (PIPE >= 10) {    // PIPE - 10 dependent-free integer VALU operations per MFMA quartet (x16 per K step)
#pragma unroll
          for (int u = 0; u < PIPE - 10; ++u) junk[u & 3] = (junk[u & 3] ^ lane) + u;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (DMA && PIPE != 3 && PIPE != 5) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BARRIER) __syncthreads();
  }
  float sum = 0.f;
  for (int t = 0; t < 4; ++t)
    for (int e = 0; e < 16; ++e) sum += acc[t][e];
  if (sum == 123.456f || junk[0] + junk[1] + junk[2] + junk[3] == 12345 || ujunk[0] + ujunk[1] + ujunk[2] + ujunk[3] == 54321) out[0] = sum;
}

template <int PIPE, bool B, bool D>
static void runp(const char* name, float* out, const float* src, int ksteps, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((pipe_kernel<PIPE, B, D>), dim3(cus * 2), dim3(256), 0, 0, out, src, ksteps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  double flops = (double)cus * 2 * 4 * ksteps * 64.0 * 4096.0;
  printf("%-60s %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.2864);
}

// The same loop with 64 pixels x 128 channels per wave (a 256-pixel tile): 8 accumulator tiles, 2 A + 4 B fragments per
// group of 32 MFMAs, the same 16 KB of weights and one barrier per K step - half the barriers / DMA / B reads per MFMA.
template <bool BARRIER, bool DMA>
__global__ __launch_bounds__(256, 2) void wide_kernel(float* out, const float* src, int ksteps) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // 80 KB: patch 48 KB + 2 x 16 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 20480; i += 256) lds[i] = 1e-6f * i;
  __syncthreads();
  f32x16 acc[2][4];
  for (int i = 0; i < 2; ++i)
    for (int t = 0; t < 4; ++t)
      for (int e = 0; e < 16; ++e) acc[i][t][e] = 0.f;
  const float* pa = lds + (lane & 31) * 36 + (lane >> 5) * 4 + wave * 2304;
  const float* pb = lds + 12288 + (lane & 31) * 32 + (lane >> 5) * 4;
  f32x4 fa[2][2], fb[2][4];
  auto rd = [&](int buf, int ks, int grp) {
    fa[buf][0] = *reinterpret_cast<const f32x4*>(pa + 8 * grp);
    fa[buf][1] = *reinterpret_cast<const f32x4*>(pa + 1152 + 8 * grp);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[buf][j] = *reinterpret_cast<const f32x4*>(pb + ((ks & 1) * 4096) + j * 1024 + 8 * grp);
  };
  const float* g = src + (size_t)(blockIdx.x & 63) * 4096 + tid * 4;
  rd(0, 0, 0);
  for (int ks = 0; ks < ksteps; ++ks) {
    float* nb = lds + 12288 + ((ks + 1) & 1) * 4096;
    if (ks > 0) rd(0, ks, 0);
#pragma unroll
    for (int grp = 0; grp < 4; ++grp) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[grp & 1][i][s], fb[grp & 1][j][s], acc[i][j], 0, 0, 0);
        if (DMA && grp == 0)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + s * 1024),
                                           (__attribute__((address_space(3))) void*)(nb + wave * 256 + s * 1024), 16, 0, 0);
        if (s == 1 && grp < 3) rd((grp + 1) & 1, ks, grp + 1);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (DMA) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (BARRIER) __syncthreads();
  }
  float sum = 0.f;
  for (int i = 0; i < 2; ++i)
    for (int t = 0; t < 4; ++t)
      for (int e = 0; e < 16; ++e) sum += acc[i][t][e];
  if (sum == 123.456f) out[0] = sum;
}

template <bool B, bool D>
static void runw(const char* name, float* out, const float* src, int ksteps, int cus) {
  hipFuncSetAttribute((const void*)wide_kernel<B, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 81920);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((wide_kernel<B, D>), dim3(cus * 2), dim3(256), 81920, 0, out, src, ksteps);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  double flops = (double)cus * 2 * 4 * ksteps * 128.0 * 4096.0;
  printf("%-60s %8.3f ms  %6.1f TFLOP/s  (%.3f of 157.3)\n", name, best, flops / best / 1e9, flops / best / 1e9 / 157.2864);
}

int main(int argc, char** argv) {
  int ksteps = argc > 1 ? atoi(argv[1]) : 2000;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  int cus = prop.multiProcessorCount;
  float *out, *src;
  hipMalloc(&out, 4);
  hipMalloc(&src, 64 * 4096 * 4 + 65536);
  hipMemset(src, 0, 64 * 4096 * 4 + 65536);
  run<false, false, false>("MFMA only", out, src, ksteps, cus);
  run<true, false, false>("+ fragment ds_read_b128 (5 per 16 MFMA)", out, src, ksteps, cus);
  run<false, true, false>("+ barrier per K step (no reads)", out, src, ksteps, cus);
  run<true, true, false>("+ reads + barrier", out, src, ksteps, cus);
  run<true, true, true>("+ reads + barrier + LDS-DMA (16 KB / K step)", out, src, ksteps, cus);
  run<true, false, true>("+ reads + LDS-DMA, no barrier", out, src, ksteps, cus);
  runp<1, false, false>("pipelined fragment reads, no barrier", out, src, ksteps, cus);
  runp<1, true, false>("pipelined reads + barrier (first group after the barrier)", out, src, ksteps, cus);
  runp<1, true, true>("pipelined reads + barrier + DMA  [= the real kernel]", out, src, ksteps, cus);
  runp<2, true, false>("first group of the next step read BEFORE the barrier", out, src, ksteps, cus);
  runp<2, true, true>("  ... + DMA", out, src, ksteps, cus);
  runp<3, true, true>("real-kernel structure, DMA never waited for (timing only)", out, src, ksteps, cus);
  runp<5, true, true>("real-kernel structure, weights staged through registers + ds_write", out, src, ksteps, cus);
  runp<12, true, true>("real-kernel structure + 32 x 2 full-rate integer VALU (xor, add) per K step", out, src, ksteps, cus);
  runp<16, true, true>("real-kernel structure + 96 x 2 full-rate VALU per K step", out, src, ksteps, cus);
  runp<22, true, true>("real-kernel structure + 192 x 2 full-rate VALU per K step", out, src, ksteps, cus);
  runp<44, true, true>("real-kernel structure + 64 x 3 scalar integer ops per K step", out, src, ksteps, cus);
  runp<48, true, true>("real-kernel structure + 128 x 3 scalar integer ops per K step", out, src, ksteps, cus);
  runw<true, false>("256-pixel tile (64 x 128 per wave): reads + barrier", out, src, ksteps / 2, cus);
  runw<true, true>("256-pixel tile: reads + barrier + DMA", out, src, ksteps / 2, cus);
  return 0;
}
