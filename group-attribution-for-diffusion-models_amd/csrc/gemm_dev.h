// Device-side building blocks shared by the contraction kernels of gemm_f32.hip and wino4_fused.hip (gfx950 only):
// launch-argument struct, LDS-DMA staging, the XOR-swizzled KC tile image and its fragment reads, the XCD-aware block remap.
#pragma once
#include "gad_common.h"

namespace gadk {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

// 64 bytes of zeros in device memory: the DMA source of padding / out-of-range tile slots
static __device__ __attribute__((aligned(64))) float g_zero_block[16];

// physical float offset of logical 16-B chunk q (0..7) of KC-tile row r
__device__ __forceinline__ int kc_off(int r, int q) { return r * BK + ((q ^ ((r >> 1) & 7)) << 2); }

// LDS-DMA: 64 lanes x 16 B -> LDS [dst, dst + 1 KiB), lane-linear; src is per lane
__device__ __forceinline__ void glds16(const float* src, float* dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, 0);
}

// A barrier that publishes LDS-DMA'd tiles: the DMA is a VMEM operation (vmcnt), and the compiler's own waitcnt placement
// at __syncthreads() only covers it when it happens to track the LDS side effect - so wait for it explicitly.
// s_waitcnt immediate (gfx9): vmcnt = 0 (bits 3:0 and 15:14), expcnt = 7 (no wait), lgkmcnt = 15 (no wait).
__device__ __forceinline__ void barrier_after_dma() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
}

// division by a launch-invariant divisor: q = (mulhi(n, mul) + n) >> shift, exact for 0 <= n < 2^31
struct FastDiv {
  unsigned mul, shift;
  __device__ __forceinline__ int div(int n) const { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shift); }
};

struct DevArgs {
  FastDiv fdC, fdKW, fdHoWo, fdWo, fdRpg, fdTaps;
  int kperm, taps;   // kperm: K steps visit (channel chunk, tap) instead of (tap, channel chunk)
  const float* A;
  const float* B;
  float* C;
  const unsigned short* Bh;   // bf16 copy of B ([N][ldb], k contiguous): bf16 patch conv streams it by LDS-DMA
  const float* A2;   // two-source conv gather (virtual channel concat): channels >= a_split come from A2
  int a_split, ldx2;
  // K-concatenated dense operands (fused LoRA side path): for k >= k_split the products read Ak2 / Bk2 at k - k_split
  const float* Ak2;
  const float* Bk2;
  int lda2, ldb2, k_split;
  int M, N, K;
  int lda, ldb, ldc;
  int batch_inner;
  long sA0, sA1, sB0, sB1, sC0, sC1;
  gad_conv_geom g;
  float alpha;
  const float* bias;
  const float* rowadd;
  int rows_per_group, ld_rowadd;
  const float* residual;
  int ldr;
  float* ws;
  int tiles_m, tiles_n, splitk, ktiles_per_split;
  int epi_vec;       // the output (and bias / rowadd / residual / workspace) can be written / read as aligned float4
};

__device__ __forceinline__ f32x4 ldg4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 zero4() { return f32x4{0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ f32x4 keep_if(f32x4 v, bool ok) { return ok ? v : zero4(); }

// ------------------------------------------------------------------------------------
// Tile loaders.  Each thread owns NS float4 slots of the tile.
//   load(k0, v, mask): issue the global loads of the K step starting at k0.  Loads are
//     UNCONDITIONAL (an invalid slot reads the tensor's base address) so that all of a
//     thread's loads are in flight together behind the MFMAs of the current step;
//     `mask` bit i says whether slot i is real data.
//   store(lds, v, mask): zero the invalid slots and write the tile into LDS.
// KC-type: slot i = (row (tid>>3)+32 i, float4 column tid&7)
// MC-type: slot i = (k row tid/F4 + (256/F4) i, float4 column tid%F4),  F4 = ROWS/4
// ------------------------------------------------------------------------------------
// the wave's index as a SCALAR: LDS-DMA destinations (M0) derived from it need no per-launch v_readfirstlane in the K loop
__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int ROWS>
struct KCSlots {
  static constexpr int NS = ROWS / 32;
  __device__ static int row(int i) { return (threadIdx.x >> 3) + 32 * i; }
  // logical float4 column (k chunk) this lane supplies for its row: the lane's LDS position is fixed by
  // the DMA (physical chunk = lane & 7), so the swizzle is applied to what it loads
  // (row(i) >> 1) & 7 == (tid >> 4) & 7 for every slot i, so the column is slot independent
  __device__ static int kq4() { return ((threadIdx.x & 7) ^ ((threadIdx.x >> 4) & 7)) * 4; }
  __device__ static float* dma_dst(float* tile, int i) { return tile + (wave_id() * 8 + 32 * i) * BK; }
  __device__ static void store(float* lds, const f32x4* v, unsigned mask) {   // register path (VEC == 1)
#pragma unroll
    for (int i = 0; i < NS; ++i)
      *reinterpret_cast<f32x4*>(lds + row(i) * BK + (threadIdx.x & 7) * 4) = keep_if(v[i], (mask >> i) & 1u);
  }
};
template <int ROWS>
struct MCSlots {
  static constexpr int NS = ROWS / 32;
  static constexpr int F4 = ROWS / 4;
  static constexpr int KSTEP = NTHREADS / F4;
  __device__ static int krow(int i) { return threadIdx.x / F4 + KSTEP * i; }
  __device__ static int rq4() { return (threadIdx.x % F4) * 4; }
  __device__ static float* dma_dst(float* tile, int i) { return tile + (wave_id() * (KSTEP / 4) + KSTEP * i) * ROWS; }
  __device__ static void store(float* lds, const f32x4* v, unsigned mask) {   // register path (VEC == 1)
#pragma unroll
    for (int i = 0; i < NS; ++i)
      *reinterpret_cast<f32x4*>(lds + krow(i) * ROWS + rq4()) = keep_if(v[i], (mask >> i) & 1u);
  }
};

// Loader protocol: prep(k0) once per K step (shared decode), then per slot i either
//   src(i)            -> per-lane source address of the float4 (or the zero block)   [VEC == 4, LDS-DMA]
//   slot(i, v, mask)  -> register-staged scalar gather of the 4 floats               [VEC == 1]
// both called from between MFMA groups so the address arithmetic sits in the MFMA shadow.
// branch-free source select: offsets are always computed (possibly from out-of-range coordinates), masked
// to 0 when invalid and added to either the tensor base or the zero block
__device__ __forceinline__ const float* sel_src(const float* base, long off, bool ok) {
  const float* b = ok ? base : (const float*)g_zero_block;
  return b + (off & -(long)ok);
}


// Workgroup id -> work item, XCD-aware and bijective: hardware deals consecutive workgroup ids round-robin to the 8 XCDs
// (each with its own L2); this hands every XCD a CONTIGUOUS range of work items, so tiles that share an operand panel
// (neighbouring tile_n of one tile_m, the halo rows of neighbouring row tiles) meet in one L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

constexpr int EPI_LD = 40;                       // scratch row stride in floats
constexpr int EPI_WAVE = 32 * EPI_LD;            // floats of scratch per wave (5 KB; 20 KB per workgroup)

// ------------------------------------------------------------------------------------
// Fragment reads.  K step of 32 = 4 groups of 8; in group g, MFMA step j (0..3) feeds
// lane half h with k = 8g + 4h + j  (so a KC tile is read as one b128 per group).
// ------------------------------------------------------------------------------------
template <bool KC, int ROWS>
__device__ __forceinline__ f32x4 read_frag(const float* lds, int row, int g, int h) {
  if (KC) {
    return *reinterpret_cast<const f32x4*>(lds + kc_off(row, 2 * g + h));
  } else {
    const float* p = lds + (8 * g + 4 * h) * ROWS + row;
    return f32x4{p[0], p[ROWS], p[2 * ROWS], p[3 * ROWS]};
  }
}

// dense [row][k] rows by LDS-DMA with a 64-bit step offset (position x panel stride + k): rows beyond the operand clamp
// to the last one (they feed outputs the epilogue never stores)
template <int ROWS>
struct WinoKC : KCSlots<ROWS> {
  using S = KCSlots<ROWS>;
  const float* ptr[S::NS];
  long off;
  __device__ void init(const float* b, int ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < S::NS; ++i) {
      int r = row0 + S::row(i);
      r = r < nrows ? r : nrows - 1;
      ptr[i] = b + (long)r * ld + S::kq4();
    }
    off = 0;
  }
  __device__ __forceinline__ const float* src(int i) const { return ptr[i] + off; }
};

// one direction of the F(4x4, 3x3) input transform: t = B^T d for the standard points {0, +-1, +-2, inf}
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void wino4_bt(const f32x4 (&d)[6], f32x4 (&t)[6]) {
  t[0] = 4.f * d[0] - 5.f * d[2] + d[4];
  t[1] = -4.f * (d[1] + d[2]) + d[3] + d[4];
  t[2] = 4.f * (d[1] - d[2]) - d[3] + d[4];
  t[3] = 2.f * (d[3] - d[1]) - d[2] + d[4];
  t[4] = 2.f * (d[1] - d[3]) - d[2] + d[4];
  t[5] = 4.f * d[1] - 5.f * d[3] + d[5];
}

}  // namespace gadk
