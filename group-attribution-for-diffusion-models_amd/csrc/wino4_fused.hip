// Winograd F(4x4, 3x3) products with the WHOLE output transform kept on chip (gfx950).
//
// The three-launch route of gemm_f32.hip (wino4_input_kernel -> wino4_gemm_kernel -> wino4_output_kernel) sends the
// half-transformed products Mh [24][tiles][Cout] out to HBM and back: 3 of the 9.5 tensor-sized streams of a convolution.
// Here one workgroup owns a (64 tiles x 64 output channels) block for ALL 36 Winograd positions and never writes a product:
//
//   for xi in 0..5:                                   // rows of the 6 x 6 position grid
//     for nu in 0..5:                                 // position (xi, nu): one [64 x Cin] x [Cin x 64] product, K loop over Cin
//       acc  = V[6 xi + nu] U[6 xi + nu]^T
//       Y[j] += A^T[j][nu] acc          (j = 0..3)    // nu direction of y = A^T M A, folded when the position is done
//     O[i][j] += A^T[i][xi] Y[j]        (i = 0..3)    // xi direction, folded when the row is done
//   y[tile pixel (i, j)] = alpha O[i][j] + bias + time-embedding row + residual        (float4 stores through an LDS transpose)
//
// 16 + 4 + 2 accumulator blocks of 32 x 32 per wave = 352 registers of the 512 a lone wave per SIMD may hold, so the
// kernel runs ONE workgroup (4 waves, 2 x 2) per CU and hides memory latency with a ring of NST LDS stages filled by
// LDS-DMA NST - 1 stage-steps ahead (counted s_waitcnt vmcnt, one barrier per stage-step) instead of with a second workgroup.
// The position that just finished is folded while the next one multiplies: positions alternate between two accumulators, and
// the fold of the idle one is dealt over the 16 MFMA gaps of the next position's first K step (vector ALU in the matrix
// pipe's shadow).  V [36][T][Cin] (wino4_input_kernel) and U [36][Cout][Cin] (wino4_weights_kernel) are k-contiguous rows:
// the lean WinoKC loaders, XOR-swizzled KC tiles, ds_read_b128 fragments - the generic engine's K step.
// HBM traffic of the convolution: x once + V once out and once back + y once = 6.5 tensor sizes instead of 9.5;
// executed MFMA work 36/144 of the direct form's.  fp32 throughout, deterministic.
#include <type_traits>

#include "gemm_dev.h"

using namespace gadk;

namespace {

template <int N> struct IC : std::integral_constant<int, N> {};

// counted wait for LDS-DMA / loads: all but the N youngest vector-memory operations of this wave are done
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
  __builtin_amdgcn_s_waitcnt((N & 15) | ((N >> 4) << 14) | 0x0F70);
}
// ... followed by the workgroup barrier.  Not __syncthreads(): its workgroup-scope fence makes the compiler drain EVERY
// outstanding vector-memory operation (vmcnt(0)) - also the LDS-DMA of the stage-steps that are meant to stay in flight.
// The "memory" clobber keeps the compiler from moving LDS reads across; the stage published here is covered by the count.
template <int N>
__device__ __forceinline__ void barrier_vm() {
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits on gfx9");
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// p.A = V [36][T][Cin] (p.sA0 = T Cin, p.lda = Cin), p.B = U [36][Cout][Cin] (p.sB0 = Cout Cin, p.ldb = Cin), p.C = y (NHWC, p.ldc),
// p.M = T tiles, p.N = Cout, p.K = Cin (a multiple of 32 SUB); p.fdHoWo / p.fdWo divide by the tiles per image / per tile row
template <int SUB, int NST>
__global__ __launch_bounds__(NTHREADS) void wino4_fused_kernel(const DevArgs p) {
  constexpr int BM = 64, BN = 64, D = NST - 1;
  constexpr int A_TILE = BK * BM, SUBT = BK * (BM + BN), STAGE = SUB * SUBT;
  constexpr int NDMA = 4 * SUB;                  // LDS-DMA wave-instructions per thread and stage-step
  static_assert(NST * STAGE >= 4 * EPI_WAVE, "the epilogue scratch lives in the ring");
  using AL = WinoKC<BM>;
  using BL = WinoKC<BN>;
  __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];

  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;     // neighbours share the V rows in one L2
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int kcs = p.K / (BK * SUB);              // stage-steps per position

  AL al;
  BL bl;
  al.init(p.A, p.lda, row0, p.M);
  bl.init(p.B, p.ldb, col0, p.N);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1, h = lane >> 5, l31 = lane & 31;
  const int arow = wm * 32 + l31, brow = wn * 32 + l31;

  f32x16 O[4][4], Y[4], accA, accB;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    accA[e] = 0.f;
    accB[e] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Y[j][e] = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) O[i][j][e] = 0.f;
    }
  }

  // ---- the prefetch stream: stage-step (pp, pk) is the next one to issue; past the end it re-reads the last one ----
  int pp = 0, pk = 0;
  const long wrapA = p.sA0 - (long)kcs * (BK * SUB), wrapB = p.sB0 - (long)kcs * (BK * SUB);
  auto advance = [&]() {                         // branch-free (scalar selects): a K step stays one basic block for the scheduler
    const bool last = pp == 35 && pk == kcs - 1, wrap = pk + 1 == kcs;
    al.off += last ? 0L : (wrap ? wrapA + BK * SUB : (long)(BK * SUB));
    bl.off += last ? 0L : (wrap ? wrapB + BK * SUB : (long)(BK * SUB));
    pk = last ? pk : (wrap ? 0 : pk + 1);
    pp = (wrap && !last) ? pp + 1 : pp;
  };
  auto issue = [&](int d, float* stage) {        // d-th DMA of a stage-step: sub-tile d / 4, slot d % 4 = A0, A1, B0, B1
    const int sub = d >> 2, slot = d & 3;
    float* base = stage + sub * SUBT;
    if (slot < 2) glds16(al.src(slot) + sub * BK, AL::dma_dst(base, slot));
    else glds16(bl.src(slot - 2) + sub * BK, BL::dma_dst(base + A_TILE, slot - 2));
  };
#pragma unroll
  for (int s = 0; s < D; ++s) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) issue(d, lds + s * STAGE);
    advance();
  }
  barrier_vm<NDMA * (D - 1)>();

  int cs = 0;                                    // ring stage being multiplied
  // One stage-step: 16 SUB MFMAs on stage cs into `cur`; the DMA of stage-step s + D goes into the stage multiplied one step
  // ago; FIRST (a position's first step): the finished product `oth` of position (., NUP) is folded into Y and cleared,
  // one accumulator register per MFMA gap.
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;  // A^T column of the position being folded
  auto set_nu = [&](int nu) {                    // [1 0 0 0], [1 1 1 1], [1 -1 1 -1], [1 2 4 8], [1 -2 4 -8], [0 0 0 1]
    const float sg = (nu == 2 || nu == 4) ? -1.f : 1.f, two = nu >= 3 ? 2.f : 1.f;
    w0 = nu == 5 ? 0.f : 1.f;
    w1 = (nu == 0 || nu == 5) ? 0.f : sg * two;
    w2 = (nu == 0 || nu == 5) ? 0.f : two * two;
    w3 = nu == 0 ? 0.f : (nu == 5 ? 1.f : sg * two * two * two);
  };
  auto step = [&](auto first_c, f32x16& cur, f32x16& oth) {
    constexpr bool FIRST = decltype(first_c)::value != 0;
    const float* st = lds + cs * STAGE;
    float* pf = lds + (cs == 0 ? NST - 1 : cs - 1) * STAGE;
    f32x4 fa[2], fb[2];
    fa[0] = read_frag<true, BM>(st, arow, 0, h);
    fb[0] = read_frag<true, BN>(st + A_TILE, brow, 0, h);
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int pc = sub * 16 + g * 4 + q4, fi = (sub * 4 + g) & 1;
          cur = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[fi][q4], fb[fi][q4], cur, 0, 0, 0);
          if ((pc & 1) == 0 && (pc >> 1) < NDMA) issue(pc >> 1, pf);
          if (q4 == 1 && !(sub == SUB - 1 && g == 3)) {          // the next group's fragments into the other register set
            const int ng = g == 3 ? 0 : g + 1, nsub = g == 3 ? sub + 1 : sub;
            fa[fi ^ 1] = read_frag<true, BM>(st + nsub * SUBT, arow, ng, h);
            fb[fi ^ 1] = read_frag<true, BN>(st + nsub * SUBT + A_TILE, brow, ng, h);
          }
          if (FIRST && sub == 0) {
            const int e = g * 4 + q4;
            const float a = oth[e];
            Y[0][e] = __builtin_fmaf(w0, a, Y[0][e]);
            Y[1][e] = __builtin_fmaf(w1, a, Y[1][e]);
            Y[2][e] = __builtin_fmaf(w2, a, Y[2][e]);
            Y[3][e] = __builtin_fmaf(w3, a, Y[3][e]);
            oth[e] = 0.f;
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    advance();
    cs = cs + 1 == NST ? 0 : cs + 1;
    barrier_vm<NDMA * (D - 1)>();                 // stage-step s + 1 has landed (s + 2 .. s + D may still be in flight)
  };
  // xi direction: O[i][j] += A^T[i][xi] Y[j], Y = 0
  auto fold_xi = [&](int xi) {
    const float sg = (xi == 2 || xi == 4) ? -1.f : 1.f, two = xi >= 3 ? 2.f : 1.f;
    const float a0 = xi == 5 ? 0.f : 1.f;
    const float a1 = (xi == 0 || xi == 5) ? 0.f : sg * two;
    const float a2 = (xi == 0 || xi == 5) ? 0.f : two * two;
    const float a3 = xi == 0 ? 0.f : (xi == 5 ? 1.f : sg * two * two * two);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float y = Y[j][e];
        O[0][j][e] = __builtin_fmaf(a0, y, O[0][j][e]);
        O[1][j][e] = __builtin_fmaf(a1, y, O[1][j][e]);
        O[2][j][e] = __builtin_fmaf(a2, y, O[2][j][e]);
        O[3][j][e] = __builtin_fmaf(a3, y, O[3][j][e]);
        Y[j][e] = 0.f;
      }
  };

  for (int xi = 0; xi < 6; ++xi) {
    for (int nu = 0; nu < 6; nu += 2) {
      set_nu(nu == 0 ? 5 : nu - 1);
      step(IC<1>{}, accA, accB);                  // position (xi, nu); folds the one before it (for nu = 0: (xi - 1, 5), into the previous row's Y)
      if (nu == 0 && xi > 0) fold_xi(xi - 1);
      for (int k = 1; k < kcs; ++k) step(IC<0>{}, accA, accB);
      set_nu(nu);
      step(IC<1>{}, accB, accA);                  // position (xi, nu + 1)
      for (int k = 1; k < kcs; ++k) step(IC<0>{}, accB, accA);
    }
  }
  set_nu(5);
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const float a = accB[e];
    Y[0][e] = __builtin_fmaf(w0, a, Y[0][e]);
    Y[1][e] = __builtin_fmaf(w1, a, Y[1][e]);
    Y[2][e] = __builtin_fmaf(w2, a, Y[2][e]);
    Y[3][e] = __builtin_fmaf(w3, a, Y[3][e]);
  }
  fold_xi(5);
  wait_vm<0>();                                   // the re-read stage-steps past the end: nothing may land in the scratch below
  __syncthreads();

  // ---- epilogue: the 16 pixels of every tile; 32 x 32 blocks transposed through a wave-private LDS scratch to float4 stores ----
  float* scratch = lds + wave * EPI_WAVE;
  const int rr = lane >> 3, c4 = (lane & 7) * 4;
  const int n = col0 + wn * 32 + c4;
  const bool n_ok = n < p.N;
  const int tiles_img = (p.g.Ho >> 2) * (p.g.Wo >> 2), TW = p.g.Wo >> 2;
  long pix0[4];
  int img[4];
  bool m_ok[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int m = row0 + wm * 32 + rr + 8 * q;
    m_ok[q] = m < p.M && n_ok;
    const int mm = m < p.M ? m : p.M - 1;
    img[q] = p.fdHoWo.div(mm);
    const int rem = mm - img[q] * tiles_img;
    const int ty = p.fdWo.div(rem), tx = rem - ty * TW;
    pix0[q] = ((long)img[q] * p.g.Ho + 4 * ty) * p.g.Wo + 4 * tx;
  }
  f32x4 bias = zero4();
  if (p.bias && n_ok) bias = ldg4(p.bias + n);
  f32x4 ra[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ra[q] = bias;
    if (p.rowadd && m_ok[q]) ra[q] += ldg4(p.rowadd + (long)img[q] * p.ld_rowadd + n);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int e = 0; e < 16; ++e) scratch[((e & 3) + 8 * (e >> 2) + 4 * h) * EPI_LD + l31] = O[i][j][e];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (rr + 8 * q) * EPI_LD + c4);
        if (!m_ok[q]) continue;
        const long pix = pix0[q] + (long)i * p.g.Wo + j;
        v = f32x4{__builtin_fmaf(v[0], p.alpha, ra[q][0]), __builtin_fmaf(v[1], p.alpha, ra[q][1]),
                  __builtin_fmaf(v[2], p.alpha, ra[q][2]), __builtin_fmaf(v[3], p.alpha, ra[q][3])};
        if (p.residual) v += ldg4(p.residual + pix * p.ldr + n);
        *reinterpret_cast<f32x4*>(p.C + pix * p.ldc + n) = v;
      }
    }
}


// ------------------------------------------------------------------------------------
// The same algorithm on (32 tiles x 64 channels) blocks with v_mfma_f32_16x16x4_f32: a wave owns 16 tiles x 32 channels, so an
// accumulator block is 8 registers instead of 16 and the 22 blocks (O 16, Y 4, two position accumulators) fit in 256
// registers - TWO workgroups per CU.  That is what the 64 x 64 form cannot have: with one workgroup per CU its epilogue (16
// pixels x 64 channels per tile out, the residual in: as many bytes as the whole V stream of a 128-channel layer) runs with
// the matrix pipe idle, and all CUs reach it together.  Here one workgroup's epilogue and prologue overlap the other's K loop.
// Fragments: lane (r = l & 15, kg = l >> 4) reads the 16-byte chunk 4 g + kg of its row for half-step g (k = 16 g + 4 kg ..+3)
// and feeds element j to MFMA step j - the same k assignment for both operands; with the KC tile's XOR swizzle the
// ds_read_b128 is conflict-free (checked by enumeration).
// ------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int EPI2_LD = 36;                      // scratch row stride (floats): rows e and e + 4 of a ds_write_b32 half land 16 banks apart
constexpr int EPI2_WAVE = 16 * EPI2_LD;

// k-contiguous operand rows by LDS-DMA for a workgroup of NTH threads (WinoKC's layout: slot i of a thread = row (tid >> 3) + (NTH / 8) i,
// 16-byte chunk (tid & 7) swizzled by the row; rows beyond the operand clamp to the last one)
template <int ROWS, int NTH>
struct RowsKC {
  static constexpr int RPS = NTH / 8, NS = ROWS / RPS;
  static_assert(ROWS % RPS == 0, "whole slots");
  const float* ptr[NS];
  long off;
  __device__ void init(const float* b, int ld, int row0, int nrows) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      int r = row0 + (int)(threadIdx.x >> 3) + RPS * i;
      r = r < nrows ? r : nrows - 1;
      ptr[i] = b + (long)r * ld + ((threadIdx.x & 7) ^ ((threadIdx.x >> 4) & 7)) * 4;
    }
    off = 0;
  }
  __device__ __forceinline__ const float* src(int i) const { return ptr[i] + off; }
  __device__ static float* dma_dst(float* tile, int i) { return tile + (wave_id() * 8 + RPS * i) * BK; }
};

// WM x WN waves of 16 tiles x 32 channels: 2 x 2 (32 tiles x 64 channels) or 4 x 1 (64 tiles x 32 channels: output widths that are
// multiples of 32 but not of 64 - the pruned models' 96 / 160 / 224 / 288 - without a quarter of the MFMA work on padding);
// 256 threads and two workgroups per CU either way.  (4 x 2 - 512 threads, one U tile for eight waves - measured no faster.)
template <int WM, int WN, int SUB, int NST>
__global__ __launch_bounds__(64 * WM * WN, 2) void wino4_fused2_kernel(const DevArgs p) {
  constexpr int NTH = 64 * WM * WN, BM = 16 * WM, BN = 32 * WN, D = NST - 1;
  constexpr int A_TILE = BK * BM, SUBT = BK * (BM + BN), STAGE = SUB * SUBT;
  using AL = RowsKC<BM, NTH>;
  using BL = RowsKC<BN, NTH>;
  constexpr int NDA = AL::NS, NDB = BL::NS, NDS = NDA + NDB, NDMA = NDS * SUB;      // LDS-DMA wave-instructions per thread: per 32-deep sub-tile, per stage-step
  static_assert(NST * STAGE >= WM * WN * EPI2_WAVE, "the epilogue scratch lives in the ring");
  __shared__ __attribute__((aligned(16))) float lds[NST * STAGE];

  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int kcs = p.K / (BK * SUB);

  AL al;
  BL bl;
  al.init(p.A, p.lda, row0, p.M);
  bl.init(p.B, p.ldb, col0, p.N);

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave - wm * WN, r16 = lane & 15, kg = lane >> 4;
  // per-lane fragment bases (floats): chunk 4 g + kg of row arow / brow; both halves g = 0, 1 are compile-time offsets apart only
  // through the swizzle, so keep both
  const int arow = wm * 16 + r16, brow = wn * 32 + r16;
  const int aoff0 = kc_off(arow, kg), aoff1 = kc_off(arow, 4 + kg);
  const int boff0 = kc_off(brow, kg), boff1 = kc_off(brow, 4 + kg);          // second channel tile: + 16 rows = + 16 BK floats (same swizzle phase: (r >> 1) & 7 moves by 8)

  f32x4 O[4][4][2], Y[4][2], accA[2], accB[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    accA[c] = zero4();
    accB[c] = zero4();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      Y[j][c] = zero4();
#pragma unroll
      for (int i = 0; i < 4; ++i) O[i][j][c] = zero4();
    }
  }

  int pp = 0, pk = 0;
  const long wrapA = p.sA0 - (long)kcs * (BK * SUB), wrapB = p.sB0 - (long)kcs * (BK * SUB);
  auto advance = [&]() {
    const bool last = pp == 35 && pk == kcs - 1, wrap = pk + 1 == kcs;
    al.off += last ? 0L : (wrap ? wrapA + BK * SUB : (long)(BK * SUB));
    bl.off += last ? 0L : (wrap ? wrapB + BK * SUB : (long)(BK * SUB));
    pk = last ? pk : (wrap ? 0 : pk + 1);
    pp = (wrap && !last) ? pp + 1 : pp;
  };
  auto issue = [&](int d, float* stage) {        // d-th DMA of a stage-step: sub-tile d / NDS, slot d % NDS = A slots, then B slots
    const int sub = d / NDS, slot = d - NDS * sub;
    float* base = stage + sub * SUBT;
    if (slot < NDA) glds16(al.src(slot) + sub * BK, AL::dma_dst(base, slot));
    else glds16(bl.src(slot - NDA) + sub * BK, BL::dma_dst(base + A_TILE, slot - NDA));
  };
#pragma unroll
  for (int s = 0; s < D; ++s) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) issue(d, lds + s * STAGE);
    advance();
  }
  barrier_vm<NDMA * (D - 1)>();

  int cs = 0;
  float w0 = 0.f, w1 = 0.f, w2 = 0.f, w3 = 0.f;
  auto set_nu = [&](int nu) {
    const float sg = (nu == 2 || nu == 4) ? -1.f : 1.f, two = nu >= 3 ? 2.f : 1.f;
    w0 = nu == 5 ? 0.f : 1.f;
    w1 = (nu == 0 || nu == 5) ? 0.f : sg * two;
    w2 = (nu == 0 || nu == 5) ? 0.f : two * two;
    w3 = nu == 0 ? 0.f : (nu == 5 ? 1.f : sg * two * two * two);
  };
  auto step = [&](auto first_c, f32x4 (&cur)[2], f32x4 (&oth)[2]) {
    constexpr bool FIRST = decltype(first_c)::value != 0;
    const float* st = lds + cs * STAGE;
    float* pf = lds + (cs == 0 ? NST - 1 : cs - 1) * STAGE;
    f32x4 fa[2], fb[2][2];
    fa[0] = *reinterpret_cast<const f32x4*>(st + aoff0);
    fb[0][0] = *reinterpret_cast<const f32x4*>(st + A_TILE + boff0);
    fb[0][1] = *reinterpret_cast<const f32x4*>(st + A_TILE + 16 * BK + boff0);
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const int fi = (sub * 2 + g) & 1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int pc = (sub * 2 + g) * 4 + j;                 // piece = two MFMAs (the two channel tiles)
          cur[0] = mfma16(fa[fi][j], fb[fi][0][j], cur[0]);
          cur[1] = mfma16(fa[fi][j], fb[fi][1][j], cur[1]);
          if (pc < NDMA) issue(pc, pf);
          if (j == 1 && !(sub == SUB - 1 && g == 1)) {         // the next half-step's fragments
            const int ng = g ^ 1, nsub = g == 1 ? sub + 1 : sub;
            const float* nb = st + nsub * SUBT;
            fa[fi ^ 1] = *reinterpret_cast<const f32x4*>(nb + (ng ? aoff1 : aoff0));
            fb[fi ^ 1][0] = *reinterpret_cast<const f32x4*>(nb + A_TILE + (ng ? boff1 : boff0));
            fb[fi ^ 1][1] = *reinterpret_cast<const f32x4*>(nb + A_TILE + 16 * BK + (ng ? boff1 : boff0));
          }
          if (FIRST && pc < 8) {                              // fold register pc of the finished position
            const int c = pc >> 2, e = pc & 3;
            const float a = oth[c][e];
            Y[0][c][e] = __builtin_fmaf(w0, a, Y[0][c][e]);
            Y[1][c][e] = __builtin_fmaf(w1, a, Y[1][c][e]);
            Y[2][c][e] = __builtin_fmaf(w2, a, Y[2][c][e]);
            Y[3][c][e] = __builtin_fmaf(w3, a, Y[3][c][e]);
            oth[c][e] = 0.f;
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    advance();
    cs = cs + 1 == NST ? 0 : cs + 1;
    barrier_vm<NDMA * (D - 1)>();
  };
  auto fold_xi = [&](int xi) {
    const float sg = (xi == 2 || xi == 4) ? -1.f : 1.f, two = xi >= 3 ? 2.f : 1.f;
    const float a0 = xi == 5 ? 0.f : 1.f;
    const float a1 = (xi == 0 || xi == 5) ? 0.f : sg * two;
    const float a2 = (xi == 0 || xi == 5) ? 0.f : two * two;
    const float a3 = xi == 0 ? 0.f : (xi == 5 ? 1.f : sg * two * two * two);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float y = Y[j][c][e];
          O[0][j][c][e] = __builtin_fmaf(a0, y, O[0][j][c][e]);
          O[1][j][c][e] = __builtin_fmaf(a1, y, O[1][j][c][e]);
          O[2][j][c][e] = __builtin_fmaf(a2, y, O[2][j][c][e]);
          O[3][j][c][e] = __builtin_fmaf(a3, y, O[3][j][c][e]);
          Y[j][c][e] = 0.f;
        }
  };

  for (int xi = 0; xi < 6; ++xi) {
    for (int nu = 0; nu < 6; nu += 2) {
      set_nu(nu == 0 ? 5 : nu - 1);
      step(IC<1>{}, accA, accB);
      if (nu == 0 && xi > 0) fold_xi(xi - 1);
      for (int k = 1; k < kcs; ++k) step(IC<0>{}, accA, accB);
      set_nu(nu);
      step(IC<1>{}, accB, accA);
      for (int k = 1; k < kcs; ++k) step(IC<0>{}, accB, accA);
    }
  }
  set_nu(5);
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float a = accB[c][e];
      Y[0][c][e] = __builtin_fmaf(w0, a, Y[0][c][e]);
      Y[1][c][e] = __builtin_fmaf(w1, a, Y[1][c][e]);
      Y[2][c][e] = __builtin_fmaf(w2, a, Y[2][c][e]);
      Y[3][c][e] = __builtin_fmaf(w3, a, Y[3][c][e]);
    }
  fold_xi(5);
  wait_vm<0>();
  __syncthreads();

  // ---- epilogue: per (i, j) the wave's [16 tiles][32 channels] block -> scratch -> float4 rows of 128 B ----
  float* scratch = lds + wave * EPI2_WAVE;
  const int rr = lane >> 3, c4 = (lane & 7) * 4;
  const int n = col0 + wn * 32 + c4;
  const bool n_ok = n < p.N;
  const int tiles_img = (p.g.Ho >> 2) * (p.g.Wo >> 2), TW = p.g.Wo >> 2;
  long pix0[2];
  int img[2];
  bool m_ok[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = row0 + wm * 16 + rr + 8 * q;
    m_ok[q] = m < p.M && n_ok;
    const int mm = m < p.M ? m : p.M - 1;
    img[q] = p.fdHoWo.div(mm);
    const int rem = mm - img[q] * tiles_img;
    const int ty = p.fdWo.div(rem), tx = rem - ty * TW;
    pix0[q] = ((long)img[q] * p.g.Ho + 4 * ty) * p.g.Wo + 4 * tx;
  }
  f32x4 bias = zero4();
  if (p.bias && n_ok) bias = ldg4(p.bias + n);
  f32x4 ra[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    ra[q] = bias;
    if (p.rowadd && m_ok[q]) ra[q] += ldg4(p.rowadd + (long)img[q] * p.ld_rowadd + n);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // the residual of the whole output row i (4 pixels x 2 tile rows per lane) is requested before any of it is needed: one
    // memory round trip per row instead of one per pixel (the loads used to sit between each transpose and its store)
    f32x4 rs[4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q)
        rs[j][q] = (p.residual && m_ok[q]) ? ldg4(p.residual + (pix0[q] + (long)i * p.g.Wo + j) * p.ldr + n) : zero4();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) scratch[(4 * kg + e) * EPI2_LD + 16 * c + r16] = O[i][j][c][e];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        f32x4 v = *reinterpret_cast<const f32x4*>(scratch + (rr + 8 * q) * EPI2_LD + c4);
        if (!m_ok[q]) continue;
        const long pix = pix0[q] + (long)i * p.g.Wo + j;
        v = f32x4{__builtin_fmaf(v[0], p.alpha, ra[q][0]), __builtin_fmaf(v[1], p.alpha, ra[q][1]),
                  __builtin_fmaf(v[2], p.alpha, ra[q][2]), __builtin_fmaf(v[3], p.alpha, ra[q][3])};
        v += rs[j][q];
        *reinterpret_cast<f32x4*>(p.C + pix * p.ldc + n) = v;      // (plain accesses: non-temporal ones measured 5-15 % slower, profiles/r04_wino4_fused_experiments.txt)
      }
    }
  }
}

}  // namespace

namespace gadk {

// grid = tiles_m x tiles_n blocks of bm tiles x 64 channels (bm = 64: one workgroup per CU, 32: two); K steps of 64 where Cin
// allows it (half the barriers)
void launch_wino4_fused(const DevArgs& w, int bm, hipStream_t st) {
  dim3 grid((unsigned)((long)w.tiles_m * w.tiles_n)), block(NTHREADS);
  if (bm == 64) {
    if (w.K % 64 == 0) hipLaunchKernelGGL((wino4_fused_kernel<2, 3>), grid, block, 0, st, w);
    else hipLaunchKernelGGL((wino4_fused_kernel<1, 4>), grid, block, 0, st, w);
  } else if (bm == 32) {                         // 32 tiles x 64 channels
    if (w.K % 64 == 0) hipLaunchKernelGGL((wino4_fused2_kernel<2, 2, 2, 3>), grid, block, 0, st, w);
    else hipLaunchKernelGGL((wino4_fused2_kernel<2, 2, 1, 4>), grid, block, 0, st, w);
  } else {                                       // bm == 65: 64 tiles x 32 channels
    if (w.K % 64 == 0) hipLaunchKernelGGL((wino4_fused2_kernel<4, 1, 2, 3>), grid, block, 0, st, w);
    else hipLaunchKernelGGL((wino4_fused2_kernel<4, 1, 1, 4>), grid, block, 0, st, w);
  }
}

}  // namespace gadk
