// Half-precision ACTIVATION path for gfx950 (MI355X): the arithmetic the reference's Stable-Diffusion jobs run
// (`--mixed_precision=fp16`, text_to_image/experiments/setup_train_commands.py:127; frozen weights cast to 16 bit and autocast
// around the U-Net, text_to_image/train_text_to_image_lora.py:752-760,1268-1270) with bf16 as the 16-bit type.
//
//   * activations and their gradients are bf16 in HBM; every accumulation, statistic and epilogue is fp32;
//   * hgemm_kernel: ONE contraction engine for every Linear / convolution / data gradient / LoRA parameter gradient of the
//     path - C[m][n] = sum_k A(m, k) B[n][k] on v_mfma_f32_32x32x16_bf16 - with both operands streamed HBM/L2 -> LDS by
//     LDS-DMA (global_load_lds_dwordx4: 16 B per lane, no VGPR round trip); the im2col gather of a convolution is nothing
//     but a per-lane SOURCE address (padding = a block of zeros in device memory), so a 3x3 convolution, its data gradient
//     (rotated weights), a stride-2 / upsample-fused convolution and a dense GEMM are the same kernel;
//   * LDS tile images are [row][BK] bf16 with the 16-B chunk index XOR-swizzled by the row (applied to the DMA's SOURCE chunk
//     and to the fragment read), which makes every ds_read_b128 of a 32-row fragment conflict-free;
//   * tiles: 256 x 320 on eight waves / 128 x 320 (SD channel counts 320 / 640 / 960 / 1280 / 1920 / 2560 are all multiples of 320: no
//     column waste) and 128 x 128 (everything else: LoRA ranks, 4-channel conv_out, the 16x16 level's Linears, ragged shapes);
//   * hgemm_tn_kernel contracts over the TOKEN axis with both operands read in place (the LoRA parameter gradients).
#include "gad_common.h"

namespace gadh {

typedef unsigned short u16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static __device__ __attribute__((aligned(64))) unsigned int g_zero[16];   // the DMA source of padding / out-of-range chunks

__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) {          // round to nearest even (NaN kept quiet)
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u16)((u >> 16) | 0x40);
  return (u16)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ unsigned pack2(float a, float b) { return (unsigned)f2bf(a) | ((unsigned)f2bf(b) << 16); }
__device__ __forceinline__ void unpack8(const u32x4& v, float (&f)[8]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f[2 * i] = __uint_as_float(v[i] << 16);
    f[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ u32x4 pack8(const float (&f)[8]) {
  return u32x4{pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
}

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  int xcd = bid & 7, q = nwg >> 3, rr = nwg & 7;
  return (xcd < rr ? xcd * (q + 1) : rr * (q + 1) + (xcd - rr) * q) + (bid >> 3);
}

struct HDev {
  const u16* A; const u16* A2; const u16* B; const u16* B2;
  int M, N;
  int lda, lda2, ldb, ldb2;
  int len1, len2;           // segment lengths of one K group (dense: whole K split in two; conv: channels of source 1 / 2)
  int n1, n2;               // K steps per segment (ceil(len / BK)); steps per group = n1 + n2
  int groups;               // K groups: the taps of a convolution (1 for a dense contraction)
  int conv, H, W, Ho, Wo, KW, stride, pad_t, pad_l, ups;
  float alpha; const float* bias; const float* rowadd; int rpg, ld_rowadd; const u16* residual; int ldr;
  void* C; int ldc, out_f32, accumulate;
  float* ws;
  int tiles_m, tiles_n, splitk, steps_per_split, steps;
  int vec;                  // the epilogue's operands (or the split-K slabs) can be moved as aligned 16-byte vectors
  int vec_out;              // the output-side operands can (the split-K reduce's epilogue)
  const u16* zero;          // the zero block (a DMA source like any other)
  int dbg_zero;             // A/B tools only (tile_hint + 100): every DMA reads the zero block - the kernel without its memory system
};

// epilogue of one output element
__device__ __forceinline__ void epi_store(const HDev& p, int m, int n, int img, float v) {
  v *= p.alpha;
  if (p.bias) v += p.bias[n];
  if (p.rowadd) v += p.rowadd[(long)img * p.ld_rowadd + n];
  if (p.residual) v += bf2f(p.residual[(long)m * p.ldr + n]);
  if (p.out_f32) {
    float* c = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
    *c = p.accumulate ? *c + v : v;
  } else {
    reinterpret_cast<u16*>(p.C)[(long)m * p.ldc + n] = f2bf(v);
  }
}

// epilogue of 8 consecutive columns of one output row (v0 | v1: the raw accumulators), or their split-K slab store
__device__ __forceinline__ void epi_row8(const HDev& p, float* slab, int m, int n, const f32x4& v0, const f32x4& v1) {
  float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
  if (slab) {
    float* dst = slab + (long)m * p.N + n;
    if (p.vec) {
      *reinterpret_cast<f32x4*>(dst) = v0;
      *reinterpret_cast<f32x4*>(dst + 4) = v1;
    } else {
      for (int k = 0; k < 8 && n + k < p.N; ++k) dst[k] = v[k];
    }
    return;
  }
  const int img = p.rowadd ? m / p.rpg : 0;
  if (p.vec) {
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= p.alpha;
    if (p.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
    }
    if (p.rowadd) {
      const float* ra = p.rowadd + (long)img * p.ld_rowadd + n;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(ra), b1 = *reinterpret_cast<const f32x4*>(ra + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
    }
    if (p.residual) {
      float rr[8];
      unpack8(*reinterpret_cast<const u32x4*>(p.residual + (long)m * p.ldr + n), rr);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += rr[k];
    }
    if (p.out_f32) {
      float* cdst = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
      if (p.accumulate) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cdst), c1 = *reinterpret_cast<const f32x4*>(cdst + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += c0[k]; v[4 + k] += c1[k]; }
      }
      *reinterpret_cast<f32x4*>(cdst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(cdst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.C) + (long)m * p.ldc + n) = pack8(v);
    }
  } else {
    for (int k = 0; k < 8 && n + k < p.N; ++k) epi_store(p, m, n + k, img, v[k]);
  }
}

template <int N>
__device__ __forceinline__ void barrier_vm() {          // all but this wave's N youngest vector-memory operations are done, then the
  static_assert(N >= 0 && N < 64, "vmcnt is 6 bits");   // workgroup barrier - NOT __syncthreads(), whose fence drains every LDS-DMA in flight
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------
// hgemm_kernel<WM, WN, TM, TN, BKT, NST, ROLES, GATHER>
//   WM x WN waves, each TM x TN tiles of 32 x 32 (v_mfma_f32_32x32x16_bf16); K steps of BKT.
//   NST = 2: two stage buffers, the next step's DMA issued before this step's MFMAs; two workgroups per CU cover each other's waits.
//   NST >= 3: a ring with the DMA NST - 1 steps ahead (counted vmcnt, one raw barrier per step), one workgroup per CU.
//   ROLES: the lower half of the waves streams B (weights), the upper half A (activations / the gather): a wave carries one kind of
//          addressing.  GATHER: 0 dense A (+ K concatenation), 1 convolution gather, 2 gather with the nearest-2x / stride-2-transposed forms.
// The DMA issue path is the part that competes with the MFMAs for the vector issue slots (an MFMA leaves 24 of its 32 cycles to other
// vector instructions), so it is kept lean: 32-bit element offsets fixed before the loop, one chunk index per wave (rows are dealt to the
// DMA instructions so that the XOR swizzle is the same for all of a wave's instructions), the walk over (tap, segment, k) carried in
// scalars, the zero block's address in a kernel argument, no branches.
// ---------------------------------------------------------------------------------------------------------------
template <int WM, int WN, int TM, int TN, int BKT, int NST, bool ROLES, int GATHER, bool MF16 = false>
__global__ __launch_bounds__(WM* WN * 64, (NST == 2 || WM * WN == 8) ? 2 : 1) void hgemm_kernel(const HDev p) {
  constexpr int NW = WM * WN, BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int CPR = BKT / 8, RPI = 64 / CPR;             // 16-B chunks per tile row; tile rows per DMA wave-instruction
  constexpr int NWA = ROLES ? NW / 2 : NW, NWB = NWA;      // waves that issue A / B DMAs
  constexpr int AI = BM / RPI / NWA, BI = BN / RPI / NWB;  // DMA instructions per issuing wave and K step
  static_assert(BM % (RPI * NWA) == 0 && BN % (RPI * NWB) == 0, "tile rows must divide over the waves' DMA instructions");
  static_assert(NWA % 2 == 0, "interleaved row dealing keeps the swizzle per wave only for an even wave count");
  constexpr int A_BYTES = BM * BKT * 2, B_BYTES = BN * BKT * 2, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

  const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
  const int row0 = tile_m * BM, col0 = tile_n * BN;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave - wm * WN;
  const bool does_b = !ROLES || wave < NWB, does_a = !ROLES || wave >= NWB;
  const int wa = ROLES ? wave - NWB : wave, wb = wave;     // index among the A / B issuing waves
  const int lrow = lane / CPR, slot = lane % CPR;
  // chunk swizzle of the [row][BKT] images.  32x32x16 fragments (lane = row l & 31, chunk by l >> 5): (row >> 2) & 3 for 64-byte rows,
  // (row >> 1) & 7 for 128-byte rows.  16x16x32 fragments (MF16: lane = row l & 15, chunk l >> 4) need the permuted form
  // {0, 2, 3, 1}[(row >> 2) & 3]: with it each of ds_read_b128's four 16-lane groups ({0-3, 12-15, 20-27}, ...) lands on 16 distinct slots.
  auto swz = [](int row) { return BKT == 64 ? (row >> 1) & 7 : MF16 ? (0x78 >> (((row >> 2) & 3) * 2)) & 3 : (row >> 2) & 3; };
  // DMA instruction ii of an operand covers tile rows [ii * RPI, (ii + 1) * RPI); wave w issues ii = i * NWx + w, so (ii * RPI + lrow)'s
  // swizzle bits do not depend on i: one source chunk per wave and operand
  const int cha = (slot ^ swz(wa * RPI + lrow)) * 8, chb = (slot ^ swz(wb * RPI + lrow)) * 8;

  int aoff[AI], aoff2[GATHER ? 1 : AI], ay[GATHER ? AI : 1], ax[GATHER ? AI : 1];
  bool aok[AI];
#pragma unroll
  for (int i = 0; i < AI; ++i) {
    const int rl = (i * NWA + wa) * RPI + lrow, m = row0 + rl;
    aok[i] = does_a && m < p.M;
    if constexpr (GATHER != 0) {
      const int hw = p.Ho * p.Wo, mm = aok[i] ? m : 0;
      const int img = mm / hw, rem = mm - img * hw, oy = rem / p.Wo, ox = rem - oy * p.Wo;
      aoff[i] = img * p.H * p.W;                      // image base pixel
      ay[i] = oy * p.stride - p.pad_t;
      ax[i] = ox * p.stride - p.pad_l;
    } else {
      aoff[i] = m * p.lda + cha;
      aoff2[i] = m * p.lda2 + cha;
    }
  }
  int boff[BI], boff2[GATHER ? 1 : BI];
  bool bok[BI];
#pragma unroll
  for (int i = 0; i < BI; ++i) {
    const int rl = (i * NWB + wb) * RPI + lrow, n = col0 + rl;
    bok[i] = does_b && n < p.N;
    boff[i] = n * p.ldb + chb;
    if constexpr (GATHER == 0) boff2[i] = n * p.ldb2 + chb;
  }

  // The walk over K steps: step t = (channel step su, tap sg) with the TAP as the fast index - the nine gathers of one 32-/64-channel
  // slice of a pixel run back to back, so their re-reads of that slice meet in L2 (tap-major order re-read each slice after a whole
  // pass over the channels: 4.9x the algorithmic bytes at the fabric counters); conv taps (sr, ss) = (sg / KW, sg % KW).
  const int spg = p.n1 + p.n2;
  const int t0 = z * p.steps_per_split;
  const int t1 = min(p.steps, t0 + p.steps_per_split);
  int su = t0 / p.groups, sg = t0 - su * p.groups, sr = 0, ss = 0;
  if constexpr (GATHER != 0) { sr = sg / p.KW; ss = sg - sr * p.KW; }
  const u16* const zero = p.zero;
  auto stage = [&](int buf, bool live) {                 // issues the DMAs of the walk's current step, then advances the walk
    const bool s2 = su >= p.n1;
    const int kk = (s2 ? su - p.n1 : su) * BKT;
    const int rem = (s2 ? p.len2 : p.len1) - kk;         // valid K left in this segment (>= BKT except on a segment's last step)
    if (does_a) {
      unsigned char* dst = lds + buf * STAGE + wa * 1024;
      const bool tail_ok = cha < rem;
      if constexpr (GATHER != 0) {
        const u16* base = s2 ? p.A2 : p.A;
        const int ld = s2 ? p.lda2 : p.lda;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          int yy = ay[i] + sr, xx = ax[i] + ss;
          bool ok = live && aok[i] && tail_ok;
          if constexpr (GATHER == 2) {
            if (p.ups) { ok = ok && (unsigned)yy < 2u * p.H && (unsigned)xx < 2u * p.W; yy >>= 1; xx >>= 1; }
            if (p.conv == 2) { ok = ok && !((yy | xx) & 1); yy >>= 1; xx >>= 1; }
          }
          ok = ok && (unsigned)yy < (unsigned)p.H && (unsigned)xx < (unsigned)p.W;
          const int off = (aoff[i] + yy * p.W + xx) * ld + kk + cha;
          const u16* src = ok ? base + off : zero;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + i * (NWA * 1024)), 16, 0, 0);
        }
      } else {
        const u16* base = s2 ? p.A2 : p.A;
#pragma unroll
        for (int i = 0; i < AI; ++i) {
          const bool ok = live && aok[i] && tail_ok;
          const u16* src = ok ? base + ((s2 ? aoff2[i] : aoff[i]) + kk) : zero;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + i * (NWA * 1024)), 16, 0, 0);
        }
      }
    }
    if (does_b) {
      unsigned char* dst = lds + buf * STAGE + A_BYTES + wb * 1024;
      const bool tail_ok = chb < rem;
      const bool second = GATHER == 0 && s2 && p.B2 != nullptr;
      const u16* base = second ? p.B2 : p.B;
      const int kb = second ? kk : sg * (p.len1 + p.len2) + (s2 ? p.len1 : 0) + kk;
#pragma unroll
      for (int i = 0; i < BI; ++i) {
        const bool ok = live && bok[i] && tail_ok;
        const int o = GATHER == 0 ? (second ? boff2[i] : boff[i]) : boff[i];
        const u16* src = ok ? base + (o + kb) : zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(dst + i * (NWB * 1024)), 16, 0, 0);
      }
    }
    if (++sg == p.groups) {
      sg = 0;
      ++su;
      if constexpr (GATHER != 0) { sr = 0; ss = 0; }
    } else if constexpr (GATHER != 0) {
      if (++ss == p.KW) { ss = 0; ++sr; }
    }
  };

  // accumulators: TM x TN tiles of 32 x 32 (f32x16), or - MF16 - the same strip as 2 TM x 2 TN tiles of 16 x 16 (f32x4) for
  // v_mfma_f32_16x16x32_bf16, the shape on which the chip holds the higher clock under load (MI355X_MICROARCH.md, DVFS item 7)
  f32x16 acc[MF16 ? 1 : TM][MF16 ? 1 : TN];
  f32x4 acc4[MF16 ? 2 * TM : 1][MF16 ? 2 * TN : 1];
  if constexpr (MF16) {
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
      for (int j = 0; j < 2 * TN; ++j) acc4[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  }

  const int l31 = lane & 31, h = lane >> 5, l15 = lane & 15, g4 = lane >> 4;
  const int swl = swz(MF16 ? l15 : l31);
  const int a_base = (wm * TM * 32 + (MF16 ? l15 : l31)) * (BKT * 2);
  const int b_base = A_BYTES + (wn * TN * 32 + (MF16 ? l15 : l31)) * (BKT * 2);
  auto compute = [&](int buf) {
    const unsigned char* sb = lds + buf * STAGE;
    if constexpr (MF16) {
      // lane = (row l & 15, 8 k of chunk l >> 4 of a 32-deep block): one ds_read_b128 per 16-row fragment and block; rows 16 apart share the swizzle
#pragma unroll
      for (int kb = 0; kb < BKT / 32; ++kb) {
        const int off = ((4 * kb + g4) ^ swl) * 16;
        bf16x8 af[2 * TM], bfr[2 * TN];
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + i * 16 * BKT * 2 + off);
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_base + j * 16 * BKT * 2 + off);
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
          for (int j = 0; j < 2 * TN; ++j) acc4[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc4[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BKT / 16; ++kk) {
        const int off = ((2 * kk + h) ^ swl) * 16;
        bf16x8 af[TM], bfr[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sb + a_base + i * 32 * BKT * 2 + off);
#pragma unroll
        for (int j = 0; j < TN; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb + b_base + j * 32 * BKT * 2 + off);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  const bool live0 = !p.dbg_zero;
  if constexpr (NST == 2) {
    if (t0 < t1) {
      stage(0, live0);
      __builtin_amdgcn_s_waitcnt(0x0F70);        // vmcnt(0): the DMA is a VMEM operation
      __syncthreads();
    }
    for (int st = t0; st < t1; ++st) {
      const int buf = (st - t0) & 1;
      if (st + 1 < t1) stage(buf ^ 1, live0);
      compute(buf);
      __builtin_amdgcn_s_waitcnt(0x0F70);
      __syncthreads();
    }
  } else {
    constexpr int D = NST - 1;
    constexpr int NDA = ROLES ? AI : AI + BI, NDB = ROLES ? BI : AI + BI;      // DMAs per step of an A-issuing / B-issuing wave
    static_assert((D - 1) * NDA < 64 && (D - 1) * NDB < 64, "vmcnt range");
#pragma unroll
    for (int s_ = 0; s_ < D; ++s_) stage(s_, live0 && t0 + s_ < t1);
    int rd = 0, wr = D;                          // ring positions: read this step / refill (the buffer read in the previous step)
    for (int st = t0; st < t1; ++st) {
      if (ROLES && wave >= NWB) barrier_vm<(D - 1) * NDA>(); else barrier_vm<(D - 1) * NDB>();   // this step's stage landed; all past the last step's reads
      stage(wr, live0 && st + D < t1);
      compute(rd);
      rd = rd + 1 == NST ? 0 : rd + 1;
      wr = wr + 1 == NST ? 0 : wr + 1;
    }
    barrier_vm<0>();                             // nothing may still be landing in LDS when the epilogue reuses it
  }

  // ---- epilogue ----
  // Every wave is past the K loop's last barrier, so the stage buffers are free.  An MFMA accumulator has its COLUMN on the lane and
  // its rows in the registers: stored as it stands a bf16 output would leave in 2-byte pieces.  Each wave therefore passes its
  // strip through a private LDS patch, 16 rows x (TN * 32) columns at a time, after which a lane holds 8 consecutive columns of
  // one row: bias / time-embedding row / residual come in and the result goes out as 16-byte vectors, whole 64..320-byte row
  // segments per wave-instruction.  (The LDS operations of one wave complete in order: no barrier inside a wave's patch.)
  constexpr int EW = TN * 32, ELD = EW + 4, CH = EW / 8, TASKS = 16 * CH / 64;
  static_assert((16 * CH) % 64 == 0, "patch chunks must divide over the lanes");
  static_assert(NW * 16 * ELD * 4 <= NST * STAGE, "the patches live in the stage buffers");
  float* patch = reinterpret_cast<float*>(lds) + wave * (16 * ELD);
  float* slab = p.splitk > 1 ? p.ws + (long)z * p.M * p.N : nullptr;
#pragma clang loop unroll(full)
  for (int i = 0; i < TM; ++i) {
#pragma clang loop unroll(full)
    for (int hf = 0; hf < 2; ++hf) {
      if constexpr (MF16) {                // 16 x 16 accumulators: column = lane & 15, rows 4 (lane >> 4) + register: one tile row per pass
#pragma clang loop unroll(full)
        for (int j = 0; j < 2 * TN; ++j)
#pragma clang loop unroll(full)
          for (int e4 = 0; e4 < 4; ++e4) patch[(4 * g4 + e4) * ELD + j * 16 + l15] = acc4[2 * i + hf][j][e4];
      } else {
#pragma clang loop unroll(full)
        for (int j = 0; j < TN; ++j)
#pragma clang loop unroll(full)
          for (int q = 0; q < 2; ++q)
#pragma clang loop unroll(full)
            for (int e4 = 0; e4 < 4; ++e4)
              patch[(e4 + 8 * q + 4 * h) * ELD + j * 32 + l31] = acc[i][j][(2 * hf + q) * 4 + e4];
      }
#pragma clang loop unroll(full)
      for (int tk = 0; tk < TASKS; ++tk) {
        const int task = lane + 64 * tk;
        const int r = task / CH, c = (task - r * CH) * 8;
        const int m = row0 + (wm * TM + i) * 32 + 16 * hf + r, n = col0 + wn * EW + c;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(patch + r * ELD + c), v1 = *reinterpret_cast<const f32x4*>(patch + r * ELD + c + 4);
        if (m >= p.M || n >= p.N) continue;
        epi_row8(p, slab, m, n, v0, v1);
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// hgemm_tn_kernel: C[m][n] = sum_k A[k][m] B[k][n] - both operands stored with the CONTRACTION index as their row (activations
// [tokens][channels]): the LoRA parameter gradients dB = dy^T mid, dA = dmid^T x, G = dy^T x read in place, no transposed copies.
// An MFMA operand wants 8 consecutive k per lane; from a [k][column] image that is a transposing LDS read: ds_read_tr16_b64 hands
// each lane of a 16-lane group one column of a 4-row x 16-column block, so two of them make the 8-deep fragment of
// v_mfma_f32_16x16x32_bf16 (k slots (g, j < 4) -> row 4 g + j, (g, j >= 4) -> row 16 + 4 g + j - 4 for BOTH operands: the contraction
// order is permuted identically on the two sides).  Images are [32 k][128 columns] (256-byte rows, lane-linear for the LDS-DMA) with
// the 16-byte chunk index XOR-ed by (k & 7) << 1 on the DMA's source and on the read: the eight rows a 32-lane half touches then
// fall in eight different 32-byte slots of the bank row.  128 x 128 outputs per workgroup (4 waves x 4 x 4 tiles of 16 x 16),
// two stage buffers, fp32 output (direct, accumulating, or split-K slabs reduced by hgemm_reduce_kernel).
// ---------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ bf16x8 tn_frag(const unsigned char* img, int c0) {      // columns c0 .. c0 + 15 of a [32][128] image, 32 k
  const int lane = threadIdx.x & 63, g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const int row = 4 * g + q;                                                          // (row + 16 has the same low three bits)
  const int chunk = ((c0 >> 3) + (pp >> 1)) ^ ((row & 7) << 1);
  const unsigned char* a0 = img + row * 256 + chunk * 16 + (pp & 1) * 8;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 16 * 256));
  const s16x8 both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, both);
}
__global__ __launch_bounds__(256, 2) void hgemm_tn_kernel(const HDev p) {           // p.len1 = K (rows of A and B); p.M x p.N outputs
  constexpr int KB = 2;                                  // 32-row sub-images per stage: 64 rows of K per barrier
  constexpr int IMG = 32 * 256, OPER = KB * IMG, STAGE = 2 * OPER;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
  const int t = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
  const int tile_m = t / p.tiles_n, tile_n = t - tile_m * p.tiles_n;
  const int row0 = tile_m * 128, col0 = tile_n * 128;
  const int z = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  // DMA: instruction ii (0 .. 8 KB - 1 per operand) covers stage rows 4 ii .. 4 ii + 3; wave w issues ii = w + 4 i
  constexpr int NI = 2 * KB;
  const int dr = lane >> 4, pos = lane & 15;
  // (ii = w + 4 i -> row 4 w + 16 i + dr: its low three bits, hence the swizzle, do not depend on i)
  const int kr0 = wave * 4 + dr;
  const int ch = (pos ^ ((kr0 & 7) << 1)) * 8;           // source chunk (elements)
  const bool oka = row0 + ch < p.M, okb = col0 + ch < p.N;
  const int offa = kr0 * p.lda + row0 + ch, offb = kr0 * p.ldb + col0 + ch;
  const u16* const zero = p.zero;
  const int t0 = z * p.steps_per_split, t1 = min(p.steps, t0 + p.steps_per_split);
  auto stage = [&](int step, int buf) {
    const int k0 = step * (32 * KB);
    unsigned char* dst = lds + buf * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      const int kr = k0 + kr0 + 16 * i;
      const bool kin = kr < p.len1 && !p.dbg_zero;
      const u16* sa = (kin && oka) ? p.A + ((long)(k0 + 16 * i) * p.lda + offa) : zero;
      const u16* sb = (kin && okb) ? p.B + ((long)(k0 + 16 * i) * p.ldb + offb) : zero;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sa,
                                       (__attribute__((address_space(3))) void*)(dst + i * 4096), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)sb,
                                       (__attribute__((address_space(3))) void*)(dst + OPER + i * 4096), 16, 0, 0);
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (t0 < t1) {
    stage(t0, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  for (int st = t0; st < t1; ++st) {
    const int buf = (st - t0) & 1;
    if (st + 1 < t1) stage(st + 1, buf ^ 1);
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
      const unsigned char* ia = lds + buf * STAGE + kb * IMG;
      const unsigned char* ib = ia + OPER;
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = tn_frag(ia, wm * 64 + 16 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[j] = tn_frag(ib, wn * 64 + 16 * j);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
  }
  // C / D of the 16 x 16 MFMA: column = lane & 15, rows 4 (lane >> 4) + register
  float* slab = p.splitk > 1 ? p.ws + (long)z * p.M * p.N : nullptr;
  const int c = lane & 15, g = lane >> 4;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = col0 + wn * 64 + 16 * j + c;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int m = row0 + wm * 64 + 16 * i + 4 * g + e;
        if (m < p.M && n < p.N) {
          if (slab) slab[(long)m * p.N + n] = acc[i][j][e];
          else {
            float* cd = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
            const float v = acc[i][j][e] * p.alpha;
            *cd = p.accumulate ? *cd + v : v;
          }
        }
      }
    }
}

// split-K: sum the slabs in a fixed order and run the epilogue (8 columns per thread where the operands allow 16-byte vectors)
__global__ __launch_bounds__(256) void hgemm_reduce_kernel(const HDev p) {
  const long total = (long)p.M * p.N;
  if (p.vec_out) {
    const long idx = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
    if (idx >= total) return;
    const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int zz = 0; zz < p.splitk; ++zz) {
      const float* src = p.ws + (long)zz * total + idx;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(src), a1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += a0[k]; v[4 + k] += a1[k]; }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] *= p.alpha;
    if (p.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(p.bias + n), b1 = *reinterpret_cast<const f32x4*>(p.bias + n + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
    }
    if (p.rowadd) {
      const float* ra = p.rowadd + (long)(m / p.rpg) * p.ld_rowadd + n;
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(ra), b1 = *reinterpret_cast<const f32x4*>(ra + 4);
#pragma unroll
      for (int k = 0; k < 4; ++k) { v[k] += b0[k]; v[4 + k] += b1[k]; }
    }
    if (p.residual) {
      float rr[8];
      unpack8(*reinterpret_cast<const u32x4*>(p.residual + (long)m * p.ldr + n), rr);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] += rr[k];
    }
    if (p.out_f32) {
      float* cdst = reinterpret_cast<float*>(p.C) + (long)m * p.ldc + n;
      if (p.accumulate) {
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cdst), c1 = *reinterpret_cast<const f32x4*>(cdst + 4);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] += c0[k]; v[4 + k] += c1[k]; }
      }
      *reinterpret_cast<f32x4*>(cdst) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(cdst + 4) = f32x4{v[4], v[5], v[6], v[7]};
    } else {
      *reinterpret_cast<u32x4*>(reinterpret_cast<u16*>(p.C) + (long)m * p.ldc + n) = pack8(v);
    }
    return;
  }
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int m = (int)(idx / p.N), n = (int)(idx - (long)m * p.N);
  float v = 0.f;
  for (int zz = 0; zz < p.splitk; ++zz) v += p.ws[(long)zz * total + idx];
  epi_store(p, m, n, p.rowadd ? m / p.rpg : 0, v);
}

// ---------------------------------------------------------------------------------------------------------------
// transpose / cast / elementwise
// ---------------------------------------------------------------------------------------------------------------
template <bool F32>
__global__ __launch_bounds__(256) void transpose_kernel(const void* src, u16* dst, int rows, int cols, int ld_src, int ld_dst) {
  __shared__ u16 tile[64][66];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    u16 v = 0;
    if (r < rows && c < cols)
      v = F32 ? f2bf(reinterpret_cast<const float*>(src)[(long)r * ld_src + c]) : reinterpret_cast<const u16*>(src)[(long)r * ld_src + c];
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst[(long)c * ld_dst + r] = tile[tx][i];
  }
}
// bf16 -> bf16 with 16-byte global accesses on both sides: 64 x 64 tiles, rows and columns multiples of 8, aligned bases / strides
__global__ __launch_bounds__(256) void transpose16_kernel(const u16* __restrict__ src, u16* __restrict__ dst, int rows, int cols, int ld_src, int ld_dst) {
  __shared__ __attribute__((aligned(16))) u16 tile[64][72];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int q = threadIdx.x & 7, rr = threadIdx.x >> 3;          // 8 chunks per tile row, 32 tile rows per pass
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + rr + 32 * i, c = c0 + q * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r < rows && c < cols) v = *reinterpret_cast<const u32x4*>(src + (long)r * ld_src + c);
    *reinterpret_cast<u32x4*>(&tile[rr + 32 * i][q * 8]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = c0 + rr + 32 * i, r = r0 + q * 8;               // output row c, 8 consecutive source rows
    if (c < cols && r < rows) {
      u16 e[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) e[k] = tile[q * 8 + k][rr + 32 * i];
      *reinterpret_cast<u32x4*>(dst + (long)c * ld_dst + r) =
          u32x4{(unsigned)e[0] | ((unsigned)e[1] << 16), (unsigned)e[2] | ((unsigned)e[3] << 16), (unsigned)e[4] | ((unsigned)e[5] << 16),
                (unsigned)e[6] | ((unsigned)e[7] << 16)};
    }
  }
}

// bf16 shadows of every 2-D matrix resident in a flat fp32 parameter buffer, in ONE launch per optimizer step: dst[off + r * cols + c] =
// bf16(src[off + r * cols + c]) and dst_t[off + c * rows + r] = the same value (the transpose at the same offset).  table rows of five
// int64 {off, rows, cols, r0, c0}: one 64 x 64 tile each.  The LoRA matrices: down / up for the forward products, their transposes for
// the data gradients (dmid = dy up, dx += dmid down read them as [n][k] operands).
__global__ __launch_bounds__(256) void shadow_pair_kernel(const float* __restrict__ src, u16* __restrict__ dst, u16* __restrict__ dst_t,
                                                          const long* __restrict__ table) {
  __shared__ u16 tile[64][66];
  const long* row = table + 5L * blockIdx.x;
  const long off = row[0];
  const int rows = (int)row[1], cols = (int)row[2], r0 = (int)row[3], c0 = (int)row[4];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (int i = ty; i < 64; i += 4) {
    const int r = r0 + i, c = c0 + tx;
    u16 v = 0;
    if (r < rows && c < cols) {
      v = f2bf(src[off + (long)r * cols + c]);
      dst[off + (long)r * cols + c] = v;
    }
    tile[i][tx] = v;
  }
  __syncthreads();
  for (int i = ty; i < 64; i += 4) {
    const int c = c0 + i, r = r0 + tx;
    if (c < cols && r < rows) dst_t[off + (long)c * rows + r] = tile[tx][i];
  }
}

__global__ void cast_to_bf16_kernel(const float* __restrict__ src, u16* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i), b = *reinterpret_cast<const f32x4*>(src + i + 4);
    *reinterpret_cast<u32x4*>(dst + i) = u32x4{pack2(a[0], a[1]), pack2(a[2], a[3]), pack2(b[0], b[1]), pack2(b[2], b[3])};
  }
  if (i < n && i + 8 > n)
    for (long j = i; j < n; ++j) dst[j] = f2bf(src[j]);
}
__global__ void cast_to_f32_kernel(const u16* __restrict__ src, float* __restrict__ dst, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    float f[8];
    unpack8(*reinterpret_cast<const u32x4*>(src + i), f);
    *reinterpret_cast<f32x4*>(dst + i) = f32x4{f[0], f[1], f[2], f[3]};
    *reinterpret_cast<f32x4*>(dst + i + 4) = f32x4{f[4], f[5], f[6], f[7]};
  }
  if (i < n && i + 8 > n)
    for (long j = i; j < n; ++j) dst[j] = bf2f(src[j]);
}

__global__ void add_kernel(const u16* __restrict__ a, const u16* __restrict__ b, u16* __restrict__ out, long n) {
  long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const long stride = (long)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    float x[8], y[8];
    unpack8(*reinterpret_cast<const u32x4*>(a + i), x);
    unpack8(*reinterpret_cast<const u32x4*>(b + i), y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] += y[k];
    *reinterpret_cast<u32x4*>(out + i) = pack8(x);
  }
  if (i < n && i + 8 > n)
    for (long j = i; j < n; ++j) out[j] = f2bf(bf2f(a[j]) + bf2f(b[j]));
}

__global__ void upsample2x_bwd_kernel(const u16* __restrict__ dy, u16* __restrict__ dx, int B, int H, int W, int C8) {
  const long total = (long)B * H * W * C8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int c = (int)(idx % C8);
    long pix = idx / C8;
    const int x = (int)(pix % W);
    pix /= W;
    const int y = (int)(pix % H), b = (int)(pix / H);
    const long rowe = 2L * W * C8;                    // one row of the expanded grid, in octets
    const long base = (((long)b * 2 * H + 2 * y) * 2 * W + 2 * x) * C8 + c;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, f[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      unpack8(reinterpret_cast<const u32x4*>(dy)[base + (q >> 1) * rowe + (q & 1) * C8], f);
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] += f[k];
    }
    reinterpret_cast<u32x4*>(dx)[idx] = pack8(s);
  }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_df(float x) {
  return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}
__global__ void geglu_fwd_kernel(const u16* __restrict__ hh, u16* __restrict__ out, long M, int F8) {
  const long total = M * F8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / F8;
    const int c = (int)(idx - m * F8);
    float a[8], g[8];
    unpack8(reinterpret_cast<const u32x4*>(hh)[m * 2 * F8 + c], a);
    unpack8(reinterpret_cast<const u32x4*>(hh)[m * 2 * F8 + F8 + c], g);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] *= gelu_f(g[k]);
    reinterpret_cast<u32x4*>(out)[idx] = pack8(a);
  }
}
__global__ void geglu_bwd_kernel(const u16* __restrict__ hh, const u16* __restrict__ dout, u16* __restrict__ dh, long M, int F8) {
  const long total = M * F8;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const long m = idx / F8;
    const int c = (int)(idx - m * F8);
    float a[8], g[8], d[8], da[8], dg[8];
    unpack8(reinterpret_cast<const u32x4*>(hh)[m * 2 * F8 + c], a);
    unpack8(reinterpret_cast<const u32x4*>(hh)[m * 2 * F8 + F8 + c], g);
    unpack8(reinterpret_cast<const u32x4*>(dout)[idx], d);
#pragma unroll
    for (int k = 0; k < 8; ++k) { da[k] = d[k] * gelu_f(g[k]); dg[k] = d[k] * a[k] * gelu_df(g[k]); }
    reinterpret_cast<u32x4*>(dh)[m * 2 * F8 + c] = pack8(da);
    reinterpret_cast<u32x4*>(dh)[m * 2 * F8 + F8 + c] = pack8(dg);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm: one wave per row, octets of the row dealt to the lanes (C / 8 <= 64 * LN_MAXO)
// ---------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXO = 4;       // rows up to 2048 channels
__global__ __launch_bounds__(256) void ln_fwd_kernel(const u16* __restrict__ x, u16* __restrict__ y, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, float* __restrict__ mean, float* __restrict__ rstd,
                                                     long rows, int C, float eps) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, C8 = C >> 3;
  float v[LN_MAXO][8];
  float s = 0.f;
#pragma unroll
  for (int o = 0; o < LN_MAXO; ++o) {
    const int c = lane + 64 * o;
    if (c < C8) {
      unpack8(reinterpret_cast<const u32x4*>(x)[row * C8 + c], v[o]);
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[o][k];
    }
  }
  const float mu = wave_sum(s) / C;
  float q = 0.f;
#pragma unroll
  for (int o = 0; o < LN_MAXO; ++o)
    if (lane + 64 * o < C8)
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = v[o][k] - mu; q += d * d; }
  const float rs = rsqrtf(wave_sum(q) / C + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int o = 0; o < LN_MAXO; ++o) {
    const int c = lane + 64 * o;
    if (c < C8) {
      float r[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = (v[o][k] - mu) * rs * gamma[c * 8 + k] + beta[c * 8 + k];
      reinterpret_cast<u32x4*>(y)[row * C8 + c] = pack8(r);
    }
  }
}
__global__ __launch_bounds__(256) void ln_bwd_kernel(const u16* __restrict__ x, const u16* __restrict__ dy, u16* __restrict__ dx,
                                                     const u16* __restrict__ dx_add, const float* __restrict__ gamma,
                                                     const float* __restrict__ mean, const float* __restrict__ rstd, long rows, int C) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63, C8 = C >> 3;
  const float mu = mean[row], rs = rstd[row];
  float xh[LN_MAXO][8], g[LN_MAXO][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int o = 0; o < LN_MAXO; ++o) {
    const int c = lane + 64 * o;
    if (c < C8) {
      float a[8], d[8];
      unpack8(reinterpret_cast<const u32x4*>(x)[row * C8 + c], a);
      unpack8(reinterpret_cast<const u32x4*>(dy)[row * C8 + c], d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        xh[o][k] = (a[k] - mu) * rs;
        g[o][k] = d[k] * gamma[c * 8 + k];
        s1 += g[o][k];
        s2 += g[o][k] * xh[o][k];
      }
    }
  }
  s1 = wave_sum(s1) / C;
  s2 = wave_sum(s2) / C;
#pragma unroll
  for (int o = 0; o < LN_MAXO; ++o) {
    const int c = lane + 64 * o;
    if (c < C8) {
      float r[8], ad[8];
      if (dx_add) unpack8(reinterpret_cast<const u32x4*>(dx_add)[row * C8 + c], ad);
#pragma unroll
      for (int k = 0; k < 8; ++k) r[k] = rs * (g[o][k] - s1 - xh[o][k] * s2) + (dx_add ? ad[k] : 0.f);
      reinterpret_cast<u32x4*>(dx)[row * C8 + c] = pack8(r);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// GroupNorm (+ SiLU), NHWC bf16.  Two launches: per-(image, row chunk, group) partial sums | apply, whose prologue reduces the
// chunks' partials to the image's statistics in LDS (every workgroup of an image does the same few KB of L2 reads: cheaper than a
// third launch).  A thread owns ONE channel octet (blockDim = NO * RP with NO = C / 8 octets, RP row phases), so its per-channel
// coefficients are loop invariants and every access is a 16-B vector.
//   forward  partials: (sum x, sum x^2)            statistics: mean, rstd
//   backward partials: (sum g, sum g xhat), g = dy silu'(pre) gamma     statistics: s1 / n, s2 / n   (dx = rstd (g - s1/n - xhat s2/n))
// ---------------------------------------------------------------------------------------------------------------
struct GnDev {
  const u16* x; const u16* x2; int C1; const u16* dy; const u16* dx_add; u16* y;
  const float* gamma; const float* beta; float* mean; float* rstd;
  float* part;              // [B][chunks][G][2]
  int B, HW, C, G, silu, chunks, rows_per_chunk, NO, RP;
  float eps;
};
__device__ __forceinline__ void gn_load(const GnDev& p, long pix, int o, float (&f)[8]) {
  const int c = o * 8;
  if (p.x2 == nullptr) unpack8(reinterpret_cast<const u32x4*>(p.x + pix * p.C + c)[0], f);
  else if (c < p.C1) unpack8(reinterpret_cast<const u32x4*>(p.x + pix * p.C1 + c)[0], f);
  else unpack8(reinterpret_cast<const u32x4*>(p.x2 + pix * (p.C - p.C1) + (c - p.C1))[0], f);
}
__device__ __forceinline__ float silu_f(float v) { return v / (1.f + __expf(-v)); }
__device__ __forceinline__ float silu_df(float v) { const float s = 1.f / (1.f + __expf(-v)); return s * (1.f + v * (1.f - s)); }

template <bool BWD>
__global__ void gn_part_kernel(const GnDev p) {
  extern __shared__ float red[];          // [RP][NO][16]
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int o = threadIdx.x % p.NO, rp = threadIdx.x / p.NO;
  const int cpg = p.C / p.G;
  float a0[8], a1[8], mu[8], rs[8], ga[8], be[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    a0[k] = a1[k] = 0.f;
    if (BWD) {
      const int c = o * 8 + k, g = c / cpg;
      mu[k] = p.mean[b * p.G + g]; rs[k] = p.rstd[b * p.G + g]; ga[k] = p.gamma[c]; be[k] = p.beta[c];
    }
  }
  const int r0 = chunk * p.rows_per_chunk, r1 = min(p.HW, r0 + p.rows_per_chunk);
  for (int r = r0 + rp; r < r1; r += p.RP) {
    const long pix = (long)b * p.HW + r;
    float f[8];
    gn_load(p, pix, o, f);
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { a0[k] += f[k]; a1[k] += f[k] * f[k]; }
    } else {
      float d[8];
      unpack8(reinterpret_cast<const u32x4*>(p.dy + pix * p.C + o * 8)[0], d);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (f[k] - mu[k]) * rs[k];
        float g = d[k];
        if (p.silu) g *= silu_df(xh * ga[k] + be[k]);
        g *= ga[k];
        a0[k] += g;
        a1[k] += g * xh;
      }
    }
  }
  float* mine = red + ((long)rp * p.NO + o) * 16;
#pragma unroll
  for (int k = 0; k < 8; ++k) { mine[k] = a0[k]; mine[8 + k] = a1[k]; }
  __syncthreads();
  if (rp == 0) {
    for (int q = 1; q < p.RP; ++q) {
      const float* other = red + ((long)q * p.NO + o) * 16;
#pragma unroll
      for (int k = 0; k < 8; ++k) { a0[k] += other[k]; a1[k] += other[8 + k]; }
    }
  }
  __syncthreads();                       // every row phase's partials are read: the array is reused as [C][2] channel sums
  if (rp == 0) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { red[(o * 8 + k) * 2] = a0[k]; red[(o * 8 + k) * 2 + 1] = a1[k]; }
  }
  __syncthreads();
  for (int g = threadIdx.x; g < p.G; g += blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s0 += red[2 * c]; s1 += red[2 * c + 1]; }
    float* dst = p.part + (((long)b * p.chunks + chunk) * p.G + g) * 2;
    dst[0] = s0;
    dst[1] = s1;
  }
}
template <bool BWD>
__global__ void gn_apply_kernel(const GnDev p) {
  extern __shared__ float stat[];         // [G][2]: forward (mean, rstd); backward (sum g / n, sum g xhat / n)
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int o = threadIdx.x % p.NO, rp = threadIdx.x / p.NO;
  const int cpg = p.C / p.G;
  // the image's statistics from the chunks' partials: thread (g, part) sums every PARTS-th chunk of group g, LDS combines the parts in a
  // fixed order (every workgroup of the image gets the same bits)
  float* parts = stat + 2 * p.G;                           // [PARTS][G][2]
  const int PARTS = max(1, (int)blockDim.x / p.G);
  for (int idx = threadIdx.x; idx < PARTS * p.G; idx += blockDim.x) {
    const int g = idx % p.G, part = idx / p.G;
    float s0 = 0.f, s1 = 0.f;
    for (int ch = part; ch < p.chunks; ch += PARTS) {
      const float* src = p.part + (((long)b * p.chunks + ch) * p.G + g) * 2;
      s0 += src[0];
      s1 += src[1];
    }
    parts[(part * p.G + g) * 2] = s0;
    parts[(part * p.G + g) * 2 + 1] = s1;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < p.G; g += blockDim.x) {
    float s0 = 0.f, s1 = 0.f;
    for (int part = 0; part < PARTS; ++part) { s0 += parts[(part * p.G + g) * 2]; s1 += parts[(part * p.G + g) * 2 + 1]; }
    const float n = (float)p.HW * cpg;
    if (!BWD) {
      const float mu = s0 / n, var = fmaxf(s1 / n - mu * mu, 0.f), rs = rsqrtf(var + p.eps);
      stat[2 * g] = mu;
      stat[2 * g + 1] = rs;
      if (chunk == 0) { p.mean[b * p.G + g] = mu; p.rstd[b * p.G + g] = rs; }     // kept for the backward pass
    } else {
      stat[2 * g] = s0 / n;
      stat[2 * g + 1] = s1 / n;
    }
  }
  __syncthreads();
  float mu[8], rs[8], ga[8], be[8], s1[8], s2[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int c = o * 8 + k, g = c / cpg;
    ga[k] = p.gamma[c]; be[k] = p.beta[c];
    if (!BWD) { mu[k] = stat[2 * g]; rs[k] = stat[2 * g + 1]; }
    else { mu[k] = p.mean[b * p.G + g]; rs[k] = p.rstd[b * p.G + g]; s1[k] = stat[2 * g]; s2[k] = stat[2 * g + 1]; }
  }
  const int r0 = chunk * p.rows_per_chunk, r1 = min(p.HW, r0 + p.rows_per_chunk);
  for (int r = r0 + rp; r < r1; r += p.RP) {
    const long pix = (long)b * p.HW + r;
    float f[8], out[8];
    gn_load(p, pix, o, f);
    if (!BWD) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float v = (f[k] - mu[k]) * rs[k] * ga[k] + be[k];
        out[k] = p.silu ? silu_f(v) : v;
      }
    } else {
      float d[8], ad[8];
      unpack8(reinterpret_cast<const u32x4*>(p.dy + pix * p.C + o * 8)[0], d);
      if (p.dx_add) unpack8(reinterpret_cast<const u32x4*>(p.dx_add + pix * p.C + o * 8)[0], ad);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xh = (f[k] - mu[k]) * rs[k];
        float g = d[k];
        if (p.silu) g *= silu_df(xh * ga[k] + be[k]);
        g *= ga[k];
        out[k] = rs[k] * (g - s1[k] - xh * s2[k]) + (p.dx_add ? ad[k] : 0.f);
      }
    }
    reinterpret_cast<u32x4*>(p.y + pix * p.C + o * 8)[0] = pack8(out);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
struct HPlan { int tile, splitk, bk, bm, bn, steps, n1, n2, len1, len2, tiles_m, tiles_n, fits32; };

static int hgemm_check(const gad_hgemm_args* a) {
  GAD_CHECK(a && a->A && a->B && a->C, "gad_hgemm: null operand");
  GAD_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gad_hgemm: empty problem %d x %d x %d", a->M, a->N, a->K);
  GAD_CHECK(gad_aligned16(a->A) && gad_aligned16(a->B) && (!a->A2 || gad_aligned16(a->A2)) && (!a->B2 || gad_aligned16(a->B2)),
            "gad_hgemm: operands must be 16-byte aligned");
  GAD_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0 && (!a->A2 || a->lda2 % 8 == 0) && (!a->B2 || a->ldb2 % 8 == 0),
            "gad_hgemm: row strides must be multiples of 8 elements");
  GAD_CHECK(a->k_split % 8 == 0 && a->K % 8 == 0, "gad_hgemm: K (%d) and k_split (%d) must be multiples of 8", a->K, a->k_split);
  if (a->conv) {
    GAD_CHECK(a->conv == 1 || a->conv == 2, "gad_hgemm: conv mode %d", a->conv);
    GAD_CHECK(a->KH > 0 && a->KW > 0 && a->Cin > 0 && a->K == a->KH * a->KW * a->Cin, "gad_hgemm: K != KH*KW*Cin");
    GAD_CHECK(a->H > 0 && a->W > 0 && a->Ho > 0 && a->Wo > 0 && a->stride > 0, "gad_hgemm: bad convolution geometry");
    GAD_CHECK(a->k_split > 0 && a->k_split <= a->Cin && (a->k_split == a->Cin || a->A2), "gad_hgemm: bad channel split");
    GAD_CHECK(a->M % (a->Ho * a->Wo) == 0, "gad_hgemm: M is not a whole number of %d x %d maps", a->Ho, a->Wo);
    GAD_CHECK((long)(a->M / (a->Ho * a->Wo)) * a->H * a->W * (long)(a->lda > a->lda2 ? a->lda : a->lda2) < (1L << 40), "gad_hgemm: tensor too large");
    GAD_CHECK(a->B2 == nullptr, "gad_hgemm: conv takes one weight matrix");
  } else {
    GAD_CHECK(a->k_split > 0 && a->k_split <= a->K && (a->k_split == a->K || (a->A2 && a->B2)), "gad_hgemm: bad K split");
  }
  GAD_CHECK(!a->rowadd || a->rows_per_group >= 1, "gad_hgemm: rows_per_group must be >= 1");
  return 0;
}

static HPlan hgemm_plan(const gad_hgemm_args* a) {
  // Measured on the SD step's shapes (tools/ab_hgemm.py, profiles/r04_ab_hgemm.txt): the 128 x 320 tile wins wherever it yields a full
  // round of workgroups (2 per CU); below that, LONG contractions (3x3 convolutions: K >= 2304) are split along K - the fp32 slabs
  // cost less than idle CUs -, SHORT ones take 128 x 128 tiles instead (more workgroups, no slab traffic).
  HPlan pl{};
  const long tm = (a->M + 127) / 128;
  const long tiles2 = tm * ((a->N + 319) / 320), tiles1 = tm * ((a->N + 127) / 128);
  const bool long_k = a->K >= 2304;
  int tile;
  // 32-bit element offsets in the eight-wave kernel
  const long amax = a->conv ? (long)(a->M / (a->Ho * a->Wo)) * a->H * a->W * (a->lda > a->lda2 ? a->lda : a->lda2)
                            : (long)a->M * (a->lda > a->lda2 ? a->lda : a->lda2);
  const bool fits32 = amax < (1L << 31) && (long)a->N * (a->ldb > a->ldb2 ? a->ldb : a->ldb2) < (1L << 31);
  const long tiles8 = ((a->M + 255) / 256) * ((a->N + 319) / 320);
  const int hint = a->tile_hint % 100;
  if (hint == 1 || hint == 2 || (hint >= 5 && hint <= 9)) tile = hint;
  else if (a->N % 320 != 0) tile = 1;
  else if (long_k) tile = tiles8 >= 48 ? 6 : 7;                    // 3x3 convolutions: the eight-wave form (one round of tiles, or of K slices); 8x8 maps: 128 x 320 split
  else if (tiles2 >= 400) tile = 7;                                // Linears with a full round of 128 x 320 tiles
  else tile = tiles1 * 2 > tiles2 * 3 ? 1 : 7;
  if (tile == 1 && hint == 0 && tiles1 >= 512 && a->K <= 4096) tile = 8;     // enough 128 x 128 tiles for four workgroups per CU: the 32 KB form
  else if (tile == 1 && hint == 0 && a->N >= 640) tile = 9;                   // 16x16 level Linears: the 64-deep form on 16x16x32 MFMAs (+5-10 %)
  // (forms 6 / 7 = the 256 x 320 / 128 x 320 tiles on v_mfma_f32_16x16x32_bf16: +6-12 % over the 32x32x16 forms 5 / 2 on the convolutions,
  //  +3-8 % on the Linears - profiles/r04_ab_hgemm.txt; 5 / 2 stay for A/B)
  pl.tile = tile;
  pl.fits32 = fits32;
  const bool t320 = tile == 2 || (tile >= 5 && tile <= 7);
  pl.bk = (t320 || tile == 8) ? 32 : 64;
  pl.bm = (tile == 5 || tile == 6) ? 256 : 128;
  pl.bn = t320 ? 320 : 128;
  const int groups = a->conv ? a->KH * a->KW : 1;
  pl.len1 = a->k_split;
  pl.len2 = (a->conv ? a->Cin : a->K) - a->k_split;
  pl.n1 = (pl.len1 + pl.bk - 1) / pl.bk;
  pl.n2 = (pl.len2 + pl.bk - 1) / pl.bk;
  pl.steps = groups * (pl.n1 + pl.n2);
  pl.tiles_m = (a->M + pl.bm - 1) / pl.bm;
  pl.tiles_n = (a->N + pl.bn - 1) / pl.bn;
  const long tiles = (long)pl.tiles_m * pl.tiles_n;
  int sk = 1;
  if (a->splitk_hint > 0) sk = a->splitk_hint;
  else if (a->out_f32 && tiles <= 16 && a->K >= 1024) {
    // parameter gradients (a LoRA matrix: a handful of tiles, the whole token axis to contract): slices of >= 256
    sk = (int)(512 / tiles);
    if (sk > a->K / 256) sk = a->K / 256;
    if (sk > 128) sk = 128;
  } else if (long_k && tiles < 400) {
    const int slots = (tile == 5 || tile == 6) ? 256 : 512;                     // workgroups the chip holds at once (the eight-wave form: one per CU)
    sk = (int)((slots + tiles / 2) / tiles);
    if ((tile == 5 || tile == 6) && tiles >= 224) sk = 1;
    const int max_sk = a->K / 1152;                               // at least 1152 of K per slice
    if (sk > max_sk) sk = max_sk;
    if (sk > 128) sk = 128;
  }
  if (sk > pl.steps) sk = pl.steps;
  if (sk < 1) sk = 1;
  const int per = (pl.steps + sk - 1) / sk;
  pl.splitk = (pl.steps + per - 1) / per;
  return pl;
}

template <int WM, int WN, int TM, int TN, int BKT, int NST, bool ROLES, int GATHER, bool MF16>
static int hgemm_launch_g(const HDev& d, hipStream_t st) {
  constexpr int LDS = NST * (WM * TM * 32 + WN * TN * 32) * BKT * 2;
  auto kern = hgemm_kernel<WM, WN, TM, TN, BKT, NST, ROLES, GATHER, MF16>;
  if (LDS > 64 * 1024) {
    static unsigned done = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned bit = 1u << (dev & 31);
    if (!(__atomic_load_n(&done, __ATOMIC_ACQUIRE) & bit)) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      GAD_CHECK(e == hipSuccess, "gad_hgemm: cannot reserve %d bytes of LDS: %s", LDS, hipGetErrorString(e));
      __atomic_fetch_or(&done, bit, __ATOMIC_RELEASE);
    }
  }
  hipLaunchKernelGGL(kern, dim3(d.tiles_m * d.tiles_n, d.splitk), dim3(WM * WN * 64), LDS, st, d);
  GAD_LAUNCH_CHECK("hgemm_kernel");
  return 0;
}
template <int WM, int WN, int TM, int TN, int BKT, int NST, bool ROLES, bool MF16 = false>
static int hgemm_launch(const HDev& d, hipStream_t st) {
  if (!d.conv) return hgemm_launch_g<WM, WN, TM, TN, BKT, NST, ROLES, 0, MF16>(d, st);
  if (d.conv == 1 && !d.ups) return hgemm_launch_g<WM, WN, TM, TN, BKT, NST, ROLES, 1, MF16>(d, st);
  return hgemm_launch_g<WM, WN, TM, TN, BKT, NST, ROLES, 2, MF16>(d, st);
}

// the zero block's device address (per device; a static __device__ array of this translation unit)
static const u16* zero_block() {
  static const u16* addr[32] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  dev &= 31;
  const u16* p = __atomic_load_n(&addr[dev], __ATOMIC_ACQUIRE);
  if (!p) {
    void* sym = nullptr;
    if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_zero)) != hipSuccess || !sym) return nullptr;
    p = reinterpret_cast<const u16*>(sym);
    __atomic_store_n(&addr[dev], p, __ATOMIC_RELEASE);
  }
  return p;
}

static int64_t gn_chunks(const gad_groupnorm_args* a, int* rows_per_chunk, int* NO, int* RP) {
  const int no = a->C / 8;
  int rp = 256 / no;
  if (rp < 1) rp = 1;
  // enough workgroups to fill the chip: B * chunks >= ~1024, at least 4 * RP rows per chunk
  int chunks = (1024 + a->B - 1) / a->B;
  const int max_chunks = (a->HW + 4 * rp - 1) / (4 * rp);
  if (chunks > max_chunks) chunks = max_chunks;
  if (chunks < 1) chunks = 1;
  const int rpc = (a->HW + chunks - 1) / chunks;
  chunks = (a->HW + rpc - 1) / rpc;
  *rows_per_chunk = rpc; *NO = no; *RP = rp;
  return chunks;
}
static int gn_check(const gad_groupnorm_args* a, const char* who) {
  GAD_CHECK(a && a->x && a->y && a->gamma && a->beta && a->mean && a->rstd, "%s: null pointer", who);
  GAD_CHECK(a->B > 0 && a->HW > 0 && a->C > 0 && a->G > 0 && a->C % a->G == 0, "%s: bad shape", who);
  GAD_CHECK(a->C % 8 == 0 && a->C / 8 <= 1024, "%s: C must be a multiple of 8 (<= 8192)", who);
  GAD_CHECK(!a->x2 || (a->C1 % 8 == 0 && a->C1 > 0 && a->C1 < a->C), "%s: channel split must be a multiple of 8", who);
  GAD_CHECK(a->ws && a->ws_bytes >= gad_h_groupnorm_workspace_bytes(a), "%s: workspace too small", who);
  return 0;
}
static GnDev gn_dev(const gad_groupnorm_args* a) {
  GnDev d{};
  d.x = (const u16*)a->x; d.x2 = (const u16*)a->x2; d.C1 = a->C1; d.dy = (const u16*)a->dy; d.dx_add = (const u16*)a->dx_add;
  d.y = (u16*)a->y; d.gamma = a->gamma; d.beta = a->beta; d.mean = a->mean; d.rstd = a->rstd;
  d.B = a->B; d.HW = a->HW; d.C = a->C; d.G = a->G; d.silu = a->silu; d.eps = a->eps;
  d.chunks = (int)gn_chunks(a, &d.rows_per_chunk, &d.NO, &d.RP);
  d.part = (float*)a->ws;
  return d;
}

}  // namespace gadh

using namespace gadh;

extern "C" int gad_hgemm_plan(const gad_hgemm_args* a, int32_t* tile, int32_t* splitk) {
  if (hgemm_check(a)) return 1;
  const HPlan pl = hgemm_plan(a);
  if (tile) *tile = pl.tile;
  if (splitk) *splitk = pl.splitk;
  return 0;
}
extern "C" int64_t gad_hgemm_workspace_bytes(const gad_hgemm_args* a) {
  if (hgemm_check(a)) return -1;
  const HPlan pl = hgemm_plan(a);
  return pl.splitk > 1 ? (int64_t)pl.splitk * a->M * a->N * 4 : 0;
}
extern "C" int gad_hgemm(const gad_hgemm_args* a, void* stream) {
  if (hgemm_check(a)) return 1;
  const HPlan pl = hgemm_plan(a);
  GAD_CHECK(pl.fits32, "gad_hgemm: an operand spans 2^31 elements or more (the kernels address in 32-bit element offsets)");
  hipStream_t st = (hipStream_t)stream;
  if (pl.splitk > 1)
    GAD_CHECK(a->ws && a->ws_bytes >= (int64_t)pl.splitk * a->M * a->N * 4, "gad_hgemm: split-K workspace too small (%lld needed)",
              (long long)pl.splitk * a->M * a->N * 4);
  HDev d{};
  d.A = (const u16*)a->A; d.A2 = (const u16*)a->A2; d.B = (const u16*)a->B; d.B2 = (const u16*)a->B2;
  d.M = a->M; d.N = a->N;
  d.lda = a->lda; d.lda2 = a->lda2; d.ldb = a->ldb; d.ldb2 = a->ldb2;
  d.len1 = pl.len1; d.len2 = pl.len2; d.n1 = pl.n1; d.n2 = pl.n2;
  d.groups = a->conv ? a->KH * a->KW : 1;
  d.conv = a->conv; d.H = a->H; d.W = a->W; d.Ho = a->Ho; d.Wo = a->Wo; d.KW = a->KW; d.stride = a->stride;
  d.pad_t = a->pad_t; d.pad_l = a->pad_l; d.ups = a->upsample;
  d.alpha = a->alpha; d.bias = a->bias; d.rowadd = a->rowadd; d.rpg = a->rows_per_group; d.ld_rowadd = a->ld_rowadd;
  d.residual = (const u16*)a->residual; d.ldr = a->ldr;
  d.C = a->C; d.ldc = a->ldc; d.out_f32 = a->out_f32; d.accumulate = a->accumulate;
  d.ws = (float*)a->ws;
  d.tiles_m = pl.tiles_m; d.tiles_n = pl.tiles_n; d.splitk = pl.splitk; d.steps = pl.steps;
  d.steps_per_split = (pl.steps + pl.splitk - 1) / pl.splitk;
  d.dbg_zero = a->tile_hint >= 100;
  d.vec_out = a->N % 8 == 0 && gad_aligned16(a->C) && a->ldc % (a->out_f32 ? 4 : 8) == 0 && (!a->bias || gad_aligned16(a->bias)) &&
              (!a->rowadd || (gad_aligned16(a->rowadd) && a->ld_rowadd % 4 == 0)) &&
              (!a->residual || (gad_aligned16(a->residual) && a->ldr % 8 == 0));
  d.vec = pl.splitk > 1 ? (a->N % 8 == 0 && gad_aligned16(a->ws)) : d.vec_out;
  if (pl.splitk > 1 && !d.vec) d.vec_out = 0;                      // (slabs with ragged rows: the scalar reduce)
  int rc;
  d.zero = zero_block();
  GAD_CHECK(d.zero != nullptr, "gad_hgemm: cannot resolve the zero block's device address");
  if (pl.tile == 1) rc = hgemm_launch<2, 2, 2, 2, 64, 2, false>(d, st);
  else if (pl.tile == 2) rc = hgemm_launch<2, 2, 2, 5, 32, 2, false>(d, st);
  else if (pl.tile == 5) rc = hgemm_launch<4, 2, 2, 5, 32, 4, true>(d, st);
  else if (pl.tile == 6) rc = hgemm_launch<4, 2, 2, 5, 32, 4, true, true>(d, st);       // A/B: the eight-wave form on 16x16x32 MFMAs
  else if (pl.tile == 7) rc = hgemm_launch<2, 2, 2, 5, 32, 2, false, true>(d, st);      // 128 x 320 on 16x16x32 MFMAs
  else if (pl.tile == 9) rc = hgemm_launch<2, 2, 2, 2, 64, 2, false, true>(d, st);      // 128 x 128, 64-deep steps, 16x16x32 MFMAs
  else if (pl.tile == 8) rc = hgemm_launch<2, 2, 2, 2, 32, 2, false, true>(d, st);      // 128 x 128, 32-deep steps, 16x16x32 MFMAs: 32 KB of LDS, four workgroups per CU
  else { gad_set_error("gad_hgemm: tile_hint %d", pl.tile); return 1; }
  if (rc) return rc;
  if (pl.splitk > 1) {
    const long total = d.vec_out ? (long)a->M * a->N / 8 : (long)a->M * a->N;
    hipLaunchKernelGGL(hgemm_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d);
    GAD_LAUNCH_CHECK("hgemm_reduce_kernel");
  }
  return 0;
}

extern "C" int gad_h_transpose(const void* src, void* dst, int32_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, int32_t src_f32,
                               void* stream) {
  GAD_CHECK(src && dst && rows > 0 && cols > 0 && ld_src >= cols && ld_dst >= rows, "gad_h_transpose: bad arguments");
  const dim3 grid((cols + 63) / 64, (rows + 63) / 64);
  if (!src_f32 && rows % 8 == 0 && cols % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && gad_aligned16(src) && gad_aligned16(dst)) {
    hipLaunchKernelGGL(transpose16_kernel, grid, dim3(256), 0, (hipStream_t)stream, (const u16*)src, (u16*)dst, rows, cols, ld_src, ld_dst);
    GAD_LAUNCH_CHECK("h_transpose");
    return 0;
  }
  if (src_f32) hipLaunchKernelGGL(transpose_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, rows, cols, ld_src, ld_dst);
  else hipLaunchKernelGGL(transpose_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, rows, cols, ld_src, ld_dst);
  GAD_LAUNCH_CHECK("h_transpose");
  return 0;
}
static unsigned ew_grid(long work) {
  long g = (work + 255) / 256;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (unsigned)g;
}
extern "C" int gad_h_shadow_pairs(const float* src, void* dst, void* dst_t, const int64_t* table, int32_t n_tiles, void* stream) {
  GAD_CHECK(src && dst && dst_t && table && n_tiles > 0, "gad_h_shadow_pairs: bad arguments");
  static_assert(sizeof(long) == sizeof(int64_t), "table rows are 64-bit");
  hipLaunchKernelGGL(shadow_pair_kernel, dim3((unsigned)n_tiles), dim3(256), 0, (hipStream_t)stream, src, (u16*)dst, (u16*)dst_t, (const long*)table);
  GAD_LAUNCH_CHECK("h_shadow_pairs");
  return 0;
}
extern "C" int gad_h_cast(const void* src, void* dst, int64_t n, int32_t to_f32, void* stream) {
  GAD_CHECK(src && dst && n > 0 && gad_aligned16(src) && gad_aligned16(dst), "gad_h_cast: bad arguments");
  if (to_f32) hipLaunchKernelGGL(cast_to_f32_kernel, dim3(ew_grid((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const u16*)src, (float*)dst, (long)n);
  else hipLaunchKernelGGL(cast_to_bf16_kernel, dim3(ew_grid((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const float*)src, (u16*)dst, (long)n);
  GAD_LAUNCH_CHECK("h_cast");
  return 0;
}
extern "C" int gad_h_add(const void* a, const void* b, void* out, int64_t n, void* stream) {
  GAD_CHECK(a && b && out && n > 0 && gad_aligned16(a) && gad_aligned16(b) && gad_aligned16(out), "gad_h_add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(ew_grid((n + 7) / 8)), dim3(256), 0, (hipStream_t)stream, (const u16*)a, (const u16*)b, (u16*)out, (long)n);
  GAD_LAUNCH_CHECK("h_add");
  return 0;
}
extern "C" int gad_h_upsample2x_bwd(const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream) {
  GAD_CHECK(dy && dx && B > 0 && H > 0 && W > 0 && C > 0 && C % 8 == 0, "gad_h_upsample2x_bwd: bad arguments (C %% 8)");
  const long total = (long)B * H * W * (C / 8);
  hipLaunchKernelGGL(upsample2x_bwd_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const u16*)dy, (u16*)dx, B, H, W, C / 8);
  GAD_LAUNCH_CHECK("h_upsample2x_bwd");
  return 0;
}
extern "C" int gad_h_geglu_fwd(const void* h, void* out, int64_t M, int32_t F, void* stream) {
  GAD_CHECK(h && out && M > 0 && F > 0 && F % 8 == 0, "gad_h_geglu_fwd: bad arguments (F %% 8)");
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(ew_grid(M * (F / 8))), dim3(256), 0, (hipStream_t)stream, (const u16*)h, (u16*)out, (long)M, F / 8);
  GAD_LAUNCH_CHECK("h_geglu_fwd");
  return 0;
}
extern "C" int gad_h_geglu_bwd(const void* h, const void* dout, void* dh, int64_t M, int32_t F, void* stream) {
  GAD_CHECK(h && dout && dh && M > 0 && F > 0 && F % 8 == 0, "gad_h_geglu_bwd: bad arguments (F %% 8)");
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(ew_grid(M * (F / 8))), dim3(256), 0, (hipStream_t)stream, (const u16*)h, (const u16*)dout, (u16*)dh, (long)M,
                     F / 8);
  GAD_LAUNCH_CHECK("h_geglu_bwd");
  return 0;
}
extern "C" int gad_h_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd, int64_t rows,
                                   int32_t C, float eps, void* stream) {
  GAD_CHECK(x && y && gamma && beta && mean && rstd && rows > 0, "gad_h_layernorm_fwd: null / empty");
  GAD_CHECK(C % 8 == 0 && C / 8 <= 64 * LN_MAXO, "gad_h_layernorm_fwd: C must be a multiple of 8, <= %d", 512 * LN_MAXO);
  hipLaunchKernelGGL(ln_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, (u16*)y, gamma, beta, mean, rstd,
                     (long)rows, C, eps);
  GAD_LAUNCH_CHECK("h_layernorm_fwd");
  return 0;
}
extern "C" int gad_h_layernorm_bwd(const void* x, const void* dy, void* dx, const void* dx_add, const float* gamma, const float* mean,
                                   const float* rstd, int64_t rows, int32_t C, void* stream) {
  GAD_CHECK(x && dy && dx && gamma && mean && rstd && rows > 0, "gad_h_layernorm_bwd: null / empty");
  GAD_CHECK(C % 8 == 0 && C / 8 <= 64 * LN_MAXO, "gad_h_layernorm_bwd: C must be a multiple of 8, <= %d", 512 * LN_MAXO);
  hipLaunchKernelGGL(ln_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const u16*)x, (const u16*)dy, (u16*)dx,
                     (const u16*)dx_add, gamma, mean, rstd, (long)rows, C);
  GAD_LAUNCH_CHECK("h_layernorm_bwd");
  return 0;
}

extern "C" int64_t gad_h_groupnorm_workspace_bytes(const gad_groupnorm_args* a) {
  if (!a || a->C <= 0 || a->C % 8 || a->B <= 0 || a->HW <= 0 || a->G <= 0) return -1;
  int rpc, no, rp;
  const int64_t chunks = gn_chunks(a, &rpc, &no, &rp);
  return (int64_t)a->B * chunks * a->G * 2 * 4;
}
template <bool BWD>
static int gn_run(const gad_groupnorm_args* a, hipStream_t st, const char* who) {
  if (gn_check(a, who)) return 1;
  const GnDev d = gn_dev(a);
  const dim3 grid(d.chunks, d.B), block(d.NO * d.RP);
  hipLaunchKernelGGL(gn_part_kernel<BWD>, grid, block, (size_t)d.RP * d.NO * 16 * 4, st, d);
  GAD_LAUNCH_CHECK("h_gn_part");
  const int parts = d.NO * d.RP / d.G > 0 ? d.NO * d.RP / d.G : 1;
  hipLaunchKernelGGL(gn_apply_kernel<BWD>, grid, block, (size_t)(1 + parts) * d.G * 2 * 4, st, d);
  GAD_LAUNCH_CHECK("h_gn_apply");
  return 0;
}
extern "C" int gad_h_groupnorm_silu_fwd(const gad_groupnorm_args* a, void* stream) {
  return gn_run<false>(a, (hipStream_t)stream, "gad_h_groupnorm_silu_fwd");
}
extern "C" int gad_h_groupnorm_silu_bwd(const gad_groupnorm_args* a, void* stream) {
  GAD_CHECK(a && a->dy, "gad_h_groupnorm_silu_bwd: dy is NULL");
  GAD_CHECK(!a->dgamma && !a->dbeta, "gad_h_groupnorm_silu_bwd: the half path computes dx only (frozen norm)");
  GAD_CHECK(!a->x2, "gad_h_groupnorm_silu_bwd: single source only");
  return gn_run<true>(a, (hipStream_t)stream, "gad_h_groupnorm_silu_bwd");
}

// ---- token-axis contraction (both operands [k][.]) ----
static int tn_check(const gad_hgemm_args* a) {
  GAD_CHECK(a && a->A && a->B && a->C, "gad_hgemm_tn: null operand");
  GAD_CHECK(a->M > 0 && a->N > 0 && a->K > 0, "gad_hgemm_tn: empty problem %d x %d x %d", a->M, a->N, a->K);
  GAD_CHECK(a->lda % 8 == 0 && a->ldb % 8 == 0 && a->lda >= (a->M + 7) / 8 * 8 && a->ldb >= (a->N + 7) / 8 * 8,
            "gad_hgemm_tn: row strides must be multiples of 8 covering M / N rounded up to 8 (rows are read in 16-byte chunks)");
  GAD_CHECK(gad_aligned16(a->A) && gad_aligned16(a->B), "gad_hgemm_tn: operands must be 16-byte aligned");
  GAD_CHECK((long)a->K * (a->lda > a->ldb ? a->lda : a->ldb) < (1L << 40), "gad_hgemm_tn: operand too large");
  GAD_CHECK(a->out_f32 && !a->bias && !a->rowadd && !a->residual && !a->A2 && !a->B2 && !a->conv,
            "gad_hgemm_tn: fp32 output, no epilogue operands, no second operand pair, no gather");
  GAD_CHECK(a->ldc >= a->N, "gad_hgemm_tn: ldc < N");
  return 0;
}
static int tn_splitk(const gad_hgemm_args* a, int* steps_out) {
  const int steps = (a->K + 63) / 64;
  const long tiles = (long)((a->M + 127) / 128) * ((a->N + 127) / 128);
  int sk = a->splitk_hint > 0 ? a->splitk_hint : (int)(512 / tiles);
  if (a->splitk_hint <= 0 && sk > steps / 4) sk = steps / 4;            // at least 256 of K per slice
  if (sk > 128) sk = 128;
  if (sk > steps) sk = steps;
  if (sk < 1) sk = 1;
  const int per = (steps + sk - 1) / sk;
  *steps_out = steps;
  return (steps + per - 1) / per;
}
extern "C" int64_t gad_hgemm_tn_workspace_bytes(const gad_hgemm_args* a) {
  if (tn_check(a)) return -1;
  int steps;
  const int sk = tn_splitk(a, &steps);
  return sk > 1 ? (int64_t)sk * a->M * a->N * 4 : 0;
}
extern "C" int gad_hgemm_tn(const gad_hgemm_args* a, void* stream) {
  if (tn_check(a)) return 1;
  int steps;
  const int sk = tn_splitk(a, &steps);
  if (sk > 1) GAD_CHECK(a->ws && a->ws_bytes >= (int64_t)sk * a->M * a->N * 4, "gad_hgemm_tn: split-K workspace too small");
  HDev d{};
  d.A = (const u16*)a->A; d.B = (const u16*)a->B;
  d.M = a->M; d.N = a->N; d.lda = a->lda; d.ldb = a->ldb; d.len1 = a->K;
  d.alpha = a->alpha; d.C = a->C; d.ldc = a->ldc; d.out_f32 = 1; d.accumulate = a->accumulate;
  d.ws = (float*)a->ws;
  d.tiles_m = (a->M + 127) / 128; d.tiles_n = (a->N + 127) / 128; d.splitk = sk; d.steps = steps;
  d.steps_per_split = (steps + sk - 1) / sk;
  d.vec_out = a->N % 8 == 0 && gad_aligned16(a->C) && a->ldc % 4 == 0 && (sk == 1 || gad_aligned16(a->ws));
  d.vec = d.vec_out;
  d.dbg_zero = a->tile_hint >= 100;
  d.zero = zero_block();
  GAD_CHECK(d.zero != nullptr, "gad_hgemm_tn: cannot resolve the zero block's device address");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(hgemm_tn_kernel, dim3(d.tiles_m * d.tiles_n, sk), dim3(256), 0, st, d);
  GAD_LAUNCH_CHECK("hgemm_tn_kernel");
  if (sk > 1) {
    const long total = d.vec_out ? (long)a->M * a->N / 8 : (long)a->M * a->N;
    hipLaunchKernelGGL(hgemm_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, d);
    GAD_LAUNCH_CHECK("hgemm_reduce_kernel");
  }
  return 0;
}
