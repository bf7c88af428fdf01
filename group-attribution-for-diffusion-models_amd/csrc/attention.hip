// Fused (flash-style) attention for gfx950: softmax(scale * Q K^T) V and its gradients with the score matrix kept in
// registers - S = [B, heads, Tq, Tk] never exists in HBM.
//
// Replaces F.scaled_dot_product_attention inside AttnProcessor2_0
// (reference src/diffusers/models/attention_processor.py:1314-1325) for UNet2DModel's attention blocks (1 head of 256 on
// CIFAR, heads of 32 on CelebA-HQ) and for the self / cross attentions of the Stable-Diffusion U-Net (head dims 40 / 80 /
// 160, Tk = Tq or 77; text_to_image/train_text_to_image_lora.py:1268-1270).
//
// Exact-fp32 arithmetic on v_mfma_f32_16x16x4_f32 (16-wide tiles: head dim 40 pads to 48, not 64).  One workgroup = 4
// waves = one block of queries (forward, dQ) or keys (dK/dV) of one (batch, head); it streams the other side through
// double-buffered LDS tiles filled by LDS-DMA (global_load_lds_dwordx4; rows padded to D+4 floats, zeros beyond D / T).
//
// Orientation trick (cdna_hip_programming.md, "an accumulator tile as the next MFMA's operand"): the first product is
// computed TRANSPOSED so that its 16x16 accumulator (col = lane & 15, row = 4 (lane >> 4) + reg) is, register by
// register, already the B operand (B[k = lane >> 4][j = lane & 15]) of the product that contracts over its row index:
//   forward   S^T[key][q] = K Q^T  ->  P^T          ->  O^T[dv][q]  += V^T[dv][key] P^T[key][q]
//   dQ        S^T, dP^T[key][q] = V dO^T -> dS^T    ->  dQ^T[k][q]  += K^T[k][key] dS^T[key][q]
//   dK / dV   S[q][key] = Q K^T, dP = dO V^T -> P,dS ->  dV^T[dv][key] += dO^T[dv][q] P[q][key];  dK^T += Q^T dS
// so probabilities never touch LDS, and the softmax statistics of a query (running max, sum, LSE, delta) are per-lane
// scalars wherever the query sits on the lane index.  Row maxima need two cross-lane exchanges (lanes l, l^16, l^32).
//
// Backward, head dims up to 96: ONE kernel per key block computes S and dP once and feeds dV, dK and dQ (the five products
// of the minimal scheme); dQ partials of the key blocks of one (b, h) go to workspace slabs that a fixed-order reduce sums.
// Wider heads (and bf16 operands) run two kernels - dQ by query block, dK/dV by key block - that each recompute S and dP
// (7 products).  Either way every output element is produced in a fixed order - no float atomics, so a training step
// stays bit-reproducible (tests/test_gpu_fullsize.py relies on that).
//
// Any head dim <= 256 and any alignment: launches that cannot stream 16-byte pieces (the pruned CelebA model's d = 23 in
// 322-float rows) run the `RG` instances of the same kernels - dword staging through registers, masks at d.
#include "gad_common.h"

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NT = 256;     // threads per workgroup (4 waves)
constexpr int KV = 32;      // streamed rows per LDS tile

__device__ __attribute__((aligned(64))) float g_attn_zero[16];

__device__ __forceinline__ void glds16(const float* src, float* dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)dst_wave_uniform, 16, 0, 0);
}
// publish LDS-DMA'd tiles: the DMA is a VMEM operation -> explicit vmcnt(0) before the barrier
__device__ __forceinline__ void barrier_after_dma() {
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __syncthreads();
}

template <int D>
struct Cfg {
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  static constexpr int DP = (D + 15) / 16 * 16;    // head dim padded to whole 16-wide output tiles
  static constexpr int S = DP + 4;                 // LDS row stride (floats): == 4 (mod 8) -> column reads conflict-free
  static constexpr int G16 = D / 16;               // 16-k groups read as one float4 per lane
  static constexpr bool TAIL8 = (D % 16) == 8;     // + one 8-k group read as float2 per lane
  static constexpr int KS = D / 4;                 // MFMA 16x16x4 steps over the head dim
  static constexpr int NDV = DP / 16;              // 16-wide tiles over the head dim
  static constexpr int TILE = (KV * S * 4 + 1023) / 1024 * 256;   // floats per LDS tile (whole 1 KiB DMA pieces)
  static constexpr int NPIECE = TILE / 256;
};

struct AttnDev {
  const float* q; const float* k; const float* v; float* o; float* lse;
  const float* d_o; const float* delta; float* dq; float* dk; float* dv;
  int B, heads, Tq, Tk;
  int d;                                                 // the head dim itself (<= the instance's D; < D only in RG kernels)
  int nblk;                                              // query / key blocks per (b, h): blockIdx.x = blk + nblk * (b * heads + h)
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;       // row strides (floats)
  long sq, sk, sv, so, sdo, sdq, sdk, sdv;              // batch strides (floats)
  float scale;                                           // 1/sqrt(d)
  float* ws;                                             // single-pass backward: dQ partial slabs [key block][B][Tq][heads*d]
};

// Fill a [KV][S] row-major tile with rows row0 .. row0+KV-1 of `base` (row stride ld, D valid columns, T valid rows);
// everything else (pad columns, rows past T, the tail of the last 1 KiB piece) reads the zero block.
template <int D>
__device__ __forceinline__ void stage_tile(float* tile, const float* base, int ld, int row0, int T) {
  using C = Cfg<D>;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < (C::NPIECE + 3) / 4; ++i) {
    const int piece = wave + 4 * i;
    if (piece < C::NPIECE) {
      const int f = (piece * 64 + lane) * 4;
      const int row = f / C::S, col = f - row * C::S;
      const bool ok = row < KV && row0 + row < T && col < D;
      const float* src = ok ? base + (long)(row0 + row) * ld + col : (const float*)g_attn_zero;
      glds16(src, tile + piece * 256);
    }
  }
}

// Lean form of stage_tile for the K / V streams of a kernel's main loop (every vector-ALU instruction there costs MFMA issue
// time, tools/micro/mfma_loop_model.hip; the general form spends ~12 per DMA piece on f / S, the validity test and a
// 64-bit select against the zero block).  The piece -> (row, column) map of a lane is fixed for the workgroup: `plan`
// keeps each piece's element offset (row * ld + column, or -1 for pad columns / the tail of the last piece, which stay on
// the zero block); a tile whose rows are all inside T then needs one wave-uniform row offset per piece.  A ragged last
// tile (Tk = 77) takes the general form.
template <int D>
struct TilePlan {
  static constexpr int NP = (Cfg<D>::NPIECE + 3) / 4;
  int off[NP];
  __device__ __forceinline__ void init(int ld) {
    using C = Cfg<D>;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int piece = wave + 4 * i;
      const int f = (piece * 64 + lane) * 4;
      const int row = f / C::S, col = f - row * C::S;
      off[i] = (piece < C::NPIECE && row < KV && col < D) ? row * ld + col : -1;
    }
  }
};
template <int D>
__device__ __forceinline__ void stage_tile_lean(float* tile, const float* base, int ld, int row0, int T, const TilePlan<D>& plan) {
  using C = Cfg<D>;
  if (row0 + KV > T) {             // ragged: rows past T must read zeros
    stage_tile<D>(tile, base, ld, row0, T);
    return;
  }
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float* rowbase = base + (long)row0 * ld;                 // wave-uniform
#pragma unroll
  for (int i = 0; i < TilePlan<D>::NP; ++i) {
    const int piece = wave + 4 * i;
    if (piece < C::NPIECE) {
      const float* src = plan.off[i] >= 0 ? rowbase + plan.off[i] : (const float*)g_attn_zero;
      glds16(src, tile + piece * 256);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// RG ("ragged") instances: any head dim d <= D and any alignment - the head-grouped-pruned CelebA model keeps its heads
// and shrinks the head dim 32 -> 23 (reference unconditional_generation/prune.py:337-342): rows of 14 x 23 = 322 floats,
// heads starting at odd offsets, nothing float4-aligned.  The kernels are the same; what changes is how operands reach
// them: streamed tiles go global -> registers (dword loads, issued before the tile's MFMAs) -> ds_write_b32 (after them)
// into the same [KV][S] image, columns d .. S-1 zeroed once; loop-invariant row fragments and the outputs use dword
// accesses masked at d.  `Stream<D, RG>` hides the difference from the kernel bodies.
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool RG>
struct Stream;
template <int D>
struct Stream<D, false> {
  TilePlan<D> plan;
  __device__ __forceinline__ void init(int ld) { plan.init(ld); }
  __device__ __forceinline__ void issue(float* tile, const float* base, int ld, int row0, int T, int) {
    stage_tile_lean<D>(tile, base, ld, row0, T, plan);
  }
  __device__ __forceinline__ void commit(float*) {}
};
template <int D>
struct Stream<D, true> {
  static constexpr int NE = (KV * D + NT - 1) / NT;
  // up to D = 96 the next tile's elements wait in registers while the current tile is computed on; wider heads have no
  // registers to spare (dK/dV at D = 256 holds 128 accumulator registers + 128 of fragments), so they fetch and write
  // after the MFMAs - a correct, slower form for shapes no reference model has
  static constexpr bool HOLD = D <= 96;
  float r[HOLD ? NE : 1];
  const float* src_;
  int ld_, row0_, T_, d_;
  __device__ __forceinline__ void init(int) {}
  __device__ __forceinline__ float fetch(int i, const float* base, int ld, int row0, int T, int d) const {
    const int e = threadIdx.x + NT * i;
    const int row = e / D, col = e - row * D;
    const bool ok = e < KV * D && row0 + row < T && col < d;
    return ok ? base[(long)(row0 + row) * ld + col] : 0.f;
  }
  __device__ __forceinline__ void issue(float*, const float* base, int ld, int row0, int T, int d) {
    if (HOLD) {
#pragma unroll
      for (int i = 0; i < NE; ++i) r[i] = fetch(i, base, ld, row0, T, d);
    } else {
      src_ = base; ld_ = ld; row0_ = row0; T_ = T; d_ = d;
    }
  }
  __device__ __forceinline__ void commit(float* tile) {
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = threadIdx.x + NT * i;
      const int row = e / D, col = e - row * D;
      const float v = HOLD ? r[HOLD ? i : 0] : fetch(i, src_, ld_, row0_, T_, d_);
      if (e < KV * D) tile[row * Cfg<D>::S + col] = v;
    }
  }
};
// first tiles of the two streams into buffer 0 (RG: the whole LDS is zeroed first - pad columns stay zero for good)
template <int D, bool RG>
__device__ __forceinline__ void stream_prologue(float* lds, Stream<D, RG>& a, Stream<D, RG>& b, const float* A, int lda,
                                                const float* Bp, int ldb, int T, int d) {
  using C = Cfg<D>;
  if (RG) {
    for (int i = threadIdx.x * 4; i < 4 * C::TILE; i += NT * 4) *reinterpret_cast<f32x4*>(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
  }
  a.issue(lds, A, lda, 0, T, d);
  b.issue(lds + C::TILE, Bp, ldb, 0, T, d);
  a.commit(lds);
  b.commit(lds + C::TILE);
  barrier_after_dma();
}
// four consecutive head-dim columns col0 .. col0+3 of one output row (col0 % 4 == 0)
template <int D, bool RG>
__device__ __forceinline__ void store_cols(float* rowptr, int col0, const f32x4& v, int d) {
  if (!RG) {
    if (col0 < D) *reinterpret_cast<f32x4*>(rowptr + col0) = v;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (col0 + j < d) rowptr[col0 + j] = v[j];
  }
}

// "row fragment": lanes along the tile's ROWS (lane l: row r0 + (l & 15), k slot g = l >> 4).  Step s = 4 grp + j uses
// k = 16 grp + 4 g + j (float4 per group), the 8-wide tail k = 16 G16 + 2 g + j (float2).  The SAME k assignment is
// used for both operands of a product, which is all a contraction needs.
template <int D>
__device__ __forceinline__ void row_frag_lds(const float* tile, int r0, float (&f)[Cfg<D>::KS], int stride = Cfg<D>::S) {
  using C = Cfg<D>;
  const int lane = threadIdx.x & 63, r = r0 + (lane & 15), g = lane >> 4;
  const float* p = tile + r * stride;
#pragma unroll
  for (int grp = 0; grp < C::G16; ++grp) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(p + 16 * grp + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[4 * grp + j] = v[j];
  }
  if (C::TAIL8) {
    const f32x2 v = *reinterpret_cast<const f32x2*>(p + 16 * C::G16 + 2 * g);
    f[4 * C::G16] = v[0];
    f[4 * C::G16 + 1] = v[1];
  }
}
// the same fragment straight from global memory (loop-invariant operands: one load per workgroup lifetime)
template <int D, bool RG = false>
__device__ __forceinline__ void row_frag_global(const float* base, int ld, int row, bool ok, float mul, float (&f)[Cfg<D>::KS], int d = D) {
  using C = Cfg<D>;
  const int g = (threadIdx.x & 63) >> 4;
  const float* p = base + (long)(ok ? row : 0) * ld;
  if (RG) {                      // dword loads, zero at and beyond d
#pragma unroll
    for (int grp = 0; grp < C::G16; ++grp)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int col = 16 * grp + 4 * g + j;
        f[4 * grp + j] = (ok && col < d) ? p[col] * mul : 0.f;
      }
    if (C::TAIL8) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = 16 * C::G16 + 2 * g + j;
        f[4 * C::G16 + j] = (ok && col < d) ? p[col] * mul : 0.f;
      }
    }
    return;
  }
#pragma unroll
  for (int grp = 0; grp < C::G16; ++grp) {
    f32x4 v = *reinterpret_cast<const f32x4*>(p + 16 * grp + 4 * g);
#pragma unroll
    for (int j = 0; j < 4; ++j) f[4 * grp + j] = ok ? v[j] * mul : 0.f;
  }
  if (C::TAIL8) {
    f32x2 v = *reinterpret_cast<const f32x2*>(p + 16 * C::G16 + 2 * g);
    f[4 * C::G16] = ok ? v[0] * mul : 0.f;
    f[4 * C::G16 + 1] = ok ? v[1] * mul : 0.f;
  }
}

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float xmax16_32(float v) {        // max over the 4 lanes {l, l^16, l^32, l^48}
  v = fmaxf(v, __shfl_xor(v, 16, 64));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xsum16_32(float v) {
  v += __shfl_xor(v, 16, 64);
  return v + __shfl_xor(v, 32, 64);
}

constexpr float LOG2E = 1.4426950408889634f;
constexpr float NEG_BIG = -1.0e30f;
constexpr float RESCALE_SLACK = 8.f;     // exp2 domain: the running reference may lag the running max by up to 8 (x256)

// ------------------------------------------------------------------------------------------------------------------
// forward: workgroup = 64 NQ queries of one (b, h); wave w owns queries [q0 + 16 NQ w, + 16 NQ)
// ------------------------------------------------------------------------------------------------------------------
template <int D, int NQ, bool RG = false>
__global__ __launch_bounds__(NT) void attn_fwd_f32_kernel(const AttnDev p) {
  using C = Cfg<D>;
  extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][K tile | V tile]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const float* Q = p.q + b * p.sq + h * p.d;
  const float* K = p.k + b * p.sk + h * p.d;
  const float* V = p.v + b * p.sv + h * p.d;
  const int qw = blk * (64 * NQ) + wave * (16 * NQ);

  float qf[NQ][C::KS];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const int row = qw + 16 * t + c;
    row_frag_global<D, RG>(Q, p.ldq, row, row < p.Tq, p.scale * LOG2E, qf[t], p.d);     // scores in the exp2 domain
  }
  f32x4 o[C::NDV][NQ];
#pragma unroll
  for (int i = 0; i < C::NDV; ++i)
#pragma unroll
    for (int t = 0; t < NQ; ++t) o[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[NQ], l[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) { m[t] = NEG_BIG; l[t] = 0.f; }

  const int ntiles = (p.Tk + KV - 1) / KV;
  Stream<D, RG> ks, vs;
  ks.init(p.ldk);
  vs.init(p.ldv);
  stream_prologue<D, RG>(lds, ks, vs, K, p.ldk, V, p.ldv, p.Tk, p.d);

  for (int it = 0; it < ntiles; ++it) {
    const float* kt_ = lds + (it & 1) * (2 * C::TILE);
    const float* vt_ = kt_ + C::TILE;
    float* const nb = lds + ((it + 1) & 1) * (2 * C::TILE);
    const bool more = it + 1 < ntiles;
    if (more) {                       // the other buffer was last read before the barrier that ended iteration it-1
      ks.issue(nb, K, p.ldk, (it + 1) * KV, p.Tk, p.d);
      vs.issue(nb + C::TILE, V, p.ldv, (it + 1) * KV, p.Tk, p.d);
    }
    // S^T[key][q] (exp2 domain), two key tiles of 16
    f32x4 s[2][NQ];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float kf[C::KS];
      row_frag_lds<D>(kt_, 16 * kt, kf);
#pragma unroll
      for (int t = 0; t < NQ; ++t) s[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < C::KS; ++st)
#pragma unroll
        for (int t = 0; t < NQ; ++t) s[kt][t] = mfma16(kf[st], qf[t][st], s[kt][t]);
    }
    if ((it + 1) * KV > p.Tk) {       // ragged last tile (Tk = 77): keys past the end do not take part
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (it * KV + 16 * kt + 4 * g + e >= p.Tk)
#pragma unroll
            for (int t = 0; t < NQ; ++t) s[kt][t][e] = -__builtin_inff();
    }
    // online softmax; query = lane & 15 -> running reference / sum / rescale factor are per-lane scalars.  The reference
    // m[t] is moved (and the accumulators rescaled - NDV accumulator tiles through the vector ALUs) only when some query
    // of the wave has a score more than RESCALE_SLACK above it: probabilities are then at most 2^RESCALE_SLACK instead of
    // 1, which fp32 carries without loss, and the softmax is the same function of the scores (LSE = m + log2 l as before).
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      float mx = fmaxf(fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3])),
                       fmaxf(fmaxf(s[1][t][0], s[1][t][1]), fmaxf(s[1][t][2], s[1][t][3])));
      mx = xmax16_32(mx);
      if (__builtin_amdgcn_ballot_w64(mx > m[t] + RESCALE_SLACK) != 0) {      // wave-uniform
        const float mn = fmaxf(m[t], mx);
        const float alpha = ex2(m[t] - mn);
        m[t] = mn;
        l[t] *= alpha;
#pragma unroll
        for (int i = 0; i < C::NDV; ++i) o[i][t] *= alpha;
      }
      const float mr = m[t];
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = ex2(s[kt][t][e] - mr);
          s[kt][t][e] = pe;
          ps += pe;
        }
      l[t] += ps;                     // per-lane partial sum (its 8 keys); the 4 partials of a query meet in the epilogue
    }
    // O^T[dv][q] += V^T[dv][key] P^T[key][q]: B operand = the probability registers as they stand
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* vrow = vt_ + (16 * kt + 4 * g + e) * C::S + c;
#pragma unroll
        for (int i = 0; i < C::NDV; ++i) {
          const float a = vrow[16 * i];
#pragma unroll
          for (int t = 0; t < NQ; ++t) o[i][t] = mfma16(a, s[kt][t][e], o[i][t]);
        }
      }
    if (RG && more) {
      ks.commit(nb);
      vs.commit(nb + C::TILE);
    }
    barrier_after_dma();              // next tiles landed; every wave is done with this buffer
  }

  float* O = p.o + b * p.so + h * p.d;
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const float lt = xsum16_32(l[t]);
    const float inv = 1.f / lt;
    const int row = qw + 16 * t + c;
    if (row < p.Tq) {
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) store_cols<D, RG>(O + (long)row * p.ldo, 16 * i + 4 * g, o[i][t] * inv, p.d);
      if (g == 0 && p.lse) p.lse[(long)bh * p.Tq + row] = m[t] + __builtin_amdgcn_logf(lt);   // v_log_f32 = log2
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// forward for WIDE heads (d >= 160: one head of 256 / 192 on CIFAR, 160 on SD's deepest level): the kernel above holds a
// query tile's whole head dim per wave (64 fragment + 64 accumulator registers at d = 256) and runs one wave per SIMD.
// Here a workgroup is 8 waves = two per SIMD: wave (w4 = wave & 3, dh = wave >> 2) owns 16 queries and ONE HALF of the head
// dim - half the contraction of S^T = K Q^T and half the output tiles of O^T = V^T P^T.  The two partial score tiles of a
// query block meet through LDS (8 floats per lane, summed half 0 + half 1 by both partners, so both hold bit-identical
// scores and take identical softmax decisions), each partner then multiplies the probabilities into its own dv half.
// Registers halve (<= 128: two waves per SIMD hide each other's LDS reads and softmax), the K / V tiles in LDS are shared by
// twice the waves.  Same arithmetic per element as the 4-wave kernel except the order of the score's k-sum (two halves).
// ------------------------------------------------------------------------------------------------------------------
constexpr int NTW = 512;
template <int D>
__global__ __launch_bounds__(NTW) void attn_fwd_wide_f32_kernel(const AttnDev p) {
  using C = Cfg<D>;
  using CH = Cfg<D / 2>;
  static_assert(D % 32 == 0, "both halves are whole 16-wide tiles");
  constexpr int KSH = C::KS / 2, NDVH = C::NDV / 2;
  extern __shared__ __attribute__((aligned(16))) float lds[];     // [2][K tile | V tile] | exchange [2][4][64][8]
  float* const xch = lds + 4 * C::TILE;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int w4 = wave & 3, dh = wave >> 2;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const float* Q = p.q + b * p.sq + h * p.d + dh * (D / 2);
  const float* K = p.k + b * p.sk + h * p.d;
  const float* V = p.v + b * p.sv + h * p.d;
  const int row = blk * 64 + w4 * 16 + c;

  float qf[KSH];
  row_frag_global<D / 2>(Q, p.ldq, row, row < p.Tq, p.scale * LOG2E, qf);
  f32x4 o[NDVH];
#pragma unroll
  for (int i = 0; i < NDVH; ++i) o[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = NEG_BIG, l = 0.f;

  // 8-wave staging: piece (wave + 8 i) of a tile, fixed (row, column) per lane as in TilePlan
  constexpr int NP8 = (C::NPIECE + 7) / 8;
  int koff[NP8], voff[NP8];
#pragma unroll
  for (int i = 0; i < NP8; ++i) {
    const int piece = wave + 8 * i;
    const int f = (piece * 64 + lane) * 4;
    const int r = f / C::S, col = f - r * C::S;
    const bool ok = piece < C::NPIECE && r < KV && col < D;
    koff[i] = ok ? r * p.ldk + col : -1;
    voff[i] = ok ? r * p.ldv + col : -1;
  }
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  auto stage = [&](float* buf, int row0) {
    const bool whole = row0 + KV <= p.Tk;
#pragma unroll
    for (int i = 0; i < NP8; ++i) {
      const int piece = wv + 8 * i;
      if (piece < C::NPIECE) {
        const int f = (piece * 64 + lane) * 4;
        const int r = f / C::S;
        const bool okk = koff[i] >= 0 && (whole || row0 + r < p.Tk);
        glds16(okk ? K + (long)row0 * p.ldk + koff[i] : (const float*)g_attn_zero, buf + piece * 256);
        glds16(okk ? V + (long)row0 * p.ldv + voff[i] : (const float*)g_attn_zero, buf + C::TILE + piece * 256);
      }
    }
  };
  const int ntiles = (p.Tk + KV - 1) / KV;
  stage(lds, 0);
  barrier_after_dma();

  for (int it = 0; it < ntiles; ++it) {
    const float* kt_ = lds + (it & 1) * (2 * C::TILE);
    const float* vt_ = kt_ + C::TILE;
    if (it + 1 < ntiles) stage(lds + ((it + 1) & 1) * (2 * C::TILE), (it + 1) * KV);
    // partial S^T over this wave's half of the head dim
    f32x4 s[2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      float kf[KSH];
      row_frag_lds<D / 2>(kt_ + dh * (D / 2), 16 * kt, kf, C::S);
      s[kt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < KSH; ++st) s[kt] = mfma16(kf[st], qf[st], s[kt]);
    }
    float* mine = xch + ((dh * 4 + w4) * 64 + lane) * 8;
    *reinterpret_cast<f32x4*>(mine) = s[0];
    *reinterpret_cast<f32x4*>(mine + 4) = s[1];
    __builtin_amdgcn_s_waitcnt(0xC07F);            // lgkmcnt(0) only: the next tile's LDS-DMA stays in flight across this barrier
    __builtin_amdgcn_s_barrier();
    {
      const float* h0 = xch + ((0 * 4 + w4) * 64 + lane) * 8;
      const float* h1 = xch + ((1 * 4 + w4) * 64 + lane) * 8;
      s[0] = *reinterpret_cast<const f32x4*>(h0) + *reinterpret_cast<const f32x4*>(h1);
      s[1] = *reinterpret_cast<const f32x4*>(h0 + 4) + *reinterpret_cast<const f32x4*>(h1 + 4);
    }
    if ((it + 1) * KV > p.Tk) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (it * KV + 16 * kt + 4 * g + e >= p.Tk) s[kt][e] = -__builtin_inff();
    }
    {
      float mx = fmaxf(fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3])), fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3])));
      mx = xmax16_32(mx);
      if (__builtin_amdgcn_ballot_w64(mx > m + RESCALE_SLACK) != 0) {
        const float mn = fmaxf(m, mx);
        const float alpha = ex2(m - mn);
        m = mn;
        l *= alpha;
#pragma unroll
        for (int i = 0; i < NDVH; ++i) o[i] *= alpha;
      }
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = ex2(s[kt][e] - m);
          s[kt][e] = pe;
          ps += pe;
        }
      l += ps;
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* vrow = vt_ + (16 * kt + 4 * g + e) * C::S + dh * (D / 2) + c;
#pragma unroll
        for (int i = 0; i < NDVH; ++i) o[i] = mfma16(vrow[16 * i], s[kt][e], o[i]);
      }
    barrier_after_dma();
  }
  float* O = p.o + b * p.so + h * p.d + dh * (D / 2);
  const float lt = xsum16_32(l);
  const float inv = 1.f / lt;
  if (row < p.Tq) {
#pragma unroll
    for (int i = 0; i < NDVH; ++i) *reinterpret_cast<f32x4*>(O + (long)row * p.ldo + 16 * i + 4 * g) = o[i] * inv;
    if (dh == 0 && g == 0 && p.lse) p.lse[(long)bh * p.Tq + row] = m + __builtin_amdgcn_logf(lt);
  }
}

// delta[b, h, q] = sum_dv dO[b, q, h, dv] * O[b, q, h, dv]
template <int D, bool RG = false>
__global__ void attn_delta_kernel(const AttnDev p, float* delta) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.B * p.Tq * p.heads;
  if (idx >= total) return;
  const int h = (int)(idx % p.heads);
  const long bq = idx / p.heads;
  const int q = (int)(bq % p.Tq), b = (int)(bq / p.Tq);
  const float* o = p.o + b * p.so + (long)q * p.ldo + h * p.d;
  const float* g = p.d_o + b * p.sdo + (long)q * p.lddo + h * p.d;
  float acc = 0.f;
  if (RG) {
    for (int i = 0; i < p.d; ++i) acc += o[i] * g[i];
  } else {
#pragma unroll
    for (int i = 0; i < D; i += 4) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(o + i), d = *reinterpret_cast<const f32x4*>(g + i);
      acc += a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
    }
  }
  delta[((long)b * p.heads + h) * p.Tq + q] = acc;
}

// ------------------------------------------------------------------------------------------------------------------
// dQ: workgroup = 64 queries of one (b, h), wave w owns 16; streams K / V tiles
// ------------------------------------------------------------------------------------------------------------------
template <int D, int NQ = 1, bool RG = false>
__global__ __launch_bounds__(NT) void attn_bwd_dq_f32_kernel(const AttnDev p) {
  // NQ query blocks of 16 per wave (workgroup = 64 NQ queries): a K / V fragment read from LDS, and every staged byte, feeds
  // NQ times the MFMAs - at d = 40 one block per wave is 64 MFMAs against ~230 other instructions per tile
  using C = Cfg<D>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const float* K = p.k + b * p.sk + h * p.d;
  const float* V = p.v + b * p.sv + h * p.d;
  int row[NQ];
  bool rok[NQ];
  float qf[NQ][C::KS], dof[NQ][C::KS], L2[NQ], dl[NQ];
  f32x4 dq[NQ][C::NDV];
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    row[t] = blk * (64 * NQ) + wave * (16 * NQ) + 16 * t + c;
    rok[t] = row[t] < p.Tq;
    row_frag_global<D, RG>(p.q + b * p.sq + h * p.d, p.ldq, row[t], rok[t], p.scale * LOG2E, qf[t], p.d);
    row_frag_global<D, RG>(p.d_o + b * p.sdo + h * p.d, p.lddo, row[t], rok[t], 1.f, dof[t], p.d);
    L2[t] = rok[t] ? p.lse[(long)bh * p.Tq + row[t]] : 0.f;
    dl[t] = rok[t] ? p.delta[(long)bh * p.Tq + row[t]] : 0.f;
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) dq[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  const int ntiles = (p.Tk + KV - 1) / KV;
  Stream<D, RG> ks, vs;
  ks.init(p.ldk);
  vs.init(p.ldv);
  stream_prologue<D, RG>(lds, ks, vs, K, p.ldk, V, p.ldv, p.Tk, p.d);
  for (int it = 0; it < ntiles; ++it) {
    const float* kt_ = lds + (it & 1) * (2 * C::TILE);
    const float* vt_ = kt_ + C::TILE;
    float* const nb = lds + ((it + 1) & 1) * (2 * C::TILE);
    const bool more = it + 1 < ntiles;
    if (more) {
      ks.issue(nb, K, p.ldk, (it + 1) * KV, p.Tk, p.d);
      vs.issue(nb + C::TILE, V, p.ldv, (it + 1) * KV, p.Tk, p.d);
    }
    f32x4 ds[2][NQ];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      f32x4 s[NQ], dp[NQ];
#pragma unroll
      for (int t = 0; t < NQ; ++t) { s[t] = f32x4{0.f, 0.f, 0.f, 0.f}; dp[t] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      {
        float kf[C::KS];
        row_frag_lds<D>(kt_, 16 * kt, kf);
#pragma unroll
        for (int st = 0; st < C::KS; ++st)
#pragma unroll
          for (int t = 0; t < NQ; ++t) s[t] = mfma16(kf[st], qf[t][st], s[t]);          // S^T[key][q]
      }
      {
        float vf[C::KS];
        row_frag_lds<D>(vt_, 16 * kt, vf);
#pragma unroll
        for (int st = 0; st < C::KS; ++st)
#pragma unroll
          for (int t = 0; t < NQ; ++t) dp[t] = mfma16(vf[st], dof[t][st], dp[t]);       // dP^T[key][q] = V dO^T
      }
      if ((it + 1) * KV <= p.Tk) {        // whole tile inside Tk (wave-uniform): no key masks
#pragma unroll
        for (int t = 0; t < NQ; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) ds[kt][t][e] = ex2(s[t][e] - L2[t]) * (dp[t][e] - dl[t]);
      } else {
#pragma unroll
        for (int t = 0; t < NQ; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool kok = it * KV + 16 * kt + 4 * g + e < p.Tk;
            const float pe = kok ? ex2(s[t][e] - L2[t]) : 0.f;
            ds[kt][t][e] = pe * (dp[t][e] - dl[t]);
          }
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* krow = kt_ + (16 * kt + 4 * g + e) * C::S + c;
#pragma unroll
        for (int i = 0; i < C::NDV; ++i) {
          const float a = krow[16 * i];
#pragma unroll
          for (int t = 0; t < NQ; ++t) dq[t][i] = mfma16(a, ds[kt][t][e], dq[t][i]);   // dQ^T[k][q] += K^T dS^T
        }
      }
    if (RG && more) {
      ks.commit(nb);
      vs.commit(nb + C::TILE);
    }
    barrier_after_dma();
  }
#pragma unroll
  for (int t = 0; t < NQ; ++t)
    if (rok[t]) {
      float* DQ = p.dq + b * p.sdq + h * p.d + (long)row[t] * p.lddq;
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) store_cols<D, RG>(DQ, 16 * i + 4 * g, dq[t][i] * p.scale, p.d);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dK / dV: workgroup = 64 keys of one (b, h), wave w owns 16; streams Q / dO tiles (+ LSE, delta of their rows)
// ------------------------------------------------------------------------------------------------------------------
template <int D, bool RG = false>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_f32_kernel(const AttnDev p) {
  using C = Cfg<D>;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const float* Q = p.q + b * p.sq + h * p.d;
  const float* DO = p.d_o + b * p.sdo + h * p.d;
  const int key = blk * 64 + wave * 16 + c;
  const bool kok = key < p.Tk;

  float kf[C::KS], vf[C::KS];
  row_frag_global<D, RG>(p.k + b * p.sk + h * p.d, p.ldk, key, kok, p.scale * LOG2E, kf, p.d);
  row_frag_global<D, RG>(p.v + b * p.sv + h * p.d, p.ldv, key, kok, 1.f, vf, p.d);
  f32x4 dk[C::NDV], dv[C::NDV];
#pragma unroll
  for (int i = 0; i < C::NDV; ++i) { dk[i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const float* lse = p.lse + (long)bh * p.Tq;
  const float* dlt = p.delta + (long)bh * p.Tq;
  const bool vec_stats = p.Tq % 4 == 0 && (reinterpret_cast<uintptr_t>(p.lse) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.delta) & 15) == 0;

  const int ntiles = (p.Tq + KV - 1) / KV;
  Stream<D, RG> qs, gs;
  qs.init(p.ldq);
  gs.init(p.lddo);
  stream_prologue<D, RG>(lds, qs, gs, Q, p.ldq, DO, p.lddo, p.Tq, p.d);
  for (int it = 0; it < ntiles; ++it) {
    const float* qt_ = lds + (it & 1) * (2 * C::TILE);
    const float* dot_ = qt_ + C::TILE;
    float* const nb = lds + ((it + 1) & 1) * (2 * C::TILE);
    const bool more = it + 1 < ntiles;
    if (more) {
      qs.issue(nb, Q, p.ldq, (it + 1) * KV, p.Tq, p.d);
      gs.issue(nb + C::TILE, DO, p.lddo, (it + 1) * KV, p.Tq, p.d);
    }
    f32x4 pr[2], ds[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
      {
        float qf[C::KS];
        row_frag_lds<D>(qt_, 16 * t, qf);
#pragma unroll
        for (int st = 0; st < C::KS; ++st) s = mfma16(qf[st], kf[st], s);              // S[q][key]
      }
      {
        float gf[C::KS];
        row_frag_lds<D>(dot_, 16 * t, gf);
#pragma unroll
        for (int st = 0; st < C::KS; ++st) dp = mfma16(gf[st], vf[st], dp);            // dP[q][key] = dO V^T
      }
      // the query of register e is row 4 g + e of the tile
      if ((it + 1) * KV <= p.Tq && vec_stats) {     // whole tile inside Tq (wave-uniform): LSE / delta as one float4 each
        const int q0 = it * KV + 16 * t + 4 * g;
        const f32x4 L4 = *reinterpret_cast<const f32x4*>(lse + q0), d4 = *reinterpret_cast<const f32x4*>(dlt + q0);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = ex2(s[e] - L4[e]);
          pr[t][e] = pe;
          ds[t][e] = pe * (dp[e] - d4[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int qrow = it * KV + 16 * t + 4 * g + e;
          const bool qok = qrow < p.Tq;
          const float L2 = qok ? lse[qrow] : 0.f, dl = qok ? dlt[qrow] : 0.f;
          const float pe = qok ? ex2(s[e] - L2) : 0.f;
          pr[t][e] = pe;
          ds[t][e] = pe * (dp[e] - dl);
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* grow = dot_ + (16 * t + 4 * g + e) * C::S + c;
        const float* qrow = qt_ + (16 * t + 4 * g + e) * C::S + c;
#pragma unroll
        for (int i = 0; i < C::NDV; ++i) {
          dv[i] = mfma16(grow[16 * i], pr[t][e], dv[i]);      // dV^T[dv][key] += dO^T[dv][q] P[q][key]
          dk[i] = mfma16(qrow[16 * i], ds[t][e], dk[i]);      // dK^T[k][key]  += Q^T[k][q]  dS[q][key]
        }
      }
    if (RG && more) {
      qs.commit(nb);
      gs.commit(nb + C::TILE);
    }
    barrier_after_dma();
  }
  if (kok) {
    float* DK = p.dk + b * p.sdk + h * p.d + (long)key * p.lddk;
    float* DV = p.dv + b * p.sdv + h * p.d + (long)key * p.lddv;
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) {
      store_cols<D, RG>(DK, 16 * i + 4 * g, dk[i] * p.scale, p.d);
      store_cols<D, RG>(DV, 16 * i + 4 * g, dv[i], p.d);
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Single-pass backward (head dims up to 96): ONE kernel computes S and dP once and feeds all three gradients - the five
// products of the minimal scheme (10 B h Tq Tk d FLOPs) instead of the seven the dQ + dK/dV pair executes.
//   workgroup = one block of KB = 64 NK keys of one (b, h); wave w owns key tiles j < NK at keys 16 (NK w + j); K and V
//   row fragments of those keys are loop-invariant registers, dK^T / dV^T accumulate in registers for the whole sweep
//   (as in the dK/dV kernel); Q / dO tiles of 32 queries stream through the double-buffered LDS tiles.
//   dQ contracts over KEYS - the lane index of the dS accumulators - so dS crosses LDS once: every wave writes its
//   [32 q][16 NK keys] block into a [32][KB] image; one barrier later (the loop's own: the image is double-buffered and
//   tile it-1's dQ is computed at the top of iteration it) the workgroup computes dQ^T[kdim][q] = K^T dS^T over all KB
//   keys with the output tiles dealt round-robin to the waves.  K sits in LDS TRANSPOSED ([kdim][key], written once per
//   workgroup), so both operands of that product are ds_read_b128 - four MFMA steps per read (a row-major K image costs a
//   dword read per MFMA: the LDS issue rate, not the matrix pipe, then paces the phase).
//   Key blocks of one (b, h) each hold a partial dQ: it goes to slab [block] of the caller's workspace and
//   `attn_dq_reduce_kernel` sums the slabs in block order - no atomics, every element written once in a fixed order, so a
//   training step stays bit-reproducible.  A single key block (cross attention, Tk = 77) writes dQ directly.
// ------------------------------------------------------------------------------------------------------------------
template <int D, int NK>
struct Bwd1 {
  using C = Cfg<D>;
  static constexpr int KB = 64 * NK;
  static constexpr int SQ = KB + 20;                 // dS image row stride: (4 g + e) * SQ covers both bank halves; 16-B aligned
  static constexpr int KIMG = (C::DP * SQ + 3) / 4 * 4;                  // floats: the block's K, TRANSPOSED: [kdim][key]
  static constexpr int DSIMG = 32 * SQ;
  static constexpr int LDS_FLOATS = KIMG + 4 * C::TILE + 2 * DSIMG;
  static constexpr int NOUT = 2 * C::NDV;            // dQ output tiles (16 kdim x 16 q) per 32-query tile
};

template <int D, int NK, bool RG>
__global__ __launch_bounds__(NT) void attn_bwd1_f32_kernel(const AttnDev p) {
  using C = Cfg<D>;
  using W = Bwd1<D, NK>;
  constexpr int KB = W::KB, SQ = W::SQ;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* const kimg = lds;                           // [DP][SQ]  K of this block, transposed: kimg[kdim][key] (zeros past Tk / d)
  float* const stream = lds + W::KIMG;               // [2][Q tile | dO tile]
  float* const dsimg = stream + 4 * C::TILE;         // [2][32 q][SQ]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const float* Q = p.q + b * p.sq + h * p.d;
  const float* DO = p.d_o + b * p.sdo + h * p.d;
  const float* Kp = p.k + b * p.sk + h * p.d;
  const float* Vp = p.v + b * p.sv + h * p.d;
  const int key0 = blk * KB + wave * (16 * NK);

  float kf[NK][C::KS], vf[NK][C::KS];
  bool kok[NK];
  f32x4 dk[NK][C::NDV], dv[NK][C::NDV];
#pragma unroll
  for (int j = 0; j < NK; ++j) {
    const int key = key0 + 16 * j + c;
    kok[j] = key < p.Tk;
    row_frag_global<D, RG>(Kp, p.ldk, key, kok[j], p.scale * LOG2E, kf[j], p.d);
    row_frag_global<D, RG>(Vp, p.ldv, key, kok[j], 1.f, vf[j], p.d);
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) { dk[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[j][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  const float* lse = p.lse + (long)bh * p.Tq;
  const float* dlt = p.delta + (long)bh * p.Tq;
  const bool vec_stats = p.Tq % 4 == 0 && (reinterpret_cast<uintptr_t>(p.lse) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.delta) & 15) == 0;

  // the K image (transposed): zero everything once (pad rows kdim >= d, keys past Tk), then scatter the block's K
  for (int i = threadIdx.x * 4; i < W::LDS_FLOATS; i += NT * 4) *reinterpret_cast<f32x4*>(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  if (RG) {
    for (int e = threadIdx.x; e < KB * D; e += NT) {
      const int row = e / D, col = e - row * D;
      if (blk * KB + row < p.Tk && col < p.d) kimg[col * SQ + row] = Kp[(long)(blk * KB + row) * p.ldk + col];
    }
  } else {
    for (int e = threadIdx.x; e < KB * (D / 4); e += NT) {
      const int row = e / (D / 4), col = (e - row * (D / 4)) * 4;
      if (blk * KB + row < p.Tk) {
        const f32x4 v4 = *reinterpret_cast<const f32x4*>(Kp + (long)(blk * KB + row) * p.ldk + col);
#pragma unroll
        for (int x = 0; x < 4; ++x) kimg[(col + x) * SQ + row] = v4[x];
      }
    }
  }
  const int ntiles = (p.Tq + KV - 1) / KV;
  Stream<D, RG> qs, gs;
  qs.init(p.ldq);
  gs.init(p.lddo);
  {                                  // first Q / dO tiles (the RG form zeroed the whole LDS above)
    qs.issue(stream, Q, p.ldq, 0, p.Tq, p.d);
    gs.issue(stream + C::TILE, DO, p.lddo, 0, p.Tq, p.d);
    qs.commit(stream);
    gs.commit(stream + C::TILE);
    barrier_after_dma();
  }

  // dQ of `ntq` (1 or 2) consecutive 32-query tiles starting at tile tq0 from the dS images they left behind (tile tq in
  // half tq & 1): output tile o = (i = o % NDV, t = o / NDV) with t counting 16-query sub-tiles -> wave o % 4.
  // PAIR (an odd number of 16-wide head-dim tiles: d = 40 / 48 has 3, d = 80 has 5): 2 NDV output tiles per query tile do
  // not split over four waves (6 -> 2, 2, 1, 1), so the product runs once per TWO query tiles (4 NDV tiles: 3 or 5 per
  // wave) right after the second one's barrier, plus one barrier that keeps the next tile's dS writes behind these reads.
  constexpr bool PAIR = (C::NDV % 2) == 1;
  const bool direct = p.nblk == 1;
  auto dq_phase = [&](int tq0, int ntq) {
    const int nout = 2 * ntq * C::NDV;
#pragma unroll
    for (int oo = 0; oo < (2 * (PAIR ? 2 : 1) * C::NDV + 3) / 4; ++oo) {
      const int o = wave + 4 * oo;
      if (o >= nout) break;                          // wave-uniform
      const int i = o % C::NDV, t = o / C::NDV, tq = tq0 + (t >> 1);
      const float* dsb = dsimg + (tq & 1) * W::DSIMG;
      f32x4 acc0 = f32x4{0.f, 0.f, 0.f, 0.f}, acc1 = f32x4{0.f, 0.f, 0.f, 0.f};      // two chains: the f32 MFMA's dependent latency
      const float* brow = dsb + (16 * (t & 1) + c) * SQ + 4 * g;    // dS[q = 16 (t & 1) + c][keys 16 m + 4 g ..+3]
      const float* arow = kimg + (16 * i + c) * SQ + 4 * g;         // K^T[kdim = 16 i + c][the same keys]
#pragma unroll 2
      for (int m = 0; m < KB / 16; m += 2) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(brow + 16 * m), b1 = *reinterpret_cast<const f32x4*>(brow + 16 * m + 16);
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(arow + 16 * m), a1 = *reinterpret_cast<const f32x4*>(arow + 16 * m + 16);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc0 = mfma16(a0[e], b0[e], acc0);
          acc1 = mfma16(a1[e], b1[e], acc1);
        }
      }
      const f32x4 acc = acc0 + acc1;
      const int q = tq * KV + 16 * (t & 1) + c;
      if (q < p.Tq) {
        if (direct) {
          store_cols<D, RG>(p.dq + b * p.sdq + h * p.d + (long)q * p.lddq, 16 * i + 4 * g, acc * p.scale, p.d);
        } else {       // slab [blk][b][q][heads*d], dense rows of heads*d floats
          const int wd = p.heads * p.d;
          float* row = p.ws + (((long)blk * p.B + b) * p.Tq + q) * wd + h * p.d;
          store_cols<D, RG>(row, 16 * i + 4 * g, acc, p.d);
        }
      }
    }
  };

  for (int it = 0; it < ntiles; ++it) {
    const float* qt_ = stream + (it & 1) * (2 * C::TILE);
    const float* dot_ = qt_ + C::TILE;
    float* const nb = stream + ((it + 1) & 1) * (2 * C::TILE);
    float* const dsw = dsimg + (it & 1) * W::DSIMG;
    const bool more = it + 1 < ntiles;
    if (more) {
      qs.issue(nb, Q, p.ldq, (it + 1) * KV, p.Tq, p.d);
      gs.issue(nb + C::TILE, DO, p.lddo, (it + 1) * KV, p.Tq, p.d);
    }
    if (!PAIR && it > 0) dq_phase(it - 1, 1);        // (deferred by one iteration: the loop's own barrier orders it)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x4 pr[NK], ds[NK];
      {
        float qf[C::KS], gf[C::KS];
        row_frag_lds<D>(qt_, 16 * t, qf);
        row_frag_lds<D>(dot_, 16 * t, gf);
        f32x4 L4, d4;
        const int q0 = it * KV + 16 * t + 4 * g;
        if ((it + 1) * KV <= p.Tq && vec_stats) {
          L4 = *reinterpret_cast<const f32x4*>(lse + q0);
          d4 = *reinterpret_cast<const f32x4*>(dlt + q0);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool qok = q0 + e < p.Tq;
            L4[e] = qok ? lse[q0 + e] : __builtin_inff();          // exp2(s - inf) = 0: rows past Tq contribute nothing
            d4[e] = qok ? dlt[q0 + e] : 0.f;
          }
        }
#pragma unroll
        for (int j = 0; j < NK; ++j) {
          f32x4 sacc = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int st = 0; st < C::KS; ++st) {
            sacc = mfma16(qf[st], kf[j][st], sacc);              // S[q][key]
            dp = mfma16(gf[st], vf[j][st], dp);                  // dP[q][key] = dO V^T
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float pe = kok[j] ? ex2(sacc[e] - L4[e]) : 0.f;     // keys past Tk: no probability, no dS
            pr[j][e] = pe;
            ds[j][e] = pe * (dp[e] - d4[e]);
          }
          // dS block -> image [q][key] for the dQ product of the next iteration
          float* drow = dsw + (16 * t + 4 * g) * SQ + 16 * (NK * wave + j) + c;
#pragma unroll
          for (int e = 0; e < 4; ++e) drow[e * SQ] = ds[j][e];
        }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float* grow = dot_ + (16 * t + 4 * g + e) * C::S + c;
        const float* qrow = qt_ + (16 * t + 4 * g + e) * C::S + c;
#pragma unroll
        for (int i = 0; i < C::NDV; ++i) {
          const float ag = grow[16 * i], aq = qrow[16 * i];
#pragma unroll
          for (int j = 0; j < NK; ++j) {
            dv[j][i] = mfma16(ag, pr[j][e], dv[j][i]);           // dV^T[dv][key] += dO^T[dv][q] P[q][key]
            dk[j][i] = mfma16(aq, ds[j][e], dk[j][i]);           // dK^T[k][key]  += Q^T[k][q]  dS[q][key]
          }
        }
      }
    }
    if (RG && more) {
      qs.commit(nb);
      gs.commit(nb + C::TILE);
    }
    barrier_after_dma();
    if (PAIR && (it & 1)) {
      dq_phase(it - 1, 2);
      __syncthreads();                               // the next tile's dS goes where these reads were
    }
  }
  if (!PAIR) dq_phase(ntiles - 1, 1);
  else if (ntiles & 1) dq_phase(ntiles - 1, 1);
#pragma unroll
  for (int j = 0; j < NK; ++j)
    if (kok[j]) {
      const int key = key0 + 16 * j + c;
      float* DK = p.dk + b * p.sdk + h * p.d + (long)key * p.lddk;
      float* DV = p.dv + b * p.sdv + h * p.d + (long)key * p.lddv;
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) {
        store_cols<D, RG>(DK, 16 * i + 4 * g, dk[j][i] * p.scale, p.d);
        store_cols<D, RG>(DV, 16 * i + 4 * g, dv[j][i], p.d);
      }
    }
}

// dq[b][q][:] = scale * sum over key blocks (in block order) of slab[block][b][q][:]
__global__ void attn_dq_reduce_kernel(const AttnDev p, int nslab, int vec) {
  const int wd = p.heads * p.d;
  const long rows = (long)p.B * p.Tq, slab = rows * wd;
  if (vec) {
    const int w4 = wd / 4;
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < rows * w4; idx += (long)gridDim.x * blockDim.x) {
      const long r = idx / w4;
      const int c4 = (int)(idx - r * w4) * 4;
      f32x4 acc = *reinterpret_cast<const f32x4*>(p.ws + r * wd + c4);
      for (int s = 1; s < nslab; ++s) acc += *reinterpret_cast<const f32x4*>(p.ws + s * slab + r * wd + c4);
      const long bq = r / p.Tq;
      *reinterpret_cast<f32x4*>(p.dq + bq * p.sdq + (r - bq * p.Tq) * p.lddq + c4) = acc * p.scale;
    }
  } else {
    for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < rows * wd; idx += (long)gridDim.x * blockDim.x) {
      const long r = idx / wd;
      const int cc = (int)(idx - r * wd);
      float acc = p.ws[r * wd + cc];
      for (int s = 1; s < nslab; ++s) acc += p.ws[s * slab + r * wd + cc];
      const long bq = r / p.Tq;
      p.dq[bq * p.sdq + (r - bq * p.Tq) * p.lddq + cc] = acc * p.scale;
    }
  }
}

// ==================================================================================================================
// bf16-operand instances (gad_attention_args.operand_precision = 1; the analogue of the reference's fp16 autocast for
// configs 4 / 5): the same three kernels on v_mfma_f32_16x16x32_bf16 - operands rounded to bf16 (RNE), products exact,
// fp32 accumulation, fp32 softmax statistics; q, k, v, o, gradients stay fp32 in HBM.
//   * streamed tiles (K, V forward / dQ; Q, dO for dK/dV) pass through registers once per workgroup: fp32 float4
//     loads issued before the tile's MFMAs, converted and written to the other LDS buffer after them (T14 split);
//   * LDS images are row-major bf16: rows of Dk32+8 elements serve the fragments whose lanes run along the tile's
//     rows (one ds_read_b128 = 8 k), rows of SV elements (SV = 16 mod 32: the 8 rows x 4 segments of a 32-lane half
//     cover the 64 banks once) serve the fragments whose lanes run along the columns through the transposing
//     ds_read_b64_tr_b16 - 4 rows x 16 columns per 16-lane group, i.e. exactly the (k slot, column) shape of an
//     MFMA operand;
//   * the accumulator-as-operand trick carries over: a 32-deep contraction step takes lane group g's slots
//     j < 4 from the first 16x16 accumulator (rows 4 g + j) and j >= 4 from the second (rows 16 + 4 g + j - 4): the
//     eight probabilities a lane holds are packed to bf16 in place and ARE the B operand; the A operand is read with
//     the same row assignment.
// ==================================================================================================================
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

template <int D>
struct CfgH {
  static_assert(D % 8 == 0, "head dim must be a multiple of 8");
  static constexpr int DK32 = (D + 31) / 32 * 32;          // contraction over the head dim: 32-deep MFMA steps
  static constexpr int KS = DK32 / 32;
  static constexpr int SK = DK32 + 8;                      // row stride (bf16) of the image read along rows
  static constexpr int DV16 = (D + 15) / 16 * 16;
  static constexpr int NDV = DV16 / 16;
  static constexpr int SV = (DV16 % 32 == 16) ? DV16 : DV16 + 16;   // row stride (bf16) of the image read transposed
  static constexpr int NF4 = KV * D / 4;                   // float4 per streamed tile
  static constexpr int SLOTS = (NF4 + NT - 1) / NT;        // per thread
};

__device__ __forceinline__ f32x4 mfma16h(bf16x8_t a, bf16x8_t b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

// storage type of q / k / v / o and their gradients: fp32 (operand_precision = 1 of gad_attention_*) or bf16 (gad_h_attention_*:
// the half-precision activation path, strides in elements)
template <bool HIO> struct IoT { typedef float type; };
template <> struct IoT<true> { typedef unsigned short type; };
__device__ __forceinline__ f32x4 load4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 load4(const unsigned short* p) {
  const unsigned long long u = *reinterpret_cast<const unsigned long long*>(p);
  const unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32);
  return f32x4{__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
}
__device__ __forceinline__ void store4(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void store4(unsigned short* p, f32x4 v) {
  const bf16x4_t h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
  *reinterpret_cast<bf16x4_t*>(p) = h;
}
// issue the loads of rows row0 .. row0+KV-1 (zeros past T)
template <int D, typename T>
__device__ __forceinline__ void tile_fetch(f32x4 (&r)[CfgH<D>::SLOTS], const T* base, int ld, int row0, int T_) {
  using C = CfgH<D>;
#pragma unroll
  for (int i = 0; i < C::SLOTS; ++i) {
    const int f = threadIdx.x + NT * i;
    const int row = f / (D / 4), c4 = (f - row * (D / 4)) * 4;
    const bool ok = f < C::NF4 && row0 + row < T_;
    r[i] = ok ? load4(base + (long)(row0 + row) * ld + c4) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
// convert and write a fetched tile into a row-major bf16 image with row stride S
template <int D, int S>
__device__ __forceinline__ void tile_commit(unsigned short* img, const f32x4 (&r)[CfgH<D>::SLOTS], float mul = 1.f) {
  using C = CfgH<D>;
#pragma unroll
  for (int i = 0; i < C::SLOTS; ++i) {
    const int f = threadIdx.x + NT * i;
    if (f < C::NF4) {
      const int row = f / (D / 4), c4 = (f - row * (D / 4)) * 4;
      const bf16x4_t h = {(__bf16)(r[i][0] * mul), (__bf16)(r[i][1] * mul), (__bf16)(r[i][2] * mul), (__bf16)(r[i][3] * mul)};
      *reinterpret_cast<bf16x4_t*>(img + row * S + c4) = h;
    }
  }
}
// one streamed tile's registers: fp32 storage -> float4 slots converted at the commit; bf16 storage -> 16-byte slots copied as they are
template <int D, bool HIO>
struct TileIO {
  f32x4 r[CfgH<D>::SLOTS];
  __device__ __forceinline__ void fetch(const float* base, int ld, int row0, int T_) { tile_fetch<D>(r, base, ld, row0, T_); }
  template <int S>
  __device__ __forceinline__ void commit(unsigned short* img) const { tile_commit<D, S>(img, r); }
};
template <int D>
struct TileIO<D, true> {
  typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
  static constexpr int N8 = KV * D / 8, SLOTS8 = (N8 + NT - 1) / NT;
  u32x4_t r[SLOTS8];
  __device__ __forceinline__ void fetch(const unsigned short* base, int ld, int row0, int T_) {
#pragma unroll
    for (int i = 0; i < SLOTS8; ++i) {
      const int f = threadIdx.x + NT * i;
      const int row = f / (D / 8), c8 = (f - row * (D / 8)) * 8;
      const bool ok = f < N8 && row0 + row < T_;
      r[i] = ok ? *reinterpret_cast<const u32x4_t*>(base + (long)(row0 + row) * ld + c8) : u32x4_t{0u, 0u, 0u, 0u};
    }
  }
  template <int S>
  __device__ __forceinline__ void commit(unsigned short* img) const {
    static_assert((S * 2) % 16 == 0, "image rows must keep 16-byte alignment");
#pragma unroll
    for (int i = 0; i < SLOTS8; ++i) {
      const int f = threadIdx.x + NT * i;
      if (f < N8) {
        const int row = f / (D / 8), c8 = (f - row * (D / 8)) * 8;
        *reinterpret_cast<u32x4_t*>(img + row * S + c8) = r[i];
      }
    }
  }
};
// fragment whose lanes run along the image's rows: lane (row r0 + (l & 15), k group g = l >> 4) -> k = 32 s + 8 g + j
template <int S>
__device__ __forceinline__ bf16x8_t row_frag_h(const unsigned short* img, int r0, int s) {
  const int lane = threadIdx.x & 63;
  return *reinterpret_cast<const bf16x8_t*>(img + (r0 + (lane & 15)) * S + 32 * s + 8 * (lane >> 4));
}
// fragment whose lanes run along the image's columns, contraction over 32 rows rb .. rb+31 with the slot assignment
// (g, j < 4) -> row rb + 4 g + j, (g, j >= 4) -> row rb + 16 + 4 g + j - 4; lane l holds column c0 + (l & 15)
template <int S>
__device__ __forceinline__ bf16x8_t col_frag_h(const unsigned short* img, int rb, int c0) {
  const int lane = threadIdx.x & 63, g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
  const unsigned short* a0 = img + (rb + 4 * g + q) * S + c0 + 4 * pp;
  const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)a0);
  const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(a0 + 16 * S));
  const s16x8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8_t, both);
}
// loop-invariant row fragments straight from global memory (-> bf16, zero past the head dim / the sequence)
template <int D, typename T>
__device__ __forceinline__ void row_frag_global_h(const T* base, int ld, int row, bool ok, float mul, bf16x8_t (&f)[CfgH<D>::KS]) {
  using C = CfgH<D>;
  const int g = (threadIdx.x & 63) >> 4;
  const T* p = base + (long)(ok ? row : 0) * ld;
#pragma unroll
  for (int s = 0; s < C::KS; ++s) {
    const int k = 32 * s + 8 * g;
    const bool in = ok && k < D;                      // D % 8 == 0: a group of 8 is all inside or all outside
    const f32x4 a = in ? load4(p + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4 b = in ? load4(p + k + 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    f[s] = bf16x8_t{(__bf16)(a[0] * mul), (__bf16)(a[1] * mul), (__bf16)(a[2] * mul), (__bf16)(a[3] * mul),
                    (__bf16)(b[0] * mul), (__bf16)(b[1] * mul), (__bf16)(b[2] * mul), (__bf16)(b[3] * mul)};
  }
}
__device__ __forceinline__ bf16x8_t pack8(const f32x4& a, const f32x4& b) {
  return bf16x8_t{(__bf16)a[0], (__bf16)a[1], (__bf16)a[2], (__bf16)a[3], (__bf16)b[0], (__bf16)b[1], (__bf16)b[2], (__bf16)b[3]};
}
__device__ __forceinline__ void zero_lds(unsigned short* lds, int n_elems) {
  for (int i = threadIdx.x * 8; i < n_elems; i += NT * 8) *reinterpret_cast<f32x4*>(lds + i) = f32x4{0.f, 0.f, 0.f, 0.f};
}

template <int D, int NQ, bool HIO = false>
__global__ __launch_bounds__(NT) void attn_fwd_bf16_kernel(const AttnDev p) {
  using C = CfgH<D>;
  using T = typename IoT<HIO>::type;
  constexpr int KIMG = KV * C::SK, VIMG = KV * C::SV, BUF = (KIMG + VIMG + 7) / 8 * 8;
  extern __shared__ __attribute__((aligned(16))) unsigned short ldsh[];          // [2][K image | V image]
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.sq + h * D;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.sk + h * D;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.sv + h * D;
  const int qw = blk * (64 * NQ) + wave * (16 * NQ);

  bf16x8_t qf[NQ][C::KS];
#pragma unroll
  for (int t = 0; t < NQ; ++t) row_frag_global_h<D>(Q, p.ldq, qw + 16 * t + c, qw + 16 * t + c < p.Tq, p.scale * LOG2E, qf[t]);
  f32x4 o[C::NDV][NQ];
#pragma unroll
  for (int i = 0; i < C::NDV; ++i)
#pragma unroll
    for (int t = 0; t < NQ; ++t) o[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m[NQ], l[NQ];
#pragma unroll
  for (int t = 0; t < NQ; ++t) { m[t] = NEG_BIG; l[t] = 0.f; }

  zero_lds(ldsh, 2 * BUF);          // pad columns (k >= D of the K image, dv >= D of the V image) stay zero for good
  __syncthreads();
  TileIO<D, HIO> rk, rv;
  rk.fetch(K, p.ldk, 0, p.Tk);
  rv.fetch(V, p.ldv, 0, p.Tk);
  rk.template commit<C::SK>(ldsh);
  rv.template commit<C::SV>(ldsh + KIMG);
  __syncthreads();

  const int ntiles = (p.Tk + KV - 1) / KV;
  for (int it = 0; it < ntiles; ++it) {
    const unsigned short* kimg = ldsh + (it & 1) * BUF;
    const unsigned short* vimg = kimg + KIMG;
    const bool more = it + 1 < ntiles;
    if (more) {
      rk.fetch(K, p.ldk, (it + 1) * KV, p.Tk);
      rv.fetch(V, p.ldv, (it + 1) * KV, p.Tk);
    }
    f32x4 s[2][NQ];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
#pragma unroll
      for (int t = 0; t < NQ; ++t) s[kt][t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < C::KS; ++st) {
        const bf16x8_t kf = row_frag_h<C::SK>(kimg, 16 * kt, st);
#pragma unroll
        for (int t = 0; t < NQ; ++t) s[kt][t] = mfma16h(kf, qf[t][st], s[kt][t]);
      }
    }
    if ((it + 1) * KV > p.Tk) {
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (it * KV + 16 * kt + 4 * g + e >= p.Tk)
#pragma unroll
            for (int t = 0; t < NQ; ++t) s[kt][t][e] = -__builtin_inff();
    }
    bf16x8_t pb[NQ];
#pragma unroll
    for (int t = 0; t < NQ; ++t) {
      float mx = fmaxf(fmaxf(fmaxf(s[0][t][0], s[0][t][1]), fmaxf(s[0][t][2], s[0][t][3])),
                       fmaxf(fmaxf(s[1][t][0], s[1][t][1]), fmaxf(s[1][t][2], s[1][t][3])));
      mx = xmax16_32(mx);
      const float mn = fmaxf(m[t], mx);
      const float alpha = ex2(m[t] - mn);
      m[t] = mn;
      float ps = 0.f;
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = ex2(s[kt][t][e] - mn);
          s[kt][t][e] = pe;
          ps += pe;
        }
      l[t] = l[t] * alpha + ps;
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) o[i][t] *= alpha;
      pb[t] = pack8(s[0][t], s[1][t]);          // slots j < 4: keys 4 g + j, j >= 4: keys 16 + 4 g + j - 4
    }
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) {
      const bf16x8_t vf = col_frag_h<C::SV>(vimg, 0, 16 * i);
#pragma unroll
      for (int t = 0; t < NQ; ++t) o[i][t] = mfma16h(vf, pb[t], o[i][t]);
    }
    if (more) {
      unsigned short* nb = ldsh + ((it + 1) & 1) * BUF;
      rk.template commit<C::SK>(nb);
      rv.template commit<C::SV>(nb + KIMG);
    }
    __syncthreads();
  }

  T* O = reinterpret_cast<T*>(p.o) + b * p.so + h * D;
#pragma unroll
  for (int t = 0; t < NQ; ++t) {
    const float lt = xsum16_32(l[t]);
    const float inv = 1.f / lt;
    const int row = qw + 16 * t + c;
    if (row < p.Tq) {
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) {
        const int dv = 16 * i + 4 * g;
        if (dv < D) store4(O + (long)row * p.ldo + dv, o[i][t] * inv);
      }
      if (g == 0 && p.lse) p.lse[(long)bh * p.Tq + row] = m[t] + __builtin_amdgcn_logf(lt);
    }
  }
}

// dQ, bf16 operands: workgroup = 64 queries, wave = 16; K and V tiles in ONE image type (row stride SV) read both along
// rows (S^T = K Q^T, dP^T = V dO^T) and transposed (dQ^T += K^T dS^T).  Row reads past the head dim meet zero B
// operands (the loop-invariant fragments are zero there), so whatever finite bf16 they pick up contributes nothing.
// NU: 16-row groups per wave (1 or 2).  With two, a wave's streamed-tile fragments (the LDS reads) serve twice the MFMA work and a
// workgroup covers 128 rows: per-tile costs - barrier, tile copy, fragment reads - are paid half as often per FLOP.
template <int D, bool HIO = false, int NU = 1>
__global__ __launch_bounds__(NT) void attn_bwd_dq_bf16_kernel(const AttnDev p) {
  using C = CfgH<D>;
  using T = typename IoT<HIO>::type;
  constexpr int IMG = KV * C::SV, BUF = 2 * IMG;
  extern __shared__ __attribute__((aligned(16))) unsigned short ldsh[];          // [2][K image | V image] + tail pad
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const T* K = reinterpret_cast<const T*>(p.k) + b * p.sk + h * D;
  const T* V = reinterpret_cast<const T*>(p.v) + b * p.sv + h * D;
  int row[NU];
  bool rok[NU];
  bf16x8_t qf[NU][C::KS], dof[NU][C::KS];
  float L2[NU], dl[NU];
  f32x4 dq[NU][C::NDV];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    row[u] = blk * (64 * NU) + wave * (16 * NU) + 16 * u + c;
    rok[u] = row[u] < p.Tq;
    row_frag_global_h<D>(reinterpret_cast<const T*>(p.q) + b * p.sq + h * D, p.ldq, row[u], rok[u], p.scale * LOG2E, qf[u]);
    row_frag_global_h<D>(reinterpret_cast<const T*>(p.d_o) + b * p.sdo + h * D, p.lddo, row[u], rok[u], 1.f, dof[u]);
    L2[u] = rok[u] ? p.lse[(long)bh * p.Tq + row[u]] : 0.f;
    dl[u] = rok[u] ? p.delta[(long)bh * p.Tq + row[u]] : 0.f;
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) dq[u][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }

  zero_lds(ldsh, 2 * BUF + 64);
  __syncthreads();
  TileIO<D, HIO> rk, rv;
  rk.fetch(K, p.ldk, 0, p.Tk);
  rv.fetch(V, p.ldv, 0, p.Tk);
  rk.template commit<C::SV>(ldsh);
  rv.template commit<C::SV>(ldsh + IMG);
  __syncthreads();
  const int ntiles = (p.Tk + KV - 1) / KV;
  for (int it = 0; it < ntiles; ++it) {
    const unsigned short* kimg = ldsh + (it & 1) * BUF;
    const unsigned short* vimg = kimg + IMG;
    const bool more = it + 1 < ntiles;
    if (more) {
      rk.fetch(K, p.ldk, (it + 1) * KV, p.Tk);
      rv.fetch(V, p.ldv, (it + 1) * KV, p.Tk);
    }
    f32x4 ds[NU][2];
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      bf16x8_t ka[C::KS], va[C::KS];
#pragma unroll
      for (int st = 0; st < C::KS; ++st) { ka[st] = row_frag_h<C::SV>(kimg, 16 * kt, st); va[st] = row_frag_h<C::SV>(vimg, 16 * kt, st); }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < C::KS; ++st) {
          s = mfma16h(ka[st], qf[u][st], s);
          dp = mfma16h(va[st], dof[u][st], dp);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool kok = it * KV + 16 * kt + 4 * g + e < p.Tk;
          const float pe = kok ? ex2(s[e] - L2[u]) : 0.f;
          ds[u][kt][e] = pe * (dp[e] - dl[u]);
        }
      }
    }
    bf16x8_t dsb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) dsb[u] = pack8(ds[u][0], ds[u][1]);
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) {
      const bf16x8_t kc = col_frag_h<C::SV>(kimg, 0, 16 * i);
#pragma unroll
      for (int u = 0; u < NU; ++u) dq[u][i] = mfma16h(kc, dsb[u], dq[u][i]);
    }
    if (more) {
      unsigned short* nb = ldsh + ((it + 1) & 1) * BUF;
      rk.template commit<C::SV>(nb);
      rv.template commit<C::SV>(nb + IMG);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (rok[u]) {
      T* DQ = reinterpret_cast<T*>(p.dq) + b * p.sdq + h * D + (long)row[u] * p.lddq;
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) {
        const int kk = 16 * i + 4 * g;
        if (kk < D) store4(DQ + kk, dq[u][i] * p.scale);
      }
    }
}

// dK / dV, bf16 operands: workgroup = 64 NU keys, wave = 16 NU; Q and dO tiles in one image type, read along rows for
// S = Q K^T and dP = dO V^T and transposed for dV^T += dO^T P and dK^T += Q^T dS.
template <int D, bool HIO = false, int NU = 1>
__global__ __launch_bounds__(NT) void attn_bwd_dkv_bf16_kernel(const AttnDev p) {
  using C = CfgH<D>;
  using T = typename IoT<HIO>::type;
  constexpr int IMG = KV * C::SV, BUF = 2 * IMG;
  extern __shared__ __attribute__((aligned(16))) unsigned short ldsh[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, c = lane & 15, g = lane >> 4;
  const int bh = blockIdx.x / p.nblk, blk = blockIdx.x - bh * p.nblk, b = bh / p.heads, h = bh - b * p.heads;
  const T* Q = reinterpret_cast<const T*>(p.q) + b * p.sq + h * D;
  const T* DO = reinterpret_cast<const T*>(p.d_o) + b * p.sdo + h * D;
  int key[NU];
  bool kok[NU];
  bf16x8_t kf[NU][C::KS], vf[NU][C::KS];
  f32x4 dk[NU][C::NDV], dv[NU][C::NDV];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    key[u] = blk * (64 * NU) + wave * (16 * NU) + 16 * u + c;
    kok[u] = key[u] < p.Tk;
    row_frag_global_h<D>(reinterpret_cast<const T*>(p.k) + b * p.sk + h * D, p.ldk, key[u], kok[u], p.scale * LOG2E, kf[u]);
    row_frag_global_h<D>(reinterpret_cast<const T*>(p.v) + b * p.sv + h * D, p.ldv, key[u], kok[u], 1.f, vf[u]);
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) { dk[u][i] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[u][i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  }
  const float* lse = p.lse + (long)bh * p.Tq;
  const float* dlt = p.delta + (long)bh * p.Tq;

  zero_lds(ldsh, 2 * BUF + 64);
  __syncthreads();
  // the tile's log-sum-exp / delta rows travel with the tile: threads 0-63 fetch one value each (lse | delta) beside the tile's
  // loads and commit it to LDS with the images - read from global memory at their point of use they were eight dependent L2 loads per
  // lane in front of every tile's exponentials (and a register-side prefetch cost occupancy: measured slower)
  float* stt = reinterpret_cast<float*>(ldsh + 2 * BUF + 64);          // [2 stages][lse | delta][KV]
  const int st_which = (threadIdx.x >> 5) & 1, st_r = threadIdx.x & 31;
  float sv = 0.f;
  auto fetch_stats = [&](int tile) {
    if (threadIdx.x < 64) {
      const int qrow = tile * KV + st_r;
      sv = qrow < p.Tq ? (st_which ? dlt[qrow] : lse[qrow]) : 0.f;
    }
  };
  auto commit_stats = [&](int stage) {
    if (threadIdx.x < 64) stt[stage * 2 * KV + st_which * KV + st_r] = sv;
  };
  TileIO<D, HIO> rq, rg;
  rq.fetch(Q, p.ldq, 0, p.Tq);
  rg.fetch(DO, p.lddo, 0, p.Tq);
  fetch_stats(0);
  rq.template commit<C::SV>(ldsh);
  rg.template commit<C::SV>(ldsh + IMG);
  commit_stats(0);
  __syncthreads();
  const int ntiles = (p.Tq + KV - 1) / KV;
  for (int it = 0; it < ntiles; ++it) {
    const unsigned short* qimg = ldsh + (it & 1) * BUF;
    const unsigned short* gimg = qimg + IMG;
    const bool more = it + 1 < ntiles;
    if (more) {
      rq.fetch(Q, p.ldq, (it + 1) * KV, p.Tq);
      rg.fetch(DO, p.lddo, (it + 1) * KV, p.Tq);
      fetch_stats(it + 1);
    }
    const float* cst = stt + (it & 1) * 2 * KV;
    f32x4 pr[NU][2], ds[NU][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      bf16x8_t qa[C::KS], ga[C::KS];
#pragma unroll
      for (int st = 0; st < C::KS; ++st) { qa[st] = row_frag_h<C::SV>(qimg, 16 * t, st); ga[st] = row_frag_h<C::SV>(gimg, 16 * t, st); }
      float L2[4], dl[4];
      bool qok[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 16 * t + 4 * g + e;
        qok[e] = it * KV + r < p.Tq;
        L2[e] = cst[r];
        dl[e] = cst[KV + r];
      }
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < C::KS; ++st) {
          s = mfma16h(qa[st], kf[u][st], s);
          dp = mfma16h(ga[st], vf[u][st], dp);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float pe = qok[e] ? ex2(s[e] - L2[e]) : 0.f;
          pr[u][t][e] = pe;
          ds[u][t][e] = pe * (dp[e] - dl[e]);
        }
      }
    }
    bf16x8_t pb[NU], dsb[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) { pb[u] = pack8(pr[u][0], pr[u][1]); dsb[u] = pack8(ds[u][0], ds[u][1]); }
#pragma unroll
    for (int i = 0; i < C::NDV; ++i) {
      const bf16x8_t gc = col_frag_h<C::SV>(gimg, 0, 16 * i), qc = col_frag_h<C::SV>(qimg, 0, 16 * i);
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        dv[u][i] = mfma16h(gc, pb[u], dv[u][i]);
        dk[u][i] = mfma16h(qc, dsb[u], dk[u][i]);
      }
    }
    if (more) {
      unsigned short* nb = ldsh + ((it + 1) & 1) * BUF;
      rq.template commit<C::SV>(nb);
      rg.template commit<C::SV>(nb + IMG);
      commit_stats((it + 1) & 1);
    }
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < NU; ++u)
    if (kok[u]) {
      T* DK = reinterpret_cast<T*>(p.dk) + b * p.sdk + h * D + (long)key[u] * p.lddk;
      T* DV = reinterpret_cast<T*>(p.dv) + b * p.sdv + h * D + (long)key[u] * p.lddv;
#pragma unroll
      for (int i = 0; i < C::NDV; ++i) {
        const int kk = 16 * i + 4 * g;
        if (kk < D) {
          store4(DK + kk, dk[u][i] * p.scale);
          store4(DV + kk, dv[u][i]);
        }
      }
    }
}

// Kernels that want more than 64 KiB of dynamic LDS need the attribute once per kernel instance and device - not per
// launch (a driver call on the sampling / training hot path, and one made inside hipGraph captures of the U-Net).
template <typename F>
static int set_lds(F kernel, int bytes, const char* what, unsigned* done_mask) {
  if (bytes <= 64 * 1024) return 0;
  int dev = 0;
  (void)hipGetDevice(&dev);
  const unsigned bit = 1u << (dev & 31);
  if (__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit) return 0;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    gad_set_error("%s: cannot reserve %d bytes of LDS: %s", what, bytes, hipGetErrorString(e));
    return 1;
  }
  __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
  return 0;
}

// head dims with an instance; any other d <= 256 runs the next larger one in its RG form
#define GAD_ATTN_DIMS(X) X(16) X(24) X(32) X(40) X(48) X(64) X(80) X(96) X(128) X(160) X(192) X(224) X(256)
static int instance_dim(int d) {
#define X(DIM) if (d <= DIM) return DIM;
  GAD_ATTN_DIMS(X)
#undef X
  return 0;
}
static bool exact_instance(int d) { return d > 0 && instance_dim(d) == d; }

}  // namespace
extern "C" int gad_attention_supported(int32_t d) { return d >= 1 && d <= 256; }
namespace {

// Does this launch meet the float4 / LDS-DMA contract (an exact instance, every row and batch stride a multiple of 4
// floats, every pointer 16-byte aligned)?  If not it runs the RG instance (dword staging, any d <= 256, any alignment).
static bool fast_contract(const gad_attention_args* a, bool bwd) {
  bool ok = exact_instance(a->d) && a->ldq % 4 == 0 && a->ldk % 4 == 0 && a->ldv % 4 == 0 && a->ldo % 4 == 0 &&
            a->stride_q % 4 == 0 && a->stride_k % 4 == 0 && a->stride_v % 4 == 0 && a->stride_o % 4 == 0 &&
            gad_aligned16(a->q) && gad_aligned16(a->k) && gad_aligned16(a->v) && gad_aligned16(a->o);
  if (bwd)
    ok = ok && a->ld_do % 4 == 0 && a->ld_dq % 4 == 0 && a->ld_dk % 4 == 0 && a->ld_dv % 4 == 0 && a->stride_do % 4 == 0 &&
         a->stride_dq % 4 == 0 && a->stride_dk % 4 == 0 && a->stride_dv % 4 == 0 && gad_aligned16(a->d_o) &&
         gad_aligned16(a->dq) && gad_aligned16(a->dk) && gad_aligned16(a->dv);
  return ok;
}

static int check_common(const gad_attention_args* a, const char* who) {
  GAD_CHECK(a && a->q && a->k && a->v && a->o, "%s: null pointer", who);
  GAD_CHECK(a->B > 0 && a->heads > 0 && a->Tq > 0 && a->Tk > 0, "%s: bad shape B=%d heads=%d Tq=%d Tk=%d", who, a->B, a->heads, a->Tq, a->Tk);
  GAD_CHECK(gad_attention_supported(a->d), "%s: head dim %d is outside 1..256", who, a->d);
  const int w = a->heads * a->d;
  GAD_CHECK(a->ldq >= w && a->ldk >= w && a->ldv >= w && a->ldo >= w, "%s: a row stride is smaller than heads*d = %d", who, w);
  GAD_CHECK((long)a->B * a->heads * gad_ceil_div(a->Tq > a->Tk ? a->Tq : a->Tk, 64) < (1L << 31), "%s: B*heads*blocks exceeds the grid", who);
  GAD_CHECK(a->operand_precision == 0 || a->operand_precision == 1, "%s: operand_precision must be 0 or 1", who);
  return 0;
}

static AttnDev make_dev(const gad_attention_args* a) {
  AttnDev d;
  memset(&d, 0, sizeof(d));
  d.q = a->q; d.k = a->k; d.v = a->v; d.o = a->o; d.lse = a->lse;
  d.d_o = a->d_o; d.delta = a->delta; d.dq = a->dq; d.dk = a->dk; d.dv = a->dv;
  d.B = a->B; d.heads = a->heads; d.Tq = a->Tq; d.Tk = a->Tk; d.d = a->d;
  d.ldq = a->ldq; d.ldk = a->ldk; d.ldv = a->ldv; d.ldo = a->ldo;
  d.lddo = a->ld_do; d.lddq = a->ld_dq; d.lddk = a->ld_dk; d.lddv = a->ld_dv;
  d.sq = a->stride_q; d.sk = a->stride_k; d.sv = a->stride_v; d.so = a->stride_o;
  d.sdo = a->stride_do; d.sdq = a->stride_dq; d.sdk = a->stride_dk; d.sdv = a->stride_dv;
  d.scale = a->scale;
  d.ws = (float*)a->ws;
  return d;
}

static dim3 grid_of(AttnDev& d, int rows, int rows_per_block) {
  d.nblk = (int)gad_ceil_div(rows, rows_per_block);
  return dim3((unsigned)((long)d.nblk * d.B * d.heads));
}

template <int D, int NQ, bool RG>
static int launch_fwd(AttnDev d, hipStream_t st) {
  static unsigned lds_set = 0;
  const int bytes = 4 * Cfg<D>::TILE * (int)sizeof(float);
  if (set_lds(attn_fwd_f32_kernel<D, NQ, RG>, bytes, "gad_attention_fwd", &lds_set)) return 1;
  const dim3 grid = grid_of(d, d.Tq, 64 * NQ);
  hipLaunchKernelGGL((attn_fwd_f32_kernel<D, NQ, RG>), grid, dim3(NT), bytes, st, d);
  return 0;
}

template <int D>
static int launch_fwd_wide(AttnDev d, hipStream_t st) {
  static unsigned lds_set = 0;
  const int bytes = (4 * Cfg<D>::TILE + 2 * 4 * 64 * 8) * (int)sizeof(float);
  if (set_lds(attn_fwd_wide_f32_kernel<D>, bytes, "gad_attention_fwd", &lds_set)) return 1;
  const dim3 grid = grid_of(d, d.Tq, 64);
  hipLaunchKernelGGL((attn_fwd_wide_f32_kernel<D>), grid, dim3(NTW), bytes, st, d);
  return 0;
}

template <int D, int NQ>
static int launch_fwd_h(AttnDev d, hipStream_t st) {
  using C = CfgH<D>;
  static unsigned lds_set = 0;
  const int bytes = 2 * ((KV * C::SK + KV * C::SV + 7) / 8 * 8) * (int)sizeof(unsigned short);
  if (set_lds(attn_fwd_bf16_kernel<D, NQ>, bytes, "gad_attention_fwd", &lds_set)) return 1;
  const dim3 grid = grid_of(d, d.Tq, 64 * NQ);
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<D, NQ>), grid, dim3(NT), bytes, st, d);
  return 0;
}

template <int D, bool RG>
static int launch_bwd(AttnDev d, float* delta, hipStream_t st) {
  static unsigned lds_set_q = 0, lds_set_kv = 0;
  const int bytes = 4 * Cfg<D>::TILE * (int)sizeof(float);
  constexpr int NQB = (D <= 80 && !RG) ? 2 : 1;      // query blocks per wave of the dQ kernel (registers allow two up to d = 80)
  if (set_lds(attn_bwd_dq_f32_kernel<D, NQB, RG>, bytes, "gad_attention_bwd", &lds_set_q) ||
      set_lds(attn_bwd_dkv_f32_kernel<D, RG>, bytes, "gad_attention_bwd", &lds_set_kv)) return 1;
  const long total = (long)d.B * d.Tq * d.heads;
  hipLaunchKernelGGL((attn_delta_kernel<D, RG>), dim3((unsigned)gad_ceil_div(total, 256)), dim3(256), 0, st, d, delta);
  dim3 grid = grid_of(d, d.Tq, 64 * NQB);
  hipLaunchKernelGGL((attn_bwd_dq_f32_kernel<D, NQB, RG>), grid, dim3(NT), bytes, st, d);
  grid = grid_of(d, d.Tk, 64);
  hipLaunchKernelGGL((attn_bwd_dkv_f32_kernel<D, RG>), grid, dim3(NT), bytes, st, d);
  return 0;
}

// ---- single-pass backward: plan and launch ---------------------------------------------------------------------------
// keys per workgroup = 64 NK; NK <= 4 up to d = 40, <= 2 up to 80, 1 at 96 (dK / dV accumulators + the K / V fragments of
// NK key tiles per wave must fit the register file); among those the NK with the fewest padded keys, ties to the larger
// (fewer dQ slabs: a single block writes dQ directly)
static int bwd1_nk(int dinst, int Tk, long bh, int Tq) {
  if (dinst > 96) return 0;
  const int nkmax = dinst <= 40 ? 4 : dinst <= 80 ? 2 : 1;
  int best = 0;
  long best_pad = 0;
  for (int nk = 1; nk <= nkmax; nk *= 2) {
    const long nblk = gad_ceil_div(Tk, 64 * nk), pad = nblk * 64 * nk;
    if (bh * nblk < 512 && nk > 1) continue;               // keep two workgroups per CU in flight
    if (!best || pad <= best_pad) { best_pad = pad; best = nk; }
  }
  // few keys, many queries (cross attention over 77 text tokens at 64 x 64 latents): a key block per workgroup leaves
  // most of the chip idle, the dQ kernel of the pair is parallel over the queries - keep the pair there
  if (best == 1 && bh * gad_ceil_div(Tk, 64) < 512 && bh * gad_ceil_div(Tq, 128) > 4 * bh * gad_ceil_div(Tk, 64)) return 0;
  return best;
}
static int64_t bwd1_ws_bytes(const gad_attention_args* a) {
  const int nk = bwd1_nk(instance_dim(a->d), a->Tk, (long)a->B * a->heads, a->Tq);
  if (!nk) return 0;
  const long nblk = gad_ceil_div(a->Tk, 64 * nk);
  return nblk > 1 ? nblk * (int64_t)a->B * a->Tq * a->heads * a->d * (int64_t)sizeof(float) : 0;
}

template <int D, int NK, bool RG>
static int launch_bwd1(AttnDev d, float* delta, hipStream_t st) {
  using W = Bwd1<D, NK>;
  static unsigned lds_set = 0;
  const int bytes = W::LDS_FLOATS * (int)sizeof(float);
  if (set_lds(attn_bwd1_f32_kernel<D, NK, RG>, bytes, "gad_attention_bwd", &lds_set)) return 1;
  const long total = (long)d.B * d.Tq * d.heads;
  hipLaunchKernelGGL((attn_delta_kernel<D, RG>), dim3((unsigned)gad_ceil_div(total, 256)), dim3(256), 0, st, d, delta);
  const dim3 grid = grid_of(d, d.Tk, W::KB);
  GAD_CHECK(d.nblk == 1 || d.ws, "gad_attention_bwd: %d key blocks but no dQ slab workspace", d.nblk);   // never a null store
  hipLaunchKernelGGL((attn_bwd1_f32_kernel<D, NK, RG>), grid, dim3(NT), bytes, st, d);
  if (d.nblk > 1) {
    const int wd = d.heads * d.d;
    const int vec = wd % 4 == 0 && d.lddq % 4 == 0 && d.sdq % 4 == 0 && gad_aligned16(d.dq) && gad_aligned16(d.ws);
    const long n = (long)d.B * d.Tq * (vec ? wd / 4 : wd);
    const long blocks = gad_ceil_div(n, 256);
    hipLaunchKernelGGL(attn_dq_reduce_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, st, d, d.nblk, vec);
  }
  return 0;
}
template <int D, bool RG>
static int bwd1_dim(const AttnDev& d, float* delta, hipStream_t st, int nk) {
  if constexpr (D <= 96) {
    if (nk == 1) return launch_bwd1<D, 1, RG>(d, delta, st);
    if constexpr (D <= 80) { if (nk == 2) return launch_bwd1<D, 2, RG>(d, delta, st); }
    if constexpr (D <= 40) { if (nk == 4) return launch_bwd1<D, 4, RG>(d, delta, st); }
  }
  gad_set_error("gad_attention_bwd: no single-pass instance for d = %d, NK = %d", D, nk);
  return 1;
}

template <int D>
static int launch_bwd_h(AttnDev d, float* delta, hipStream_t st) {
  using C = CfgH<D>;
  static unsigned lds_set_q = 0, lds_set_kv = 0;
  const int bytes = (4 * KV * C::SV + 64) * (int)sizeof(unsigned short) + 4 * KV * (int)sizeof(float);     // (+ the dK/dV kernel's lse / delta rows)
  if (set_lds(attn_bwd_dq_bf16_kernel<D>, bytes, "gad_attention_bwd", &lds_set_q) ||
      set_lds(attn_bwd_dkv_bf16_kernel<D>, bytes, "gad_attention_bwd", &lds_set_kv)) return 1;
  const long total = (long)d.B * d.Tq * d.heads;
  hipLaunchKernelGGL((attn_delta_kernel<D>), dim3((unsigned)gad_ceil_div(total, 256)), dim3(256), 0, st, d, delta);
  dim3 grid = grid_of(d, d.Tq, 64);
  hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<D>), grid, dim3(NT), bytes, st, d);
  grid = grid_of(d, d.Tk, 64);
  hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<D>), grid, dim3(NT), bytes, st, d);
  return 0;
}

// ---- half-precision I/O launchers (gad_h_attention_*): the bf16-operand kernels with 16-bit loads / stores ----
template <int D>
__global__ void attn_delta_h_kernel(const AttnDev p, float* delta) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)p.B * p.Tq * p.heads;
  if (idx >= total) return;
  const int h = (int)(idx % p.heads);
  const long bq = idx / p.heads;
  const int q = (int)(bq % p.Tq), b = (int)(bq / p.Tq);
  const unsigned short* o = reinterpret_cast<const unsigned short*>(p.o) + b * p.so + (long)q * p.ldo + h * D;
  const unsigned short* g = reinterpret_cast<const unsigned short*>(p.d_o) + b * p.sdo + (long)q * p.lddo + h * D;
  float acc = 0.f;
#pragma unroll
  for (int i = 0; i < D; i += 4) {
    const f32x4 a = load4(o + i), d = load4(g + i);
    acc += a[0] * d[0] + a[1] * d[1] + a[2] * d[2] + a[3] * d[3];
  }
  delta[((long)b * p.heads + h) * p.Tq + q] = acc;
}
template <int D, int NQ>
static int launch_fwd_hio(AttnDev d, hipStream_t st) {
  using C = CfgH<D>;
  static unsigned lds_set = 0;
  const int bytes = 2 * ((KV * C::SK + KV * C::SV + 7) / 8 * 8) * (int)sizeof(unsigned short);
  if (set_lds(attn_fwd_bf16_kernel<D, NQ, true>, bytes, "gad_h_attention_fwd", &lds_set)) return 1;
  const dim3 grid = grid_of(d, d.Tq, 64 * NQ);
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<D, NQ, true>), grid, dim3(NT), bytes, st, d);
  return 0;
}
template <int D>
static int launch_bwd_hio(AttnDev d, float* delta, hipStream_t st) {
  using C = CfgH<D>;
  static unsigned lds_set_q = 0, lds_set_kv = 0;
  const int bytes = (4 * KV * C::SV + 64) * (int)sizeof(unsigned short) + 4 * KV * (int)sizeof(float);     // (+ the dK/dV kernel's lse / delta rows)
  if (set_lds(attn_bwd_dq_bf16_kernel<D, true>, bytes, "gad_h_attention_bwd", &lds_set_q) ||
      set_lds(attn_bwd_dkv_bf16_kernel<D, true>, bytes, "gad_h_attention_bwd", &lds_set_kv)) return 1;
  const long total = (long)d.B * d.Tq * d.heads;
  hipLaunchKernelGGL((attn_delta_h_kernel<D>), dim3((unsigned)gad_ceil_div(total, 256)), dim3(256), 0, st, d, delta);
  // 32 rows per wave (128 per workgroup) while the grid still fills the chip twice over (head dims up to 96)
  bool two_q = false, two_k = false;
  if constexpr (D <= 96) {
    two_q = gad_ceil_div(d.Tq, 128) * d.B * d.heads >= 512;
    two_k = gad_ceil_div(d.Tk, 128) * d.B * d.heads >= 512;
  }
  if constexpr (D <= 96) {
    if (two_q) {
      const dim3 grid = grid_of(d, d.Tq, 128);
      hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<D, true, 2>), grid, dim3(NT), bytes, st, d);
    }
  }
  if (!two_q) {
    const dim3 grid = grid_of(d, d.Tq, 64);
    hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<D, true>), grid, dim3(NT), bytes, st, d);
  }
  if constexpr (D <= 96) {
    if (two_k) {
      const dim3 grid = grid_of(d, d.Tk, 128);
      hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<D, true, 2>), grid, dim3(NT), bytes, st, d);
    }
  }
  if (!two_k) {
    const dim3 grid = grid_of(d, d.Tk, 64);
    hipLaunchKernelGGL((attn_bwd_dkv_bf16_kernel<D, true>), grid, dim3(NT), bytes, st, d);
  }
  return 0;
}
template <int D>
static int fwd_dim_hio(const AttnDev& d, hipStream_t st, bool wide) {
  // (64 queries per wave measured slower than 32 at d = 40, T = 4096: 731 vs 675 us)
  if constexpr (D <= 96) { if (wide) return launch_fwd_hio<D, 2>(d, st); }
  return launch_fwd_hio<D, 1>(d, st);
}
// bf16 I/O contract: head dim an exact instance (multiple of 8, <= 160), 16-byte aligned rows (strides multiples of 8 elements)
static int check_hio(const gad_attention_args* a, bool bwd, const char* who) {
  if (check_common(a, who)) return 1;
  GAD_CHECK(exact_instance(a->d) && a->d <= 160, "%s: head dim %d has no bf16-I/O instance", who, a->d);
  auto al = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
  GAD_CHECK(al(a->q) && al(a->k) && al(a->v) && al(a->o), "%s: q / k / v / o must be 16-byte aligned", who);
  GAD_CHECK(a->ldq % 8 == 0 && a->ldk % 8 == 0 && a->ldv % 8 == 0 && a->ldo % 8 == 0 && a->stride_q % 8 == 0 && a->stride_k % 8 == 0 &&
            a->stride_v % 8 == 0 && a->stride_o % 8 == 0, "%s: strides must be multiples of 8 elements", who);
  if (bwd) {
    GAD_CHECK(a->lse && a->d_o && a->delta && a->dq && a->dk && a->dv, "%s: null pointer (lse / d_o / delta / dq / dk / dv)", who);
    GAD_CHECK(al(a->d_o) && al(a->dq) && al(a->dk) && al(a->dv), "%s: gradients must be 16-byte aligned", who);
    GAD_CHECK(a->ld_do % 8 == 0 && a->ld_dq % 8 == 0 && a->ld_dk % 8 == 0 && a->ld_dv % 8 == 0 && a->stride_do % 8 == 0 &&
              a->stride_dq % 8 == 0 && a->stride_dk % 8 == 0 && a->stride_dv % 8 == 0, "%s: gradient strides must be multiples of 8 elements", who);
    const int w = a->heads * a->d;
    GAD_CHECK(a->ld_do >= w && a->ld_dq >= w && a->ld_dk >= w && a->ld_dv >= w, "%s: a gradient row stride is smaller than heads*d = %d", who, w);
  }
  return 0;
}

// one instance dim: pick the form (RG / bf16 / f32, queries per workgroup)
template <int D>
static int fwd_dim(const AttnDev& d, hipStream_t st, bool rg, bool bf16, bool wide, bool two_kernel_legacy) {
  if (rg) return launch_fwd<D, 1, true>(d, st);
  if (bf16) {
    if constexpr (D <= 96) { if (wide) return launch_fwd_h<D, 2>(d, st); }
    return launch_fwd_h<D, 1>(d, st);
  }
  if constexpr (D <= 80) { if (wide) return launch_fwd<D, 2, false>(d, st); }
  if constexpr (D >= 160 && D % 32 == 0) { if (!two_kernel_legacy) return launch_fwd_wide<D>(d, st); }
  return launch_fwd<D, 1, false>(d, st);
}
template <int D>
static int bwd_dim(const AttnDev& d, float* delta, hipStream_t st, bool rg, bool bf16, int nk1) {
  if (nk1) return rg ? bwd1_dim<D, true>(d, delta, st, nk1) : bwd1_dim<D, false>(d, delta, st, nk1);
  if (rg) return launch_bwd<D, true>(d, delta, st);
  if (bf16) return launch_bwd_h<D>(d, delta, st);
  return launch_bwd<D, false>(d, delta, st);
}

}  // namespace

extern "C" int64_t gad_attention_bwd_workspace_bytes(const gad_attention_args* a) {
  if (!a || !gad_attention_supported(a->d) || a->B <= 0 || a->heads <= 0 || a->Tq <= 0 || a->Tk <= 0) return 0;
  if (a->operand_precision == 1 && fast_contract(a, true)) return 0;        // bf16 launches run the two-kernel pair
  return bwd1_ws_bytes(a);
}

// 1 if this launch runs bf16-operand kernels (operand_precision = 1 AND the float4 contract holds; RG launches are fp32)
extern "C" int gad_attention_uses_bf16(const gad_attention_args* a, int32_t backward) {
  return a && a->operand_precision == 1 && fast_contract(a, backward != 0);
}

extern "C" int gad_attention_fwd(const gad_attention_args* a, void* stream) {
  if (check_common(a, "gad_attention_fwd")) return 1;
  const AttnDev d = make_dev(a);
  hipStream_t st = (hipStream_t)stream;
  const bool rg = !fast_contract(a, false), bf16 = a->operand_precision == 1;
  // queries per workgroup: 128 while the (b, h, block) grid still fills the chip twice over, else 64
  const bool wide = gad_ceil_div(a->Tq, 128) * a->B * a->heads >= 512;
  int rc = 1;
  switch (instance_dim(a->d)) {
#define X(DIM) case DIM: rc = fwd_dim<DIM>(d, st, rg, bf16, wide, (a->flags & GAD_ATTN_NARROW_FWD) != 0); break;
    GAD_ATTN_DIMS(X)
#undef X
  }
  if (rc) return rc;
  GAD_LAUNCH_CHECK("gad_attention_fwd");
  return 0;
}

extern "C" int gad_attention_bwd(const gad_attention_args* a, void* stream) {
  if (check_common(a, "gad_attention_bwd")) return 1;
  GAD_CHECK(a->lse && a->d_o && a->delta && a->dq && a->dk && a->dv, "gad_attention_bwd: null pointer (lse / d_o / delta / dq / dk / dv)");
  const int w = a->heads * a->d;
  GAD_CHECK(a->ld_do >= w && a->ld_dq >= w && a->ld_dk >= w && a->ld_dv >= w, "gad_attention_bwd: a gradient row stride is smaller than heads*d = %d", w);
  const AttnDev d = make_dev(a);
  hipStream_t st = (hipStream_t)stream;
  const bool rg = !fast_contract(a, true), bf16 = a->operand_precision == 1;
  // single-pass kernel: exact-fp32 launches up to d = 96 whose dQ slabs fit the caller's workspace (flags bit 0 keeps the
  // dQ + dK/dV pair: A/B tools, tests); everything else - wider heads, bf16 operands - runs the pair
  int nk1 = (bf16 && !rg) || (a->flags & GAD_ATTN_TWO_KERNEL_BWD) ? 0 : bwd1_nk(instance_dim(a->d), a->Tk, (long)a->B * a->heads, a->Tq);
  if (nk1) {
    const int64_t need = bwd1_ws_bytes(a);
    if (need > 0 && !(a->ws && a->ws_bytes >= need && gad_aligned16(a->ws))) nk1 = 0;
  }
  int rc = 1;
  switch (instance_dim(a->d)) {
#define X(DIM) case DIM: rc = bwd_dim<DIM>(d, a->delta, st, rg, bf16, nk1); break;
    GAD_ATTN_DIMS(X)
#undef X
  }
  if (rc) return rc;
  GAD_LAUNCH_CHECK("gad_attention_bwd");
  return 0;
}

#define GAD_ATTN_HIO_DIMS(X) X(16) X(24) X(32) X(40) X(48) X(64) X(80) X(96) X(128) X(160)
extern "C" int gad_h_attention_fwd(const gad_attention_args* a, void* stream) {
  if (check_hio(a, false, "gad_h_attention_fwd")) return 1;
  const AttnDev d = make_dev(a);
  hipStream_t st = (hipStream_t)stream;
  const bool wide = gad_ceil_div(a->Tq, 128) * a->B * a->heads >= 512;
  int rc = 1;
  switch (a->d) {
#define X(DIM) case DIM: rc = fwd_dim_hio<DIM>(d, st, wide); break;
    GAD_ATTN_HIO_DIMS(X)
#undef X
  }
  if (rc) return rc;
  GAD_LAUNCH_CHECK("gad_h_attention_fwd");
  return 0;
}
extern "C" int gad_h_attention_bwd(const gad_attention_args* a, void* stream) {
  if (check_hio(a, true, "gad_h_attention_bwd")) return 1;
  const AttnDev d = make_dev(a);
  hipStream_t st = (hipStream_t)stream;
  int rc = 1;
  switch (a->d) {
#define X(DIM) case DIM: rc = launch_bwd_hio<DIM>(d, a->delta, st); break;
    GAD_ATTN_HIO_DIMS(X)
#undef X
  }
  if (rc) return rc;
  GAD_LAUNCH_CHECK("gad_h_attention_bwd");
  return 0;
}
