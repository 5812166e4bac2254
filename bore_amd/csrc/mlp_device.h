// mlp_device.h -- device building blocks shared by every kernel of the path.
// gfx950 only: 64-wide wavefronts, v_mfma_f32_16x16x4_f32 tiles, operands in LDS.
//
// Unit of work: a ROW-BLOCK = 16 rows of the tile.  One wave carries its row-block through
// the whole network: layer l+1 of those rows needs only layer l of the same rows, which the
// same wave produced, so forward and backward need no workgroup barrier -- only the weight
// gradient (a sum over ALL rows) does.
//
// A_l is the OUTPUT of layer l (A_0 = the input rows), D_l = d objective / d pre-activation
// of layer l (D_0 = d objective / d input).  Activation derivatives are written in terms of
// the activation output, so pre-activations are never stored.
//
// MFMA operand maps (v_mfma_f32_16x16x4_f32, lane l, m = l & 15, q = l >> 4):
//   A[16x4]: lane holds A[m][q]      B[4x16]: lane holds B[q][m]
//   C/D[16x16]: lane holds C[4q + r][m], r = 0..3
// The product is bit-for-bit a k-ordered fp32 fmaf chain (no wider accumulation).
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_layout.h"
#include "mlp_shapes.h"

// A phase body that the fused iteration kernel runs inside its resident loop derives its per-lane
// addresses from an OPAQUE copy of the work-item id: computed from threadIdx.x itself they are
// loop-invariant, get hoisted out of the BO-iteration loop and stay live across every other phase.
#define BORE_OPAQUE_TID(t) asm volatile("" : "+v"(t))

namespace bore {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float sigmoid_stable(float x) {
  float e = expf(-fabsf(x));  // expf: ocml, <=1 ulp
  float d = 1.f + e;
  return x >= 0.f ? 1.f / d : e / d;
}

// ---- the FIT's arithmetic: "<= 1 ulp per operation" (DESIGN.md 2), hardware rcp / sqrt / 2^t ----
__device__ __forceinline__ float fit_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fit_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
// exp(x) for x <= 0: 2^(x log2 e) with the product's rounding error fed back (the argument's error
// would otherwise be |x| 2^-24 in the exponent); flushes to zero below 2^-126 like the result's use
// (1 + e) does not notice
__device__ __forceinline__ float fit_exp_neg(float x) {
  const float L2E = 1.442695040888963f, L2E_LO = 1.925963033500e-8f;  // log2(e) = hi + lo
  const float t = x * L2E;
  const float r = fmaf(x, L2E_LO, fmaf(x, L2E, -t));                 // exact remainder of the product
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * 0.6931471805599453f, e);                          // 2^(t + r) = 2^t (1 + r ln 2)
}
// elu(x) in the fit (round 6).  The reference's layers are Keras `activation="elu"` (plugins/hpbandster/base.py:
// 152-155), whose TF kernel forms exp(x) - 1 for x < 0 (Eigen: features.exp() - 1).  ocml's expm1f is ~60
// instructions with branches, 24 calls per lane in a forward pass of 16->32-32-32-1: 6.7 k of an 18 k-cycle Adam
// step (profiles/r6/fit_marks_plugin.txt).  Here, for x < 0: the hardware's 2^(x log2 e) minus one -- the error of
// the rounded exponent is e^x |x| 2^-24 <= 0.37 x 2^-24 in absolute terms, under the half-ulp the exponential itself
// may be off near 1, so no feedback term; the subtraction is exact from x >= -0.69 on -- and the series
// x + x^2/2 + x^3/6 + x^4/24 where that difference would cancel (|x| < 1/32: truncation x^5 / 120 < 2^-26 |x|).
// Absolute error <= 2^-24 everywhere against the float64 oracle's expm1, relative <= 2e-6 (at the seam).
__device__ __forceinline__ float fit_elu(float x) {
  const float xn = fminf(x, 0.f);
  const float big = __builtin_amdgcn_exp2f(xn * 1.442695040888963f) - 1.f;
  const float ser = xn * fmaf(xn, fmaf(xn, fmaf(xn, 1.f / 24.f, 1.f / 6.f), 0.5f), 1.f);
  const float neg = xn > -0.03125f ? ser : big;
  return x > 0.f ? x : neg;
}

// FIT: the caller is a fit kernel (fit_elu instead of ocml's expm1f; everything else alike)
template <bool FIT = false>
__device__ __forceinline__ float act_fwd(int a, float x) {
  switch (a) {
    case BORE_ACT_RELU: return fmaxf(x, 0.f);
    case BORE_ACT_ELU:
      if constexpr (FIT) return fit_elu(x);
      else return x > 0.f ? x : expm1f(x);
    case BORE_ACT_SIGMOID: return sigmoid_stable(x);
    case BORE_ACT_TANH: return tanhf(x);
    default: return x;
  }
}

// d act / d pre-activation, from the activation OUTPUT h.
__device__ __forceinline__ float act_grad(int a, float h) {
  switch (a) {
    case BORE_ACT_RELU: return h > 0.f ? 1.f : 0.f;
    case BORE_ACT_ELU: return h > 0.f ? 1.f : h + 1.f;
    case BORE_ACT_SIGMOID: return h * (1.f - h);
    case BORE_ACT_TANH: return 1.f - h * h;
    default: return 1.f;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// Sum over the four 16-lane rows of a wave, the same bits in every lane: (s0 + s1) + (s2 + s3).
// permlane16_swap(x, y): x.row1 <-> y.row0, x.row3 <-> y.row2;  permlane32_swap(x, y):
// x.rows{2,3} <-> y.rows{0,1}  (gfx950 VALU, no LDS).
__device__ __forceinline__ float rows_sum4(float s) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(s), __float_as_uint(s), false, false);
  const float x = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// 4 x 4 transpose inside every lane quad: g[j] <- (lane of the quad with lane & 3 == j)'s g[lane & 3].
// Two butterfly stages (quad_perm [1,0,3,2], then [2,3,0,1]), written as selects on purpose:
// if / else assignments to g[] compile to execution-mask branches and per-register compares
// (60 instructions instead of 16).  odd = lane & 1, hi = lane & 2.
__device__ __forceinline__ float quad_xchg(float x, bool far) {
  const unsigned u = __float_as_uint(x);
  return __uint_as_float(far ? __builtin_amdgcn_update_dpp(0u, u, 0x4E, 0xF, 0xF, true)
                             : __builtin_amdgcn_update_dpp(0u, u, 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ void quad_transpose4(float (&g)[4], bool odd, bool hi) {
  const float x = quad_xchg(odd ? g[0] : g[1], false), y = quad_xchg(odd ? g[2] : g[3], false);
  const float a0 = odd ? x : g[0], a1 = odd ? g[1] : x, a2 = odd ? y : g[2], a3 = odd ? g[3] : y;
  const float s = quad_xchg(hi ? a0 : a2, true), t = quad_xchg(hi ? a1 : a3, true);
  g[0] = hi ? s : a0;
  g[1] = hi ? t : a1;
  g[2] = hi ? a2 : s;
  g[3] = hi ? a3 : t;
}

// A float array addressed as a raw buffer: element offset u (wave-uniform: scalar register) + byte
// offset lb of the lane.  buffer_load / buffer_store take both, so an access is one instruction;
// the same access through a flat pointer costs a 64-bit vector add first.  Accesses past the
// array read zero / are dropped (the descriptor carries the size), which nothing relies on.
struct BufF32 {
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  __amdgpu_buffer_rsrc_t r;
  __device__ __forceinline__ BufF32(float *base, int n_floats)
      : r(__builtin_amdgcn_make_buffer_rsrc(base, 0, n_floats * 4, 0x00020000)) {}
  __device__ __forceinline__ f4u ld4(int u, unsigned lb) const {
    return __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lb, u * 4, 0));
  }
  __device__ __forceinline__ float ld1(int u, unsigned lb) const {
    return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, (int)lb, u * 4, 0));
  }
  __device__ __forceinline__ void st4(f4u v, int u, unsigned lb) const {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, (int)lb, u * 4, 0);
  }
  __device__ __forceinline__ void st1(float v, int u, unsigned lb) const {
    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, (int)lb, u * 4, 0);
  }
};

// Lanes of ONE wave exchange data through LDS between two program points: all of the
// wave's earlier LDS accesses complete and the compiler may not move memory operations
// across (a wave runs in lock-step, so no s_barrier is involved).
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// packed (Keras order) parameter index -> which tensor element it is.
struct ParamRef {
  int l;     // layer, 1-based
  int k;     // input index (row of W_l); -1 for a bias
  int j;     // output index
  int lds;   // index in the padded LDS image
};

// (the layer loop is unrolled so that every table lookup has a constant index)
__device__ __forceinline__ ParamRef param_ref(const MlpLayout &L, int p, int n) {
  ParamRef r;
  r.l = 1;
  r.k = -1;
  r.j = 0;
  r.lds = 0;
#pragma unroll
  for (int l = 1; l <= n; ++l) {
    if (p >= L.goff_w[l] && p < L.goff_b[l]) {
      const int q = p - L.goff_w[l];
      r.l = l;
      r.k = q / L.w[l];
      r.j = q - r.k * L.w[l];
      r.lds = L.woff[l] + r.k * L.ldw[l] + r.j;
    } else if (p >= L.goff_b[l] && p < L.goff_b[l] + L.w[l]) {
      r.l = l;
      r.k = -1;
      r.j = p - L.goff_b[l];
      r.lds = L.boff[l] + r.j;
    }
  }
  return r;
}

__device__ __forceinline__ void zero_lds(float *p, int n_floats) {
  for (int i = threadIdx.x; i < n_floats; i += blockDim.x) p[i] = 0.f;
}

// Zero the workgroup's LDS and park a copy of the layout at smem[o_layout].  The kernels
// index the layout tables by a run-time layer number; from the kernarg segment that is a
// ~1 us global load per lookup, from LDS a broadcast ds_read.  Ends with a barrier.
__device__ __forceinline__ const MlpLayout &stage_layout(const MlpLayout &Lk, float *smem,
                                                        int total_floats, int o_layout) {
  zero_lds(smem, total_floats);
  __syncthreads();
  const int *src = reinterpret_cast<const int *>(&Lk);
  int *dst = reinterpret_cast<int *>(smem + o_layout);
  for (int i = threadIdx.x; i < (int)(sizeof(MlpLayout) / 4); i += blockDim.x) dst[i] = src[i];
  __syncthreads();
  return *reinterpret_cast<const MlpLayout *>(smem + o_layout);
}

#define BORE_LAYOUT_FLOATS ((int)((sizeof(MlpLayout) + 15) / 16 * 4))
// A value every lane of the wave holds alike, moved to scalar registers (v_readfirstlane): what is read
// through a pointer into ordinary global memory -- a loop's id, its iteration count -- arrives in vector
// registers, and everything derived from it (data-set size, trip counts, base addresses) then stays
// there: loops over it become execution-mask loops, tests vector compares.
__device__ __forceinline__ int uniform_i32(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uniform_i64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v & 0xffffffffULL));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}

// fit_body's `stage`: one slot per work-item and k-chunk of the first layer (static shapes 1, 2: at most two)
// plus one for the label -- each lane's share of its next-step row (bore_hip.hip, pipe_perm)
#define BORE_FIT_STAGE_FLOATS (3 * BORE_THREADS)
// (16->32-32-32-1 and the fit-only shapes -- 16 inputs -- have four k-chunks in their first layer)
// (2->16-16-1 has one: two slots, which is part of what lets three loops of the fused kernel share a CU's LDS)
#define BORE_FIT_STAGE_FLOATS_OF(shape) ((shape) >= 5 ? 5 * BORE_THREADS : (shape) == 1 ? 2 * BORE_THREADS : BORE_FIT_STAGE_FLOATS)

// Start of every kernel: zero the LDS, then hand out the layout.  Three kernel flavours:
//   SHAPE > 0   a shape of mlp_shapes.h: the caller holds a constexpr layout (everything folds)
//   SHAPE < 0   any widths, -SHAPE layers: the layer loops are unrolled, so the layout tables
//               are read from the kernarg segment at CONSTANT offsets (scalar loads)
//   SHAPE == 0  any widths, any depth: tables indexed at run time -> LDS copy
// Ends with a barrier.
template <int SHAPE>
__device__ __forceinline__ const MlpLayout &begin_kernel(const MlpLayout &Lstatic,
                                                        const MlpLayout &Lk, float *smem,
                                                        int total_floats, int o_layout) {
  if constexpr (SHAPE == 0) {
    return stage_layout(Lk, smem, total_floats, o_layout);
  } else {
    zero_lds(smem, total_floats);
    __syncthreads();
    if constexpr (SHAPE > 0) return Lstatic;
    else return Lk;
  }
}

// layer count as a compile-time constant where the kernel flavour fixes it
template <int SHAPE>
__device__ __forceinline__ int layer_count(const MlpLayout &L) {
  if constexpr (SHAPE < 0) return -SHAPE;
  else return L.n_layers;
}

// HBM -> LDS: packed vector into the (already zeroed) padded image; coalesced reads.
__device__ __forceinline__ void load_theta(const MlpLayout &L, int n, const float *__restrict__ g,
                                           float *th) {
  for (int p = threadIdx.x; p < L.P; p += blockDim.x) th[param_ref(L, p, n).lds] = g[p];
}

__device__ __forceinline__ void store_theta(const MlpLayout &L, int n, const float *th,
                                            float *__restrict__ g) {
  for (int p = threadIdx.x; p < L.P; p += blockDim.x) g[p] = th[param_ref(L, p, n).lds];
}

// One 16x16 tile: sum over kchunks*4 of A[m][k] * B[k][n].  ap / bp are THIS LANE's operand
// addresses for k-chunk 0; sa / sb the address step per k-chunk (4 values of k).
// Operands are fetched four k-chunks at a time (8 independent ds_reads in flight) ahead of
// the dependent MFMA chain.
__device__ __forceinline__ f32x4 tile_mma(const float *ap, int sa, const float *bp, int sb,
                                          int kchunks) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  int kc = 0;
  for (; kc + 4 <= kchunks; kc += 4) {
    const float a0 = ap[kc * sa], a1 = ap[(kc + 1) * sa], a2 = ap[(kc + 2) * sa],
                a3 = ap[(kc + 3) * sa];
    const float b0 = bp[kc * sb], b1 = bp[(kc + 1) * sb], b2 = bp[(kc + 2) * sb],
                b3 = bp[(kc + 3) * sb];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
  }
  for (; kc < kchunks; ++kc)
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kc * sa], bp[kc * sb], acc, 0, 0, 0);
  return acc;
}

// Rows [16 rb, 16 rb + 16) of A_l = act_l(A_{l-1} W_l + b_l), by the calling wave.
template <bool FIT = false>
__device__ __forceinline__ void fwd_rowblock(const MlpLayout &L, const float *th, float *tile,
                                             int l, int rb, bool keep_logits) {
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  const int lda_in = L.lda[l - 1], lda_out = L.lda[l], ldw = L.ldw[l];
  const float *ap = tile + L.aoff[l - 1] + (rb * 16 + m) * lda_in + q;
  const float *W = th + L.woff[l];
  const float *bias = th + L.boff[l];
  float *Aout = tile + L.aoff[l] + (rb * 16 + q * 4) * lda_out;
  const int kch = (L.w[l - 1] + 3) >> 2;
  const int a = keep_logits ? BORE_ACT_LINEAR : L.act[l];
  const int ncb = L.Np[l] >> 4, wl = L.w[l];
#pragma unroll
  for (int cb = 0; cb < ncb; ++cb) {
    const int col = cb * 16 + m;
    const f32x4 acc = tile_mma(ap, 4, W + q * ldw + col, 4 * ldw, kch);
    const bool valid = col < wl;
    const float b = valid ? bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) Aout[r * lda_out + col] = valid ? act_fwd<FIT>(a, acc[r] + b) : 0.f;
  }
}

// Rows [16 rb, 16 rb + 16) of D_{l-1} = (D_l W_l^T) .* act'_{l-1}(A_{l-1}).
__device__ __forceinline__ void bwd_rowblock(const MlpLayout &L, const float *th, float *tile,
                                             int l, int rb) {
  const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
  const int ld_in = L.lda[l], ld_out = L.lda[l - 1], ldw = L.ldw[l];
  const float *ap = tile + L.doff[l] + (rb * 16 + m) * ld_in + q;
  const float *W = th + L.woff[l];
  float *Dout = tile + L.doff[l - 1] + (rb * 16 + q * 4) * ld_out;
  const float *Aprev = tile + L.aoff[l - 1] + (rb * 16 + q * 4) * ld_out;
  const int kch = (L.w[l] + 3) >> 2;
  const int a = L.act[l - 1];
  const int nkb = L.Np[l - 1] >> 4, wp = L.w[l - 1];
#pragma unroll
  for (int kb = 0; kb < nkb; ++kb) {
    const int col = kb * 16 + m;  // input index of layer l == column of D_{l-1}
    const f32x4 acc = tile_mma(ap, 4, W + col * ldw + q, 4, kch);
    const bool valid = col < wp;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = valid ? acc[r] : 0.f;
      if (l > 1 && valid) v *= act_grad(a, Aprev[r * ld_out + col]);
      Dout[r * ld_out + col] = v;
    }
  }
}

// Forward through every layer for one row-block (calling wave).
template <bool FIT = false>
__device__ __forceinline__ void fwd_all(const MlpLayout &L, int n, const float *th, float *tile,
                                        int rb, bool keep_logits) {
#pragma unroll
  for (int l = 1; l <= n; ++l) {
    fwd_rowblock<FIT>(L, th, tile, l, rb, keep_logits && l == n);
    wave_lds_sync();
  }
}

// Objective + input gradient for one row-block whose rows sit in A_0: forward, T(sign*f)
// into val_out[row] (LDS or global, indexed by tile row), d T / d x into D_0.
__device__ __forceinline__ void fg_rowblock(const MlpLayout &L, int n, const float *th,
                                            float *tile, int rb, int transform, float sign,
                                            float *val_out) {
  const int lane = threadIdx.x & 63;
  fwd_all(L, n, th, tile, rb, false);
  if (lane < 16) {
    const int row = rb * 16 + lane;
    const float f = tile[L.aoff[n] + row * L.lda[n]];
    const float u = sign * f;
    float T, dT;
    if (transform == BORE_T_SIGMOID) {
      T = sigmoid_stable(u);
      dT = T * (1.f - T);
    } else if (transform == BORE_T_EXP) {
      T = expf(u);
      dT = T;
    } else {
      T = u;
      dT = 1.f;
    }
    val_out[row] = T;
    tile[L.doff[n] + row * L.lda[n]] = sign * dT * act_grad(L.act[n], f);
  }
  wave_lds_sync();
#pragma unroll
  for (int l = n; l >= 1; --l) {
    bwd_rowblock(L, th, tile, l, rb);
    wave_lds_sync();
  }
}

// ---- shuffle stream ---------------------------------------------------------
// One permutation of range(N) per (seed, model, epoch): row i draws the 32-bit key
// mix(base + (i+1)*C3) >> 32 and the permutation lists the rows by ascending
// (key, i).  bore_amd/shuffle.py holds the identical numpy statement.
__host__ __device__ __forceinline__ unsigned long long mix64(unsigned long long z) {
  z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27; z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}

__host__ __device__ __forceinline__ unsigned long long shuffle_base(unsigned long long seed,
                                                                    long long model,
                                                                    long long epoch) {
  unsigned long long h = mix64(seed + 0x9E3779B97F4A7C15ULL * (unsigned long long)(model + 1));
  return mix64(h + 0xD1B54A32D192ED03ULL * (unsigned long long)(epoch + 1));
}

__host__ __device__ __forceinline__ unsigned shuffle_key(unsigned long long base, int i) {
  return (unsigned)(mix64(base + 0x8CB92BA72F3D8DD7ULL * (unsigned long long)(i + 1)) >> 32);
}

// floats of LDS scratch make_perm needs: round_up(N, 16) 64-bit words + N rank counters
__host__ __device__ __forceinline__ long long perm_scratch_floats(long long N) {
  return 2 * ((N + 15) & ~15LL) + ((N + 3) & ~3LL) + (N > 128 ? 8 : 0);  // (+ 8: make_perm_buckets' wave totals)
}

// More than 128 rows: rank by BUCKETS instead of all pairs (N^2 / threads compares: 8.4 k cycles per shuffle at
// 256 rows, a sixth of the 6->32-32-1 fit's step at four steps per epoch, 134 k at 1024 rows).  NB = the largest
// power of two <= N buckets by the top bits of the 32-bit key, about one row each.  Count the rows per bucket
// (LDS atomics), scan the counts, drop every row's word (key << 32 | row) into its bucket's range of a second array
// (the order inside a bucket is the order of arrival: any), then every POSITION ranks its word among the members
// of its bucket: rank = bucket's first position + members below.  The same words and the same rule as the
// all-pairs count: the same permutation.  Scratch: the same perm_scratch_floats(N) -- N 64-bit words, then NB <= N
// counters, then 8 ints.  Ends with a barrier.
__device__ __forceinline__ int wave_inclusive_scan(int v);
// ONE_WAVE: the calling wave does it alone (wave-level LDS hand-overs instead of barriers), in three STAGES that
// keep their state in the scratch -- the eight-wave fit's fifth wave draws the next epoch's shuffle this way, one
// stage in each of the epoch's first three steps, while the first four waves run the forward / backward pass.
// Stages: 0 count the rows per bucket, 1 scan the counts and scatter the words, 2 rank inside the buckets.
template <bool ONE_WAVE = false>
__device__ __forceinline__ void make_perm_buckets(unsigned long long base, int N, unsigned *keys, int *perm_out,
                                                  int stage_lo = 0, int stage_hi = 2) {
  const int lane = threadIdx.x & 63, wv = ONE_WAVE ? 0 : (int)(threadIdx.x >> 6);
  const int tid = ONE_WAVE ? lane : (int)threadIdx.x, nthr = ONE_WAVE ? 64 : (int)blockDim.x;
  auto sync = [] {
    if constexpr (ONE_WAVE) wave_lds_sync();
    else __syncthreads();
  };
  const int N16 = (N + 15) & ~15;
  unsigned long long *mem = reinterpret_cast<unsigned long long *>(keys);  // [N] words, bucket by bucket
  int *cur = reinterpret_cast<int *>(mem + N16);                           // [NB] count -> first -> end
  int *tot = cur + ((N + 3) & ~3);                                          // [8] wave totals
  int lg = 0;
  while ((2 << lg) <= N) ++lg;
  const int NB = 1 << lg, sh = 32 - lg;  // (N > 128: lg >= 7)
  if (stage_lo <= 0 && 0 <= stage_hi) {
    for (int b = tid; b < NB; b += nthr) cur[b] = 0;
    sync();
    for (int i = tid; i < N; i += nthr) atomicAdd(&cur[shuffle_key(base, i) >> sh], 1);
    sync();
  }
  if (stage_lo <= 1 && 1 <= stage_hi) {
    // exclusive scan of the NB counts in place: thread t owns the `per` consecutive buckets from t * per
    const int per = (NB + nthr - 1) / nthr, b0 = tid * per;
    int sum = 0;
    for (int b = b0; b < min(b0 + per, NB); ++b) sum += cur[b];
    const int incl = wave_inclusive_scan(sum);
    int run = incl - sum;
    if constexpr (!ONE_WAVE) {
      if (lane == 63) tot[wv] = incl;
      __syncthreads();
      for (int w = 0; w < wv; ++w) run += tot[w];
    }
    for (int b = b0; b < min(b0 + per, NB); ++b) {
      const int c = cur[b];
      cur[b] = run;
      run += c;
    }
    sync();
    // (a row's key is drawn again rather than kept: held in registers across the scan it was slower)
    for (int i = tid; i < N; i += nthr) {
      const unsigned key = shuffle_key(base, i);
      mem[atomicAdd(&cur[key >> sh], 1)] = ((unsigned long long)key << 32) | (unsigned)i;
    }
    sync();  // (cur[b] is now the END of bucket b = the first position of bucket b + 1)
  }
  if (stage_lo <= 2 && 2 <= stage_hi) {
    for (int p = tid; p < N; p += nthr) {
      const unsigned long long w = mem[p];
      const int b = (int)((unsigned)(w >> 32) >> sh);
      const int lo = b ? cur[b - 1] : 0, hi = cur[b];
      int r = lo;
      for (int q = lo; q < hi; ++q) r += mem[q] < w;
      perm_out[r] = (int)(unsigned)w;
    }
    sync();
  }
}

// keys: LDS scratch of perm_scratch_floats(N) floats, 16-byte aligned; perm_out: [N] (LDS or
// global).  Ends with a barrier.  Row i's rank is the number of 64-bit words
// (key_j << 32 | j) below its own.  The words are fetched 16 at a time (8 broadcast
// ds_read_b128 in flight); for small N the word range is split over up to 4 threads per row
// (partial counts meet in an LDS counter), so that a 64-row shuffle keeps all 256 threads busy.
__device__ __forceinline__ void make_perm(unsigned long long base, int N, unsigned *keys,
                                          int *perm_out) {
  if (N > 128) {  // (workgroup-uniform)
    make_perm_buckets(base, N, keys, perm_out);
    return;
  }
  unsigned long long *k64 = reinterpret_cast<unsigned long long *>(keys);
  // (NOT opaque here: the shuffle's per-thread constants -- a division by the padded row count among
  // them -- are wanted outside the epoch loop; made opaque the fit lost 3 %)
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int N16 = (N + 15) & ~15;
  int *rank = reinterpret_cast<int *>(k64 + N16);
  int parts = 1;
  while (parts < 4 && N * parts * 2 <= nthr) parts *= 2;
  for (int i = tid; i < N16; i += nthr)  // the padding words (all ones) are never below a key
    k64[i] = i < N ? ((unsigned long long)shuffle_key(base, i) << 32) | (unsigned)i : ~0ULL;
  if (parts > 1)
    for (int i = tid; i < N; i += nthr) rank[i] = 0;
  __syncthreads();
  const ulonglong2 *kk = reinterpret_cast<const ulonglong2 *>(keys);
  const int nbat = N16 >> 4;
  for (int u = tid; u < N * parts; u += nthr) {
    const int p = u / N, i = u - p * N;
    const unsigned long long ki = k64[i];
    const int b0 = p * nbat / parts, b1 = (p + 1) * nbat / parts;
    int r = 0;
    for (int b = b0; b < b1; ++b) {
      ulonglong2 k[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) k[j] = kk[b * 8 + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r += k[j].x < ki;
        r += k[j].y < ki;
      }
    }
    if (parts == 1) perm_out[r] = i;
    else atomicAdd(&rank[i], r);
  }
  __syncthreads();
  if (parts > 1) {
    for (int i = tid; i < N; i += nthr) perm_out[rank[i]] = i;
    __syncthreads();
  }
}

// ---- several epochs' shuffles at once -------------------------------------------------------
// A data set of N <= 128 rows leaves most of the workgroup idle in make_perm (one key per row).
// perm_group(N, threads) = how many consecutive epochs fit side by side (one thread per
// (epoch, row)): their keys are drawn together and each thread ranks its row against the N
// keys of its own epoch -- the same permutations as make_perm, three barriers per GROUP of
// epochs instead of per epoch.
__host__ __device__ __forceinline__ int perm_group(long long N, int threads) {
  const long long slot = (N + 63) & ~63LL;  // rows of one epoch occupy whole waves
  const long long g = threads / slot;
  return g >= 4 ? 4 : g >= 2 ? 2 : 1;  // (a power of two: the kernels mask the epoch index)
}
__host__ __device__ __forceinline__ long long perm_group_scratch_floats(long long N, int G) {
  return G <= 1 ? perm_scratch_floats(N) : (long long)G * 2 * ((N + 15) & ~15LL);
}

// perm_out: [G][N] (LDS); epochs epoch .. epoch + n_ep - 1 (n_ep <= G) of `model`.
__device__ __forceinline__ void make_perm_group(unsigned long long seed, long long model,
                                                long long epoch, int n_ep, int N, unsigned *keys,
                                                int *perm_out) {
  unsigned long long *k64 = reinterpret_cast<unsigned long long *>(keys);
  const int tid = threadIdx.x;
  const int N16 = (N + 15) & ~15, slot = (N + 63) & ~63;
  const int g = tid / slot, i = tid - g * slot;  // this thread's (epoch in the group, row)
  const bool mine = g < n_ep && i < N16;
  if (mine) {
    const unsigned long long base = shuffle_base(seed, model, epoch + g);
    k64[g * N16 + i] = i < N ? ((unsigned long long)shuffle_key(base, i) << 32) | (unsigned)i : ~0ULL;
  }
  __syncthreads();
  if (mine && i < N) {
    const ulonglong2 *kk = reinterpret_cast<const ulonglong2 *>(k64 + g * N16);
    const unsigned long long ki = k64[g * N16 + i];
    int r = 0;
    for (int b = 0; b < (N16 >> 4); ++b) {
      ulonglong2 k[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) k[j] = kk[b * 8 + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r += k[j].x < ki;
        r += k[j].y < ki;
      }
    }
    perm_out[g * N + r] = i;
  }
  __syncthreads();
}


// ONE epoch's permutation by ONE wave (N <= 128), no workgroup barrier: what the fit's idle wave
// does for the NEXT epoch while the other waves run the current step's forward / backward (a
// static-shape fit of at most 48 rows per last step leaves its fourth wave without rows; formed by
// the whole workgroup between two steps, the shuffles were a sixth of an Adam step of the
// 2->16-16-1 fit: profiles/r3/fit_marks.txt).  Same keys, same ranking rule as make_perm: the same
// permutation.  k64: round_up(N, 16) words of scratch that only this wave touches.
__device__ __forceinline__ void make_perm_wave(unsigned long long base, int N,
                                               unsigned long long *k64, int *perm_out) {
  const int lane = threadIdx.x & 63;
  const int N16 = (N + 15) & ~15;
  for (int i = lane; i < N16; i += 64)  // the padding words (all ones) are never below a key
    k64[i] = i < N ? ((unsigned long long)shuffle_key(base, i) << 32) | (unsigned)i : ~0ULL;
  wave_lds_sync();
  const ulonglong2 *kk = reinterpret_cast<const ulonglong2 *>(k64);
  for (int i = lane; i < N; i += 64) {
    const unsigned long long ki = k64[i];
    int r = 0;
    for (int b = 0; b < (N16 >> 4); ++b) {
      ulonglong2 k[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) k[j] = kk[b * 8 + j];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        r += k[j].x < ki;
        r += k[j].y < ki;
      }
    }
    perm_out[r] = i;
  }
}

// The same permutation by ONE wave in a third of the cycles at 100 rows (65 <= N <= 128): bucket
// ranking.  A row's rank = (rows in buckets below its own) + (rows of its bucket with a smaller word);
// the bucket is the top 7 bits of the row's 32-bit key, so the 128 buckets hold about one row each.
// Rows enter their bucket's member list through an LDS counter (the ORDER of arrival does not matter:
// a row is compared with every member), a wave scan of the counts gives the buckets' first ranks, and
// every row walks its own bucket (<= CAP members, predicated compares).  Members are 32-bit: inside a
// bucket the top 7 key bits agree, so (low 25 key bits << 7 | row) orders like (key << 32 | row).
// Exact: the same permutation as make_perm.  A bucket with more than CAP rows (~2e-4 of the shuffles
// of 128 rows) sends the whole shuffle through make_perm_wave.
// scratch: BORE_PERM_WAVE_FLOATS floats, 16-byte aligned, touched by this wave only.
constexpr int BORE_PERM_BUCKETS = 128, BORE_PERM_CAP = 8;
#define BORE_PERM_WAVE_FLOATS (BORE_PERM_BUCKETS + BORE_PERM_BUCKETS * BORE_PERM_CAP)
// inclusive prefix sum over the 64 lanes: four DPP shifts inside each row of 16, the rows' totals by
// v_readlane (no LDS round trips)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31),
            t2 = __builtin_amdgcn_readlane(v, 47);
  const int row = (threadIdx.x & 63) >> 4;
  return v + (row > 0 ? t0 : 0) + (row > 1 ? t1 : 0) + (row > 2 ? t2 : 0);
}
__device__ __forceinline__ void make_perm_wave_buckets(unsigned long long base, int N,
                                                       unsigned long long *scratch, int *perm_out) {
  const int lane = threadIdx.x & 63;
  int *cnt = reinterpret_cast<int *>(scratch);                            // [128] counts, then first rank | count << 16
  unsigned *mem = reinterpret_cast<unsigned *>(scratch) + BORE_PERM_BUCKETS;  // [128][CAP] member words
  cnt[lane] = 0;
  cnt[lane + 64] = 0;
  wave_lds_sync();
  unsigned w[2];
  int b[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = lane + 64 * j;
    const unsigned key = shuffle_key(base, i);
    w[j] = (key << 7) | (unsigned)i;  // (i < 128)
    b[j] = (int)(key >> 25);
  }
  int slot[2] = {0, 0};
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (lane + 64 * j < N) slot[j] = atomicAdd(&cnt[b[j]], 1);
#pragma unroll
  for (int j = 0; j < 2; ++j)
    if (lane + 64 * j < N && slot[j] < BORE_PERM_CAP) mem[b[j] * BORE_PERM_CAP + slot[j]] = w[j];
  wave_lds_sync();
  const int2 c01 = *reinterpret_cast<const int2 *>(cnt + 2 * lane);  // this lane's two buckets
  if (__any(c01.x > BORE_PERM_CAP || c01.y > BORE_PERM_CAP)) {       // (wave-uniform; the member area is free again)
    make_perm_wave(base, N, scratch + BORE_PERM_BUCKETS / 2, perm_out);
    return;
  }
  const int first = wave_inclusive_scan(c01.x + c01.y) - (c01.x + c01.y);
  *reinterpret_cast<int2 *>(cnt + 2 * lane) = make_int2(first | (c01.x << 16), (first + c01.x) | (c01.y << 16));
  wave_lds_sync();
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = lane + 64 * j;
    if (i < N) {
      const int pc = cnt[b[j]];
      int r = pc & 0xffff;
      const int nb = pc >> 16;
      const uint4 *mb = reinterpret_cast<const uint4 *>(mem + b[j] * BORE_PERM_CAP);
      const uint4 m0 = mb[0], m1 = mb[1];
      const unsigned mm[8] = {m0.x, m0.y, m0.z, m0.w, m1.x, m1.y, m1.z, m1.w};
#pragma unroll
      for (int q = 0; q < BORE_PERM_CAP; ++q) r += (q < nb && mm[q] < w[j]) ? 1 : 0;
      perm_out[r] = i;
    }
  }
}

}  // namespace bore
