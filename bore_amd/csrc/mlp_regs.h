// mlp_regs.h -- a row-block's trip through the network with the activations in REGISTERS.
// gfx950 only; used by the kernels instantiated for a static shape (mlp_shapes.h).
//
// mlp_device.h's fwd_rowblock/bwd_rowblock pass a layer's output to the next layer through
// LDS (write, s_waitcnt, read): ~250 cycles of pure latency per layer on a chain that one wave
// walks alone.  Here the products are formed TRANSPOSED,
//     H_l^T [units x 16 rows] = W_l^T [units x in] * H_{l-1}^T [in x 16 rows]
// so the weights are the MFMA A operand (LDS, no dependence on the chain: the loads of every
// layer can be in flight from the start) and the activations the B operand.  The output of
// v_mfma_f32_16x16x4_f32 leaves lane (q, m) = (lane >> 4, lane & 15) holding
//     C[4q + r][m] = H_l[row m][unit 16t + 4q + r],   r = 0..3
// and the B operand of the next layer's k-chunk kc wants H_l[row m][unit 4 kc + q] in that
// lane: a 4x4 transpose between the register index r and the 16-lane row index q, which
// gfx950 does in four VALU instructions (v_permlane32_swap + v_permlane16_swap) -- no LDS.
// The same holds backwards (D_{l-1}^T = W_l * D_l^T, A operand = W_l), and act' needs
// H_{l-1} in exactly the C layout the forward pass left in this lane's registers.
//
// Arithmetic: identical to the LDS path, bit for bit -- the same k-ordered fmaf chain per
// output (a*b commutes), the same bias add and activation expressions.
#pragma once
#include <type_traits>

#include "mlp_device.h"

namespace bore {

// v[r] in lane-row q holds M[r][q]  ->  v[r] in lane-row q holds M[q][r]  (per column m)
__device__ __forceinline__ void rows_transpose4(float (&v)[4]) {
  // permlane32_swap(x, y): x.rows{2,3} <-> y.rows{0,1};  permlane16_swap(x, y): x.row1 <-> y.row0,
  // x.row3 <-> y.row2  (rows of 16 lanes)
  const auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]),
                                                  false, false);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]),
                                                  false, false);
  const auto c = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  const auto d = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  v[0] = __uint_as_float(c[0]);
  v[1] = __uint_as_float(c[1]);
  v[2] = __uint_as_float(d[0]);
  v[3] = __uint_as_float(d[1]);
}

// bf16 <-> fp32 (round to nearest even; NaN kept quiet)
__device__ __forceinline__ unsigned short f32_to_bf16(float x) {
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) {
  return __uint_as_float((unsigned)h << 16);
}
__device__ __forceinline__ float bf16_round(float x) { return bf16_to_f32(f32_to_bf16(x)); }

// BF16 = true: the LDS images (weights, and the A_l / D_l copies) hold bfloat16 and every
// layer output / delta is rounded to bfloat16 -- "bf16 weights and activations, fp32
// accumulate"; the products still run on the fp32 MFMA (bf16 x bf16 is exact in fp32), so the
// lane maps and the k-ordered sums are those of the fp32 path.
template <int SHAPE, int DELTAS, bool BF16 = false>
struct RegNet {
  using WT = typename std::conditional<BF16, unsigned short, float>::type;  // LDS element
  static __device__ __forceinline__ float ld(const WT *p) {
    if constexpr (BF16) return bf16_to_f32(*p);
    else return *p;
  }
  static __device__ __forceinline__ float rnd(float x) {
    if constexpr (BF16) return bf16_round(x);
    else return x;
  }
  // an LDS element as loaded (raw) and as an MFMA operand: the widening shift of a bfloat16 is
  // kept out of the load phase, where it would wait for the load to return
  static __device__ __forceinline__ float cvt(WT raw) {
    if constexpr (BF16) return bf16_to_f32(raw);
    else return raw;
  }
  static constexpr MlpLayout L = bore_static_layout(SHAPE, DELTAS, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers;
  static constexpr int max_tiles() {
    int t = 1;
    for (int l = 0; l <= L.n_layers; ++l)
      if (L.Np[l] / 16 > t) t = L.Np[l] / 16;
    return t;
  }
  static constexpr int T = max_tiles();
  static constexpr int KC0 = (L.w[0] + 3) / 4;  // k-chunks of the input layer

  // ---- operand bookkeeping (all compile-time) ----
  static constexpr int fkch(int l) { return (L.w[l - 1] + 3) / 4; }   // k-chunks, forward
  static constexpr int ftiles(int l) { return L.Np[l] / 16; }          // output tiles, forward
  static constexpr int fofs(int l) {
    int o = 0;
    for (int i = 1; i < l; ++i) o += fkch(i) * ftiles(i);
    return o;
  }
  static constexpr int biasofs(int l) {
    int o = 0;
    for (int i = 1; i < l; ++i) o += 4 * ftiles(i);
    return o;
  }
  static constexpr int bkch(int l) { return (L.w[l] + 3) / 4; }        // k-chunks, backward
  static constexpr int btiles(int l) { return L.Np[l - 1] / 16; }      // output tiles, backward
  static constexpr int bofs(int l) {  // layers are visited n, n-1, ..
    int o = 0;
    for (int i = L.n_layers; i > l; --i) o += bkch(i) * btiles(i);
    return o;
  }

  // Small nets: every layer's operands are requested up front and stay in registers (the
  // lbfgsb / rows kernels reuse them across row-blocks).  Wide nets: one output tile's operands
  // at a time, inside the layer.
  static constexpr bool PRELOAD = fofs(L.n_layers + 1) + bofs(0) <= 64;
  // Activations: part of the shape, or read at run time (set_acts) when the shape fixes only
  // the widths (mlp_shapes.h: act[0] < 0).
  static constexpr bool RT_ACT = kShapes[SHAPE].act[0] < 0;
  int acts[n + 1];
  __device__ __forceinline__ void set_acts(const MlpLayout &Lrt) {
#pragma unroll
    for (int l = 0; l <= n; ++l) acts[l] = Lrt.act[l];
  }
  template <int l>
  __device__ __forceinline__ int act_of() const {
    if constexpr (RT_ACT) return acts[l];
    else return L.act[l];
  }

  // Activation over a layer's registers.  A run-time activation id is wave-uniform: ONE scalar
  // branch per layer picks the loop (a per-element switch makes the compiler evaluate several
  // activations and select).
  template <int A, int NT>
  static __device__ __forceinline__ void act_tiles(float (&v)[T][4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][r] = act_fwd<DELTAS == 1 && !BF16>(A, v[t][r]);  // (DELTAS == 1: the fit's network)
  }
  template <int NT>
  static __device__ __forceinline__ void act_tiles_rt(int a, float (&v)[T][4]) {
    switch (__builtin_amdgcn_readfirstlane(a)) {
      case BORE_ACT_RELU: act_tiles<BORE_ACT_RELU, NT>(v); break;
      case BORE_ACT_ELU: act_tiles<BORE_ACT_ELU, NT>(v); break;
      case BORE_ACT_SIGMOID: act_tiles<BORE_ACT_SIGMOID, NT>(v); break;
      case BORE_ACT_TANH: act_tiles<BORE_ACT_TANH, NT>(v); break;
      default: break;
    }
  }
  // v *= act'(h), same dispatch
  template <int A, int NT>
  static __device__ __forceinline__ void grad_tiles(float (&v)[T][4], const float (&hh)[T][4]) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) v[t][r] *= act_grad(A, hh[t][r]);
  }
  template <int NT>
  static __device__ __forceinline__ void grad_tiles_rt(int a, float (&v)[T][4],
                                                       const float (&hh)[T][4]) {
    switch (__builtin_amdgcn_readfirstlane(a)) {
      case BORE_ACT_RELU: grad_tiles<BORE_ACT_RELU, NT>(v, hh); break;
      case BORE_ACT_ELU: grad_tiles<BORE_ACT_ELU, NT>(v, hh); break;
      case BORE_ACT_SIGMOID: grad_tiles<BORE_ACT_SIGMOID, NT>(v, hh); break;
      case BORE_ACT_TANH: grad_tiles<BORE_ACT_TANH, NT>(v, hh); break;
      default: break;  // linear: derivative 1
    }
  }

  // h[l][t][r] = A_l[row m][unit 16t + 4q + r] (l >= 1);  d[l][t][r] = D_l, same map
  float h[n + 1][T][4];
  float d[n + 1][T][4];
  // this lane's MFMA A operands (weights) and biases, fetched ahead of the chain that uses them
  float wf[PRELOAD ? fofs(n + 1) : 1], bf[PRELOAD ? biasofs(n + 1) : 1], wb[PRELOAD ? bofs(0) : 1];

  // Request every forward operand: W_l[4kc + q][16t + m] and b_l[16t + 4q + r].
  template <int l = 1>
  __device__ __forceinline__ void load_fwd(const WT *th) {
    if constexpr (PRELOAD && l <= n) {
      const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
      constexpr int kch = fkch(l), ldw = L.ldw[l];
#pragma unroll
      for (int t = 0; t < ftiles(l); ++t) {
        const WT *wp = th + L.woff[l] + q * ldw + 16 * t + m;
#pragma unroll
        for (int kc = 0; kc < kch; ++kc) wf[fofs(l) + t * kch + kc] = ld(wp + kc * 4 * ldw);
        const WT *bp = th + L.boff[l] + 16 * t + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) bf[biasofs(l) + 4 * t + r] = ld(bp + r);
      }
      load_fwd<l + 1>(th);
    }
  }

  // Request the backward operands of layers from, from-1, .., to: W_l[16t + m][4kc + q].
  template <int from, int to>
  __device__ __forceinline__ void load_bwd(const WT *th) {
    if constexpr (PRELOAD && from >= to) {
      const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
      constexpr int kch = bkch(from), ldw = L.ldw[from];
#pragma unroll
      for (int t = 0; t < btiles(from); ++t) {
        const WT *wp = th + L.woff[from] + (16 * t + m) * ldw + q;
#pragma unroll
        for (int kc = 0; kc < kch; ++kc) wb[bofs(from) + t * kch + kc] = ld(wp + kc * 4);
      }
      load_bwd<from - 1, to>(th);
    }
  }

  // B operands of a product whose inner dimension is a layer of width w held in C layout
  template <int w>
  static __device__ __forceinline__ void make_bop(const float (&src)[T][4], float (&bop)[4 * T]) {
    const int q = (threadIdx.x & 63) >> 4;
    if constexpr (w == 1) {
      bop[0] = q == 0 ? src[0][0] : 0.f;
    } else {
      constexpr int kch = (w + 3) / 4;
#pragma unroll
      for (int t = 0; t < (kch + 3) / 4; ++t) {
        float v[4] = {src[t][0], src[t][1], src[t][2], src[t][3]};
        rows_transpose4(v);
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (4 * t + j < kch) bop[4 * t + j] = v[j];
      }
    }
  }

  // A_l = act_l(A_{l-1} W_l + b_l); xin = the input rows' B operands (layer 1 only)
  template <int l>
  __device__ __forceinline__ void fwd_layer(const WT *th, const float (&xin)[KC0],
                                            bool keep_logits) {
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    constexpr int kch = fkch(l), ldw = L.ldw[l];
    float bop[4 * T];
    if constexpr (l == 1) {
#pragma unroll
      for (int kc = 0; kc < KC0; ++kc) bop[kc] = xin[kc];
    } else {
      make_bop<L.w[l - 1]>(h[l - 1], bop);
    }
    const int a = (keep_logits && l == n) ? BORE_ACT_LINEAR : act_of<l>();
    // Wide nets fetch one output tile's operands at a time -- W_l[4kc + q][16t + m],
    // b_l[16t + 4q + r] -- and ONE TILE AHEAD of the MFMA chain that uses them: left to the
    // scheduler (at its register limit in these kernels) each ds_read sat directly in front of
    // its MFMA with a full s_waitcnt -- an LDS round trip per 32-cycle MFMA, forward and backward
    // 2.8x their MFMA time (profiles/r2: wide_stamps).
    WT wraw[PRELOAD ? 1 : 2][PRELOAD ? 1 : kch], braw[PRELOAD ? 1 : 2][4];
    if constexpr (!PRELOAD) {
      const WT *wp = th + L.woff[l] + q * ldw + m;
      const WT *bp = th + L.boff[l] + 4 * q;
#pragma unroll
      for (int kc = 0; kc < kch; ++kc) wraw[0][kc] = wp[kc * 4 * ldw];
#pragma unroll
      for (int r = 0; r < 4; ++r) braw[0][r] = bp[r];
    }
#pragma unroll
    for (int t = 0; t < ftiles(l); ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      float bias[4];
      if constexpr (PRELOAD) {
#pragma unroll
        for (int kc = 0; kc < kch; ++kc)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[fofs(l) + t * kch + kc], bop[kc], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = bf[biasofs(l) + 4 * t + r];
      } else {
        if (t + 1 < ftiles(l)) {  // the next tile's operands
          const WT *wp = th + L.woff[l] + q * ldw + 16 * (t + 1) + m;
          const WT *bp = th + L.boff[l] + 16 * (t + 1) + 4 * q;
#pragma unroll
          for (int kc = 0; kc < kch; ++kc) wraw[(t + 1) & 1][kc] = wp[kc * 4 * ldw];
#pragma unroll
          for (int r = 0; r < 4; ++r) braw[(t + 1) & 1][r] = bp[r];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < kch; ++kc)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cvt(wraw[t & 1][kc]), bop[kc], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) bias[r] = cvt(braw[t & 1][r]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (RT_ACT) {  // pre-activations; the activation follows for the whole layer
#pragma unroll
        for (int r = 0; r < 4; ++r) h[l][t][r] = acc[r] + bias[r];
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool valid = 16 * t + 4 * q + r < L.w[l];
          h[l][t][r] = valid ? rnd(act_fwd<DELTAS == 1 && !BF16>(a, acc[r] + bias[r])) : 0.f;
        }
      }
    }
    if constexpr (RT_ACT) {
      act_tiles_rt<ftiles(l)>(a, h[l]);
#pragma unroll
      for (int t = 0; t < ftiles(l); ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          h[l][t][r] = 16 * t + 4 * q + r < L.w[l] ? rnd(h[l][t][r]) : 0.f;  // padding stays zero
    }
  }

  // D_{l-1} = (D_l W_l^T) .* act'_{l-1}(A_{l-1})
  template <int l>
  __device__ __forceinline__ void bwd_layer(const WT *th) {
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    constexpr int kch = bkch(l), ldw = L.ldw[l];
    float bop[4 * T];
    make_bop<L.w[l]>(d[l], bop);
    const int ap = act_of<(l > 1 ? l - 1 : 1)>();
    WT wraw[PRELOAD ? 1 : 2][PRELOAD ? 1 : kch];  // W_l[16t + m][4kc + q], one tile ahead (fwd_layer)
    if constexpr (!PRELOAD) {
      const WT *wp = th + L.woff[l] + m * ldw + q;
#pragma unroll
      for (int kc = 0; kc < kch; ++kc) wraw[0][kc] = wp[kc * 4];
    }
#pragma unroll
    for (int t = 0; t < btiles(l); ++t) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if constexpr (PRELOAD) {
#pragma unroll
        for (int kc = 0; kc < kch; ++kc)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[bofs(l) + t * kch + kc], bop[kc], acc, 0, 0, 0);
      } else {
        if (t + 1 < btiles(l)) {
          const WT *wp = th + L.woff[l] + (16 * (t + 1) + m) * ldw + q;
#pragma unroll
          for (int kc = 0; kc < kch; ++kc) wraw[(t + 1) & 1][kc] = wp[kc * 4];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kc = 0; kc < kch; ++kc)
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(cvt(wraw[t & 1][kc]), bop[kc], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool valid = 16 * t + 4 * q + r < L.w[l - 1];
        float v = valid ? acc[r] : 0.f;
        if constexpr (!RT_ACT) {
          if (l > 1 && valid) v *= act_grad(ap, h[l - 1][t][r]);
          v = rnd(v);
        }
        d[l - 1][t][r] = v;
      }
    }
    if constexpr (RT_ACT && l > 1) {
      grad_tiles_rt<btiles(l)>(ap, d[l - 1], h[l - 1]);
      if constexpr (BF16) {
#pragma unroll
        for (int t = 0; t < btiles(l); ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) d[l - 1][t][r] = rnd(d[l - 1][t][r]);
      }
    }
  }

  // (small nets: operands requested by load_fwd; wide nets: fetched per tile from th)
  template <int l = 1>
  __device__ __forceinline__ void forward(const WT *th, const float (&xin)[KC0],
                                          bool keep_logits) {
    if constexpr (l <= n) {
      fwd_layer<l>(th, xin, keep_logits);
      forward<l + 1>(th, xin, keep_logits);
    }
  }

  // bwd_layer for l = from, from-1, ..., to  (leaves D_{to-1}; operands requested by load_bwd)
  template <int from, int to>
  __device__ __forceinline__ void backward(const WT *th) {
    if constexpr (from >= to) {
      bwd_layer<from>(th);
      backward<from - 1, to>(th);
    }
  }

  // d loss / d logit (or d objective / d pre-activation of the output unit) of row m, given by
  // the lanes < 16; every other slot of D_n is zero.
  __device__ __forceinline__ void set_output_delta(float delta) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) d[n][t][r] = 0.f;
    d[n][0][0] = lane < 16 ? rnd(delta) : 0.f;
  }

  // Objective + input gradient (the register form of fg_rowblock): returns T(sign*f) of row
  // m in the lanes < 16; D_0 (= d T / d x) is left in d[0].  Operands: load_fwd + load_bwd<n, 1>.
  __device__ __forceinline__ float fg(const WT *th, const float (&xin)[KC0], int transform,
                                      float sign) {
    forward(th, xin, false);
    const float f = h[n][0][0];
    const float u = sign * f;
    float Tv, dT;
    if (transform == BORE_T_SIGMOID) {
      Tv = sigmoid_stable(u);
      dT = Tv * (1.f - Tv);
    } else if (transform == BORE_T_EXP) {
      Tv = expf(u);
      dT = Tv;
    } else {
      Tv = u;
      dT = 1.f;
    }
    set_output_delta(sign * dT * act_grad(act_of<n>(), f));
    backward<n, 1>(th);
    return Tv;
  }

  // float32 rows regardless of the compute type (staging for per-lane consumers)
  template <int l>
  static __device__ __forceinline__ void store_rows_f32(const float (&src)[T][4], float *img,
                                                        int rb) {
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    float *p = img + (rb * 16 + m) * L.lda[l] + 4 * q;
#pragma unroll
    for (int t = 0; t < L.Np[l] / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) p[16 * t + r] = src[t][r];
  }

  template <int l, int hi>
  __device__ __forceinline__ void store_A(WT *tile, int rb) const {
    if constexpr (l <= hi) {
      store_rows<l>(h[l], tile + L.aoff[l], rb);
      store_A<l + 1, hi>(tile, rb);
    }
  }
  template <int l, int hi>
  __device__ __forceinline__ void store_D(WT *tile, int rb) const {
    if constexpr (l <= hi) {
      store_rows<l>(d[l], tile + L.doff[l], rb);
      store_D<l + 1, hi>(tile, rb);
    }
  }

  // Store layer l's C-layout registers into the padded LDS image rows [16 rb, 16 rb + 16)
  // (the weight-gradient phase reads A_l / D_l of ALL rows from there).
  template <int l>
  static __device__ __forceinline__ void store_rows(const float (&src)[T][4], WT *img, int rb) {
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    WT *p = img + (rb * 16 + m) * L.lda[l] + 4 * q;
#pragma unroll
    for (int t = 0; t < L.Np[l] / 16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (BF16) p[16 * t + r] = f32_to_bf16(src[t][r]);  // (exact: already rounded)
        else p[16 * t + r] = src[t][r];
      }
  }
};

// HBM -> LDS image of theta: float32 (load_theta) or rounded to bfloat16 (2-byte elements at the
// same padded indices).
template <bool BF16>
__device__ __forceinline__ void stage_theta(const MlpLayout &L, int n, const float *__restrict__ g,
                                            float *smem) {
  if constexpr (BF16) {
    unsigned short *t16 = reinterpret_cast<unsigned short *>(smem);
    for (int p = threadIdx.x; p < L.P; p += blockDim.x) t16[param_ref(L, p, n).lds] = f32_to_bf16(g[p]);
  } else {
    load_theta(L, n, g, smem);
  }
}

// The same for a static shape's kernel serving a net of FEWER inputs (d_in <= L.w[0]; mlp_shapes.h:
// bore_acq_flavour): the packed vector holds d_in rows of W_1 (its first block, row-major [in][out]) where the
// static layout counts L.w[0] -- same prefix, then everything behind W_1 shifted by the missing rows; the rows
// d_in .. of the LDS image keep the zeros begin_kernel put there.  `g_all`: the first model's vector.
template <bool BF16>
__device__ __forceinline__ void stage_theta_in(const MlpLayout &L, int n, const float *__restrict__ g_all,
                                               long long model, int d_in, float *smem) {
  const int gap = (L.w[0] - d_in) * L.w[1], head = d_in * L.w[1], P_in = L.P - gap;
  const float *g = g_all + model * P_in;
  if (gap == 0) {
    stage_theta<BF16>(L, n, g, smem);
    return;
  }
  static_assert(!BF16, "float32 nets");
  for (int p = threadIdx.x; p < P_in; p += blockDim.x) smem[param_ref(L, p < head ? p : p + gap, n).lds] = g[p];
}

}  // namespace bore
