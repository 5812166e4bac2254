// mlp_point.h -- objective and input gradient of ONE point per wave on the vector ALU.
//
// The restart kernel's one-problem-per-wave mode evaluates a single point per round.  Through
// the matrix pipeline that is a 16-row block with 15 dead rows: 680 matrix instructions of 32 cycles
// for 32->128-128-1, 52 k cycles per evaluation with their operand fetches (profiles/r2:
// lbfgsb_phase_stamps).  A matrix-vector product needs 1/16 of that arithmetic, and
// v_mfma_f32_16x16x4_f32 is bit for bit a k-ordered fmaf chain (mlp_device.h), so the SAME numbers
// come out of a plain chain on the vector ALU:
//   * unit u of a layer lives in lane u & 63, slot u >> 6;
//   * forward:  acc_u = fmaf(W_l[k][u], a_k, acc_u), k = 0 .. K-1 in order -- a_k by v_readlane from
//     the lane that holds it (compile-time lane), W_l[k][u] one conflict-free LDS read per k for
//     the 64 units of a slot; then act(acc + b), rounded to bfloat16 when the compute type is;
//   * backward: d_{l-1}[k] = (sum_j fmaf(W_l[k][j], d_l[j]))  .* act'(a_{l-1}[k]), j in order;
//   * a one-unit last layer is the same chain formed by every lane (its value is wave-uniform).
// ~2.3 k instructions per evaluation for 32->128-128-1 instead of 52 k cycles; the row kernels
// (predict, value + gradient of many rows, screening) keep the matrix pipeline, where all 16 rows
// of a block are live.  Equality with those kernels is what the optimiser tests assert (the host
// build of the optimiser is fed through bore_mlp_value_and_input_grad and must reproduce the
// device run bit for bit).  Measured (profiles/r2/lbfgsb_phase_stamps.txt): cycles per evaluation
// 17.6 k -> 9.5 k (16->64-64-64-1), 52 k -> 33 k (32->128-128-1 bf16), 5.8 k -> 6.3 k (6->32-32-1).
// The 2->16-16-1 net keeps the matrix path: in the fused iteration kernel the vector form measured
// 4 % slower (A/B, round 2); round 3 tried it again with every layer's operands requested before
// the first term (one LDS round trip instead of six, 70 registers): restart phase 399.8 against
// 398.9 us per loop-iteration, same bits -- a 16-term readlane + fmac chain per layer is as long as
// the matrix path's four dependent matrix instructions plus transposes.  Not kept.
#pragma once
#include "mlp_regs.h"

namespace bore {

// Activations picked at run time (shapes whose widths alone are static), as REAL calls: inlined, the
// four activation bodies (exp, expm1, tanh ...) at each of a dozen sites made the 128-wide restart
// kernel spill 320 registers (668 B of scratch per lane; 156 B with relu alone inlined).  Leaf
// functions of a dozen registers: the call costs ~50 cycles, a layer of 128 units takes thousands.
// Same expressions as act_fwd / act_grad inlined: same bits.
static __device__ __attribute__((noinline)) float point_act_call(int a, float x) {
  switch (a) {
    case BORE_ACT_RELU: return act_fwd(BORE_ACT_RELU, x);
    case BORE_ACT_ELU: return act_fwd(BORE_ACT_ELU, x);
    case BORE_ACT_SIGMOID: return act_fwd(BORE_ACT_SIGMOID, x);
    case BORE_ACT_TANH: return act_fwd(BORE_ACT_TANH, x);
    default: return x;
  }
}
static __device__ __attribute__((noinline)) float point_grad_call(int a, float v, float hh) {
  switch (a) {
    case BORE_ACT_RELU: return v * act_grad(BORE_ACT_RELU, hh);
    case BORE_ACT_ELU: return v * act_grad(BORE_ACT_ELU, hh);
    case BORE_ACT_SIGMOID: return v * act_grad(BORE_ACT_SIGMOID, hh);
    case BORE_ACT_TANH: return v * act_grad(BORE_ACT_TANH, hh);
    default: return v;  // linear: derivative 1
  }
}

#define BORE_POINT_KB(U) ((U) > 1 ? 8 : 16)

template <int SHAPE, bool BF16 = false>
struct PointNet {
  using R = RegNet<SHAPE, 2, BF16>;
  using WT = typename R::WT;
  static constexpr MlpLayout L = R::L;
  static constexpr int n = L.n_layers;
  static constexpr int slots(int l) { return (L.w[l] + 63) / 64; }
  static constexpr int max_slots() {
    int s = 1;
    for (int l = 0; l <= L.n_layers; ++l)
      if (slots(l) > s) s = slots(l);
    return s;
  }
  static constexpr int U = max_slots();
  // terms whose operands are in flight together (two slots per lane double the registers a batch
  // takes: 128-wide layers batch 8 terms)
  static constexpr int KB = BORE_POINT_KB(U);
  float h[n + 1][U];  // h[l][s] = A_l[unit lane + 64 s]
  float d[n + 1][U];
  int acts[n + 1];

  __device__ __forceinline__ void set_acts(const MlpLayout &Lrt) {
#pragma unroll
    for (int l = 0; l <= n; ++l) acts[l] = Lrt.act[l];
  }
  template <int l>
  __device__ __forceinline__ int act_of() const {
    if constexpr (R::RT_ACT) return acts[l];
    else return L.act[l];
  }
  // act_fwd / act_grad with a wave-uniform run-time id: one scalar branch, then the same
  // constant-id code the matrix path runs (act_tiles / grad_tiles)
  static __device__ __forceinline__ float act_rt(int a, float x) {
    return point_act_call(__builtin_amdgcn_readfirstlane(a), x);
  }
  static __device__ __forceinline__ float grad_rt(int a, float v, float hh) {
    return point_grad_call(__builtin_amdgcn_readfirstlane(a), v, hh);
  }
  static __device__ __forceinline__ float lane_value(float v, int src) {  // src: compile-time after unrolling
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src));
  }

  // A_l = act_l(W_l^T a_{l-1} + b_l)
  template <int l>
  __device__ __forceinline__ void fwd_layer(const WT *th) {
    const int lane = threadIdx.x & 63;
    constexpr int K = L.w[l - 1], Nw = L.w[l], ldw = L.ldw[l], S = slots(l);
    float acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.f;
    // operands are requested KB values of k at a time, one batch ahead of the chain that uses them
    // (left to the scheduler every ds_read sat in front of its fmaf with a full s_waitcnt: an LDS
    // round trip per two terms)
    const WT *wp = Nw == 1 ? th + L.woff[l] : th + L.woff[l] + lane;
    WT wb[2][KB][S];
    auto fetch = [&](int buf, int k0) {
#pragma unroll
      for (int kk = 0; kk < KB; ++kk)
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (k0 + kk < K) wb[buf][kk][s] = (Nw == 1 || lane + 64 * s < Nw) ? wp[(k0 + kk) * ldw + 64 * s] : WT(0);
    };
    fetch(0, 0);
#pragma unroll
    for (int k0 = 0; k0 < K; k0 += KB) {
      const int buf = (k0 / KB) & 1;
      if (k0 + KB < K) fetch(buf ^ 1, k0 + KB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < KB; ++kk) {
        const int k = k0 + kk;
        if (k < K) {
          const float ak = lane_value(h[l - 1][k >> 6], k & 63);
#pragma unroll
          for (int s = 0; s < S; ++s) acc[s] = fmaf(R::cvt(wb[buf][kk][s]), ak, acc[s]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int a = act_of<l>();
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int u = Nw == 1 ? 0 : lane + 64 * s;
      float v = 0.f;
      if (u < Nw) {
        const float pre = acc[s] + R::ld(th + L.boff[l] + u);
        v = R::rnd(R::RT_ACT ? act_rt(a, pre) : act_fwd(a, pre));
      }
      h[l][s] = v;
    }
  }
  template <int l = 1>
  __device__ __forceinline__ void forward(const WT *th) {
    if constexpr (l <= n) {
      fwd_layer<l>(th);
      forward<l + 1>(th);
    }
  }

  // D_{l-1} = (W_l d_l) .* act'_{l-1}(A_{l-1})
  template <int l>
  __device__ __forceinline__ void bwd_layer(const WT *th) {
    const int lane = threadIdx.x & 63;
    constexpr int K = L.w[l - 1], Nw = L.w[l], ldw = L.ldw[l], S = slots(l - 1);
    float acc[S];
#pragma unroll
    for (int s = 0; s < S; ++s) acc[s] = 0.f;
    const WT *wp = th + L.woff[l] + lane * ldw;
    WT wb[2][KB][S];
    auto fetch = [&](int buf, int j0) {
#pragma unroll
      for (int jj = 0; jj < KB; ++jj)
#pragma unroll
        for (int s = 0; s < S; ++s)
          if (j0 + jj < Nw) wb[buf][jj][s] = lane + 64 * s < K ? wp[64 * s * ldw + j0 + jj] : WT(0);
    };
    fetch(0, 0);
#pragma unroll
    for (int j0 = 0; j0 < Nw; j0 += KB) {
      const int buf = (j0 / KB) & 1;
      if (j0 + KB < Nw) fetch(buf ^ 1, j0 + KB);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int jj = 0; jj < KB; ++jj) {
        const int j = j0 + jj;
        if (j < Nw) {
          // (a one-unit layer's delta is wave-uniform already)
          const float dj = Nw == 1 ? d[l][0] : lane_value(d[l][j >> 6], j & 63);
#pragma unroll
          for (int s = 0; s < S; ++s) acc[s] = fmaf(R::cvt(wb[buf][jj][s]), dj, acc[s]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const int ap = act_of<(l > 1 ? l - 1 : 1)>();
#pragma unroll
    for (int s = 0; s < S; ++s) {
      float v = lane + 64 * s < K ? acc[s] : 0.f;
      if constexpr (l > 1) {
        if (lane + 64 * s < K) v = R::RT_ACT ? grad_rt(ap, v, h[l - 1][s]) : v * act_grad(ap, h[l - 1][s]);
      }
      // (the matrix path with run-time activations leaves D_0 unrounded: mirrored)
      d[l - 1][s] = (R::RT_ACT && l == 1) ? v : R::rnd(v);
    }
  }
  template <int from, int to>
  __device__ __forceinline__ void backward(const WT *th) {
    if constexpr (from >= to) {
      bwd_layer<from>(th);
      backward<from - 1, to>(th);
    }
  }

  // T(sign * f(x)) (wave-uniform) for the point whose component k lane k holds in x_lane (already
  // rounded to the compute type); d T / d x_k is left in d[0][0] of lane k.
  __device__ __forceinline__ float fg(const WT *th, float x_lane, int transform, float sign) {
    static_assert(L.w[0] <= 64 && L.w[n] == 1, "one point per wave: at most 64 inputs, one output unit");
    h[0][0] = x_lane;
#pragma unroll
    for (int s = 1; s < U; ++s) h[0][s] = 0.f;
    forward(th);
    const float f = h[n][0];
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
    d[n][0] = R::rnd(sign * dT * act_grad(act_of<n>(), f));
    backward<n, 1>(th);
    return Tv;
  }
};


}  // namespace bore
