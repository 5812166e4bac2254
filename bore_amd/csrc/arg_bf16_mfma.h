// arg_bf16_mfma.h -- the acquisition side of a bfloat16 model on the bf16 matrix cores:
// prediction, objective value and input gradient of 16 rows per wave through
// v_mfma_f32_16x16x32_bf16.  gfx950 only; wide static shapes (mlp_shapes.h ids 3 and 4).
//
// Round 3 ran these products on the fp32 MFMA with operands widened from bfloat16 (exact, k-ordered:
// mlp_regs.h, BF16 = true) and, where a wave of the restart kernel evaluates ONE point, as fmaf
// chains on the vector ALU (mlp_point.h): 25 k cycles per evaluation of 32->128-128-1, half of a
// restart's time (profiles/r3/wide_restart_phases_final.txt).  One v_mfma_f32_16x16x32_bf16 covers 32
// k-values in 16 cycles: the whole evaluation is 84 matrix instructions, ~1.4 k cycles of matrix
// pipe, even with 15 of the 16 rows dead.
//
// Arithmetic (= oracle.forward_bf16 / value_and_input_grad_bf16): bfloat16 inputs, weights, biases,
// layer outputs, prediction and deltas; float32 accumulation inside the MFMA (its summation order is
// the hardware's, not k-ordered: results differ from the round-3 kernels in the last bits of a sum,
// i.e. by a flipped bfloat16 rounding now and then); the transform T, d T / d f and the input
// gradient are float32.  EVERY acquisition-side kernel of a bfloat16 model runs this same code per row
// (rows_kernel, screen_topk_kernel, lbfgsb_kernel*), and a row's result does not depend on the other
// rows of its block, so the device optimiser still equals its host build fed through
// bore_mlp_value_and_input_grad bit for bit (tests/test_gpu_argmax.py).
//
// LDS images (fit_bf16_mfma.h has the fragment order): Wf of layers 1..n, Wb of layers 1..n-1 (the
// input gradient needs layer 1's; the one-unit last layer is an outer product from float32 copies),
// biases and the last layer's weights as float32 values of their bfloat16 roundings.
#pragma once
#include "fit_bf16_mfma.h"

namespace bore {

template <int SHAPE>
struct ArgBf16Plan {
  static constexpr MlpLayout L = bore_static_layout(SHAPE, 2, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers;
  static constexpr bool TIGHT = true;  // (Bf16Net: operand loads one tile ahead, not hoisted)
  static constexpr int T(int l) { return L.Np[l] / 16; }
  static constexpr int CF(int l) { return (L.w[l - 1] + 31) / 32; }
  static constexpr int CB(int l) { return (L.w[l] + 31) / 32; }
  static constexpr int wf_off(int l) {  // bf16 elements
    int o = 0;
    for (int i = 1; i < l; ++i) o += T(i) * CF(i) * 512;
    return o;
  }
  static constexpr int wf_total() { return wf_off(n + 1); }
  static constexpr int wb_off(int l) {  // layers 1..n-1
    int o = 0;
    for (int i = 1; i < l; ++i) o += T(i - 1) * CB(i) * 512;
    return o;
  }
  static constexpr int wb_total() { return wb_off(n); }
  static constexpr int bias_off(int l) {  // floats
    int o = 0;
    for (int i = 1; i < l; ++i) o += L.Np[i];
    return o;
  }
  static constexpr int wlast_off() { return bias_off(n + 1); }
  static constexpr int bias_total() { return wlast_off() + L.Np[n - 1]; }
  // byte offsets from the (16-byte aligned) start of the dynamic LDS
  static constexpr int o_wf = 0;
  static constexpr int o_wb = 2 * wf_total();
  static constexpr int o_bias = (o_wb + 2 * wb_total() + 15) & ~15;
  static constexpr int o_end = (o_bias + 4 * bias_total() + 15) & ~15;
  static constexpr int floats = o_end / 4;  // what the kernels' LDS carves reserve for the images
};

// the LDS images of an acquisition-side kernel
struct ArgBf16Images {
  const unsigned short *wf, *wb;
  const float *bias;
};
template <int SHAPE>
__device__ __forceinline__ ArgBf16Images arg_bf16_images(float *smem) {
  using Pl = ArgBf16Plan<SHAPE>;
  unsigned char *b = reinterpret_cast<unsigned char *>(smem);
  return ArgBf16Images{reinterpret_cast<const unsigned short *>(b + Pl::o_wf),
                       reinterpret_cast<const unsigned short *>(b + Pl::o_wb),
                       reinterpret_cast<const float *>(b + Pl::o_bias)};
}

// HBM (packed float32 theta) -> the images, by all threads of the workgroup; the region must have
// been zeroed (padding k-slots and units read as zero).  No barrier inside.
template <int SHAPE>
__device__ __forceinline__ void arg_bf16_stage(const float *__restrict__ g, float *smem) {
  using Pl = ArgBf16Plan<SHAPE>;
  unsigned char *b = reinterpret_cast<unsigned char *>(smem);
  unsigned short *wf = reinterpret_cast<unsigned short *>(b + Pl::o_wf);
  unsigned short *wb = reinterpret_cast<unsigned short *>(b + Pl::o_wb);
  float *bias = reinterpret_cast<float *>(b + Pl::o_bias);
  for (int p = threadIdx.x; p < Pl::L.P; p += blockDim.x) {
    const Bf16Where<SHAPE> w = bf16_where<SHAPE>(p);
    const unsigned short h = f32_to_bf16(g[p]);
#pragma unroll
    for (int l = 1; l <= Pl::n; ++l) {
      if (w.l != l) continue;
      if (w.k < 0) {
        bias[Pl::bias_off(l) + w.j] = bf16_to_f32(h);
      } else {
        wf[wf_index<SHAPE>(Pl::wf_off(l), Pl::CF(l), w.k, w.j)] = h;
        if (l < Pl::n) wb[wb_index<SHAPE>(Pl::wb_off(l), Pl::CB(l), w.k, w.j)] = h;
        if (l == Pl::n) bias[Pl::wlast_off() + w.k] = bf16_to_f32(h);
      }
    }
  }
}

template <int SHAPE>
struct ArgBf16Net : Bf16Net<SHAPE, ArgBf16Plan<SHAPE>> {
  using Base = Bf16Net<SHAPE, ArgBf16Plan<SHAPE>>;
  using Pl = ArgBf16Plan<SHAPE>;
  static constexpr int n = Pl::n;
  static constexpr int CF1 = Pl::CF(1);
  static constexpr int T0 = Pl::T(0);  // 16-input tiles of the input gradient

  // Input fragments of one row from a getter x(d) -> float (d < D guaranteed by the caller's padding
  // rule: the getter returns 0 beyond the row / the inputs).  k-slot 8q + i of chunk c = input
  // 32c + 16 (i >> 2) + 4q + (i & 3), as for every other layer.
  template <typename F>
  static __device__ __forceinline__ void make_xfrag(bf16x8_t (&xf)[CF1], F &&x) {
    const int q = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int c = 0; c < CF1; ++c) {
      float lo[4], hi[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        lo[i] = x(32 * c + 4 * q + i);
        hi[i] = x(32 * c + 16 + 4 * q + i);
      }
      xf[c] = pack_frag(lo, hi);
    }
  }

  // Objective + input gradient of the wave's 16 rows: returns T(sign * f) of row m in the lanes < 16;
  // d T / d x of row m, inputs 16t + 4q + r, is left in d[0][t][r] (float32).
  __device__ __forceinline__ float fg(const ArgBf16Images &im, const bf16x8_t (&xf)[CF1], int transform,
                                      float sign) {
    this->predict(im.wf, im.bias, xf);
    const float f = this->h[n][0][0];
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
    const int lane = threadIdx.x & 63;
    // d T / d (pre-activation of the output unit) of row m, in lane m (bwd_layer<n> reads it from there)
    this->d[n][0][0] = lane < 16 ? bf16_round_hw(sign * dT * act_grad(this->acts[n], f)) : 0.f;
    this->template backward<n, 1>(im.wb, im.bias);
    return Tv;
  }
};

}  // namespace bore
