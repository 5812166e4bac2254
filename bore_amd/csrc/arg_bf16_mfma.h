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
// ONE weight image in LDS serves both directions.  W_l sits row-major, [k][n], bfloat16, rows padded
// with zeros to a multiple of 32, row pitch = 16 (mod 32) elements (144 for 128 units):
//   * forward, H_l^T = W_l^T H_{l-1}^T: the A operand of output tile t (units n = 16t + m) and k-chunk c
//     wants, in lane (q, m), eight k-values of ONE column n.  ds_read_b64_tr_b16 delivers exactly that:
//     a 16-lane group reads a block of 4 rows x 16 columns and lane i receives column i of the four
//     rows (checked on the device: tools/ubench/tr_read.hip).  Group q reads rows 32c + 4q + 0..3 and
//     32c + 16 + 4q + 0..3 -- two reads per fragment -- which labels k-slot 8q + i of the chunk as unit
//     32c + 16 (i >> 2) + 4q + (i & 3): the B fragment is then the lane's OWN result registers of tiles
//     2c, 2c + 1 of the previous layer (fit_bf16_mfma.h has the argument), no cross-lane movement;
//   * backward, D_{l-1}^T = W_l D_l^T: the A operand of output tile t (k = 16t + m) wants eight
//     n-values of ONE row k: two plain 8-byte reads of row 16t + m.
// With the pitch = 16 (mod 32) elements the transposed reads of a 32-lane half touch 64 distinct banks;
// the row reads are 2-way (rows m and m + 8 share banks) -- the LDS is not what bounds an evaluation.
// Round 4's first form kept two fragment-order images (88 KB for 32->128-128-1): four restarts per CU.
// This one takes 52 KB; with the optimiser's two 2m x 2m matrices outside the LDS eight fit.
//
// Registers: a layer's outputs are kept PACKED (two bfloat16 per register -- they are bfloat16 values
// anyway, and the packed pair is what the next layer's B operand is made of): 16 registers for 128
// units instead of 32, unpacked with one shift / mask where the backward pass needs act'(h).
#pragma once
#include "fit_bf16_mfma.h"

namespace bore {

typedef short v4i16_t __attribute__((ext_vector_type(4)));

template <int SHAPE>
struct ArgBf16Plan {
  static constexpr MlpLayout L = bore_static_layout(SHAPE, 2, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers;
  static constexpr int T(int l) { return L.Np[l] / 16; }                // 16-unit tiles of layer l
  static constexpr int CF(int l) { return (L.w[l - 1] + 31) / 32; }      // k-chunks, forward of l
  static constexpr int CB(int l) { return (L.Np[l] + 31) / 32; }         // k-chunks, backward of l
  static constexpr int rows(int l) { return 32 * CF(l); }                // image rows of W_l (zero padded)
  // elements; = 16 (mod 32).  (The one-unit last layer is never a backward A operand: its 16 columns.)
  static constexpr int pitch(int l) { return l == L.n_layers ? 16 : 32 * CB(l) + 16; }
  static constexpr int w_off(int l) {  // bf16 elements
    int o = 0;
    for (int i = 1; i < l; ++i) o += rows(i) * pitch(i);
    return o;
  }
  static constexpr int w_total() { return w_off(n + 1); }
  static constexpr int bias_off(int l) {  // floats
    int o = 0;
    for (int i = 1; i < l; ++i) o += L.Np[i];
    return o;
  }
  static constexpr int wlast_off() { return bias_off(n + 1); }  // W_n[k][0] as floats, k < Np[n-1]
  static constexpr int bias_total() { return wlast_off() + L.Np[n - 1]; }
  static constexpr int max_tiles() {
    int t = 1;
    for (int l = 0; l <= n; ++l)
      if (T(l) > t) t = T(l);
    return t;
  }
  static constexpr int TM = max_tiles();
  // byte offsets from the (16-byte aligned) start of the dynamic LDS
  static constexpr int o_w = 0;
  static constexpr int o_bias = (2 * w_total() + 15) & ~15;
  static constexpr int o_end = (o_bias + 4 * bias_total() + 15) & ~15;
  static constexpr int floats = o_end / 4;  // what the kernels' LDS carves reserve for the images
  // (a backward chunk reads 32 columns of a row: the pitch covers them even where Np_l is 16)
  static_assert(n >= 2, "one hidden layer at least");
};

struct ArgBf16Images {
  const unsigned short *w;
  const float *bias;
};
template <int SHAPE>
__device__ __forceinline__ ArgBf16Images arg_bf16_images(float *smem) {
  using Pl = ArgBf16Plan<SHAPE>;
  unsigned char *b = reinterpret_cast<unsigned char *>(smem);
  return ArgBf16Images{reinterpret_cast<const unsigned short *>(b + Pl::o_w),
                       reinterpret_cast<const float *>(b + Pl::o_bias)};
}

// HBM (packed float32 theta) -> the image, by all threads of the workgroup; the region must have
// been zeroed (padding rows / columns read as zero).  No barrier inside.
template <int SHAPE>
__device__ __forceinline__ void arg_bf16_stage(const float *__restrict__ g, float *smem) {
  using Pl = ArgBf16Plan<SHAPE>;
  unsigned char *b = reinterpret_cast<unsigned char *>(smem);
  unsigned short *w = reinterpret_cast<unsigned short *>(b + Pl::o_w);
  float *bias = reinterpret_cast<float *>(b + Pl::o_bias);
  for (int p = threadIdx.x; p < Pl::L.P; p += blockDim.x) {
    const Bf16Where<SHAPE> wh = bf16_where<SHAPE>(p);
    const unsigned short h = f32_to_bf16(g[p]);
#pragma unroll
    for (int l = 1; l <= Pl::n; ++l) {
      if (wh.l != l) continue;
      if (wh.k < 0) {
        bias[Pl::bias_off(l) + wh.j] = bf16_to_f32(h);
      } else {
        w[Pl::w_off(l) + wh.k * Pl::pitch(l) + wh.j] = h;
        if (l == Pl::n) bias[Pl::wlast_off() + wh.k] = bf16_to_f32(h);
      }
    }
  }
}

template <int SHAPE>
struct ArgBf16Net {
  using Pl = ArgBf16Plan<SHAPE>;
  static constexpr MlpLayout L = Pl::L;
  static constexpr int n = Pl::n;
  static constexpr int TM = Pl::TM;
  static constexpr int CF1 = Pl::CF(1);
  static constexpr int T0 = Pl::T(0);  // 16-input tiles of the input gradient
  // hp[l][t] = this lane's four outputs of tile t of layer l (row m, units 16t + 4q + 0..3), packed:
  // [0] = units +0 (low half), +1 (high half); [1] = units +2, +3.  l = 1 .. n-1.
  unsigned hp[n][TM][2];
  float out;          // prediction of row m, in lane m (the lanes q = 0)
  float d[1][TM][4];  // d[0][t][r] = d T / d x of row m, inputs 16t + 4q + r (float32), after fg()
  int acts[n + 1];

  __device__ __forceinline__ void set_acts(const MlpLayout &Lrt) {
#pragma unroll
    for (int l = 0; l <= n; ++l) acts[l] = Lrt.act[l];
  }

  static __device__ __forceinline__ float lo16(unsigned u) { return __uint_as_float(u << 16); }
  static __device__ __forceinline__ float hi16(unsigned u) { return __uint_as_float(u & 0xffff0000u); }
  static __device__ __forceinline__ bf16x8_t frag4(unsigned a, unsigned b, unsigned c, unsigned d4) {
    u32x4_t u = {a, b, c, d4};
    return __builtin_bit_cast(bf16x8_t, u);
  }
  // B fragment of k-chunk c of a layer held packed (tiles 2c, 2c + 1; an odd tile count: zeros)
  template <int NT>
  static __device__ __forceinline__ bf16x8_t frag_of(const unsigned (&src)[TM][2], int c) {
    if (2 * c + 1 < NT) return frag4(src[2 * c][0], src[2 * c][1], src[2 * c + 1][0], src[2 * c + 1][1]);
    return frag4(src[2 * c][0], src[2 * c][1], 0u, 0u);
  }

  // Input fragments of one row from a getter x(d) -> float (0 beyond the row / the inputs): k-slot
  // 8q + i of chunk c = input 32c + 16 (i >> 2) + 4q + (i & 3), as for every other layer.
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

  // forward A operand (tile t, chunk c of layer l): two transposed reads.  EXEC must be all ones
  // (the gather crosses lanes): every caller is in wave-uniform control flow.
  template <int l>
  static __device__ __forceinline__ bf16x8_t a_fwd(const unsigned short *w, int t, int c) {
    typedef __attribute__((address_space(3))) v4i16_t *lds_v4;
    const int lane = threadIdx.x & 63, q = lane >> 4, i = lane & 15;
    const unsigned short *p = w + Pl::w_off(l) + (32 * c + 4 * q + (i >> 2)) * Pl::pitch(l) + 16 * t + 4 * (i & 3);
    const v4i16_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)p);
    const v4i16_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4)(p + 16 * Pl::pitch(l)));
    typedef short v8i16_t __attribute__((ext_vector_type(8)));
    const v8i16_t v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8_t, v);
  }
  // backward A operand (output tile t = rows k of W_l, chunk c over its columns): two row reads
  template <int l>
  static __device__ __forceinline__ bf16x8_t a_bwd(const unsigned short *w, int t, int c) {
    const int lane = threadIdx.x & 63, q = lane >> 4, m = lane & 15;
    const unsigned short *p = w + Pl::w_off(l) + (16 * t + m) * Pl::pitch(l) + 32 * c + 4 * q;
    const uint2 lo = *reinterpret_cast<const uint2 *>(p), hi = *reinterpret_cast<const uint2 *>(p + 16);
    return frag4(lo.x, lo.y, hi.x, hi.y);
  }

  // A layer's code is instantiated once per activation and ONE scalar branch picks the copy (a per-value
  // switch makes the compiler evaluate several activations and select; real calls per value cost ~50
  // cycles each, mlp_point.h).
  template <int l, int A>
  __device__ __forceinline__ void fwd_layer_a(const ArgBf16Images &im, const bf16x8_t (&xf)[CF1]) {
    const int lane = threadIdx.x & 63, q = lane >> 4;
    constexpr int CFl = Pl::CF(l), Tl = Pl::T(l);
    bf16x8_t bfr[CFl];
#pragma unroll
    for (int c = 0; c < CFl; ++c) {
      if constexpr (l == 1) bfr[c] = xf[c];
      else bfr[c] = frag_of<Pl::T(l - 1)>(hp[l - 1], c);
    }
    const float *bp = im.bias + Pl::bias_off(l) + 4 * q;
    // operands one tile ahead of their matrix instructions (and no further: the kernel shares its
    // registers with an optimiser)
    bf16x8_t afr[2][CFl];
    float4 br[2];
#pragma unroll
    for (int c = 0; c < CFl; ++c) afr[0][c] = a_fwd<l>(im.w, 0, c);
    br[0] = *reinterpret_cast<const float4 *>(bp);
#pragma unroll
    for (int t = 0; t < Tl; ++t) {
      if (t + 1 < Tl) {
#pragma unroll
        for (int c = 0; c < CFl; ++c) afr[(t + 1) & 1][c] = a_fwd<l>(im.w, t + 1, c);
        br[(t + 1) & 1] = *reinterpret_cast<const float4 *>(bp + 16 * (t + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CFl; ++c)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[t & 1][c], bfr[c], acc, 0, 0, 0);
      const float b4[4] = {br[t & 1].x, br[t & 1].y, br[t & 1].z, br[t & 1].w};
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (l < n) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 16 * t + 4 * q + r < L.w[l] ? act_fwd(A, acc[r] + b4[r]) : 0.f;
        hp[l][t][0] = pack2_bf16(v[0], v[1]);  // (round to nearest even: v_cvt_pk_bf16_f32)
        hp[l][t][1] = pack2_bf16(v[2], v[3]);
      } else if (t == 0) {
        out = bf16_round_hw(act_fwd(A, acc[0] + b4[0]));  // (unit 0: valid in the lanes q = 0, i.e. lane m)
      }
    }
  }
  template <int l>
  __device__ __forceinline__ void fwd_layer(const ArgBf16Images &im, const bf16x8_t (&xf)[CF1]) {
    switch (__builtin_amdgcn_readfirstlane(acts[l])) {
      case BORE_ACT_RELU: fwd_layer_a<l, BORE_ACT_RELU>(im, xf); break;
      case BORE_ACT_ELU: fwd_layer_a<l, BORE_ACT_ELU>(im, xf); break;
      case BORE_ACT_SIGMOID: fwd_layer_a<l, BORE_ACT_SIGMOID>(im, xf); break;
      case BORE_ACT_TANH: fwd_layer_a<l, BORE_ACT_TANH>(im, xf); break;
      default: fwd_layer_a<l, BORE_ACT_LINEAR>(im, xf); break;
    }
  }

  // D_{l-1} = (D_l W_l^T) .* act'_{l-1}(A_{l-1}); dl = D_l packed (l < n) -- the result is left packed in
  // dprev (l >= 2) or, for l = 1, as float32 in d[0] (the input gradient).  AP = activation of layer l-1.
  template <int l, int AP>
  __device__ __forceinline__ void bwd_layer_a(const ArgBf16Images &im, const unsigned (&dl)[TM][2],
                                              unsigned (&dprev)[TM][2]) {
    const int lane = threadIdx.x & 63, q = lane >> 4;
    constexpr int CBl = Pl::CB(l), Tp = Pl::T(l - 1);
    bf16x8_t bfr[CBl];
#pragma unroll
    for (int c = 0; c < CBl; ++c) bfr[c] = frag_of<Pl::T(l)>(dl, c);
    bf16x8_t afr[2][CBl];
#pragma unroll
    for (int c = 0; c < CBl; ++c) afr[0][c] = a_bwd<l>(im.w, 0, c);
#pragma unroll
    for (int t = 0; t < Tp; ++t) {
      if (t + 1 < Tp) {
#pragma unroll
        for (int c = 0; c < CBl; ++c) afr[(t + 1) & 1][c] = a_bwd<l>(im.w, t + 1, c);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CBl; ++c)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(afr[t & 1][c], bfr[c], acc, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (l == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) d[0][t][r] = 16 * t + 4 * q + r < L.w[0] ? acc[r] : 0.f;
      } else {
        const float hh[4] = {lo16(hp[l - 1][t][0]), hi16(hp[l - 1][t][0]), lo16(hp[l - 1][t][1]), hi16(hp[l - 1][t][1])};
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = 16 * t + 4 * q + r < L.w[l - 1] ? acc[r] * act_grad(AP, hh[r]) : 0.f;
        dprev[t][0] = pack2_bf16(v[0], v[1]);
        dprev[t][1] = pack2_bf16(v[2], v[3]);
      }
    }
  }
  template <int l>
  __device__ __forceinline__ void bwd_layer(const ArgBf16Images &im, const unsigned (&dl)[TM][2],
                                            unsigned (&dprev)[TM][2]) {
    if constexpr (l == 1) {
      bwd_layer_a<1, BORE_ACT_LINEAR>(im, dl, dprev);
    } else {
      switch (__builtin_amdgcn_readfirstlane(acts[l - 1])) {
        case BORE_ACT_RELU: bwd_layer_a<l, BORE_ACT_RELU>(im, dl, dprev); break;
        case BORE_ACT_ELU: bwd_layer_a<l, BORE_ACT_ELU>(im, dl, dprev); break;
        case BORE_ACT_SIGMOID: bwd_layer_a<l, BORE_ACT_SIGMOID>(im, dl, dprev); break;
        case BORE_ACT_TANH: bwd_layer_a<l, BORE_ACT_TANH>(im, dl, dprev); break;
        default: bwd_layer_a<l, BORE_ACT_LINEAR>(im, dl, dprev); break;
      }
    }
  }

  template <int l = 1>
  __device__ __forceinline__ void predict(const ArgBf16Images &im, const bf16x8_t (&xf)[CF1]) {
    if constexpr (l <= n) {
      fwd_layer<l>(im, xf);
      predict<l + 1>(im, xf);
    }
  }
  template <int l>
  __device__ __forceinline__ void backward_from(const ArgBf16Images &im, const unsigned (&dl)[TM][2]) {
    unsigned dprev[TM][2];
    bwd_layer<l>(im, dl, dprev);
    if constexpr (l > 1) backward_from<l - 1>(im, dprev);
  }

  // the one-unit last layer backwards: D_{n-1}[row][k] = delta[row] * W_n[k][0] (exact in float32)
  // times act'_{n-1}, rounded and packed
  template <int AP>
  __device__ __forceinline__ void last_layer_back(const ArgBf16Images &im, float delta, unsigned (&dlast)[TM][2]) {
    const int q = (threadIdx.x & 63) >> 4;
    constexpr int Tp = Pl::T(n - 1);
#pragma unroll
    for (int t = 0; t < Tp; ++t) {
      const float4 w4 = *reinterpret_cast<const float4 *>(im.bias + Pl::wlast_off() + 16 * t + 4 * q);
      const float wv4[4] = {w4.x, w4.y, w4.z, w4.w};
      const float hh[4] = {lo16(hp[n - 1][t][0]), hi16(hp[n - 1][t][0]), lo16(hp[n - 1][t][1]), hi16(hp[n - 1][t][1])};
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
        v[r] = 16 * t + 4 * q + r < L.w[n - 1] ? (delta * wv4[r]) * act_grad(AP, hh[r]) : 0.f;
      dlast[t][0] = pack2_bf16(v[0], v[1]);
      dlast[t][1] = pack2_bf16(v[2], v[3]);
    }
  }

  // Objective + input gradient of the wave's 16 rows: returns T(sign * f) of row m in the lanes < 16;
  // d T / d x of row m, inputs 16t + 4q + r, is left in d[0][t][r] (float32).
  __device__ __forceinline__ float fg(const ArgBf16Images &im, const bf16x8_t (&xf)[CF1], int transform,
                                      float sign) {
    predict(im, xf);
    const float f = out;
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
    // d T / d (pre-activation of the output unit) of row m, formed by lane m and handed to the four
    // lanes (q, m) of the row's column
    const float delta = __shfl(bf16_round_hw(sign * dT * act_grad(acts[n], f)), lane & 15, 64);
    unsigned dlast[TM][2];
    switch (__builtin_amdgcn_readfirstlane(acts[n - 1])) {
      case BORE_ACT_RELU: last_layer_back<BORE_ACT_RELU>(im, delta, dlast); break;
      case BORE_ACT_ELU: last_layer_back<BORE_ACT_ELU>(im, delta, dlast); break;
      case BORE_ACT_SIGMOID: last_layer_back<BORE_ACT_SIGMOID>(im, delta, dlast); break;
      case BORE_ACT_TANH: last_layer_back<BORE_ACT_TANH>(im, delta, dlast); break;
      default: last_layer_back<BORE_ACT_LINEAR>(im, delta, dlast); break;
    }
    backward_from<n - 1>(im, dlast);
    return Tv;
  }
};

}  // namespace bore
