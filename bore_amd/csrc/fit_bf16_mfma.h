// fit_bf16_mfma.h -- the mixed-precision fit on the bf16 matrix cores (v_mfma_f32_16x16x32_bf16).
// gfx950 only; wide static shapes (mlp_shapes.h ids 3 and 4: 16->64-64-64-1, 32->128-128-1).
// BASELINE config 5: "128-128-1 MLP bf16 ... fused Adam + input-grad kernel".
//
// Arithmetic (the same rounding points as oracle.fit_bf16): bfloat16 weights, biases, inputs,
// layer outputs, logits and deltas; float32 accumulation inside the MFMA, float32 loss and
// d loss / d logit; float32 master weights (in registers for the launch) + Adam slots (in HBM), updated every step.
//
// The fp32-MFMA form this replaces spent, per Adam step of 32->128-128-1, 163 k cycles: 968
// v_mfma_f32_16x16x4_f32 per wave (31 k cycles of matrix-core time alone), each fed by its own
// 2-byte LDS read, plus a per-tile HBM round trip in the update (profiles/r2/wide_stamps_*.txt).
// Here one MFMA covers 32 k-values (16x fewer instructions, each half as long), its weight
// operand is ONE 16-byte LDS read, and the activations never leave the registers:
//
//  * products are formed transposed, H_l^T = W_l^T H_{l-1}^T, as in mlp_regs.h: the result of a
//    16x16 tile leaves lane (q, m) = (lane >> 4, lane & 15) holding H_l[row m][unit 16t + 4q + r];
//  * the k-slot j = 8q + i of a 32-wide k-chunk c is assigned to unit
//        u(c, q, i) = 32c + 16 (i >> 2) + 4q + (i & 3)
//    so that lane (q, m)'s B fragment is exactly its OWN eight registers of tiles 2c, 2c + 1 --
//    no cross-lane transpose at all (the MFMA sums over k in any order we label it, as long as the
//    A operand uses the same labels);
//  * the weights are kept in LDS in FRAGMENT ORDER, two images per layer: Wf (forward: unit-out m,
//    eight unit-in k-slots per lane) and Wb (backward: unit-in m, eight unit-out k-slots), both
//    lane-linear 16-byte reads -- conflict-free;
//  * for the weight gradients (k = the 64 batch rows) the activations / deltas are stored
//    TRANSPOSED, [unit][row] -- 128-byte rows whose 16-byte chunks are XOR-swizzled by the unit
//    index so that the 16 lanes of a fragment read hit distinct banks -- and both operands of a
//    16x16 tile of dW_l are 16-byte reads too: 2 MFMAs per tile instead of 16;
//  * the last layer has ONE unit: its backward product is an outer product, formed elementwise
//    from a float32 copy of its weights (no Wb image for it);
//  * the update runs tile by tile in the MFMA's own result layout (TileOrder below): a wave's lanes hold the
//    float32 master weights of its tiles in registers for the whole launch; m / v stream from HBM as 16-byte
//    loads three tiles ahead of use.
#pragma once
#include "mlp_device.h"
#include "mlp_regs.h"
#include "mlp_shapes.h"

namespace bore {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_t;

template <int SHAPE>
struct Bf16Plan {
  static constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers;
  static constexpr bool TIGHT = false;  // (see Bf16Net: the fit leaves the operand loads to the scheduler)
  static constexpr int RS = 64;  // row pitch of the transposed A / D images, in elements (128 B)
  // element (unit, row) of a transposed image: the row's 16-byte chunk (8 rows) is XOR-ed with
  // bits 1..3 of the unit -- a fragment read (16 consecutive units, one chunk) then touches 64
  // distinct banks (units of one parity share a 128-byte bank half: 8 of them, 8 chunk positions)
  static __host__ __device__ constexpr int t_index(int unit, int row) {
    return unit * RS + ((((row >> 3) ^ ((unit >> 1) & 7)) << 3) | (row & 7));
  }
  static constexpr int T(int l) { return L.Np[l] / 16; }                // 16-unit tiles of layer l
  static constexpr int CF(int l) { return (L.w[l - 1] + 31) / 32; }      // k-chunks, forward of l
  static constexpr int CB(int l) { return (L.w[l] + 31) / 32; }          // k-chunks, backward of l
  // bf16 element offsets of the weight images
  static constexpr int wf_off(int l) {
    int o = 0;
    for (int i = 1; i < l; ++i) o += T(i) * CF(i) * 512;
    return o;
  }
  static constexpr int wf_total() { return wf_off(n + 1); }
  static constexpr int wb_off(int l) {  // layers 2..n-1 (no input gradient; layer n: wlast)
    int o = 0;
    for (int i = 2; i < l; ++i) o += T(i - 1) * CB(i) * 512;
    return o;
  }
  static constexpr int wb_total() { return wb_off(n); }
  static constexpr int bias_off(int l) {  // floats
    int o = 0;
    for (int i = 1; i < l; ++i) o += L.Np[i];
    return o;
  }
  static constexpr int wlast_off() { return bias_off(n + 1); }  // W_n[k][0] as floats, k < Np[n-1]
  static constexpr int bias_total() { return wlast_off() + L.Np[n - 1]; }
  static constexpr int at_off(int l) {  // A_l^T, l = 0..n-1
    int o = 0;
    for (int i = 0; i < l; ++i) o += L.Np[i] * RS;
    return o;
  }
  static constexpr int at_total() { return at_off(n); }
  static constexpr int dt_off(int l) {  // D_l^T, l = 1..n, behind the A images
    int o = at_total();
    for (int i = 1; i < l; ++i) o += L.Np[i] * RS;
    return o;
  }
  static constexpr int img_total() { return dt_off(n + 1); }
  // byte offsets inside the dynamic LDS
  static constexpr int o_wf = 0;
  static constexpr int o_wb = 2 * wf_total();
  static constexpr int o_bias = (o_wb + 2 * wb_total() + 15) & ~15;
  static constexpr int o_img = (o_bias + 4 * bias_total() + 15) & ~15;
  static constexpr int img_bytes = (2 * img_total() + 15) & ~15;
  static constexpr int o_end = o_img + img_bytes;
  // The packed gradient image (float32) is parked over the A / D images, one GROUP of consecutive
  // layers at a time: group g covers layers [gfirst(g), gfirst(g + 1)).
  static constexpr int layer_params(int l) { return L.w[l - 1] * L.w[l] + L.w[l]; }
  static constexpr int n_groups() {
    int g = 0, used = 0;
    for (int l = 1; l <= n; ++l) {
      if (used + layer_params(l) > img_bytes / 4) { ++g; used = 0; }
      used += layer_params(l);
    }
    return g + 1;
  }
  static constexpr int gfirst(int g) {  // first layer of group g (n + 1 past the last group)
    int gi = 0, used = 0;
    if (g == 0) return 1;
    for (int l = 1; l <= n; ++l) {
      if (used + layer_params(l) > img_bytes / 4) {
        ++gi;
        used = 0;
        if (gi == g) return l;
      }
      used += layer_params(l);
    }
    return n + 1;
  }
  // weight-gradient tiles, in layer order
  static constexpr int tiles_before(int l) {
    int t = 0;
    for (int i = 1; i < l; ++i) t += T(i - 1) * T(i);
    return t;
  }
  static constexpr int total_tiles() { return tiles_before(n + 1); }
  static constexpr int layer_of_tile(int t) {
    for (int l = 1; l <= n; ++l)
      if (t < tiles_before(l + 1)) return l;
    return n;
  }
  // (the tiles of a layer start at a multiple of 4: the four waves' i-th tiles, t = wave + 4i, then
  // always lie in ONE layer, known at compile time)
  static constexpr bool layers_aligned() {
    for (int l = 1; l <= n + 1; ++l)
      if (tiles_before(l) % 4 != 0) return false;
    return true;
  }
  static constexpr int tiles_per_wave() { return (total_tiles() + 3) / 4; }
  static constexpr bool fits() {
    for (int l = 1; l <= n; ++l)
      if (layer_params(l) > img_bytes / 4) return false;
    return true;
  }
};

// where element (k, j) of W_l sits in the fragment-order images (bf16 element index)
template <int SHAPE>
__device__ __forceinline__ int wf_index(int l_off, int CFl, int k, int j) {
  const int t = j >> 4, m = j & 15, c = k >> 5, q = (k & 15) >> 2, i = ((k >> 4) & 1) * 4 + (k & 3);
  return l_off + ((t * CFl + c) * 64 + q * 16 + m) * 8 + i;
}
template <int SHAPE>
__device__ __forceinline__ int wb_index(int l_off, int CBl, int k, int j) {
  const int t = k >> 4, m = k & 15, c = j >> 5, q = (j & 15) >> 2, i = ((j >> 4) & 1) * 4 + (j & 3);
  return l_off + ((t * CBl + c) * 64 + q * 16 + m) * 8 + i;
}

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
// round to nearest even through v_cvt_pk_bf16_f32 (the same bits as mlp_regs.h's integer form for
// every finite value)
__device__ __forceinline__ float bf16_round_hw(float x) { return (float)(__bf16)x; }
__device__ __forceinline__ unsigned pack2_bf16(float a, float b) {  // a in the low half
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
// eight floats -> one MFMA fragment (element i in bits 16 (i & 1) .. of dword i >> 1)
__device__ __forceinline__ bf16x8_t pack_frag(const float (&lo)[4], const float (&hi)[4]) {
  u32x4_t u;
  u[0] = pack2_bf16(lo[0], lo[1]);
  u[1] = pack2_bf16(lo[2], lo[3]);
  u[2] = pack2_bf16(hi[0], hi[1]);
  u[3] = pack2_bf16(hi[2], hi[3]);
  return __builtin_bit_cast(bf16x8_t, u);
}
__device__ __forceinline__ bf16x8_t lds_frag(const unsigned short *p) {
  return __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const u32x4_t *>(p));
}

// (packed parameter index -> layer / row / column: the layers' widths are powers of two or 1)
template <int SHAPE>
struct Bf16Where {
  int l, k, j;  // k < 0: bias j
};
template <int SHAPE>
__device__ __forceinline__ Bf16Where<SHAPE> bf16_where(int p) {
  using Pl = Bf16Plan<SHAPE>;
  Bf16Where<SHAPE> r{1, -1, 0};
#pragma unroll
  for (int l = 1; l <= Pl::n; ++l) {
    if (p >= Pl::L.goff_w[l] && p < Pl::L.goff_b[l]) {
      const int e = p - Pl::L.goff_w[l];
      r.l = l;
      r.k = e / Pl::L.w[l];
      r.j = e - r.k * Pl::L.w[l];
    } else if (p >= Pl::L.goff_b[l] && p < Pl::L.goff_b[l] + Pl::L.w[l]) {
      r.l = l;
      r.k = -1;
      r.j = p - Pl::L.goff_b[l];
    }
  }
  return r;
}

// ---------------------------------------------------------------------------------------------
// TILE ORDER of the per-parameter state (theta master / m / v in HBM) while a wide fit runs.
// The weight-gradient phase updates W_l in 16 x 16 tiles; the MFMA leaves lane (q, m) holding rows
// 16kb + 4q + 0..3 of column 16cb + m.  In the packed (Keras) order those four are a whole row of W_l
// apart (4-byte accesses), and even after a 4 x 4 transpose inside the lane quads (four contiguous
// columns per lane, 16-byte accesses) one tile is 16 row pieces of 64 bytes, every 128-byte line
// shared with the neighbouring tile: each access of a wave touched 16 half lines, and the CU's one
// vector-memory pipe -- 12 such instructions per tile -- bounded the phase (measured: removing
// either the loads or the stores took 4.5 us off a 18.5 us step).  For the duration of the launch
// the weights of every layer whose K and width are multiples of 16 are therefore kept in tile
// order = the MFMA's own result layout: tile (kb, cb) = 256 consecutive floats, lane's four
// registers at 4 * lane -- one instruction = 1 KiB contiguous = 8 full lines, and no transpose.
// The kernel permutes theta / m / v in place on entry and back on exit (staged through LDS, a few
// microseconds per launch); biases and a one-column last layer stay where they are.
// ---------------------------------------------------------------------------------------------
template <int SHAPE>
struct TileOrder {
  static constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers;
  static constexpr bool full(int l) { return L.w[l - 1] % 16 == 0 && L.w[l] % 16 == 0 && L.w[l] != 1; }
  // offset inside layer l's weight block of element q = k * Nw + j (packed) in tile order
  static __device__ __forceinline__ int local(int l_Nw, int q) {
    const int k = q / l_Nw, j = q - k * l_Nw;
    const int kb = k >> 4, q4 = (k >> 2) & 3, r = k & 3, cb = j >> 4, m = j & 15;
    return (kb * (l_Nw >> 4) + cb) * 256 + (16 * q4 + m) * 4 + r;
  }
  // position of packed parameter p during the launch
  static __device__ __forceinline__ int index(int p) {
    int r = p;
#pragma unroll
    for (int l = 1; l <= n; ++l)
      if (full(l) && p >= L.goff_w[l] && p < L.goff_b[l]) r = L.goff_w[l] + local(L.w[l], p - L.goff_w[l]);
    return r;
  }
  // in-place conversion of one array (all threads of the workgroup; stage = LDS, >= the largest
  // weight block; ends with a barrier).  to_tiles: packed -> tile order, else the way back.
  static __device__ __forceinline__ void convert(float *g, float *stage, bool to_tiles) {
#pragma unroll
    for (int l = 1; l <= n; ++l) {
      if (!full(l)) continue;
      const int cnt = L.w[l - 1] * L.w[l], Nw = L.w[l];
      float *blk = g + L.goff_w[l];
      for (int q = threadIdx.x; q < cnt; q += blockDim.x) {
        if (to_tiles) stage[local(Nw, q)] = blk[q];
        else stage[q] = blk[local(Nw, q)];
      }
      __syncthreads();
      for (int q = threadIdx.x; q < cnt; q += blockDim.x) blk[q] = stage[q];
      __syncthreads();
    }
  }
  static constexpr int stage_floats() {
    int m = 0;
    for (int l = 1; l <= n; ++l)
      if (full(l) && L.w[l - 1] * L.w[l] > m) m = L.w[l - 1] * L.w[l];
    return m;
  }
};

// write parameter p's new value into the LDS images (bf16 weights in both fragment orders, the
// bias as the float32 value of its bfloat16 rounding)
template <int SHAPE>
__device__ __forceinline__ void bf16_put(unsigned short *wf, unsigned short *wb, float *bias, int p,
                                         float value) {
  using Pl = Bf16Plan<SHAPE>;
  const Bf16Where<SHAPE> w = bf16_where<SHAPE>(p);
  const unsigned short h = f32_to_bf16(value);
#pragma unroll
  for (int l = 1; l <= Pl::n; ++l) {
    if (w.l != l) continue;
    if (w.k < 0) {
      bias[Pl::bias_off(l) + w.j] = bf16_to_f32(h);
    } else {
      wf[wf_index<SHAPE>(Pl::wf_off(l), Pl::CF(l), w.k, w.j)] = h;
      if (l >= 2 && l < Pl::n) wb[wb_index<SHAPE>(Pl::wb_off(l), Pl::CB(l), w.k, w.j)] = h;
      if (l == Pl::n) bias[Pl::wlast_off() + w.k] = bf16_to_f32(h);
    }
  }
}

// PLAN: where the weight images live -- Bf16Plan (the fit: no input gradient, backward images of
// layers 2..n-1) or ArgBf16Plan (arg_bf16_mfma.h: value + input gradient, backward images 1..n-1).
template <int SHAPE, typename PLAN = Bf16Plan<SHAPE>>
struct Bf16Net {
  using Pl = PLAN;
  static constexpr MlpLayout L = Pl::L;
  static constexpr int n = Pl::n;
  static constexpr int TM = RegNet<SHAPE, 1, true>::T;  // widest layer, in tiles
  float h[n + 1][TM][4];  // h[l][t][r] = A_l[row m][unit 16t + 4q + r] (bf16-exact), l >= 1
  float d[n + 1][TM][4];
  int acts[n + 1];

  __device__ __forceinline__ void set_acts(const MlpLayout &Lrt) {
#pragma unroll
    for (int l = 0; l <= n; ++l) acts[l] = Lrt.act[l];
  }

  // B fragments of a product whose k runs over the units of a layer held in C layout
  template <int NT>
  static __device__ __forceinline__ bf16x8_t frag_of(const float (&src)[TM][4], int c) {
    const float zero[4] = {0.f, 0.f, 0.f, 0.f};
    if (2 * c + 1 < NT) return pack_frag(src[2 * c], src[2 * c + 1]);
    return pack_frag(src[2 * c], zero);
  }

  template <int l>
  __device__ __forceinline__ void fwd_layer(const unsigned short *wf, const float *bias,
                                            const bf16x8_t (&xfrag)[Pl::CF(1)], bool keep_logits) {
    const int lane = threadIdx.x & 63, q = lane >> 4;
    constexpr int CFl = Pl::CF(l), Tl = Pl::T(l);
    bf16x8_t bfr[CFl];
#pragma unroll
    for (int c = 0; c < CFl; ++c) {
      if constexpr (l == 1) bfr[c] = xfrag[c];
      else bfr[c] = frag_of<Pl::T(l - 1)>(h[l - 1], c);
    }
    const unsigned short *wp = wf + Pl::wf_off(l) + lane * 8;
    const float *bp = bias + Pl::bias_off(l) + 4 * q;
    bf16x8_t wfr[2][CFl];
    float4 br[2];
#pragma unroll
    for (int c = 0; c < CFl; ++c) wfr[0][c] = lds_frag(wp + c * 512);
    br[0] = *reinterpret_cast<const float4 *>(bp);
#pragma unroll
    for (int t = 0; t < Tl; ++t) {
      if (t + 1 < Tl) {  // the next tile's operands, one tile ahead of their MFMAs
#pragma unroll
        for (int c = 0; c < CFl; ++c) wfr[(t + 1) & 1][c] = lds_frag(wp + ((t + 1) * CFl + c) * 512);
        br[(t + 1) & 1] = *reinterpret_cast<const float4 *>(bp + 16 * (t + 1));
      }
      // (TIGHT plans -- kernels that share their registers with an optimiser: the operand loads stay ONE
      // tile ahead instead of being hoisted, all tiles at once, to the top of the layer)
      if constexpr (Pl::TIGHT) __builtin_amdgcn_sched_barrier(0);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CFl; ++c)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[t & 1][c], bfr[c], acc, 0, 0, 0);
      const float b4[4] = {br[t & 1].x, br[t & 1].y, br[t & 1].z, br[t & 1].w};
#pragma unroll
      for (int r = 0; r < 4; ++r) h[l][t][r] = acc[r] + b4[r];
      if constexpr (Pl::TIGHT) __builtin_amdgcn_sched_barrier(0);
    }
    const int a = (keep_logits && l == n) ? BORE_ACT_LINEAR : acts[l];
    RegNet<SHAPE, 1, true>::template act_tiles_rt<Tl>(a, h[l]);
#pragma unroll
    for (int t = 0; t < Tl; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        h[l][t][r] = 16 * t + 4 * q + r < L.w[l] ? bf16_round_hw(h[l][t][r]) : 0.f;  // padding stays zero
  }

  // D_{l-1} = (D_l W_l^T) .* act'_{l-1}(A_{l-1}), l >= 2;  l = 1 (plans with a backward image of
  // layer 1): D_0 = D_1 W_1^T, the input gradient, left in float32
  template <int l>
  __device__ __forceinline__ void bwd_layer(const unsigned short *wb, const float *bias) {
    const int lane = threadIdx.x & 63, q = lane >> 4;
    constexpr int CBl = Pl::CB(l), Tp = Pl::T(l - 1);
    static_assert(n >= 2, "one hidden layer at least");
    if constexpr (l == n) {  // one output unit: D_{n-1}[row][k] = delta[row] * W_n[k][0] (exact in float32)
      static_assert(L.w[n] == 1, "the classifier's last layer has one unit");
      const float dl = __shfl(d[n][0][0], lane & 15, 64);  // row m's delta sits in lane m
#pragma unroll
      for (int t = 0; t < Tp; ++t) {
        const float4 w4 = *reinterpret_cast<const float4 *>(bias + Pl::wlast_off() + 16 * t + 4 * q);
        const float wv4[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) d[l - 1][t][r] = 16 * t + 4 * q + r < L.w[l - 1] ? dl * wv4[r] : 0.f;
      }
      RegNet<SHAPE, 1, true>::template grad_tiles_rt<Tp>(acts[l - 1], d[l - 1], h[l - 1]);
#pragma unroll
      for (int t = 0; t < Tp; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[l - 1][t][r] = bf16_round_hw(d[l - 1][t][r]);
      return;
    }
    bf16x8_t bfr[CBl];
#pragma unroll
    for (int c = 0; c < CBl; ++c) bfr[c] = frag_of<Pl::T(l)>(d[l], c);
    const unsigned short *wp = wb + Pl::wb_off(l) + lane * 8;
    bf16x8_t wfr[2][CBl];
#pragma unroll
    for (int c = 0; c < CBl; ++c) wfr[0][c] = lds_frag(wp + c * 512);
#pragma unroll
    for (int t = 0; t < Tp; ++t) {
      if (t + 1 < Tp) {
#pragma unroll
        for (int c = 0; c < CBl; ++c) wfr[(t + 1) & 1][c] = lds_frag(wp + ((t + 1) * CBl + c) * 512);
      }
      if constexpr (Pl::TIGHT) __builtin_amdgcn_sched_barrier(0);
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < CBl; ++c)
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wfr[t & 1][c], bfr[c], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) d[l - 1][t][r] = 16 * t + 4 * q + r < L.w[l - 1] ? acc[r] : 0.f;
      if constexpr (Pl::TIGHT) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (l >= 2) {
      RegNet<SHAPE, 1, true>::template grad_tiles_rt<Tp>(acts[l - 1], d[l - 1], h[l - 1]);
#pragma unroll
      for (int t = 0; t < Tp; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) d[l - 1][t][r] = bf16_round_hw(d[l - 1][t][r]);
    }
  }

  template <int l = 1>
  __device__ __forceinline__ void forward(const unsigned short *wf, const float *bias,
                                          const bf16x8_t (&xfrag)[Pl::CF(1)]) {
    if constexpr (l <= n) {
      fwd_layer<l>(wf, bias, xfrag, true);
      forward<l + 1>(wf, bias, xfrag);
    }
  }
  // (the acquisition side: the last layer's activation applied -- predictions, not logits)
  template <int l = 1>
  __device__ __forceinline__ void predict(const unsigned short *wf, const float *bias,
                                          const bf16x8_t (&xfrag)[Pl::CF(1)]) {
    if constexpr (l <= n) {
      fwd_layer<l>(wf, bias, xfrag, false);
      predict<l + 1>(wf, bias, xfrag);
    }
  }
  template <int l = n, int to = 2>
  __device__ __forceinline__ void backward(const unsigned short *wb, const float *bias) {
    if constexpr (l >= to) {
      bwd_layer<l>(wb, bias);
      backward<l - 1, to>(wb, bias);
    }
  }

  // C-layout registers of layer l -> the transposed image img[unit][row].  A lane holds ONE row and
  // four units; the image wants rows contiguous per unit.  The four lanes of a quad (rows 4g .. 4g+3)
  // transpose their 4 x 4 block of bfloat16 among themselves (two DPP exchanges), after which lane
  // a = lane & 3 holds unit 4q + a for the quad's four rows: one 8-byte store per tile instead of four
  // 2-byte stores that collide pairwise inside their dwords.
  template <int l>
  static __device__ __forceinline__ void store_t(const float (&src)[TM][4], unsigned short *img, int row) {
    const int lane = threadIdx.x & 63, q = lane >> 4, a = lane & 3;
    const int r0 = row & ~3;
    const unsigned sel = (a & 1) ? 0x07060302u : 0x01000504u;
#pragma unroll
    for (int t = 0; t < Pl::T(l); ++t) {
      const unsigned p01 = pack2_bf16(src[t][0], src[t][1]), p23 = pack2_bf16(src[t][2], src[t][3]);
      // lane ^ 1: quad_perm [1, 0, 3, 2]
      const unsigned x01 = __builtin_amdgcn_update_dpp(0u, p01, 0xB1, 0xF, 0xF, true);
      const unsigned x23 = __builtin_amdgcn_update_dpp(0u, p23, 0xB1, 0xF, 0xF, true);
      // even lane: units 0 / 2 of rows (m, m + 1); odd lane: units 1 / 3 of rows (m - 1, m)
      // (one v_perm with a per-lane selector; bytes 0..3 = the neighbour's word, 4..7 = this lane's)
      const unsigned A = __builtin_amdgcn_perm(p01, x01, sel);  // odd: (hi(x01), hi(p01)); even: (lo(p01), lo(x01))
      const unsigned B = __builtin_amdgcn_perm(p23, x23, sel);
      // lane ^ 2: quad_perm [2, 3, 0, 1]; lanes 0, 1 keep A (units 0, 1), lanes 2, 3 keep B (units 2, 3)
      const unsigned send = (a & 2) ? A : B;
      const unsigned recv = __builtin_amdgcn_update_dpp(0u, send, 0x4E, 0xF, 0xF, true);
      uint2 v;
      v.x = (a & 2) ? recv : A;
      v.y = (a & 2) ? B : recv;
      *reinterpret_cast<uint2 *>(img + Pl::t_index(16 * t + 4 * q + a, r0)) = v;
    }
  }
  template <int l = 1>
  __device__ __forceinline__ void store_images(unsigned short *img, int row) const {
    if constexpr (l <= n) {
      if constexpr (l < n) store_t<l>(h[l], img + Pl::at_off(l), row);
      store_t<l>(d[l], img + Pl::dt_off(l), row);
      store_images<l + 1>(img, row);
    }
  }
};

// one weight-gradient tile: which layer, which 16x16 block
struct Bf16Tile {
  int l, kb, cb, K, Nw, goff_w, goff_b, at, dt;
};
template <int SHAPE>
__device__ __forceinline__ Bf16Tile bf16_tile(int t) {
  using Pl = Bf16Plan<SHAPE>;
  Bf16Tile w{1, 0, 0, 0, 0, 0, 0, 0, 0};
  int ncb = 1, r = 0;
#pragma unroll
  for (int l = 1; l <= Pl::n; ++l) {
    if (t >= Pl::tiles_before(l) && t < Pl::tiles_before(l + 1)) {
      w.l = l; w.K = Pl::L.w[l - 1]; w.Nw = Pl::L.w[l];
      w.goff_w = Pl::L.goff_w[l]; w.goff_b = Pl::L.goff_b[l];
      w.at = Pl::at_off(l - 1); w.dt = Pl::dt_off(l);
      ncb = Pl::T(l);
      r = t - Pl::tiles_before(l);
    }
  }
  w.kb = r / ncb;
  w.cb = r - w.kb * ncb;
  return w;
}

// compile-time loop: f(std::integral_constant<int, I>) for I = B .. E - 1
template <int B, int E, typename F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (B < E) {
    f(std::integral_constant<int, B>{});
    static_for<B + 1, E>(f);
  }
}

}  // namespace bore
