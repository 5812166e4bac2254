// mlp_device.h -- device building blocks shared by the fit / forward / input-gradient
// kernels.  gfx950 only: 64-wide wavefronts, one 256-thread workgroup per model,
// weights + activations of a 64-row tile resident in LDS.
//
// Tile convention: a tile is up to 64 rows (BORE_BATCH_MAX, one Keras mini-batch).
// A_l is the OUTPUT of layer l (A_0 = the input rows), D_l = d objective / d
// pre-activation of layer l (D_0 = d objective / d input).  Every activation
// derivative is written in terms of the activation output, so pre-activations
// are never stored.
#pragma once
#include <hip/hip_runtime.h>

#include "mlp_layout.h"

namespace bore {

__device__ __forceinline__ float sigmoid_stable(float x) {
  float e = expf(-fabsf(x));  // expf: ocml, <=1 ulp
  float d = 1.f + e;
  return x >= 0.f ? 1.f / d : e / d;
}

__device__ __forceinline__ float act_fwd(int a, float x) {
  switch (a) {
    case BORE_ACT_RELU: return fmaxf(x, 0.f);
    case BORE_ACT_ELU: return x > 0.f ? x : expm1f(x);
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

// packed (Keras order) parameter index -> which tensor element it is.
struct ParamRef {
  int l;     // layer, 1-based
  int k;     // input index (row of W_l); -1 for a bias
  int j;     // output index
  int lds;   // index in the padded LDS copy
};

__device__ __forceinline__ ParamRef param_ref(const MlpLayout &L, int p) {
  ParamRef r;
  r.l = 1;
  for (int l = 1; l <= L.n_layers; ++l)
    if (p >= L.goff_w[l]) r.l = l;
  const int l = r.l;
  if (p < L.goff_b[l]) {
    int q = p - L.goff_w[l];
    r.k = q / L.w[l];
    r.j = q - r.k * L.w[l];
    r.lds = L.woff[l] + r.k * L.ldw[l] + r.j;
  } else {
    r.k = -1;
    r.j = p - L.goff_b[l];
    r.lds = L.boff[l] + r.j;
  }
  return r;
}

// HBM -> LDS: packed theta into the padded LDS image (coalesced reads).
__device__ __forceinline__ void load_theta(const MlpLayout &L, const float *__restrict__ g,
                                           float *th) {
  for (int p = threadIdx.x; p < L.P; p += blockDim.x) th[param_ref(L, p).lds] = g[p];
}

__device__ __forceinline__ void store_theta(const MlpLayout &L, const float *th,
                                            float *__restrict__ g) {
  for (int p = threadIdx.x; p < L.P; p += blockDim.x) g[p] = th[param_ref(L, p).lds];
}

// A_l = act_l(A_{l-1} W_l + b_l) for the nb rows of the tile.  Adjacent threads take
// adjacent output units j: W_l[k][j..] is a contiguous LDS read, A_{l-1}[b][k] a
// broadcast.  Accumulation order: bias, then k ascending, one fmaf each.
__device__ __forceinline__ void fwd_layer(const MlpLayout &L, const float *th, float *tile,
                                          int l, int nb, bool keep_logits) {
  const int K = L.w[l - 1], N = L.w[l];
  const float *Ain = tile + L.aoff[l - 1];
  float *Aout = tile + L.aoff[l];
  const int lda_in = L.lda[l - 1], lda_out = L.lda[l], ldw = L.ldw[l];
  const float *W = th + L.woff[l];
  const float *bias = th + L.boff[l];
  const int a = keep_logits ? BORE_ACT_LINEAR : L.act[l];
  for (int idx = threadIdx.x; idx < nb * N; idx += blockDim.x) {
    const int b = idx / N, j = idx - b * N;
    const float *arow = Ain + b * lda_in;
    const float *wcol = W + j;
    float acc = bias[j];
    for (int k = 0; k < K; ++k) acc = fmaf(arow[k], wcol[k * ldw], acc);
    Aout[b * lda_out + j] = act_fwd(a, acc);
  }
}

// D_{l-1} = (D_l W_l^T) .* act'_{l-1}(A_{l-1}).  Adjacent threads take adjacent k:
// rows of W_l sit ldw (odd) floats apart, so the walk is bank-conflict free.
__device__ __forceinline__ void bwd_delta(const MlpLayout &L, const float *th, float *tile,
                                          int l, int nb) {
  const int K = L.w[l - 1], N = L.w[l];
  const float *Din = tile + L.doff[l];
  float *Dout = tile + L.doff[l - 1];
  const float *Aprev = tile + L.aoff[l - 1];
  const int ld_in = L.lda[l], ld_out = L.lda[l - 1], ldw = L.ldw[l];
  const float *W = th + L.woff[l];
  const int a = L.act[l - 1];
  for (int idx = threadIdx.x; idx < nb * K; idx += blockDim.x) {
    const int b = idx / K, k = idx - b * K;
    const float *drow = Din + b * ld_in;
    const float *wrow = W + k * ldw;
    float acc = 0.f;
    for (int j = 0; j < N; ++j) acc = fmaf(drow[j], wrow[j], acc);
    if (l > 1) acc *= act_grad(a, Aprev[b * ld_out + k]);
    Dout[b * ld_out + k] = acc;
  }
}

// Objective + input gradient of one tile whose rows sit in A_0: forward, T(sign*f) into
// val_out[0..nb) (LDS or global), dT/dx into D_0.  Ends with a barrier; D_0 and val_out
// are then readable by every thread.
__device__ __forceinline__ void fg_tile(const MlpLayout &L, const float *th, float *tile, int nb,
                                        int transform, float sign, float *val_out) {
  const int n = L.n_layers;
  for (int l = 1; l <= n; ++l) {
    fwd_layer(L, th, tile, l, nb, false);
    __syncthreads();
  }
  if ((int)threadIdx.x < nb) {
    const int b = threadIdx.x;
    const float f = tile[L.aoff[n] + b * L.lda[n]];
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
    val_out[b] = T;
    tile[L.doff[n] + b * L.lda[n]] = sign * dT * act_grad(L.act[n], f);
  }
  __syncthreads();
  for (int l = n; l >= 1; --l) {
    bwd_delta(L, th, tile, l, nb);
    __syncthreads();
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

// keys: LDS scratch [N]; perm_out: [N] (LDS or global).  Ends with a barrier.
__device__ __forceinline__ void make_perm(unsigned long long base, int N, unsigned *keys,
                                          int *perm_out) {
  for (int i = threadIdx.x; i < N; i += blockDim.x) keys[i] = shuffle_key(base, i);
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const unsigned ki = keys[i];
    int r = 0;
    for (int j = 0; j < N; ++j) {
      const unsigned kj = keys[j];
      r += (kj < ki) || (kj == ki && j < i);
    }
    perm_out[r] = i;
  }
  __syncthreads();
}

}  // namespace bore
