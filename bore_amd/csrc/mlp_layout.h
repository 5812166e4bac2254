// mlp_layout.h -- host+device description of where one model's state lives in LDS.
//
// One workgroup owns one model (one BO loop's classifier).  All of its state that
// is touched every Adam step -- theta, m, v, the activations A_l and the deltas
// D_l of the current <=64-row tile -- sits in the CU's 160 KiB LDS for the whole
// launch; HBM sees the packed vectors once on entry and once on exit.
#pragma once
#include <stdint.h>

#include "../../include/bore_hip.h"

#define BORE_LDS_BYTES (160 * 1024)
#define BORE_THREADS 256

struct MlpLayout {
  int n_layers;
  int P;      // packed parameter count (global layout, Keras order)
  int P_lds;  // padded parameter count (LDS layout, odd row strides)
  int w[BORE_MAX_LAYERS + 1];    // widths, w[0] = input_dim
  int act[BORE_MAX_LAYERS + 1];  // act[l] = activation of layer l (1-based), act[0] unused
  int goff_w[BORE_MAX_LAYERS + 1], goff_b[BORE_MAX_LAYERS + 1];  // packed offsets
  int woff[BORE_MAX_LAYERS + 1], boff[BORE_MAX_LAYERS + 1];      // LDS offsets
  int ldw[BORE_MAX_LAYERS + 1];   // LDS row stride of W_l (odd => column walks are conflict-free)
  int lda[BORE_MAX_LAYERS + 1];   // row stride of A_l and D_l (odd)
  int aoff[BORE_MAX_LAYERS + 1];  // offset of A_l inside the tile region
  int doff[BORE_MAX_LAYERS + 1];  // offset of D_l inside the tile region
  int tb;                         // rows per tile (<= BORE_BATCH_MAX)
  int tile_floats;                // floats of A_0..A_n (+ D_* when with_deltas)
  float l2_w[BORE_MAX_LAYERS + 1], l2_b[BORE_MAX_LAYERS + 1];
  int any_l2;
};

static inline int bore_odd(int x) { return x | 1; }

// Returns 0 on success.  with_deltas: 0 = forward only, 1 = deltas D_1..D_n (fit),
// 2 = deltas D_0..D_n (input gradient).  tile_rows: rows held per tile, 1..BORE_BATCH_MAX.
static inline int bore_make_layout(const bore_mlp_desc *d, int with_deltas, int tile_rows,
                                   MlpLayout *L) {
  if (!d || d->n_layers < 1 || d->n_layers > BORE_MAX_LAYERS || d->input_dim < 1) return -1;
  if (tile_rows < 1 || tile_rows > BORE_BATCH_MAX) return -1;
  L->tb = tile_rows;
  L->n_layers = d->n_layers;
  L->w[0] = d->input_dim;
  L->act[0] = 0;
  L->any_l2 = 0;
  int g = 0, s = 0;
  for (int l = 1; l <= d->n_layers; ++l) {
    int n = d->units[l - 1];
    if (n < 1) return -1;
    if (d->act[l - 1] < BORE_ACT_LINEAR || d->act[l - 1] > BORE_ACT_TANH) return -1;
    L->w[l] = n;
    L->act[l] = d->act[l - 1];
    L->l2_w[l] = d->l2_kernel[l - 1];
    L->l2_b[l] = d->l2_bias[l - 1];
    if (L->l2_w[l] != 0.f || L->l2_b[l] != 0.f) L->any_l2 = 1;
    int k = L->w[l - 1];
    L->goff_w[l] = g; g += k * n;
    L->goff_b[l] = g; g += n;
    L->ldw[l] = bore_odd(n);
    L->woff[l] = s; s += k * L->ldw[l];
    L->boff[l] = s; s += n;
  }
  L->l2_w[0] = L->l2_b[0] = 0.f;
  L->goff_w[0] = L->goff_b[0] = L->woff[0] = L->boff[0] = L->ldw[0] = 0;
  L->P = g;
  L->P_lds = s;
  int t = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    L->lda[l] = bore_odd(L->w[l]);
    L->aoff[l] = t; t += tile_rows * L->lda[l];
  }
  for (int l = 0; l <= d->n_layers; ++l) {
    L->doff[l] = t;
    if ((with_deltas == 1 && l >= 1) || with_deltas == 2) t += tile_rows * L->lda[l];
  }
  L->tile_floats = t;
  return 0;
}
