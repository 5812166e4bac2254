// mlp_layout.h -- host+device description of where one model's state lives in LDS.
//
// One workgroup owns one model (one BO loop's classifier).  All of its state that is
// touched every Adam step -- theta, m, v, the activations A_l and the deltas D_l of the
// current <=64-row tile -- sits in the CU's 160 KiB LDS for the whole launch; HBM sees the
// packed vectors once on entry and once on exit.
//
// Every matrix product of the path (forward, backward, weight gradient) is tiled into
// 16x16 outputs computed by v_mfma_f32_16x16x4_f32 (exact fp32: a k-ordered fmaf chain),
// so the LDS images are padded to the MFMA tile grid:
//   A_l, D_l : [rows][lda_l],  lda_l = round_up(w_l, 16) + 2, columns >= w_l are ZERO
//   W_l      : [round_up(w_{l-1}, 16)][ldw_l], ldw_l = round_up(w_l, 16) + 2, padding ZERO
//   b_l      : [round_up(w_l, 16)]
// Row strides of the form 16k+2 = 2*odd make the 16-row x 2-column operand fetch of one
// MFMA (lane -> row l&15, column l>>4) hit 32 distinct LDS banks.
#pragma once
#include <stdint.h>

#include "../../include/bore_hip.h"

#define BORE_LDS_BYTES (160 * 1024)
#define BORE_THREADS 256

struct MlpLayout {
  int n_layers;
  int P;      // packed parameter count (global layout, Keras order)
  int P_lds;  // floats of the padded LDS image of theta (same for m and v)
  int w[BORE_MAX_LAYERS + 1];    // widths, w[0] = input_dim
  int act[BORE_MAX_LAYERS + 1];  // act[l] = activation of layer l (1-based), act[0] unused
  int Np[BORE_MAX_LAYERS + 1];   // round_up(w[l], 16)
  int goff_w[BORE_MAX_LAYERS + 1], goff_b[BORE_MAX_LAYERS + 1];  // packed offsets
  int woff[BORE_MAX_LAYERS + 1], boff[BORE_MAX_LAYERS + 1];      // LDS offsets
  int ldw[BORE_MAX_LAYERS + 1];   // LDS row stride of W_l
  int lda[BORE_MAX_LAYERS + 1];   // row stride of A_l and D_l
  int aoff[BORE_MAX_LAYERS + 1];  // offset of A_l inside the tile region
  int doff[BORE_MAX_LAYERS + 1];  // offset of D_l inside the tile region
  int tb;                         // rows per tile (<= BORE_BATCH_MAX)
  int tbp;                        // rows allocated per tile buffer: round_up(tb, 16)
  int tile_floats;                // floats of A_0..A_n (+ D_* when with_deltas)
  float l2_w[BORE_MAX_LAYERS + 1], l2_b[BORE_MAX_LAYERS + 1];
  int any_l2;
};

static constexpr int bore_round_up(int x, int m) { return (x + m - 1) / m * m; }

// Returns 0 on success.  with_deltas: 0 = forward only, 1 = deltas D_1..D_n (fit),
// 2 = deltas D_0..D_n (input gradient).  tile_rows: rows held per tile, 1..BORE_BATCH_MAX.
// constexpr: the kernels specialised for a fixed network shape (mlp_shapes.h) evaluate it at
// compile time, so every offset below becomes an immediate.
static constexpr int bore_make_layout(const bore_mlp_desc *d, int with_deltas, int tile_rows,
                                      MlpLayout *L) {
  if (!d || d->n_layers < 1 || d->n_layers > BORE_MAX_LAYERS || d->input_dim < 1) return -1;
  if (d->compute != BORE_COMPUTE_F32 && d->compute != BORE_COMPUTE_BF16) return -1;
  if (tile_rows < 1 || tile_rows > BORE_BATCH_MAX) return -1;
  L->tb = tile_rows;
  L->tbp = bore_round_up(tile_rows, 16);
  L->n_layers = d->n_layers;
  L->w[0] = d->input_dim;
  L->Np[0] = bore_round_up(d->input_dim, 16);
  L->act[0] = 0;
  L->any_l2 = 0;
  int g = 0, s = 0;
  for (int l = 1; l <= d->n_layers; ++l) {
    int n = d->units[l - 1];
    if (n < 1) return -1;
    if (d->act[l - 1] < BORE_ACT_LINEAR || d->act[l - 1] > BORE_ACT_TANH) return -1;
    L->w[l] = n;
    L->Np[l] = bore_round_up(n, 16);
    L->act[l] = d->act[l - 1];
    L->l2_w[l] = d->l2_kernel[l - 1];
    L->l2_b[l] = d->l2_bias[l - 1];
    if (L->l2_w[l] != 0.f || L->l2_b[l] != 0.f) L->any_l2 = 1;
    int k = L->w[l - 1];
    L->goff_w[l] = g; g += k * n;
    L->goff_b[l] = g; g += n;
    L->ldw[l] = L->Np[l] + 2;
    L->woff[l] = s; s += L->Np[l - 1] * L->ldw[l];
    L->boff[l] = s; s += L->Np[l];
  }
  L->l2_w[0] = L->l2_b[0] = 0.f;
  L->goff_w[0] = L->goff_b[0] = L->woff[0] = L->boff[0] = L->ldw[0] = 0;
  L->P = g;
  L->P_lds = s;
  int t = 0;
  for (int l = 0; l <= d->n_layers; ++l) {
    L->lda[l] = L->Np[l] + 2;
    L->aoff[l] = t; t += L->tbp * L->lda[l];
  }
  for (int l = 0; l <= d->n_layers; ++l) {
    L->doff[l] = t;
    if ((with_deltas == 1 && l >= 1) || with_deltas == 2) t += L->tbp * L->lda[l];
  }
  L->tile_floats = t;
  return 0;
}
