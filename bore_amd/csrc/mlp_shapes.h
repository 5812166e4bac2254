// mlp_shapes.h -- network shapes with a compile-time layout.
//
// The generic kernels read the layout tables (widths, strides, LDS offsets) at run time,
// indexed by the layer number: every lookup is a dependent LDS read in the middle of a
// latency-bound chain.  For the shapes listed here -- the configurations BASELINE.json
// quotes -- the same kernel source is instantiated with the layout as a constant
// expression: layer loops unroll, offsets fold into ds_read immediates, the k-loops of
// the MFMA tiles have constant trip counts.  SHAPE 0 is the generic path.
#pragma once
#include "mlp_layout.h"

#define BORE_N_SHAPES 5  // ids 1..BORE_N_SHAPES: every kernel family (fit, rows, screen, restarts, ...)
// Two more compile-time layouts for the FIT kernels alone (bore_mlp_fit): 16->32-32-1 (DenseSequential with
// num_layers = 1, num_units = 32) and 16->16-16-1 with any activations (README.rst:60-63's widths on more than two
// inputs), both at the largest input dimension the first layer's 16-row tile holds.  Nets with these widths and
// fewer inputs are fitted on them zero-padded (exact: fit_padded in bore_hip.hip) -- the generic flavour takes 2.5x
// as long.  The acquisition kernels of such nets stay generic.
#define BORE_FIT_SHAPE_16_32 6
#define BORE_FIT_SHAPE_16_16 7
#define BORE_N_FIT_SHAPES 2  // ids BORE_N_SHAPES + 1 .. BORE_N_SHAPES + BORE_N_FIT_SHAPES

// Which kernel flavours a build instantiates (experiment builds only: tools/build_variant.sh fast1
// -DBORE_SHAPE_MASK=0x2 compiles the 2->16-16-1 kernels alone in a fifth of the time; a request for
// a flavour that was left out is refused with BORE_E_UNSUPPORTED).  Bit 0: the generic flavour 0,
// bits 1..5: static shapes 1..5, bits 6..9: flavours -1..-4, bits 10, 11: the fit-only shapes 6, 7.  The shipped
// library has them all.
#ifndef BORE_SHAPE_MASK
#define BORE_SHAPE_MASK 0xfff
#endif
#define BORE_ON_0 ((BORE_SHAPE_MASK) & 0x001)
#define BORE_ON_1 ((BORE_SHAPE_MASK) & 0x002)
#define BORE_ON_2 ((BORE_SHAPE_MASK) & 0x004)
#define BORE_ON_3 ((BORE_SHAPE_MASK) & 0x008)
#define BORE_ON_4 ((BORE_SHAPE_MASK) & 0x010)
#define BORE_ON_5 ((BORE_SHAPE_MASK) & 0x020)
#define BORE_ON_N1 ((BORE_SHAPE_MASK) & 0x040)
#define BORE_ON_N2 ((BORE_SHAPE_MASK) & 0x080)
#define BORE_ON_N3 ((BORE_SHAPE_MASK) & 0x100)
#define BORE_ON_N4 ((BORE_SHAPE_MASK) & 0x200)
#define BORE_ON_6 ((BORE_SHAPE_MASK) & 0x400)
#define BORE_ON_7 ((BORE_SHAPE_MASK) & 0x800)
#define BORE_FLAVOUR_LEFT_OUT "this build of the library leaves the kernel flavour out (BORE_SHAPE_MASK)"

struct ShapeSpec {
  int D, n_layers;
  int units[4];
  int act[4];  // act[0] < 0: any activations (read at run time; only the widths are static)
};

// 1: README.rst:60-63 / BASELINE config 1 and 4 (Branin-2D, 16-16-1, relu relu sigmoid)
// 2: BASELINE config 2 (Hartmann-6D, 32-32-1)
// 3: BASELINE config 3 (16-D, 64-64-64-1)
// 4: BASELINE config 5 (32-D, 128-128-1; fp32 forward / input gradient / L-BFGS-B only)
// 5: the network the reference's only in-repo caller builds -- BORE(num_layers=2, num_units=32, activation="elu",
//    transform="sigmoid") (bore/plugins/hpbandster/base.py:23-33) goes through DenseSequential's fall-through
//    (bore/models.py:16-19: num_layers + 1 hidden layers) and is D -> 32-32-32-1, elu x3 + a linear output,
//    from_logits BCE.  Static at 16 inputs; a search space of fewer dimensions runs the same kernels with the first
//    layer's unused rows zero (fit: fit_padded; acquisition: the true input dimension is a run-time argument, the
//    LDS image is the 16-input one either way).
// 6, 7: fit kernels only (BORE_FIT_SHAPE_16_32, BORE_FIT_SHAPE_16_16)
static constexpr ShapeSpec kShapes[BORE_N_SHAPES + 1 + BORE_N_FIT_SHAPES] = {
    {0, 0, {0, 0, 0, 0}, {0, 0, 0, 0}},
    {2, 3, {16, 16, 1, 0}, {BORE_ACT_RELU, BORE_ACT_RELU, BORE_ACT_SIGMOID, 0}},
    {6, 3, {32, 32, 1, 0}, {-1, 0, 0, 0}},
    {16, 4, {64, 64, 64, 1}, {-1, 0, 0, 0}},
    {32, 3, {128, 128, 1, 0}, {-1, 0, 0, 0}},
    {16, 4, {32, 32, 32, 1}, {-1, 0, 0, 0}},
    {16, 3, {32, 32, 1, 0}, {-1, 0, 0, 0}},
    {16, 3, {16, 16, 1, 0}, {-1, 0, 0, 0}},
};
// shapes with a register-row-block fit (4 does not fit fp32 theta + a 64-row tile in LDS)
static constexpr bool bore_shape_has_static_fit(int shape) {
  return (shape >= 1 && shape <= 3) || (shape >= 5 && shape <= BORE_N_SHAPES + BORE_N_FIT_SHAPES);
}
// wide shapes: the fit walks its weight-gradient tiles in a run-time loop, Adam slots in HBM
static constexpr bool bore_shape_is_wide(int shape) { return shape == 3 || shape == 4; }

static constexpr bore_mlp_desc bore_shape_desc(int shape) {
  bore_mlp_desc d{};
  d.input_dim = kShapes[shape].D;
  d.n_layers = kShapes[shape].n_layers;
  for (int i = 0; i < 4; ++i) {
    d.units[i] = kShapes[shape].units[i];
    d.act[i] = kShapes[shape].act[0] < 0 ? BORE_ACT_LINEAR : kShapes[shape].act[i];  // placeholder
  }
  return d;
}

static constexpr MlpLayout bore_static_layout(int shape, int with_deltas, int tile_rows) {
  MlpLayout L{};
  bore_mlp_desc d = bore_shape_desc(shape);
  bore_make_layout(&d, with_deltas, tile_rows, &L);
  return L;
}

template <int SHAPE, int DELTAS, int ROWS>
struct StaticLayout {
  static constexpr MlpLayout value = bore_static_layout(SHAPE, DELTAS, ROWS);
};

// Which static shape (if any) a descriptor matches: same widths and activations, no l2.
static inline int bore_match_shape(const bore_mlp_desc *d) {
  for (int s = 1; s <= BORE_N_SHAPES; ++s) {
    if (d->input_dim != kShapes[s].D || d->n_layers != kShapes[s].n_layers) continue;
    bool ok = true;
    for (int i = 0; i < d->n_layers; ++i)
      ok = ok && d->units[i] == kShapes[s].units[i] &&
           (kShapes[s].act[0] < 0 || d->act[i] == kShapes[s].act[i]) &&
           d->l2_kernel[i] == 0.f && d->l2_bias[i] == 0.f;
    if (ok) return s;
  }
  return 0;
}

// Kernel flavour for a descriptor (template argument SHAPE of the kernels): a static shape id
// when it matches one (and the launch uses the 64-row tile that layout was built for),
// -n_layers for up to 4 layers of any width, 0 otherwise.
static inline int bore_kernel_flavour(const bore_mlp_desc *d, bool full_tile) {
  const int s = full_tile ? bore_match_shape(d) : 0;
  if (s) return s;
  return d->n_layers <= 4 ? -d->n_layers : 0;
}

// The flavour of the ACQUISITION kernels (forward, value + input gradient, screening, restarts): as
// bore_kernel_flavour, and a float32 net with a narrow static shape's widths and activations on FEWER inputs takes
// that shape's kernels too -- their LDS image of the first layer is the padded one either way (a 16-row tile), the
// unused rows stay zero, and the true input dimension reaches the kernel as a run-time argument (d_in: sampling,
// row addresses, the optimiser's problem size).  Regularisers do not matter here (the fit's business).  The BASELINE
// shapes with fixed input dimensions keep exact matching where padding would buy nothing.
// (which shapes: the BASELINE shapes stay exact -- their kernels, the fused loop kernel among them, are compiled
// for their input dimension as a constant)
static constexpr bool bore_shape_takes_fewer_inputs(int shape) { return shape == 5; }
static inline int bore_acq_flavour(const bore_mlp_desc *d, bool full_tile) {
  const int f = bore_kernel_flavour(d, full_tile);
  if (f > 0 || !full_tile || d->compute != BORE_COMPUTE_F32) return f;
  for (int s : {5}) {
    if (d->input_dim > kShapes[s].D || d->n_layers != kShapes[s].n_layers) continue;
    bool ok = true;
    for (int i = 0; i < d->n_layers; ++i)
      ok = ok && d->units[i] == kShapes[s].units[i] && (kShapes[s].act[0] < 0 || d->act[i] == kShapes[s].act[i]);
    if (ok) return s;
  }
  return f;
}

// The flavour of the FIT kernels: as above, plus the fit-only shape (exact match: same widths, 16 inputs, no l2).
static inline int bore_fit_flavour(const bore_mlp_desc *d, bool full_tile) {
  const int f = bore_kernel_flavour(d, full_tile);
  if (f > 0 || !full_tile) return f;
  for (int s = BORE_N_SHAPES + 1; s <= BORE_N_SHAPES + BORE_N_FIT_SHAPES; ++s) {
    if (d->input_dim != kShapes[s].D || d->n_layers != kShapes[s].n_layers) continue;
    bool ok = true;
    for (int i = 0; i < d->n_layers; ++i)
      ok = ok && d->units[i] == kShapes[s].units[i] && d->l2_kernel[i] == 0.f && d->l2_bias[i] == 0.f;
    if (ok) return s;
  }
  return f;
}

// (experiment builds: was this flavour compiled in?)
static inline bool bore_flavour_built(int flavour) {
  const int bit = flavour > BORE_N_SHAPES ? 4 + flavour : (flavour >= 0 ? flavour : 5 - flavour);  // (6, 7 -> 10, 11)
  return ((BORE_SHAPE_MASK) >> bit) & 1;
}
