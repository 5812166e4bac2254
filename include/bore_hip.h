/*
 * bore_hip.h -- C-ABI of libbore_hip.so, the MI355X (gfx950) implementation of the
 * BORE density-ratio classifier hot path: MLP fit (BCE + Adam) and the batched
 * value / input-gradient operator used by the multi-start argmax.
 *
 * The reference (ltiao/bore v1.5.0) is pure Python over TensorFlow/Keras and SciPy;
 * it has no FFI of its own.  Each entry point below replaces the TF op sequence a
 * reference call site triggers; the Python host (bore_amd/) binds them with ctypes
 * (INTEGRATION.md shows the stub a reference maintainer would add).
 *
 * Conventions
 *   - plain C, no C++/torch types; every pointer marked "device" is a HIP device
 *     pointer owned by the caller (torch.Tensor.data_ptr()); `stream` is a
 *     hipStream_t passed as void* (NULL = default stream).  Calls only ENQUEUE
 *     work; the caller synchronises the stream.
 *   - return 0 on success, <0 on error (BORE_E_*); bore_last_error() returns the
 *     message for the calling thread.  Nothing throws across the boundary.
 *   - row-major; network arithmetic is fp32; coordinates crossing the SciPy
 *     boundary are fp64 (bore/decorators.py:54-61).
 *   - parameters are ONE packed fp32 vector per model in Keras get_weights()
 *     order [W1 (in,out), b1, W2, b2, ...]; `n_models` models are stored back to
 *     back (leading replica dimension: independent BO loops).
 */
#ifndef BORE_HIP_H
#define BORE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BORE_ABI_VERSION 12
#define BORE_MAX_LAYERS 8
#define BORE_BATCH_MAX 64 /* rows per tile == wavefront width */
#define BORE_DIM_MAX 64   /* largest input dimension for the by-value bound arrays */

enum bore_status {
  BORE_OK = 0,
  BORE_E_INVALID = -1,   /* bad argument */
  BORE_E_UNSUPPORTED = -2, /* shape does not fit this build's kernels (e.g. LDS budget) */
  BORE_E_HIP = -3,       /* HIP runtime error (message has hipGetErrorString) */
  BORE_E_CALLBACK = -4,  /* a host callback asked to stop (bore_engine_run) */
  /* bore_mlp_fit without `perm`: the network fits, but N rows are too many for the epoch's shuffle to be
   * drawn in LDS; the same call WITH explicit permutations (any N) works.  Callers test this code,
   * not the message (a kind of BORE_E_UNSUPPORTED: bore_amd raises a subclass of its error). */
  BORE_E_NEEDS_PERM = -5
};

/* Keras activation names accepted by Dense(activation=...) on this path
 * (README.rst:61-63 relu/sigmoid; plugins/hpbandster/base.py:27 elu). */
enum bore_activation {
  BORE_ACT_LINEAR = 0,
  BORE_ACT_RELU = 1,
  BORE_ACT_ELU = 2,
  BORE_ACT_SIGMOID = 3,
  BORE_ACT_TANH = 4
};

/* TRANSFORMS of bore/plugins/hpbandster/base.py:18; the objective handed to
 * L-BFGS-B is transform(-f(x)) (bore/mixins.py:20). */
enum bore_transform {
  BORE_T_IDENTITY = 0,
  BORE_T_SIGMOID = 1,
  BORE_T_EXP = 2
};

/* Arithmetic of the network (what a Keras dtype policy selects).  BORE_COMPUTE_BF16 =
 * "mixed_bfloat16" (BASELINE.json config 5): weights, layer outputs, logits and deltas rounded to
 * bfloat16, sums in float32, float32 master weights (`theta`) and Adam slots; the objective value
 * and input gradient of the argmax are evaluated with the same rounding.  Available for the wide
 * static shapes (16->64-64-64-1 and 32->128-128-1, any activations, no l2), fit with
 * batch_size 64; every entry point returns BORE_E_UNSUPPORTED otherwise (bore_mlp_evaluate
 * always reads the float32 master weights). */
enum bore_compute {
  BORE_COMPUTE_F32 = 0,
  BORE_COMPUTE_BF16 = 1
};

/* A stack of Dense layers (bore/models.py:9-33; README.rst:60-63). */
typedef struct bore_mlp_desc {
  int32_t input_dim;                 /* D */
  int32_t n_layers;                  /* number of Dense layers, 1..BORE_MAX_LAYERS */
  int32_t units[BORE_MAX_LAYERS];    /* output width of each layer */
  int32_t act[BORE_MAX_LAYERS];      /* enum bore_activation */
  float l2_kernel[BORE_MAX_LAYERS];  /* kernel_regularizer=l2(f): f (0 = none) */
  float l2_bias[BORE_MAX_LAYERS];    /* bias_regularizer=l2(f) */
  int32_t compute;                   /* enum bore_compute */
} bore_mlp_desc;

/* tf.keras.optimizers.Adam hyper-parameters (README.rst:66 optimizer="adam":
 * lr 1e-3, beta 0.9/0.999, epsilon 1e-7). */
typedef struct bore_adam_cfg {
  float lr, beta1, beta2, eps;
} bore_adam_cfg;

int bore_abi_version(void);
/* ABI 11: sha256 (hex) of the kernel sources this library was compiled from -- the .hip and .h files of bore_amd/csrc and this
 * header, names and contents in sorted order (bore_amd._lib.source_digest) -- or "unknown" for a build that was not
 * given one.  bore_amd._lib.build_native() rebuilds when it differs from the tree's. */
const char *bore_source_digest(void);
const char *bore_last_error(void);

/* Number of fp32 parameters P of the packed vector; <0 on a bad descriptor. */
int64_t bore_param_count(const bore_mlp_desc *desc);

/*
 * Keras predict / model(x)  (bore/mixins.py:50 screening; bore/base.py:40).
 *   theta  device [n_models][P]
 *   X      device fp32 [n_models][n_rows][D], or [n_rows][D] shared by every
 *          model when x_shared != 0
 *   out    device fp32 [n_models][n_rows]  (units[last] must be 1)
 */
int bore_mlp_forward(const bore_mlp_desc *desc, int n_models, const float *theta,
                     const float *X, int64_t n_rows, int x_shared, float *out,
                     void *stream);

/*
 * convert(model, transform) evaluated on a batch of points (bore/base.py:7-42,
 * bore/decorators.py:24-79).  negate != 0 gives the minimisation form
 * transform(-f(x)) that MaximizableMixin hands to SciPy (bore/mixins.py:20);
 * negate == 0 gives transform(f(x)) (BatchMaximizableMixin._func_max,
 * bore/mixins.py:98).
 *   X     device fp64 [n_models][n_rows][D]
 *   val   device fp32 [n_models][n_rows]
 *   grad  device fp64 [n_models][n_rows][D]     d val / d x (dtype of the watched input)
 */
int bore_mlp_value_and_input_grad(const bore_mlp_desc *desc, int n_models,
                                  const float *theta, const double *X,
                                  int64_t n_rows, int transform, int negate,
                                  float *val, double *grad, void *stream);

/*
 * Keras fit(X, z, epochs, batch_size) for n_models independent models in ONE
 * launch (README.rst:93; bore/plugins/hpbandster/base.py:184): per epoch one
 * shuffle, ceil(N/batch) steps of {forward, mean BCE-from-logits, backward,
 * Adam}; the last partial batch still steps (bore/math.py:12-13).
 *   theta, adam_m, adam_v  device fp32 [n_models][P], updated in place
 *   adam_t   device int64 [n_models], Adam iteration counters, advanced by
 *            epochs*ceil(N/batch) (state persists across calls: warm start)
 *   X        device fp32 [n_models][N][D];  z device fp32 [n_models][N] in {0,1}
 *   perm     device int32 [n_models][epochs][N] explicit shuffles, or NULL to
 *            draw them in-kernel from (seed, model_index0+model, epoch0+epoch)
 *            -- the same stream bore_shuffle_perm() writes out.  In-kernel shuffles are
 *            drawn and ranked in LDS: N up to what a workgroup's LDS holds beside the
 *            network (BORE_E_UNSUPPORTED beyond, the message gives the bound); an
 *            explicit perm may have any length N <= 2^24 (read from memory step by step
 *            when it does not fit in LDS)
 *   epoch_loss  device fp32 [n_models][epochs] or NULL: Keras' logged loss
 *   batch_size  >= 1 (more than BORE_BATCH_MAX rows: 64-row sub-tiles of one Adam step)
 * Kernel flavours: the networks BASELINE.json quotes have kernels with a compile-time layout (2->16-16-1,
 * 6->32-32-1, 16->64-64-64-1; bfloat16: the last and 32->128-128-1), and so has 16->32-32-32-1 -- what the
 * reference's HpBandSter plugin builds by default (num_layers=2, num_units=32 go through DenseSequential's
 * num_layers + 1 hidden layers, bore/models.py:16-19; fit AND acquisition kernels).  The fit also for 16->32-32-1
 * (num_layers=1) and 16->16-16-1.  A float32 network with the widths and activations of one of these (2->16-16-1
 * and the three 16-input layouts) but FEWER inputs is fitted on those kernels zero-padded (exact: the same bits as
 * the generic flavour, 2.5x sooner; the repack lives in a stream-ordered scratch, hipMallocAsync); anything else on
 * the generic flavour (any widths, up to 8 layers).
 */
int bore_mlp_fit(const bore_mlp_desc *desc, int n_models, float *theta,
                 float *adam_m, float *adam_v, int64_t *adam_t, const float *X,
                 const float *z, int64_t N, int epochs, int batch_size,
                 const int32_t *perm, uint64_t seed, int64_t model_index0,
                 int64_t epoch0, const bore_adam_cfg *adam, float *epoch_loss,
                 void *stream);

/*
 * Keras evaluate(X, z): mean BCE(+l2) and binary accuracy (threshold 0.5 on the
 * model output) over all N rows (bore/plugins/hpbandster/base.py:186).
 *   loss, acc  device fp32 [n_models]
 */
int bore_mlp_evaluate(const bore_mlp_desc *desc, int n_models, const float *theta,
                      const float *X, const float *z, int64_t N, float *loss,
                      float *acc, void *stream);

/*
 * Label step of the BO loop: tau = quantile(y, gamma) with numpy's default linear
 * interpolation, z = (y < tau) (bore/data.py:31-35; README.rst:89-90).  fp64, bit-exact
 * with numpy for finite y.
 *   y    device fp64 [n_models][N];  z device fp32 [n_models][N] in {0,1};  tau device fp64 [n_models]
 */
int bore_labels(int n_models, const double *y, int64_t N, double gamma, float *z, double *tau,
                void *stream);

/*
 * Candidate sampling of maxima(): X ~ U(low, high), shape [n_models][n_samples][D]
 * (bore/mixins.py:49).  The reference draws from numpy's RandomState on the host; this
 * entry point is the device-resident alternative for replica runs: a counter-based
 * stream keyed by (seed, model_index0 + model, draw_index) -- bore_amd/sampling.py holds
 * the identical numpy statement.  low/high are HOST arrays of length D (D <= 64).
 */
int bore_uniform_candidates(uint64_t seed, int64_t model_index0, int n_models, int64_t draw_index,
                            int64_t n_samples, int D, const double *low, const double *high,
                            double *X, void *stream);

/*
 * Screening of maxima(): predict on the candidates and keep the num_starts best
 * (bore/mixins.py:50-56: z_init = predict(X_init); argpartition(-z_init, num_starts-1)).
 * Selection is by descending prediction, ties to the lower index (numpy's argpartition
 * returns the same SET in an unspecified order).
 *   X_init  device fp64 [n_models][n_samples][D] (or [n_samples][D] when x_shared != 0)
 *   x0      device fp64 [n_models][num_starts][D]   the chosen rows
 *   idx     device int32 [n_models][num_starts]     their row numbers
 *   pred    device fp32 [n_models][n_samples] or NULL
 */
int bore_screen_topk(const bore_mlp_desc *desc, int n_models, const float *theta,
                     const double *X_init, int64_t n_samples, int x_shared, int num_starts,
                     double *x0, int32_t *idx, float *pred, void *stream);

/*
 * bore_uniform_candidates + bore_screen_topk in one launch, without the candidates ever being
 * written to memory: row i of model m is recomputed from the counter stream (seed,
 * model_index0 + m, draw_index) where the forward pass and the final gather need it.  Same
 * picks, bit for bit, as the two calls (tested); saves the [n_models][n_samples][D] fp64
 * round trip through HBM (16 KB per model at 1024 x 2).
 */
int bore_sample_screen_topk(const bore_mlp_desc *desc, int n_models, const float *theta,
                            uint64_t seed, int64_t model_index0, int64_t draw_index,
                            int64_t n_samples, const double *low, const double *high,
                            int num_starts, double *x0, int32_t *idx, float *pred, void *stream);

/* scipy.optimize.minimize(method="L-BFGS-B") options as the reference passes them
 * (bore/mixins.py:23: maxiter=1000, ftol=1e-9; SciPy defaults for the rest). */
typedef struct bore_lbfgsb_opts {
  int32_t maxcor;   /* 10 */
  int32_t maxiter;  /* 15000 (reference: 1000) */
  int32_t maxfun;   /* 15000 */
  int32_t maxls;    /* 20 */
  double ftol;      /* 2.22e-9 (reference: 1e-9) */
  double gtol;      /* 1e-5 */
} bore_lbfgsb_opts;

/*
 * The restart loop of maxima() (bore/mixins.py:57-66): num_starts independent
 * bound-constrained L-BFGS-B minimisations of transform(+-f(x)) per model, all inside ONE
 * launch -- the optimiser state machines run on the device (bore_amd/csrc/lbfgsb.h), the
 * f/g requests of a workgroup's problems are evaluated together against weights held in
 * LDS.  Same algorithm, constants, stopping rules and status codes as SciPy's L-BFGS-B.
 *   x0       device fp64 [n_models][num_starts][D]  (clipped into the box like SciPy does)
 *   lb, ub   HOST fp64 [D]; +-INFINITY for an open side (D <= 64)
 *   x, jac   device fp64 [n_models][num_starts][D];  fun device fp64 [n_models][num_starts]
 *            (the fp32 objective value, widened)
 *   info     device int32 [n_models][num_starts][5] = {nit, nfev, status, task, message}
 *            status 0 converged / 1 maxiter-or-maxfun / 2 abnormal (OptimizeResult.status);
 *            (task, message) index scipy's status_messages / task_messages
 * Launch geometry (round 4; the results do not depend on it): one restart per wave at a time; with many
 * restarts per model a model's restarts go to a few workgroups whose waves draw problem after problem
 * from a queue in LDS (the weights are staged once per workgroup).  For 32 -> 128-128-1 in bfloat16 the
 * optimiser's two 2m x 2m matrices live in a device buffer the library allocates on first use and keeps
 * (1 024 slots x 8 waves x 8 m^2 doubles: 52 MB at maxcor 10), so that eight restarts share a CU's LDS.
 */
int bore_lbfgsb_minimize(const bore_mlp_desc *desc, int n_models, const float *theta,
                         int transform, int negate, const double *x0, int num_starts,
                         const double *lb, const double *ub, const bore_lbfgsb_opts *opts,
                         double *x, double *fun, double *jac, int32_t *info, void *stream);

/*
 * Record.append + Record.load_regression_data for replica runs (bore/data.py:14-29): the
 * device keeps every loop's observations in fp64, X_seen [n_models][cap][D] and y_seen
 * [n_models][cap] (the first n_seen rows valid).  This call appends row n_seen = (x_new, y_new)
 * of every model (skipped when x_new == NULL) and writes the DENSE training views the label
 * step and the fit consume: X32 fp32 [n_models][n][D] (Keras casts the float64 features to
 * float32) and y_dense fp64 [n_models][n], n = n_seen + (x_new != NULL).
 *   x_new device fp64 [n_models][D], y_new device fp64 [n_models]   (or both NULL)
 */
int bore_append_observations(int n_models, int D, double *X_seen, double *y_seen, int64_t n_seen,
                             int64_t cap, const double *x_new, const double *y_new, float *X32,
                             double *y_dense, void *stream);

/*
 * The selection of MaximizableMixin.argmax (bore/mixins.py:74-89) over the results
 * bore_lbfgsb_minimize wrote, with the duplicate filter the plugin passes as filter_fn
 * (bore/plugins/hpbandster/base.py:210-214 -> Record.is_duplicate, bore/data.py:43-48):
 * per model, among the restarts with status 0 or 1 (`res.success or res.status == 1`) that
 * are not np.allclose(x_prev, x, rtol, atol) to any stored x_prev -- i.e. not
 * all_d |x_prev[d] - x[d]| <= atol + rtol*|x[d]| -- the one with the smallest fun, the
 * earliest on ties (the reference compares with a strict <).  fun is assumed not NaN.
 *   x, fun, info  device, as bore_lbfgsb_minimize wrote them
 *   X_seen        device fp64 [n_models][cap][D], first n_seen rows valid; NULL = no filter
 *   x_best        device fp64 [n_models][D]   (left untouched for a model without a result)
 *   best          device int32 [n_models]     restart index, or -1 where the reference returns None
 */
int bore_select_best(int n_models, int num_starts, int D, const double *x, const double *fun,
                     const int32_t *info, const double *X_seen, int64_t n_seen, int64_t cap,
                     double rtol, double atol, double *x_best, int32_t *best, void *stream);

/* SVGD hyper-parameters as BatchMaximizableMixin.argmax_batch passes them (bore/mixins.py:100-116;
 * defaults n_iter 1000, step_size 1e-3, alpha 0.9, eps 1e-6, tau 1.0, length_scale None). */
typedef struct bore_svgd_opts {
  int32_t n_iter;
  int32_t distortion;       /* 0: omega = distortion_param (DistortionConstant c);
                               1: omega = rank^-distortion_param (DistortionExpDecay lambd) */
  double step_size, alpha, eps, tau;
  double length_scale;      /* RadialBasis length scale; < 0: the median heuristic (None) */
  double distortion_param;
} bore_svgd_opts;

/*
 * SVGD.optimize_from_init (bore/optimizers/svgd/base.py:79-119) with the RadialBasis kernel
 * (bore/optimizers/svgd/kernels.py:4-28) on the objective transform(f(x)) -- every iteration of
 * every particle inside ONE launch per model (particles, kernel matrix and Adagrad history in LDS).
 *   x_init, x_out  device fp64 [n_models][n_particles][D]   (up to BORE_BATCH_MAX particles with the kernel matrix
 *                  in LDS, more -- as many as fit: 32 n_particles D bytes of state -- with its entries formed on the fly)
 *   desc           float32 networks of any shape; compute = bfloat16 for the wide static shapes (as everywhere)
 *   lb, ub         HOST fp64 [D]: particles are clipped into the box after every update;
 *                  both NULL = no clipping
 * BORE_E_UNSUPPORTED when particles x dimension do not fit the LDS beside the network.
 */
int bore_svgd_optimize(const bore_mlp_desc *desc, int n_models, const float *theta, int transform,
                       const double *x_init, int n_particles, const double *lb, const double *ub,
                       const bore_svgd_opts *opts, double *x_out, void *stream);

/* The in-kernel shuffle stream of bore_mlp_fit, written out:
 * perm device int32 [n_models][epochs][N]. */
int bore_shuffle_perm(uint64_t seed, int64_t model_index0, int n_models,
                      int64_t epoch0, int epochs, int64_t N, int32_t *perm,
                      void *stream);

/*
 * Batch mode (plumbing of the asynchronous replica engine; off by default).  While a batch is set
 * for the calling thread, the leading dimension of bore_append_observations, bore_labels,
 * bore_mlp_fit, bore_sample_screen_topk and bore_lbfgsb_minimize is a BATCH of arbitrary loops at
 * different iteration counts instead of consecutive models of one size: slot b works on loop
 * ids[b] with N = n_init + its[b] observations.
 *   - per-LOOP arrays are indexed by ids[b]: theta/adam_m/adam_v [n_loops][P], adam_t, and the
 *     record buffers, all with `cap` rows per loop: X_seen fp64 [n_loops][cap][D], y_seen
 *     [n_loops][cap], the float32 training view X (fit) [n_loops][cap][D], z [n_loops][cap];
 *   - per-SLOT arrays are indexed by b: x_new, y_new, x0, idx, x, fun, jac, info;
 *   - the N / n_seen / epoch0 / draw_index arguments give the LARGEST value in the batch (they
 *     size the LDS); each slot uses its own: epoch0 = its[b] * epochs, draw_index = its[b];
 *     stream keys use model_index0 + ids[b];
 *   - bore_append_observations writes row N - 1 of loop ids[b] (its[b] > 0) into X_seen, y_seen
 *     and X32 (all `cap`-strided; y_dense unused);
 *   - bore_lbfgsb_minimize (num_starts <= 4) ends each loop with the pick of bore_select_best
 *     (filter over X_seen when deduplicate != 0) and publishes it as soon as THAT loop's restarts
 *     are done: result[ids[b]] = {x_best[D], best, sum nfev, max nfev, then the device-clock
 *     ticks (hipDeviceAttributeWallClockRate) the loop-iteration spent in labels, fit, sample +
 *     screen and restarts + pick -- zeros unless `stamps` is given and the launch is the fused
 *     iteration kernel --, the restarts' evaluations that ran the network} (D + 8 doubles), then, after a system-scope fence,
 *     flag[ids[b]] = its[b] + 1.  result and flag must be host-visible (pinned) memory.
 * bore_set_batch(NULL) returns to the plain meaning.  The struct is copied.
 */
typedef struct bore_batch {
  const int32_t *ids;    /* device [n_models] */
  const int32_t *its;    /* device [n_models] */
  int32_t n_init;
  int32_t deduplicate;
  int64_t cap;
  const double *X_seen;  /* device fp64 [n_loops][cap][D] */
  double *result;        /* pinned [n_loops][D + 8] */
  int32_t *flag;         /* pinned [n_loops] */
  int64_t *stamps;       /* device [n_loops][4] scratch for the phase clock stamps, or NULL */
  /* Residency (fused iteration kernel only; all five pointers or none).  A workgroup runs
   * iterations its[b] .. targets[b] - 1 of its loop in ONE launch as long as the host keeps up:
   * after publishing iteration k's suggestion it waits up to wait_ticks of the device clock for
   * yseq[ids[b]] >= k + 1, meaning ynew[ids[b]] = {x[D], y} holds the row iteration k + 1
   * appends, and goes on without a launch.  If the row does not come in time (or wait_ticks is 0,
   * or *abort_flag is set) it stores parked[ids[b]] = k + 1 and exits; the caller launches
   * iteration k + 1 in a later batch.  Every wave therefore ends within wait_ticks of host
   * silence.  ynew, yseq, parked and abort_flag are host-visible (pinned) memory. */
  const int32_t *targets;    /* device [n_models] */
  const double *ynew;        /* pinned [n_loops][D + 1], written by the host */
  const int32_t *yseq;       /* pinned [n_loops], written by the host after ynew */
  int32_t *parked;           /* pinned [n_loops], written by the device */
  const int32_t *abort_flag; /* pinned [1] */
  int64_t wait_ticks;        /* hipDeviceAttributeWallClockRate ticks; 0 = never wait */
  int32_t resident_loops;    /* workgroups the caller wants resident at once (all its loops): the
                                wait is dropped when the device cannot hold that many */
} bore_batch;
void bore_set_batch(const bore_batch *batch);

/* ---------------------------------------------------------------------------
 * Replica engine: n_loops independent BO loops on one GPU (BASELINE.json config 4), each
 * iteration being label -> fit -> sample + screen -> L-BFGS-B restarts -> pick as in the
 * reference's loop (README.rst:83-103; bore/plugins/hpbandster/base.py:216-262), driven by a
 * native host loop.  Two schedules (bore_engine_cfg.async_loops):
 *   0  the loops are split into `groups`, each stepping in lock-step on its own HIP stream; per
 *      iteration a group uploads one row per loop, makes six launches (append, labels, fit,
 *      sample + screen, restarts, pick) and downloads the suggestions;
 *   1  every loop advances on its own: ready loops are launched in batches (batch mode above; for
 *      the 2 -> 16-16-1 BASELINE model ONE fused kernel per loop-iteration, otherwise the launch
 *      chain), a loop's suggestion reaches the host through a flag the moment ITS restarts are
 *      done, and the loop joins the next launch without waiting for the slowest loop of a group.
 *      Needs num_starts <= 4 and ~14 hardware queues (GPU_MAX_HW_QUEUES, read at HIP start-up).
 * Same trajectories either way.  Only the objective is evaluated on the host, through the callback.
 * Not thread-safe per engine; engines are independent of each other.
 * ------------------------------------------------------------------------- */

/* y[i] = objective(x[i]) for n points: x host fp64 [n][D], y host fp64 [n].  Returns 0, or
 * non-zero to stop: bore_engine_run then drains the work in flight and returns BORE_E_CALLBACK. */
typedef int (*bore_objective_fn)(const double *x, int64_t n, int32_t D, double *y, void *user);
/* A built-in objective of that type: Branin-Hoo on its native box, rescaled to [0, 1]^2 (D = 2; the
 * synthetic objective of BASELINE.json's configs 1 and 4, bore_amd.engine.branin01 -- the same
 * operations in the same order in fp64).  Passed to bore_engine_create it keeps the interpreter out
 * of the host loop: a Python callback costs ~40 us per call, which at 512 loops was ALL of the host
 * thread's time and ~50 us of every loop-iteration's turn-around.  `user` is ignored. */
int bore_objective_branin01(const double *x, int64_t n, int32_t D, double *y, void *user);

typedef struct bore_engine_cfg {
  int32_t n_loops;      /* loops on this GPU */
  int32_t groups;       /* stream groups (clamped to n_loops) */
  int64_t loop_id0;     /* global id of the first loop: keys the device shuffle/candidate streams */
  int32_t n_init;       /* observations per loop at the start */
  int32_t epochs;       /* fit(epochs=..) per iteration */
  int32_t batch_size;   /* <= BORE_BATCH_MAX */
  int32_t num_starts;   /* argmax(num_starts=..) */
  int32_t num_samples;  /* argmax(num_samples=..) */
  int32_t transform;    /* enum bore_transform */
  int32_t deduplicate;  /* != 0: Record.is_duplicate as filter_fn (the plugin's rule) */
  int32_t async_loops;  /* != 0: loops advance individually (batch mode above; num_starts <= 4):
                           a loop re-enters the next launch as soon as ITS restarts are done */
  uint64_t seed;
  double gamma;
  bore_adam_cfg adam;
  bore_lbfgsb_opts lbfgsb;
  const double *low, *high; /* HOST fp64 [D]: the search box (copied) */
  /* ABI 12 (round 6; were the environment variables BORE_ASYNC_RESIDENT_US / _WORKERS / _QUEUE).  Asynchronous schedule: */
  int32_t resident_wait_us; /* how long a loop's workgroup waits on its CU for the objective value before it parks:
                               < 0 the default (2000), 0 = one launch per loop-iteration */
  int32_t worker_streams;   /* <= 0 the default (12 single-kernel launches in flight when fused, bounded by
                               GPU_MAX_HW_QUEUES - 2; 4 launch chains otherwise) */
  int32_t work_queue;       /* < 0 by size (more loops than the device holds at once), 0 never (the launch-per-batch
                               schedule of round 3), 1 always (one persistent launch fed by the host) */
  int32_t reserved;         /* 0 */
} bore_engine_cfg;

/* Sums since creation / the last reset (HIP-event durations of the two big kernels, the
 * algorithmic bytes of SURVEY.md 8d, counts). */
typedef struct bore_engine_stats {
  double fit_ms, fit_bytes, argmax_ms, argmax_bytes;
  double host_enqueue_s, host_finalize_s;
  int64_t fit_launches, argmax_launches, n_fg_rows, n_rounds, none_results;
  /* asynchronous schedule, fused iteration kernel: device-clock time of the phases of a
   * loop-iteration summed over `phase_iterations` of them (ns), and the host's view of the same
   * iterations: ready -> launched, launched -> result seen, result seen -> ready again (s) */
  double phase_ns_labels, phase_ns_fit, phase_ns_screen, phase_ns_lbfgsb;
  double ready_to_launch_s, launch_to_result_s, result_to_ready_s;
  int64_t phase_iterations, batches;
  /* asynchronous schedule: worker streams, and how many of them the device was seen to run at
   * once when the engine was created (a probe of spinning kernels).  Fewer than worker_streams
   * means streams share hardware queues -- GPU_MAX_HW_QUEUES was below worker_streams + 2 when
   * the HIP runtime initialised -- and launches of different batches serialise.  Not reset. */
  int64_t worker_streams, stream_concurrency;
  /* ABI 9: n_fg_rows counts the evaluations that RAN the network (what the roofline's algorithmic
   * bytes are made of); n_fg_requests = the optimisers' nfev, which also counts trial points served
   * from the image shortcut of the fused kernel (SciPy's nfev counts them too).  Launch-chain
   * schedules only see nfev: there the two are equal. */
  int64_t n_fg_requests;
  /* ABI 11 (asynchronous schedule, fused kernel; not reset): loops of this model one CU holds at the records'
   * current capacity, and the workgroups that serve the loops side by side -- n_loops when every loop is resident,
   * the size of the work-queue kernel's grid otherwise (0: launch chains).  Also: host threads of the last
   * work-queue run. */
  int64_t loops_per_cu, side_by_side_workgroups, host_threads;
} bore_engine_stats;

typedef struct bore_engine bore_engine;

/*
 *   theta0    HOST fp32 [n_loops][P]              initial weights (Keras defaults drawn by the caller)
 *   X0, y0    HOST fp64 [n_loops][n_init][D], [n_loops][n_init]
 *   mt_state  HOST uint32 [n_loops][625]          each loop's numpy RandomState (624 key words + pos):
 *             the fallback point of a loop whose argmax returns None is drawn from it
 *             (rs.uniform(low, high), bore/plugins/hpbandster/base.py:255-262)
 */
int bore_engine_create(const bore_mlp_desc *desc, const bore_engine_cfg *cfg, const float *theta0,
                       const double *X0, const double *y0, const uint32_t *mt_state,
                       bore_objective_fn objective, void *user, bore_engine **out);
/* Advance every loop by n_steps BO iterations (returns when all have finished). */
int bore_engine_run(bore_engine *engine, int n_steps);
/* Observations per loop so far (<0: bad engine). */
int64_t bore_engine_size(const bore_engine *engine);
/* X host fp64 [n_loops][N][D], y host fp64 [n_loops][N], N = bore_engine_size(). */
int bore_engine_observations(bore_engine *engine, double *X, double *y);
/* Classifier state, host buffers [n_loops][P] / [n_loops]; any may be NULL. */
int bore_engine_state(bore_engine *engine, float *theta, float *adam_m, float *adam_v,
                      int64_t *adam_t);
int bore_engine_get_stats(bore_engine *engine, bore_engine_stats *out, int reset);
/* Frees the engine's device and pinned memory and its streams.  NOT from inside an objective callback, nor from
 * another thread while ANY engine of the process is inside bore_engine_run on the same device: freeing device memory
 * waits for the device to be idle, and a running engine's resident / work-queue kernel waits on its CUs for objective
 * values that the host, stuck in the free, would never deliver.  (The Python wrapper defers such a close to the end of
 * the outermost run: bore_amd.engine.NativeEngine.close.) */
void bore_engine_destroy(bore_engine *engine);

#ifdef __cplusplus
}
#endif
#endif /* BORE_HIP_H */
