// bore_argmax.hip -- the acquisition side of the hot path on the device: label step,
// candidate sampling, screening + top-k, and the multi-start L-BFGS-B itself.
// gfx950 (MI355X) only.  C-ABI: include/bore_hip.h.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host_common.h"
#include "lbfgsb.h"
#include "mlp_device.h"
#include "mlp_regs.h"
#include "mlp_point.h"
#include "arg_bf16_mfma.h"

using namespace bore;

// ---------------------------------------------------------------------------
// labels: tau = np.quantile(y, gamma) (linear interpolation), z = y < tau
// ---------------------------------------------------------------------------
// (batch mode: slot -> loop ids[slot] with its own N = n_init + its[slot]; y and z `cap`-strided)
__device__ __forceinline__ void labels_body(const double *y, int N, double vi, float *z,
                                            double *tau_out, const int *ids, const int *its,
                                            int n_init, long long cap, double gamma,
                                            long long model, int it_now = -1) {
  // (it_now >= 0: the slot's iteration count, given by a caller that runs several iterations of
  // the loop in one launch -- bore_iter.hip -- instead of its[model])
  extern __shared__ float smem[];
  double *ys = reinterpret_cast<double *>(smem);  // [N] + 2 (a, b)
  long long stride = N;
  if (ids) {
    N = n_init + (it_now >= 0 ? it_now : uniform_i32(its[model]));
    model = uniform_i64(ids[model]);
    stride = cap;
    vi = (double)(N - 1) * gamma;  // numpy's virtual index for this slot's N
  }
  const double *ym = y + model * stride;
  for (int i = threadIdx.x; i < N; i += blockDim.x) ys[i] = ym[i];
  // numpy's _get_indexes: floor/ceil neighbours of the virtual index, clamped to the ends
  int lo, hi;
  if (vi >= (double)(N - 1)) {
    lo = hi = N - 1;
  } else if (vi < 0.0) {
    lo = hi = 0;
  } else {
    lo = (int)floor(vi);
    hi = lo + 1;
  }
  const double gam = (vi >= (double)(N - 1)) ? vi - (-1.0) : (vi < 0.0 ? vi : vi - floor(vi));
  __syncthreads();
  for (int i = threadIdx.x; i < N; i += blockDim.x) {
    const double yi = ys[i];
    int r = 0;
    for (int j = 0; j < N; ++j) {
      const double yj = ys[j];
      r += (yj < yi) || (yj == yi && j < i);
    }
    if (r == lo) ys[N] = yi;
    if (r == hi) ys[N + 1] = yi;
  }
  __syncthreads();
  // numpy's _lerp
  const double a = ys[N], b = ys[N + 1];
  const double diff = b - a;
  double tau = a + diff * gam;
  if (gam >= 0.5) tau = b - diff * (1.0 - gam);
  for (int i = threadIdx.x; i < N; i += blockDim.x) z[model * stride + i] = ys[i] < tau ? 1.f : 0.f;
  if (threadIdx.x == 0 && tau_out) tau_out[model] = tau;
}

__global__ __launch_bounds__(BORE_THREADS) void labels_kernel(const double *y, int N, double vi,
                                                              float *z, double *tau_out,
                                                              const int *ids, const int *its,
                                                              int n_init, long long cap,
                                                              double gamma) {
  labels_body(y, N, vi, z, tau_out, ids, its, n_init, cap, gamma, blockIdx.x);
}

extern "C" int bore_labels(int n_models, const double *y, int64_t N, double gamma, float *z,
                           double *tau, void *stream) {
  if (n_models < 1 || !y || !z) return fail(BORE_E_INVALID, "labels: bad argument");
  if (N < 1 || N > 16384) return fail(BORE_E_UNSUPPORTED, "labels: N=%lld outside 1..16384", (long long)N);
  if (!(gamma >= 0.0 && gamma <= 1.0)) return fail(BORE_E_INVALID, "labels: gamma outside [0, 1]");
  // numpy's virtual index of the default ("linear") method: (n - 1) * q
  const double vi = (double)(N - 1) * gamma;
  const size_t bytes = ((size_t)N + 2) * 8;
  int rc = allow_lds(labels_kernel, bytes);
  if (rc) return rc;
  if (g_batch && tau) return fail(BORE_E_INVALID, "labels: no tau output in batch mode");
  hipLaunchKernelGGL(labels_kernel, dim3(n_models), dim3(BORE_THREADS), bytes, (hipStream_t)stream,
                     y, (int)N, vi, z, tau, g_batch ? g_batch->ids : nullptr,
                     g_batch ? g_batch->its : nullptr, g_batch ? g_batch->n_init : 0,
                     g_batch ? (long long)g_batch->cap : 0LL, gamma);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// candidates: X ~ U(low, high) from a counter-based stream
// ---------------------------------------------------------------------------
struct BoxArgs {
  double lo[BORE_DIM_MAX], hi[BORE_DIM_MAX];
};

__host__ __device__ __forceinline__ unsigned long long candidate_base(unsigned long long seed,
                                                                      long long model,
                                                                      long long draw) {
  unsigned long long h = mix64(seed ^ 0xA0761D6478BD642FULL);  // domain tag: not the shuffle stream
  h = mix64(h + 0x9E3779B97F4A7C15ULL * (unsigned long long)(model + 1));
  return mix64(h + 0xD1B54A32D192ED03ULL * (unsigned long long)(draw + 1));
}

__global__ __launch_bounds__(BORE_THREADS) void candidates_kernel(unsigned long long seed,
                                                                 long long model0, long long draw,
                                                                 long long n_samples, int D,
                                                                 const BoxArgs box, double *X) {
  const long long model = blockIdx.y;
  const unsigned long long base = candidate_base(seed, model0 + model, draw);
  const long long total = n_samples * D;
  double *Xm = X + model * total;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total;
       i += (long long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const unsigned long long r = mix64(base + 0x8CB92BA72F3D8DD7ULL * (unsigned long long)(i + 1));
    const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);  // 53 bits -> [0, 1)
    Xm[i] = box.lo[d] + (box.hi[d] - box.lo[d]) * u;
  }
}

extern "C" int bore_uniform_candidates(uint64_t seed, int64_t model_index0, int n_models,
                                       int64_t draw_index, int64_t n_samples, int D,
                                       const double *low, const double *high, double *X,
                                       void *stream) {
  if (n_models < 1 || n_samples < 1 || !low || !high || !X)
    return fail(BORE_E_INVALID, "uniform_candidates: bad argument");
  if (D < 1 || D > BORE_DIM_MAX)
    return fail(BORE_E_UNSUPPORTED, "uniform_candidates: D must be 1..%d", BORE_DIM_MAX);
  if (n_models > 65535) return fail(BORE_E_UNSUPPORTED, "uniform_candidates: n_models > 65535");
  BoxArgs box;
  for (int d = 0; d < D; ++d) {
    box.lo[d] = low[d];
    box.hi[d] = high[d];
  }
  long long gx = (n_samples * D + BORE_THREADS - 1) / BORE_THREADS;
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(candidates_kernel, dim3((unsigned)gx, n_models), dim3(BORE_THREADS), 0,
                     (hipStream_t)stream, seed, (long long)model_index0, (long long)draw_index,
                     (long long)n_samples, D, box, X);
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// screening + top-k: one workgroup per model
// ---------------------------------------------------------------------------
struct ScreenArgs {
  MlpLayout L;
  const float *theta;
  const double *X;
  double *x0;
  int *idx;
  float *pred;
  long long n_samples;
  int x_shared, R, n_pad;
  int o_tile, o_keys, o_layout, total;  // float offsets (o_keys is 8-byte aligned)
  // sampled != 0: no X in memory -- row i of model m IS row i of bore_uniform_candidates(seed,
  // model0 + m, draw), recomputed from the counter stream where it is needed
  int sampled, o_box;
  unsigned long long seed;
  long long model0, draw;
  const int *ids, *its;  // batch mode: theta and the stream of loop ids[slot], draw its[slot]
  BoxArgs box;
  int d_in;  // the net's input dimension (<= the static shape's: stage_theta_in)
};

// float -> unsigned that sorts like the float
__device__ __forceinline__ unsigned orderable(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const unsigned long long o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  return v;
}

// MODE 0: the whole screen in one workgroup per model.  For ONE (or a few) wide models that leaves
// the device idle while a single CU walks thousands of candidates, so the launcher may split it:
// MODE 1 = the predictions only, grid (models, parts), workgroup `part` taking every parts-th
// round of row-blocks and writing a.pred; MODE 2 = the selection, reading the predictions back
// (same predictions, same keys, same order: the two forms give the same bits).
template <int SHAPE, bool BF16 = false, int MODE = 0>
__device__ __forceinline__ void screen_body(const ScreenArgs &a, const long long model,
                                            const int it_now = -1) {
  extern __shared__ float smem[];
  constexpr MlpLayout Lc = bore_static_layout(SHAPE > 0 ? SHAPE : 0, 0, BORE_BATCH_MAX);
  const MlpLayout &L = begin_kernel<SHAPE>(Lc, a.L, smem, a.total, a.o_layout);
  int tid = threadIdx.x;
  BORE_OPAQUE_TID(tid);
  const int nthr = blockDim.x;
  // `model` is the output slot
  const long long lid = a.ids ? uniform_i64(a.ids[model]) : model;        // whose weights and stream
  const int n = layer_count<SHAPE>(L), D = (bore_shape_takes_fewer_inputs(SHAPE) && !BF16) ? a.d_in : L.w[0];
  const int Ns = (int)a.n_samples;
  float *th = smem, *tile = smem + a.o_tile;
  unsigned long long *keys = reinterpret_cast<unsigned long long *>(smem + a.o_keys);
  const int wv = tid >> 6, lane = tid & 63, m16 = lane & 15, q4 = lane >> 4;
  if constexpr (MODE != 2) {
    if constexpr (BF16) arg_bf16_stage<SHAPE>(a.theta + lid * L.P, smem);  // (arg_bf16_mfma.h)
    else if constexpr (bore_shape_takes_fewer_inputs(SHAPE)) stage_theta_in<false>(L, n, a.theta, lid, D, smem);
    else stage_theta<false>(L, n, a.theta + lid * L.P, smem);
  }
  const double *X = a.sampled ? nullptr : a.X + (a.x_shared ? 0 : model * a.n_samples * D);
  double *blo = reinterpret_cast<double *>(smem + a.o_box), *bhi = blo + D;
  if (a.sampled && tid < D) {  // the box is indexed per lane: LDS copy
    blo[tid] = a.box.lo[tid];
    bhi[tid] = a.box.hi[tid];
  }
  const unsigned long long cbase =
      a.sampled ? candidate_base(a.seed, a.model0 + lid,
                                 a.ids ? (long long)(it_now >= 0 ? it_now : uniform_i32(a.its[model])) : a.draw)
                : 0ULL;
  auto xval = [&](long long row, int d) -> double {
    if (a.sampled) {
      const long long i = row * D + d;
      const unsigned long long r = mix64(cbase + 0x8CB92BA72F3D8DD7ULL * (unsigned long long)(i + 1));
      const double u = (double)(r >> 11) * (1.0 / 9007199254740992.0);
      return blo[d] + (bhi[d] - blo[d]) * u;
    }
    return X[row * D + d];
  };
  __syncthreads();
  // predictions -> sort keys: ascending key == descending prediction, ties to the lower row.
  // Every wave walks its own 16-row blocks of the candidates.
  const int waves = L.tbp >> 4;
  const int n_blocks = (Ns + 15) >> 4;
  using Net = RegNet<(SHAPE > 0 ? SHAPE : 1), 0, BF16>;
  const typename Net::WT *thw = reinterpret_cast<const typename Net::WT *>(smem);
  Net net;  // static shapes: the weights stay in this lane's registers for every row-block
  if constexpr (SHAPE > 0 && MODE != 2 && !BF16) {
    if constexpr (Net::RT_ACT) net.set_acts(a.L);
    net.load_fwd(thw);
  }
  using ANet = ArgBf16Net<(BF16 ? SHAPE : 3)>;  // a bfloat16 model: the bf16 matrix cores
  ANet anet;
  if constexpr (BF16 && MODE != 2) anet.set_acts(a.L);
  const int g0 = MODE == 1 ? wv + waves * (int)blockIdx.y : wv;
  const int gs = MODE == 1 ? waves * (int)gridDim.y : waves;
  if (MODE != 2 && wv < waves)
    for (int g = g0; g < n_blocks; g += gs) {
      const int row = g * 16 + m16;
      float p;  // prediction of row g*16 + lane, in the lanes < 16
      if constexpr (BF16) {
        static_assert(!BF16 || bore_shape_is_wide(SHAPE), "bfloat16: wide static shapes");
        bf16x8_t xf[ANet::CF1];
        ANet::make_xfrag(xf, [&](int d) -> float { return (d < D && row < Ns) ? (float)xval(row, d) : 0.f; });
        anet.predict(arg_bf16_images<SHAPE>(smem), xf);
        p = anet.out;
      } else if constexpr (SHAPE > 0) {  // activations in registers (mlp_regs.h)
        float xin[Net::KC0];
#pragma unroll
        for (int kc = 0; kc < Net::KC0; ++kc) {
          const int d = 4 * kc + q4;
          xin[kc] = (d < D && row < Ns) ? Net::rnd((float)xval(row, d)) : 0.f;
        }
        net.forward(thw, xin, false);
        p = net.h[Net::n][0][0];
      } else {
        float *A0 = tile + L.aoff[0] + (wv * 16 + m16) * L.lda[0];
        for (int d = q4; d < D; d += 4) A0[d] = row < Ns ? (float)xval(row, d) : 0.f;
        wave_lds_sync();
        fwd_all(L, n, th, tile, wv, false);
        p = tile[L.aoff[n] + (wv * 16 + (lane & 15)) * L.lda[n]];
        wave_lds_sync();
      }
      if (lane < 16 && g * 16 + lane < Ns) {
        const int r = g * 16 + lane;
        if constexpr (MODE == 0) keys[r] = ((unsigned long long)(~orderable(p)) << 32) | (unsigned)r;
        if (MODE == 1 || a.pred) a.pred[model * a.n_samples + r] = p;
      }
    }
  if constexpr (MODE == 1) return;
  if constexpr (MODE == 2)
    for (int r = tid; r < Ns; r += nthr)
      keys[r] = ((unsigned long long)(~orderable(a.pred[model * a.n_samples + r])) << 32) | (unsigned)r;
  for (int i = Ns + tid; i < a.n_pad; i += nthr) keys[i] = ~0ULL;
  __syncthreads();

  int *idx_out = a.idx + model * a.R;
  if (a.R <= 16) {
    // R passes of "smallest key greater than the previous pick" (keys are distinct)
    unsigned long long *red = keys + a.n_pad;  // [4] wave minima + [1] the pick; picks at [8..8+R)
    unsigned long long prev = 0;
    for (int r = 0; r < a.R; ++r) {
      unsigned long long best = ~0ULL;
      for (int i = tid; i < Ns; i += nthr) {
        const unsigned long long k = keys[i];
        if ((r == 0 || k > prev) && k < best) best = k;
      }
      best = wave_min_u64(best);
      if ((tid & 63) == 0) red[tid >> 6] = best;
      __syncthreads();
      if (tid == 0) {
        unsigned long long b = red[0];
        for (int w = 1; w < (nthr >> 6); ++w) b = red[w] < b ? red[w] : b;
        red[4] = b;
        red[8 + r] = b;
        idx_out[r] = (int)(unsigned)(b & 0xFFFFFFFFu);
      }
      __syncthreads();
      prev = red[4];
    }
  } else {
    // bitonic sort of the padded key array
    for (int k = 2; k <= a.n_pad; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = tid; i < a.n_pad; i += nthr) {
          const int ixj = i ^ j;
          if (ixj > i) {
            const unsigned long long x = keys[i], y = keys[ixj];
            const bool up = (i & k) == 0;
            if ((x > y) == up) {
              keys[i] = y;
              keys[ixj] = x;
            }
          }
        }
        __syncthreads();
      }
    for (int r = tid; r < a.R; r += nthr) idx_out[r] = (int)(unsigned)(keys[r] & 0xFFFFFFFFu);
  }
  __syncthreads();
  // gather the chosen rows; the picks are still in LDS
  const unsigned long long *picks = a.R <= 16 ? keys + a.n_pad + 8 : keys;
  double *x0 = a.x0 + model * (long long)a.R * D;
  for (int i = tid; i < a.R * D; i += nthr) {
    const int r = i / D, d = i - r * D;
    x0[i] = xval((long long)(unsigned)(picks[r] & 0xFFFFFFFFu), d);
  }
}

template <int SHAPE, bool BF16 = false, int MODE = 0>
__global__ __launch_bounds__(BORE_THREADS) void screen_topk_kernel(const ScreenArgs a) {
  screen_body<SHAPE, BF16, MODE>(a, blockIdx.x);
}

struct SampleSpec {
  uint64_t seed;
  int64_t model_index0, draw_index;
  const double *low, *high;
};

static int screen_build(const bore_mlp_desc *desc, int n_models, const float *theta,
                        const double *X_init, const SampleSpec *spec, int64_t n_samples,
                        int x_shared, int num_starts, double *x0, int32_t *idx, float *pred,
                        ScreenArgs &a, size_t &lds_floats, int &shape_out) {
  if (n_samples < 1 || n_samples > (1 << 20))
    return fail(BORE_E_INVALID, "screen_topk: n_samples out of range");
  if (num_starts < 1 || num_starts > n_samples)
    return fail(BORE_E_INVALID, "screen_topk: need 1 <= num_starts <= n_samples");
  int n_pad = 1;
  while (n_pad < n_samples) n_pad <<= 1;
  const size_t key_floats = 2 * ((size_t)n_pad + 32) + 1 + BORE_LAYOUT_FLOATS + 4 + 4 * BORE_DIM_MAX + 4;
  int rc = check_common(desc, n_models, 0, BORE_BATCH_MAX, true, key_floats, &a.L);
  if (rc) return rc;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "screen_topk: the last Dense layer must have 1 unit");
  if (!theta || (!X_init && !spec) || !x0 || !idx) return fail(BORE_E_INVALID, "screen_topk: null pointer");
  a.theta = theta; a.X = X_init; a.x0 = x0; a.idx = idx; a.pred = pred;
  a.n_samples = n_samples; a.x_shared = x_shared; a.R = num_starts; a.n_pad = n_pad;
  a.sampled = spec != nullptr;
  a.seed = 0; a.model0 = 0; a.draw = 0;
  a.ids = a.its = nullptr;
  if (g_batch) {
    if (!spec) return fail(BORE_E_INVALID, "screen_topk: batch mode needs the sampled form");
    a.ids = g_batch->ids; a.its = g_batch->its;
  }
  if (spec) {
    const int D = desc->input_dim;
    if (D > BORE_DIM_MAX) return fail(BORE_E_UNSUPPORTED, "sample_screen_topk: D must be 1..%d", BORE_DIM_MAX);
    if (!spec->low || !spec->high) return fail(BORE_E_INVALID, "sample_screen_topk: null bounds");
    a.seed = spec->seed; a.model0 = spec->model_index0; a.draw = spec->draw_index;
    for (int d = 0; d < D; ++d) {
      a.box.lo[d] = spec->low[d];
      a.box.hi[d] = spec->high[d];
    }
  }
  size_t off = a.L.P_lds;
  a.d_in = desc->input_dim;
  const int flav = bore_acq_flavour(desc, true);
  const bool bf_img = desc->compute == BORE_COMPUTE_BF16 && bore_shape_is_wide(flav);
  // (a bfloat16 model: the fragment-order images of arg_bf16_mfma.h, activations in registers)
  if (bf_img) off = flav == 3 ? ArgBf16Plan<3>::floats : ArgBf16Plan<4>::floats;
  a.o_tile = (int)off; off += bf_img ? 0 : a.L.tile_floats;
  off = (off + 3) & ~(size_t)3;
  a.o_box = (int)off; off += 4 * (size_t)desc->input_dim;
  off = (off + 1) & ~(size_t)1;
  a.o_keys = (int)off; off += 2 * ((size_t)n_pad + 32);
  a.total = (int)off;
  off = (off + 3) & ~(size_t)3;
  a.o_layout = (int)off; off += BORE_LAYOUT_FLOATS;
  lds_floats = off;
  shape_out = flav;  // (static flavours do not use the tile)
  return 0;
}

static int screen_launch(const bore_mlp_desc *desc, int n_models, const float *theta,
                         const double *X_init, const SampleSpec *spec, int64_t n_samples,
                         int x_shared, int num_starts, double *x0, int32_t *idx, float *pred,
                         void *stream) {
  ScreenArgs a;
  size_t off = 0;
  int shape = 0;
  int rc = screen_build(desc, n_models, theta, X_init, spec, n_samples, x_shared, num_starts, x0, idx,
                        pred, a, off, shape);
  if (rc) return rc;
  // A few wide models, many candidates, and the caller gave a prediction buffer: the predictions run
  // on `parts` workgroups per model, then one workgroup per model selects (screen_body, MODE 1 / 2).
  // BORE_SCREEN_SPLIT = 0 / 1 forces either form.
  if (pred && !g_batch && bore_shape_is_wide(shape)) {
    const int forced = getenv("BORE_SCREEN_SPLIT") ? atoi(getenv("BORE_SCREEN_SPLIT")) : -1;
    const int n_blocks = (int)((n_samples + 15) >> 4);
    int parts = device_cus() / (n_models > 0 ? n_models : 1);
    if (parts > n_blocks / 8) parts = n_blocks / 8;  // at least two row-blocks per wave
    if (parts > 64) parts = 64;
    if (forced < 0 ? parts >= 4 : (forced != 0 && parts >= 2)) {
      const bool bf = desc->compute == BORE_COMPUTE_BF16;
#define BORE_LAUNCH_SPLIT(S, B)                                                                          \
  {                                                                                                      \
    if ((rc = allow_lds((screen_topk_kernel<S, B, 1>), off * 4)) ||                                       \
        (rc = allow_lds((screen_topk_kernel<S, B, 2>), off * 4)))                                         \
      return rc;                                                                                         \
    hipLaunchKernelGGL((screen_topk_kernel<S, B, 1>), dim3(n_models, parts), dim3(BORE_THREADS), off * 4, \
                       (hipStream_t)stream, a);                                                          \
    hipLaunchKernelGGL((screen_topk_kernel<S, B, 2>), dim3(n_models), dim3(BORE_THREADS), off * 4,        \
                       (hipStream_t)stream, a);                                                          \
  }
      if (!bore_flavour_built(shape)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#if BORE_ON_3
      if (shape == 3 && !bf) BORE_LAUNCH_SPLIT(3, false)
      if (shape == 3 && bf) BORE_LAUNCH_SPLIT(3, true)
#endif
#if BORE_ON_4
      if (shape == 4 && !bf) BORE_LAUNCH_SPLIT(4, false)
      if (shape == 4 && bf) BORE_LAUNCH_SPLIT(4, true)
#endif
#undef BORE_LAUNCH_SPLIT
      HIP_TRY(hipGetLastError());
      return 0;
    }
  }
  if (desc->compute == BORE_COMPUTE_BF16) {
    if (!bore_shape_is_wide(shape)) return fail(BORE_E_UNSUPPORTED, kBf16Shapes);
    if (!bore_flavour_built(shape)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#if BORE_ON_3
    if (shape == 3) {
      rc = allow_lds((screen_topk_kernel<3, true>), off * 4);
      if (rc) return rc;
      hipLaunchKernelGGL((screen_topk_kernel<3, true>), dim3(n_models), dim3(BORE_THREADS), off * 4,
                         (hipStream_t)stream, a);
    }
#endif
#if BORE_ON_4
    if (shape == 4) {
      rc = allow_lds((screen_topk_kernel<4, true>), off * 4);
      if (rc) return rc;
      hipLaunchKernelGGL((screen_topk_kernel<4, true>), dim3(n_models), dim3(BORE_THREADS), off * 4,
                         (hipStream_t)stream, a);
    }
#endif
    HIP_TRY(hipGetLastError());
    return 0;
  }
#define BORE_LAUNCH_SCREEN(S)                                                                 \
  case S:                                                                                     \
    rc = allow_lds(screen_topk_kernel<S>, off * 4);                                           \
    if (rc) return rc;                                                                        \
    hipLaunchKernelGGL(screen_topk_kernel<S>, dim3(n_models), dim3(BORE_THREADS), off * 4,    \
                       (hipStream_t)stream, a);                                               \
    break;
  if (!bore_flavour_built(shape)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
  switch (shape) {
#if BORE_ON_1
    BORE_LAUNCH_SCREEN(1)
#endif
#if BORE_ON_2
    BORE_LAUNCH_SCREEN(2)
#endif
#if BORE_ON_3
    BORE_LAUNCH_SCREEN(3)
#endif
#if BORE_ON_4
    BORE_LAUNCH_SCREEN(4)
#endif
#if BORE_ON_5
    BORE_LAUNCH_SCREEN(5)
#endif
#if BORE_ON_N1
    BORE_LAUNCH_SCREEN(-1)
#endif
#if BORE_ON_N2
    BORE_LAUNCH_SCREEN(-2)
#endif
#if BORE_ON_N3
    BORE_LAUNCH_SCREEN(-3)
#endif
#if BORE_ON_N4
    BORE_LAUNCH_SCREEN(-4)
#endif
    default:
#if !BORE_ON_0
      return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#endif
#if BORE_ON_0
    BORE_LAUNCH_SCREEN(0)
#endif
  }
#undef BORE_LAUNCH_SCREEN
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int bore_screen_topk(const bore_mlp_desc *desc, int n_models, const float *theta,
                                const double *X_init, int64_t n_samples, int x_shared,
                                int num_starts, double *x0, int32_t *idx, float *pred,
                                void *stream) {
  if (!X_init) return fail(BORE_E_INVALID, "screen_topk: null pointer");
  return screen_launch(desc, n_models, theta, X_init, nullptr, n_samples, x_shared, num_starts, x0,
                       idx, pred, stream);
}

extern "C" int bore_sample_screen_topk(const bore_mlp_desc *desc, int n_models, const float *theta,
                                       uint64_t seed, int64_t model_index0, int64_t draw_index,
                                       int64_t n_samples, const double *low, const double *high,
                                       int num_starts, double *x0, int32_t *idx, float *pred,
                                       void *stream) {
  const SampleSpec spec{seed, model_index0, draw_index, low, high};
  return screen_launch(desc, n_models, theta, nullptr, &spec, n_samples, 0, num_starts, x0, idx,
                       pred, stream);
}

// ---------------------------------------------------------------------------
// multi-start L-BFGS-B: grid = (models, problem blocks)
//
// A workgroup takes up to 64 restarts of one model.  Problem p runs on wave p & 3, lane
// p >> 2, and its point lives in tile row 16 (p & 3) + (p >> 2) -- i.e. in the 16-row block
// its OWN wave evaluates.  So a wave advances its (<= 16) state machines, evaluates their
// pending points with one MFMA pass over weights held in LDS, feeds f/g back and repeats,
// without ever waiting for another wave: no workgroup barrier inside the optimisation.
// ---------------------------------------------------------------------------
struct LbfgsbArgs {
  MlpLayout L;
  const float *theta;
  const double *x0;
  double *x, *fun, *jac;
  int *info;
  BoxArgs box;
  int nbd[BORE_DIM_MAX];
  lbfgsb::Options opt;
  int R, PB, transform, max_rounds;  // PB = problems per workgroup
  // queue != 0: one problem per wave at a time, PB may exceed the waves: a wave that finishes a problem
  // draws the workgroup's next one from a counter in LDS (o_queue) and reuses its workspace slot
  int queue, o_queue;
  // big != NULL: the optimiser's two 2m x 2m matrices of every wave live in a slot of this device pool
  // (host_common.h: big_pool) instead of in LDS: big_wave doubles per wave, slots of 8 waves
  double *big;
  int *big_locks;
  int big_wave;
  float sign;
  // LDS carve (float offsets; the fp64 regions are 8-byte aligned)
  int o_tile, o_vals, o_box, o_prob, prob_floats, o_state, o_dw, o_iw, o_layout, total;
  // batch mode (bore_set_batch): weights of loop ids[slot]; each loop's pick is published to the
  // host as soon as ITS restarts are done (one workgroup per loop: R <= 4)
  const int *ids, *its;
  int n_init, dedup, o_res;
  long long cap;
  const double *X_seen;
  double *result;
  int *flag;
  const long long *stamps;  // [n_loops][4] clock stamps of the fused kernel's earlier phases, or NULL
  int d_in;  // the net's input dimension = the optimiser's problem size (<= the static shape's: stage_theta_in)
  // PUBLISH = 1 launches: 0 = the flag goes out by a system-scope atomic exchange on the pinned block (the link to
  // the host does native atomics: host_common.h, device_host_atomics), 1 = by a release store behind a system-scope
  // fence (on a link without them the exchange may never land and the host would wait for a flag forever)
  int flag_by_store;
};

// -DBORE_STAMPS: cycles spent in the optimiser / in f-g evaluation by wave 0 of workgroup 0
#ifdef BORE_STAMPS
__device__ long long g_lstamps[8];
extern "C" int bore_debug_lstamps(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstamps), sizeof(long long) * 8);
}
extern "C" int bore_debug_lphases(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lbfgsb::g_lb_phase), sizeof(long long) * 16);
}
extern "C" int bore_debug_lphases_reset(void) {
  long long z[16] = {0};
  static unsigned long long zz[LB_PP_MAX][64];
  (void)hipMemcpyToSymbol(HIP_SYMBOL(lbfgsb::g_lb_pp), zz, sizeof(zz));
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(lbfgsb::g_lb_phase), z, sizeof(z));
}
extern "C" int bore_debug_lpp(unsigned long long *out) {  // [LB_PP_MAX][64]
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(lbfgsb::g_lb_pp),
                                  sizeof(unsigned long long) * LB_PP_MAX * 64);
}
#define BORE_LCLOCK() clock64()
#else
#define BORE_LCLOCK() 0LL
#endif

// which static shapes evaluate a wave's single point on the vector ALU (mlp_point.h)
#define BORE_POINT_SHAPE(S) ((S) != 1)


// LEAN: the network's weight operands are re-requested from LDS for every evaluation instead of
// living in registers across the optimiser (the fused iteration kernel is held to 256 VGPRs).
// ALWAYS_COOP: the caller's launches never give a wave more than one problem at a time (the fused
// iteration kernel): the lane-per-problem loop is not compiled.
// MAYQ: the caller's launches may set a.queue (the fused iteration kernel's never do).
// PUBLISH (batch mode: how a loop's pick reaches the host).  0: behind a system-scope release -- on gfx950 a
// write-back of the XCD's whole L2 (buffer_wbl2 sc0 sc1), after which ANY later reader of the loop's state, on
// any XCD or the host, is served.  1: the caller guarantees that the loop's next iteration runs on this XCD (the
// resident kernel: the same workgroup; the work-queue kernel: one queue per XCD) -- the pick and the flag go to the
// pinned block as write-through stores, the flag after the others have been acknowledged, and the L2 keeps its
// dirty lines: that write-back, one per loop-iteration and serialised per XCD, was what held every schedule of
// round 4 at ~0.71 M loop-iterations per second (profiles/r5/ab_log.txt).
template <int SHAPE, bool BF16 = false, bool LEAN = false, bool ALWAYS_COOP = false, bool MAYQ = true, int PUBLISH = 0>
__device__ __forceinline__ void lbfgsb_body(const LbfgsbArgs &a, const long long model,
                                            const int block_y, const int it_now = -1) {
  extern __shared__ float smem[];
  constexpr MlpLayout Lc = bore_static_layout(SHAPE > 0 ? SHAPE : 0, 2, BORE_BATCH_MAX);
  const long long c_enter = BORE_LCLOCK();
  const MlpLayout &L = begin_kernel<SHAPE>(Lc, a.L, smem, a.total, a.o_layout);
  const long long c_begun = BORE_LCLOCK();
  int tid = threadIdx.x;
  BORE_OPAQUE_TID(tid);
  // (the wave number is wave-uniform: said so, the optimiser's two dozen workspace addresses -- all of the form
  // base(wave) + offset -- are scalar values instead of a vector register each)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // `model` is the slot: it indexes x0 / x / fun / jac / info
  const long long lid = a.ids ? uniform_i64(a.ids[model]) : model;  // whose weights (and record, and result)
  const int n_lay = layer_count<SHAPE>(L), D = (bore_shape_takes_fewer_inputs(SHAPE) && !BF16) ? a.d_in : L.w[0];
  const int p0 = block_y * a.PB;                  // first problem of this workgroup
  const int np = min(a.PB, a.R - p0);             // problems here (>= 1 by grid construction)
  float *th = smem, *tile = smem + a.o_tile, *vals = smem + a.o_vals;
  // the box, too, is indexed at run time by the optimiser: LDS copy (see stage_layout)
  double *blo = reinterpret_cast<double *>(smem + a.o_box), *bhi = blo + D;
  int *bnbd = reinterpret_cast<int *>(bhi + D);
  if (tid < D) {
    blo[tid] = a.box.lo[tid];
    bhi[tid] = a.box.hi[tid];
    bnbd[tid] = a.nbd[tid];
  }
  if constexpr (BF16) arg_bf16_stage<SHAPE>(a.theta + lid * L.P, smem);  // (arg_bf16_mfma.h)
  else if constexpr (bore_shape_takes_fewer_inputs(SHAPE)) stage_theta_in<false>(L, n_lay, a.theta, lid, D, smem);
  else stage_theta<false>(L, n_lay, a.theta + lid * L.P, smem);
  __syncthreads();

  // Which problem this thread works on.  With at most one problem per wave (np <= 4: the
  // BASELINE config-1 shape, 3 restarts) ALL 64 lanes of wave wv run problem wv together
  // (lbfgsb::Coop); otherwise lane s < 16 of wave wv runs problem 4 s + wv on its own.
  // Batch mode with 5..16 restarts (the reference's default is num_starts = 5) keeps the loop in
  // ONE workgroup, as the per-loop pick below needs: wave wv runs problems wv, wv + 4, .. one
  // after the other, each with all its lanes (`passes`).  Same bits every way (tested).
  // (launches with more workgroups than CUs may run 5..8 waves per workgroup, one problem each:
  // lbfgsb_kernel_w8; batch mode always has four)
#ifdef BORE_STAMPS
  lbfgsb::g_lb_lds[wv & 15][lane] = 0;
#endif
  const int NW = (int)(blockDim.x >> 6);
  const bool multi = a.result && np > 4 && np <= 16;
  // Many problems per workgroup, one per wave at a time (a.queue, round 4).  Round 3 launched one
  // workgroup per 4 (8) restarts: every workgroup zeroed its LDS and staged the model's weights for
  // four problems, and lived as long as the SLOWEST of them while the other waves idled.  Now a model's
  // restarts go to a few workgroups with hundreds of problems each, and a wave that finishes one draws
  // the next from a counter in LDS: the weights are staged once per workgroup, the waves stay busy
  // until the workgroup's problems run out.  Same results (a problem's arithmetic does not depend on
  // the wave or the order).
  const bool queue = MAYQ && a.queue != 0;
  const bool coop = np <= NW || multi || queue;
  const int passes = multi ? (np + 3) / 4 : (queue ? 0x7fffffff : 1);
  int *qnext = reinterpret_cast<int *>(smem + a.o_queue);
  if (queue && tid == 0) *qnext = NW;  // (the first NW problems go to the waves in order)
  // (the pooled matrices are compiled into the one kernel whose launches use them: the eight-wave kernel of
  // the 128-wide shape; lbfgsb_build decides per launch)
  constexpr bool BIG_OK = ALWAYS_COOP && SHAPE == 4;
  const bool use_big = BIG_OK && a.big != nullptr;
  if (use_big && tid == 0) {
    // a free slot of the pool: more slots than workgroups can be resident at once, so the walk ends
    int sl = (int)((blockIdx.x * gridDim.y + blockIdx.y) % BORE_BIG_SLOTS);
    while (__hip_atomic_exchange(a.big_locks + sl, 1, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != 0)
      sl = sl + 1 == BORE_BIG_SLOTS ? 0 : sl + 1;
    qnext[1] = sl;
    qnext[2] = 0;  // waves of this workgroup that are done with the slot
  }
  const int myrow = coop ? wv * 16 : wv * 16 + lane;
  // (ALWAYS_COOP kernels run one problem per wave or nothing: the lane count is a compile-time 64
  // there, which is what lets lbfgsb.h's LB_UNI move the optimiser's integers to scalar registers)
  const lbfgsb::Coop cp = (ALWAYS_COOP || coop) ? lbfgsb::Coop{lane, 64} : lbfgsb::Coop{0, 1};
  __syncthreads();  // weights staged; from here on the waves are independent
  double *my_big = nullptr;
  int big_slot = 0;
  if (use_big) {
    big_slot = uniform_i32(qnext[1]);
    my_big = a.big + ((size_t)big_slot * 8 + wv) * (size_t)a.big_wave;
  }
  const long long c_staged = BORE_LCLOCK();
  if (wv >= np) {  // wave without problems (np < 4)
    if (use_big && lane == 0 && atomicAdd(qnext + 2, 1) == NW - 1)
      __hip_atomic_store(a.big_locks + big_slot, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  double *res = reinterpret_cast<double *>(smem + a.o_res);  // batch mode: [np][D + 3] fun, status, nfev, x
  int *cnt = reinterpret_cast<int *>(res + (multi ? np : 4) * (D + 3));
  for (int pass = 0; pass < passes; ++pass) {
  int myp;
  if (queue) {
    int nxt = wv;
    if (pass > 0) {
      if (lane == 0) nxt = atomicAdd(qnext, 1);
      nxt = __builtin_amdgcn_readfirstlane(nxt);
    }
    myp = nxt < np ? nxt : -1;
  } else {
    myp = coop ? (wv + 4 * pass < np ? wv + 4 * pass : -1)
               : ((lane < 16 && lane * 4 + wv < np) ? lane * 4 + wv : -1);
  }
  if (coop && myp < 0) break;
  // The scalar state of the optimiser stays in this thread's REGISTERS for the whole launch
  // (its vectors and matrices are in LDS): kept in memory, every store to a workspace array
  // would force the compiler to reload the state fields it may alias.
  lbfgsb::State st;
  lbfgsb::Work wk;
  if (myp >= 0) {
    float *base = smem + a.o_prob + (size_t)(queue ? wv : myp) * a.prob_floats;  // (queue: the wave's slot)
    if (queue && pass > 0) {  // a reused slot starts as the zeroed LDS a first problem finds (prob_floats % 4 == 0)
      float4 *b4 = reinterpret_cast<float4 *>(base);
      for (int i = lane; i < a.prob_floats / 4; i += 64) b4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      wave_lds_sync();
    }
    wk = lbfgsb::make_work(reinterpret_cast<double *>(base + a.o_dw),
                           reinterpret_cast<int *>(base + a.o_iw), D, a.opt.m, my_big);
    // (the pool slot holds the last problem's matrices: the optimiser zeroes them if and when this problem's first
    // subspace minimisation comes -- as the zeroed LDS would have been found -- not 6.4 KB of writes per problem)
    if (my_big) wk.vm = lbfgsb::LB_BIG_LAZY;
    const double *x0 = a.x0 + (model * a.R + p0 + myp) * (long long)D;
    lbfgsb::lbfgsb_init(st, wk, D, a.opt.m, x0, blo, bhi, bnbd, cp);
#ifdef BORE_STAMPS
    st.lb_last = clock64();
#endif
  }
  wave_lds_sync();  // (the workspace of a problem is private to its wave / lane)

  // static shapes: the network (small ones: with its weight operands, 30 registers for shape 1)
  // lives in registers for the whole optimisation
  using Net = RegNet<(SHAPE > 0 ? SHAPE : 1), 2, BF16>;
  const typename Net::WT *thw = reinterpret_cast<const typename Net::WT *>(smem);
  Net net;
  if constexpr (SHAPE > 0 && !BF16) {
    if constexpr (Net::RT_ACT) net.set_acts(a.L);
    if constexpr (!LEAN) {
      if (!(coop && BORE_POINT_SHAPE(SHAPE))) {  // (one point per wave goes through PointNet: no matrix operands)
        net.load_fwd(thw);
        net.template load_bwd<Net::n, 1>(thw);
      }
    }
  }
  // A bfloat16 model: every evaluation -- one point per wave included -- is a 16-row pass over the bf16
  // matrix cores (arg_bf16_mfma.h): 84 matrix instructions for 32->128-128-1 where the vector-ALU chain
  // of mlp_point.h took 25 k cycles.
  static_assert(!BF16 || bore_shape_is_wide(SHAPE), "bfloat16: wide static shapes");
  using ANet = ArgBf16Net<(BF16 ? SHAPE : 3)>;
  ANet anet;
  if constexpr (BF16) anet.set_acts(a.L);
  bool done = (myp < 0);
  const long long c_init = BORE_LCLOCK();
  long long t_adv = 0, t_fg = 0, n_rounds = 0;
  // evaluations that RAN the network (nfev also counts requests served by the image shortcut): batch mode
  // counts them per problem in LDS -- one ds_add per evaluation, no register carried through the optimiser
  int *execs = cnt + 4;
  // ---- one problem per wave, static shape: the optimiser calls the evaluation (lbfgsb.h, DIRECT form) ----
  bool direct = false;
  if constexpr (SHAPE > 0) {
    direct = coop;
    if (coop && !done) {
      const int m16 = lane & 15, q4 = lane >> 4;
      auto evaluate = [&](lbfgsb::State &s, const lbfgsb::Work &w) {
        wave_lds_sync();  // (x was written by the lanes that own its components)
        const long long c1 = BORE_LCLOCK();
        // (the image shortcut of evaluate2 below for a bfloat16 network, which reads the point's bfloat16
        // image: the point evaluated last, its gradient and its value are the optimiser's own cache --
        // from the second evaluation on.  BASELINE config 5, 256 loops: restarts 111 -> 91 ms.  Not for
        // float32 networks of more than two inputs: configs 2 / 3 hardly ever repeat an image and paid
        // 1 - 2 % for the test, profiles/r3/ab_headline.txt)
        static_assert(BORE_DIM_MAX <= 64, "one input per lane");
        if (BF16 && s.nfev > 1) {
          bool other = false;
          if (lane < D)
            other = __float_as_uint(Net::rnd((float)w.x[lane])) != __float_as_uint(Net::rnd((float)w.xlast[lane]));
          if (!__any(other)) {
            s.f = s.flast;
            if (lane < D) w.g[lane] = w.glast[lane];
            wave_lds_sync();
            ++n_rounds;
            return;
          }
        }
        if (a.result && lane == 0) atomicAdd(execs + myp, 1);
        if constexpr (LEAN && !BF16) {
          if (!BORE_POINT_SHAPE(SHAPE)) {
            net.load_fwd(thw);
            net.template load_bwd<Net::n, 1>(thw);
          }
        }
        if constexpr (BF16) {
          // every row of the wave's block is the point, read from the optimiser's fp64 x (autocast)
          bf16x8_t xf[ANet::CF1];
          ANet::make_xfrag(xf, [&](int d) -> float { return d < D ? (float)w.x[d] : 0.f; });
          __builtin_amdgcn_sched_barrier(0);
          const float Tv = anet.fg(arg_bf16_images<SHAPE>(smem), xf, a.transform, a.sign);
          s.f = (double)__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(Tv)));
          if (m16 == 0) {
#pragma unroll
            for (int t = 0; t < ANet::T0; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int d = 16 * t + 4 * q4 + r;
                if (d < D) w.g[d] = (double)anet.d[0][t][r];
              }
          }
        } else if constexpr (BORE_POINT_SHAPE(SHAPE)) {
          // one point per wave, on the vector ALU (mlp_point.h: the same k-ordered fmaf chains as the
          // matrix path, 1 / 16 of its arithmetic); x straight from the optimiser's fp64 vector (Keras
          // autocast fp64 -> fp32)
          PointNet<(SHAPE > 0 ? SHAPE : 1), BF16> pnet;
          if constexpr (Net::RT_ACT) pnet.set_acts(a.L);
          const float xl = lane < D ? Net::rnd((float)w.x[lane]) : 0.f;
          const float Tv = pnet.fg(thw, xl, a.transform, a.sign);
          s.f = (double)Tv;
          if (lane < D) w.g[lane] = (double)pnet.d[0][0];
        } else {
          // every row of the wave's 16-row block evaluates the point, read straight from the
          // optimiser's fp64 x (Keras autocast fp64 -> fp32)
          float xin[Net::KC0];
#pragma unroll
          for (int kc = 0; kc < Net::KC0; ++kc) {
            const int d = 4 * kc + q4;
            xin[kc] = d < D ? Net::rnd((float)w.x[d]) : 0.f;
          }
          __builtin_amdgcn_sched_barrier(0);
          const float Tv = net.fg(thw, xin, a.transform, a.sign);
          s.f = (double)__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(Tv)));
          if (m16 == 0) {
#pragma unroll
            for (int t = 0; t < Net::L.Np[0] / 16; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int d = 16 * t + 4 * q4 + r;
                if (d < D) w.g[d] = (double)net.d[0][t][r];
              }
          }
        }
        wave_lds_sync();
        t_fg += BORE_LCLOCK() - c1;
        ++n_rounds;
      };
      const long long c0 = BORE_LCLOCK();
      if constexpr (SHAPE == 1 && !BORE_POINT_SHAPE(SHAPE) && !BF16) {
        // two inputs: the line search keeps its vectors in registers (lbfgsb.h, TWO-VARIABLE form) and
        // hands the point over in registers as well
        auto evaluate2 = [&](lbfgsb::State &s, double x0, double x1, double &g0, double &g1,
                             const double xl0, const double xl1, const double gl0, const double gl1) {
          const long long c1 = BORE_LCLOCK();
          // The network reads the point's float32 image (Keras autocast): a trial point whose image has
          // the bits of the last evaluated point's gets that point's value and gradient -- what the
          // network would return -- without the evaluation.  A line search that fails in float32 noise
          // shrinks its step below the float32 spacing of x long before it gives up: a fifth of all
          // requests (tools/fp32_image_hits.py; counted as evaluations all the same: SciPy's cache
          // compares the float64 points).
          if (__float_as_uint((float)x0) == __float_as_uint((float)xl0) &&
              __float_as_uint((float)x1) == __float_as_uint((float)xl1)) {
            s.f = s.flast;
            g0 = gl0;
            g1 = gl1;
            ++n_rounds;
            return;
          }
          // (tried and dropped: the search's base point as a second entry -- another 7 % of the requests,
          // no gain over its compare and four more registers in the loop, profiles/r3/ab_headline.txt)
          if (a.result && lane == 0) atomicAdd(execs + myp, 1);
          if constexpr (LEAN) {
            net.load_fwd(thw);
            net.template load_bwd<Net::n, 1>(thw);
          }
          static_assert(Net::KC0 == 1, "two inputs: one k-chunk");
          float xin[Net::KC0];
          xin[0] = q4 == 0 ? (float)x0 : (q4 == 1 ? (float)x1 : 0.f);  // Keras autocast fp64 -> fp32
          __builtin_amdgcn_sched_barrier(0);
          const float Tv = net.fg(thw, xin, a.transform, a.sign);
          s.f = (double)__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(Tv)));
          // lane 0 (row 0, inputs 0..3) holds d T / d x_0, d T / d x_1
          g0 = (double)__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(net.d[0][0][0])));
          g1 = (double)__uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(net.d[0][0][1])));
          t_fg += BORE_LCLOCK() - c1;
          ++n_rounds;
        };
        lbfgsb::lbfgsb_advance<SHAPE != 1>(st, wk, blo, bhi, bnbd, a.opt, cp, evaluate, evaluate2);
      } else {
        lbfgsb::lbfgsb_advance<SHAPE != 1>(st, wk, blo, bhi, bnbd, a.opt, cp, evaluate);
      }
      t_adv = BORE_LCLOCK() - c0 - t_fg;
      done = true;
    }
  }
  if constexpr (ALWAYS_COOP) {
    // (cannot happen: these kernels' launches give a wave one problem at a time -- engine_create,
    // lbfgsb_build.  If a caller ever breaks that, the problems are REPORTED, unoptimised, with status 2
    // below -- every flag is published, no host waits -- instead of being dropped silently.)
  } else {
  for (int round = 0; !direct && round < a.max_rounds; ++round) {
    int pending = 0;
    const long long c0 = BORE_LCLOCK();
    if (!done) {
      // (2-D problems -- static shape 1, the fused iteration kernel -- keep the forms without a variable per lane)
      const int rc = lbfgsb::lbfgsb_advance<SHAPE != 1>(st, wk, blo, bhi, bnbd, a.opt, cp);
      if (rc == lbfgsb::LB_NEED_FG) {
        float *row = tile + L.aoff[0] + myrow * L.lda[0];
        // Keras autocast fp64 -> fp32 (a wave's single problem: a component per lane -- cp = {lane, 64} -- else the
        // lane's own problem, every component)
        for (int d = cp.lane; d < D; d += cp.nl) row[d] = (float)wk.x[d];
        pending = 1;
      } else {
        done = true;
      }
    }
    if (!__any(pending)) break;  // every problem of this wave has terminated
    wave_lds_sync();
    const long long c1 = BORE_LCLOCK();
    if constexpr (SHAPE > 0) {
      // static shape: the 16-row block goes through the network in registers (mlp_regs.h); lane
      // s < 16 owns row s (a wave's single problem takes the DIRECT form above)
      const int m16 = lane & 15, q4 = lane >> 4;
      float xin[Net::KC0];
      if constexpr (LEAN && !BF16) {
        net.load_fwd(thw);
        net.template load_bwd<Net::n, 1>(thw);
      }
      const float *A0 = tile + L.aoff[0] + (wv * 16 + m16) * L.lda[0];
      float Tv;
      if constexpr (BF16) {
        bf16x8_t xf[ANet::CF1];
        ANet::make_xfrag(xf, [&](int d) -> float { return d < D ? A0[d] : 0.f; });
        __builtin_amdgcn_sched_barrier(0);
        Tv = anet.fg(arg_bf16_images<SHAPE>(smem), xf, a.transform, a.sign);
        Net::template store_rows_f32<0>(anet.d[0], tile + BORE_BATCH_MAX * L.lda[0], wv);
      } else {
#pragma unroll
        for (int kc = 0; kc < Net::KC0; ++kc) xin[kc] = Net::rnd(A0[4 * kc + q4]);
        __builtin_amdgcn_sched_barrier(0);
        Tv = net.fg(thw, xin, a.transform, a.sign);
        Net::template store_rows_f32<0>(net.d[0], tile + BORE_BATCH_MAX * L.lda[0], wv);
      }
      wave_lds_sync();
      if (pending) {  // lane s < 16 owns row s: its value is already in this lane
        st.f = (double)Tv;
        const float *g = tile + BORE_BATCH_MAX * L.lda[0] + myrow * L.lda[0];
        for (int d = cp.lane; d < D; d += cp.nl) wk.g[d] = (double)g[d];
      }
    } else {
      fg_rowblock(L, n_lay, th, tile, wv, a.transform, a.sign, vals);
      if (pending) {
        st.f = (double)vals[myrow];
        const float *g = tile + L.doff[0] + myrow * L.lda[0];
        for (int d = cp.lane; d < D; d += cp.nl) wk.g[d] = (double)g[d];
      }
    }
    wave_lds_sync();
    t_adv += c1 - c0;
    t_fg += BORE_LCLOCK() - c1;
    ++n_rounds;
  }
  }
  (void)t_adv; (void)t_fg; (void)n_rounds; (void)c_enter; (void)c_begun; (void)c_staged; (void)c_init;
#ifdef BORE_STAMPS
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) {
    g_lstamps[0] = t_adv; g_lstamps[1] = t_fg; g_lstamps[2] = n_rounds; g_lstamps[3] = st.nit;
    g_lstamps[4] = c_begun - c_enter; g_lstamps[5] = c_staged - c_begun; g_lstamps[6] = c_init - c_staged;
    g_lstamps[7] = BORE_LCLOCK() - c_enter;
  }
  if (coop && lane == 0 && 4 * blockIdx.x + wv < LB_PP_MAX) {  // (one writer per row: plain adds)
    unsigned long long *pp = lbfgsb::g_lb_pp[4 * blockIdx.x + wv];
    for (int i = 0; i < 64; ++i) {
      pp[i] += lbfgsb::g_lb_lds[wv & 15][i];
      lbfgsb::g_lb_lds[wv & 15][i] = 0;
    }
    pp[13] += (unsigned long long)t_adv;
    pp[14] += (unsigned long long)t_fg;
    pp[32 + 13] += (unsigned long long)n_rounds;
  }
  if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)   // (the r2 tool's view: cycles 0..7, calls 8..15)
    for (int i = 0; i < 8; ++i) {
      lbfgsb::g_lb_phase[i] = (long long)lbfgsb::g_lb_pp[0][i];
      lbfgsb::g_lb_phase[8 + i] = (long long)lbfgsb::g_lb_pp[0][32 + i];
    }
#endif

  // (the result addresses below depend on the problem number alone: an opaque copy keeps them from
  // being formed before the optimisation and held -- spilled -- across it)
  int myq = myp;
  BORE_OPAQUE_TID(myq);
  if (myp >= 0 && st.stage != lbfgsb::S_FINISHED) {  // round cap hit (cannot happen with a sane cap)
    st.status = 2;
    st.task = lbfgsb::T_STOP;
    st.msg = lbfgsb::M_MAXFUN;
  }
  // (round 6, the 32-input shape: the shared problem's point and gradient go out a component per lane -- one lane
  // walking 32 components was 64 dependent LDS round trips at the end of every problem of BASELINE config 5, whose
  // restarts last one iteration: 20.4 -> 19.7 ms.  Not for the narrower shapes: 16->64-64-64-1 14.35 -> 15.0 ms with it,
  // profiles/r6/ab_log.txt)
  constexpr bool OUT_PER_LANE = SHAPE == 4;
  if (OUT_PER_LANE && myp >= 0 && coop) {
    const long long q = model * a.R + p0 + myq;
    for (int d = lane; d < D; d += 64) {
      a.x[q * D + d] = wk.x[d];
      a.jac[q * D + d] = wk.g[d];
    }
  }
  if (myp >= 0 && (!coop || lane == 0)) {  // (coop: one lane reports the shared problem)
    const long long q = model * a.R + p0 + myq;
    if (!(OUT_PER_LANE && coop))
      for (int d = 0; d < D; ++d) {
        a.x[q * D + d] = wk.x[d];
        a.jac[q * D + d] = wk.g[d];
      }
    a.fun[q] = st.f;
    int *inf = a.info + q * 5;
    inf[0] = st.nit; inf[1] = st.nfev; inf[2] = st.status; inf[3] = st.task; inf[4] = st.msg;
  }
  if (!a.result) continue;  // (not batch mode: one pass)
  // ---- batch mode: the loop's pick (bore_select_best's rule), published by whichever of its
  // waves finishes the loop's last problem, while other loops of the launch are still optimising ----
  if (lane == 0) {
    double *r = res + myq * (D + 3);
    r[0] = st.f;
    r[1] = (double)(st.status + 8 * execs[myq]);  // (status 0..2 | executed evaluations: unpacked below)
    r[2] = (double)st.nfev;
    for (int d = 0; d < D; ++d) r[3 + d] = wk.x[d];
  }
  wave_lds_sync();
  int last = 0;
  if (lane == 0) last = atomicAdd(cnt, 1) == np - 1;
  last = __shfl(last, 0, 64);
  if (!last) continue;
  wave_lds_sync();
  const int it_done = it_now >= 0 ? it_now : uniform_i32(a.its[model]);
  const int N = a.n_init + it_done;
  const double *Xs = a.dedup ? a.X_seen + lid * a.cap * D : nullptr;
  int best = -1;
  double best_fun = 0.0, nfev_sum = 0.0, nfev_max = 0.0, exec_sum = 0.0;
  for (int r = 0; r < np; ++r) {
    const double *rr = res + r * (D + 3);
    nfev_sum += rr[2];
    nfev_max = fmax(nfev_max, rr[2]);
    const int status = (int)rr[1] & 7;
    exec_sum += (double)((int)rr[1] >> 3);
    if (status != 0 && status != 1) continue;  // res.success or res.status == 1
    bool dup = false;
    if (a.dedup) {  // any(np.allclose(x_prev, x) for x_prev in record): rows dealt to lanes
      bool mine = false;
      for (int i = lane; i < N && !mine; i += 64) {
        bool close = true;
        for (int d = 0; d < D && close; ++d) {
          const double b = rr[3 + d];
          close = fabs(Xs[(long long)i * D + d] - b) <= 1e-8 + 1e-5 * fabs(b);
        }
        mine = close;
      }
      dup = __any(mine) != 0;
    }
    if (dup) continue;
    if (best < 0 || rr[0] < best_fun) {
      best = r;
      best_fun = rr[0];
    }
  }
  if (lane == 0) {
    double *out = a.result + lid * (D + 8);
    auto put = [&](int i, double v) {
      if constexpr (PUBLISH == 1)
        __hip_atomic_store(reinterpret_cast<long long *>(out) + i, __double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
      else
        out[i] = v;
    };
    for (int d = 0; d < D; ++d) put(d, best >= 0 ? res[best * (D + 3) + 3 + d] : 0.0);
    put(D, (double)best);
    put(D + 1, nfev_sum);
    put(D + 2, nfev_max);
    if (a.stamps) {  // device-clock ticks per phase of this loop-iteration (fused kernel)
      const long long *s = a.stamps + lid * 4;
      const long long now = wall_clock64();
      put(D + 3, (double)(s[1] - s[0]));
      put(D + 4, (double)(s[2] - s[1]));
      put(D + 5, (double)(s[3] - s[2]));
      put(D + 6, (double)(now - s[3]));
    } else {
      for (int i = 3; i < 7; ++i) put(D + i, 0.0);
    }
    put(D + 7, exec_sum);  // evaluations that ran the network (<= nfev_sum: the image shortcut serves the rest)
    if constexpr (PUBLISH == 1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (this lane's stores -- the pick above among them -- are acknowledged)
      // (an exchange, not a store: the wave waits for it to have happened.  As a plain write-through store the flag
      // reached the host late and in bursts -- the host's result -> row turn-around went from 1.5 to 17 us per
      // loop-iteration, all of what the cheaper hand-overs had saved; a buffer_wbl2 after the store changed nothing,
      // profiles/r5/ab_log.txt)
      // Ordering: the pick's stores are write-through and ACKNOWLEDGED (vmcnt(0) above) before the flag's operation is
      // issued -- what gfx942 / gfx950 guarantee for system-scope (sc0 sc1) stores to fine-grained host memory, not
      // a formal release.  Where the link has no native atomics (a.flag_by_store) the flag is a release store behind a
      // system-scope fence instead: slower (the XCD's L2 is written back), correct everywhere.
      if (a.flag_by_store) {
        __threadfence_system();
        __hip_atomic_store(a.flag + lid, it_done + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        (void)__hip_atomic_exchange(a.flag + lid, it_done + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    } else {
      __threadfence_system();
      __hip_atomic_store(a.flag + lid, it_done + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  }  // passes
  if (use_big) {  // the workgroup's last wave to get here frees the pool slot
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && atomicAdd(qnext + 2, 1) == NW - 1)
      __hip_atomic_store(a.big_locks + big_slot, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int SHAPE, bool BF16 = false>
__global__ __launch_bounds__(BORE_THREADS) void lbfgsb_kernel(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, BF16>(a, blockIdx.x, blockIdx.y);
}

// Two workgroups per CU (256 registers per lane, operands re-requested per evaluation): for launches
// with many more workgroups than the device has CUs.
// (3 / 4 workgroups per CU measured slower in round 3: spills)
template <int SHAPE, bool BF16 = false>
__global__ __launch_bounds__(BORE_THREADS, 2) void lbfgsb_kernel_occ2(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, BF16, true, true>(a, blockIdx.x, blockIdx.y);
}
#ifdef BORE_RU_PROBE_WIDE  // (register-pressure probes: the wide restart kernels at three waves per SIMD; never launched)
template <int SHAPE, bool BF16>
__global__ __launch_bounds__(BORE_THREADS, 3) void lbfgsb_kernel_occ3_probe(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, BF16, true, true>(a, blockIdx.x, blockIdx.y);
}
template <int SHAPE, bool BF16>
__global__ __launch_bounds__(3 * BORE_THREADS) void lbfgsb_kernel_w12_probe(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, BF16, true, true>(a, blockIdx.x, blockIdx.y);
}
template __global__ void lbfgsb_kernel_occ3_probe<2, false>(const LbfgsbArgs a);
template __global__ void lbfgsb_kernel_w12_probe<3, false>(const LbfgsbArgs a);
template __global__ void lbfgsb_kernel_w12_probe<4, true>(const LbfgsbArgs a);
#endif
#ifdef BORE_RU_PROBE  // (register-pressure probe of the fused kernel's restart phase on its own; never launched)
template <int SHAPE>
__global__ __launch_bounds__(BORE_THREADS, BORE_RU_PROBE) void lbfgsb_kernel_probe(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, false, true, true, false>(a, blockIdx.x, 0, 0);
}
template __global__ void lbfgsb_kernel_probe<1>(const LbfgsbArgs a);
#endif
// The same for the wide shapes, whose weights take too much LDS for two workgroups: ONE workgroup of
// up to eight waves (two per SIMD), one problem per wave, the weights staged once for all of them.
template <int SHAPE, bool BF16 = false>
__global__ __launch_bounds__(2 * BORE_THREADS) void lbfgsb_kernel_w8(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, BF16, true, true>(a, blockIdx.x, blockIdx.y);
}

// Twelve waves (three per SIMD, 168 registers) for the narrow shapes, whose problems are small enough for twelve
// workspaces beside the weights in ONE workgroup's LDS: the restart kernels run chains of dependent LDS round trips
// and float64 operations -- a call of cauchy takes the same 13 - 15 k cycles alone on the device and at two waves
// per SIMD on a full one (profiles/r5/ab_log.txt) -- so a third wave per SIMD is throughput for free.
template <int SHAPE, bool BF16 = false>
__global__ __launch_bounds__(3 * BORE_THREADS) void lbfgsb_kernel_w12(const LbfgsbArgs a) {
  lbfgsb_body<SHAPE, BF16, true, true>(a, blockIdx.x, blockIdx.y);
}

// (Sixteen waves -- four per SIMD, 128 registers -- fit since round 6 stores the optimiser's two 2m x 2m matrices as the
// two triangles of one block, lbfgsb.h: 90 registers spilled, 8.45 ms per million evaluation requests of config 2
// against 7.93 for twelve waves on the same inputs, profiles/r6/ab_log.txt.  Not kept.)

static int lbfgsb_build(const bore_mlp_desc *desc, int n_models, const float *theta,
                        int transform, int negate, const double *x0, int num_starts,
                        const double *lb, const double *ub, const bore_lbfgsb_opts *opts, double *x,
                        double *fun, double *jac, int32_t *info, LbfgsbArgs &a, size_t &lds_floats,
                        int &flavour_out, int &blocks_out, int *waves_out = nullptr) {
  if (!desc || !theta || !x0 || !lb || !ub || !opts || !x || !fun || !jac || !info)
    return fail(BORE_E_INVALID, "lbfgsb_minimize: null pointer");
  if (n_models < 1) return fail(BORE_E_INVALID, "n_models must be >= 1 (got %d)", n_models);
  const int D = desc->input_dim;
  if (D < 1 || D > BORE_DIM_MAX)
    return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: input_dim must be 1..%d", BORE_DIM_MAX);
  if (num_starts < 1) return fail(BORE_E_INVALID, "lbfgsb_minimize: num_starts < 1");
  if (transform < BORE_T_IDENTITY || transform > BORE_T_EXP)
    return fail(BORE_E_INVALID, "lbfgsb_minimize: unknown transform %d", transform);
  if (opts->maxcor < 1 || opts->maxcor > 32)
    return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: maxcor must be 1..32");
  if (opts->maxls < 1) return fail(BORE_E_INVALID, "lbfgsb_minimize: maxls must be positive.");
  if (opts->maxiter < 0 || opts->maxfun < 0 || opts->ftol < 0)
    return fail(BORE_E_INVALID, "lbfgsb_minimize: negative limit/tolerance");
  for (int d = 0; d < D; ++d) {
    const bool lo = std::isfinite(lb[d]), up = std::isfinite(ub[d]);
    if (lo && up && lb[d] > ub[d])
      return fail(BORE_E_INVALID,
                  "LBFGSB - one of the lower bounds is greater than an upper bound.");
    a.nbd[d] = lo && up ? 2 : lo ? 1 : up ? 3 : 0;
    a.box.lo[d] = lo ? lb[d] : 0.0;
    a.box.hi[d] = up ? ub[d] : 0.0;
  }
  const int m = opts->maxcor;
  a.opt.m = m;
  a.opt.factr = opts->ftol / 2.220446049250313e-16;
  a.opt.pgtol = opts->gtol;
  a.opt.maxiter = opts->maxiter;
  a.opt.maxfun = opts->maxfun;
  a.opt.maxls = opts->maxls;
  // per-problem LDS block: fp64 workspace | int workspace (8-byte aligned); the scalar
  // State lives in registers
  const size_t state_f = 0;
  size_t dw_f = 2 * (size_t)lbfgsb::dwork_size(D, m);
  const size_t iw_f = ((size_t)lbfgsb::iwork_size(D) + 3) & ~(size_t)3;
  a.o_state = 0;
  a.o_dw = (int)state_f;
  a.o_iw = (int)(state_f + dw_f);
  a.prob_floats = (int)(state_f + dw_f + iw_f);
  a.big = nullptr; a.big_locks = nullptr; a.big_wave = 0;
  // largest number of problems per workgroup whose state fits beside theta and the tile
  // (tile rows: one 16-row block per wave that has a problem)
  int PB = num_starts < BORE_BATCH_MAX ? num_starts : BORE_BATCH_MAX;
  // One problem per WAVE (all 64 lanes on it, lbfgsb::Coop) finishes a problem ~1.6x sooner than one
  // problem per lane and uses every lane: 4 problems per workgroup, grid = models x ceil(R / 4).
  // (round 2: measured on BASELINE configs 2 / 3 with 256 loops -- 16 384 and 65 536 workgroups --
  // the one-problem-per-wave mapping is 2.0x / 1.5x faster than 64 problems per workgroup there as
  // well: lanes that each run their own state machine diverge; profiles/r2/lbfgsb_mapping.txt.  The
  // lane-per-problem mapping is kept for grids beyond 4 M workgroups and for BORE_LBFGSB_COOP_GRID.)
  const long long coop_grid_max =
      getenv("BORE_LBFGSB_COOP_GRID") ? atoll(getenv("BORE_LBFGSB_COOP_GRID")) : (1LL << 22);
  if (PB > 4 && !g_batch && (long long)n_models * ((num_starts + 3) / 4) <= coop_grid_max) PB = 4;
  const int flavour = bore_acq_flavour(desc, true);
  a.d_in = D;
  // More workgroups than CUs, static shape: ONE workgroup per CU with a problem per wave and as many waves as
  // workspaces fit beside the weights, up to the kernel's launch bound -- 8 for the wide shapes (lbfgsb_kernel_w8),
  // 12 for the narrow ones (lbfgsb_kernel_w12, the same body at 168 registers).  The restart kernels are chains of
  // dependent LDS round trips and float64 operations: waves in flight are throughput -- while the registers last: ten
  // waves of 16->64-64-64-1 at 168 registers (154 spilled) measured no faster than eight at 256, sixteen of 6->32-32-1
  // at 128 (90 spilled) slower than twelve (profiles/r6/ab_log.txt).  BORE_LBFGSB_WAVES = 4 keeps the four-wave
  // kernels, = 8 / 12 forces the many-wave kernel with at most that many waves (tests, A/B).
  const int waves_env = getenv("BORE_LBFGSB_WAVES") ? atoi(getenv("BORE_LBFGSB_WAVES")) : -1;
  const bool bf16_model = desc->compute == BORE_COMPUTE_BF16;
  int wmax = 4;  // waves (= problems at a time) per workgroup of this launch's kernel
  if (waves_out && !g_batch && PB == 4) {
    int bound = 4;
    if ((flavour == 3 || (flavour == 4 && bf16_model)) && num_starts >= 8) bound = 8;
    if ((flavour == 2 || flavour == 5) && !bf16_model && num_starts >= 12) bound = 12;
    if (waves_env >= 4) wmax = bound < waves_env ? bound : waves_env;
    else if ((long long)n_models * ((num_starts + 3) / 4) > device_cus()) wmax = bound;
    if (wmax > num_starts) wmax = num_starts;
    if (wmax < 8) wmax = 4;
    PB = wmax;
  }
  const bool w8 = wmax > 4 && (flavour == 3 || flavour == 4);
  const int shape = flavour > 0 ? flavour : 0;  // constexpr-layout kernels assume a 64-row tile
  size_t off = 0;
  for (;; --PB) {
    if (PB < 1) return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: one problem's state does not fit in LDS");
    // static flavours keep the row-blocks in registers; their tile region only stages the
    // points (A_0) and gradients (D_0) of waves that run several problems: 2 x 64 rows
    const int rows = shape ? 16 : 16 * (PB < 4 ? PB : 4);
    if (bore_make_layout(desc, 2, rows, &a.L)) return fail(BORE_E_INVALID, "bad bore_mlp_desc");
    // (a bfloat16 image holds 2-byte elements at the float image's indices: half its bytes.  Round 2
    // reserved the float size, which left BASELINE config 5 -- 128-128-1 -- room for three problems per
    // workgroup and kept it out of the eight-wave kernel; with the true size six fit.)
    off = desc->compute == BORE_COMPUTE_BF16 ? ((size_t)a.L.P_lds + 1) / 2 : (size_t)a.L.P_lds;
    // (round 4: wide static shapes in bfloat16 hold the fragment-order images of arg_bf16_mfma.h -- forward
    // and backward operands of the bf16 matrix instructions, about twice the plain bfloat16 image)
    if (desc->compute == BORE_COMPUTE_BF16 && bore_shape_is_wide(flavour))
      off = flavour == 3 ? ArgBf16Plan<3>::floats : ArgBf16Plan<4>::floats;
    off = (off + 3) & ~(size_t)3;
    a.o_tile = (int)off;
    // (one problem per wave -- up to 4 problems, 8 in the eight-wave kernel, 16 in batch mode -- reads
    // its point straight from the optimiser's vector: no staging region at all)
    const bool per_wave = PB <= wmax || (g_batch && PB <= 16);
    off += shape ? (per_wave ? 0 : 2 * (size_t)BORE_BATCH_MAX * a.L.lda[0]) : (size_t)a.L.tile_floats;
    a.o_vals = (int)off; off += BORE_BATCH_MAX;
    off = (off + 3) & ~(size_t)3;  // 16-byte boundary for the fp64 regions
    a.o_box = (int)off; off += 4 * (size_t)D + (((size_t)D + 3) & ~(size_t)3) + (D & 1 ? 2 : 0);
    off = (off + 3) & ~(size_t)3;
    a.o_prob = (int)off; off += (size_t)a.prob_floats * PB;
    off = (off + 3) & ~(size_t)3;
    a.o_res = (int)off;  // batch mode: [max(R, 4)][D + 3] fp64 + the finished-problem counter
    // (+ the finished-problem counter and the per-problem counts of executed evaluations)
    off += g_batch ? 2 * (size_t)(num_starts > 4 ? num_starts : 4) * ((size_t)D + 3) + 4 + 16 : 0;
    a.o_queue = (int)off; off += 4;  // (the queue's next-problem counter)
    a.total = (int)off;
    off = (off + 3) & ~(size_t)3;
    a.o_layout = (int)off; off += BORE_LAYOUT_FLOATS;
    if (off * 4 <= BORE_LDS_BYTES) break;
  }
  // The eight-wave kernel with fewer than eight problems beside the weights (32->128-128-1: six): with the
  // optimiser's two 2m x 2m matrices in a slot of the device pool instead of in LDS eight fit (the matrices
  // are touched by the subspace minimisation only -- once in a hundred evaluations of BASELINE config 5).
  // BORE_LBFGSB_BIG = 0 / 1 forces the choice (tests, A/B).
  if (w8 && !g_batch && flavour == 4) {  // (lbfgsb_body: BIG_OK)
    const int forced = getenv("BORE_LBFGSB_BIG") ? atoi(getenv("BORE_LBFGSB_BIG")) : -1;
    const size_t small_f = 2 * (size_t)lbfgsb::dwork_size(D, m, true) + iw_f;
    const size_t fixed = off - (size_t)a.prob_floats * PB;   // everything but the problems (floats)
    int PB2 = (int)((BORE_LDS_BYTES / 4 - fixed) / small_f);
    if (PB2 > 8) PB2 = 8;
    if (PB2 >= 1 && (forced < 0 ? PB2 > PB : forced != 0)) {
      BigPool pool;
      const int rcp = big_pool(8 * (size_t)lbfgsb::big_size(m), &pool);
      if (rcp) return rcp;
      a.big = pool.buf; a.big_locks = pool.locks; a.big_wave = (int)(pool.per_slot / 8);
      // re-carve with the smaller problem blocks (same order of regions as in the loop above)
      PB = PB2;
      dw_f = 2 * (size_t)lbfgsb::dwork_size(D, m, true);
      a.o_iw = (int)(state_f + dw_f);
      a.prob_floats = (int)(state_f + dw_f + iw_f);
      off = (size_t)a.o_prob + (size_t)a.prob_floats * PB;
      off = (off + 3) & ~(size_t)3;
      a.o_res = (int)off;
      a.o_queue = (int)off; off += 4;
      a.total = (int)off;
      off = (off + 3) & ~(size_t)3;
      a.o_layout = (int)off; off += BORE_LAYOUT_FLOATS;
      if (off * 4 > BORE_LDS_BYTES) return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: LDS carve with the pooled matrices");
    }
  }
  // One problem per wave and many restarts per model: a few workgroups per model, each drawing its
  // share of the restarts from a queue (lbfgsb_body).  About four workgroups per CU over the launch;
  // BORE_LBFGSB_QUEUE = 0 keeps one workgroup per PB restarts, = n > 0 asks for n workgroups in all
  // (A/B, tests).
  const int slots = PB;  // workspaces = waves with a problem
  a.queue = 0;
  const long long q_env = getenv("BORE_LBFGSB_QUEUE") ? atoll(getenv("BORE_LBFGSB_QUEUE")) : -1;
  if (!g_batch && PB <= wmax && num_starts > PB && q_env != 0) {
    // (the 32-32-1 flavour stages 6 KB of weights and runs two workgroups per CU: finer shares -- 32 per CU
    // over the launch -- measured best there, profiles/r4/ab_log.txt; the wide flavours stage 40 - 50 KB)
    // (round 6, tools/ab_restarts.py on fixed inputs, profiles/r6/ab_queue_shares.txt: 4 per CU is the best or within
    // 0.2 % of it for the twelve- / eight-wave float32 kernels; 32->128-128-1 in bfloat16 gains 4 % from 8 per CU)
    const long long want = q_env > 0 ? q_env
                                     : ((flavour == 2 || flavour == 5) && wmax == 4 ? 32LL : (flavour == 4 ? 8LL : 4LL)) * device_cus();
    long long per_model = (want + n_models - 1) / n_models;
    const long long most = (num_starts + slots - 1) / slots;
    if (per_model > most) per_model = most;
    if (per_model < 1) per_model = 1;
    PB = (int)((num_starts + per_model - 1) / per_model);
    a.queue = PB > slots;
  }
  a.PB = PB;
  if (waves_out) *waves_out = wmax > 4 && slots > 4 ? slots : 4;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "lbfgsb_minimize: the last Dense layer must have 1 unit");
  a.theta = theta; a.x0 = x0; a.x = x; a.fun = fun; a.jac = jac; a.info = info;
  a.R = num_starts; a.transform = transform; a.sign = negate ? -1.f : 1.f;
  a.ids = a.its = nullptr; a.n_init = a.dedup = 0; a.cap = 0;
  a.X_seen = nullptr; a.result = nullptr; a.flag = nullptr; a.stamps = nullptr;
  a.flag_by_store = device_host_atomics() ? 0 : 1;
  if (g_batch) {
    if (num_starts > 16 || PB < num_starts)
      return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: batch mode needs num_starts <= 16 in one workgroup");
    if (!g_batch->ids || !g_batch->its || !g_batch->result || !g_batch->flag ||
        (g_batch->deduplicate && !g_batch->X_seen))
      return fail(BORE_E_INVALID, "lbfgsb_minimize: incomplete bore_batch");
    a.ids = g_batch->ids; a.its = g_batch->its; a.n_init = g_batch->n_init;
    a.dedup = g_batch->deduplicate; a.cap = g_batch->cap; a.X_seen = g_batch->X_seen;
    a.result = g_batch->result; a.flag = g_batch->flag;
  }
  long long cap = (long long)opts->maxfun + opts->maxls + 64;
  a.max_rounds = (int)(cap > (1 << 24) ? (1 << 24) : cap);
  const int blocks = (num_starts + PB - 1) / PB;
  if (blocks > 65535) return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: too many restarts");
  lds_floats = off;
  flavour_out = flavour;
  blocks_out = blocks;
  return 0;
}

extern "C" int bore_lbfgsb_minimize(const bore_mlp_desc *desc, int n_models, const float *theta,
                                    int transform, int negate, const double *x0, int num_starts,
                                    const double *lb, const double *ub,
                                    const bore_lbfgsb_opts *opts, double *x, double *fun,
                                    double *jac, int32_t *info, void *stream) {
  LbfgsbArgs a;
  size_t off = 0;
  int flavour = 0, blocks = 0, waves = 4;
  int rc = lbfgsb_build(desc, n_models, theta, transform, negate, x0, num_starts, lb, ub, opts, x, fun,
                        jac, info, a, off, flavour, blocks, &waves);
  if (rc) return rc;
  if (waves > 4) {  // (wide static shape, more workgroups than CUs: see lbfgsb_build)
    const bool bf = desc->compute == BORE_COMPUTE_BF16;
#define BORE_LAUNCH_W8(K)                                                                          \
  {                                                                                                \
    rc = allow_lds((K), off * 4);                                                                  \
    if (rc) return rc;                                                                             \
    hipLaunchKernelGGL((K), dim3(n_models, blocks), dim3(64 * waves), off * 4, (hipStream_t)stream, a); \
  }
    if (!bore_flavour_built(flavour)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
    if (!(flavour == 3 || (flavour == 4 && bf) || ((flavour == 2 || flavour == 5) && !bf)))
      return fail(BORE_E_UNSUPPORTED, "lbfgsb_minimize: no many-wave kernel for this flavour");
    // (the kernel whose launch bound covers the waves: a launch of fewer waves than the bound is fine)
#if BORE_ON_2
    if (flavour == 2) BORE_LAUNCH_W8((lbfgsb_kernel_w12<2, false>))
#endif
#if BORE_ON_5
    if (flavour == 5) BORE_LAUNCH_W8((lbfgsb_kernel_w12<5, false>))
#endif
#if BORE_ON_3
    if (flavour == 3 && !bf) BORE_LAUNCH_W8((lbfgsb_kernel_w8<3, false>))
    if (flavour == 3 && bf) BORE_LAUNCH_W8((lbfgsb_kernel_w8<3, true>))
#endif
#if BORE_ON_4
    if (flavour == 4 && bf) BORE_LAUNCH_W8((lbfgsb_kernel_w8<4, true>))
#endif
#undef BORE_LAUNCH_W8
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (desc->compute == BORE_COMPUTE_BF16) {
    if (!bore_shape_is_wide(flavour)) return fail(BORE_E_UNSUPPORTED, kBf16Shapes);
    if (!bore_flavour_built(flavour)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#if BORE_ON_3
    if (flavour == 3) {
      rc = allow_lds((lbfgsb_kernel<3, true>), off * 4);
      if (rc) return rc;
      hipLaunchKernelGGL((lbfgsb_kernel<3, true>), dim3(n_models, blocks), dim3(BORE_THREADS),
                         off * 4, (hipStream_t)stream, a);
    }
#endif
#if BORE_ON_4
    if (flavour == 4) {
      rc = allow_lds((lbfgsb_kernel<4, true>), off * 4);
      if (rc) return rc;
      hipLaunchKernelGGL((lbfgsb_kernel<4, true>), dim3(n_models, blocks), dim3(BORE_THREADS),
                         off * 4, (hipStream_t)stream, a);
    }
#endif
    HIP_TRY(hipGetLastError());
    return 0;
  }
  {
    // More workgroups than CUs (many loops x many restarts): the 32-32-1 flavour, whose workgroup
    // needs 48 KB of LDS, runs two workgroups per CU (256 registers per lane, operands re-requested
    // per evaluation, 132 B of scratch) -- restarts of BASELINE config 2 with 256 loops 78.7 -> 43.1 ms,
    // same bits.  A single loop's launch (64 workgroups) keeps the one-per-CU kernel: it finishes a
    // problem sooner.  BORE_LBFGSB_OCC2 = 0 / 1 forces either (tests, measurements).
    const int forced = getenv("BORE_LBFGSB_OCC2") ? atoi(getenv("BORE_LBFGSB_OCC2")) : -1;
    const bool many = (long long)n_models * blocks > device_cus();
    (void)forced; (void)many;
    // (one problem per wave only: the kernel does not carry the lane-per-problem loop)
    // (launches with many workgroups of a narrow shape take the twelve-wave kernel above first: lbfgsb_build)
#define BORE_LAUNCH_OCC2(S)                                                                                                 \
    if (flavour == (S) && (a.PB <= 4 || a.queue) && off * 4 <= BORE_LDS_BYTES / 2 && (forced < 0 ? many : forced != 0)) { \
      rc = allow_lds(lbfgsb_kernel_occ2<S>, off * 4);                                                                       \
      if (rc) return rc;                                                                                                    \
      hipLaunchKernelGGL(lbfgsb_kernel_occ2<S>, dim3(n_models, blocks), dim3(BORE_THREADS), off * 4,                       \
                         (hipStream_t)stream, a);                                                                           \
      HIP_TRY(hipGetLastError());                                                                                           \
      return 0;                                                                                                             \
    }
#if BORE_ON_2
    BORE_LAUNCH_OCC2(2)
#endif
#if BORE_ON_5
    BORE_LAUNCH_OCC2(5)  // (the plugin's default network: 5 restarts per loop, many loops)
#endif
#undef BORE_LAUNCH_OCC2
  }
  if (!bore_flavour_built(flavour)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#define BORE_LAUNCH_LBFGSB(S)                                                                  \
  case S:                                                                                      \
    rc = allow_lds(lbfgsb_kernel<S>, off * 4);                                                 \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL(lbfgsb_kernel<S>, dim3(n_models, blocks), dim3(BORE_THREADS), off * 4,  \
                       (hipStream_t)stream, a);                                                \
    break;
  switch (flavour) {
#if BORE_ON_1
    BORE_LAUNCH_LBFGSB(1)
#endif
#if BORE_ON_2
    BORE_LAUNCH_LBFGSB(2)
#endif
#if BORE_ON_3
    BORE_LAUNCH_LBFGSB(3)
#endif
#if BORE_ON_4
    BORE_LAUNCH_LBFGSB(4)
#endif
#if BORE_ON_5
    BORE_LAUNCH_LBFGSB(5)
#endif
#if BORE_ON_N1
    BORE_LAUNCH_LBFGSB(-1)
#endif
#if BORE_ON_N2
    BORE_LAUNCH_LBFGSB(-2)
#endif
#if BORE_ON_N3
    BORE_LAUNCH_LBFGSB(-3)
#endif
#if BORE_ON_N4
    BORE_LAUNCH_LBFGSB(-4)
#endif
    default:
#if !BORE_ON_0
      return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#endif
#if BORE_ON_0
    BORE_LAUNCH_LBFGSB(0)
#endif
  }
#undef BORE_LAUNCH_LBFGSB
  HIP_TRY(hipGetLastError());
  return 0;
}

// ---------------------------------------------------------------------------
// device-resident observation store (Record.append / load_regression_data) and the
// selection of argmax() with the plugin's duplicate filter
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BORE_THREADS) void append_kernel(int D, double *X_seen, double *y_seen,
                                                              long long n_seen, long long cap,
                                                              const double *x_new,
                                                              const double *y_new, float *X32,
                                                              double *y_dense, const int *ids,
                                                              const int *its, int n_init) {
  const long long model = blockIdx.x;
  if (ids) {  // batch mode: row N - 1 of loop ids[slot] into the three cap-strided buffers
    const int it = its[model];
    if (it <= 0) return;  // nothing new before the first iteration
    const long long lid = ids[model], row = n_init + it - 1;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
      const double v = x_new[model * D + d];
      X_seen[(lid * cap + row) * D + d] = v;
      X32[(lid * cap + row) * D + d] = (float)v;
    }
    if (threadIdx.x == 0) y_seen[lid * cap + row] = y_new[model];
    return;
  }
  double *Xs = X_seen + model * cap * D, *ys = y_seen + model * cap;
  if (x_new) {  // row n_seen of the store; the same threads read it back below
    for (int d = threadIdx.x; d < D; d += blockDim.x) Xs[n_seen * D + d] = x_new[model * D + d];
    if (threadIdx.x == 0) ys[n_seen] = y_new[model];
    __syncthreads();
  }
  const long long n = n_seen + (x_new ? 1 : 0);
  float *Xo = X32 + model * n * D;
  double *yo = y_dense + model * n;
  for (long long i = threadIdx.x; i < n * D; i += blockDim.x) Xo[i] = (float)Xs[i];
  for (long long i = threadIdx.x; i < n; i += blockDim.x) yo[i] = ys[i];
}

extern "C" int bore_append_observations(int n_models, int D, double *X_seen, double *y_seen,
                                        int64_t n_seen, int64_t cap, const double *x_new,
                                        const double *y_new, float *X32, double *y_dense,
                                        void *stream) {
  if (g_batch) {
    if (n_models < 1 || D < 1 || !X_seen || !y_seen || !X32 || !x_new || !y_new)
      return fail(BORE_E_INVALID, "append_observations: bad argument");
    hipLaunchKernelGGL(append_kernel, dim3(n_models), dim3(64), 0, (hipStream_t)stream, D, X_seen,
                       y_seen, 0LL, (long long)g_batch->cap, x_new, y_new, X32, y_dense,
                       g_batch->ids, g_batch->its, g_batch->n_init);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (n_models < 1 || D < 1 || !X_seen || !y_seen || !X32 || !y_dense)
    return fail(BORE_E_INVALID, "append_observations: bad argument");
  if ((x_new == nullptr) != (y_new == nullptr))
    return fail(BORE_E_INVALID, "append_observations: x_new and y_new go together");
  if (n_seen < 0 || n_seen + (x_new ? 1 : 0) > cap || n_seen + (x_new ? 1 : 0) < 1)
    return fail(BORE_E_INVALID, "append_observations: %lld rows (+%d) do not fit cap %lld",
                (long long)n_seen, x_new ? 1 : 0, (long long)cap);
  hipLaunchKernelGGL(append_kernel, dim3(n_models), dim3(BORE_THREADS), 0, (hipStream_t)stream, D,
                     X_seen, y_seen, (long long)n_seen, (long long)cap, x_new, y_new, X32, y_dense,
                     (const int *)nullptr, (const int *)nullptr, 0);
  HIP_TRY(hipGetLastError());
  return 0;
}

// fun as a key that orders like the doubles do (NaN cannot occur: sigmoid/exp/identity of a
// finite logit); the restart number below it breaks ties towards the earliest
__device__ __forceinline__ unsigned long long orderable64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ULL);
}

__global__ __launch_bounds__(BORE_THREADS) void select_kernel(int R, int D, const double *x,
                                                              const double *fun, const int *info,
                                                              const double *X_seen,
                                                              long long n_seen, long long cap,
                                                              double rtol, double atol,
                                                              double *x_best, int *best) {
  __shared__ unsigned long long s_key;
  __shared__ int s_idx;
  const long long model = blockIdx.x;
  if (threadIdx.x == 0) {
    s_key = ~0ULL;
    s_idx = 0x7fffffff;
  }
  __syncthreads();
  const double *Xs = X_seen ? X_seen + model * cap * D : nullptr;
  unsigned long long my_key = ~0ULL;
  int my_idx = 0x7fffffff;
  for (int r = threadIdx.x; r < R; r += blockDim.x) {
    const long long q = model * R + r;
    const int status = info[q * 5 + 2];
    if (status != 0 && status != 1) continue;  // res.success or res.status == 1
    const double *xr = x + q * D;
    bool dup = false;
    if (Xs) {
      for (long long i = 0; i < n_seen && !dup; ++i) {  // any(np.allclose(x_prev, x) ...)
        bool close = true;
        for (int d = 0; d < D && close; ++d) {
          const double b = xr[d];
          close = fabs(Xs[i * D + d] - b) <= atol + rtol * fabs(b);
        }
        dup = close;
      }
    }
    if (dup) continue;
    const unsigned long long k = orderable64(fun[q]);
    if (k < my_key) {  // (r ascends within a thread: the earliest of equal keys stays)
      my_key = k;
      my_idx = r;
    }
  }
  if (my_idx != 0x7fffffff) atomicMin(&s_key, my_key);
  __syncthreads();
  if (my_idx != 0x7fffffff && my_key == s_key) atomicMin(&s_idx, my_idx);
  __syncthreads();
  const int b = s_idx == 0x7fffffff ? -1 : s_idx;
  if (threadIdx.x == 0) best[model] = b;
  if (b >= 0)
    for (int d = threadIdx.x; d < D; d += blockDim.x) x_best[model * D + d] = x[(model * R + b) * D + d];
}

extern "C" int bore_select_best(int n_models, int num_starts, int D, const double *x,
                                const double *fun, const int32_t *info, const double *X_seen,
                                int64_t n_seen, int64_t cap, double rtol, double atol,
                                double *x_best, int32_t *best, void *stream) {
  if (n_models < 1 || num_starts < 1 || D < 1 || !x || !fun || !info || !x_best || !best)
    return fail(BORE_E_INVALID, "select_best: bad argument");
  if (X_seen && (n_seen < 0 || n_seen > cap))
    return fail(BORE_E_INVALID, "select_best: n_seen %lld outside 0..cap %lld", (long long)n_seen,
                (long long)cap);
  hipLaunchKernelGGL(select_kernel, dim3(n_models), dim3(BORE_THREADS), 0, (hipStream_t)stream,
                     num_starts, D, x, fun, info, X_seen, (long long)n_seen, (long long)cap, rtol,
                     atol, x_best, best);
  HIP_TRY(hipGetLastError());
  return 0;
}
