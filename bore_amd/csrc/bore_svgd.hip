// bore_svgd.hip -- Stein variational gradient descent for batch acquisition, all iterations of
// all particles in ONE launch per model.  gfx950 (MI355X) only.  C-ABI: include/bore_hip.h.
//
// What it stands in for: SVGD.optimize_from_init (bore/optimizers/svgd/base.py:79-119) with the
// RadialBasis kernel (bore/optimizers/svgd/kernels.py:4-28), as BatchMaximizableMixin.argmax_batch
// drives it (bore/mixins.py:100-116).  Per iteration, for n particles x in R^(n x D):
//     sq_ij = |x_i - x_j|^2;  h = length_scale or sqrt(median(sq) / (2 log(n + 1))), >= 1e-6
//     K = exp(-sq / (2 h^2));  K_grad_i = 2 sum_j gamma (x_i - x_j) K_ij,  gamma = 1 / (2 h^2)
//     f, f_grad = value and input gradient of transform(f(x))        (fp32 network, fp64 edges)
//     zeta = distortion(rank(f))                                      (constant c | rank^-lambda)
//     grad = (K (zeta * f_grad) + tau K_grad) / n
//     hist = grad^2 (first) | alpha hist + (1 - alpha) grad^2;  x += step grad / (eps + sqrt(hist))
//     x = clip(x, low, high)
// The host statement of the same arithmetic (bore_amd/optimizers/svgd.py, bit-equal to the
// reference) needs a launch, a download and an upload per iteration; here the particles, the
// kernel matrix and the Adagrad history stay in LDS for all n_iter iterations; the median of the
// n^2 distances is a radix select.  Sums run in plain
// index order and exp() is the device's, so particles agree with the host statement to rounding
// (tests: 1e-9 after 200 iterations), not bit for bit.
//
// BF = 0: float32 network of any shape (the LDS row-block path); BF = 3 / 4: a bfloat16 network of the wide
// static shapes (the only ones the library has bfloat16 kernels for): value and input gradient of the
// particles on the bf16 matrix cores (arg_bf16_mfma.h), the same arithmetic as
// bore_mlp_value_and_input_grad gives the host driver for such a model.
#include <hip/hip_runtime.h>

#include <cmath>

#include "arg_bf16_mfma.h"
#include "host_common.h"
#include "mlp_device.h"
#include "mlp_regs.h"

using namespace bore;

#define BORE_SVGD_MAX_PARTICLES 4096

struct SvgdArgs {
  MlpLayout L;
  const float *theta;
  const double *x_init;
  double *x_out;
  double lo[BORE_DIM_MAX], hi[BORE_DIM_MAX];
  int clip, n, n_iter, transform, distortion, n_sort;
  double step, alpha, eps, tau, length_scale, dparam;
  // LDS carve (float offsets; the fp64 regions start on 16-byte boundaries)
  int o_tile, o_vals, o_x, o_fg, o_grad, o_hist, o_K, o_sort, o_f, total, o_layout;
};

// value and input gradient of all n particles for a bfloat16 network: the waves take the 16-row blocks in turn
template <int BF>
__device__ __forceinline__ void svgd_eval_bf16(const MlpLayout &Lrt, int transform, float *smem, const double *x,
                                               double *f, double *fg, int n, int D) {
  using ANet = ArgBf16Net<BF>;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, m16 = lane & 15, q4 = lane >> 4;
  const ArgBf16Images im = arg_bf16_images<BF>(smem);
  ANet net;
  net.set_acts(Lrt);
  for (int rb = wv; rb * 16 < n; rb += BORE_THREADS / 64) {
    const int row = rb * 16 + m16;
    bf16x8_t xf[ANet::CF1];
    ANet::make_xfrag(xf, [&](int d) -> float {
      return d < D && row < n ? (float)x[row * D + d] : 0.f;  // Keras autocast fp64 -> fp32
    });
    const float Tv = net.fg(im, xf, transform, 1.f);
    if (lane < 16 && row < n) f[row] = (double)Tv;
    if (row < n) {
#pragma unroll
      for (int t = 0; t < ANet::T0; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 16 * t + 4 * q4 + r;
          if (d < D) fg[row * D + d] = (double)net.d[0][t][r];
        }
    }
  }
}

template <int BF>
__global__ __launch_bounds__(BORE_THREADS) void svgd_kernel(const SvgdArgs a) {
  extern __shared__ float smem[];
  constexpr MlpLayout Lc = bore_static_layout(BF, 2, BORE_BATCH_MAX);
  const MlpLayout &L = begin_kernel<BF>(Lc, a.L, smem, a.total, a.o_layout);
  const int tid = threadIdx.x, wv = tid >> 6;
  const long long model = blockIdx.x;
  const int n_lay = L.n_layers, D = L.w[0], n = a.n, nn = n * n, nD = n * D;
  float *th = smem, *tile = smem + a.o_tile, *vals = smem + a.o_vals;
  double *x = reinterpret_cast<double *>(smem + a.o_x);
  double *fg = reinterpret_cast<double *>(smem + a.o_fg);
  double *grad = reinterpret_cast<double *>(smem + a.o_grad);
  double *hist = reinterpret_cast<double *>(smem + a.o_hist);
  double *K = reinterpret_cast<double *>(smem + a.o_K);
  double *srt = reinterpret_cast<double *>(smem + a.o_sort);
  double *f = reinterpret_cast<double *>(smem + a.o_f), *zeta = f + n;
  if constexpr (BF) arg_bf16_stage<BF>(a.theta + model * L.P, smem);
  else stage_theta<false>(L, n_lay, a.theta + model * L.P, smem);
  for (int e = tid; e < nD; e += BORE_THREADS) x[e] = a.x_init[model * nD + e];
  __syncthreads();

  for (int it = 0; it < a.n_iter; ++it) {
    // squared distances (and a copy to sort for the median heuristic)
    for (int e = tid; e < nn; e += BORE_THREADS) {
      const int i = e / n, j = e - i * n;
      double s = 0.0;
      for (int d = 0; d < D; ++d) {
        const double t = x[i * D + d] - x[j * D + d];
        s += t * t;
      }
      K[e] = s;
      if (a.length_scale < 0.0) srt[e] = s;
    }
    // the particles as network inputs (Keras autocast fp64 -> fp32)
    if constexpr (!BF)
      for (int e = tid; e < nD; e += BORE_THREADS) {
        const int i = e / D, d = e - i * D;
        tile[L.aoff[0] + i * L.lda[0] + d] = (float)x[e];
      }
    double h = a.length_scale;
    if (a.length_scale < 0.0 && a.n_sort <= 1024) {  // np.median, few entries: bitonic sort in LDS
      for (int e = nn + tid; e < a.n_sort; e += BORE_THREADS) srt[e] = INFINITY;
      __syncthreads();
      for (int k = 2; k <= a.n_sort; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
          for (int t = tid; t < a.n_sort; t += BORE_THREADS) {
            const int u = t ^ j;
            if (u > t) {
              const double p = srt[t], q = srt[u];
              if ((p > q) == ((t & k) == 0)) {
                srt[t] = q;
                srt[u] = p;
              }
            }
          }
          __syncthreads();
        }
      const double med = (nn & 1) ? srt[nn >> 1] : (srt[(nn >> 1) - 1] + srt[nn >> 1]) / 2.0;
      h = sqrt(.5 * med / log((double)(n + 1)));
    } else if (a.length_scale < 0.0) {
      // np.median over all n^2 entries = the order statistics k1 = (nn - 1) / 2 and k2 = nn / 2.
      // Radix select on the bit patterns (squared distances are >= 0: their bits order like the
      // values), 8 bits per pass from the top: a 256-bin histogram of the entries that share the
      // prefix found so far, then the bin holding rank k1 (thread b owns bin b: a scan over the
      // 256 counts finds it).  (A full bitonic sort of 4096 entries costs 78 barrier stages.)
      const unsigned long long *sb = reinterpret_cast<const unsigned long long *>(srt);
      unsigned *hist_s = reinterpret_cast<unsigned *>(srt + a.n_sort);  // [256] bins + [8] scratch
      int *scan_s = reinterpret_cast<int *>(hist_s + 256);  // [0..3] wave totals, [4] bin, [5] below
      const int k1 = (nn - 1) >> 1, k2 = nn >> 1;
      unsigned long long prefix = 0;
      int rank = k1;  // rank of the wanted entry among those sharing `prefix`
      for (int shift = 56; shift >= 0; shift -= 8) {
        hist_s[tid] = 0u;
        __syncthreads();
        const unsigned long long hi_mask = shift == 56 ? 0ULL : ~0ULL << (shift + 8);
        for (int e = tid; e < nn; e += BORE_THREADS) {
          const unsigned long long v = sb[e];
          if ((v & hi_mask) == prefix) atomicAdd(&hist_s[(unsigned)(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        // inclusive scan of the 256 counts: within each wave by shuffles, then the wave totals
        const int cnt = (int)hist_s[tid];
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int up = __shfl_up(incl, off, 64);
          if ((tid & 63) >= off) incl += up;
        }
        if ((tid & 63) == 63) scan_s[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += scan_s[w];
        incl += base;
        if (incl - cnt <= rank && rank < incl) {  // exactly one bin holds the rank
          scan_s[4] = tid;
          scan_s[5] = incl - cnt;
        }
        __syncthreads();
        rank -= scan_s[5];
        prefix |= (unsigned long long)scan_s[4] << shift;
        __syncthreads();
      }
      const double v1 = __longlong_as_double((long long)prefix);
      double med = v1;
      if (k2 != k1) {  // even count: the next entry in order = v1 again if it repeats, else min above
        unsigned *cnt_le = hist_s;                                       // #entries <= v1
        unsigned long long *min_gt = reinterpret_cast<unsigned long long *>(hist_s + 2);
        if (tid == 0) {
          *cnt_le = 0u;
          *min_gt = ~0ULL;
        }
        __syncthreads();
        unsigned c = 0;
        unsigned long long mg = ~0ULL;
        for (int e = tid; e < nn; e += BORE_THREADS) {
          const unsigned long long v = sb[e];
          if (v <= prefix) ++c;
          else if (v < mg) mg = v;
        }
        atomicAdd(cnt_le, c);
        atomicMin(min_gt, mg);
        __syncthreads();
        const double v2 = (int)*cnt_le > k2 ? v1 : __longlong_as_double((long long)*min_gt);
        med = (v1 + v2) / 2.0;
        __syncthreads();
      }
      h = sqrt(.5 * med / log((double)(n + 1)));
    } else {
      __syncthreads();
    }
    h = fmax(h, 1e-6);
    const double gamma = .5 / (h * h);
    for (int e = tid; e < nn; e += BORE_THREADS) K[e] = exp(-gamma * K[e]);
    // value and input gradient of every particle: one row-block per wave
    if constexpr (BF) {
      svgd_eval_bf16<BF>(a.L, a.transform, smem, x, f, fg, n, D);
    } else {
      if (wv * 16 < n) fg_rowblock(L, n_lay, th, tile, wv, a.transform, 1.f, vals);
      __syncthreads();
      for (int e = tid; e < nD; e += BORE_THREADS) {
        const int i = e / D, d = e - i * D;
        fg[e] = (double)tile[L.doff[0] + i * L.lda[0] + d];
      }
      for (int i = tid; i < n; i += BORE_THREADS) f[i] = (double)vals[i];
    }
    __syncthreads();
    for (int i = tid; i < n; i += BORE_THREADS) {  // zeta = distortion(rank(f))
      double z = a.dparam;
      if (a.distortion == 1) {
        int c = 0;
        for (int j = 0; j < n; ++j) c += f[j] <= f[i];
        z = pow((double)c / (double)n, -a.dparam);
      }
      zeta[i] = z;
    }
    __syncthreads();
    for (int e = tid; e < nD; e += BORE_THREADS) {
      const int i = e / D, d = e - i * D;
      double drive = 0.0, rep = 0.0;
      for (int j = 0; j < n; ++j) {
        const double kij = K[i * n + j];
        drive += kij * (zeta[j] * fg[j * D + d]);
        rep += gamma * (x[i * D + d] - x[j * D + d]) * kij;
      }
      grad[e] = (drive + a.tau * (2.0 * rep)) / (double)n;
    }
    __syncthreads();
    for (int e = tid; e < nD; e += BORE_THREADS) {
      const int d = e % D;
      const double g = grad[e];
      const double hs = it == 0 ? g * g : a.alpha * hist[e] + (1.0 - a.alpha) * (g * g);
      hist[e] = hs;
      double xn = x[e] + a.step * (g / (a.eps + sqrt(hs)));
      if (a.clip) xn = fmin(fmax(xn, a.lo[d]), a.hi[d]);
      x[e] = xn;
    }
    __syncthreads();
  }
  for (int e = tid; e < nD; e += BORE_THREADS) a.x_out[model * nD + e] = x[e];
}


// More than 64 particles (one thread per particle in the interaction, in turns beyond 256): the n x n kernel
// matrix does not fit in LDS beside the rest (n = 128: 128 KB), so it is never stored -- thread i
// walks j = 0 .. n-1, forms |x_i - x_j|^2 and exp(-gamma .) on the fly and adds particle j's drive
// and repulsion terms to its D accumulators (8 coordinates per pass over j); the median of the n^2
// distances is the same radix select, each of its passes recomputing the distances; the network
// sees the particles in chunks of 64 rows.  Same sums in the same (index) order as svgd_kernel.
template <int BF>
__global__ __launch_bounds__(BORE_THREADS) void svgd_big_kernel(const SvgdArgs a) {
  extern __shared__ float smem[];
  constexpr MlpLayout Lc = bore_static_layout(BF, 2, BORE_BATCH_MAX);
  const MlpLayout &L = begin_kernel<BF>(Lc, a.L, smem, a.total, a.o_layout);
  const int tid = threadIdx.x, wv = tid >> 6;
  const long long model = blockIdx.x;
  const int n_lay = L.n_layers, D = L.w[0], n = a.n, nD = n * D;
  const long long nn = (long long)n * n;
  float *th = smem, *tile = smem + a.o_tile, *vals = smem + a.o_vals;
  double *x = reinterpret_cast<double *>(smem + a.o_x);
  double *fg = reinterpret_cast<double *>(smem + a.o_fg);
  double *grad = reinterpret_cast<double *>(smem + a.o_grad);
  double *hist = reinterpret_cast<double *>(smem + a.o_hist);
  unsigned *hist_s = reinterpret_cast<unsigned *>(smem + a.o_sort);  // [256] bins + [8] scratch
  int *scan_s = reinterpret_cast<int *>(hist_s + 256);               // [0..3] wave totals, [4] bin, [5] below
  double *f = reinterpret_cast<double *>(smem + a.o_f), *zeta = f + n;
  if constexpr (BF) arg_bf16_stage<BF>(a.theta + model * L.P, smem);
  else stage_theta<false>(L, n_lay, a.theta + model * L.P, smem);
  for (int e = tid; e < nD; e += BORE_THREADS) x[e] = a.x_init[model * nD + e];
  __syncthreads();
  auto sqdist = [&](int i, int j) {
    double s = 0.0;
    for (int d = 0; d < D; ++d) {
      const double t = x[i * D + d] - x[j * D + d];
      s += t * t;
    }
    return s;
  };

  for (int it = 0; it < a.n_iter; ++it) {
    double h = a.length_scale;
    if (a.length_scale < 0.0) {
      // np.median over all n^2 squared distances: order statistics k1 = (nn - 1) / 2 and k2 = nn / 2
      // by a radix select on the bit patterns (svgd_kernel), 8 bits per pass from the top
      const long long k1 = (nn - 1) >> 1, k2 = nn >> 1;
      unsigned long long prefix = 0;
      long long rank = k1;
      for (int shift = 56; shift >= 0; shift -= 8) {
        hist_s[tid] = 0u;
        __syncthreads();
        const unsigned long long hi_mask = shift == 56 ? 0ULL : ~0ULL << (shift + 8);
        for (long long e = tid; e < nn; e += BORE_THREADS) {
          const int i = (int)(e / n), j = (int)(e - (long long)i * n);
          const unsigned long long v = (unsigned long long)__double_as_longlong(sqdist(i, j));
          if ((v & hi_mask) == prefix) atomicAdd(&hist_s[(unsigned)(v >> shift) & 255u], 1u);
        }
        __syncthreads();
        const int cnt = (int)hist_s[tid];
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
          const int up = __shfl_up(incl, off, 64);
          if ((tid & 63) >= off) incl += up;
        }
        if ((tid & 63) == 63) scan_s[tid >> 6] = incl;
        __syncthreads();
        int base = 0;
        for (int w = 0; w < (tid >> 6); ++w) base += scan_s[w];
        incl += base;
        if (incl - cnt <= rank && rank < incl) {  // exactly one bin holds the rank
          scan_s[4] = tid;
          scan_s[5] = incl - cnt;
        }
        __syncthreads();
        rank -= scan_s[5];
        prefix |= (unsigned long long)scan_s[4] << shift;
        __syncthreads();
      }
      const double v1 = __longlong_as_double((long long)prefix);
      double med = v1;
      if (k2 != k1) {  // even count: the next entry in order = v1 again if it repeats, else min above
        unsigned *cnt_le = hist_s;
        unsigned long long *min_gt = reinterpret_cast<unsigned long long *>(hist_s + 2);
        if (tid == 0) {
          *cnt_le = 0u;
          *min_gt = ~0ULL;
        }
        __syncthreads();
        unsigned c = 0;
        unsigned long long mg = ~0ULL;
        for (long long e = tid; e < nn; e += BORE_THREADS) {
          const int i = (int)(e / n), j = (int)(e - (long long)i * n);
          const unsigned long long v = (unsigned long long)__double_as_longlong(sqdist(i, j));
          if (v <= prefix) ++c;
          else if (v < mg) mg = v;
        }
        atomicAdd(cnt_le, c);
        atomicMin(min_gt, mg);
        __syncthreads();
        const double v2 = (long long)*cnt_le > k2 ? v1 : __longlong_as_double((long long)*min_gt);
        med = (v1 + v2) / 2.0;
        __syncthreads();
      }
      h = sqrt(.5 * med / log((double)(n + 1)));
    }
    h = fmax(h, 1e-6);
    const double gamma = .5 / (h * h);
    // value and input gradient of every particle, 64 rows of the tile at a time
    if constexpr (BF) {
      svgd_eval_bf16<BF>(a.L, a.transform, smem, x, f, fg, n, D);
      __syncthreads();
    } else
    for (int c0 = 0; c0 < n; c0 += BORE_BATCH_MAX) {
      const int nc = min(BORE_BATCH_MAX, n - c0);
      for (int e = tid; e < BORE_BATCH_MAX * D; e += BORE_THREADS) {
        const int i = e / D, d = e - i * D;
        tile[L.aoff[0] + i * L.lda[0] + d] = i < nc ? (float)x[(c0 + i) * D + d] : 0.f;  // Keras autocast
      }
      __syncthreads();
      if (wv * 16 < nc) fg_rowblock(L, n_lay, th, tile, wv, a.transform, 1.f, vals);
      __syncthreads();
      for (int e = tid; e < nc * D; e += BORE_THREADS) {
        const int i = e / D, d = e - i * D;
        fg[(c0 + i) * D + d] = (double)tile[L.doff[0] + i * L.lda[0] + d];
      }
      for (int i = tid; i < nc; i += BORE_THREADS) f[c0 + i] = (double)vals[i];
      __syncthreads();
    }
    for (int i = tid; i < n; i += BORE_THREADS) {  // zeta = distortion(rank(f))
      double z = a.dparam;
      if (a.distortion == 1) {
        int c = 0;
        for (int j = 0; j < n; ++j) c += f[j] <= f[i];
        z = pow((double)c / (double)n, -a.dparam);
      }
      zeta[i] = z;
    }
    __syncthreads();
    for (int i = tid; i < n; i += BORE_THREADS) {  // particle i: drive and repulsion, 8 coordinates per pass over the others
      for (int d0 = 0; d0 < D; d0 += 8) {
        double drive[8], rep[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) drive[q] = rep[q] = 0.0;
        for (int j = 0; j < n; ++j) {
          const double kij = exp(-gamma * sqdist(i, j));
          const double zj = zeta[j];
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (d0 + q < D) {
              drive[q] += kij * (zj * fg[j * D + d0 + q]);
              rep[q] += gamma * (x[i * D + d0 + q] - x[j * D + d0 + q]) * kij;
            }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q)
          if (d0 + q < D) grad[i * D + d0 + q] = (drive[q] + a.tau * (2.0 * rep[q])) / (double)n;
      }
    }
    __syncthreads();
    for (int e = tid; e < nD; e += BORE_THREADS) {
      const int d = e % D;
      const double g = grad[e];
      const double hs = it == 0 ? g * g : a.alpha * hist[e] + (1.0 - a.alpha) * (g * g);
      hist[e] = hs;
      double xn = x[e] + a.step * (g / (a.eps + sqrt(hs)));
      if (a.clip) xn = fmin(fmax(xn, a.lo[d]), a.hi[d]);
      x[e] = xn;
    }
    __syncthreads();
  }
  for (int e = tid; e < nD; e += BORE_THREADS) a.x_out[model * nD + e] = x[e];
}

extern "C" int bore_svgd_optimize(const bore_mlp_desc *desc, int n_models, const float *theta,
                                  int transform, const double *x_init, int n_particles,
                                  const double *lb, const double *ub, const bore_svgd_opts *opts,
                                  double *x_out, void *stream) {
  if (!desc || !theta || !x_init || !x_out || !opts) return fail(BORE_E_INVALID, "svgd_optimize: NULL argument");
  if (n_models < 1) return fail(BORE_E_INVALID, "svgd_optimize: n_models must be >= 1");
  const int flav = bore_kernel_flavour(desc, true);
  const bool bf = desc->compute == BORE_COMPUTE_BF16;
  if (bf && !bore_shape_is_wide(flav)) return fail(BORE_E_UNSUPPORTED, kBf16Shapes);
  if (bf && !bore_flavour_built(flav)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
  const int D = desc->input_dim, n = n_particles;
  if (D < 1 || D > BORE_DIM_MAX) return fail(BORE_E_UNSUPPORTED, "svgd_optimize: input_dim must be 1..%d", BORE_DIM_MAX);
  if (n < 1 || n > BORE_SVGD_MAX_PARTICLES)  // (what fits is decided by the LDS check below: 32 n D bytes of state)
    return fail(BORE_E_UNSUPPORTED, "svgd_optimize: 1..%d particles per launch", BORE_SVGD_MAX_PARTICLES);
  const bool big = n > BORE_BATCH_MAX;  // (no kernel matrix in LDS: svgd_big_kernel)
  if (transform < BORE_T_IDENTITY || transform > BORE_T_EXP)
    return fail(BORE_E_INVALID, "svgd_optimize: unknown transform %d", transform);
  if (opts->n_iter < 0 || (opts->distortion != 0 && opts->distortion != 1))
    return fail(BORE_E_INVALID, "svgd_optimize: bad options");
  if ((lb == nullptr) != (ub == nullptr)) return fail(BORE_E_INVALID, "svgd_optimize: lb and ub go together");
  SvgdArgs a;
  if (bore_make_layout(desc, 2, big ? BORE_BATCH_MAX : 16 * ((n + 15) / 16), &a.L)) return fail(BORE_E_INVALID, "bad bore_mlp_desc");
  if (a.L.w[a.L.n_layers] != 1) return fail(BORE_E_INVALID, "svgd_optimize: the last Dense layer must have 1 unit");
  a.clip = lb != nullptr;
  for (int d = 0; d < D; ++d) {
    a.lo[d] = lb ? lb[d] : 0.0;
    a.hi[d] = ub ? ub[d] : 0.0;
  }
  a.theta = theta; a.x_init = x_init; a.x_out = x_out;
  a.n = n; a.n_iter = opts->n_iter; a.transform = transform; a.distortion = opts->distortion;
  a.step = opts->step_size; a.alpha = opts->alpha; a.eps = opts->eps; a.tau = opts->tau;
  a.length_scale = opts->length_scale; a.dparam = opts->distortion_param;
  int ns = 1;
  while (ns < n * n) ns <<= 1;
  a.n_sort = ns;
  const size_t nD2 = 2 * (((size_t)n * D + 1) & ~(size_t)1);  // floats of an [n][D] fp64 array, 16-B multiple
  // (a bfloat16 network: its image instead of the padded float32 theta, and no activation tile)
  size_t off = bf ? (flav == 3 ? ArgBf16Plan<3>::floats : ArgBf16Plan<4>::floats) : (size_t)a.L.P_lds;
  a.o_tile = (int)off; off += bf ? 0 : a.L.tile_floats;
  a.o_vals = (int)off; off += BORE_BATCH_MAX;
  off = (off + 3) & ~(size_t)3;
  a.o_x = (int)off; off += nD2;
  a.o_fg = (int)off; off += nD2;
  a.o_grad = (int)off; off += nD2;
  a.o_hist = (int)off; off += nD2;
  a.o_K = (int)off; off += big ? 0 : 2 * (size_t)n * n + 2;
  off = (off + 3) & ~(size_t)3;
  a.o_sort = (int)off; off += big ? 256 + 8 : (opts->length_scale < 0.0 ? 2 * (size_t)ns + 256 + 8 : 0);
  a.o_f = (int)off; off += 4 * (size_t)n + 4;
  a.total = (int)off;
  off = (off + 3) & ~(size_t)3;
  a.o_layout = (int)off; off += BORE_LAYOUT_FLOATS;
  if (off * 4 > BORE_LDS_BYTES)
    return fail(BORE_E_UNSUPPORTED, "svgd_optimize: %d particles in %d dimensions need %zu B of LDS (> %d)",
                n, D, off * 4, BORE_LDS_BYTES);
  int rc = 0;
#define BORE_SVGD_LAUNCH(BF)                                                                                   \
  do {                                                                                                         \
    rc = big ? allow_lds(svgd_big_kernel<BF>, off * 4) : allow_lds(svgd_kernel<BF>, off * 4);                  \
    if (rc) return rc;                                                                                         \
    if (big)                                                                                                   \
      hipLaunchKernelGGL(svgd_big_kernel<BF>, dim3(n_models), dim3(BORE_THREADS), off * 4, (hipStream_t)stream, a); \
    else                                                                                                       \
      hipLaunchKernelGGL(svgd_kernel<BF>, dim3(n_models), dim3(BORE_THREADS), off * 4, (hipStream_t)stream, a); \
  } while (0)
  if (!bf) BORE_SVGD_LAUNCH(0);
#if BORE_ON_3
  else if (flav == 3) BORE_SVGD_LAUNCH(3);
#endif
#if BORE_ON_4
  else if (flav == 4) BORE_SVGD_LAUNCH(4);
#endif
#undef BORE_SVGD_LAUNCH
  HIP_TRY(hipGetLastError());
  return 0;
}
