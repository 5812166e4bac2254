// bore_hip.hip -- kernels + C-ABI of libbore_hip.so (include/bore_hip.h).
// gfx950 (MI355X) only.  See DESIGN.md for the data layout and per-kernel rooflines.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "host_common.h"
#include "mlp_device.h"

using namespace bore;

extern "C" int bore_abi_version(void) { return BORE_ABI_VERSION; }
extern "C" const char *bore_last_error(void) { return g_bore_err; }

extern "C" int64_t bore_param_count(const bore_mlp_desc *desc) {
  MlpLayout L;
  if (bore_make_layout(desc, 0, 1, &L)) return fail(BORE_E_INVALID, "bad bore_mlp_desc");
  return L.P;
}

// ---------------------------------------------------------------------------
// fit: one workgroup per model, all Adam steps of the call inside one launch
// ---------------------------------------------------------------------------
struct FitArgs {
  MlpLayout L;
  float *theta, *am, *av;
  long long *at;
  const float *X, *z;
  const int *perm;
  float *epoch_loss;
  unsigned long long seed;
  long long model0, epoch0;
  int N, epochs, B;
  float lr, beta1, beta2, eps;
  int state_in_lds, data_in_lds;
  // LDS carve (float offsets)
  int o_tile, o_zt, o_misc, o_m, o_v, o_perm, o_keys, o_X, o_z;
};

__global__ __launch_bounds__(BORE_THREADS) void fit_kernel(const FitArgs a) {
  extern __shared__ float smem[];
  const MlpLayout &L = a.L;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long long model = blockIdx.x;
  const int P = L.P, n = L.n_layers, D = L.w[0], N = a.N;

  float *th = smem;
  float *tile = smem + a.o_tile;
  float *zt = smem + a.o_zt;
  float *misc = smem + a.o_misc;  // [0] = l2 penalty accumulator
  int *perm_s = reinterpret_cast<int *>(smem + a.o_perm);
  unsigned *keys = reinterpret_cast<unsigned *>(smem + a.o_keys);

  float *theta_g = a.theta + model * P;
  float *m_g = a.am + model * P;
  float *v_g = a.av + model * P;
  const float *X_g = a.X + model * (long long)N * D;
  const float *z_g = a.z + model * (long long)N;

  // state: m, v indexed by PACKED index p (adjacent threads, adjacent p)
  float *sm = a.state_in_lds ? smem + a.o_m : m_g;
  float *sv = a.state_in_lds ? smem + a.o_v : v_g;
  const float *Xs = a.data_in_lds ? smem + a.o_X : X_g;
  const float *zs = a.data_in_lds ? smem + a.o_z : z_g;

  load_theta(L, theta_g, th);
  if (a.state_in_lds)
    for (int p = tid; p < P; p += nthr) {
      smem[a.o_m + p] = m_g[p];
      smem[a.o_v + p] = v_g[p];
    }
  if (a.data_in_lds) {
    for (int i = tid; i < N * D; i += nthr) smem[a.o_X + i] = X_g[i];
    for (int i = tid; i < N; i += nthr) smem[a.o_z + i] = z_g[i];
  }
  if (tid == 0) misc[0] = 0.f;
  __syncthreads();
  if (L.any_l2) {  // l2 penalty of the incoming weights (what the first step's loss sees)
    float reg = 0.f;
    for (int p = tid; p < P; p += nthr) {
      const ParamRef r = param_ref(L, p);
      const float l2 = r.k >= 0 ? L.l2_w[r.l] : L.l2_b[r.l];
      const float w = th[r.lds];
      reg = fmaf(l2 * w, w, reg);
    }
    reg = wave_sum(reg);
    if ((tid & 63) == 0) atomicAdd(&misc[0], reg);
  }

  // running beta powers in fp64 (rounded to fp32 at use; see DESIGN.md "Adam")
  const long long t0 = a.at[model];
  double b1p = pow((double)a.beta1, (double)t0);
  double b2p = pow((double)a.beta2, (double)t0);
  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  const int steps = (N + a.B - 1) / a.B;
  __syncthreads();

  for (int e = 0; e < a.epochs; ++e) {
    if (a.perm) {
      const int *pg = a.perm + (model * a.epochs + e) * (long long)N;
      for (int i = tid; i < N; i += nthr) perm_s[i] = pg[i];
      __syncthreads();
    } else {
      make_perm(shuffle_base(a.seed, a.model0 + model, a.epoch0 + e), N, keys, perm_s);
    }
    float eloss = 0.f;  // thread 0: sum over the epoch of per-row losses (+ nb * penalty)

    for (int s = 0; s < steps; ++s) {
      const int row0 = s * a.B;
      const int nb = min(a.B, N - row0);
      // gather the mini-batch rows
      {
        float *A0 = tile + L.aoff[0];
        const int lda0 = L.lda[0];
        for (int idx = tid; idx < nb * D; idx += nthr) {
          const int b = idx / D, d = idx - b * D;
          A0[b * lda0 + d] = Xs[perm_s[row0 + b] * D + d];
        }
        if (tid < nb) zt[tid] = zs[perm_s[row0 + tid]];
      }
      __syncthreads();
      for (int l = 1; l <= n; ++l) {
        fwd_layer(L, th, tile, l, nb, /*keep_logits=*/l == n);
        __syncthreads();
      }
      // loss + d loss / d logit  (the final layer has one unit)
      if (tid < 64) {
        float lossb = 0.f;
        if (tid < nb) {
          const float x = tile[L.aoff[n] + tid * L.lda[n]];
          const float zz = zt[tid];
          lossb = fmaxf(x, 0.f) - x * zz + log1pf(expf(-fabsf(x)));
          tile[L.doff[n] + tid * L.lda[n]] = (sigmoid_stable(x) - zz) / (float)nb;
        }
        lossb = wave_sum(lossb);
        if (tid == 0) eloss += lossb + (L.any_l2 ? misc[0] * (float)nb : 0.f);
      }
      __syncthreads();
      for (int l = n; l >= 2; --l) {
        bwd_delta(L, th, tile, l, nb);
        __syncthreads();
      }
      if (L.any_l2 && tid == 0) misc[0] = 0.f;  // consumed above; re-accumulated below
      // weight gradients + Adam, one thread per parameter (packed order)
      b1p *= (double)a.beta1;
      b2p *= (double)a.beta2;
      const float alpha = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
      float reg = 0.f;
      for (int p = tid; p < P; p += nthr) {
        const ParamRef r = param_ref(L, p);
        const float *Dl = tile + L.doff[r.l] + r.j;
        const int ldd = L.lda[r.l];
        float g = 0.f;
        if (r.k >= 0) {
          const float *Ap = tile + L.aoff[r.l - 1] + r.k;
          const int ldap = L.lda[r.l - 1];
          for (int b = 0; b < nb; ++b) g = fmaf(Ap[b * ldap], Dl[b * ldd], g);
        } else {
          for (int b = 0; b < nb; ++b) g += Dl[b * ldd];
        }
        float w = th[r.lds];
        const float l2 = r.k >= 0 ? L.l2_w[r.l] : L.l2_b[r.l];
        if (l2 != 0.f) g = fmaf(2.f * l2, w, g);
        float mm = sm[p], vv = sv[p];
        mm += (g - mm) * omb1;
        vv += (g * g - vv) * omb2;
        w -= (mm * alpha) / (sqrtf(vv) + a.eps);
        sm[p] = mm;
        sv[p] = vv;
        th[r.lds] = w;
        if (l2 != 0.f) reg = fmaf(l2 * w, w, reg);
      }
      __syncthreads();
      if (L.any_l2) {  // penalty of the UPDATED weights = the one the next step's loss sees
        reg = wave_sum(reg);
        if ((tid & 63) == 0) atomicAdd(&misc[0], reg);
        __syncthreads();
      }
    }
    if (tid == 0 && a.epoch_loss) a.epoch_loss[model * a.epochs + e] = eloss / (float)N;
  }

  store_theta(L, th, theta_g);
  if (a.state_in_lds)
    for (int p = tid; p < P; p += nthr) {
      m_g[p] = smem[a.o_m + p];
      v_g[p] = smem[a.o_v + p];
    }
  if (tid == 0) a.at[model] = t0 + (long long)a.epochs * steps;
}

// ---------------------------------------------------------------------------
// forward (predict) / value + input gradient: grid = (models, tile slots)
// ---------------------------------------------------------------------------
struct RowArgs {
  MlpLayout L;
  const float *theta;
  const float *Xf;   // forward: fp32 rows
  const double *Xd;  // input-gradient: fp64 rows
  float *out;        // forward: [models][rows]; input-gradient: val
  double *grad;
  long long n_rows;
  int x_shared, transform;
  float sign;  // -1: T(-f) (minimisation form), +1: T(f)
  int o_tile;
};

__global__ __launch_bounds__(BORE_THREADS) void forward_kernel(const RowArgs a) {
  extern __shared__ float smem[];
  const MlpLayout &L = a.L;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long long model = blockIdx.x;
  const int n = L.n_layers, D = L.w[0];
  float *th = smem, *tile = smem + a.o_tile;
  load_theta(L, a.theta + model * L.P, th);
  const float *X = a.Xf + (a.x_shared ? 0 : model * a.n_rows * D);
  float *out = a.out + model * a.n_rows;
  const long long n_tiles = (a.n_rows + L.tb - 1) / L.tb;
  __syncthreads();
  for (long long t = blockIdx.y; t < n_tiles; t += gridDim.y) {
    const long long row0 = t * L.tb;
    const int nb = (int)min((long long)L.tb, a.n_rows - row0);
    float *A0 = tile + L.aoff[0];
    for (int idx = tid; idx < nb * D; idx += nthr) {
      const int b = idx / D, d = idx - b * D;
      A0[b * L.lda[0] + d] = X[row0 * D + idx];
    }
    __syncthreads();
    for (int l = 1; l <= n; ++l) {
      fwd_layer(L, th, tile, l, nb, false);
      __syncthreads();
    }
    if (tid < nb) out[row0 + tid] = tile[L.aoff[n] + tid * L.lda[n]];
    // next tile's gather only touches A_0, whose readers passed a barrier already
  }
}

__global__ __launch_bounds__(BORE_THREADS) void value_grad_kernel(const RowArgs a) {
  extern __shared__ float smem[];
  const MlpLayout &L = a.L;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long long model = blockIdx.x;
  const int n = L.n_layers, D = L.w[0];
  float *th = smem, *tile = smem + a.o_tile;
  load_theta(L, a.theta + model * L.P, th);
  const double *X = a.Xd + model * a.n_rows * D;
  float *val = a.out + model * a.n_rows;
  double *grad = a.grad + model * a.n_rows * D;
  const long long n_tiles = (a.n_rows + L.tb - 1) / L.tb;
  __syncthreads();
  for (long long t = blockIdx.y; t < n_tiles; t += gridDim.y) {
    const long long row0 = t * L.tb;
    const int nb = (int)min((long long)L.tb, a.n_rows - row0);
    float *A0 = tile + L.aoff[0];
    for (int idx = tid; idx < nb * D; idx += nthr) {
      const int b = idx / D, d = idx - b * D;
      A0[b * L.lda[0] + d] = (float)X[row0 * D + idx];  // Keras autocast fp64 -> fp32
    }
    __syncthreads();
    fg_tile(L, th, tile, nb, a.transform, a.sign, val + row0);
    const float *D0 = tile + L.doff[0];
    for (int idx = tid; idx < nb * D; idx += nthr) {
      const int b = idx / D, d = idx - b * D;
      grad[row0 * D + idx] = (double)D0[b * L.lda[0] + d];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------
// evaluate: one workgroup per model
// ---------------------------------------------------------------------------
struct EvalArgs {
  MlpLayout L;
  const float *theta, *X, *z;
  float *loss, *acc;
  long long N;
  int o_tile, o_misc;
};

__global__ __launch_bounds__(BORE_THREADS) void evaluate_kernel(const EvalArgs a) {
  extern __shared__ float smem[];
  const MlpLayout &L = a.L;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const long long model = blockIdx.x;
  const int n = L.n_layers, D = L.w[0];
  float *th = smem, *tile = smem + a.o_tile, *misc = smem + a.o_misc;
  load_theta(L, a.theta + model * L.P, th);
  if (tid == 0) misc[0] = 0.f;
  const float *X = a.X + model * a.N * D;
  const float *z = a.z + model * a.N;
  const long long n_tiles = (a.N + L.tb - 1) / L.tb;
  float lsum = 0.f, csum = 0.f;
  __syncthreads();
  for (long long t = 0; t < n_tiles; ++t) {
    const long long row0 = t * L.tb;
    const int nb = (int)min((long long)L.tb, a.N - row0);
    float *A0 = tile + L.aoff[0];
    for (int idx = tid; idx < nb * D; idx += nthr) {
      const int b = idx / D, d = idx - b * D;
      A0[b * L.lda[0] + d] = X[row0 * D + idx];
    }
    __syncthreads();
    for (int l = 1; l <= n; ++l) {
      fwd_layer(L, th, tile, l, nb, l == n);
      __syncthreads();
    }
    if (tid < nb) {
      const float x = tile[L.aoff[n] + tid * L.lda[n]];
      const float zz = z[row0 + tid];
      lsum += fmaxf(x, 0.f) - x * zz + log1pf(expf(-fabsf(x)));
      const float o = L.act[n] == BORE_ACT_SIGMOID ? sigmoid_stable(x) : x;
      csum += ((o > 0.5f) == (zz > 0.5f)) ? 1.f : 0.f;
    }
  }
  float reg = 0.f;
  if (L.any_l2)
    for (int p = tid; p < L.P; p += nthr) {
      const ParamRef r = param_ref(L, p);
      const float l2 = r.k >= 0 ? L.l2_w[r.l] : L.l2_b[r.l];
      const float w = th[r.lds];
      reg = fmaf(l2 * w, w, reg);
    }
  reg = wave_sum(reg);
  if (L.any_l2 && (tid & 63) == 0) atomicAdd(&misc[0], reg);
  __syncthreads();
  if (tid < 64) {
    lsum = wave_sum(lsum);
    csum = wave_sum(csum);
    if (tid == 0) {
      a.loss[model] = lsum / (float)a.N + misc[0];
      a.acc[model] = csum / (float)a.N;
    }
  }
}

// ---------------------------------------------------------------------------
// shuffle stream dump
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BORE_THREADS) void shuffle_kernel(unsigned long long seed,
                                                              long long model0, long long epoch0,
                                                              int epochs, int N, int *perm) {
  extern __shared__ float smem[];
  unsigned *keys = reinterpret_cast<unsigned *>(smem);
  const long long model = blockIdx.x, e = blockIdx.y;
  make_perm(shuffle_base(seed, model0 + model, epoch0 + e), N, keys,
            perm + (model * epochs + e) * (long long)N);
}

// ---------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------
extern "C" int bore_mlp_fit(const bore_mlp_desc *desc, int n_models, float *theta, float *adam_m,
                            float *adam_v, int64_t *adam_t, const float *X, const float *z,
                            int64_t N, int epochs, int batch_size, const int32_t *perm,
                            uint64_t seed, int64_t model_index0, int64_t epoch0,
                            const bore_adam_cfg *adam, float *epoch_loss, void *stream) {
  FitArgs a;
  if (batch_size < 1 || batch_size > BORE_BATCH_MAX)
    return fail(BORE_E_UNSUPPORTED, "fit: batch_size must be 1..%d (got %d)", BORE_BATCH_MAX,
                batch_size);
  if (N < 1 || N > (1 << 20)) return fail(BORE_E_INVALID, "fit: N=%lld out of range", (long long)N);
  // the whole mini-batch is one tile; perm (+ keys) and the batch targets ride along
  int rc = check_common(desc, n_models, 1, batch_size, false,
                        BORE_BATCH_MAX + 8 + (size_t)N * (perm ? 1 : 2), &a.L);
  if (rc) return rc;
  const MlpLayout &L = a.L;
  if (L.w[L.n_layers] != 1)
    return fail(BORE_E_INVALID, "fit: the last Dense layer must have 1 unit (binary classifier)");
  if (L.act[L.n_layers] != BORE_ACT_SIGMOID && L.act[L.n_layers] != BORE_ACT_LINEAR)
    return fail(BORE_E_INVALID, "fit: BCE needs a sigmoid or linear (from_logits) output layer");
  if (!theta || !adam_m || !adam_v || !adam_t || !X || !z || !adam)
    return fail(BORE_E_INVALID, "fit: null pointer");
  if (epochs < 0) return fail(BORE_E_INVALID, "fit: epochs < 0");
  if (epochs == 0) return 0;

  a.theta = theta; a.am = adam_m; a.av = adam_v; a.at = (long long *)adam_t;
  a.X = X; a.z = z; a.perm = perm; a.epoch_loss = epoch_loss;
  a.seed = seed; a.model0 = model_index0; a.epoch0 = epoch0;
  a.N = (int)N; a.epochs = epochs; a.B = batch_size;
  a.lr = adam->lr; a.beta1 = adam->beta1; a.beta2 = adam->beta2; a.eps = adam->eps;

  // LDS carve: theta | tile | zt | misc | perm | keys | [m v] | [X z]
  size_t off = 0;
  off += L.P_lds;
  a.o_tile = (int)off; off += L.tile_floats;
  a.o_zt = (int)off; off += BORE_BATCH_MAX;
  a.o_misc = (int)off; off += 8;
  a.o_perm = (int)off; off += N;
  a.o_keys = (int)off; off += perm ? 0 : N;
  if (off * 4 > BORE_LDS_BYTES)
    return fail(BORE_E_UNSUPPORTED, "fit: theta+tile+perm need %zu B of LDS (> %d)", off * 4,
                BORE_LDS_BYTES);
  a.state_in_lds = (off + 2 * (size_t)L.P) * 4 <= BORE_LDS_BYTES;
  a.o_m = a.o_v = 0;
  if (a.state_in_lds) {
    a.o_m = (int)off; off += L.P;
    a.o_v = (int)off; off += L.P;
  }
  const size_t data = (size_t)N * (L.w[0] + 1);
  a.data_in_lds = (off + data) * 4 <= BORE_LDS_BYTES;
  a.o_X = a.o_z = 0;
  if (a.data_in_lds) {
    a.o_X = (int)off; off += (size_t)N * L.w[0];
    a.o_z = (int)off; off += N;
  }
  rc = allow_lds(fit_kernel, off * 4);
  if (rc) return rc;
  hipLaunchKernelGGL(fit_kernel, dim3(n_models), dim3(BORE_THREADS), off * 4,
                     (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

static int row_launch(bool with_grad, const bore_mlp_desc *desc, int n_models, RowArgs &a,
                      void *stream) {
  const MlpLayout &L = a.L;
  size_t off = L.P_lds;
  a.o_tile = (int)off;
  off += L.tile_floats;
  const long long n_tiles = (a.n_rows + L.tb - 1) / L.tb;
  // enough workgroups to fill 256 CUs a few times over, never more than tiles
  long long gy = n_tiles;
  const long long cap = (2048 + n_models - 1) / n_models;
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  if (gy > 65535) gy = 65535;
  int rc;
  if (with_grad) {
    rc = allow_lds(value_grad_kernel, off * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(value_grad_kernel, dim3(n_models, (unsigned)gy), dim3(BORE_THREADS),
                       off * 4, (hipStream_t)stream, a);
  } else {
    rc = allow_lds(forward_kernel, off * 4);
    if (rc) return rc;
    hipLaunchKernelGGL(forward_kernel, dim3(n_models, (unsigned)gy), dim3(BORE_THREADS), off * 4,
                       (hipStream_t)stream, a);
  }
  HIP_TRY(hipGetLastError());
  (void)desc;
  return 0;
}

extern "C" int bore_mlp_forward(const bore_mlp_desc *desc, int n_models, const float *theta,
                                const float *X, int64_t n_rows, int x_shared, float *out,
                                void *stream) {
  RowArgs a;
  int rc = check_common(desc, n_models, 0, BORE_BATCH_MAX, true, 0, &a.L);
  if (rc) return rc;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "forward: the last Dense layer must have 1 unit");
  if (!theta || !X || !out) return fail(BORE_E_INVALID, "forward: null pointer");
  if (n_rows < 0) return fail(BORE_E_INVALID, "forward: n_rows < 0");
  if (n_rows == 0) return 0;
  a.theta = theta; a.Xf = X; a.Xd = nullptr; a.out = out; a.grad = nullptr;
  a.n_rows = n_rows; a.x_shared = x_shared; a.transform = 0; a.sign = 1.f;
  return row_launch(false, desc, n_models, a, stream);
}

extern "C" int bore_mlp_value_and_input_grad(const bore_mlp_desc *desc, int n_models,
                                             const float *theta, const double *X, int64_t n_rows,
                                             int transform, int negate, float *val,
                                             double *grad, void *stream) {
  RowArgs a;
  int rc = check_common(desc, n_models, 2, BORE_BATCH_MAX, true, 0, &a.L);
  if (rc) return rc;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "value_and_input_grad: the last Dense layer must have 1 unit");
  if (transform < BORE_T_IDENTITY || transform > BORE_T_EXP)
    return fail(BORE_E_INVALID, "value_and_input_grad: unknown transform %d", transform);
  if (!theta || !X || !val || !grad) return fail(BORE_E_INVALID, "value_and_input_grad: null pointer");
  if (n_rows < 0) return fail(BORE_E_INVALID, "value_and_input_grad: n_rows < 0");
  if (n_rows == 0) return 0;
  a.theta = theta; a.Xf = nullptr; a.Xd = X; a.out = val; a.grad = grad;
  a.n_rows = n_rows; a.x_shared = 0; a.transform = transform; a.sign = negate ? -1.f : 1.f;
  return row_launch(true, desc, n_models, a, stream);
}

extern "C" int bore_mlp_evaluate(const bore_mlp_desc *desc, int n_models, const float *theta,
                                 const float *X, const float *z, int64_t N, float *loss,
                                 float *acc, void *stream) {
  EvalArgs a;
  int rc = check_common(desc, n_models, 0, BORE_BATCH_MAX, true, 8, &a.L);
  if (rc) return rc;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "evaluate: the last Dense layer must have 1 unit");
  if (!theta || !X || !z || !loss || !acc) return fail(BORE_E_INVALID, "evaluate: null pointer");
  if (N < 1) return fail(BORE_E_INVALID, "evaluate: N < 1");
  a.theta = theta; a.X = X; a.z = z; a.loss = loss; a.acc = acc; a.N = N;
  size_t off = a.L.P_lds;
  a.o_tile = (int)off; off += a.L.tile_floats;
  a.o_misc = (int)off; off += 8;
  rc = allow_lds(evaluate_kernel, off * 4);
  if (rc) return rc;
  hipLaunchKernelGGL(evaluate_kernel, dim3(n_models), dim3(BORE_THREADS), off * 4,
                     (hipStream_t)stream, a);
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int bore_shuffle_perm(uint64_t seed, int64_t model_index0, int n_models,
                                 int64_t epoch0, int epochs, int64_t N, int32_t *perm,
                                 void *stream) {
  if (n_models < 1 || epochs < 0 || N < 1 || !perm)
    return fail(BORE_E_INVALID, "shuffle_perm: bad argument");
  if (epochs == 0) return 0;
  if (epochs > 65535) return fail(BORE_E_UNSUPPORTED, "shuffle_perm: epochs > 65535");
  const size_t bytes = (size_t)N * 4;
  int rc = allow_lds(shuffle_kernel, bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(shuffle_kernel, dim3(n_models, epochs), dim3(BORE_THREADS), bytes,
                     (hipStream_t)stream, seed, (long long)model_index0, (long long)epoch0, epochs,
                     (int)N, perm);
  HIP_TRY(hipGetLastError());
  return 0;
}
