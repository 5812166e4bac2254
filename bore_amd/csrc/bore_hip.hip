// bore_hip.hip -- fit / forward / value+input-gradient / evaluate / shuffle kernels and their
// C-ABI entry points (include/bore_hip.h).  gfx950 (MI355X) only.
// See DESIGN.md for the data layout and per-kernel rooflines.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "host_common.h"
#include "mlp_device.h"
#include "mlp_regs.h"
#include "fit_bf16_mfma.h"
#include "arg_bf16_mfma.h"

using namespace bore;

// -DBORE_STAMPS: cycle stamps of one Adam step's phases (diagnostic builds only)
#ifdef BORE_STAMPS
__device__ long long g_stamps[64];
#define BORE_STAMP(i)                                                                  \
  do {                                                                                 \
    if (blockIdx.x == 0 && (tid & 63) == 0 && e == 1 && s == 0) g_stamps[16 * (tid >> 6) + (i)] = clock64(); \
  } while (0)
extern "C" int bore_debug_stamps(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(long long) * 64);
}
__device__ int g_stamp_on;
#define BORE_TSTAMP(i)                                                                         \
  do {                                                                                         \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && g_stamp_on) g_stamps[16 * (threadIdx.x >> 6) + (i)] = clock64(); \
  } while (0)
#else
#define BORE_STAMP(i)
#define BORE_TSTAMP(i)
#endif

// -DBORE_FIT_MARKS: where an Adam step of the static-shape fit spends its cycles, summed over EVERY step
// of every workgroup (tools/fit_marks.py).  Each wave keeps the time of its previous mark in a
// register and adds the interval to a per-wave LDS row (ds_add without return); the rows go to
// global memory once, at the end of the fit.  Buckets: 0 gather + operand requests, 1 forward,
// 2 loss + delta, 3 backward + A / D copies, 4 wait at the mid-step barrier, 5 weight gradients +
// Adam (whole phase), 6 wait at the step's last barrier, 7 step loop top -> first instruction of the
// step; inside a weight-gradient task: 8 operand requests, 9 matrix chain, 10 Adam + stores; 11 two
// marks back to back (what a mark costs: ~85 cycles, included in every bucket); 12 end of a step ->
// top of the next epoch, 13 -> shuffle chosen, 14 -> step loop top; 15 mid-step barrier -> the wave's
// weight-gradient task done (5 then holds what follows it: the next step size, the l2 sums).
#ifdef BORE_FIT_MARKS
__device__ unsigned long long g_fit_acc[8][32];  // (eight waves: fit_kernel_w8)
__shared__ unsigned g_fit_lds[8][32];
#define FIT_MARK_DECL long long fm_last = clock64()
#define FIT_MARK(i)                                                                                    \
  do {                                                                                                 \
    const long long fm_now = clock64();                                                                \
    if ((threadIdx.x & 63) == 0) {                                                                     \
      __hip_atomic_fetch_add(&g_fit_lds[(threadIdx.x >> 6) & 7][i], (unsigned)(fm_now - fm_last), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      __hip_atomic_fetch_add(&g_fit_lds[(threadIdx.x >> 6) & 7][16 + (i)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    }                                                                                                  \
    fm_last = fm_now;                                                                                  \
  } while (0)
extern "C" int bore_debug_fit_marks(unsigned long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_fit_acc), sizeof(unsigned long long) * 256);
  if (reset) {
    unsigned long long z[256] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_fit_acc), z, sizeof(z));
  }
  return rc;
}
#else
#define FIT_MARK_DECL
#define FIT_MARK(i)
#endif

// -DBORE_WIDE_STAMPS: cycles per phase of the wide fits' Adam step, summed over the launch by
// every wave's lane 0 of workgroup 0 (diagnostic builds only; tools/wide_stamps.py)
#ifdef BORE_WIDE_STAMPS
__device__ long long g_wstamps[4][16];
#define BORE_WSTAMP_DECL long long ws_t = clock64(); (void)ws_t
#define BORE_WSTAMP(i)                                                   \
  do {                                                                   \
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) {                    \
      const long long ws_n = clock64();                                  \
      g_wstamps[threadIdx.x >> 6][(i)] += ws_n - ws_t;                   \
      ws_t = ws_n;                                                       \
    }                                                                    \
  } while (0)
extern "C" int bore_debug_wide_stamps(long long *out, int reset) {
  int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wstamps), sizeof(long long) * 64);
  if (reset) {
    long long z[64] = {0};
    rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_wstamps), z, sizeof(z));
  }
  return rc;
}
#else
#define BORE_WSTAMP_DECL
#define BORE_WSTAMP(i)
#endif

extern "C" int bore_abi_version(void) { return BORE_ABI_VERSION; }
#ifndef BORE_SRC_DIGEST
#define BORE_SRC_DIGEST "unknown"
#endif
extern "C" const char *bore_source_digest(void) { return BORE_SRC_DIGEST; }

extern "C" void bore_set_batch(const bore_batch *batch) {
  if (batch) {
    g_batch_store = *batch;
    g_batch = &g_batch_store;
  } else {
    g_batch = nullptr;
  }
}
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
  int perm_in_lds;  // 0: an explicit perm too long for LDS is read from memory per step (generic flavours)
  int perm_ahead;   // 1: room for two shuffles: the eight-wave kernel's fifth wave draws an epoch ahead (fit_body)
  // LDS carve (float offsets)
  int o_tile, o_zt, o_misc, o_stage, o_m, o_v, o_perm, o_perm2, o_keys, o_X, o_z, o_g, o_layout, total;
  // batch mode (bore_set_batch): slot -> loop ids[slot] at iteration its[slot]; N above is the
  // largest of the batch and the data buffers are `cap`-strided per loop
  const int *ids, *its;
  int n_init;
  long long cap;
};

// The fit's arithmetic is "at most 1 ulp per operation", not IEEE-correctly-rounded: square root,
// reciprocal and exp2 are the hardware's v_sqrt_f32 / v_rcp_f32 / v_exp_f32 (1 ulp each).  TensorFlow's
// own Eigen kernels are not correctly rounded either; the parity gate is the oracle
// tolerance of tests/test_gpu_parity.py, not the bits of a previous build.  The correctly rounded forms
// (ocml expf, IEEE division and sqrt: ~30 dependent instructions per updated register) were 0.8 - 1.2 k
// cycles of a 3.3 - 3.7 k-cycle Adam step (profiles/r3/fit_marks_final.txt).  The L-BFGS-B (fp64,
// lbfgsb.h) and the objective evaluation keep their IEEE forms.
// (fit_rcp, fit_sqrt, fit_exp_neg, fit_elu: mlp_device.h -- the register network of mlp_regs.h uses them too)

// Adam update of one parameter (ResourceApplyAdam, non-nesterov); returns the new weight.
__device__ __forceinline__ float adam_update(float w, float g, float &m, float &v, float alpha,
                                             float omb1, float omb2, float eps) {
  m += (g - m) * omb1;
  v += (g * g - v) * omb2;
  return fmaf(-(m * alpha), fit_rcp(fit_sqrt(v) + eps), w);
}

// Weight gradients + Adam for a static shape, NBLK = row-blocks of the batch that hold live
// rows (the other blocks' rows are never read).  Same sums as the generic loop in fit_kernel
// (k-ordered over the batch rows), but with every trip count a constant: the 2 x 4 NBLK MFMA
// operands and the task's theta / m / v slots are all requested up front, so one LDS latency
// is paid per task instead of one per group of k-chunks.  The Adam slots are in LDS (the host
// only picks a static flavour when they fit).
//
// One TASK = one 16x16 tile of some dW_l, or half of its registers; tasks go round-robin over
// the four waves.  What bounds a task is the serial sqrt/divide chain per register it updates,
// not its MFMAs, so:
//   * a layer with ONE unit is formed transposed (dW_l^T = D_l^T A_{l-1}): its 16 gradients land
//     in one register of lanes 0..15 and the bias rides along in lane 16 -- 1 chain, not 5;
//   * when the net has fewer tiles than the workgroup has waves, a full tile becomes two tasks
//     (registers {0,1} + bias | {2,3}); both redo the tile's MFMAs, on different SIMDs.
__device__ __forceinline__ void dw_task(const MlpLayout &L, float *smem, int o_tile, int o_m,
                                        int o_v, int l, int kb, int cb, int r_lo, int r_hi,
                                        bool want_bias, bool transposed, int kch, float alpha,
                                        float omb1, float omb2, float eps) {
  const int lane = threadIdx.x & 63, m16 = lane & 15, q4 = lane >> 4;
  float *th = smem, *tile = smem + o_tile, *sm = smem + o_m, *sv = smem + o_v;
  const int K = L.w[l - 1], Nw = L.w[l], ldw = L.ldw[l];
  const int lda_p = L.lda[l - 1], ldd = L.lda[l];
  const float *ap = tile + L.aoff[l - 1] + q4 * lda_p + kb * 16 + m16;
  const float *bp = tile + L.doff[l] + q4 * ldd + cb * 16 + m16;
  float av[4 * (BORE_BATCH_MAX / 16)], bv[4 * (BORE_BATCH_MAX / 16)];
  FIT_MARK_DECL;
  BORE_TSTAMP(8);
#pragma unroll
  for (int kc = 0; kc < kch; ++kc) {
    av[kc] = ap[kc * 4 * lda_p];
    bv[kc] = bp[kc * 4 * ldd];
  }
  // the slots this lane updates: index 0..3 = registers of the tile, 4 = the bias
  int li[5];
  bool ok[5];
  const int col = cb * 16 + m16;
  if (transposed) {  // register 0 of lanes 0..15 = dW_l[kb*16 + m16][0]; lane 16 = the bias
    const bool is_b = want_bias && lane == 16;
    li[0] = is_b ? L.boff[l] : L.woff[l] + (kb * 16 + m16) * ldw;
    ok[0] = is_b || (q4 == 0 && kb * 16 + m16 < K);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      li[r] = L.woff[l] + (kb * 16 + q4 * 4 + r) * ldw + col;
      ok[r] = col < Nw && kb * 16 + q4 * 4 + r < K;
    }
    li[4] = L.boff[l] + col;
    ok[4] = q4 == 0 && col < Nw;
  }
  float w[5], mm[5], vv[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const bool used = transposed ? r == 0 : (r == 4 ? want_bias : (r >= r_lo && r < r_hi));
    if (!used) continue;
    w[r] = th[li[r]];
    mm[r] = sm[li[r]];
    vv[r] = sv[li[r]];
  }
  // (the scheduler would otherwise re-interleave loads and MFMAs pair by pair, paying the LDS
  // latency kch / 2 times)
  __builtin_amdgcn_sched_barrier(0);
  BORE_TSTAMP(9);
  FIT_MARK(8);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
#pragma unroll
  for (int kc = 0; kc < kch; ++kc) {
    acc = transposed ? __builtin_amdgcn_mfma_f32_16x16x4f32(bv[kc], av[kc], acc, 0, 0, 0)
                     : __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc], bv[kc], acc, 0, 0, 0);
    bsum += bv[kc];
  }
  // (round 4, tried and dropped: 2 - 4 interleaved partial sums instead of one chain of kch dependent
  // matrix instructions -- fit 586 -> 582 us at N ~ 100, no change at N ~ 25, 14 spilled registers)
  float g[5] = {acc[0], acc[1], acc[2], acc[3], 0.f};
  BORE_TSTAMP(10);
  FIT_MARK(9);
  if (want_bias) {
    const float gb = rows_sum4(bsum);  // column sums of D_l (every lane of the column)
    if (transposed) g[0] = lane == 16 ? gb : g[0];
    else g[4] = gb;
  }
  // branch-free: a lane updates all of the task's slots (padding slots work on zeros and are
  // not stored), so the dependent sqrt/divide chains interleave
  float wn[5];
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const bool used = transposed ? r == 0 : (r == 4 ? want_bias : (r >= r_lo && r < r_hi));
    if (!used) continue;
    wn[r] = adam_update(w[r], g[r], mm[r], vv[r], alpha, omb1, omb2, eps);
  }
#pragma unroll
  for (int r = 0; r < 5; ++r) {
    const bool used = transposed ? r == 0 : (r == 4 ? want_bias : (r >= r_lo && r < r_hi));
    if (!used) continue;
    if (!ok[r]) continue;
    th[li[r]] = wn[r];
    sm[li[r]] = mm[r];
    sv[li[r]] = vv[r];
  }
  BORE_TSTAMP(11);
  FIT_MARK(10);
}

template <int SHAPE, int NBLK>
__device__ __forceinline__ void dw_adam_static(const FitArgs &a, float *smem, float alpha,
                                               float omb1, float omb2) {
  constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  constexpr int KCH = 4 * NBLK;
  int n_tiles = 0;
#pragma unroll
  for (int l = 1; l <= L.n_layers; ++l) n_tiles += (L.Np[l - 1] >> 4) * (L.Np[l] >> 4);
  const bool split = n_tiles < BORE_THREADS / 64;
  // (the wave number as a SCALAR: the task tests below are then scalar compares and branches, not
  // vector compares that narrow the execution mask -- the way from the mid-step barrier to the
  // wave's task was ~380 cycles of an Adam step, profiles/r3/fit_marks_dispatch.txt)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  // (tasks go round the workgroup's waves: four, or eight in fit_kernel_w8 -- a power of two)
  const int wmask = __builtin_amdgcn_readfirstlane((int)(blockDim.x >> 6)) - 1;
  int t = 0;
#pragma unroll
  for (int l = 1; l <= L.n_layers; ++l) {
    const int nkb = L.Np[l - 1] >> 4, ncb = L.Np[l] >> 4;
#pragma unroll
    for (int kb = 0; kb < nkb; ++kb)
#pragma unroll
      for (int cb = 0; cb < ncb; ++cb) {
        const bool transposed = L.w[l] == 1;
        const bool two = !transposed && split && L.w[l - 1] - kb * 16 > 2;
        if ((t & wmask) == wv)
          // (registers past the layer's inputs hold padding: a 2-input layer updates two, not four; a tile
          // split into two tasks: the bias rides with the SECOND half -- the first half's wave was the
          // heavier one: fit 392.7 -> 385.8 us per loop-iteration.  Tried beyond that and dropped: the bias
          // of a narrow layer in the idle lanes of register 0, the split tile's bias as second register of
          // the transposed task -- an unpaired register costs ~340 cycles, but the fused kernel paid for the
          // extra operands in spills: profiles/r3/ab_headline.txt)
          dw_task(L, smem, a.o_tile, a.o_m, a.o_v, l, kb, cb, 0,
                  two ? 2 : (L.w[l - 1] - kb * 16 < 4 ? L.w[l - 1] - kb * 16 : 4), kb == 0 && !two, transposed,
                  KCH, alpha, omb1, omb2, a.eps);
        ++t;
        if (two) {
          if ((t & wmask) == wv)
            dw_task(L, smem, a.o_tile, a.o_m, a.o_v, l, kb, cb, 2, 4, kb == 0, false, KCH, alpha,
                    omb1, omb2, a.eps);
          ++t;
        }
      }
  }
}

// Weight gradients + Adam for a WIDE static shape (mlp_shapes.h: 64-wide hidden layers): too
// many tiles to unroll (40 for 16->64-64-64-1), so the wave walks its tiles t = wv, wv + 4, ...
// in a run-time loop; per tile the same plan as dw_task -- all 2 x 16 MFMA operands and the
// tile's theta / m / v slots requested up front (m, v from HBM when they do not fit in LDS: the
// loads then fly under the MFMA chain), k-ordered MFMA sum over the 64 batch rows (rows past
// the batch hold zeros), bias = column sums, branch-free Adam, one-unit layers transposed.
// where one weight-gradient tile of a wide static shape reads and writes
struct WideTile {
  int K, Nw, ldw, lda_p, ldd, aoff, doff, kb, cb;
  int li[5], gi[5];  // LDS / packed-HBM index of the slots this lane updates (4 = the bias)
  bool ok[5], transposed, want_bias;
};

template <int SHAPE>
__device__ __forceinline__ WideTile wide_tile(int t, int lane_in = -1) {
  constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  const int lane = lane_in >= 0 ? lane_in : (int)(threadIdx.x & 63), m16 = lane & 15, q4 = lane >> 4;
  WideTile w;
  // tile t -> (layer, kb, cb); every per-layer quantity is picked by a scalar compare chain
  int woff = 0, boff = 0, goff_w = 0, goff_b = 0, ncb = 1, r = t, start = 0;
  w.K = w.Nw = w.ldw = w.lda_p = w.ldd = w.aoff = w.doff = 0;
#pragma unroll
  for (int l = 1; l <= L.n_layers; ++l) {
    const int nt = (L.Np[l - 1] >> 4) * (L.Np[l] >> 4);
    if (t >= start && t < start + nt) {
      w.K = L.w[l - 1]; w.Nw = L.w[l]; w.ldw = L.ldw[l]; w.lda_p = L.lda[l - 1]; w.ldd = L.lda[l];
      w.aoff = L.aoff[l - 1]; w.doff = L.doff[l]; woff = L.woff[l]; boff = L.boff[l];
      goff_w = L.goff_w[l]; goff_b = L.goff_b[l]; ncb = L.Np[l] >> 4; r = t - start;
    }
    start += nt;
  }
  w.kb = r / ncb;
  w.cb = r - w.kb * ncb;
  w.transposed = w.Nw == 1;
  w.want_bias = w.kb == 0;
  const int col = w.cb * 16 + m16;
  if (w.transposed) {  // register 0 of lanes 0..15 = dW_l[kb*16 + m16][0]; lane 16 = the bias
    const bool is_b = w.want_bias && lane == 16;
    w.li[0] = is_b ? boff : woff + (w.kb * 16 + m16) * w.ldw;
    w.gi[0] = is_b ? goff_b : goff_w + (w.kb * 16 + m16) * w.Nw;
    w.ok[0] = is_b || (q4 == 0 && w.kb * 16 + m16 < w.K);
#pragma unroll
    for (int q = 1; q < 5; ++q) { w.li[q] = w.li[0]; w.gi[q] = w.gi[0]; w.ok[q] = false; }
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      w.li[q] = woff + (w.kb * 16 + q4 * 4 + q) * w.ldw + col;
      w.gi[q] = goff_w + (w.kb * 16 + q4 * 4 + q) * w.Nw + col;
      w.ok[q] = col < w.Nw && w.kb * 16 + q4 * 4 + q < w.K;
    }
    w.li[4] = boff + col;
    w.gi[4] = goff_b + col;
    w.ok[4] = w.want_bias && q4 == 0 && col < w.Nw;
  }
  return w;
}

// ---------------------------------------------------------------------------------------------
// Weight gradients + Adam of a WIDE static shape whose Adam slots live in HBM, in two phases.
//
// Round 1 updated tile by tile: each tile's m / v (and, for the mixed-precision fit, master weight)
// slots made an HBM round trip that one tile of look-ahead (~1 k cycles of MFMAs)
// does not cover -- measured 4.3 k cycles per tile for 16->64-64-64-1 (40 tiles: 43 k of the
// step's 65 k cycles) and 5.8 k for 32->128-128-1 in bf16 (88 tiles).  Here
//   A. every wave forms ALL of its gradient tiles back to back (pure LDS + MFMA; the 16x16
//      results stay in registers: 5 per tile),
//   B. after a barrier (the A / D copies are dead) the gradients go to LDS in PACKED parameter
//      order, over the tile region,
//   C. after another barrier all 256 threads walk the packed vector: g from LDS, m / v (/ master
//      weight) from HBM with coalesced loads, a dozen elements in flight per thread -- one memory
//      latency per dozen elements instead of one per tile.
// Same sums, same Adam expression per element: results are bit-identical to the tile-by-tile form.
// ---------------------------------------------------------------------------------------------
template <int SHAPE>
constexpr int wide_total_tiles() {
  constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  int total = 0;
  for (int l = 1; l <= L.n_layers; ++l) total += (L.Np[l - 1] >> 4) * (L.Np[l] >> 4);
  return total;
}
template <int SHAPE>
constexpr int wide_tiles_per_wave() {
  return (wide_total_tiles<SHAPE>() + BORE_THREADS / 64 - 1) / (BORE_THREADS / 64);
}

// Phase A: G[i][q] = gradient slot q of this wave's i-th tile (t = wave + 4 i).  ET = element type
// of the A / D images (float, or unsigned short holding bfloat16).  A tile's 2 x 16 operands are
// requested one tile ahead of its MFMA chain (raw: the widening of a bfloat16 happens beside the
// MFMAs, not in front of them).
template <int SHAPE, typename ET>
__device__ __forceinline__ void wide_grads(const ET *tile, float (&G)[wide_tiles_per_wave<SHAPE>()][5],
                                           int tid_o) {
  // (tid_o: an opaque copy of the thread id made inside the caller's step loop -- from the
  // loop-invariant id the per-tile indices of all tiles are hoisted out of that loop and spilled)
  constexpr int KCH = BORE_BATCH_MAX / 4, TOTAL = wide_total_tiles<SHAPE>();
  constexpr int TPW = wide_tiles_per_wave<SHAPE>(), STEP = BORE_THREADS / 64;
  const int wv = __builtin_amdgcn_readfirstlane(tid_o >> 6), lane = tid_o & 63, m16 = lane & 15, q4 = lane >> 4;
  auto widen = [](ET raw) -> float {
    if constexpr (sizeof(ET) == 2) return bf16_to_f32(raw);
    else return raw;
  };
  ET av[2][KCH], bv[2][KCH];
  {
    const WideTile w0 = wide_tile<SHAPE>(wv < TOTAL ? wv : 0, lane);
    const ET *ap = tile + w0.aoff + q4 * w0.lda_p + w0.kb * 16 + m16;
    const ET *bp = tile + w0.doff + q4 * w0.ldd + w0.cb * 16 + m16;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      av[0][kc] = ap[kc * 4 * w0.lda_p];
      bv[0][kc] = bp[kc * 4 * w0.ldd];
    }
  }
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wv + STEP * i;
#pragma unroll
    for (int q = 0; q < 5; ++q) G[i][q] = 0.f;
    if (TOTAL % STEP != 0 && t >= TOTAL) continue;
    const WideTile cur = wide_tile<SHAPE>(t, lane);
    if (i + 1 < TPW) {  // the next tile's operands (clamped: a wave past the end re-reads its last)
      const WideTile nx = wide_tile<SHAPE>(t + STEP < TOTAL ? t + STEP : t, lane);
      const ET *ap = tile + nx.aoff + q4 * nx.lda_p + nx.kb * 16 + m16;
      const ET *bp = tile + nx.doff + q4 * nx.ldd + nx.cb * 16 + m16;
#pragma unroll
      for (int kc = 0; kc < KCH; ++kc) {
        av[(i + 1) & 1][kc] = ap[kc * 4 * nx.lda_p];
        bv[(i + 1) & 1][kc] = bp[kc * 4 * nx.ldd];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
#pragma unroll
    for (int kc = 0; kc < KCH; ++kc) {
      const float a = widen(av[i & 1][kc]), b = widen(bv[i & 1][kc]);
      const float x = cur.transposed ? b : a, y = cur.transposed ? a : b;
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc, 0, 0, 0);
      bsum += b;
    }
    G[i][0] = acc[0]; G[i][1] = acc[1]; G[i][2] = acc[2]; G[i][3] = acc[3];
    if (cur.want_bias) {
      const float gb = rows_sum4(bsum);
      if (cur.transposed) G[i][0] = lane == 16 ? gb : G[i][0];
      else G[i][4] = gb;
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Phase B: the gradients to gl[packed parameter index].
template <int SHAPE>
__device__ __forceinline__ void wide_scatter(const float (&G)[wide_tiles_per_wave<SHAPE>()][5], float *gl,
                                             int tid_o) {
  constexpr int TOTAL = wide_total_tiles<SHAPE>(), TPW = wide_tiles_per_wave<SHAPE>();
  constexpr int STEP = BORE_THREADS / 64;
  const int wv = __builtin_amdgcn_readfirstlane(tid_o >> 6), lane = tid_o & 63;
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    const int t = wv + STEP * i;
    if (TOTAL % STEP != 0 && t >= TOTAL) continue;
    const WideTile cur = wide_tile<SHAPE>(t, lane);
#pragma unroll
    for (int q = 0; q < 5; ++q)
      if (cur.ok[q]) gl[cur.gi[q]] = G[i][q];
    // (tile by tile: hoisted, the index arithmetic of all tiles is live at once -- 22 x 10 registers)
    __builtin_amdgcn_sched_barrier(0);
  }
}

// Phase C: all threads walk the packed parameter vector in batches of U elements per thread.  The
// NEXT batch's loads are issued before this batch's arithmetic and stores: gfx950 counts loads and
// stores on one counter (vmcnt), so a load issued after a store cannot be waited for without
// waiting for the store's round trip as well -- batch by batch that was two memory latencies per U
// elements (64 k of the 163 k cycles of a 32->128-128-1 step, profiles/r2 wide_stamps).
// MASTER: float32 master weights packed in HBM beside m / v, LDS image bfloat16 (mixed precision);
// otherwise theta is the float32 LDS image.
template <int SHAPE, bool MASTER>
__device__ __forceinline__ void wide_adam(void *th_lds, const float *gl, float *theta_g, float *m_g,
                                          float *v_g, float alpha, float omb1, float omb2, float eps,
                                          int tid_o) {
  constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  constexpr int U = 8, STRIDE = BORE_THREADS * U;
  constexpr int NB = (L.P + STRIDE - 1) / STRIDE;
  float *th = reinterpret_cast<float *>(th_lds);
  unsigned short *th16 = reinterpret_cast<unsigned short *>(th_lds);
  float g[2][U], mm[2][U], vv[2][U], w[2][U];
  int li[2][U];
  auto request = [&](int b, int buf) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = b * STRIDE + BORE_THREADS * u + tid_o;
      const bool ok = p < L.P;
      li[buf][u] = param_ref(L, ok ? p : 0, L.n_layers).lds;
      mm[buf][u] = ok ? m_g[p] : 0.f;
      vv[buf][u] = ok ? v_g[p] : 0.f;
      g[buf][u] = ok ? gl[p] : 0.f;
      if constexpr (MASTER) w[buf][u] = ok ? theta_g[p] : 0.f;
      else w[buf][u] = th[li[buf][u]];
    }
  };
  request(0, 0);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int cur = b & 1;
    if (b + 1 < NB) request(b + 1, cur ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    float wn[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      wn[u] = adam_update(w[cur][u], g[cur][u], mm[cur][u], vv[cur][u], alpha, omb1, omb2, eps);
#pragma unroll
    for (int u = 0; u < U; ++u) asm volatile("" : "+v"(wn[u]), "+v"(mm[cur][u]), "+v"(vv[cur][u]));
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int p = b * STRIDE + BORE_THREADS * u + tid_o;
      if (p < L.P) {
        if constexpr (MASTER) {
          th16[li[cur][u]] = f32_to_bf16(wn[u]);
          theta_g[p] = wn[u];
        } else {
          th[li[cur][u]] = wn[u];
        }
        m_g[p] = mm[cur][u];
        v_g[p] = vv[cur][u];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ---------------------------------------------------------------------------------------------
// The float32 fit of a WIDE static shape (16->64-64-64-1), round-2 form.  Same arithmetic as before
// (v_mfma_f32_16x16x4_f32 chains k-ordered over the batch rows, IEEE Adam): bit-identical results.
// What changed is how the weight-gradient phase is fed and how its results are used:
//  * the copies of A_l / D_l that the gradients sum over are stored TRANSPOSED and row-permuted,
//    img[unit][(row & 3) * 16 + (row >> 2)] with a 68-float pitch: lane (q, m)'s sixteen operands
//    of a tile -- rows 4 kc + q, kc = 0..15, of unit 16 kb + m -- are then contiguous: four 16-byte
//    LDS reads per operand instead of sixteen 4-byte ones (the pitch keeps the 16 lanes of a read
//    on distinct banks);
//  * the four waves' i-th tiles lie in ONE layer, known at compile time (static_for): no per-tile
//    layer dispatch;
//  * the wave that formed a tile updates it; m and v are held in the MFMA's own result layout for the
//    launch (TileOrder, fit_bf16_mfma.h), so they are ONE 16-byte load and store each per tile and lane,
//    1 KiB contiguous per wave, requested three tiles ahead (and ahead of the stores in between:
//    loads and stores share a counter).
// ---------------------------------------------------------------------------------------------
template <int SHAPE>
struct WideTp {
  static constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers, PITCH = 68;
  static constexpr int a_off(int l) {  // image of A_l, l = 0..n-1 (float offset inside the tile region)
    int o = 0;
    for (int i = 0; i < l; ++i) o += L.Np[i] * PITCH;
    return o;
  }
  static constexpr int d_off(int l) {  // image of D_l, l = 1..n
    int o = a_off(n);
    for (int i = 1; i < l; ++i) o += L.Np[i] * PITCH;
    return o;
  }
  static constexpr int total() { return d_off(n + 1); }
  static __host__ __device__ constexpr int idx(int unit, int row) {
    return unit * PITCH + (row & 3) * 16 + (row >> 2);
  }
  static constexpr int tiles_before(int l) {
    int t = 0;
    for (int i = 1; i < l; ++i) t += (L.Np[i - 1] >> 4) * (L.Np[i] >> 4);
    return t;
  }
  static constexpr int total_tiles() { return tiles_before(n + 1); }
  static constexpr int layer_of_tile(int t) {
    for (int l = 1; l <= n; ++l)
      if (t < tiles_before(l + 1)) return l;
    return n;
  }
  static constexpr bool aligned() {
    for (int l = 1; l <= n + 1; ++l)
      if (tiles_before(l) % 4 != 0) return false;
    return true;
  }
};

// C-layout registers of a layer (lane (q, m): row m of the wave's block, units 16t + 4q + r) -> its
// transposed image
template <int SHAPE, int NP, int TMAX>
__device__ __forceinline__ void store_rows_tp(const float (&src)[TMAX][4], float *img, int row) {
  const int q = (threadIdx.x & 63) >> 4;
#pragma unroll
  for (int t = 0; t < NP / 16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) img[WideTp<SHAPE>::idx(16 * t + 4 * q + r, row)] = src[t][r];
}

template <int SHAPE>
__device__ __forceinline__ void wide_fused_f32(float *th, const float *img, float *m_g, float *v_g,
                                               float alpha, float omb1, float omb2, float eps, int tid_o) {
  using Tp = WideTp<SHAPE>;
  constexpr MlpLayout L = Tp::L;
  constexpr int TOTAL = Tp::total_tiles(), TPW = TOTAL / 4, AHEAD = 3, PITCH = Tp::PITCH;
  static_assert(Tp::aligned() && TOTAL % 4 == 0, "tiles of a layer must start at a multiple of 4");
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int wv = __builtin_amdgcn_readfirstlane(tid_o >> 6), lane = tid_o & 63, m16 = lane & 15, q4 = lane >> 4;
  constexpr int RING = AHEAD + 2;  // (tile I - 1 is still in use when tile I + AHEAD is requested)
  f4u pm[RING], pv[RING];
  float bm[RING], bv[RING];
  // m / v are kept in tile order for the launch (TileOrder, fit_bf16_mfma.h: the MFMA's result layout,
  // tile (kb, cb) = 256 consecutive floats) and addressed as buffers: index = a wave-uniform part pu
  // (scalar) + the lane's byte offset lb4; pbu / 4 m16 = the same for the bias.
  constexpr int P = L.P;
  const BufF32 b_m(m_g, P), b_v(v_g, P);
  const unsigned lbb = 4u * (unsigned)m16;
  auto slots = [&](auto ic, int &kb, int &cb, int &pu, unsigned &lb4, bool &ok4, int &pbu, bool &okb) {
    constexpr int I = decltype(ic)::value, l = Tp::layer_of_tile(4 * I);
    constexpr int K = L.w[l - 1], Nw = L.w[l], ncb = L.Np[l] >> 4;
    constexpr bool FULL = K % 16 == 0 && Nw % 16 == 0;
    const int r = wv + 4 * I - Tp::tiles_before(l);
    kb = r / ncb;
    cb = r - kb * ncb;
    if constexpr (Nw == 1) {  // one column: the C layout's four rows ARE contiguous (lanes m = 0)
      pu = L.goff_w[l] + 16 * kb;
      lb4 = 16u * (unsigned)q4;
      ok4 = m16 == 0 && 16 * kb + 4 * q4 < K;
    } else {  // tile order: tile (kb, cb) = 256 consecutive floats, the lane's four at 4 * lane
      static_assert(FULL, "a wide static shape has layer sizes that are multiples of 16");
      pu = L.goff_w[l] + (kb * ncb + cb) * 256;
      lb4 = 16u * (unsigned)lane;
      ok4 = true;
    }
    okb = kb == 0 && q4 == 0 && (FULL || 16 * cb + m16 < Nw);
    pbu = L.goff_b[l] + 16 * cb;
  };
  auto request = [&](auto ic) {
    constexpr int I = decltype(ic)::value, l = Tp::layer_of_tile(4 * I);
    constexpr bool FULL = L.w[l - 1] % 16 == 0 && L.w[l] % 16 == 0;
    int kb, cb, pu, pbu;
    unsigned lb4;
    bool ok4, okb;
    slots(ic, kb, cb, pu, lb4, ok4, pbu, okb);
    const f4u z4 = {0.f, 0.f, 0.f, 0.f};
    if constexpr (FULL) {
      pm[I % RING] = b_m.ld4(pu, lb4);
      pv[I % RING] = b_v.ld4(pu, lb4);
    } else {
      pm[I % RING] = ok4 ? b_m.ld4(pu, lb4) : z4;
      pv[I % RING] = ok4 ? b_v.ld4(pu, lb4) : z4;
    }
    if (kb == 0) {  // (wave-uniform: only these tiles carry a bias)
      bm[I % RING] = okb ? b_m.ld1(pbu, lbb) : 0.f;
      bv[I % RING] = okb ? b_v.ld1(pbu, lbb) : 0.f;
    }
  };
  static_for<0, (AHEAD < TPW ? AHEAD : TPW)>([&](auto ic) { request(ic); });
  // Software pipeline: the MFMA chain of tile I (16 dependent matrix instructions, 32 cycles each,
  // 8 of them issue) shares its scheduling region with the Adam arithmetic of tile I - 1, which
  // fills the chain's gaps instead of following it.
  f32x4 acc_prev = {0.f, 0.f, 0.f, 0.f};
  float bsum_prev = 0.f;
  auto finish = [&](auto ic, const f32x4 &acc, float bsum) {  // quad transpose, Adam, stores of tile I
    constexpr int I = decltype(ic)::value, l = Tp::layer_of_tile(4 * I);
    constexpr int Nw = L.w[l], ldw = L.ldw[l];
    int kb, cb, pu, pbu;
    unsigned lb4;
    bool ok4, okb;
    slots(ic, kb, cb, pu, lb4, ok4, pbu, okb);
    // g[r] = dW_l[16kb + 4q + r][16cb + m]: the tile order of m / v IS this layout
    const float g[4] = {acc[0], acc[1], acc[2], acc[3]};
    // this lane's four weights in the padded LDS image of theta
    static_assert(Nw == 1 || (L.w[l - 1] % 16 == 0 && Nw % 16 == 0), "tile-order layers only");
    const int li0 = Nw == 1 ? L.woff[l] + (16 * kb + 4 * q4) * ldw : L.woff[l] + (16 * kb + 4 * q4) * ldw + 16 * cb + m16;
    constexpr int lstep = ldw;
    constexpr int cur = I % RING;
    float w[4], wn[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w[r] = ok4 ? th[li0 + r * lstep] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float mm = pm[cur][r], vv = pv[cur][r];
      wn[r] = adam_update(w[r], g[r], mm, vv, alpha, omb1, omb2, eps);
      pm[cur][r] = mm;
      pv[cur][r] = vv;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(wn[r]));
    if (ok4) {
      b_m.st4(pm[cur], pu, lb4);
      b_v.st4(pv[cur], pu, lb4);
#pragma unroll
      for (int r = 0; r < 4; ++r) th[li0 + r * lstep] = wn[r];
    }
    if (kb == 0) {  // bias: the column sums of D_l (every lane of the column holds them)
      const float gb = rows_sum4(bsum);
      const int lb = L.boff[l] + 16 * cb + m16;
      float mm = bm[cur], vv = bv[cur];
      const float wb0 = okb ? th[lb] : 0.f;
      const float wnb = adam_update(wb0, gb, mm, vv, alpha, omb1, omb2, eps);
      if (okb) {
        th[lb] = wnb;
        b_m.st1(mm, pbu, lbb);
        b_v.st1(vv, pbu, lbb);
      }
    }
  };
  static_for<0, TPW + 1>([&](auto ic) {
    constexpr int I = decltype(ic)::value;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float bsum = 0.f;
    if constexpr (I < TPW) {
      constexpr int l = Tp::layer_of_tile(4 * I);
      // (tile I + AHEAD's slots must be requested after tile I - 1's stores were... no: the slots of
      // different tiles are disjoint; the request only has to precede its use by a few tiles)
      if constexpr (I + AHEAD < TPW) request(std::integral_constant<int, I + AHEAD>{});
      int kb, cb, pu, pbu;
      unsigned lb4;
      bool ok4, okb;
      slots(ic, kb, cb, pu, lb4, ok4, pbu, okb);
      // operands: rows 4 kc + q4 (kc = 0..15) of unit 16 kb|cb + m16 = sixteen consecutive floats
      // (16-byte aligned: pitch 272 B, image offsets multiples of 16 B -> ds_read_b128)
      const float4 *ap = reinterpret_cast<const float4 *>(img + Tp::a_off(l - 1) + (16 * kb + m16) * PITCH + 16 * q4);
      const float4 *bp = reinterpret_cast<const float4 *>(img + Tp::d_off(l) + (16 * cb + m16) * PITCH + 16 * q4);
      float av[4][4], bq[4][4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float4 x = ap[c], y = bp[c];
        av[c][0] = x.x; av[c][1] = x.y; av[c][2] = x.z; av[c][3] = x.w;
        bq[c][0] = y.x; bq[c][1] = y.y; bq[c][2] = y.z; bq[c][3] = y.w;
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kc = 0; kc < 16; ++kc) {
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[kc >> 2][kc & 3], bq[kc >> 2][kc & 3], acc, 0, 0, 0);
        bsum += bq[kc >> 2][kc & 3];
      }
    }
    if constexpr (I > 0) finish(std::integral_constant<int, I - 1>{}, acc_prev, bsum_prev);
    if constexpr (I < TPW) {  // one MFMA, then a handful of the other tile's vector instructions
#pragma unroll
      for (int kc = 0; kc < 16; ++kc) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    acc_prev = acc;
    bsum_prev = bsum;
  });
}

// ---------------------------------------------------------------------------------------------
// The float32 fit of 32->128-128-1: theta (84 KB) and the 64-row A / D images (152 KB) do not share
// a CU's LDS.  Forward and backward still run on all four waves, 16 rows each, in registers; the
// weight gradients are then formed in FOUR ROUNDS: in round r wave r publishes ITS 16 rows of
// A_l / D_l (transposed, 20-float pitch: 45 KB), every wave adds those rows' four k-chunks to the
// accumulators of its own tiles (22 tiles x 4 registers, kept across the rounds), and after the
// last round updates them.  Rows enter every chain in the order 0..63, as in the one-pass form.
// ---------------------------------------------------------------------------------------------
template <int SHAPE>
struct WideTp16 {
  static constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  static constexpr int n = L.n_layers, PITCH = 20;
  static constexpr int a_off(int l) {
    int o = 0;
    for (int i = 0; i < l; ++i) o += L.Np[i] * PITCH;
    return o;
  }
  static constexpr int d_off(int l) {
    int o = a_off(n);
    for (int i = 1; i < l; ++i) o += L.Np[i] * PITCH;
    return o;
  }
  static constexpr int total() { return d_off(n + 1); }
  static __host__ __device__ constexpr int idx(int unit, int row) { return unit * PITCH + (row & 3) * 4 + (row >> 2); }
};
// shapes whose one-pass images do not fit beside theta
static constexpr bool bore_shape_fit_in_rounds(int shape) {
  return shape == 4;
}

template <int SHAPE, typename Net>
__device__ __forceinline__ void wide_rounds_f32(const Net &net, const float (&xin)[Net::KC0], float delta,
                                                float *th, float *img, float *m_g, float *v_g, float alpha,
                                                float omb1, float omb2, float eps, int tid_o) {
  using Tp = WideTp<SHAPE>;
  using T16 = WideTp16<SHAPE>;
  constexpr MlpLayout L = Tp::L;
  constexpr int n = L.n_layers, TOTAL = Tp::total_tiles(), TPW = TOTAL / 4, PITCH = T16::PITCH, P = L.P;
  static_assert(Tp::aligned() && TOTAL % 4 == 0, "tiles of a layer must start at a multiple of 4");
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  const int wv = __builtin_amdgcn_readfirstlane(tid_o >> 6), lane = tid_o & 63, m16 = lane & 15, q4 = lane >> 4;
  f32x4 acc[TPW];
  float bsum[TPW];
#pragma unroll
  for (int i = 0; i < TPW; ++i) {
    acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bsum[i] = 0.f;
  }
  auto tile_of = [&](auto ic, int &kb, int &cb) {
    constexpr int I = decltype(ic)::value, l = Tp::layer_of_tile(4 * I);
    constexpr int ncb = L.Np[l] >> 4;
    const int r = wv + 4 * I - Tp::tiles_before(l);
    kb = r / ncb;
    cb = r - kb * ncb;
  };
  for (int round = 0; round < 4; ++round) {
    if (wv == round) {  // this wave's 16 rows become the shared images: row m16 of the round
#pragma unroll
      for (int kc = 0; kc < Net::KC0; ++kc)
        if (4 * kc + q4 < L.Np[0]) img[T16::a_off(0) + T16::idx(4 * kc + q4, m16)] = xin[kc];
      static_for<1, n>([&](auto lc) {
        constexpr int l = decltype(lc)::value;
#pragma unroll
        for (int t = 0; t < L.Np[l] / 16; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            img[T16::a_off(l) + T16::idx(16 * t + 4 * q4 + r, m16)] = net.h[l][t][r];
            img[T16::d_off(l) + T16::idx(16 * t + 4 * q4 + r, m16)] = net.d[l][t][r];
          }
      });
      if (lane < 16) img[T16::d_off(n) + T16::idx(0, m16)] = delta;
    }
    __syncthreads();
    static_for<0, TPW>([&](auto ic) {
      constexpr int I = decltype(ic)::value, l = Tp::layer_of_tile(4 * I);
      int kb, cb;
      tile_of(ic, kb, cb);
      // rows 4 kc + q4 (kc = 0..3) of unit 16 kb|cb + m16: four consecutive floats
      const float4 av = *reinterpret_cast<const float4 *>(img + T16::a_off(l - 1) + (16 * kb + m16) * PITCH + 4 * q4);
      const float4 bq = *reinterpret_cast<const float4 *>(img + T16::d_off(l) + (16 * cb + m16) * PITCH + 4 * q4);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bq.x, acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bq.y, acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, bq.z, acc[I], 0, 0, 0);
      acc[I] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, bq.w, acc[I], 0, 0, 0);
      bsum[I] += bq.x;
      bsum[I] += bq.y;
      bsum[I] += bq.z;
      bsum[I] += bq.w;
    });
    __syncthreads();  // (the next round's wave overwrites the images)
  }
  // update: m / v in tile order (the MFMA result layout), theta in its LDS image
  const BufF32 b_m(m_g, P), b_v(v_g, P);
  const unsigned lbb = 4u * (unsigned)m16;
  static_for<0, TPW>([&](auto ic) {
    constexpr int I = decltype(ic)::value, l = Tp::layer_of_tile(4 * I);
    constexpr int K = L.w[l - 1], Nw = L.w[l], ncb = L.Np[l] >> 4, ldw = L.ldw[l];
    static_assert(Nw == 1 || (K % 16 == 0 && Nw % 16 == 0), "tile-order layers only");
    int kb, cb;
    tile_of(ic, kb, cb);
    int pu;
    unsigned lb4;
    bool ok4;
    if constexpr (Nw == 1) {
      pu = L.goff_w[l] + 16 * kb;
      lb4 = 16u * (unsigned)q4;
      ok4 = m16 == 0 && 16 * kb + 4 * q4 < K;
    } else {
      pu = L.goff_w[l] + (kb * ncb + cb) * 256;
      lb4 = 16u * (unsigned)lane;
      ok4 = true;
    }
    const bool okb = kb == 0 && q4 == 0 && 16 * cb + m16 < Nw;
    const int pbu = L.goff_b[l] + 16 * cb;
    const int li0 = Nw == 1 ? L.woff[l] + (16 * kb + 4 * q4) * ldw : L.woff[l] + (16 * kb + 4 * q4) * ldw + 16 * cb + m16;
    const f4u z4 = {0.f, 0.f, 0.f, 0.f};
    f4u pm = ok4 ? b_m.ld4(pu, lb4) : z4, pv = ok4 ? b_v.ld4(pu, lb4) : z4;
    const float g[4] = {acc[I][0], acc[I][1], acc[I][2], acc[I][3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float w = ok4 ? th[li0 + r * ldw] : 0.f;
      float mm = pm[r], vv = pv[r];
      const float wn = adam_update(w, g[r], mm, vv, alpha, omb1, omb2, eps);
      pm[r] = mm;
      pv[r] = vv;
      if (ok4) th[li0 + r * ldw] = wn;
    }
    if (ok4) {
      b_m.st4(pm, pu, lb4);
      b_v.st4(pv, pu, lb4);
    }
    if (kb == 0) {  // bias: the column sums of D_l
      const float gb = rows_sum4(bsum[I]);
      const int lb = L.boff[l] + 16 * cb + m16;
      float mm = okb ? b_m.ld1(pbu, lbb) : 0.f, vv = okb ? b_v.ld1(pbu, lbb) : 0.f;
      const float wb0 = okb ? th[lb] : 0.f;
      const float wnb = adam_update(wb0, gb, mm, vv, alpha, omb1, omb2, eps);
      if (okb) {
        th[lb] = wnb;
        b_m.st1(mm, pbu, lbb);
        b_v.st1(vv, pbu, lbb);
      }
    }
  });
}

// (the body is a device function so that the fused iteration kernel of bore_iter.hip can run it)
template <int SHAPE, int NW = 4>
__device__ __forceinline__ void fit_body(const FitArgs &a, const long long slot,
                                         const int it_now = -1) {
  extern __shared__ float smem[];
  constexpr MlpLayout Lc = bore_static_layout(SHAPE > 0 ? SHAPE : 0, 1, BORE_BATCH_MAX);
  constexpr bool WIDE = bore_shape_is_wide(SHAPE > 0 ? SHAPE : 0);
  constexpr bool ROUNDS = bore_shape_fit_in_rounds(SHAPE > 0 ? SHAPE : 0);  // (wide_rounds_f32)
  if constexpr (WIDE) {  // m / v (in HBM for a wide net) go to tile order for the launch (TileOrder)
    const long long mdl = a.ids ? uniform_i64(a.ids[slot]) : slot;
    TileOrder<WIDE ? SHAPE : 1>::convert(a.am + mdl * Lc.P, smem, true);
    TileOrder<WIDE ? SHAPE : 1>::convert(a.av + mdl * Lc.P, smem, true);
  }
  const MlpLayout &L = begin_kernel<SHAPE>(Lc, a.L, smem, a.total, a.o_layout);
  int tid = threadIdx.x;
  BORE_OPAQUE_TID(tid);
  const int nthr = blockDim.x;
  const int wv = tid >> 6, lane = tid & 63, m16 = lane & 15, q4 = lane >> 4;
  const long long model = a.ids ? uniform_i64(a.ids[slot]) : slot;  // the loop this workgroup fits
  const int P = L.P, n = layer_count<SHAPE>(L), D = L.w[0];
  const int it_cur = a.ids ? (it_now >= 0 ? it_now : uniform_i32(a.its[slot])) : 0;
  const int N = a.ids ? a.n_init + it_cur : a.N;
  const long long epoch0 = a.ids ? (long long)it_cur * a.epochs : a.epoch0;

  float *th = smem;
  float *tile = smem + a.o_tile;
  float *zt = smem + a.o_zt;
  float *misc = smem + a.o_misc;  // [0] l2 penalty of the current weights, [1..4] per-wave loss
  int *perm_all = reinterpret_cast<int *>(smem + a.o_perm);
  int *perm_s = perm_all;  // the current epoch's permutation
  int *perm_two = reinterpret_cast<int *>(smem + a.o_perm2);  // (perm_ahead: the other epoch's)
  unsigned *keys = reinterpret_cast<unsigned *>(smem + a.o_keys);
  const int PG = a.perm ? 1 : perm_group(N, BORE_THREADS);

  float *theta_g = a.theta + model * P;
  float *m_g = a.am + model * P;
  float *v_g = a.av + model * P;
  const float *X_g = a.X + model * (a.ids ? a.cap : (long long)N) * D;
  const float *z_g = a.z + model * (a.ids ? a.cap : (long long)N);
  float *sm = smem + a.o_m, *sv = smem + a.o_v;  // padded images (when state_in_lds)
  float *gacc = smem + a.o_g;  // weight-gradient sums carried between sub-tiles (batch_size > 64)

  load_theta(L, n, theta_g, th);
  if (a.state_in_lds) {
    load_theta(L, n, m_g, sm);
    load_theta(L, n, v_g, sv);
  }
  if (a.data_in_lds) {
    for (int i = tid; i < N * D; i += nthr) smem[a.o_X + i] = X_g[i];
    for (int i = tid; i < N; i += nthr) smem[a.o_z + i] = z_g[i];
  }
  __syncthreads();
  if (L.any_l2) {  // l2 penalty of the incoming weights (what the first step's loss sees)
    float reg = 0.f;
    for (int p = tid; p < P; p += nthr) {
      const ParamRef r = param_ref(L, p, n);
      const float l2 = r.k >= 0 ? L.l2_w[r.l] : L.l2_b[r.l];
      const float w = th[r.lds];
      reg = fmaf(l2 * w, w, reg);
    }
    reg = wave_sum(reg);
    if (lane == 0) atomicAdd(&misc[0], reg);
  }

  // running beta powers in fp64 (rounded to fp32 at use; see DESIGN.md "Adam")
  const long long t0 = a.at[model];
  double b1p = pow((double)a.beta1, (double)t0);
  double b2p = pow((double)a.beta2, (double)t0);
  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  const int steps = (N + a.B - 1) / a.B;
  // Shuffles and row gathers off the step chain: when the LAST step of an epoch leaves the fourth wave
  // without rows (at most 48 rows in it; static shapes whose waves own their row-blocks), that wave
  // draws and ranks the permutation of the epoch AFTER the next during the step (make_perm_wave) into
  // the buffer of the current epoch's, which nobody reads any more: every step requests the NEXT
  // step's rows (permutation entry -> row of X, z) beside its own weight operands and parks each
  // lane's share in that lane's slot of `stage`, so that a step starts from one LDS round trip with
  // loop-invariant addresses instead of the perm -> row -> operand chain.  (The slot is private to
  // its lane: no synchronisation; LDS rather than a register carried around the loop, which the
  // compiler rotated with copies that waited for the request at once.)  Same permutations, same rows.
  // Round 6: the eight-wave kernel too -- its fifth wave never has rows, so every data set of up to 128 rows qualifies;
  // before, it drew the shuffles of two epochs with the whole workgroup at the top of every other epoch: 2.1 k cycles
  // per epoch of 16->32-32-32-1 at N 100, 1.07 k of a 13.5 k-cycle step (profiles/r6/ab_log.txt).
  const bool pipe_perm = SHAPE > 0 && !WIDE && !a.perm && PG >= 2 && a.data_in_lds &&
                         ((blockDim.x == BORE_THREADS && N - (steps - 1) * a.B <= 16 * (BORE_THREADS / 64 - 1)) ||
                          (NW == 8 && blockDim.x == 2 * BORE_THREADS));
  const int draw_wave = NW == 8 && blockDim.x == 2 * BORE_THREADS ? BORE_THREADS / 64 : BORE_THREADS / 64 - 1;
  // The shuffles of the pipelined form are drawn by the fourth wave, two epochs ahead, during an epoch's
  // last step (make_perm_wave_buckets: ~1 k cycles at 100 rows; round 3's all-pairs count took ~6 k there,
  // more than a step's front half, and the workgroup drew the shuffles of 65..112 rows at the top of every
  // odd epoch instead -- 2.2 k cycles per epoch on the step chain).  (Tried in round 3: the pipelined form
  // also for last steps of 49..64 rows, drawn by the workgroup -- 623 against 593 us per fit at N 48..67,
  // profiles/r3/ab_headline.txt: those keep the four-epoch groups and the gather in the step.)
  // Eight-wave kernel, more than 128 rows (one shuffle per epoch, 4 k cycles of the whole workgroup on the step
  // chain): the fifth wave, which owns no rows, draws the NEXT epoch's shuffle alone, a third of it in each of the
  // epoch's first three steps while the first four run the forward / backward pass (make_perm_buckets<true>) -- off
  // the chain.
  const bool ahead = NW == 8 && a.perm_ahead && !a.perm && PG == 1 && blockDim.x == 2 * BORE_THREADS;
  constexpr int PRE_KC = RegNet<(SHAPE > 0 ? SHAPE : 1), 1>::KC0;
  static_assert(WIDE || SHAPE <= 0 || (PRE_KC + 1) * BORE_THREADS <= BORE_FIT_STAGE_FLOATS_OF(SHAPE), "stage region");
  // [PRE_KC + 1][BORE_THREADS]: inputs 4 kc + q4, then the label (the waves past the fourth own no rows and park none)
  float *stage = smem + a.o_stage + (tid & (BORE_THREADS - 1));
  // (pipe_perm) request this lane's share of row `srow`: every address valid, dead shares zeroed at the store
  auto request_row = [&](float (&gx)[PRE_KC], float &gz, const int srow) {  // (pipe_perm: the data is in LDS)
#pragma unroll
    for (int kc = 0; kc < PRE_KC; ++kc) gx[kc] = smem[a.o_X + srow * D + min(4 * kc + q4, D - 1)];
    gz = smem[a.o_z + srow];
  };
  auto park_row = [&](const float (&gx)[PRE_KC], const float gz, const bool live_row) {
#pragma unroll
    for (int kc = 0; kc < PRE_KC; ++kc)
      stage[kc * BORE_THREADS] = (4 * kc + q4 < D && live_row) ? gx[kc] : 0.f;
    stage[PRE_KC * BORE_THREADS] = (q4 == 0 && live_row) ? gz : 0.f;
  };
  // Step size of Adam step t: lr * sqrt(1 - beta2^t) / (1 - beta1^t).  Every wave forms the
  // first one; after that the last wave computes the NEXT step's while it waits at the end of
  // the weight-gradient phase and leaves it in misc[5] (one sqrt + divide per step per
  // workgroup instead of per wave, and off the step's critical path).
  b1p *= (double)a.beta1;
  b2p *= (double)a.beta2;
  const float alpha_first = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
  bool first_step = true;
  __syncthreads();
#ifdef BORE_FIT_MARKS
  if (lane < 32) g_fit_lds[wv & 7][lane] = 0;
#endif
  FIT_MARK_DECL;

  if (pipe_perm) {  // the first two epochs' shuffles by everybody (ends with a barrier); the first step's rows
    make_perm_group(a.seed, a.model0 + model, epoch0, min(2, a.epochs), N, keys, perm_all);
    const bool live0 = wv * 16 + m16 < min(a.B, N);
    float gx[PRE_KC], gz;
    request_row(gx, gz, perm_all[live0 ? wv * 16 + m16 : 0]);
    if (wv < BORE_THREADS / 64) park_row(gx, gz, live0);
  }

  // (the epochs in two instantiations -- rows parked ahead or gathered in the step -- chosen once: tested
  // inside the step, the two forms met in blocks where the compiler waited for every pending request)
  auto run_epochs = [&](auto pipe_c) {
  constexpr bool PIPE = decltype(pipe_c)::value;
  for (int e = 0; e < a.epochs; ++e) {
    FIT_MARK(12);
    if constexpr (PIPE) {  // (first: the test every step of the headline run takes)
      // two shuffle buffers: epoch e reads buffer e & 1; during its last step the fourth wave draws epoch
      // e + 2's into the same buffer, which nobody reads any more (that step's rows were parked a step ago)
      perm_s = perm_all + (e & 1) * N;
    } else if (a.perm) {
      if (a.perm_in_lds) {
        const int *pg = a.perm + (model * a.epochs + e) * (long long)N;
        for (int i = tid; i < N; i += nthr) perm_s[i] = pg[i];
        __syncthreads();
      }
    } else if (PG > 1) {  // small data set: the shuffles of PG consecutive epochs at once
      const int eg = e & (PG - 1);  // PG is 2 or 4
      if (eg == 0)
        make_perm_group(a.seed, a.model0 + model, epoch0 + e, min(PG, a.epochs - e), N, keys,
                        perm_all);
      perm_s = perm_all + eg * N;
    } else if (ahead) {  // two buffers; every epoch but the first was drawn during the one before it (below)
      perm_s = (e & 1) ? perm_two : perm_all;
      if (e == 0) make_perm(shuffle_base(a.seed, a.model0 + model, epoch0), N, keys, perm_s);
    } else {
      make_perm(shuffle_base(a.seed, a.model0 + model, epoch0 + e), N, keys, perm_s);
    }
    float eloss = 0.f;  // per lane: sum over the epoch of the losses of its row slot
    FIT_MARK(13);

    for (int s = 0; s < steps; ++s) {
      FIT_MARK(14);
      const int row0 = s * a.B;
      const int nb = min(a.B, N - row0);
      const float inv_nb = fit_rcp((float)nb);  // (wave-uniform; the mean over the step's rows as a multiply)
      const float alpha = first_step ? alpha_first : misc[5];
      first_step = false;
      float reg = 0.f;
      // A mini-batch of more than 64 rows (Keras takes any batch_size; flavours without a
      // compile-time layout) is walked in 64-row SUB-TILES: forward / loss / backward per sub-tile,
      // the weight-gradient sums carried from one sub-tile to the next in an LDS image (the MFMA
      // chain of a tile starts from the partial sum, so the k-ordered sum runs over all rows of the
      // batch), Adam once after the last.  One sub-tile = the path every batch_size <= 64 takes.
      const int nsub = SHAPE > 0 ? 1 : (nb + BORE_BATCH_MAX - 1) / BORE_BATCH_MAX;
      for (int sub = 0; sub < nsub; ++sub) {
      const int r0 = row0 + sub * BORE_BATCH_MAX;  // the sub-tile's first row in the epoch's order
      const int nr = SHAPE > 0 ? nb : min(BORE_BATCH_MAX, nb - sub * BORE_BATCH_MAX);
      const bool first_sub = sub == 0, last_sub = sub + 1 == nsub;
      // ---- forward / loss / backward: wave wv owns rows [16 wv, 16 wv + 16), no barriers ----
      BORE_STAMP(0);
      FIT_MARK(7);
      FIT_MARK(11);  // (two marks back to back: what a mark costs)
      BORE_WSTAMP_DECL;
      int src = 0;  // static path: this lane's mini-batch row, requested ahead of the arithmetic below
      bool live_n = false;  // (PIPE: src is the NEXT step's row of this lane, live_n whether it has one)
      float cx[PRE_KC], cz = 0.f;  // (PIPE) this step's row, parked during the last step
      if constexpr (SHAPE > 0) {
        if constexpr (PIPE) {
#pragma unroll
          for (int kc = 0; kc < PRE_KC; ++kc) cx[kc] = stage[kc * BORE_THREADS];
          cz = stage[PRE_KC * BORE_THREADS];
          const bool last_s = s == steps - 1;
          const int *perm_n = last_s ? perm_all + ((e + 1) & 1) * N : perm_s;
          const int row0n = last_s ? 0 : row0 + a.B;
          const int nbn = last_s && e + 1 >= a.epochs ? 0 : min(a.B, N - row0n);
          live_n = wv * 16 + m16 < nbn;
          src = perm_n[live_n ? row0n + wv * 16 + m16 : 0];
        } else if (wv * 16 + m16 < nb) {
          src = perm_s[row0 + wv * 16 + m16];
        }
      }
      // (wide shapes: every wave runs, rows past the batch are dead -- x = 0, delta = 0 -- so
      // that the weight-gradient tiles can always sum over all 64 rows)
      if (wv * 16 < nr || (bore_shape_is_wide(SHAPE > 0 ? SHAPE : 0) && wv < BORE_THREADS / 64)) {
        const int rb = wv;
        if constexpr (SHAPE > 0) {
          // static shape: the row-block's activations and deltas stay in registers
          // (mlp_regs.h); LDS only receives the copies the weight-gradient phase sums over
          using Net = RegNet<SHAPE, 1>;
          Net net;
          if constexpr (Net::RT_ACT) net.set_acts(a.L);
          // the weight operands do not depend on the gather: request them first, so that they
          // are in flight under the perm -> row -> A_0 chain below
          net.load_fwd(th);
          net.template load_bwd<Net::n, 2>(th);
          const int row = rb * 16 + m16;
          const bool live = row < nb;
          static_assert(Net::KC0 == PRE_KC, "the parked row has the first layer's k-chunks");
          float xin[Net::KC0];
          float *A0 = tile + L.aoff[0] + row * L.lda[0];
#pragma unroll
          for (int kc = 0; kc < Net::KC0; ++kc) xin[kc] = 0.f;
          float zz = 0.f;
          float nx[PRE_KC], nz = 0.f;  // (PIPE) the next step's row, on its way
          if constexpr (PIPE) {  // this step's row was parked during the last one; request the next step's
#pragma unroll
            for (int kc = 0; kc < Net::KC0; ++kc) xin[kc] = cx[kc];
            zz = cz;
            // (opaque here: the row's address is the same expression in the branch of the waves without
            // rows, and hoisted in front of both it waited for `src` before the weight requests above)
            int srow = src;
            asm volatile("" : "+v"(srow));
            request_row(nx, nz, srow);
          } else
          if (a.data_in_lds) {  // (two branches: a selected pointer would make these flat loads)
#pragma unroll
            for (int kc = 0; kc < Net::KC0; ++kc)
              if (4 * kc + q4 < D && live) xin[kc] = smem[a.o_X + src * D + 4 * kc + q4];
            if (q4 == 0 && live) zz = smem[a.o_z + src];
          } else {
#pragma unroll
            for (int kc = 0; kc < Net::KC0; ++kc)
              if (4 * kc + q4 < D && live) xin[kc] = X_g[src * D + 4 * kc + q4];
            if (q4 == 0 && live) zz = z_g[src];
          }
#pragma unroll
          for (int kc = 0; kc < Net::KC0; ++kc)
            if (4 * kc + q4 < D) {
              if constexpr (ROUNDS) {
              } else if constexpr (bore_shape_is_wide(SHAPE)) tile[WideTp<SHAPE>::a_off(0) + WideTp<SHAPE>::idx(4 * kc + q4, row)] = xin[kc];
              else A0[4 * kc + q4] = xin[kc];
            }
          __builtin_amdgcn_sched_barrier(0);  // every operand load is in flight before the chain
          BORE_STAMP(1);
          FIT_MARK(0);
          BORE_WSTAMP(0);
          net.forward(th, xin, /*keep_logits=*/true);
          BORE_STAMP(2);
          FIT_MARK(1);
          BORE_WSTAMP(1);
          if constexpr (ROUNDS) {  // (the rows stay in registers until their round: wide_rounds_f32)
          } else if constexpr (bore_shape_is_wide(SHAPE)) {
            static_for<1, Net::n>([&](auto lc) {
              constexpr int l = decltype(lc)::value;
              store_rows_tp<SHAPE, Net::L.Np[l], Net::T>(net.h[l], tile + WideTp<SHAPE>::a_off(l), row);
            });
          } else {
            net.template store_A<1, Net::n - 1>(tile, rb);
          }
          float delta = 0.f;
          if (lane < 16 && live) {
            const float x = net.h[Net::n][0][0];
            const float ex = fit_exp_neg(-fabsf(x));
            const float rden = fit_rcp(1.f + ex);
            const float sig = x >= 0.f ? rden : ex * rden;
            if (a.epoch_loss) eloss += fmaxf(x, 0.f) - x * zz + log1pf(ex);
            delta = (sig - zz) * inv_nb;
          }
          if (lane < 16) {
            if constexpr (ROUNDS) {
            } else if constexpr (bore_shape_is_wide(SHAPE)) tile[WideTp<SHAPE>::d_off(Net::n) + WideTp<SHAPE>::idx(0, row)] = delta;
            else tile[L.doff[Net::n] + row * L.lda[Net::n]] = delta;
          }
          net.set_output_delta(delta);
          BORE_STAMP(3);
          FIT_MARK(2);
          BORE_WSTAMP(2);
          net.template backward<Net::n, 2>(th);
          BORE_WSTAMP(3);
          if constexpr (ROUNDS) {
            int tid_r = tid;  // (opaque per step, as for the other wide fits)
            asm volatile("" : "+v"(tid_r));
            wide_rounds_f32<SHAPE>(net, xin, delta, th, tile, m_g, v_g, alpha, omb1, omb2, a.eps, tid_r);
          } else if constexpr (bore_shape_is_wide(SHAPE)) {
            static_for<1, Net::n>([&](auto lc) {
              constexpr int l = decltype(lc)::value;
              store_rows_tp<SHAPE, Net::L.Np[l], Net::T>(net.d[l], tile + WideTp<SHAPE>::d_off(l), row);
            });
          } else {
            net.template store_D<1, Net::n - 1>(tile, rb);
          }
          if constexpr (PIPE) park_row(nx, nz, live_n);
          BORE_STAMP(4);
          FIT_MARK(3);
          BORE_WSTAMP(4);
        } else {
        {  // gather the mini-batch rows of this row-block (rows past the sub-tile: zeros)
          float *A0 = tile + L.aoff[0] + (rb * 16 + m16) * L.lda[0];
          const int row = rb * 16 + m16;
          // (a data set whose shuffle does not fit in LDS: the explicit permutation is read from
          // memory, 64 indices per step)
          const int src = row < nr ? (a.perm_in_lds ? perm_s[r0 + row]
                                                    : a.perm[(model * a.epochs + e) * (long long)N + r0 + row])
                                   : 0;
          if (a.data_in_lds) {  // (two branches: a selected pointer would make these flat loads)
            for (int d = q4; d < D; d += 4) A0[d] = row < nr ? smem[a.o_X + src * D + d] : 0.f;
            if (q4 == 0) zt[row] = row < nr ? smem[a.o_z + src] : 0.f;
          } else {
            for (int d = q4; d < D; d += 4) A0[d] = row < nr ? X_g[src * D + d] : 0.f;
            if (q4 == 0) zt[row] = row < nr ? z_g[src] : 0.f;
          }
        }
        wave_lds_sync();
        fwd_all<true>(L, n, th, tile, rb, /*keep_logits=*/true);
        if (lane < 16) {  // loss + d loss / d logit (the final layer has one unit)
          const int row = rb * 16 + lane;
          float delta = 0.f;
          if (row < nr) {
            const float x = tile[L.aoff[n] + row * L.lda[n]];
            const float zz = zt[row];
            const float ex = fit_exp_neg(-fabsf(x));  // shared by the loss and the sigmoid
            const float rden = fit_rcp(1.f + ex);
            const float sig = x >= 0.f ? rden : ex * rden;
            if (a.epoch_loss)  // per-lane; reduced once per epoch
              eloss += fmaxf(x, 0.f) - x * zz + log1pf(ex);
            delta = (sig - zz) * inv_nb;
          }
          tile[L.doff[n] + row * L.lda[n]] = delta;
        }
        wave_lds_sync();
#pragma unroll
        for (int l = n; l >= 2; --l) {
          bwd_rowblock(L, th, tile, l, rb);
          wave_lds_sync();
        }
        }
        if (tid == 0 && L.any_l2 && first_sub) {
          eloss += misc[0] * (float)nb;
          misc[0] = 0.f;  // consumed; re-accumulated from the updated weights below
        }
      } else if constexpr (PIPE) {  // a wave without rows in this step may have some in the next
        if (wv < BORE_THREADS / 64) {
          float gx[PRE_KC], gz;
          request_row(gx, gz, src);
          park_row(gx, gz, live_n);
        }
        if (wv == draw_wave && s == steps - 1 && e + 2 < a.epochs) {
          // (an opaque copy keeps the epoch's hash in THIS wave's branch: wave-uniform scalar code is
          // otherwise hoisted in front of every wave's step)
          long long draw_epoch = epoch0 + e + 2;
          asm volatile("" : "+v"(draw_epoch));
          // (up to 64 rows: one row per lane and N compares each -- cheaper than the buckets' fixed cost)
          if (N <= 64)
            make_perm_wave(shuffle_base(a.seed, a.model0 + model, draw_epoch), N,
                           reinterpret_cast<unsigned long long *>(keys), perm_all + (e & 1) * N);
          else
            make_perm_wave_buckets(shuffle_base(a.seed, a.model0 + model, draw_epoch), N,
                                   reinterpret_cast<unsigned long long *>(keys), perm_all + (e & 1) * N);
        }
      } else {
        if (ahead && wv == BORE_THREADS / 64 && s < 3 && first_sub && e + 1 < a.epochs) {  // (> 128 rows: >= 3 steps)
          long long draw_epoch = epoch0 + e + 1;  // (opaque: the hash stays in this wave's branch, as above)
          asm volatile("" : "+v"(draw_epoch));
          make_perm_buckets<true>(shuffle_base(a.seed, a.model0 + model, draw_epoch), N, keys,
                                  ((e + 1) & 1) ? perm_two : perm_all, s, s);
        }
      }
      __syncthreads();
#ifdef BORE_STAMPS
      if (tid == 0) g_stamp_on = (e == 1 && s == 0);
      __syncthreads();
#endif
      BORE_STAMP(5);
      FIT_MARK(4);

      // ---- weight gradients (sums over all rows) + Adam, one 16x16 tile per wave at a time ----
      const int kch = (nr + 3) >> 2;
      const int wmask_g = (int)(blockDim.x >> 6) - 1;
      int t = 0;
      if constexpr (ROUNDS) {
        // (gradients and update ran inside the row-block scope above: wide_rounds_f32)
      } else if constexpr (bore_shape_is_wide(SHAPE > 0 ? SHAPE : 0)) {
        // (a wide net's m / v never fit in LDS beside theta and the 64-row images: fit_build)
        int tid_o = tid;  // opaque per step: nothing derived from it is hoisted out of the step loop
        asm volatile("" : "+v"(tid_o));
        BORE_WSTAMP(5);
        wide_fused_f32<SHAPE>(th, tile, m_g, v_g, alpha, omb1, omb2, a.eps, tid_o);
        BORE_WSTAMP(6);
      } else if constexpr (SHAPE > 0) {
        switch ((nb + 15) >> 4) {
          case 1: dw_adam_static<SHAPE, 1>(a, smem, alpha, omb1, omb2); break;
          case 2: dw_adam_static<SHAPE, 2>(a, smem, alpha, omb1, omb2); break;
          case 3: dw_adam_static<SHAPE, 3>(a, smem, alpha, omb1, omb2); break;
          default: dw_adam_static<SHAPE, 4>(a, smem, alpha, omb1, omb2); break;
        }
        FIT_MARK(15);  // (bucket 15: the mid-step barrier -> the wave's task done)
      } else
#pragma unroll
      for (int l = 1; l <= n; ++l) {
        const int K = L.w[l - 1], Nw = L.w[l], ldw = L.ldw[l];
        const int lda_p = L.lda[l - 1], ldd = L.lda[l];
        const int nkb = L.Np[l - 1] >> 4, ncb = L.Np[l] >> 4;
#pragma unroll
        for (int kb = 0; kb < nkb; ++kb)
#pragma unroll
          for (int cb = 0; cb < ncb; ++cb, ++t) {
            if ((t & wmask_g) != wv) continue;  // (round-robin over the workgroup's four or eight waves)
            // dW[k][j] = sum_rows A_{l-1}[row][k] * D_l[row][j]
            const float *ap = tile + L.aoff[l - 1] + q4 * lda_p + kb * 16 + m16;
            const float *bp = tile + L.doff[l] + q4 * ldd + cb * 16 + m16;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            float bsum = 0.f;  // this lane's share of the column sums of D_l (rows q4, q4 + 4, ..)
            const bool want_bias = kb == 0;
            const int col = cb * 16 + m16;
            const bool cvalid = col < Nw;
            if (!first_sub) {  // go on from the sums over the batch's earlier sub-tiles
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = kb * 16 + q4 * 4 + r;
                if (cvalid && row < K) acc[r] = gacc[L.woff[l] + row * ldw + col];
              }
            }
            // Adam slots that live in HBM (wide nets): fetch this tile's m, v now, so that the
            // loads are in flight under the MFMA chain instead of in front of every update
            float pm[4] = {0.f, 0.f, 0.f, 0.f}, pv[4] = {0.f, 0.f, 0.f, 0.f}, pmb = 0.f, pvb = 0.f;
            if (!a.state_in_lds && last_sub) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = kb * 16 + q4 * 4 + r;
                if (cvalid && row < K) {
                  pm[r] = m_g[L.goff_w[l] + row * Nw + col];
                  pv[r] = v_g[L.goff_w[l] + row * Nw + col];
                }
              }
              if (want_bias && q4 == 0 && cvalid) {
                pmb = m_g[L.goff_b[l] + col];
                pvb = v_g[L.goff_b[l] + col];
              }
            }
            int kc = 0;
            for (; kc + 4 <= kch; kc += 4) {  // operands of four row-chunks in flight
              const float a0 = ap[kc * 4 * lda_p], a1 = ap[(kc + 1) * 4 * lda_p],
                          a2 = ap[(kc + 2) * 4 * lda_p], a3 = ap[(kc + 3) * 4 * lda_p];
              const float b0 = bp[kc * 4 * ldd], b1 = bp[(kc + 1) * 4 * ldd],
                          b2 = bp[(kc + 2) * 4 * ldd], b3 = bp[(kc + 3) * 4 * ldd];
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a2, b2, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a3, b3, acc, 0, 0, 0);
              bsum = (((bsum + b0) + b1) + b2) + b3;
            }
            for (; kc < kch; ++kc) {
              const float av = ap[kc * 4 * lda_p], bv = bp[kc * 4 * ldd];
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
              bsum += bv;
            }
            if (!last_sub) {  // park the partial sums; Adam comes after the batch's last sub-tile
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int row = kb * 16 + q4 * 4 + r;
                if (cvalid && row < K) gacc[L.woff[l] + row * ldw + col] = acc[r];
              }
              if (want_bias) {
                const float gb = rows_sum4(bsum);
                if (q4 == 0 && cvalid)
                  gacc[L.boff[l] + col] = first_sub ? gb : gacc[L.boff[l] + col] + gb;
              }
              continue;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = kb * 16 + q4 * 4 + r;
              if (cvalid && row < K) {
                const int li = L.woff[l] + row * ldw + col;
                float w = th[li];
                float g = acc[r];
                const float l2 = L.l2_w[l];
                if (l2 != 0.f) g = fmaf(2.f * l2, w, g);
                if (a.state_in_lds) {
                  float mm = sm[li], vv = sv[li];
                  w = adam_update(w, g, mm, vv, alpha, omb1, omb2, a.eps);
                  sm[li] = mm;
                  sv[li] = vv;
                } else {
                  const int gi = L.goff_w[l] + row * Nw + col;
                  float mm = pm[r], vv = pv[r];
                  w = adam_update(w, g, mm, vv, alpha, omb1, omb2, a.eps);
                  m_g[gi] = mm;
                  v_g[gi] = vv;
                }
                th[li] = w;
                if (l2 != 0.f) reg = fmaf(l2 * w, w, reg);
              }
            }
            if (want_bias) {  // bias gradient: the column sums of D_l
              const float gb = rows_sum4(bsum);
              if (q4 == 0 && cvalid) {
                const int li = L.boff[l] + col;
                float w = th[li];
                float g = first_sub ? gb : gacc[li] + gb;
                const float l2 = L.l2_b[l];
                if (l2 != 0.f) g = fmaf(2.f * l2, w, g);
                if (a.state_in_lds) {
                  float mm = sm[li], vv = sv[li];
                  w = adam_update(w, g, mm, vv, alpha, omb1, omb2, a.eps);
                  sm[li] = mm;
                  sv[li] = vv;
                } else {
                  const int gi = L.goff_b[l] + col;
                  float mm = pmb, vv = pvb;
                  w = adam_update(w, g, mm, vv, alpha, omb1, omb2, a.eps);
                  m_g[gi] = mm;
                  v_g[gi] = vv;
                }
                th[li] = w;
                if (l2 != 0.f) reg = fmaf(l2 * w, w, reg);
              }
            }
          }
      }
      if (!last_sub) __syncthreads();  // (the next sub-tile overwrites the A_l / D_l copies)
      }  // sub-tiles
      if (L.any_l2) {  // penalty of the UPDATED weights = the one the next step's loss sees
        reg = wave_sum(reg);
        if (lane == 0) atomicAdd(&misc[0], reg);
      }
      if (wv == (BORE_THREADS / 64) - 1) {  // the next step's size (read after the barrier below)
        b1p *= (double)a.beta1;
        b2p *= (double)a.beta2;
        const float an = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
        if (lane == 0) misc[5] = an;
      }
      BORE_STAMP(6);
      FIT_MARK(5);
      __syncthreads();
      BORE_STAMP(7);
      FIT_MARK(6);
    }
    if (a.epoch_loss) {
      eloss = wave_sum(eloss);
      if (lane == 0 && wv < 4) misc[1 + wv] = eloss;  // (waves past the fourth own no rows: fit_kernel_w8)
      __syncthreads();
      if (tid == 0)
        a.epoch_loss[model * a.epochs + e] = (misc[1] + misc[2] + misc[3] + misc[4]) / (float)N;
    }
  }
  };
  if constexpr (SHAPE > 0 && !WIDE) {
    if (pipe_perm) run_epochs(std::true_type{});
    else run_epochs(std::false_type{});
  } else {
    run_epochs(std::false_type{});
  }

  __syncthreads();
  store_theta(L, n, th, theta_g);
  if (a.state_in_lds) {
    store_theta(L, n, sm, m_g);
    store_theta(L, n, sv, v_g);
  }
  if constexpr (WIDE) {  // m / v back to the packed order (the LDS copy of theta is stored)
    __syncthreads();
    TileOrder<WIDE ? SHAPE : 1>::convert(m_g, smem, false);
    TileOrder<WIDE ? SHAPE : 1>::convert(v_g, smem, false);
  }
  if (tid == 0) a.at[model] = t0 + (long long)a.epochs * steps;
#ifdef BORE_FIT_MARKS
  if (lane < 32) atomicAdd(&g_fit_acc[wv & 7][lane], (unsigned long long)g_fit_lds[wv & 7][lane]);
#endif
}

template <int SHAPE>
__global__ __launch_bounds__(BORE_THREADS) void fit_kernel(const FitArgs a) {
  fit_body<SHAPE>(a, blockIdx.x);
}
// Eight waves for the same workgroup (static shapes with two weight-gradient tiles per wave: 6->32-32-1): rows
// and row-blocks are still the first four waves'; the other four wait at the mid-step barrier and then take half
// of the weight-gradient tasks -- one tile per wave instead of two in a row (a task is a chain of LDS requests,
// dependent matrix instructions and sqrt / reciprocal chains that one wave walks alone).  For launches with no
// more models than compute units: with more, two four-wave workgroups per CU do better.
template <int SHAPE>
__global__ __launch_bounds__(2 * BORE_THREADS) void fit_kernel_w8(const FitArgs a) {
  fit_body<SHAPE, 8>(a, blockIdx.x);
}

// ---------------------------------------------------------------------------
// mixed-precision fit (BASELINE config 5: "128-128-1 MLP bf16 ... fused Adam"): bf16 weights and
// activations, fp32 accumulate, fp32 master weights + Adam.  Wide static shapes only.
//
// LDS holds the bf16 image of theta (same padded tile layout, 2-byte elements) and the bf16
// copies of A_l / D_l for the weight-gradient tiles: half the bytes of the fp32 fit, which is
// what lets a 128-wide net fit at all (fp32: 248 KB > 160 KB).  The fp32 master weights, m and
// v stay in HBM (wide_grads / wide_scatter / wide_adam below), updated in fp32
// and written back together with the tile's new bf16 image.  Products run on the fp32 MFMA
// (bf16 x bf16 is exact in fp32; k-ordered fp32 sums) -- the step is bound by the Adam chains
// and the barrier structure, not by MFMA rate.  Every layer output, logit and delta is rounded
// to bf16 (RegNet<.., BF16>); loss and d loss / d logit are fp32.
// ---------------------------------------------------------------------------
struct FitBf16Args {
  MlpLayout L;  // run-time layout (activations)
  float *theta, *am, *av;
  long long *at;
  const float *X, *z;
  const int *perm;
  float *epoch_loss;
  unsigned long long seed;
  long long model0, epoch0;
  int N, epochs, B;
  float lr, beta1, beta2, eps;
  int o_tile, o_misc, o_perm, o_keys, total;  // BYTE offsets into the dynamic LDS
};

template <int SHAPE>
__global__ __launch_bounds__(BORE_THREADS) void fit_bf16_kernel(const FitBf16Args a) {
  extern __shared__ float smem[];
  constexpr MlpLayout L = bore_static_layout(SHAPE, 1, BORE_BATCH_MAX);
  using Net = RegNet<SHAPE, 1, true>;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wv = tid >> 6, lane = tid & 63, m16 = lane & 15, q4 = lane >> 4;
  const long long model = blockIdx.x;
  constexpr int P = L.P, n = L.n_layers, D = L.w[0];
  const int N = a.N;
  for (int i = tid; i < (a.total >> 2); i += nthr) smem[i] = 0.f;
  __syncthreads();
  char *base = reinterpret_cast<char *>(smem);
  unsigned short *th16 = reinterpret_cast<unsigned short *>(base);
  unsigned short *tile16 = reinterpret_cast<unsigned short *>(base + a.o_tile);
  float *misc = reinterpret_cast<float *>(base + a.o_misc);
  int *perm_all = reinterpret_cast<int *>(base + a.o_perm);
  int *perm_s = perm_all;
  unsigned *keys = reinterpret_cast<unsigned *>(base + a.o_keys);
  const int PG = a.perm ? 1 : perm_group(N, BORE_THREADS);

  float *theta_g = a.theta + model * P;
  float *m_g = a.am + model * P;
  float *v_g = a.av + model * P;
  const float *X_g = a.X + model * (long long)N * D;
  const float *z_g = a.z + model * (long long)N;
  for (int p = tid; p < P; p += nthr) th16[param_ref(L, p, n).lds] = f32_to_bf16(theta_g[p]);

  const long long t0 = a.at[model];
  double b1p = pow((double)a.beta1, (double)t0);
  double b2p = pow((double)a.beta2, (double)t0);
  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  const int steps = (N + a.B - 1) / a.B;
  b1p *= (double)a.beta1;
  b2p *= (double)a.beta2;
  const float alpha_first = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
  bool first_step = true;
  __syncthreads();
#ifdef BORE_FIT_MARKS
  if (lane < 32) g_fit_lds[wv & 7][lane] = 0;
#endif
  FIT_MARK_DECL;

  for (int e = 0; e < a.epochs; ++e) {
    if (a.perm) {
      const int *pg = a.perm + (model * a.epochs + e) * (long long)N;
      for (int i = tid; i < N; i += nthr) perm_s[i] = pg[i];
      __syncthreads();
    } else if (PG > 1) {
      const int eg = e & (PG - 1);
      if (eg == 0)
        make_perm_group(a.seed, a.model0 + model, a.epoch0 + e, min(PG, a.epochs - e), N, keys,
                        perm_all);
      perm_s = perm_all + eg * N;
    } else {
      make_perm(shuffle_base(a.seed, a.model0 + model, a.epoch0 + e), N, keys, perm_s);
    }
    float eloss = 0.f;
    for (int s = 0; s < steps; ++s) {
      const int row0 = s * a.B;
      const int nb = min(a.B, N - row0);
      const float inv_nb = fit_rcp((float)nb);  // (wave-uniform; the mean over the step's rows as a multiply)
      const float alpha = first_step ? alpha_first : misc[5];
      first_step = false;
      BORE_WSTAMP_DECL;
      {  // every wave runs its 16 rows; rows past the batch are dead (x = 0, delta = 0)
        Net net;
        if constexpr (Net::RT_ACT) net.set_acts(a.L);
        const int rb = wv, row = rb * 16 + m16;
        const bool live = row < nb;
        const int src = live ? perm_s[row0 + row] : 0;
        float xin[Net::KC0];
        unsigned short *A0 = tile16 + L.aoff[0] + row * L.lda[0];
#pragma unroll
        for (int kc = 0; kc < Net::KC0; ++kc) {
          const int d = 4 * kc + q4;
          float x = 0.f;
          if (d < D && live) x = bf16_round(X_g[src * D + d]);
          xin[kc] = x;
          if (d < D) A0[d] = f32_to_bf16(x);
        }
        float zz = 0.f;
        if (q4 == 0 && live) zz = z_g[src];
        BORE_WSTAMP(0);
        net.forward(th16, xin, /*keep_logits=*/true);
        BORE_WSTAMP(1);
        net.template store_A<1, n - 1>(tile16, rb);
        float delta = 0.f;
        if (lane < 16 && live) {
          const float x = net.h[n][0][0];
          const float ex = fit_exp_neg(-fabsf(x));
          const float rden = fit_rcp(1.f + ex);
          const float sig = x >= 0.f ? rden : ex * rden;
          if (a.epoch_loss) eloss += fmaxf(x, 0.f) - x * zz + log1pf(ex);
          delta = bf16_round((sig - zz) * inv_nb);
        }
        if (lane < 16) tile16[L.doff[n] + row * L.lda[n]] = f32_to_bf16(delta);
        net.set_output_delta(delta);
        BORE_WSTAMP(2);
        net.template backward<n, 2>(th16);
        BORE_WSTAMP(3);
        net.template store_D<1, n - 1>(tile16, rb);
        BORE_WSTAMP(4);
      }
      __syncthreads();
      BORE_WSTAMP(5);
      {
        float G[wide_tiles_per_wave<SHAPE>()][5];
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        wide_grads<SHAPE, unsigned short>(tile16, G, tid_o);
        BORE_WSTAMP(6);
        __syncthreads();
        float *gl = reinterpret_cast<float *>(tile16);  // [P] floats: the tile region + its extension
        wide_scatter<SHAPE>(G, gl, tid_o);
        __syncthreads();
        BORE_WSTAMP(7);
        wide_adam<SHAPE, true>(th16, gl, theta_g, m_g, v_g, alpha, omb1, omb2, a.eps, tid_o);
        BORE_WSTAMP(8);
        __syncthreads();
        BORE_WSTAMP(9);
        // the image overwrote D_n, whose columns past the single output unit must read zero (the
        // next step stores column 0 only); the other A_l / D_l rows are rewritten in full
        for (int i = tid; i < BORE_BATCH_MAX * L.lda[n]; i += nthr) tile16[L.doff[n] + i] = 0;
      }
      if (wv == (BORE_THREADS / 64) - 1) {
        b1p *= (double)a.beta1;
        b2p *= (double)a.beta2;
        const float an = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
        if (lane == 0) misc[5] = an;
      }
      __syncthreads();
    }
    if (a.epoch_loss) {
      eloss = wave_sum(eloss);
      if (lane == 0) misc[1 + wv] = eloss;
      __syncthreads();
      if (tid == 0)
        a.epoch_loss[model * a.epochs + e] = (misc[1] + misc[2] + misc[3] + misc[4]) / (float)N;
    }
  }
  if (tid == 0) a.at[model] = t0 + (long long)a.epochs * steps;
}

// This lane's slots of weight-gradient tile t = wave + 4 I of the bf16-MFMA fit (layer l known at compile
// time): kb / cb = the tile's block row / column; the index of the lane's four weights = a wave-uniform
// part pu (scalar registers) + a lane part lb4 (bytes); pbu / 4 m16 = the same for its bias.
template <int SHAPE, int I>
__device__ __forceinline__ void bf16_tile_slots(int wv, int lane, int &kb, int &cb, int &pu, unsigned &lb4,
                                                bool &ok4, int &pbu, bool &okb) {
  using Pl = Bf16Plan<SHAPE>;
  constexpr MlpLayout L = Pl::L;
  constexpr int l = Pl::layer_of_tile(4 * I);
  constexpr int K = L.w[l - 1], Nw = L.w[l], ncb = Pl::T(l);
  constexpr bool FULL = K % 16 == 0 && Nw % 16 == 0;
  const int m16 = lane & 15, q4 = lane >> 4;
  const int r = wv + 4 * I - Pl::tiles_before(l);
  kb = r / ncb;
  cb = r - kb * ncb;
  if constexpr (Nw == 1) {  // one column: the C layout's four rows ARE contiguous (lanes m = 0)
    pu = L.goff_w[l] + 16 * kb;
    lb4 = 16u * (unsigned)q4;
    ok4 = m16 == 0 && 16 * kb + 4 * q4 < K;
  } else {  // tile order: tile (kb, cb) = 256 consecutive floats, the lane's four at 4 * lane
    static_assert(FULL, "a wide static shape has layer sizes that are multiples of 16");
    pu = L.goff_w[l] + (kb * ncb + cb) * 256;
    lb4 = 16u * (unsigned)lane;
    ok4 = true;
  }
  okb = kb == 0 && q4 == 0 && (FULL || 16 * cb + m16 < Nw);
  pbu = L.goff_b[l] + 16 * cb;
}

// The mixed-precision fit on the bf16 matrix cores (fit_bf16_mfma.h has the design).
template <int SHAPE>
__global__ __launch_bounds__(BORE_THREADS) void fit_bf16_mfma_kernel(const FitBf16Args a) {
  extern __shared__ float smem[];
  using Pl = Bf16Plan<SHAPE>;
  using Net = Bf16Net<SHAPE>;
  constexpr MlpLayout L = Pl::L;
  constexpr int P = L.P, n = Pl::n, D = L.w[0], RS = Pl::RS;
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wv = tid >> 6, lane = tid & 63, m16 = lane & 15, q4 = lane >> 4;
  const long long model = blockIdx.x;
  const int N = a.N;
  float *theta_g = a.theta + model * P;
  float *m_g = a.am + model * P;
  float *v_g = a.av + model * P;
  // the per-parameter state goes to tile order for the launch (TileOrder, fit_bf16_mfma.h)
  using TO = TileOrder<SHAPE>;
  TO::convert(theta_g, smem, true);
  TO::convert(m_g, smem, true);
  TO::convert(v_g, smem, true);
  for (int i = tid; i < (a.total >> 2); i += nthr) smem[i] = 0.f;
  __syncthreads();
  char *base = reinterpret_cast<char *>(smem);
  unsigned short *wf = reinterpret_cast<unsigned short *>(base + Pl::o_wf);
  unsigned short *wb = reinterpret_cast<unsigned short *>(base + Pl::o_wb);
  float *bias = reinterpret_cast<float *>(base + Pl::o_bias);
  unsigned short *img = reinterpret_cast<unsigned short *>(base + Pl::o_img);
  float *misc = reinterpret_cast<float *>(base + a.o_misc);
  int *perm_all = reinterpret_cast<int *>(base + a.o_perm);
  int *perm_s = perm_all;
  unsigned *keys = reinterpret_cast<unsigned *>(base + a.o_keys);
  const int PG = a.perm ? 1 : perm_group(N, BORE_THREADS);

  const float *X_g = a.X + model * (long long)N * D;
  const float *z_g = a.z + model * (long long)N;
  for (int p = tid; p < P; p += nthr) bf16_put<SHAPE>(wf, wb, bias, p, theta_g[TO::index(p)]);

  const long long t0 = a.at[model];
  double b1p = pow((double)a.beta1, (double)t0);
  double b2p = pow((double)a.beta2, (double)t0);
  const float omb1 = 1.f - a.beta1, omb2 = 1.f - a.beta2;
  const int steps = (N + a.B - 1) / a.B;
  b1p *= (double)a.beta1;
  b2p *= (double)a.beta2;
  const float alpha_first = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
  bool first_step = true;
  __syncthreads();
#ifdef BORE_FIT_MARKS
  if (lane < 32) g_fit_lds[wv & 7][lane] = 0;
#endif
  FIT_MARK_DECL;
  // The float32 MASTER weights live in registers for the launch: the lane's four of each of this wave's tiles
  // t = wave + 4 I (4 registers per tile, 88 for 32->128-128-1; a wave alone on its SIMD has 512).  Read and
  // written in HBM every step beside m and v they were a third of the update's traffic, which at 256 loops is
  // what the memory system delivers (profiles/r4/ab_log.txt).
  constexpr int TPW = Pl::tiles_per_wave();
  typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
  f4u sw[TPW];
  float sbw[TPW];  // (only the tiles of block row 0 carry a bias: the other entries are never touched)
  {
    const BufF32 b_w(theta_g, P);
    static_for<0, TPW>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      int kb, cb, pu, pbu;
      unsigned lb4;
      bool ok4, okb;
      bf16_tile_slots<SHAPE, I>(wv, lane, kb, cb, pu, lb4, ok4, pbu, okb);
      const f4u z4 = {0.f, 0.f, 0.f, 0.f};
      sw[I] = ok4 ? b_w.ld4(pu, lb4) : z4;
      if (kb == 0) sbw[I] = okb ? b_w.ld1(pbu, 4u * (unsigned)m16) : 0.f;
    });
  }

  for (int e = 0; e < a.epochs; ++e) {
    if (a.perm) {
      const int *pg = a.perm + (model * a.epochs + e) * (long long)N;
      for (int i = tid; i < N; i += nthr) perm_s[i] = pg[i];
      __syncthreads();
    } else if (PG > 1) {
      const int eg = e & (PG - 1);
      if (eg == 0)
        make_perm_group(a.seed, a.model0 + model, a.epoch0 + e, min(PG, a.epochs - e), N, keys,
                        perm_all);
      perm_s = perm_all + eg * N;
    } else {
      make_perm(shuffle_base(a.seed, a.model0 + model, a.epoch0 + e), N, keys, perm_s);
    }
    float eloss = 0.f;
    for (int s = 0; s < steps; ++s) {
      const int row0 = s * a.B;
      const int nb = min(a.B, N - row0);
      const float inv_nb = fit_rcp((float)nb);  // (wave-uniform; the mean over the step's rows as a multiply)
      const float alpha = first_step ? alpha_first : misc[5];
      first_step = false;
      BORE_WSTAMP_DECL;
      // (index arithmetic below must not be hoisted out of the step loop as tables: see wide_adam)
      int tid_o = tid;
      asm volatile("" : "+v"(tid_o));
      {  // every wave runs its 16 rows; rows past the batch are dead (x = 0, delta = 0)
        Net net;
        net.set_acts(a.L);
        const int row = wv * 16 + m16;
        const bool live = row < nb;
        const int src = live ? perm_s[row0 + row] : 0;
        // the input rows as the B fragments of layer 1 (k-slot (q, i) = column 32c + 16(i>>2) + 4q + (i&3)),
        // and, transposed, into the image of A_0 for the weight gradients
        bf16x8_t xfrag[Pl::CF(1)];
#pragma unroll
        for (int c = 0; c < Pl::CF(1); ++c) {
          float lo[4] = {0.f, 0.f, 0.f, 0.f}, hi[4] = {0.f, 0.f, 0.f, 0.f};
          const int c_lo = 32 * c + 4 * q4, c_hi = c_lo + 16;
          if (live && c_lo < D) {
            const float4 v = *reinterpret_cast<const float4 *>(X_g + (long long)src * D + c_lo);
            lo[0] = bf16_round_hw(v.x); lo[1] = bf16_round_hw(v.y); lo[2] = bf16_round_hw(v.z); lo[3] = bf16_round_hw(v.w);
          }
          if (live && c_hi < D) {
            const float4 v = *reinterpret_cast<const float4 *>(X_g + (long long)src * D + c_hi);
            hi[0] = bf16_round_hw(v.x); hi[1] = bf16_round_hw(v.y); hi[2] = bf16_round_hw(v.z); hi[3] = bf16_round_hw(v.w);
          }
          xfrag[c] = pack_frag(lo, hi);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if (c_lo < L.Np[0]) img[Pl::at_off(0) + Pl::t_index(c_lo + i, row)] = (unsigned short)(__float_as_uint(lo[i]) >> 16);
            if (c_hi < L.Np[0]) img[Pl::at_off(0) + Pl::t_index(c_hi + i, row)] = (unsigned short)(__float_as_uint(hi[i]) >> 16);
          }
        }
        float zz = 0.f;
        if (q4 == 0 && live) zz = z_g[src];
        BORE_WSTAMP(0);
        net.forward(wf, bias, xfrag);
        BORE_WSTAMP(1);
        float delta = 0.f;
        if (lane < 16 && live) {
          const float x = net.h[n][0][0];
          const float ex = fit_exp_neg(-fabsf(x));
          const float rden = fit_rcp(1.f + ex);
          const float sig = x >= 0.f ? rden : ex * rden;
          if (a.epoch_loss) eloss += fmaxf(x, 0.f) - x * zz + log1pf(ex);
          delta = bf16_round_hw((sig - zz) * inv_nb);
        }
#pragma unroll
        for (int t = 0; t < Net::TM; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) net.d[n][t][r] = 0.f;
        net.d[n][0][0] = lane < 16 ? delta : 0.f;
        BORE_WSTAMP(2);
        net.backward(wb, bias);
        BORE_WSTAMP(3);
        net.store_images(img, row);
        BORE_WSTAMP(4);
      }
      __syncthreads();
      BORE_WSTAMP(5);
      {
        // ---- weight gradients + Adam, tile by tile: this wave's tiles t = wave, wave + 4, ...  A tile
        // is two MFMAs (k = the 64 batch rows) and leaves lane (q, m) holding dW_l[16kb + 4q + r][16cb + m],
        // r = 0..3.  The master weight / m / v are kept in exactly that order for the launch
        // (TileOrder, fit_bf16_mfma.h): one 16-byte load and store each per lane and tile, 1 KiB
        // contiguous per wave -- the CU's one vector-memory pipe was what bounded this phase (30 dword
        // instructions per tile, then 12 16-byte ones on half lines).  They are requested three tiles
        // ahead of use and ahead of the stores in between (loads and stores share one counter); the
        // new weights go to HBM and, rounded, into the LDS images. ----
        constexpr int TOTAL = Pl::total_tiles(), AHEAD = 3;
        constexpr int RING = AHEAD + 2;  // (tile I - 1 is still in use when tile I + AHEAD is requested)
        // (lane coordinates re-derived from an opaque copy of the thread id: computed from the
        // loop-invariant ones, the slot indices of all tiles are hoisted out of the step loop and
        // kept live across it -- 22 tiles x 10 registers, most of them spilled)
        const int wv = __builtin_amdgcn_readfirstlane(tid_o >> 6), lane = tid_o & 63, m16 = lane & 15, q4 = lane >> 4;
        f4u pm[RING], pv[RING];
        float bm[RING], bv[RING];
        static_assert(Pl::layers_aligned() && TOTAL % 4 == 0, "tiles of a layer must start at a multiple of 4");
        // this lane's slots of tile t = wave + 4 I (layer l known at compile time): kb / cb = the tile's
        // block row / column; the index of the lane's four weights = a wave-uniform part pu (scalar
        // registers) + a lane part lb4 (bytes); pbu / 4 m16 = the same for its bias
        auto slots = [&](auto ic, int &kb, int &cb, int &pu, unsigned &lb4, bool &ok4, int &pbu, bool &okb) {
          bf16_tile_slots<SHAPE, decltype(ic)::value>(wv, lane, kb, cb, pu, lb4, ok4, pbu, okb);
        };
        // buffer addressing (resource + 32-bit lane offset + scalar offset): one instruction per access
        // and no 64-bit address arithmetic (as flat pointers every access cost a 64-bit vector add)
        const BufF32 b_m(m_g, P), b_v(v_g, P);
        const unsigned lbb = 4u * (unsigned)m16;
        auto request = [&](auto ic) {
          constexpr int I = decltype(ic)::value, l = Pl::layer_of_tile(4 * I);
          constexpr bool FULL = L.w[l - 1] % 16 == 0 && L.w[l] % 16 == 0;
          int kb, cb, pu, pbu;
          unsigned lb4;
          bool ok4, okb;
          slots(ic, kb, cb, pu, lb4, ok4, pbu, okb);
          const f4u z4 = {0.f, 0.f, 0.f, 0.f};
          if constexpr (FULL) {
            pm[I % RING] = b_m.ld4(pu, lb4);
            pv[I % RING] = b_v.ld4(pu, lb4);
          } else {
            pm[I % RING] = ok4 ? b_m.ld4(pu, lb4) : z4;
            pv[I % RING] = ok4 ? b_v.ld4(pu, lb4) : z4;
          }
          if (kb == 0) {  // (wave-uniform: only these tiles carry a bias)
            bm[I % RING] = okb ? b_m.ld1(pbu, lbb) : 0.f;
            bv[I % RING] = okb ? b_v.ld1(pbu, lbb) : 0.f;
          }
        };
        static_for<0, (AHEAD < TPW ? AHEAD : TPW)>([&](auto ic) { request(ic); });
        // Software pipeline, two stages per tile: FRONT(I) = the tile's two MFMAs on operands that were
        // read from LDS one iteration earlier, then the reads for tile I + 1; FINISH(I - 1) = transpose,
        // Adam, stores of the previous tile.  Both sit in one scheduling region, so the LDS and MFMA
        // latencies of one tile run under the other's vector arithmetic (one wave per SIMD: nobody
        // else hides them; in tile-after-tile form they were half of the phase).
        u32x4_t oa0, oa1, ob0, ob1;
        auto fetch = [&](auto ic) {
          constexpr int I = decltype(ic)::value, l = Pl::layer_of_tile(4 * I);
          int kb, cb, pu, pbu;
          unsigned lb4;
          bool ok4, okb;
          slots(ic, kb, cb, pu, lb4, ok4, pbu, okb);
          // rows 32c + 8 q4 .. + 7 of unit row 16 kb|cb + m16: chunk 4c + q4, swizzled (Bf16Plan::t_index)
          const int ua = 16 * kb + m16, ub = 16 * cb + m16;
          const unsigned short *ap = img + Pl::at_off(l - 1) + ua * RS, *bp = img + Pl::dt_off(l) + ub * RS;
          const int sa = (ua >> 1) & 7, sb = (ub >> 1) & 7;
          oa0 = *reinterpret_cast<const u32x4_t *>(ap + ((q4 ^ sa) << 3));
          oa1 = *reinterpret_cast<const u32x4_t *>(ap + (((4 + q4) ^ sa) << 3));
          ob0 = *reinterpret_cast<const u32x4_t *>(bp + ((q4 ^ sb) << 3));
          ob1 = *reinterpret_cast<const u32x4_t *>(bp + (((4 + q4) ^ sb) << 3));
        };
        fetch(std::integral_constant<int, 0>{});
        f32x4 acc_prev = {0.f, 0.f, 0.f, 0.f};
        float bs_prev = 0.f;
        auto finish = [&](auto ic, const f32x4 &acc, const float bs) {
          constexpr int I = decltype(ic)::value, l = Pl::layer_of_tile(4 * I);
          int kb, cb, pu, pbu;
          unsigned lb4;
          bool ok4, okb;
          slots(ic, kb, cb, pu, lb4, ok4, pbu, okb);
          // g[r] = dW_l[16kb + 4q + r][16cb + m]: the tile order of theta / m / v IS this layout
          const float g[4] = {acc[0], acc[1], acc[2], acc[3]};
          // Adam (ResourceApplyAdam form; v_sqrt_f32 / v_rcp_f32, 1 ulp: the new weight is rounded to
          // bfloat16 for the next step anyway and the master copy carries 24 bits either way)
          constexpr int cur = I % RING;
          float wn[4];
          // (two elements per instruction where the operation has a packed form -- v_pk_add_f32 /
          // v_pk_mul_f32; the same IEEE operations in the same order)
          typedef float f2 __attribute__((ext_vector_type(2)));
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const f2 gg = {g[2 * h], g[2 * h + 1]};
            f2 mm = {pm[cur][2 * h], pm[cur][2 * h + 1]}, vv = {pv[cur][2 * h], pv[cur][2 * h + 1]};
            const f2 ww = {sw[I][2 * h], sw[I][2 * h + 1]};
            mm += (gg - mm) * omb1;
            vv += (gg * gg - vv) * omb2;
            const f2 den = {__builtin_amdgcn_sqrtf(vv.x) + a.eps, __builtin_amdgcn_sqrtf(vv.y) + a.eps};
            const f2 rcp = {__builtin_amdgcn_rcpf(den.x), __builtin_amdgcn_rcpf(den.y)};
            const f2 w2 = ww - (mm * alpha) * rcp;
            wn[2 * h] = w2.x; wn[2 * h + 1] = w2.y;
            pm[cur][2 * h] = mm.x; pm[cur][2 * h + 1] = mm.y;
            pv[cur][2 * h] = vv.x; pv[cur][2 * h + 1] = vv.y;
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(wn[r]));
          sw[I] = f4u{wn[0], wn[1], wn[2], wn[3]};
          if (ok4) {
            b_m.st4(pm[cur], pu, lb4);
            b_v.st4(pv[cur], pu, lb4);
          }
          // the LDS images (fragment orders: fit_bf16_mfma.h) and the float copies
          if (ok4) {
            uint2 h4;
            h4.x = pack2_bf16(wn[0], wn[1]);
            h4.y = pack2_bf16(wn[2], wn[3]);
            const unsigned short hs[4] = {(unsigned short)h4.x, (unsigned short)(h4.x >> 16),
                                          (unsigned short)h4.y, (unsigned short)(h4.y >> 16)};
            if constexpr (l == n) {  // (k = 16kb + 4q + r, column 0): consecutive k-slots of one forward fragment
              *reinterpret_cast<uint2 *>(wf + Pl::wf_off(l) + ((kb >> 1) * 64 + q4 * 16) * 8 + (kb & 1) * 4) = h4;
              float4 f4;
              f4.x = bf16_to_f32(hs[0]); f4.y = bf16_to_f32(hs[1]); f4.z = bf16_to_f32(hs[2]); f4.w = bf16_to_f32(hs[3]);
              *reinterpret_cast<float4 *>(bias + Pl::wlast_off() + 16 * kb + 4 * q4) = f4;
            } else {  // (k = 16kb + 4q + r, column 16cb + m): four consecutive k-slots of one forward fragment
              static_assert(L.w[l - 1] % 16 == 0 && L.w[l] % 16 == 0, "tile-order layers only");
              *reinterpret_cast<uint2 *>(wf + Pl::wf_off(l) + ((cb * Pl::CF(l) + (kb >> 1)) * 64 + q4 * 16 + m16) * 8 + (kb & 1) * 4) = h4;
              if constexpr (l >= 2) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  wb[Pl::wb_off(l) + ((kb * Pl::CB(l) + (cb >> 1)) * 64 + (m16 >> 2) * 16 + 4 * q4 + r) * 8 + (cb & 1) * 4 + (m16 & 3)] = hs[r];
              }
            }
          }
          if (kb == 0) {  // bias: gradient = the sums of D_l's columns = of this lane's 16 rows (bs), then of the 4 lane rows
            const float gb = rows_sum4(bs);
            float mm = bm[cur], vv = bv[cur];
            mm += (gb - mm) * omb1;
            vv += (gb * gb - vv) * omb2;
            const float wnb = sbw[I] - (mm * alpha) * __builtin_amdgcn_rcpf(__builtin_amdgcn_sqrtf(vv) + a.eps);
            sbw[I] = wnb;
            if (okb) {
              b_m.st1(mm, pbu, lbb);
              b_v.st1(vv, pbu, lbb);
              bias[Pl::bias_off(l) + 16 * cb + m16] = bf16_round_hw(wnb);
            }
          }
        };
        static_for<0, TPW + 1>([&](auto ic) {
          constexpr int I = decltype(ic)::value;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          float bs = 0.f;
          if constexpr (I < TPW) {
            constexpr int l = Pl::layer_of_tile(4 * I);
            if constexpr (I + AHEAD < TPW) request(std::integral_constant<int, I + AHEAD>{});
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, oa0), __builtin_bit_cast(bf16x8_t, ob0), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, oa1), __builtin_bit_cast(bf16x8_t, ob1), acc, 0, 0, 0);
            const int r = wv + 4 * I - Pl::tiles_before(l);
            if (r < Pl::T(l)) {  // kb == 0 (wave-uniform): the tile carries its columns' bias
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                bs += __uint_as_float(ob0[w] << 16);
                bs += __uint_as_float(ob0[w] & 0xffff0000u);
              }
#pragma unroll
              for (int w = 0; w < 4; ++w) {
                bs += __uint_as_float(ob1[w] << 16);
                bs += __uint_as_float(ob1[w] & 0xffff0000u);
              }
            }
            if constexpr (I + 1 < TPW) fetch(std::integral_constant<int, I + 1>{});
          }
          if constexpr (I > 0) finish(std::integral_constant<int, I - 1>{}, acc_prev, bs_prev);
          __builtin_amdgcn_sched_barrier(0);
          acc_prev = acc;
          bs_prev = bs;
        });
        BORE_WSTAMP(6);
        BORE_WSTAMP(8);
      }
      if (wv == (BORE_THREADS / 64) - 1) {
        b1p *= (double)a.beta1;
        b2p *= (double)a.beta2;
        const float an = a.lr * sqrtf(1.f - (float)b2p) / (1.f - (float)b1p);
        if (lane == 0) misc[5] = an;
      }
      __syncthreads();
      BORE_WSTAMP(9);
    }
    if (a.epoch_loss) {
      eloss = wave_sum(eloss);
      if (lane == 0) misc[1 + wv] = eloss;
      __syncthreads();
      if (tid == 0)
        a.epoch_loss[model * a.epochs + e] = (misc[1] + misc[2] + misc[3] + misc[4]) / (float)N;
    }
  }
  {  // the master weights back to memory (tile order)
    const BufF32 b_w(theta_g, P);
    static_for<0, TPW>([&](auto ic) {
      constexpr int I = decltype(ic)::value;
      int kb, cb, pu, pbu;
      unsigned lb4;
      bool ok4, okb;
      bf16_tile_slots<SHAPE, I>(wv, lane, kb, cb, pu, lb4, ok4, pbu, okb);
      if (ok4) b_w.st4(sw[I], pu, lb4);
      if (kb == 0 && okb) b_w.st1(sbw[I], pbu, 4u * (unsigned)m16);
    });
  }
  // back to the packed order (the LDS images are dead by now)
  __syncthreads();
  TO::convert(theta_g, smem, false);
  TO::convert(m_g, smem, false);
  TO::convert(v_g, smem, false);
  if (tid == 0) a.at[model] = t0 + (long long)a.epochs * steps;
}

// ---------------------------------------------------------------------------
// forward (predict) / value + input gradient: grid = (models, workgroups); every wave walks
// its own 16-row blocks of the input -- no barrier after the weights are staged
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
  int o_tile, o_vals, o_layout, total, shape, bf16;
  int d_in;  // the net's input dimension (<= the static shape's: stage_theta_in)
};

template <bool WITH_GRAD, int SHAPE, bool BF16 = false>
__global__ __launch_bounds__(BORE_THREADS) void rows_kernel(const RowArgs a) {
  extern __shared__ float smem[];
  constexpr MlpLayout Lc = bore_static_layout(SHAPE > 0 ? SHAPE : 0, WITH_GRAD ? 2 : 0, BORE_BATCH_MAX);
  const MlpLayout &L = begin_kernel<SHAPE>(Lc, a.L, smem, a.total, a.o_layout);
  const int tid = threadIdx.x;
  const int wv = tid >> 6, lane = tid & 63, m16 = lane & 15, q4 = lane >> 4;
  const long long model = blockIdx.x;
  const int n = layer_count<SHAPE>(L), D = (bore_shape_takes_fewer_inputs(SHAPE) && !BF16) ? a.d_in : L.w[0];
  float *th = smem, *tile = smem + a.o_tile, *vals = smem + a.o_vals;
  if constexpr (BF16) arg_bf16_stage<SHAPE>(a.theta + model * L.P, smem);
  else if constexpr (bore_shape_takes_fewer_inputs(SHAPE)) stage_theta_in<false>(L, n, a.theta, model, D, smem);
  else stage_theta<false>(L, n, a.theta + model * L.P, smem);
  __syncthreads();
  // waves that own a 16-row slice of the tile buffers (the bf16-MFMA form has no tile: all four)
  const int waves = BF16 ? BORE_THREADS / 64 : L.tbp >> 4;
  if (wv >= waves) return;
  const long long xoff = a.x_shared ? 0 : model * a.n_rows * D;
  float *out = a.out + model * a.n_rows;
  const long long n_blocks = (a.n_rows + 15) >> 4;
  if constexpr (BF16) {  // a bfloat16 model: the bf16 matrix cores (arg_bf16_mfma.h)
    static_assert(!BF16 || bore_shape_is_wide(SHAPE), "bfloat16: wide static shapes");
    using ANet = ArgBf16Net<SHAPE>;
    const ArgBf16Images im = arg_bf16_images<SHAPE>(smem);
    ANet net;
    net.set_acts(a.L);
    for (long long g = (long long)blockIdx.y * waves + wv; g < n_blocks;
         g += (long long)gridDim.y * waves) {
      const long long row = g * 16 + m16;
      bf16x8_t xf[ANet::CF1];
      ANet::make_xfrag(xf, [&](int d) -> float {
        if (d >= D || row >= a.n_rows) return 0.f;
        return WITH_GRAD ? (float)a.Xd[xoff + row * D + d] : a.Xf[xoff + row * D + d];  // Keras autocast
      });
      if (WITH_GRAD) {
        const float Tv = net.fg(im, xf, a.transform, a.sign);
        if (lane < 16 && row < a.n_rows) out[row] = Tv;
        double *grad = a.grad + (model * a.n_rows) * D;
        if (row < a.n_rows) {
#pragma unroll
          for (int t = 0; t < ANet::T0; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int d = 16 * t + 4 * q4 + r;
              if (d < D) grad[row * D + d] = (double)net.d[0][t][r];
            }
        }
      } else {
        net.predict(im, xf);
        if (lane < 16 && row < a.n_rows) out[row] = net.out;
      }
    }
    return;
  }
  using Net = RegNet<(SHAPE > 0 ? SHAPE : 1), WITH_GRAD ? 2 : 0, BF16>;
  const typename Net::WT *thw = reinterpret_cast<const typename Net::WT *>(smem);
  Net net;  // static shapes: the weights stay in this lane's registers for every row-block
  if constexpr (SHAPE > 0) {
    if constexpr (Net::RT_ACT) net.set_acts(a.L);
    net.load_fwd(thw);
    if (WITH_GRAD) net.template load_bwd<Net::n, 1>(thw);
  }
  for (long long g = (long long)blockIdx.y * waves + wv; g < n_blocks;
       g += (long long)gridDim.y * waves) {
    const long long row = g * 16 + m16;  // the global row of this lane's operand slot
    if constexpr (SHAPE > 0) {  // activations in registers (mlp_regs.h): no LDS tile at all
      float xin[Net::KC0];
#pragma unroll
      for (int kc = 0; kc < Net::KC0; ++kc) {
        const int d = 4 * kc + q4;
        float x = 0.f;
        if (d < D && row < a.n_rows)
          x = WITH_GRAD ? (float)a.Xd[xoff + row * D + d] : a.Xf[xoff + row * D + d];
        xin[kc] = Net::rnd(x);
      }
      if (WITH_GRAD) {
        const float Tv = net.fg(thw, xin, a.transform, a.sign);
        if (lane < 16 && row < a.n_rows) out[row] = Tv;
        double *grad = a.grad + (model * a.n_rows) * D;
        if (row < a.n_rows) {
#pragma unroll
          for (int t = 0; t < Net::L.Np[0] / 16; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int d = 16 * t + 4 * q4 + r;
              if (d < D) grad[row * D + d] = (double)net.d[0][t][r];
            }
        }
      } else {
        net.forward(thw, xin, false);
        if (lane < 16 && row < a.n_rows) out[row] = net.h[Net::n][0][0];
      }
      continue;
    }
    float *A0 = tile + L.aoff[0] + (wv * 16 + m16) * L.lda[0];
    for (int d = q4; d < D; d += 4) {
      float x = 0.f;
      if (row < a.n_rows)
        x = WITH_GRAD ? (float)a.Xd[xoff + row * D + d]  // Keras autocast fp64 -> fp32
                      : a.Xf[xoff + row * D + d];
      A0[d] = x;
    }
    wave_lds_sync();
    if (WITH_GRAD) {
      fg_rowblock(L, n, th, tile, wv, a.transform, a.sign, vals);
      if (lane < 16 && g * 16 + lane < a.n_rows) out[g * 16 + lane] = vals[wv * 16 + lane];
      const float *D0 = tile + L.doff[0] + (wv * 16 + m16) * L.lda[0];
      double *grad = a.grad + (model * a.n_rows) * D;
      if (row < a.n_rows)
        for (int d = q4; d < D; d += 4) grad[row * D + d] = (double)D0[d];
    } else {
      fwd_all(L, n, th, tile, wv, false);
      if (lane < 16 && g * 16 + lane < a.n_rows)
        out[g * 16 + lane] = tile[L.aoff[n] + (wv * 16 + lane) * L.lda[n]];
    }
    wave_lds_sync();
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
  int o_tile, o_misc, o_layout, total;
};

__global__ __launch_bounds__(BORE_THREADS) void evaluate_kernel(const EvalArgs a) {
  extern __shared__ float smem[];
  const MlpLayout &L = stage_layout(a.L, smem, a.total, a.o_layout);
  const int tid = threadIdx.x, nthr = blockDim.x;
  const int wv = tid >> 6, lane = tid & 63, m16 = lane & 15, q4 = lane >> 4;
  const long long model = blockIdx.x;
  const int n = L.n_layers, D = L.w[0];
  float *th = smem, *tile = smem + a.o_tile, *misc = smem + a.o_misc;
  load_theta(L, n, a.theta + model * L.P, th);
  __syncthreads();
  const float *X = a.X + model * a.N * D;
  const float *z = a.z + model * a.N;
  const int waves = L.tbp >> 4;
  const long long n_blocks = (a.N + 15) >> 4;
  float lsum = 0.f, csum = 0.f;
  if (wv < waves)
    for (long long g = wv; g < n_blocks; g += waves) {
      const long long row = g * 16 + m16;
      float *A0 = tile + L.aoff[0] + (wv * 16 + m16) * L.lda[0];
      for (int d = q4; d < D; d += 4) A0[d] = row < a.N ? X[row * D + d] : 0.f;
      wave_lds_sync();
      fwd_all(L, n, th, tile, wv, true);
      if (lane < 16 && g * 16 + lane < a.N) {
        const float x = tile[L.aoff[n] + (wv * 16 + lane) * L.lda[n]];
        const float zz = z[g * 16 + lane];
        lsum += fmaxf(x, 0.f) - x * zz + log1pf(expf(-fabsf(x)));
        const float o = L.act[n] == BORE_ACT_SIGMOID ? sigmoid_stable(x) : x;
        csum += ((o > 0.5f) == (zz > 0.5f)) ? 1.f : 0.f;
      }
      wave_lds_sync();
    }
  float reg = 0.f;
  if (L.any_l2)
    for (int p = tid; p < L.P; p += nthr) {
      const ParamRef r = param_ref(L, p, n);
      const float l2 = r.k >= 0 ? L.l2_w[r.l] : L.l2_b[r.l];
      const float w = th[r.lds];
      reg = fmaf(l2 * w, w, reg);
    }
  lsum = wave_sum(lsum);
  csum = wave_sum(csum);
  reg = wave_sum(reg);
  if (lane == 0) {
    atomicAdd(&misc[0], lsum);
    atomicAdd(&misc[1], csum);
    atomicAdd(&misc[2], reg);
  }
  __syncthreads();
  if (tid == 0) {
    a.loss[model] = misc[0] / (float)a.N + misc[2];
    a.acc[model] = misc[1] / (float)a.N;
  }
}

// ---------------------------------------------------------------------------
// shuffle stream dump
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(BORE_THREADS) void shuffle_kernel(unsigned long long seed,
                                                              long long model0, long long epoch0,
                                                              int epochs, int N, int *perm,
                                                              int wave_form) {
  extern __shared__ float smem[];
  unsigned *keys = reinterpret_cast<unsigned *>(smem);
  const long long model = blockIdx.x, e = blockIdx.y;
  if (wave_form) {  // (tests: the one-wave bucket ranking the pipelined fit draws its shuffles with)
    if (threadIdx.x < 64)
      make_perm_wave_buckets(shuffle_base(seed, model0 + model, epoch0 + e), N,
                             reinterpret_cast<unsigned long long *>(smem),
                             perm + (model * epochs + e) * (long long)N);
    return;
  }
  make_perm(shuffle_base(seed, model0 + model, epoch0 + e), N, keys,
            perm + (model * epochs + e) * (long long)N);
}

// ---------------------------------------------------------------------------
// host side of the C-ABI
// ---------------------------------------------------------------------------
static int fit_bf16_impl(const bore_mlp_desc *, int, float *, float *, float *, int64_t *,
                         const float *, const float *, int64_t, int, int, const int32_t *, uint64_t,
                         int64_t, int64_t, const bore_adam_cfg *, float *, void *);

// Arguments, LDS bytes and kernel flavour of a float32 fit launch (what bore_mlp_fit launches;
// the fused iteration kernel of bore_iter.hip builds its fit phase with it).  Returns 1 when
// there is nothing to do (epochs == 0).
static int fit_build(const bore_mlp_desc *desc, int n_models, float *theta, float *adam_m,
                     float *adam_v, int64_t *adam_t, const float *X, const float *z, int64_t N,
                     int epochs, int batch_size, const int32_t *perm, uint64_t seed,
                     int64_t model_index0, int64_t epoch0, const bore_adam_cfg *adam,
                     float *epoch_loss, FitArgs &a, size_t &lds_floats, int &shape_out) {
  if (batch_size < 1) return fail(BORE_E_INVALID, "fit: batch_size must be positive (got %d)", batch_size);
  if (N < 1 || N > (1 << 24)) return fail(BORE_E_INVALID, "fit: N=%lld out of range", (long long)N);
  // a mini-batch of up to 64 rows is one tile (larger ones: 64-row sub-tiles, fit_body); perm
  // (+ keys) and the batch targets ride along
  const int tile_rows = batch_size < BORE_BATCH_MAX ? batch_size : BORE_BATCH_MAX;
  // (the parked-row slots of fit_body: the narrow static shapes only -- a wide one has no LDS to spare)
  // (the fit-only static shape: not in batch mode -- the fused iteration kernels have no case for it)
  auto fit_flavour = [&](const bore_mlp_desc *d) {
    return g_batch ? bore_kernel_flavour(d, batch_size == BORE_BATCH_MAX) : bore_fit_flavour(d, batch_size == BORE_BATCH_MAX);
  };
  const int flavour_early = desc ? fit_flavour(desc) : 0;
  const size_t stage_f = flavour_early > 0 && !bore_shape_is_wide(flavour_early) ? BORE_FIT_STAGE_FLOATS_OF(flavour_early) : 0;
  const size_t fixed_extra = BORE_BATCH_MAX + 8 + stage_f + BORE_LAYOUT_FLOATS + 12;
  // 32->128-128-1 in float32 with 64-row batches: theta and the 64-row images do not share the LDS;
  // the weight gradients are formed in four rounds over 16-row images instead (wide_rounds_f32)
  const bool rounds = desc && desc->compute != BORE_COMPUTE_BF16 && batch_size == BORE_BATCH_MAX && !g_batch &&
                      bore_shape_fit_in_rounds(bore_kernel_flavour(desc, true));
  int rc;
  if (rounds) {
    if (n_models < 1) return fail(BORE_E_INVALID, "n_models must be >= 1 (got %d)", n_models);
    if (bore_make_layout(desc, 1, BORE_BATCH_MAX, &a.L)) return fail(BORE_E_INVALID, "bad bore_mlp_desc");
    a.L.tile_floats = WideTp16<4>::total();  // (the kernel's images; the row-major tile is not used)
    const size_t need = (size_t)a.L.P_lds + a.L.tile_floats + fixed_extra + (size_t)N * (perm ? 1 : 3);
    rc = need * 4 <= BORE_LDS_BYTES ? 0 : BORE_E_UNSUPPORTED;
    if (rc) {
      const size_t avail = BORE_LDS_BYTES / 4 - ((size_t)a.L.P_lds + a.L.tile_floats + fixed_extra);
      return fail(BORE_E_UNSUPPORTED,
                  "fit: N=%lld rows exceed what one workgroup's LDS holds beside this network (the "
                  "epoch's shuffle lives there): at most %zu rows",
                  (long long)N, avail / (perm ? 1 : 3));
    }
  } else
  rc = check_common(desc, n_models, 1, tile_rows, false, fixed_extra + (size_t)N * (perm ? 1 : 3),
                    &a.L);
  a.perm_in_lds = 1;
  if (rc == BORE_E_UNSUPPORTED && !check_common(desc, n_models, 1, tile_rows, false, fixed_extra, &a.L)) {
    // the network fits, the epoch's shuffle (drawn and ranked in LDS) does not
    if (perm && !g_batch) {
      // an EXPLICIT permutation of any length is read from memory step by step (round 3: the
      // reference's fit takes whatever the record holds; bore_amd.models draws such shuffles with
      // the host statement of the stream and passes them here)
      a.perm_in_lds = 0;
      rc = 0;
    } else {
      const size_t avail = BORE_LDS_BYTES / 4 - ((size_t)a.L.P_lds + a.L.tile_floats + fixed_extra);
      return fail(g_batch ? BORE_E_UNSUPPORTED : BORE_E_NEEDS_PERM,
                  "fit: N=%lld rows exceed what one workgroup's LDS holds beside this network when the "
                  "epoch's shuffle is drawn on the device (at most %zu rows): pass explicit shuffles "
                  "(`perm`, e.g. from bore_amd.shuffle.permutations) -- any N then",
                  (long long)N, avail / 3);
    }
  }
  if (rc) return rc;
  const MlpLayout &L = a.L;
  if (L.w[L.n_layers] != 1)
    return fail(BORE_E_INVALID, "fit: the last Dense layer must have 1 unit (binary classifier)");
  if (L.act[L.n_layers] != BORE_ACT_SIGMOID && L.act[L.n_layers] != BORE_ACT_LINEAR)
    return fail(BORE_E_INVALID, "fit: BCE needs a sigmoid or linear (from_logits) output layer");
  if (!theta || !adam_m || !adam_v || !adam_t || !X || !z || !adam)
    return fail(BORE_E_INVALID, "fit: null pointer");
  if (epochs < 0) return fail(BORE_E_INVALID, "fit: epochs < 0");
  if (epochs == 0) return 1;

  a.theta = theta; a.am = adam_m; a.av = adam_v; a.at = (long long *)adam_t;
  a.X = X; a.z = z; a.perm = perm; a.epoch_loss = epoch_loss;
  a.seed = seed; a.model0 = model_index0; a.epoch0 = epoch0;
  a.N = (int)N; a.epochs = epochs; a.B = batch_size;
  a.lr = adam->lr; a.beta1 = adam->beta1; a.beta2 = adam->beta2; a.eps = adam->eps;
  a.ids = a.its = nullptr; a.n_init = 0; a.cap = 0;
  if (g_batch) {
    if (perm || epoch_loss) return fail(BORE_E_INVALID, "fit: no perm / epoch_loss in batch mode");
    a.ids = g_batch->ids; a.its = g_batch->its; a.n_init = g_batch->n_init; a.cap = g_batch->cap;
  }

  // LDS carve: theta | tile | zt | misc | stage | perm | keys | [m v] | [X z]
  size_t off = 0;
  off += L.P_lds;
  a.o_tile = (int)off; off += L.tile_floats;
  a.o_zt = (int)off; off += BORE_BATCH_MAX;
  a.o_misc = (int)off; off += 8;
  a.o_stage = (int)off; off += stage_f;  // each lane's share of its next-step row (fit_body)
  const int PG = perm ? 1 : perm_group(N, BORE_THREADS);  // epochs shuffled together (N <= 128)
  size_t perm_f = a.perm_in_lds ? (size_t)PG * N : 0, keys_f = perm ? 0 : (size_t)perm_group_scratch_floats(N, PG);
  // (the pipelined fit of 65..128 rows keeps four epochs' shuffles: fit_body, pipe_perm)
  if (!perm && stage_f && N <= 128 && perm_f < 4 * (size_t)N) perm_f = 4 * (size_t)N;
  // (... and its drawing wave ranks by buckets: make_perm_wave_buckets' scratch)
  if (!perm && stage_f && keys_f < BORE_PERM_WAVE_FLOATS) keys_f = BORE_PERM_WAVE_FLOATS;
  a.perm_ahead = 0;  // (decided below, once everything else has its place)
  a.o_perm2 = 0;
  if (g_batch)  // a slot's own N (<= this N) may shuffle more epochs together: room for each case
    for (long long nn : {(long long)(N < 64 ? N : 64), (long long)(N < 128 ? N : 128)}) {
      const int pg = perm_group(nn, BORE_THREADS);
      if ((size_t)pg * nn > perm_f) perm_f = (size_t)pg * nn;
      if ((size_t)perm_group_scratch_floats(nn, pg) > keys_f)
        keys_f = (size_t)perm_group_scratch_floats(nn, pg);
    }
  if (g_batch && stage_f) {  // (a slot of up to 128 rows in pipelined form: four shuffles)
    const size_t nn = N < 128 ? (size_t)N : 128;
    if (perm_f < 4 * nn) perm_f = 4 * nn;
  }
  a.o_perm = (int)off; off += perm_f;
  off = (off + 3) & ~(size_t)3;  // keys: 64-bit words fetched two at a time
  a.o_keys = (int)off; off += keys_f;
  a.o_g = (int)off;
  if (batch_size > BORE_BATCH_MAX) off += L.P_lds;  // gradient sums carried between sub-tiles
  if ((off + BORE_LAYOUT_FLOATS + 4) * 4 > BORE_LDS_BYTES)
    return fail(BORE_E_UNSUPPORTED, "fit: theta+tile+perm need %zu B of LDS (> %d)", off * 4,
                BORE_LDS_BYTES);
  const size_t tail = BORE_LAYOUT_FLOATS + 4;  // the layout copy parked behind everything
  a.state_in_lds = (off + tail + 2 * (size_t)L.P_lds) * 4 <= BORE_LDS_BYTES;
  a.o_m = a.o_v = 0;
  if (a.state_in_lds) {
    a.o_m = (int)off; off += L.P_lds;
    a.o_v = (int)off; off += L.P_lds;
  }
  const size_t data = (size_t)N * (L.w[0] + 1);
  a.data_in_lds = (off + tail + data) * 4 <= BORE_LDS_BYTES;
  a.o_X = a.o_z = 0;
  if (a.data_in_lds) {
    a.o_X = (int)off; off += (size_t)N * L.w[0];
    a.o_z = (int)off; off += N;
  }
  {  // 129..512 rows and a flavour with an eight-wave kernel: a second shuffle buffer (fit_body `ahead`) -- behind
     // everything else and only when it still fits, so that it moves nothing and decides nothing
    const int fl = fit_flavour(desc);
    if (!perm && !g_batch && a.perm_in_lds && PG == 1 && N <= 512 && batch_size <= BORE_BATCH_MAX &&
        (fl == 2 || fl == 5 || fl == BORE_FIT_SHAPE_16_32 || (fl < 0 && fl >= -4)) &&
        (off + (size_t)N + tail) * 4 <= BORE_LDS_BYTES) {
      a.perm_ahead = 1;
      a.o_perm2 = (int)off; off += (size_t)N;
    }
  }
  a.total = (int)off;
  off = (off + 3) & ~(size_t)3;
  a.o_layout = (int)off;
  const size_t off_layout = off;
  // the constexpr-layout instantiation needs the layout it was compiled for (64-row tile)
  // (and keeps the Adam slots in LDS unconditionally)
  int shape = fit_flavour(desc);
  if (!a.perm_in_lds && shape > 0) {  // (only the generic flavours read the permutation from memory)
    if (bore_shape_is_wide(shape))
      return fail(BORE_E_UNSUPPORTED, "fit: N=%lld rows with this wide network: the shuffle must fit in LDS", (long long)N);
    shape = -desc->n_layers;
  }
  if (shape > 0 && !rounds && (!bore_shape_has_static_fit(shape) || (!a.state_in_lds && !bore_shape_is_wide(shape))))
    shape = -desc->n_layers;
  if (shape > 0 && bore_shape_is_wide(shape)) {
    // the wide static fit keeps m / v in HBM and its A / D images transposed (WideTp)
    static_assert(WideTp<3>::total() <= bore_static_layout(3, 1, BORE_BATCH_MAX).tile_floats,
                  "the transposed images must fit the tile region");
    if (a.state_in_lds) return fail(BORE_E_UNSUPPORTED, "fit: internal: wide shape with Adam slots in LDS");
  }
  // (the LDS copy of the layout tables is the generic flavour's alone, begin_kernel: the others read a constant
  // expression or the kernel arguments -- 496 B that decide whether three loops of the fused kernel share a CU)
  lds_floats = off_layout + (shape == 0 ? BORE_LAYOUT_FLOATS : 0);
  shape_out = shape;
  return 0;
}

// A float32 net with the widths and activations of a static shape (mlp_shapes.h) but FEWER inputs -- the plugin's
// default 32-32-1 on a four-dimensional search space -- would run the generic flavour (layout tables read at run
// time: 2.7x the static fit's time, tools/fit_generic_time.py).  Zero-padding is exact: padded inputs are 0, so the
// padded rows of the first layer see zero gradients and Adam leaves them (and their slots) at 0, and every sum
// gains terms that are exactly 0.  Which static shape the descriptor pads to, 0 if none:
static int bore_pads_to_shape(const bore_mlp_desc *d) {
  if (d->compute != BORE_COMPUTE_F32 || bore_match_shape(d)) return 0;  // (a static shape itself: nothing to pad)
  // (2->16-16-1, 6->32-32-1 and the two fit-only shapes: their static fits give the generic flavour's bits.  The
  // wide 16->64-64-64-1 fit does not -- same tolerance against the oracle, other low bits -- so a net padded onto it would change with the
  // path it takes: left on the generic flavour.)
  for (int s : {1, 2, 5, BORE_FIT_SHAPE_16_32, BORE_FIT_SHAPE_16_16}) {  // (the fewest zero columns first)
    if (!bore_flavour_built(s) || d->input_dim >= kShapes[s].D || d->n_layers != kShapes[s].n_layers) continue;
    bool ok = true;
    for (int i = 0; i < d->n_layers; ++i)
      ok = ok && d->units[i] == kShapes[s].units[i] && (kShapes[s].act[0] < 0 || d->act[i] == kShapes[s].act[i]) &&
           d->l2_kernel[i] == 0.f && d->l2_bias[i] == 0.f;
    if (ok) return s;
  }
  return 0;
}

// The fit of such a net on the static kernels: theta / m / v and X repacked into the padded shape's layout in a
// stream-ordered scratch (W_1 is the packed vector's first block, row-major [in][out]: the padded vector is the
// same prefix, a gap of zeros, the same suffix), the static fit, the way back.  rc < 0 with nothing launched when
// the padded shape's fit cannot take the request (the caller then runs the generic flavour).
static int fit_padded(const bore_mlp_desc *desc, int S, int n_models, float *theta, float *adam_m, float *adam_v,
                      int64_t *adam_t, const float *X, const float *z, int64_t N, int epochs, int batch_size,
                      const int32_t *perm, uint64_t seed, int64_t model_index0, int64_t epoch0,
                      const bore_adam_cfg *adam, float *epoch_loss, void *stream) {
  bore_mlp_desc dp = *desc;
  dp.input_dim = kShapes[S].D;
  const size_t D = desc->input_dim, DS = dp.input_dim, H = desc->units[0];
  const int64_t P = bore_param_count(desc), PS = bore_param_count(&dp);
  {  // (would the static fit take it?  the same checks, nothing launched)
    FitArgs probe;
    size_t off = 0;
    int shape = 0;
    const int rc = fit_build(&dp, n_models, theta, adam_m, adam_v, adam_t, X, z, N, epochs, batch_size, perm, seed,
                             model_index0, epoch0, adam, epoch_loss, probe, off, shape);
    if (rc != 0 || shape != S) return BORE_E_UNSUPPORTED;
  }
  hipStream_t st = (hipStream_t)stream;
  const size_t n_par = (size_t)n_models * PS, n_x = (size_t)n_models * N * DS;
  float *buf = nullptr;
  HIP_TRY(hipMallocAsync((void **)&buf, (3 * n_par + n_x) * sizeof(float), st));
  float *th_p = buf, *m_p = buf + n_par, *v_p = buf + 2 * n_par, *X_p = buf + 3 * n_par;
  const size_t head = D * H * 4, tail = ((size_t)P - D * H) * 4;  // bytes of W_1, and of everything behind it
  auto copy_rows = [&](float *dst, size_t dpitch, size_t doff, const float *src, size_t spitch, size_t soff,
                       size_t width, size_t rows) {
    return hipMemcpy2DAsync((char *)dst + doff, dpitch, (const char *)src + soff, spitch, width, rows,
                            hipMemcpyDeviceToDevice, st);
  };
  hipError_t e = hipMemsetAsync(buf, 0, (3 * n_par + n_x) * sizeof(float), st);
  const float *user[3] = {theta, adam_m, adam_v};
  float *padded[3] = {th_p, m_p, v_p};
  for (int k = 0; k < 3 && e == hipSuccess; ++k) {
    e = copy_rows(padded[k], PS * 4, 0, user[k], P * 4, 0, head, n_models);
    if (e == hipSuccess) e = copy_rows(padded[k], PS * 4, DS * H * 4, user[k], P * 4, head, tail, n_models);
  }
  if (e == hipSuccess) e = copy_rows(X_p, DS * 4, 0, X, D * 4, 0, D * 4, (size_t)n_models * N);
  int rc = e == hipSuccess ? 0 : fail(BORE_E_HIP, "fit (padded to a static shape): %s", hipGetErrorString(e));
  if (rc == 0)
    rc = bore_mlp_fit(&dp, n_models, th_p, m_p, v_p, adam_t, X_p, z, N, epochs, batch_size, perm, seed, model_index0,
                      epoch0, adam, epoch_loss, stream);
  float *back[3] = {theta, adam_m, adam_v};
  for (int k = 0; k < 3 && rc == 0; ++k) {
    e = copy_rows(back[k], P * 4, 0, padded[k], PS * 4, 0, head, n_models);
    if (e == hipSuccess) e = copy_rows(back[k], P * 4, head, padded[k], PS * 4, DS * H * 4, tail, n_models);
    if (e != hipSuccess) rc = fail(BORE_E_HIP, "fit (padded to a static shape): %s", hipGetErrorString(e));
  }
  (void)hipFreeAsync(buf, st);
  return rc;
}

extern "C" int bore_mlp_fit(const bore_mlp_desc *desc, int n_models, float *theta, float *adam_m,
                            float *adam_v, int64_t *adam_t, const float *X, const float *z,
                            int64_t N, int epochs, int batch_size, const int32_t *perm,
                            uint64_t seed, int64_t model_index0, int64_t epoch0,
                            const bore_adam_cfg *adam, float *epoch_loss, void *stream) {
  // (BORE_FIT_PAD = 0: such nets on the generic flavour, as before round 4 -- A/B, tests)
  if (desc && !g_batch && theta && adam_m && adam_v && adam_t && X && z && adam && epochs > 0 && n_models >= 1 &&
      N >= 1 && batch_size == BORE_BATCH_MAX && !(getenv("BORE_FIT_PAD") && !atoi(getenv("BORE_FIT_PAD")))) {
    const int S = bore_pads_to_shape(desc);
    if (S) {
      const int rc = fit_padded(desc, S, n_models, theta, adam_m, adam_v, adam_t, X, z, N, epochs, batch_size, perm,
                                seed, model_index0, epoch0, adam, epoch_loss, stream);
      if (rc != BORE_E_UNSUPPORTED) return rc;
    }
  }
  if (desc && desc->compute == BORE_COMPUTE_BF16) {
    if (g_batch) return fail(BORE_E_UNSUPPORTED, "fit: batch mode is float32 only");
    return fit_bf16_impl(desc, n_models, theta, adam_m, adam_v, adam_t, X, z, N, epochs, batch_size,
                         perm, seed, model_index0, epoch0, adam, epoch_loss, stream);
  }
  FitArgs a;
  size_t off = 0;
  int shape = 0;
  int rc = fit_build(desc, n_models, theta, adam_m, adam_v, adam_t, X, z, N, epochs, batch_size, perm,
                     seed, model_index0, epoch0, adam, epoch_loss, a, off, shape);
  if (rc) return rc < 0 ? rc : 0;
  {  // Eight waves (fit_kernel_w8): 6->32-32-1, and any net of up to four layers with more than four weight-gradient
     // tiles whose Adam slots are in LDS -- launches with no more models than CUs.  BORE_FIT_W8 = 0 / 1 forces the
     // choice (A/B, tests).
    const int forced = getenv("BORE_FIT_W8") ? atoi(getenv("BORE_FIT_W8")) : -1;
    int tiles = 0;
    for (int l = 1; l <= a.L.n_layers; ++l) tiles += (a.L.Np[l - 1] >> 4) * (a.L.Np[l] >> 4);
    const bool can = !g_batch && (shape == 2 || shape == 5 || shape == BORE_FIT_SHAPE_16_32 ||
                                  (shape < 0 && shape >= -4 && a.state_in_lds && tiles > 4));
    if (can && (forced < 0 ? n_models <= device_cus() : forced != 0)) {
#define BORE_LAUNCH_FIT_W8(S)                                                                               \
  case S:                                                                                                   \
    rc = allow_lds(fit_kernel_w8<S>, off * 4);                                                              \
    if (rc) return rc;                                                                                      \
    hipLaunchKernelGGL(fit_kernel_w8<S>, dim3(n_models), dim3(2 * BORE_THREADS), off * 4, (hipStream_t)stream, a); \
    HIP_TRY(hipGetLastError());                                                                             \
    return 0;
      switch (shape) {
#if BORE_ON_2
        BORE_LAUNCH_FIT_W8(2)
#endif
#if BORE_ON_5
        BORE_LAUNCH_FIT_W8(5)
#endif
#if BORE_ON_6
        BORE_LAUNCH_FIT_W8(BORE_FIT_SHAPE_16_32)
#endif
#if BORE_ON_N1
        BORE_LAUNCH_FIT_W8(-1)
#endif
#if BORE_ON_N2
        BORE_LAUNCH_FIT_W8(-2)
#endif
#if BORE_ON_N3
        BORE_LAUNCH_FIT_W8(-3)
#endif
#if BORE_ON_N4
        BORE_LAUNCH_FIT_W8(-4)
#endif
        default: break;
      }
#undef BORE_LAUNCH_FIT_W8
    }
  }
#define BORE_LAUNCH_FIT(S)                                                              \
  case S:                                                                               \
    rc = allow_lds(fit_kernel<S>, off * 4);                                             \
    if (rc) return rc;                                                                  \
    hipLaunchKernelGGL(fit_kernel<S>, dim3(n_models), dim3(BORE_THREADS), off * 4,      \
                       (hipStream_t)stream, a);                                         \
    break;
  if (!bore_flavour_built(shape)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
  switch (shape) {
#if BORE_ON_1
    BORE_LAUNCH_FIT(1)
#endif
#if BORE_ON_2
    BORE_LAUNCH_FIT(2)
#endif
#if BORE_ON_3
    BORE_LAUNCH_FIT(3)
#endif
#if BORE_ON_4
    BORE_LAUNCH_FIT(4)
#endif
#if BORE_ON_5
    BORE_LAUNCH_FIT(5)
#endif
#if BORE_ON_6
    BORE_LAUNCH_FIT(BORE_FIT_SHAPE_16_32)
#endif
#if BORE_ON_7
    BORE_LAUNCH_FIT(BORE_FIT_SHAPE_16_16)
#endif
#if BORE_ON_N1
    BORE_LAUNCH_FIT(-1)
#endif
#if BORE_ON_N2
    BORE_LAUNCH_FIT(-2)
#endif
#if BORE_ON_N3
    BORE_LAUNCH_FIT(-3)
#endif
#if BORE_ON_N4
    BORE_LAUNCH_FIT(-4)
#endif
    default:
#if !BORE_ON_0
      return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
#endif
#if BORE_ON_0
    BORE_LAUNCH_FIT(0)
#endif
  }
#undef BORE_LAUNCH_FIT
  HIP_TRY(hipGetLastError());
  return 0;
}

static int fit_bf16_impl(const bore_mlp_desc *desc, int n_models, float *theta, float *adam_m,
                         float *adam_v, int64_t *adam_t, const float *X, const float *z, int64_t N,
                         int epochs, int batch_size, const int32_t *perm, uint64_t seed,
                         int64_t model_index0, int64_t epoch0, const bore_adam_cfg *adam,
                         float *epoch_loss, void *stream) {
  FitBf16Args a;
  if (batch_size != BORE_BATCH_MAX)
    return fail(BORE_E_UNSUPPORTED, "fit_bf16: batch_size must be %d (got %d)", BORE_BATCH_MAX,
                batch_size);
  if (N < 1 || N > (1 << 20)) return fail(BORE_E_INVALID, "fit_bf16: N=%lld out of range", (long long)N);
  if (epochs < 0) return fail(BORE_E_INVALID, "fit_bf16: epochs < 0");
  if (n_models < 1) return fail(BORE_E_INVALID, "n_models must be >= 1 (got %d)", n_models);
  if (!desc || !theta || !adam_m || !adam_v || !adam_t || !X || !z || !adam)
    return fail(BORE_E_INVALID, "fit_bf16: null pointer");
  const int shape = bore_match_shape(desc);
  if (!bore_shape_is_wide(shape))
    return fail(BORE_E_UNSUPPORTED, kBf16Shapes);
  if (bore_make_layout(desc, 1, BORE_BATCH_MAX, &a.L)) return fail(BORE_E_INVALID, "bad bore_mlp_desc");
  if (epochs == 0) return 0;
  a.theta = theta; a.am = adam_m; a.av = adam_v; a.at = (long long *)adam_t;
  a.X = X; a.z = z; a.perm = perm; a.epoch_loss = epoch_loss;
  a.seed = seed; a.model0 = model_index0; a.epoch0 = epoch0;
  a.N = (int)N; a.epochs = epochs; a.B = batch_size;
  a.lr = adam->lr; a.beta1 = adam->beta1; a.beta2 = adam->beta2; a.eps = adam->eps;
  // The bf16-MFMA kernel keeps two fragment-order weight images in LDS; with a long shuffle
  // (perm + keys grow with N) it no longer fits beside them and the fp32-MFMA form takes over.
  const int PGn = perm ? 1 : perm_group(N, BORE_THREADS);
  const size_t perm_bytes = 4 * (size_t)PGn * N + (perm ? 0 : 4 * (size_t)perm_group_scratch_floats(N, PGn)) + 96;
  const size_t new_bytes = (shape == 3 ? Bf16Plan<3>::o_end : Bf16Plan<4>::o_end) + perm_bytes;
  const bool old_form = new_bytes > BORE_LDS_BYTES;
  // LDS carve (bytes).  bf16-MFMA form: weight images | biases | A^T / D^T images (= gradient
  // image) | misc | perm | keys;  fp32-MFMA form: theta bf16 | A/D copies bf16 | misc | perm | keys
  size_t off;
  if (old_form) {
    off = 2 * (size_t)a.L.P_lds;
    off = (off + 15) & ~(size_t)15;
    // (the weight-gradient phase parks the packed fp32 gradient image over the A / D copies)
    a.o_tile = (int)off;
    off += 2 * (size_t)a.L.tile_floats > 4 * (size_t)a.L.P ? 2 * (size_t)a.L.tile_floats : 4 * (size_t)a.L.P;
  } else {
    a.o_tile = 0;
    off = shape == 3 ? Bf16Plan<3>::o_end : Bf16Plan<4>::o_end;
    static_assert(Bf16Plan<3>::fits() && Bf16Plan<4>::fits(), "a layer's gradients must fit the image region");
  }
  off = (off + 15) & ~(size_t)15;
  a.o_misc = (int)off; off += 8 * 4;
  const int PG = perm ? 1 : perm_group(N, BORE_THREADS);
  a.o_perm = (int)off; off += 4 * (size_t)PG * N;
  off = (off + 15) & ~(size_t)15;
  a.o_keys = (int)off; off += perm ? 0 : 4 * (size_t)perm_group_scratch_floats(N, PG);
  off = (off + 15) & ~(size_t)15;
  a.total = (int)off;
  if (off > BORE_LDS_BYTES)
    return fail(BORE_E_UNSUPPORTED, "fit_bf16: weights+images+perm need %zu B of LDS (> %d)", off,
                BORE_LDS_BYTES);
  int rc = 0;
#define BORE_LAUNCH_BF16(KERNEL)                                                             \
  {                                                                                          \
    rc = allow_lds(KERNEL, off);                                                             \
    if (rc) return rc;                                                                       \
    hipLaunchKernelGGL(KERNEL, dim3(n_models), dim3(BORE_THREADS), off, (hipStream_t)stream, a); \
  }
  if (!bore_flavour_built(shape)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
  if (old_form) {
#if BORE_ON_3
    if (shape == 3) BORE_LAUNCH_BF16(fit_bf16_kernel<3>)
#endif
#if BORE_ON_4
    if (shape == 4) BORE_LAUNCH_BF16(fit_bf16_kernel<4>)
#endif
  } else {
#if BORE_ON_3
    if (shape == 3) BORE_LAUNCH_BF16(fit_bf16_mfma_kernel<3>)
#endif
#if BORE_ON_4
    if (shape == 4) BORE_LAUNCH_BF16(fit_bf16_mfma_kernel<4>)
#endif
  }
#undef BORE_LAUNCH_BF16
  HIP_TRY(hipGetLastError());
  return 0;
}

static int row_launch(bool with_grad, int n_models, RowArgs &a, void *stream) {
  const MlpLayout &L = a.L;
  size_t off = L.P_lds;
  // (a bfloat16 model: the fragment-order weight images of arg_bf16_mfma.h, and no tile)
  if (a.bf16) off = a.shape == 3 ? ArgBf16Plan<3>::floats : ArgBf16Plan<4>::floats;
  a.o_tile = (int)off; off += a.bf16 ? 0 : L.tile_floats;
  a.o_vals = (int)off; off += BORE_BATCH_MAX;  // objective values of the tile rows
  a.total = (int)off;
  off = (off + 3) & ~(size_t)3;
  a.o_layout = (int)off; off += BORE_LAYOUT_FLOATS;
  const int waves = a.bf16 ? BORE_THREADS / 64 : L.tbp >> 4;
  const long long n_blocks = (a.n_rows + 15) / 16;
  // enough workgroups to fill 256 CUs a few times over, never more than there is work
  long long gy = (n_blocks + waves - 1) / waves;
  const long long cap = (2048 + n_models - 1) / n_models;
  if (gy > cap) gy = cap < 1 ? 1 : cap;
  if (gy > 65535) gy = 65535;
  int rc = 0;
  const int shape = a.shape;  // (the static flavours keep their row-blocks in registers: no tile)
#define BORE_LAUNCH_ROWS(G, S)                                                              \
  {                                                                                         \
    rc = allow_lds(rows_kernel<G, S>, off * 4);                                             \
    if (rc) return rc;                                                                      \
    hipLaunchKernelGGL((rows_kernel<G, S>), dim3(n_models, (unsigned)gy), dim3(BORE_THREADS), \
                       off * 4, (hipStream_t)stream, a);                                    \
  }
#define BORE_LAUNCH_ROWS16(G, S)                                                               \
  {                                                                                            \
    rc = allow_lds((rows_kernel<G, S, true>), off * 4);                                        \
    if (rc) return rc;                                                                         \
    hipLaunchKernelGGL((rows_kernel<G, S, true>), dim3(n_models, (unsigned)gy),                \
                       dim3(BORE_THREADS), off * 4, (hipStream_t)stream, a);                   \
  }
  if (!bore_flavour_built(shape)) return fail(BORE_E_UNSUPPORTED, BORE_FLAVOUR_LEFT_OUT);
  if (a.bf16) {  // (the entry points have checked: wide static shape)
#if BORE_ON_3
    if (shape == 3 && with_grad) BORE_LAUNCH_ROWS16(true, 3)
    if (shape == 3 && !with_grad) BORE_LAUNCH_ROWS16(false, 3)
#endif
#if BORE_ON_4
    if (shape == 4 && with_grad) BORE_LAUNCH_ROWS16(true, 4)
    if (shape == 4 && !with_grad) BORE_LAUNCH_ROWS16(false, 4)
#endif
  } else {
#define BORE_ROWS_BOTH(S)                      \
  if (shape == (S)) {                          \
    if (with_grad) BORE_LAUNCH_ROWS(true, S)   \
    else BORE_LAUNCH_ROWS(false, S)            \
  }
#if BORE_ON_1
    BORE_ROWS_BOTH(1)
#endif
#if BORE_ON_2
    BORE_ROWS_BOTH(2)
#endif
#if BORE_ON_3
    BORE_ROWS_BOTH(3)
#endif
#if BORE_ON_4
    BORE_ROWS_BOTH(4)
#endif
#if BORE_ON_5
    BORE_ROWS_BOTH(5)
#endif
#if BORE_ON_N1
    BORE_ROWS_BOTH(-1)
#endif
#if BORE_ON_N2
    BORE_ROWS_BOTH(-2)
#endif
#if BORE_ON_N3
    BORE_ROWS_BOTH(-3)
#endif
#if BORE_ON_N4
    BORE_ROWS_BOTH(-4)
#endif
#if BORE_ON_0
    BORE_ROWS_BOTH(0)
#endif
#undef BORE_ROWS_BOTH
  }
#undef BORE_LAUNCH_ROWS
#undef BORE_LAUNCH_ROWS16
  HIP_TRY(hipGetLastError());
  return 0;
}

extern "C" int bore_mlp_forward(const bore_mlp_desc *desc, int n_models, const float *theta,
                                const float *X, int64_t n_rows, int x_shared, float *out,
                                void *stream) {
  RowArgs a;
  int rc = check_common(desc, n_models, 0, BORE_BATCH_MAX, true,
                        BORE_BATCH_MAX + BORE_LAYOUT_FLOATS + 4, &a.L);
  if (rc) return rc;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "forward: the last Dense layer must have 1 unit");
  if (!theta || !X || !out) return fail(BORE_E_INVALID, "forward: null pointer");
  if (n_rows < 0) return fail(BORE_E_INVALID, "forward: n_rows < 0");
  if (n_rows == 0) return 0;
  a.theta = theta; a.Xf = X; a.Xd = nullptr; a.out = out; a.grad = nullptr;
  a.n_rows = n_rows; a.x_shared = x_shared; a.transform = 0; a.sign = 1.f;
  a.shape = bore_acq_flavour(desc, true);
  a.d_in = desc->input_dim;
  a.bf16 = desc->compute == BORE_COMPUTE_BF16;
  if (a.bf16 && !bore_shape_is_wide(a.shape)) return fail(BORE_E_UNSUPPORTED, kBf16Shapes);
  return row_launch(false, n_models, a, stream);
}

extern "C" int bore_mlp_value_and_input_grad(const bore_mlp_desc *desc, int n_models,
                                             const float *theta, const double *X, int64_t n_rows,
                                             int transform, int negate, float *val,
                                             double *grad, void *stream) {
  RowArgs a;
  int rc = check_common(desc, n_models, 2, BORE_BATCH_MAX, true,
                        BORE_BATCH_MAX + BORE_LAYOUT_FLOATS + 4, &a.L);
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
  a.shape = bore_acq_flavour(desc, true);
  a.d_in = desc->input_dim;
  a.bf16 = desc->compute == BORE_COMPUTE_BF16;
  if (a.bf16 && !bore_shape_is_wide(a.shape)) return fail(BORE_E_UNSUPPORTED, kBf16Shapes);
  return row_launch(true, n_models, a, stream);
}

extern "C" int bore_mlp_evaluate(const bore_mlp_desc *desc, int n_models, const float *theta,
                                 const float *X, const float *z, int64_t N, float *loss,
                                 float *acc, void *stream) {
  EvalArgs a;
  int rc = check_common(desc, n_models, 0, BORE_BATCH_MAX, true, 8 + BORE_LAYOUT_FLOATS + 4, &a.L);
  if (rc) return rc;
  if (a.L.w[a.L.n_layers] != 1)
    return fail(BORE_E_INVALID, "evaluate: the last Dense layer must have 1 unit");
  if (!theta || !X || !z || !loss || !acc) return fail(BORE_E_INVALID, "evaluate: null pointer");
  if (N < 1) return fail(BORE_E_INVALID, "evaluate: N < 1");
  a.theta = theta; a.X = X; a.z = z; a.loss = loss; a.acc = acc; a.N = N;
  size_t off = a.L.P_lds;
  a.o_tile = (int)off; off += a.L.tile_floats;
  a.o_misc = (int)off; off += 8;
  a.total = (int)off;
  off = (off + 3) & ~(size_t)3;
  a.o_layout = (int)off; off += BORE_LAYOUT_FLOATS;
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
  // BORE_SHUFFLE_WAVE = 1 (tests): N <= 128 rows through make_perm_wave_buckets, the form the
  // pipelined fit uses inside its steps -- the same permutations
  const int wave_form = N <= 128 && getenv("BORE_SHUFFLE_WAVE") && atoi(getenv("BORE_SHUFFLE_WAVE")) ? 1 : 0;
  const size_t bytes = (wave_form ? (size_t)BORE_PERM_WAVE_FLOATS : (size_t)perm_scratch_floats(N)) * 4;
  int rc = allow_lds(shuffle_kernel, bytes);
  if (rc) return rc;
  hipLaunchKernelGGL(shuffle_kernel, dim3(n_models, epochs), dim3(BORE_THREADS), bytes,
                     (hipStream_t)stream, seed, (long long)model_index0, (long long)epoch0, epochs,
                     (int)N, perm, wave_form);
  HIP_TRY(hipGetLastError());
  return 0;
}
