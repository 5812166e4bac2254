// lbfgsb.h -- bound-constrained limited-memory BFGS (L-BFGS-B), written from the
// published algorithm for execution INSIDE a HIP kernel, one problem per lane.
//
// What it stands in for: scipy.optimize.minimize(method="L-BFGS-B", jac=True, bounds)
// as called by the reference at bore/mixins.py:59-60 (SciPy is a third-party dependency
// of the reference, pinned scipy==1.7.0 in setup.py:15; its core is the Fortran/C code
// "L-BFGS-B 3.0" of Zhu, Byrd, Lu, Nocedal & Morales).  The algorithm restated here:
//   Byrd, Lu, Nocedal, Zhu, "A limited memory algorithm for bound constrained
//   optimization", SIAM J. Sci. Comput. 16 (1995): generalized Cauchy point (sec. 4),
//   subspace minimisation by the direct primal method (sec. 5.1), compact limited-memory
//   matrices (sec. 3);
//   Morales, Nocedal, "Remark on Algorithm 778" (2011): projected subspace step with
//   backtracking fallback;
//   More', Thuente, "Line search algorithms with guaranteed sufficient decrease",
//   ACM TOMS 20 (1994): dcsrch / dcstep, with ftol=1e-3, gtol=0.9, xtol=0.1.
// Same stopping rules, constants and failure modes as the SciPy driver
// (_minimize_lbfgsb): projected-gradient test (pgtol), relative-reduction test
// (factr*epsmch), maxiter / maxfun -> status 1, abnormal line search -> status 2, at
// most `maxls` function evaluations per line search, memory refresh on a failed
// factorisation or line search.
//
// The routine is a REVERSE-COMMUNICATION state machine: lbfgsb_advance() runs until the
// problem needs f and g at `x` (returns LB_NEED_FG) or has terminated (LB_DONE).  That is
// what lets a workgroup evaluate all its problems' points in one cooperative MLP pass.
//
// All arithmetic is fp64.  Results are not bit-identical to SciPy (its BLAS sums in a
// different order) but follow the same trajectory to rounding; tests/test_lbfgsb_host.py
// compares iterates, iteration and evaluation counts against scipy on the CPU build.
#pragma once
#include <math.h>
#include <stdint.h>

#include <type_traits>

#if defined(__HIPCC__)
#define LB_HD __host__ __device__ __forceinline__
#define LB_HDN __host__ __device__ __forceinline__  // (real calls measured slower: +20 % kernel time)
#else
#define LB_HD inline
#define LB_HDN
#endif

namespace lbfgsb {

enum { LB_NEED_FG = 1, LB_DONE = 2 };

// task codes: the (task[0], task[1]) pairs of scipy's status_messages/task_messages
enum {
  T_START = 0, T_NEW_X = 1, T_FG = 3, T_CONVERGENCE = 4, T_STOP = 5, T_ERROR = 7, T_ABNORMAL = 8
};
enum {
  M_NONE = 0, M_PGTOL = 401, M_FACTR = 402, M_MAXFUN = 502, M_MAXITER = 504,
  M_NO_FEASIBLE = 701, M_FACTR_NEG = 702, M_M_LE_0 = 711, M_N_LE_0 = 712, M_INVALID_NBD = 713
};

// stages of the state machine (where to resume after the caller supplied f and g)
enum { S_INIT = 0, S_FG_START = 1, S_FG_LNSRCH = 2, S_FINISHED = 3 };

struct Options {
  int m;          // maxcor
  double factr;   // ftol / epsmch
  double pgtol;   // gtol
  int maxiter, maxfun, maxls;
};

// Fixed-size scalar state of one problem.  The vectors/matrices live in a caller-provided
// double/int workspace (sizes from dwork_size()/iwork_size()).
struct State {
  // problem
  int n, m;
  // mainlb scalars
  double theta, f, fold, sbgnrm, dnorm, stp, gd, gdold, dtd, xstep, stpmx;
  int col, head, itail, iupdat, iter, nfgv, ifun, iback, nfree, nact, nenter, ileave, nseg;
  int info, iword;
  int prjctd, cnstnd, boxed, updatd, wrk;
  // dcsrch saved locals
  int brackt, ls_stage, ls_task;  // ls_task: 0 START, 1 FG, 2 CONVERGENCE, 3 WARNING, 4 ERROR
  double ginit, gtest, gx, gy, finit, fx, fy, stx, sty, stmin, stmax, width, width1;
  // driver
  int stage, task, msg, nit, nfev, status;
  double flast;
#if defined(BORE_STAMPS)
  long long lb_last;  // (diagnostic builds: time of the previous mark, LB_MARK)
#endif
};

// Every sub-array starts on a 16-byte boundary (vectors are padded to an even length) and the
// caller must hand in a 16-byte aligned base: the device compiler merges adjacent fp64 LDS
// loads into ds_read_b128, which returns WRONG data at 8-byte alignment on gfx950 (found as a
// host/device mismatch for odd n; see tests/test_gpu_argmax.py).
// The two 2m x 2m matrices of the subspace minimisation share ONE (2m) x (2m + 1) block (round 6): formk keeps
// the LOWER triangle of `snd` (the "WN1" of the published code: blocks (1,1), (2,1), (2,2) by rows >= columns) and
// builds / factors the UPPER triangle of `wn`; dpofa and the triangular solves touch the upper triangle only (the
// wavefront forms read, and never use, entries below the diagonal).  With `wn` one column to the right of `snd`
// -- wn(i, j) at block[(j + 1) * 2m + i], i <= j; snd(i, j) at block[j * 2m + i], i >= j -- the two triangles tile
// the block exactly, every index expression keeps its leading dimension 2m and its 16-byte column alignment, and
// a problem's workspace loses 4 m^2 - 2 m doubles (m = 10: 3 040 of 11.3 KB at 6 variables) -- which is what bounds
// the problems, i.e. the latency chains, in flight per CU.  Same arithmetic, same bits.
// big_outside: that block lives in a buffer of the caller's (big_size doubles, 16-byte aligned,
// any address space -- the device kernels put it in global memory when that lets more problems share
// the LDS) instead of inside dw.
LB_HD int big_size(int m) { return 2 * m * (2 * m + 1); }
LB_HD int dwork_size(int n, int m, bool big_outside = false) {
  const int ne = (n + 1) & ~1, mne = (m * n + 1) & ~1, mm = (m * m + 1) & ~1;
  return 2 * mne + 3 * mm + (big_outside ? 0 : big_size(m)) + 8 * m + 9 * ne + 6 * m;
}
LB_HD int iwork_size(int n) { return 3 * n; }

// Views into the workspaces (all 0-based; matrices column-major like the original).
struct Work {
  double *ws, *wy;          // [m][n]: correction j is ws[j*n .. j*n+n)
  double *sy, *ss, *wt;     // m x m, element (i,j) at [j*m + i]
  double *wn, *snd;         // 2m x 2m, element (i,j) at [j*2m + i]; wn == snd + 2m: two triangles of one block (above)
  double *z, *r, *d, *t, *xp, *x, *g;
  double *xlast, *glast;    // last point actually evaluated (SciPy's ScalarFunction cache)
  double *wa;               // 8m: p | c | wbp | v
  // Reciprocals kept beside the factors: the substitution chains multiply by them instead of
  // dividing (an fp64 divide is ~150 cycles of dependent issue per unknown on the device).
  double *rwt;              // m:  1 / diag of the Cholesky factor in wt
  double *rwn;              // 2m: 1 / diag of the two Cholesky factors in wn
  double *rsy, *rsq;        // m each: 1 / sy_ii and 1 / sqrt(sy_ii)
  int *index, *iwhere, *indx2;
  // wn / snd are outside the caller's dw (make_work's `big`: global memory on the device).  0: inside dw, zeroed
  // with it by the caller; 1: outside, zeroed by the caller; 2: outside and NOT zeroed -- lbfgsb_advance zeroes them
  // itself, before formk first touches them (LB_BIG_LAZY)
  int vm;
};

LB_HD Work make_work(double *dw, int *iw, int n, int m, double *big = nullptr) {
  Work w;
  const int ne = (n + 1) & ~1, mne = (m * n + 1) & ~1, mm = (m * m + 1) & ~1;
  w.ws = dw; dw += mne;
  w.wy = dw; dw += mne;
  w.sy = dw; dw += mm;
  w.ss = dw; dw += mm;
  w.wt = dw; dw += mm;
  w.vm = big != nullptr;
  if (big) {  // (dwork_size(n, m, true) + big_size(m))
    w.snd = big;
  } else {
    w.snd = dw; dw += big_size(m);
  }
  w.wn = w.snd + 2 * m;  // (one column to the right: the upper triangle beside snd's lower one)
  w.z = dw; dw += ne;
  w.r = dw; dw += ne;
  w.d = dw; dw += ne;
  w.t = dw; dw += ne;
  w.xp = dw; dw += ne;
  w.x = dw; dw += ne;
  w.g = dw; dw += ne;
  w.xlast = dw; dw += ne;
  w.glast = dw; dw += ne;
  w.wa = dw; dw += 8 * m;
  w.rwt = dw; dw += m;
  w.rwn = dw; dw += 2 * m;
  w.rsy = dw; dw += m;
  w.rsq = dw;
  w.index = iw;
  w.iwhere = iw + n;
  w.indx2 = iw + 2 * n;
  return w;
}

#define LB_EPSMCH 2.220446049250313e-16

// ---- lanes cooperating on ONE problem ---------------------------------------------------
// When a wave owns a single problem, all 64 lanes run the state machine in lock-step with
// identical scalars (uniform control flow, redundant stores of identical values) and SHARE
// the loops whose iterations produce independent outputs: lane `lane` of `nl` takes
// iterations lane, lane+nl, ...  Every output is still computed by one lane with the
// sequential operation order, so results do not depend on nl (host build: nl = 1).
// LB_LANES_SYNC separates such a loop from readers of its outputs in other lanes.
struct Coop {
  int lane, nl;
};
// LB_UNI: an integer of the State that every lane of a cooperating wave holds alike (c.nl > 1: ONE
// problem per wave), moved to a scalar register.  The State lives in vector registers and is updated
// under tests on fp64 values, which the compiler must take for lane-dependent: without the hint every
// loop over col / nfree / ... is an execution-mask loop with its counter and bounds in vector registers.
// (Tried beside it and dropped: the line search's own tests as scalar branches -- the compare's lane mask
// against zero -- instead of execution-mask regions: no change, profiles/r3/ab_headline.txt.)
#if defined(__HIP_DEVICE_COMPILE__)
#define LB_UNI(v, c) ((c).nl > 1 ? __builtin_amdgcn_readfirstlane(v) : (v))
#else
#define LB_UNI(v, c) (v)
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define LB_LANES_SYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
// Where lanes hand entries of the 2m x 2m matrices to each other (formk, dpofa): those may live in
// global memory (make_work's `big`, Work::vm) -- then the vector-memory counter is waited for as well.
// (Not everywhere: a kernel with register spills has scratch stores in flight all the time, and waiting
// for them at every hand-over cost the 32-32-1 restart kernel 30 %.)
#define LB_LANES_SYNC_VM(vm)                                          \
  do {                                                                \
    if (vm) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          \
    LB_LANES_SYNC();                                                  \
  } while (0)
#define LB_OPAQUE_LANE(x) asm volatile("" : "+v"(x))
#else
#define LB_LANES_SYNC() ((void)0)
#define LB_LANES_SYNC_VM(vm) ((void)(vm))
#define LB_OPAQUE_LANE(x) ((void)0)
#endif

// Work::vm == LB_BIG_LAZY: the two 2m x 2m matrices (big_size(m) doubles from w.snd on: make_work) start as whatever the
// buffer held.  formk builds them up incrementally and counts on zeros (SciPy hands setulb a zeroed work array), but
// only a problem that reaches a subspace minimisation ever looks at them -- one in twenty of BASELINE config 5's,
// whose launches spent 6.4 GB of writes per million restarts on zeroing pool slots nobody read.
enum { LB_BIG_INSIDE = 0, LB_BIG_ZEROED = 1, LB_BIG_LAZY = 2 };
LB_HD void zero_big(const Work &w, int m, const Coop c) {
  for (int i = 2 * c.lane; i < big_size(m); i += 2 * c.nl) {  // (pairs: 16-byte stores; 2m (2m + 1) is even)
    w.snd[i] = 0.0;
    w.snd[i + 1] = 0.0;
  }
  LB_LANES_SYNC_VM(1);
}

// -DBORE_STAMPS: per-phase cycle accumulators, diagnostics only.  g_lb_phase: workgroup 0,
// thread 0.  g_lb_pp[q][i]: every problem q = 4*blockIdx.x + wave of a one-problem-per-wave launch
// (i < 16: cycles in phase i, 16 + i: calls; 13 / 14: whole advance / f-g cycles, added by the kernel;
// phases: 0 cauchy 1 formk 2 cmprlb 3 subsm 4 lnsrlb 5 matupd 6 formt 7 head of advance (saves of the
// evaluated point) 8 freev 9 accept (projgr + tests) 10 cache check 11 BFGS pair (r, rr, d) 12 d = z - x).
#if defined(BORE_STAMPS) && defined(__HIPCC__)
__device__ long long g_lb_phase[16];
#define LB_PP_MAX 4096
__device__ unsigned long long g_lb_pp[LB_PP_MAX][64];
// Accumulated per wave in LDS (ds_add without return: no round trip in the optimiser's chain; the
// r2 form added to global memory with returning atomics, ~2 k cycles per stamped phase); the kernel
// flushes a wave's row to g_lb_pp when its problem ends.
__shared__ unsigned g_lb_lds[16][64];
#endif
#if defined(BORE_STAMPS) && defined(__HIP_DEVICE_COMPILE__)
// Every cycle of lbfgsb_advance lands in exactly one bucket: LB_PHASE_BEGIN(g) charges the time
// since the previous mark to the GAP bucket g (what precedes the phase: control flow, copies between
// register assignments, ...), LB_PHASE_END(i) charges the time since then to phase i.
#define LB_MARK(s_, i)                                                                    \
  do {                                                                                    \
    const long long lb_now_ = clock64();                                                  \
    const unsigned lb_dt_ = (unsigned)(lb_now_ - (s_).lb_last);                           \
    (s_).lb_last = lb_now_;                                                               \
    if ((threadIdx.x & 63) == 0) {                                                        \
      __hip_atomic_fetch_add(&g_lb_lds[(threadIdx.x >> 6) & 15][i], lb_dt_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
      __hip_atomic_fetch_add(&g_lb_lds[(threadIdx.x >> 6) & 15][32 + (i)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); \
    }                                                                                     \
  } while (0)
#if defined(BORE_STAMPS_FORMK) || defined(BORE_STAMPS_CAUCHY)
// (-DBORE_STAMPS_FORMK: the gaps are lumped into bucket 15 and their buckets 19..27 take the stages of
// formk instead: new rows, old parts, assembly, first factorisation, triangular solves, (2,2) block,
// second factorisation; tools/engine_phases.py formk.  -DBORE_STAMPS_CAUCHY: the stages of cauchy -- 19 variables
// classified + sums over the moving ones, 20 first product with the middle matrix, per breakpoint 21 the heap, 22 the
// workspace updates, 23 the middle-matrix part (col > 0), 24 the tail; tools/wide_phases.py)
#define LB_PHASE_BEGIN(g) LB_MARK(s, 15)
struct LbLocalClock { long long lb_last; };
#ifdef BORE_STAMPS_CAUCHY
#define CK_MARK_DECL LbLocalClock ck_clock{clock64()}
#define CK_MARK(i) LB_MARK(ck_clock, i)
#else
#define FK_MARK_DECL LbLocalClock fk_clock{clock64()}
#define FK_MARK(i) LB_MARK(fk_clock, i)
#endif
#else
#define LB_PHASE_BEGIN(g) LB_MARK(s, g)
#endif
#define LB_PHASE_END(i) LB_MARK(s, i)
#else
#define LB_MARK(s_, i) ((void)0)
#define LB_PHASE_BEGIN(g) ((void)0)
#define LB_PHASE_END(i) ((void)0)
#endif

#ifndef FK_MARK
#define FK_MARK_DECL ((void)0)
#define FK_MARK(i) ((void)0)
#endif
#ifndef CK_MARK
#define CK_MARK_DECL ((void)0)
#define CK_MARK(i) ((void)0)
#endif

// The value `v` holds in lane `src` (wave-uniform), in every lane of the wave.
#if defined(__HIP_DEVICE_COMPILE__)
LB_HD double lane_bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
LB_HD bool lanes_any(bool p) { return __any(p) != 0; }
LB_HD unsigned long long lanes_ballot(bool p) { return __ballot(p); }
LB_HD int bits_below(unsigned long long mask, int lane) {  // set bits of mask below bit `lane`
  return __popcll(mask & ((1ull << lane) - 1ull));
}
LB_HD int bits_set(unsigned long long mask) { return __popcll(mask); }

#else
LB_HD double lane_bcast(double v, int) { return v; }
LB_HD bool lanes_any(bool p) { return p; }
LB_HD unsigned long long lanes_ballot(bool p) { return p ? 1ull : 0ull; }
LB_HD int bits_below(unsigned long long mask, int lane) {
  int c = 0;
  for (int i = 0; i < lane; ++i) c += (int)((mask >> i) & 1ull);
  return c;
}
LB_HD int bits_set(unsigned long long mask) { return bits_below(mask, 64); }
#endif

// p mod m for 0 <= p < 2m (circular indices into the m correction pairs; no integer divide)
LB_HD int wrap(int p, int m) { return p >= m ? p - m : p; }

// sum_{i < n} term_i, left to right, where lane i holds term_i (n <= the number of lanes): the
// same additions in the same order as the sequential loop, fed from registers instead of memory.
LB_HD double lanes_sum_ordered(double term, int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += lane_bcast(term, i);
  return s;
}

#if defined(__HIP_DEVICE_COMPILE__)
// a float64 as an unsigned that sorts like it (-0.0 first made +0.0: the sequential code's `<` does not tell them apart)
LB_HD unsigned long long orderable_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v + 0.0);
  return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
// the smallest of the 64 lanes' keys, in every lane: four DPP shifts inside each row of 16 lanes (a lane without a
// source keeps its own), the four rows' results by v_readlane -- no LDS round trips (__shfl_xor: six of them)
LB_HD unsigned long long wave_min_key(unsigned long long v) {
  unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
#define LB_MIN_STEP(ctrl)                                                                        \
  {                                                                                              \
    const unsigned plo = (unsigned)__builtin_amdgcn_update_dpp((int)lo, (int)lo, ctrl, 0xf, 0xf, false); \
    const unsigned phi = (unsigned)__builtin_amdgcn_update_dpp((int)hi, (int)hi, ctrl, 0xf, 0xf, false); \
    const bool less = phi < hi || (phi == hi && plo < lo);                                       \
    lo = less ? plo : lo;                                                                        \
    hi = less ? phi : hi;                                                                        \
  }
  LB_MIN_STEP(0x111)  // row_shr:1
  LB_MIN_STEP(0x112)  // row_shr:2
  LB_MIN_STEP(0x114)  // row_shr:4
  LB_MIN_STEP(0x118)  // row_shr:8
#undef LB_MIN_STEP
  unsigned long long best = ~0ull;
#pragma unroll
  for (int r = 15; r < 64; r += 16) {
    const unsigned long long k = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)hi, r) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((int)lo, r);
    best = k < best ? k : best;
  }
  return best;
}
#endif

// ---- small dense kernels ------------------------------------------------------
// Sums strictly left to right (results do not depend on the unrolling); operands are fetched
// eight at a time so that their LDS latencies overlap -- on the device a dependent
// load-multiply-add per element costs a full LDS round trip (~130 cycles).
LB_HD double ddot(int n, const double *a, const double *b) {
  double s = 0.0;
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    const double a0 = a[i], a1 = a[i + 1], a2 = a[i + 2], a3 = a[i + 3];
    const double b0 = b[i], b1 = b[i + 1], b2 = b[i + 2], b3 = b[i + 3];
    s += a0 * b0;
    s += a1 * b1;
    s += a2 * b2;
    s += a3 * b3;
  }
  for (; i < n; ++i) s += a[i] * b[i];
  return s;
}

// ddot for lanes that cooperate on one problem: lane i forms term i, the sum runs over the lanes in
// the order of i (bit for bit ddot's result)
// (VL = false compiles the variable-per-lane forms out: the 2-D problems of the fused iteration
// kernel gain nothing from them, and their code in that kernel cost its fit phase 5 %)
template <bool VL = true>
LB_HD double ddot_c(int n, const double *a, const double *b, const Coop c) {
  if (VL && c.nl > 1 && n <= c.nl) return lanes_sum_ordered(c.lane < n ? a[c.lane] * b[c.lane] : 0.0, n);
  return ddot(n, a, b);
}

// sum_i a[i*sa] * b[i*sb] * r[i*sr], left to right (r: stored reciprocals), operands fetched
// four terms at a time
LB_HD double dot_rcp(int n, const double *a, int sa, const double *b, int sb, const double *r,
                     int sr) {
  double s = 0.0;
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    const double a0 = a[i * sa], a1 = a[(i + 1) * sa], a2 = a[(i + 2) * sa], a3 = a[(i + 3) * sa];
    const double b0 = b[i * sb], b1 = b[(i + 1) * sb], b2 = b[(i + 2) * sb], b3 = b[(i + 3) * sb];
    const double r0 = r[i * sr], r1 = r[(i + 1) * sr], r2 = r[(i + 2) * sr], r3 = r[(i + 3) * sr];
    s += a0 * b0 * r0;
    s += a1 * b1 * r1;
    s += a2 * b2 * r2;
    s += a3 * b3 * r3;
  }
  for (; i < n; ++i) s += a[i * sa] * b[i * sb] * r[i * sr];
  return s;
}

// y[i] += alpha * x[i], i < n  (x and y do not overlap)
LB_HD void daxpy(int n, double alpha, const double *x, double *y) {
  int i = 0;
  for (; i + 4 <= n; i += 4) {
    const double x0 = x[i], x1 = x[i + 1], x2 = x[i + 2], x3 = x[i + 3];
    const double y0 = y[i], y1 = y[i + 1], y2 = y[i + 2], y3 = y[i + 3];
    y[i] = y0 + alpha * x0;
    y[i + 1] = y1 + alpha * x1;
    y[i + 2] = y2 + alpha * x2;
    y[i + 3] = y3 + alpha * x3;
  }
  for (; i < n; ++i) y[i] += alpha * x[i];
}

// Cholesky of the leading n x n block of a (leading dimension ld), upper triangle:
// A = R'R with R stored in the upper triangle.  Returns 0, or k>0 if the leading minor
// of order k is not positive definite.
//
// With one lane per column (c.nl >= n) the factorisation runs as a wavefront: at step k lane
// k finishes the diagonal R_kk, then every lane j > k computes its entry R_kj.  Each entry is
// produced by the same operations in the same order as in the sequential loop nest (its
// dot product runs over i ascending, the column's sum of squares over k ascending), so the
// two forms give identical bits; the chain is n steps long instead of n^2/2.
//
// rd[k] receives 1/R_kk, formed as sqrt(d) * (1/d) from the pivot d = R_kk^2: the square root and
// the reciprocal do not depend on each other, so a step's chain is one of them plus two
// multiplies instead of a square root followed by a divide.  The entries R_kj are scaled by it.
LB_HD int dpofa(double *a, int ld, int n, double *rd, const Coop c = Coop{0, 1}, const int vm = 0) {
  if (c.nl >= n && c.nl > 1) {
    const int j = c.lane;
    double s = 0.0;
    for (int k = 0; k < n; ++k) {
      // every lane forms "its" diagonal; lane k's is the real one (v_readlane broadcast)
      const double dkk = lane_bcast(a[k * ld + k] - s, k);
      if (dkk <= 0.0) {  // not positive definite
        if (j == k) a[k * ld + k] = dkk;
        LB_LANES_SYNC_VM(vm);
        return k + 1;
      }
      const double rkk = sqrt(dkk);
      const double rinv = rkk * (1.0 / dkk);
      if (j == k) {
        a[k * ld + k] = rkk;
        rd[k] = rinv;
      }
      if (j > k && j < n) {
        double t = a[j * ld + k] - ddot(k, a + k * ld, a + j * ld);
        t = t * rinv;
        a[j * ld + k] = t;
        s += t * t;
      }
      LB_LANES_SYNC_VM(vm);
    }
    return 0;
  }
  for (int j = 0; j < n; ++j) {
    double s = 0.0;
    for (int k = 0; k < j; ++k) {
      double t = a[j * ld + k] - ddot(k, a + k * ld, a + j * ld);
      t = t * rd[k];
      a[j * ld + k] = t;
      s += t * t;
    }
    s = a[j * ld + j] - s;
    if (s <= 0.0) return j + 1;
    const double r = sqrt(s);
    a[j * ld + j] = r;
    rd[j] = r * (1.0 / s);
  }
  return 0;
}

// Triangular solves with the UPPER triangle of t (leading dimension ld):
// trans == 0:  T x = b;   trans != 0:  T' x = b.   Returns 0, or k>0 for a zero diagonal.
// c.nl >= n: lane i owns x_i.  As soon as an unknown is final it is broadcast and every lane
// still waiting folds it into its own running sum -- the same products, added in the same
// order, as the sequential substitution (identical bits), n steps instead of n^2/2.
// rd[j] = 1 / T_jj (stored by dpofa): every unknown is a product with it, not a quotient.
LB_HD int dtrsl_upper(const double *t, int ld, int n, double *b, int trans, const double *rd,
                      const Coop c = Coop{0, 1}) {
  if (c.nl >= n && c.nl > 1) {
    if (lanes_any(c.lane < n && t[(c.lane < n ? c.lane : 0) * (ld + 1)] == 0.0)) {
      for (int j = 0; j < n; ++j)
        if (t[j * ld + j] == 0.0) return j + 1;
    }
  } else {
    for (int j = 0; j < n; ++j)
      if (t[j * ld + j] == 0.0) return j + 1;
  }
  if (c.nl >= n && c.nl > 1) {
    // Every lane scales its own running value; the owner's result is broadcast with v_readlane
    // (no LDS round trip in the chain) and T's entries for the next step are requested before
    // the product of this one.
    const int me = c.lane;
    const bool mine = me < n;
    const int mc = mine ? me : 0;  // (lanes without an unknown read a valid address)
    double xme = 0.0;
    if (!trans) {  // T x = b, backward; the sequential form applies b[i] += (-x_j) T_ij, j descending
      double bi = mine ? b[me] : 0.0;
      double rjj = rd[n - 1], tji = t[(n - 1) * ld + mc];
      for (int j = n - 1; j >= 0; --j) {
        const int jn = j > 0 ? j - 1 : 0;
        const double rjj_n = rd[jn], tji_n = t[jn * ld + mc];
        const double xj = lane_bcast(bi * rjj, j);
        if (me == j) xme = xj;
        if (mine && me < j) bi = bi + (-xj) * tji;
        rjj = rjj_n;
        tji = tji_n;
      }
    } else {  // T' x = b, forward; the sequential form is x_j = (b_j - sum_{i<j} T_ij x_i) / T_jj
      const double bj = mine ? b[me] : 0.0;
      double acc = 0.0;
      double rii = rd[0], tmi = t[mc * ld];
      for (int i = 0; i < n; ++i) {
        const int in = i + 1 < n ? i + 1 : i;
        const double rii_n = rd[in], tmi_n = t[mc * ld + in];
        const double xi = lane_bcast(i == 0 ? bj * rii : (bj - acc) * rii, i);
        if (me == i) xme = xi;
        if (mine && me > i) acc += tmi * xi;
        rii = rii_n;
        tmi = tmi_n;
      }
    }
    LB_LANES_SYNC();  // (earlier reads of b by other lanes are complete)
    if (mine) b[me] = xme;
    LB_LANES_SYNC();
    return 0;
  }
  if (!trans) {
    b[n - 1] = b[n - 1] * rd[n - 1];
    for (int j = n - 2; j >= 0; --j) {
      daxpy(j + 1, -b[j + 1], t + (j + 1) * ld, b);
      b[j] = b[j] * rd[j];
    }
  } else {
    b[0] = b[0] * rd[0];
    for (int j = 1; j < n; ++j) {
      b[j] = b[j] - ddot(j, t + j * ld, b);
      b[j] = b[j] * rd[j];
    }
  }
  return 0;
}

// T' x = b for one right-hand side by the calling thread alone; the caller has checked the
// diagonal (same operations as dtrsl_upper(.., trans = 1) past its check).
LB_HD void dtrsl_lower_rhs(const double *t, int ld, int n, double *b, const double *rd) {
  b[0] = b[0] * rd[0];
  for (int j = 1; j < n; ++j) {
    b[j] = b[j] - ddot(j, t + j * ld, b);
    b[j] = b[j] * rd[j];
  }
}

// Three dot products against the same vector, each summed left to right; the operands of all
// three are fetched together (one LDS latency per four terms instead of three).
LB_HD void ddot3(int n, const double *a, const double *b, const double *c, const double *v,
                 double &av, double &bv, double &cv) {
  double sa = 0.0, sb = 0.0, sc = 0.0;
  int i = 0;
  for (; i + 2 <= n; i += 2) {
    const double a0 = a[i], a1 = a[i + 1], b0 = b[i], b1 = b[i + 1], c0 = c[i], c1 = c[i + 1];
    const double v0 = v[i], v1 = v[i + 1];
    sa += a0 * v0;
    sb += b0 * v0;
    sc += c0 * v0;
    sa += a1 * v1;
    sb += b1 * v1;
    sc += c1 * v1;
  }
  for (; i < n; ++i) {
    sa += a[i] * v[i];
    sb += b[i] * v[i];
    sc += c[i] * v[i];
  }
  av = sa;
  bv = sb;
  cv = sc;
}

// ---- limited-memory matrix products ---------------------------------------------
// Product of the 2col x 2col middle matrix of the compact L-BFGS formula with v -> p.
LB_HD int bmv(int m, const Work &w, int col, const double *v, double *p, const Coop c) {
  const double *sy = w.sy, *wt = w.wt;
  if (col == 0) return 0;
  // solve [  D^(1/2)      O ] [ p1 ] = [ v1 ]
  //       [ -L*D^(-1/2)   J ] [ p2 ]   [ v2 ]
  for (int i = c.lane; i < col; i += c.nl)
    p[col + i] = i == 0 ? v[col] : v[col + i] + dot_rcp(i, sy + i, m, v, 1, w.rsy, 1);
  LB_LANES_SYNC();
  int info = dtrsl_upper(wt, m, col, p + col, 1, w.rwt, c);
  if (info) return info;
  // solve [ -D^(1/2)   D^(-1/2)*L' ] [ p1 ] = [ p1 ]
  //       [  0         J'          ] [ p2 ]   [ p2 ]
  info = dtrsl_upper(wt, m, col, p + col, 0, w.rwt, c);
  if (info) return info;
  for (int i = c.lane; i < col; i += c.nl) {
    const double rs = w.rsq[i];
    double pi = v[i] * rs;
    pi = -pi * rs;
    p[i] = pi + dot_rcp(col - i - 1, sy + i * m + i + 1, 1, p + col + i + 1, 1, w.rsy + i, 0);
  }
  LB_LANES_SYNC();
  return 0;
}

// T = theta*SS + L*D^(-1)*L' (upper triangle), then its Cholesky factor J' in wt.
LB_HD int formt(int m, const Work &w, int col, double theta, const Coop c) {
  double *wt = w.wt;
  const double *sy = w.sy, *ss = w.ss;
  for (int e = c.lane; e < col * col; e += c.nl) {  // the entries (i, j >= i) are independent
    const int i = e / col, j = e - i * col;
    if (j < i) continue;
    if (i == 0) wt[j * m] = theta * ss[j * m];
    else wt[j * m + i] = dot_rcp(i, sy + i, m, sy + j, m, w.rsy, 1) + theta * ss[j * m + i];
  }
  LB_LANES_SYNC();
  return dpofa(wt, m, col, w.rwt, c) ? -3 : 0;
}

// ---- projected gradient norm -------------------------------------------------------
LB_HD double projgr(int n, const double *l, const double *u, const int *nbd, const double *x,
                    const double *g, const Coop c = Coop{0, 1}) {
  double sb = 0.0;
  if (c.nl > 1 && n <= c.nl) {  // lane i forms term i; the max runs over them in the same order
    double v = 0.0;
    if (c.lane < n) {
      const int i = c.lane;
      double gi = g[i];
      if (nbd[i] != 0) {
        if (gi < 0.0) {
          if (nbd[i] >= 2) gi = fmax(x[i] - u[i], gi);
        } else {
          if (nbd[i] <= 2) gi = fmin(x[i] - l[i], gi);
        }
      }
      v = fabs(gi);
    }
    for (int i = 0; i < n; ++i) sb = fmax(sb, lane_bcast(v, i));
    return sb;
  }
  for (int i = 0; i < n; ++i) {
    double gi = g[i];
    if (nbd[i] != 0) {
      if (gi < 0.0) {
        if (nbd[i] >= 2) gi = fmax(x[i] - u[i], gi);
      } else {
        if (nbd[i] <= 2) gi = fmin(x[i] - l[i], gi);
      }
    }
    sb = fmax(sb, fabs(gi));
  }
  return sb;
}

// (the heap of breakpoints of the published code -- hpsolb -- is gone: see cauchy's tie rule)

struct IterArgs {
  int n, m, col, head, nfree, nenter, ileave, updatd, iupdat;
  double theta, sbgnrm;
  Coop c;
};

// ---- generalized Cauchy point ----------------------------------------------------------
// xcp = w.z, breakpoints in w.t, search direction in w.d, iorder = w.indx2,
// p | c | wbp | v = w.wa.  Returns info (0 ok) in the low 8 bits and nseg above them.
template <bool VL = true>
LB_HDN int cauchy(const IterArgs s, const Work w, const double *l, const double *u,
                  const int *nbd) {
  const int n = s.n, m = s.m, col = s.col, col2 = 2 * s.col;
  int nseg = 0;
  double *x = w.x, *g = w.g, *t = w.t, *d = w.d, *xcp = w.z;
  double *p = w.wa, *c = w.wa + 2 * m, *wbp = w.wa + 4 * m, *v = w.wa + 6 * m;
  int *iorder = w.indx2, *iwhere = w.iwhere;
  const double theta = s.theta;

  if (s.sbgnrm <= 0.0) {
    if (VL) {
      for (int i = s.c.lane; i < n; i += s.c.nl) xcp[i] = x[i];
      LB_LANES_SYNC();
    } else {
      for (int i = 0; i < n; ++i) xcp[i] = x[i];
    }
    return 0;
  }
#define LB_CAUCHY_RET(info) (((info) & 0xff) | (nseg << 8))
  bool bnded = true;
  int nfree = n + 1, nbreak = 0, ibkmin = 0;
  double bkmin = 0.0, f1 = 0.0;
  const Coop cp = s.c;
  CK_MARK_DECL;
  for (int i = cp.lane; i < col2; i += cp.nl) p[i] = 0.0;  // lane j owns p[j] throughout

  if (VL && cp.nl > 1 && n <= cp.nl && m <= cp.nl) {
    // One variable per lane: the classification of variable i, its direction component and its
    // breakpoint depend on variable i alone.  What the sequential loop carries from one variable to
    // the next -- the sums f1 and p (accumulated in the order of i), the compacted lists of
    // breakpoints (front of iorder / t, in the order of i) and of free variables (back of iorder),
    // the first smallest breakpoint -- is rebuilt in that order from the lanes' values, so every
    // number is the one the loop below produces (this loop cost 2.4 k cycles per variable: 80 k of
    // a 100 k-cycle iteration at n = 32).
    const int i0 = cp.lane;
    double neggi = 0.0, tbrk = 0.0;
    int kind = 0;  // 0: stays (d = 0), 1: moves towards a bound (breakpoint), 2: moves freely
    bool unb = false;
    if (i0 < n) {
      neggi = -g[i0];
      const int nb = nbd[i0];
      int iw = iwhere[i0];
      double tl = 0.0, tu = 0.0;
      if (iw != 3 && iw != -1) {
        if (nb <= 2) tl = x[i0] - l[i0];
        if (nb >= 2) tu = u[i0] - x[i0];
        const bool xlower = nb <= 2 && tl <= 0.0;
        const bool xupper = nb >= 2 && tu <= 0.0;
        iw = 0;
        if (xlower) {
          if (neggi <= 0.0) iw = 1;
        } else if (xupper) {
          if (neggi >= 0.0) iw = 2;
        } else {
          if (fabs(neggi) <= 0.0) iw = -3;
        }
        iwhere[i0] = iw;
      }
      if (iw != 0 && iw != -1) {
        d[i0] = 0.0;
      } else {
        d[i0] = neggi;
        if (nb <= 2 && nb != 0 && neggi < 0.0) {
          kind = 1;
          tbrk = tl / (-neggi);
        } else if (nb >= 2 && neggi > 0.0) {
          kind = 1;
          tbrk = tu / neggi;
        } else {
          kind = 2;
          unb = fabs(neggi) > 0.0;
        }
      }
    }
    const unsigned long long mbrk = lanes_ballot(kind == 1), mfree = lanes_ballot(kind == 2);
    const unsigned long long mmove = mbrk | mfree;
    bnded = !lanes_any(unb);
    // f1, p: sequential sums over the moving variables; lane j keeps p[j], p[col + j] in registers
    double pa = 0.0, pb = 0.0;
    const int jrow = cp.lane < col ? wrap(s.head + cp.lane, m) * n : 0;
#if defined(__HIP_DEVICE_COMPILE__)
    // (round 5: the sums alone in the loop over the variables -- col == 0, the first iteration of every restart,
    // needs f1 only -- and the first smallest breakpoint by a wave minimum: this loop was 10 k of cauchy's 35 k
    // cycles per call at n = 32)
    if (col > 0) {
      for (int i = 0; i < n; ++i) {
        if (!((mmove >> i) & 1ull)) continue;  // (wave-uniform)
        const double ng = lane_bcast(neggi, i);
        f1 -= ng * ng;
        if (cp.lane < col) {
          pa += w.wy[jrow + i] * ng;
          pb += w.ws[jrow + i] * ng;
        }
      }
    } else {
      for (int i = 0; i < n; ++i) {
        if (!((mmove >> i) & 1ull)) continue;
        const double ng = lane_bcast(neggi, i);
        f1 -= ng * ng;
      }
    }
    if (mbrk) {  // the first smallest breakpoint, in list order = the lowest lane among the smallest
      const unsigned long long key = kind == 1 ? orderable_f64(tbrk) : ~0ull;
      const unsigned long long kmin = wave_min_key(key);
      const int lp = __builtin_ctzll(lanes_ballot(kind == 1 && key == kmin));
      bkmin = lane_bcast(tbrk, lp);
      ibkmin = bits_below(mbrk, lp) + 1;
    }
#else
    int k = 0;
    for (int i = 0; i < n; ++i) {
      if (!((mmove >> i) & 1ull)) continue;  // (wave-uniform)
      const double ng = lane_bcast(neggi, i);
      f1 -= ng * ng;
      if (cp.lane < col) {
        pa += w.wy[jrow + i] * ng;
        pb += w.ws[jrow + i] * ng;
      }
      if ((mbrk >> i) & 1ull) {  // the first smallest breakpoint, in list order
        ++k;
        const double tb = lane_bcast(tbrk, i);
        if (k == 1 || tb < bkmin) {
          bkmin = tb;
          ibkmin = k;
        }
      }
    }
#endif
    if (cp.lane < col) {
      p[cp.lane] = pa;
      p[col + cp.lane] = pb;
    }
    nbreak = bits_set(mbrk);
    nfree = n + 1 - bits_set(mfree);
    if (kind == 1) {
      const int pos = bits_below(mbrk, i0);
      iorder[pos] = i0 + 1;
      t[pos] = tbrk;
    } else if (kind == 2) {
      iorder[n - 1 - bits_below(mfree, i0)] = i0 + 1;
    }
    LB_LANES_SYNC();
  } else
  for (int i = 1; i <= n; ++i) {
    const double neggi = -g[i - 1];
    double tl = 0.0, tu = 0.0;
    if (iwhere[i - 1] != 3 && iwhere[i - 1] != -1) {
      if (nbd[i - 1] <= 2) tl = x[i - 1] - l[i - 1];
      if (nbd[i - 1] >= 2) tu = u[i - 1] - x[i - 1];
      const bool xlower = nbd[i - 1] <= 2 && tl <= 0.0;
      const bool xupper = nbd[i - 1] >= 2 && tu <= 0.0;
      iwhere[i - 1] = 0;
      if (xlower) {
        if (neggi <= 0.0) iwhere[i - 1] = 1;
      } else if (xupper) {
        if (neggi >= 0.0) iwhere[i - 1] = 2;
      } else {
        if (fabs(neggi) <= 0.0) iwhere[i - 1] = -3;
      }
    }
    int pointr = s.head;
    if (iwhere[i - 1] != 0 && iwhere[i - 1] != -1) {
      d[i - 1] = 0.0;
    } else {
      d[i - 1] = neggi;
      f1 -= neggi * neggi;
      for (int j = cp.lane; j < col; j += cp.nl) {
        const int pj = wrap(pointr + j, m);
        p[j] += w.wy[pj * n + (i - 1)] * neggi;
        p[col + j] += w.ws[pj * n + (i - 1)] * neggi;
      }
      if (nbd[i - 1] <= 2 && nbd[i - 1] != 0 && neggi < 0.0) {
        ++nbreak;
        iorder[nbreak - 1] = i;
        t[nbreak - 1] = tl / (-neggi);
        if (nbreak == 1 || t[nbreak - 1] < bkmin) {
          bkmin = t[nbreak - 1];
          ibkmin = nbreak;
        }
      } else if (nbd[i - 1] >= 2 && neggi > 0.0) {
        ++nbreak;
        iorder[nbreak - 1] = i;
        t[nbreak - 1] = tu / neggi;
        if (nbreak == 1 || t[nbreak - 1] < bkmin) {
          bkmin = t[nbreak - 1];
          ibkmin = nbreak;
        }
      } else {
        --nfree;
        iorder[nfree - 1] = i;
        if (fabs(neggi) > 0.0) bnded = false;
      }
    }
  }
  if (theta != 1.0)
    for (int j = cp.lane; j < col; j += cp.nl) p[col + j] *= theta;
  if (VL) {
    for (int i = cp.lane; i < n; i += cp.nl) xcp[i] = x[i];
    LB_LANES_SYNC();
  } else {
    for (int i = 0; i < n; ++i) xcp[i] = x[i];
  }
  CK_MARK(19);
  if (nbreak == 0 && nfree == n + 1) return LB_CAUCHY_RET(0);  // d is zero: GCP = x
  for (int j = cp.lane; j < col2; j += cp.nl) c[j] = 0.0;
  LB_LANES_SYNC();

  double f2 = -theta * f1;
  const double f2_org = f2;
  if (col > 0) {
    const int info = bmv(m, w, col, p, v, s.c);
    if (info) return LB_CAUCHY_RET(info);
    f2 -= ddot(col2, v, p);
  }
  double dtm = -f1 / f2;
  double tsum = 0.0;
  nseg = 1;
  bool skip_to_999 = false;
  CK_MARK(20);

  if (nbreak > 0) {
    int nleft = nbreak, iter = 1, ibp = 0;
    double tj = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
    // One variable per lane above: breakpoint k of the list sits in lane k's registers and the next one is the
    // smallest still there -- a wave minimum (ties: the lowest list position) instead of the heap in the workspace,
    // whose every step is a dependent LDS round trip made by all 64 lanes together (building it for the second
    // breakpoint and sifting: 4.2 k cycles per breakpoint at n = 32, 21 k of cauchy's 35 k per call).  The forms
    // without a variable per lane (host build, one problem per lane) scan for the same entry: see below.
    const bool in_lanes = VL && cp.nl > 1 && n <= cp.nl && m <= cp.nl;
    double t_l = 0.0;
    int o_l = 0;
    bool alive = false;
    if (in_lanes && cp.lane < nbreak) {
      t_l = t[cp.lane];
      o_l = iorder[cp.lane];
      alive = true;
    }
    const unsigned long long key_l = orderable_f64(t_l);
#endif
    for (;;) {
      const double tj0 = tj;
#if defined(__HIP_DEVICE_COMPILE__)
      if (in_lanes) {
        int lp;
        if (iter == 1) {
          tj = bkmin;
          lp = ibkmin - 1;
        } else {
          const unsigned long long kmin = wave_min_key(alive ? key_l : ~0ull);
          lp = __builtin_ctzll(lanes_ballot(alive && key_l == kmin));
          tj = lane_bcast(t_l, lp);
        }
        ibp = __builtin_amdgcn_readlane(o_l, lp);
        alive = alive && cp.lane != lp;
      } else
#endif
      {
        // ONE tie rule in every form: the next breakpoint is the smallest still in the list, and among
        // EQUAL ones the lowest list position -- what the wave minimum above takes.  The published code's heap hands
        // the breakpoints out in ascending order too, but where two are exactly equal (symmetric gradients and bounds)
        // its order is an accident of the heap's shape; a host build with the heap could then differ from the device
        // in the last bits.  A taken entry is marked with -1 (breakpoints are >= 0).
        int best = ibkmin - 1;
        if (iter > 1) {
          best = -1;
          double tb = 0.0;
          for (int k = 0; k < nbreak; ++k) {
            const double tk = t[k];
            if (tk != -1.0 && (best < 0 || tk < tb)) {  // (!= : a NaN entry still counts as present)
              best = k;
              tb = tk;
            }
          }
        }
        tj = iter == 1 ? bkmin : t[best];
        ibp = iorder[best];
        t[best] = -1.0;
      }
      CK_MARK(21);
      const double dt = tj - tj0;
      if (dtm < dt) break;  // the minimiser lies in this segment
      tsum += dt;
      --nleft;
      ++iter;
      const double dibp = d[ibp - 1];
      d[ibp - 1] = 0.0;
      double zibp;
      if (dibp > 0.0) {
        zibp = u[ibp - 1] - x[ibp - 1];
        xcp[ibp - 1] = u[ibp - 1];
        iwhere[ibp - 1] = 2;
      } else {
        zibp = l[ibp - 1] - x[ibp - 1];
        xcp[ibp - 1] = l[ibp - 1];
        iwhere[ibp - 1] = 1;
      }
      if (nleft == 0 && nbreak == n) {  // every variable is fixed
        dtm = dt;
        skip_to_999 = true;
        break;
      }
      ++nseg;
      const double dibp2 = dibp * dibp;
      f1 = f1 + dt * f2 + dibp2 - theta * dibp * zibp;
      f2 = f2 - theta * dibp2;
      CK_MARK(22);
      if (col > 0) {
        for (int j = cp.lane; j < col2; j += cp.nl) c[j] += dt * p[j];
        for (int j = cp.lane; j < col; j += cp.nl) {
          const int pj = wrap(s.head + j, m);
          wbp[j] = w.wy[pj * n + (ibp - 1)];
          wbp[col + j] = theta * w.ws[pj * n + (ibp - 1)];
        }
        LB_LANES_SYNC();
        const int info = bmv(m, w, col, wbp, v, s.c);
        if (info) return LB_CAUCHY_RET(info);
        double wmc, wmp, wmw;
        ddot3(col2, c, p, wbp, v, wmc, wmp, wmw);
        for (int j = cp.lane; j < col2; j += cp.nl) p[j] -= dibp * wbp[j];
        LB_LANES_SYNC();
        f1 += dibp * wmc;
        f2 += 2.0 * dibp * wmp - dibp2 * wmw;
      }
      f2 = fmax(LB_EPSMCH * f2_org, f2);
      CK_MARK(23);
      if (nleft > 0) {
        dtm = -f1 / f2;
        continue;
      } else if (bnded) {
        f1 = 0.0;
        f2 = 0.0;
        dtm = 0.0;
      } else {
        dtm = -f1 / f2;
      }
      break;
    }
  }
  if (!skip_to_999) {
    if (dtm <= 0.0) dtm = 0.0;
    tsum += dtm;
    if (VL) {
      for (int i = cp.lane; i < n; i += cp.nl) xcp[i] += tsum * d[i];
    } else {
      for (int i = 0; i < n; ++i) xcp[i] += tsum * d[i];
    }
  }
  if (col > 0)
    for (int j = cp.lane; j < col2; j += cp.nl) c[j] += dtm * p[j];
  LB_LANES_SYNC();
  CK_MARK(24);
  return LB_CAUCHY_RET(0);
#undef LB_CAUCHY_RET
}

// ---- free / active bookkeeping at the GCP ------------------------------------------------
template <bool VL = true>
LB_HD void freev(State &s, const Work &w, const Coop c = Coop{0, 1}) {
  const int n = s.n;
  s.nenter = 0;
  s.ileave = n + 1;
  if (VL && c.nl > 1 && n <= c.nl) {
    // one list position (first two loops) / one variable (third) per lane; the lists are compacted
    // in the order of the sequential loops
    if (s.iter > 0 && s.cnstnd) {
      const int pos = c.lane;  // 0-based position in the old index
      const int k = pos < n ? w.index[pos] : 1;
      const int iw = pos < n ? w.iwhere[k - 1] : 0;
      const unsigned long long mleave = lanes_ballot(pos < s.nfree && iw > 0);
      const unsigned long long menter = lanes_ballot(pos >= s.nfree && pos < n && iw <= 0);
      if ((mleave >> pos) & 1ull) w.indx2[n - 1 - bits_below(mleave, pos)] = k;
      if ((menter >> pos) & 1ull) w.indx2[bits_below(menter, pos)] = k;
      s.ileave = n + 1 - bits_set(mleave);
      s.nenter = bits_set(menter);
    }
    s.wrk = (s.ileave < n + 1) || (s.nenter > 0) || s.updatd;
    LB_LANES_SYNC();  // (the old index has been read by every lane before it is rewritten)
    const int i0 = c.lane;
    const bool isfree = i0 < n && w.iwhere[i0] <= 0;
    const unsigned long long mfree = lanes_ballot(isfree), mact = lanes_ballot(i0 < n && !isfree);
    if (isfree) w.index[bits_below(mfree, i0)] = i0 + 1;
    else if (i0 < n) w.index[n - 1 - bits_below(mact, i0)] = i0 + 1;
    s.nfree = bits_set(mfree);
    LB_LANES_SYNC();
    return;
  }
  if (s.iter > 0 && s.cnstnd) {
    for (int i = 1; i <= s.nfree; ++i) {
      const int k = w.index[i - 1];
      if (w.iwhere[k - 1] > 0) {
        --s.ileave;
        w.indx2[s.ileave - 1] = k;
      }
    }
    for (int i = 1 + s.nfree; i <= n; ++i) {
      const int k = w.index[i - 1];
      if (w.iwhere[k - 1] <= 0) {
        ++s.nenter;
        w.indx2[s.nenter - 1] = k;
      }
    }
  }
  s.wrk = (s.ileave < n + 1) || (s.nenter > 0) || s.updatd;
  s.nfree = 0;
  int iact = n + 1;
  for (int i = 1; i <= n; ++i) {
    if (w.iwhere[i - 1] <= 0) {
      ++s.nfree;
      w.index[s.nfree - 1] = i;
    } else {
      --iact;
      w.index[iact - 1] = i;
    }
  }
}

// ---- LEL^T factorisation of the reduced middle matrix (subspace minimisation) ---------------
LB_HDN int formk(const IterArgs s, const Work w) {
  const int n = s.n, m = s.m, col = s.col, nsub = s.nfree, m2 = 2 * s.m;
  const Coop c = s.c;
  double *wn = w.wn, *wn1 = w.snd;
  const int *ind = w.index, *indx2 = w.indx2;
#define WN(i, j) wn[(j) * m2 + (i)]
#define WN1(i, j) wn1[(j) * m2 + (i)]
#define WS(k, p) w.ws[(p) * n + (k)]
#define WY(k, p) w.wy[(p) * n + (k)]
  int upcl;
  FK_MARK_DECL;
  if (s.updatd) {
    if (s.iupdat > m) {  // shift the old part of WN1: every entry moves one up and one to the left
      if (c.nl > 1) {
        // Round 6.  The sequential form below is 2 (m - 1) + (m - 1)^2 + ... = 171 moves at m = 10, each a load that its
        // store waits for, done by all 64 lanes alike: ~15 k cycles per call once the memory is full (BASELINE config 2:
        // formk 36 k cycles per call against 24 k for config 3, whose restarts end before it fills).  The moves of the
        // three blocks touch disjoint entries, and within a block an entry's source lies one COLUMN to the right of
        // it: taken in column order, a source is overwritten by a LATER move only.  So the (jy, i) grid is dealt to
        // the lanes in that order, a pass loads its sources, waits, then stores -- and a source of this pass was the
        // destination of no earlier one.  Same values moved to the same places.
        const int m1 = m - 1;
        for (int e0 = 0; e0 < m1 * m1; e0 += c.nl) {
          const int e = e0 + c.lane;
          const bool on = e < m1 * m1;
          const int jy = on ? e / m1 : 0, i = on ? e - jy * m1 : 0, js = m + jy;
          const bool tri = on && i < m1 - jy;
          double a = 0.0, b = 0.0, cc = 0.0;
          if (tri) {
            a = WN1(jy + 1 + i, jy + 1);
            b = WN1(js + 1 + i, js + 1);
          }
          if (on) cc = WN1(m + 1 + i, jy + 1);
          LB_LANES_SYNC_VM(w.vm);
          if (tri) {
            WN1(jy + i, jy) = a;
            WN1(js + i, js) = b;
          }
          if (on) WN1(m + i, jy) = cc;
          LB_LANES_SYNC_VM(w.vm);
        }
      } else {
      for (int jy = 0; jy < m - 1; ++jy) {
        const int js = m + jy;
        for (int i = 0; i < m - 1 - jy; ++i) {
          WN1(jy + i, jy) = WN1(jy + 1 + i, jy + 1);
          WN1(js + i, js) = WN1(js + 1 + i, js + 1);
        }
        for (int i = 0; i < m - 1; ++i) WN1(m + i, jy) = WN1(m + 1 + i, jy + 1);
      }
      }
      LB_LANES_SYNC_VM(w.vm);
    }
    // new rows in blocks (1,1), (2,1), (2,2) and the new column in block (2,1): one entry
    // set per jy, independent of each other
    {
      const int ipntr = wrap(s.head + col - 1, m);
      const int iy = col - 1, is = m + col - 1;
      for (int jy = c.lane; jy < col; jy += c.nl) {
        const int js = m + jy;
        const int jpntr = wrap(s.head + jy, m);
        double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0;
        for (int k = 0; k < nsub; ++k) {
          const int k1 = ind[k] - 1;
          temp1 += WY(k1, ipntr) * WY(k1, jpntr);
        }
        for (int k = nsub; k < n; ++k) {
          const int k1 = ind[k] - 1;
          temp2 += WS(k1, ipntr) * WS(k1, jpntr);
          temp3 += WS(k1, ipntr) * WY(k1, jpntr);
        }
        WN1(iy, jy) = temp1;
        WN1(is, js) = temp2;
        WN1(is, jy) = temp3;
      }
      LB_LANES_SYNC_VM(w.vm);
      const int jyc = col - 1;
      const int jpntr = wrap(s.head + col - 1, m);
      for (int i = c.lane; i < col; i += c.nl) {
        const int is2 = m + i;
        const int ip = wrap(s.head + i, m);
        double temp3 = 0.0;
        for (int k = 0; k < nsub; ++k) {
          const int k1 = ind[k] - 1;
          temp3 += WS(k1, ip) * WY(k1, jpntr);
        }
        WN1(is2, jyc) = temp3;
      }
      LB_LANES_SYNC_VM(w.vm);
    }
    upcl = col - 1;
  } else {
    upcl = col;
  }
  FK_MARK(19);
  // old parts of blocks (1,1), (2,2) and (2,1): variables that entered / left the free set.
  // Each (iy, jy) entry is independent.
  for (int e = c.lane; e < upcl * upcl; e += c.nl) {
    const int iy = e / upcl, jy = e - iy * upcl;
    const int ipntr = wrap(s.head + iy, m), jpntr = wrap(s.head + jy, m);
    if (jy <= iy) {
      const int is = m + iy, js = m + jy;
      double temp1 = 0.0, temp2 = 0.0, temp3 = 0.0, temp4 = 0.0;
      for (int k = 0; k < s.nenter; ++k) {
        const int k1 = indx2[k] - 1;
        temp1 += WY(k1, ipntr) * WY(k1, jpntr);
        temp2 += WS(k1, ipntr) * WS(k1, jpntr);
      }
      for (int k = s.ileave - 1; k < n; ++k) {
        const int k1 = indx2[k] - 1;
        temp3 += WY(k1, ipntr) * WY(k1, jpntr);
        temp4 += WS(k1, ipntr) * WS(k1, jpntr);
      }
      WN1(iy, jy) += temp1 - temp3;
      WN1(is, js) += -temp2 + temp4;
    }
    {  // block (2,1), entry (m + iy, jy)
      const int is = m + iy;
      double temp1 = 0.0, temp3 = 0.0;
      for (int k = 0; k < s.nenter; ++k) {
        const int k1 = indx2[k] - 1;
        temp1 += WS(k1, ipntr) * WY(k1, jpntr);
      }
      for (int k = s.ileave - 1; k < n; ++k) {
        const int k1 = indx2[k] - 1;
        temp3 += WS(k1, ipntr) * WY(k1, jpntr);
      }
      if (is <= jy + m)
        WN1(is, jy) += temp1 - temp3;
      else
        WN1(is, jy) += -temp1 + temp3;
    }
  }
  LB_LANES_SYNC_VM(w.vm);
  FK_MARK(20);
  // upper triangle of WN = [D+Y'ZZ'Y/theta   -L_a'+R_z'] [-L_a+R_z   S'AA'S*theta]
  // (each iy writes its own columns iy and col+iy)
  const double theta = s.theta, rtheta = 1.0 / s.theta;
  if (c.nl > 1) {
    // (round 6: one ENTRY per lane instead of one column -- a column was up to 3 col loads each followed by its
    // store, one after the other; every entry is its own product of one stored value)
    for (int e = c.lane; e < col * col; e += c.nl) {
      const int iy = e / col, jy = e - iy * col;
      const int is = col + iy, is1 = m + iy, js = col + jy, js1 = m + jy;
      const double v21 = WN1(is1, jy);
      if (jy <= iy) {
        const double v11 = WN1(iy, jy), v22 = WN1(is1, js1);
        double t11 = v11 * rtheta;
        if (jy == iy) t11 += w.sy[iy * m + iy];
        WN(jy, iy) = t11;
        WN(js, is) = v22 * theta;
      }
      WN(jy, is) = jy < iy ? -v21 : v21;
    }
  } else {
  for (int iy = c.lane; iy < col; iy += c.nl) {
    const int is = col + iy, is1 = m + iy;
    for (int jy = 0; jy <= iy; ++jy) {
      const int js = col + jy, js1 = m + jy;
      WN(jy, iy) = WN1(iy, jy) * rtheta;
      WN(js, is) = WN1(is1, js1) * theta;
    }
    for (int jy = 0; jy < iy; ++jy) WN(jy, is) = -WN1(is1, jy);
    for (int jy = iy; jy < col; ++jy) WN(jy, is) = WN1(is1, jy);
    WN(iy, iy) += w.sy[iy * m + iy];
  }
  }
  LB_LANES_SYNC_VM(w.vm);
  FK_MARK(21);
  // Cholesky of the (1,1) block, then L^-1(-L_a'+R_z') in the (1,2) block (one right-hand
  // side per lane)
  if (dpofa(wn, m2, col, w.rwn, c, w.vm)) return -1;
  FK_MARK(22);
  const int col2 = 2 * col;
  if (c.nl > 1 && c.nl >= col) {
    if (lanes_any(c.lane < col && wn[(c.lane < col ? c.lane : 0) * (m2 + 1)] == 0.0)) return -1;
  } else {
    for (int j = 0; j < col; ++j)
      if (wn[j * m2 + j] == 0.0) return -1;
  }
  for (int js = col + c.lane; js < col2; js += c.nl)
    dtrsl_lower_rhs(wn, m2, col, wn + js * m2, w.rwn);
  LB_LANES_SYNC_VM(w.vm);
  FK_MARK(23);
  // (2,2) block: S'AA'S*theta + (L^-1(-L_a'+R_z'))'(L^-1(-L_a'+R_z')), then its Cholesky
  for (int e = c.lane; e < col * col; e += c.nl) {
    const int is = col + e / col, js = col + e % col;
    if (js >= is) WN(is, js) += ddot(col, wn + is * m2, wn + js * m2);
  }
  LB_LANES_SYNC_VM(w.vm);
  FK_MARK(24);
  if (dpofa(wn + col * m2 + col, m2, col, w.rwn + col, c, w.vm)) return -2;
  FK_MARK(25);
  return 0;
#undef WN
#undef WN1
#undef WS
#undef WY
}

// r = -Z'B(xcp - x) - Z'g   (uses c = wa[2m..4m) from cauchy; p = wa[0..2m) as scratch)
LB_HD int cmprlb(State &s, const Work &w, const Coop c) {
  const int n = LB_UNI(s.n, c), m = LB_UNI(s.m, c), col = LB_UNI(s.col, c);
  const int nfree = LB_UNI(s.nfree, c), head = LB_UNI(s.head, c);
  if (!s.cnstnd && col > 0) {
    for (int i = c.lane; i < n; i += c.nl) w.r[i] = -w.g[i];
    LB_LANES_SYNC();
    return 0;
  }
  if (bmv(m, w, col, w.wa + 2 * m, w.wa, c)) return -8;
  for (int i = c.lane; i < nfree; i += c.nl) {  // each r[i]: its terms in the order j = 0, 1, ...
    const int k = w.index[i] - 1;
    double ri = -s.theta * (w.z[k] - w.x[k]) - w.g[k];
    int p0 = head;
    int j = 0;
    for (; j + 2 <= col; j += 2) {  // operands of two terms in flight
      const int p1 = p0 + 1 == m ? 0 : p0 + 1;
      const double a10 = w.wa[j], a11 = w.wa[j + 1];
      const double b20 = w.wa[col + j], b21 = w.wa[col + j + 1];
      const double y0 = w.wy[p0 * n + k], s0 = w.ws[p0 * n + k];
      const double y1 = w.wy[p1 * n + k], s1 = w.ws[p1 * n + k];
      ri += y0 * a10 + s0 * (s.theta * b20);
      ri += y1 * a11 + s1 * (s.theta * b21);
      p0 = p1 + 1 == m ? 0 : p1 + 1;
    }
    if (j < col) ri += w.wy[p0 * n + k] * w.wa[j] + w.ws[p0 * n + k] * (s.theta * w.wa[col + j]);
    w.r[i] = ri;
  }
  LB_LANES_SYNC();
  return 0;
}

// Subspace minimisation: on entry w.z = xcp and w.r = reduced gradient; on exit w.z is
// the (projected) subspace minimiser.  wv = wa[0..2m).
// Returns info (0 ok) in the low 8 bits and iword (1: the step hit a bound) in bit 8.
template <bool VL = true>
LB_HDN int subsm(const IterArgs s, const Work w, const double *l, const double *u,
                 const int *nbd) {
  const int n = s.n, m = s.m, col = s.col, nsub = s.nfree, m2 = 2 * s.m, col2 = 2 * s.col;
  int iword = 0;
  double *x = w.z, *d = w.r, *xp = w.xp, *wv = w.wa;
  const double *xx = w.x, *gg = w.g;
  const int *ind = w.index;
  const double theta = s.theta;
  if (nsub <= 0) return 0;
  const Coop c = s.c;
  for (int i = c.lane; i < col; i += c.nl) {  // wv = W'Zd, one entry pair per i
    const int pointr = wrap(s.head + i, m);
    double temp1 = 0.0, temp2 = 0.0;
    for (int j = 0; j < nsub; ++j) {
      const int k = ind[j] - 1;
      temp1 += w.wy[pointr * n + k] * d[j];
      temp2 += w.ws[pointr * n + k] * d[j];
    }
    wv[i] = temp1;
    wv[col + i] = theta * temp2;
  }
  LB_LANES_SYNC();
  if (dtrsl_upper(w.wn, m2, col2, wv, 1, w.rwn, c)) return 1;
  for (int i = c.lane; i < col; i += c.nl) wv[i] = -wv[i];
  LB_LANES_SYNC();
  if (dtrsl_upper(w.wn, m2, col2, wv, 0, w.rwn, c)) return 1;
  const double rtheta = 1.0 / theta;
  for (int i = c.lane; i < n; i += c.nl) xp[i] = x[i];  // (the free variables are saved below too)
  for (int i = c.lane; i < nsub; i += c.nl) {  // d = (1/theta)d + (1/theta^2)Z'W wv, per entry
    const int k = ind[i] - 1;
    double di = d[i];
    int p0 = s.head;
    int jy = 0;
    for (; jy + 2 <= col; jy += 2) {  // operands (and the two divides) of two terms in flight
      const int p1 = p0 + 1 == m ? 0 : p0 + 1;
      const double v0 = wv[jy], v1 = wv[jy + 1], u0 = wv[col + jy], u1 = wv[col + jy + 1];
      const double y0 = w.wy[p0 * n + k], s0 = w.ws[p0 * n + k];
      const double y1 = w.wy[p1 * n + k], s1 = w.ws[p1 * n + k];
      di += y0 * v0 * rtheta + s0 * u0;
      di += y1 * v1 * rtheta + s1 * u1;
      p0 = p1 + 1 == m ? 0 : p1 + 1;
    }
    if (jy < col) di += w.wy[p0 * n + k] * wv[jy] * rtheta + w.ws[p0 * n + k] * wv[col + jy];
    di = di * rtheta;
    d[i] = di;
    // projected Newton step of this free variable (distinct k per i)
    const double xk = x[k];
    const int nb = nbd[k];
    double xn = xk + di;
    bool hit = false;
    if (nb == 1) {
      xn = fmax(l[k], xn);
      hit = xn == l[k];
    } else if (nb == 2) {
      xn = fmin(u[k], fmax(l[k], xn));
      hit = xn == l[k] || xn == u[k];
    } else if (nb == 3) {
      xn = fmin(u[k], xn);
      hit = xn == u[k];
    }
    x[k] = xn;
    if (hit) iword = 1;
  }
  LB_LANES_SYNC();
  if (c.nl > 1) iword = lanes_any(iword != 0) ? 1 : 0;
  if (iword == 0) return 0;
  // sign of the directional derivative along the projected step
  double dd_p = 0.0;
  if (VL && c.nl > 1 && n <= c.nl) {
    dd_p = lanes_sum_ordered(c.lane < n ? (x[c.lane] - xx[c.lane]) * gg[c.lane] : 0.0, n);
  } else {
    for (int i = 0; i < n; ++i) dd_p += (x[i] - xx[i]) * gg[i];
  }
  if (dd_p > 0.0) {  // not a descent direction: fall back to the backtracking step
    for (int i = 0; i < n; ++i) x[i] = xp[i];
    double alpha = 1.0, temp1 = alpha;
    int ibd = 0;
    for (int i = 0; i < nsub; ++i) {
      const int k = ind[i] - 1;
      const double dk = d[i];
      if (nbd[k] != 0) {
        if (dk < 0.0 && nbd[k] <= 2) {
          const double temp2 = l[k] - x[k];
          if (temp2 >= 0.0) temp1 = 0.0;
          else if (dk * alpha < temp2) temp1 = temp2 / dk;
        } else if (dk > 0.0 && nbd[k] >= 2) {
          const double temp2 = u[k] - x[k];
          if (temp2 <= 0.0) temp1 = 0.0;
          else if (dk * alpha > temp2) temp1 = temp2 / dk;
        }
        if (temp1 < alpha) {
          alpha = temp1;
          ibd = i;
        }
      }
    }
    if (alpha < 1.0) {
      const double dk = d[ibd];
      const int k = ind[ibd] - 1;
      if (dk > 0.0) {
        x[k] = u[k];
        d[ibd] = 0.0;
      } else if (dk < 0.0) {
        x[k] = l[k];
        d[ibd] = 0.0;
      }
    }
    for (int i = 0; i < nsub; ++i) {
      const int k = ind[i] - 1;
      x[k] += alpha * d[i];
    }
  }
  return iword << 8;
}

// ---- More'-Thuente line search -------------------------------------------------------------
// (Tried and dropped, round 3: the quotients of dcstep written out as LLVM's fp64 division sequence for
// three or four operands in lockstep -- same bits, the instructions interleaved in the ISA as intended,
// restart phase 331.0 -> 329.8 us: the chain is not what bounds it.  profiles/r3/ab_headline.txt,
// lbfgsb_formk_dcsrch_stages.txt: dcstep ~1.0 k of dcsrch's ~1.9 k cycles per call.)
LB_HD void dcstep(double &stx, double &fx, double &dx, double &sty, double &fy, double &dy,
                  double &stp, double fp, double dp, int &brackt, double stpmin, double stpmax) {
  const double p66 = 0.66;
  const double sgnd = dp * (dx / fabs(dx));
  double stpf;
  if (fp > fx) {  // higher function value: bracketed
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double sc = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dx / sc) * (dp / sc));
    if (stp < stx) gamma = -gamma;
    const double p = (gamma - dx) + theta, q = ((gamma - dx) + gamma) + dp, r = p / q;
    const double stpc = stx + r * (stp - stx);
    const double stpq = stx + ((dx / ((fx - fp) / (stp - stx) + dx)) / 2.0) * (stp - stx);
    if (fabs(stpc - stx) < fabs(stpq - stx)) stpf = stpc;
    else stpf = stpc + (stpq - stpc) / 2.0;
    brackt = 1;
  } else if (sgnd < 0.0) {  // lower value, derivatives of opposite sign: bracketed
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double sc = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dx / sc) * (dp / sc));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dx, r = p / q;
    const double stpc = stp + r * (stx - stp);
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
    else stpf = stpq;
    brackt = 1;
  } else if (fabs(dp) < fabs(dx)) {  // lower value, same sign, derivative shrinks
    const double theta = 3.0 * (fx - fp) / (stp - stx) + dx + dp;
    const double sc = fmax(fabs(theta), fmax(fabs(dx), fabs(dp)));
    double gamma = sc * sqrt(fmax(0.0, (theta / sc) * (theta / sc) - (dx / sc) * (dp / sc)));
    if (stp > stx) gamma = -gamma;
    const double p = (gamma - dp) + theta, q = (gamma + (dx - dp)) + gamma, r = p / q;
    double stpc;
    if (r < 0.0 && gamma != 0.0) stpc = stp + r * (stx - stp);
    else if (stp > stx) stpc = stpmax;
    else stpc = stpmin;
    const double stpq = stp + (dp / (dp - dx)) * (stx - stp);
    if (brackt) {
      if (fabs(stpc - stp) < fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      if (stp > stx) stpf = fmin(stp + p66 * (sty - stp), stpf);
      else stpf = fmax(stp + p66 * (sty - stp), stpf);
    } else {
      if (fabs(stpc - stp) > fabs(stpq - stp)) stpf = stpc;
      else stpf = stpq;
      stpf = fmin(stpmax, stpf);
      stpf = fmax(stpmin, stpf);
    }
  } else {  // lower value, same sign, derivative does not shrink
    if (brackt) {
      const double theta = 3.0 * (fp - fy) / (sty - stp) + dy + dp;
      const double sc = fmax(fabs(theta), fmax(fabs(dy), fabs(dp)));
      double gamma = sc * sqrt((theta / sc) * (theta / sc) - (dy / sc) * (dp / sc));
      if (stp > sty) gamma = -gamma;
      const double p = (gamma - dp) + theta, q = ((gamma - dp) + gamma) + dy, r = p / q;
      stpf = stp + r * (sty - stp);
    } else if (stp > stx) {
      stpf = stpmax;
    } else {
      stpf = stpmin;
    }
  }
  // (value selects, not conditional stores: the device compiler turned the latter into stores
  // through a selected ADDRESS, which forced all six variables into scratch memory)
  const bool higher = fp > fx, opposite = !higher && sgnd < 0.0;
  const double nsty = higher ? stp : (opposite ? stx : sty);
  const double nfy = higher ? fp : (opposite ? fx : fy);
  const double ndy = higher ? dp : (opposite ? dx : dy);
  const double nstx = higher ? stx : stp, nfx = higher ? fx : fp, ndx = higher ? dx : dp;
  sty = nsty; fy = nfy; dy = ndy;
  stx = nstx; fx = nfx; dx = ndx;
  stp = stpf;
}

enum { LS_START = 0, LS_FG = 1, LS_CONV = 2, LS_WARN = 3, LS_ERROR = 4 };

LB_HD void dcsrch(State &s, double f, double g, double &stp, double ftol, double gtol,
                  double xtol, double stpmin, double stpmax) {
  const double xtrapl = 1.1, xtrapu = 4.0, p5 = 0.5, p66 = 0.66;
  FK_MARK_DECL;
  if (s.ls_task == LS_START) {
    if (stp < stpmin || stp > stpmax || g >= 0.0 || ftol < 0.0 || gtol < 0.0 || xtol < 0.0 ||
        stpmin < 0.0 || stpmax < stpmin) {
      s.ls_task = LS_ERROR;
      return;
    }
    s.brackt = 0;
    s.ls_stage = 1;
    s.finit = f;
    s.ginit = g;
    s.gtest = ftol * s.ginit;
    s.width = stpmax - stpmin;
    s.width1 = s.width / p5;
    s.stx = 0.0; s.fx = s.finit; s.gx = s.ginit;
    s.sty = 0.0; s.fy = s.finit; s.gy = s.ginit;
    s.stmin = 0.0;
    s.stmax = stp + xtrapu * stp;
    s.ls_task = LS_FG;
    return;
  }
  const double ftest = s.finit + stp * s.gtest;
  if (s.ls_stage == 1 && f <= ftest && g >= 0.0) s.ls_stage = 2;
  int task = LS_FG;
  if (s.brackt && (stp <= s.stmin || stp >= s.stmax)) task = LS_WARN;
  if (s.brackt && s.stmax - s.stmin <= xtol * s.stmax) task = LS_WARN;
  if (stp == stpmax && f <= ftest && g <= s.gtest) task = LS_WARN;
  if (stp == stpmin && (f > ftest || g >= s.gtest)) task = LS_WARN;
  if (f <= ftest && fabs(g) <= gtol * (-s.ginit)) task = LS_CONV;
  if (task != LS_FG) {
    s.ls_task = task;
    return;
  }
  {
    // ONE set of locals goes through dcstep whichever branch is taken: handing it either the
    // state's fields or modified copies by reference made the device compiler keep all eight in
    // scratch memory (a dozen dependent ~500-cycle round trips per evaluation).
    const bool modified = s.ls_stage == 1 && f <= s.fx && f > ftest;
    double stx = s.stx, sty = s.sty, fx = s.fx, fy = s.fy, gx = s.gx, gy = s.gy, fp = f, gp = g;
    int brackt = s.brackt;
    if (modified) {
      fp = f - stp * s.gtest;
      fx = s.fx - s.stx * s.gtest;
      fy = s.fy - s.sty * s.gtest;
      gp = g - s.gtest;
      gx = s.gx - s.gtest;
      gy = s.gy - s.gtest;
    }
    FK_MARK(26);
    dcstep(stx, fx, gx, sty, fy, gy, stp, fp, gp, brackt, s.stmin, s.stmax);
    FK_MARK(27);
    if (modified) {
      fx = fx + stx * s.gtest;
      fy = fy + sty * s.gtest;
      gx = gx + s.gtest;
      gy = gy + s.gtest;
    }
    s.stx = stx; s.sty = sty; s.fx = fx; s.fy = fy; s.gx = gx; s.gy = gy;
    s.brackt = brackt;
  }
  if (s.brackt) {
    if (fabs(s.sty - s.stx) >= p66 * s.width1) stp = s.stx + p5 * (s.sty - s.stx);
    s.width1 = s.width;
    s.width = fabs(s.sty - s.stx);
  }
  if (s.brackt) {
    s.stmin = fmin(s.stx, s.sty);
    s.stmax = fmax(s.stx, s.sty);
  } else {
    s.stmin = stp + xtrapl * (stp - s.stx);
    s.stmax = stp + xtrapu * (stp - s.stx);
  }
  stp = fmax(stp, stpmin);
  stp = fmin(stp, stpmax);
  if ((s.brackt && (stp <= s.stmin || stp >= s.stmax)) ||
      (s.brackt && s.stmax - s.stmin <= xtol * s.stmax))
    stp = s.stx;
  s.ls_task = LS_FG;
}

// One call of the line-search driver.  first != 0 starts a new search along w.d.
// Returns 1 if f/g is needed at the new w.x, 0 if the search ended (s.info tells how).
// (the element-wise loops over the n variables are dealt to the lanes of a cooperating wave;
// the dot products stay sequential -- their summation order is part of the result)
template <bool VL = true>
LB_HD int lnsrlb(State &s, const Work &w, const double *l, const double *u, const int *nbd,
                 int first, const Coop c = Coop{0, 1}) {
  const int n = s.n;
  const double big = 1e10, ftol = 1e-3, gtol = 0.9, xtol = 0.1;
  if (first) {
    s.dtd = ddot_c<VL>(n, w.d, w.d, c);
    s.dnorm = sqrt(s.dtd);
    s.stpmx = big;
    if (s.cnstnd) {
      if (s.iter == 0) {
        s.stpmx = 1.0;
      } else if (VL && c.nl > 1 && n <= c.nl) {
        // lane i fetches variable i's operands; the scan itself (each step uses the stpmx so far)
        // runs over the lanes in the order of i
        double a1v = 0.0, a2lo = 0.0, a2hi = 0.0;
        int nb = 0;
        if (c.lane < n) {
          a1v = w.d[c.lane];
          nb = nbd[c.lane];
          a2lo = l[c.lane] - w.x[c.lane];
          a2hi = u[c.lane] - w.x[c.lane];
        }
        const unsigned long long mlo = lanes_ballot(nb != 0 && a1v < 0.0 && nb <= 2);
        const unsigned long long mhi = lanes_ballot(nb != 0 && !(a1v < 0.0 && nb <= 2) && a1v > 0.0 && nb >= 2);
        for (int i = 0; i < n; ++i) {
          if ((mlo >> i) & 1ull) {
            const double a1 = lane_bcast(a1v, i), a2 = lane_bcast(a2lo, i);
            if (a2 >= 0.0) s.stpmx = 0.0;
            else if (a1 * s.stpmx < a2) s.stpmx = a2 / a1;
          } else if ((mhi >> i) & 1ull) {
            const double a1 = lane_bcast(a1v, i), a2 = lane_bcast(a2hi, i);
            if (a2 <= 0.0) s.stpmx = 0.0;
            else if (a1 * s.stpmx > a2) s.stpmx = a2 / a1;
          }
        }
      } else {
        for (int i = 0; i < n; ++i) {
          const double a1 = w.d[i];
          if (nbd[i] != 0) {
            if (a1 < 0.0 && nbd[i] <= 2) {
              const double a2 = l[i] - w.x[i];
              if (a2 >= 0.0) s.stpmx = 0.0;
              else if (a1 * s.stpmx < a2) s.stpmx = a2 / a1;
            } else if (a1 > 0.0 && nbd[i] >= 2) {
              const double a2 = u[i] - w.x[i];
              if (a2 <= 0.0) s.stpmx = 0.0;
              else if (a1 * s.stpmx > a2) s.stpmx = a2 / a1;
            }
          }
        }
      }
    }
    if (s.iter == 0 && !s.boxed) s.stp = fmin(1.0 / s.dnorm, s.stpmx);
    else s.stp = 1.0;
    for (int i = c.lane; i < n; i += c.nl) {
      w.t[i] = w.x[i];
      w.r[i] = w.g[i];
    }
    LB_LANES_SYNC();
    s.fold = s.f;
    s.ifun = 0;
    s.iback = 0;
    s.ls_task = LS_START;
  }
  s.gd = ddot_c<VL>(n, w.g, w.d, c);
  if (s.ifun == 0) {
    s.gdold = s.gd;
    if (s.gd >= 0.0) {  // ascent direction in projection: line search impossible
      s.info = -4;
      return 0;
    }
  }
  dcsrch(s, s.f, s.gd, s.stp, ftol, gtol, xtol, 0.0, s.stpmx);
  s.xstep = s.stp * s.dnorm;
  if (s.ls_task != LS_CONV && s.ls_task != LS_WARN) {
    // (LS_ERROR cannot arise: stp in [0, stpmx], gd < 0, constants valid; treated as FG by the
    // original as well, whose caller then trips on info/iback)
    ++s.ifun;
    ++s.nfgv;
    s.iback = s.ifun - 1;
    if (s.stp == 1.0) {
      for (int i = c.lane; i < n; i += c.nl) w.x[i] = w.z[i];
    } else {
      for (int i = c.lane; i < n; i += c.nl) w.x[i] = s.stp * w.d[i] + w.t[i];
    }
    LB_LANES_SYNC();
    return 1;
  }
  return 0;
}

// ---- BFGS matrix update -----------------------------------------------------------------------
template <bool VL = true>
LB_HD void matupd(State &s, const Work &w, double rr, double dr, const Coop c = Coop{0, 1}) {
  const int n = LB_UNI(s.n, c), m = LB_UNI(s.m, c);
  s.iupdat = LB_UNI(s.iupdat, c);
  s.head = LB_UNI(s.head, c);
  s.itail = LB_UNI(s.itail, c);
  s.col = LB_UNI(s.col, c);
  if (s.iupdat <= m) {
    s.col = s.iupdat;
    s.itail = wrap(s.head + s.iupdat - 1, m);
  } else {
    s.itail = wrap(s.itail + 1, m);
    s.head = wrap(s.head + 1, m);
  }
  if (VL) {
    for (int i = c.lane; i < n; i += c.nl) {
      w.ws[s.itail * n + i] = w.d[i];
      w.wy[s.itail * n + i] = w.r[i];
    }
    LB_LANES_SYNC();
  } else {
    for (int i = 0; i < n; ++i) {
      w.ws[s.itail * n + i] = w.d[i];
      w.wy[s.itail * n + i] = w.r[i];
    }
  }
  s.theta = rr / dr;
  const int col = s.col;
  if (VL && c.nl > 1 && s.iupdat > m) {
    // (round 6: the same moves dealt to the lanes -- see formk's shift: sources lie one column to the right of their
    // destinations, the (j, i) grid goes out in column order, a pass loads, waits, stores.  108 dependent load /
    // store pairs by every lane alike were ~5 k of matupd's 7 k cycles once the memory is full)
    const int c1 = col - 1;
    for (int e0 = 0; e0 < c1 * c1; e0 += c.nl) {
      const int e = e0 + c.lane;
      const bool on = e < c1 * c1;
      const int j = on ? e / c1 : 0, i = on ? e - j * c1 : 0;
      const bool up = on && i <= j, lo = on && i < c1 - j, first = on && i == 0;
      double a = 0.0, b = 0.0, r1 = 0.0, r2 = 0.0;
      if (up) a = w.ss[(j + 1) * m + (i + 1)];
      if (lo) b = w.sy[(j + 1) * m + (j + 1 + i)];
      if (first) { r1 = w.rsy[j + 1]; r2 = w.rsq[j + 1]; }
      LB_LANES_SYNC();
      if (up) w.ss[j * m + i] = a;
      if (lo) w.sy[j * m + (j + i)] = b;
      if (first) { w.rsy[j] = r1; w.rsq[j] = r2; }
      LB_LANES_SYNC();
    }
  } else if (s.iupdat > m) {  // move old information
    for (int j = 0; j < col - 1; ++j) {
      w.rsy[j] = w.rsy[j + 1];
      w.rsq[j] = w.rsq[j + 1];
      for (int i = 0; i <= j; ++i) w.ss[j * m + i] = w.ss[(j + 1) * m + (i + 1)];
      for (int i = 0; i < col - 1 - j; ++i) w.sy[j * m + (j + i)] = w.sy[(j + 1) * m + (j + 1 + i)];
    }
  }
  if (VL) {
    LB_LANES_SYNC();
    // the new column of SY and the new row of SS: 2 (col - 1) independent inner products, one per lane
    for (int e = c.lane; e < 2 * (col - 1); e += c.nl) {
      const int j = e < col - 1 ? e : e - (col - 1);
      const int pointr = wrap(s.head + j, m);
      if (e < col - 1) w.sy[j * m + (col - 1)] = ddot(n, w.d, w.wy + pointr * n);
      else w.ss[(col - 1) * m + j] = ddot(n, w.ws + pointr * n, w.d);
    }
  } else {
    int pointr = s.head;
    for (int j = 0; j < col - 1; ++j) {
      w.sy[j * m + (col - 1)] = ddot(n, w.d, w.wy + pointr * n);
      w.ss[(col - 1) * m + j] = ddot(n, w.ws + pointr * n, w.d);
      pointr = wrap(pointr + 1, m);
    }
  }
  if (s.stp == 1.0) w.ss[(col - 1) * m + (col - 1)] = s.dtd;
  else w.ss[(col - 1) * m + (col - 1)] = s.stp * s.stp * s.dtd;
  w.sy[(col - 1) * m + (col - 1)] = dr;
  w.rsy[col - 1] = 1.0 / dr;
  w.rsq[col - 1] = 1.0 / sqrt(dr);
  if (VL) LB_LANES_SYNC();
}

LB_HD void refresh_memory(State &s) {
  s.info = 0;
  s.col = 0;
  s.head = 0;
  s.theta = 1.0;
  s.iupdat = 0;
  s.updatd = 0;
}

LB_HD void finish(State &s, int task, int msg) {
  s.task = task;
  s.msg = msg;
  s.stage = S_FINISHED;
  // scipy's warnflag: 0 converged, 1 maxfun/maxiter, 2 anything else
  if (task == T_CONVERGENCE) s.status = 0;
  else if (msg == M_MAXFUN || msg == M_MAXITER) s.status = 1;
  else s.status = 2;
}

// Initialise a problem.  x0 is clipped into the box (scipy does that before the solver
// sees it).  nbd[i]: 0 unbounded, 1 lower, 2 both, 3 upper.
// (c: the lanes of a wave that runs ONE problem share the loop over the variables -- round 6: done by every lane alike it
// was 32 dependent LDS round trips per problem of BASELINE config 5, whose restarts last one iteration)
LB_HD void lbfgsb_init(State &s, const Work &w, int n, int m, const double *x0, const double *l,
                       const double *u, const int *nbd, const Coop c = Coop{0, 1}) {
  s = State{};
  s.n = n;
  s.m = m;
  for (int i = c.lane; i < n; i += c.nl) {
    double xi = x0[i];
    if (nbd[i] == 1 || nbd[i] == 2) xi = fmax(xi, l[i]);
    if (nbd[i] == 2 || nbd[i] == 3) xi = fmin(xi, u[i]);
    w.x[i] = xi;
    w.g[i] = 0.0;
  }
  s.f = 0.0;
  s.stage = S_INIT;
  s.task = T_START;
  s.msg = M_NONE;
  s.nit = 0;
  s.nfev = 0;
  s.status = 2;
}

// Advance the problem until it needs f and g at w.x (LB_NEED_FG: the caller stores them in
// s.f / w.g and calls again) or terminates (LB_DONE: result in w.x, s.f, w.g, s.nit,
// s.nfev, s.status, s.task, s.msg).
// `coop`: {0, 1} for a thread that owns its problem alone; {lane, 64} when the 64 lanes of a
// wave run ONE problem together (every lane calls with identical State; see struct Coop).
//
// DIRECT form: with an evaluation functor `fg(State &, const Work &)` -- which leaves f in s.f and
// g in w.g for the point in w.x -- the routine does not return for evaluations; it calls fg where
// the reverse-communication form returns LB_NEED_FG and goes on exactly where a re-entry would
// (same operations in the same order: the two forms give the same bits, tested).  What it saves
// is the way out of and back into a state machine that the device compiler has inlined into its
// caller's loop: ~2 k cycles per evaluation of exits through nested loops, re-dispatch on the
// stage and copies between register assignments (profiles/r3/lbfgsb_gap_stamps.txt).  For a
// caller whose lanes all work on ONE problem (Coop) -- divergent callers need the returning form.
//
// TWO-VARIABLE form of the line search (DIRECT callers may add `fg2(State &, double x0, double x1,
// double &g0, double &g1, double xlast0, double xlast1, double glast0, double glast1)` -- the point
// evaluated last with its gradient; its value is s.flast): from a search's second call on, the trial
// point, the direction, the gradient and the cache of the point evaluated last live in REGISTERS (the same values in every
// lane) instead of the workspace -- per evaluation that is the trial point written, read back for the
// cache check and again for the evaluation, the gradient written and read, and both copied to the
// cache: half a dozen LDS round trips on a chain that one wave walks alone.  The first call of a
// search (step bound, saves) keeps the workspace form; the registers go back to the workspace when
// the search ends.  Same operations on the same values in the same order: same bits (tested against
// the other two forms on the host, tests/test_lbfgsb_host.py, and on the device).
struct ReverseCommunication {};
struct NoTwoVariableForm {};
template <bool VL = true, class FG = ReverseCommunication, class FG2 = NoTwoVariableForm>
LB_HD int lbfgsb_advance(State &s, const Work &w, const double *l, const double *u,
                         const int *nbd, const Options &opt, const Coop coop_in = Coop{0, 1},
                         FG fg = FG{}, FG2 fg2 = FG2{}) {
  constexpr bool DIRECT = !std::is_same<FG, ReverseCommunication>::value;
  constexpr bool TWO = DIRECT && !std::is_same<FG2, NoTwoVariableForm>::value;
  Coop coop = coop_in;
  const int n = LB_UNI(s.n, coop_in), m = LB_UNI(s.m, coop_in);
  bool first_ls = false;
  bool resume_ls = (s.stage == S_FG_LNSRCH);

  if (s.stage == S_FINISHED) return LB_DONE;
  // (LB_BIG_LAZY: this call sees the whole problem in the DIRECT form and zeroes the outside matrices in front of
  // the first formk; the reverse-communication form returns in between and zeroes them on its first call)
  bool big_stale = w.vm == LB_BIG_LAZY;
  if (!DIRECT && big_stale && s.stage == S_INIT) zero_big(w, m, coop_in);
  LB_MARK(s, 15);  // (everything since the previous return: the evaluation)

  // fresh f, g at w.x have arrived: remember the point (SciPy's ScalarFunction cache)
  auto save_evaluated_point = [&]() {
    LB_PHASE_BEGIN(20);
    for (int i = coop.lane; i < n; i += coop.nl) { w.xlast[i] = w.x[i]; w.glast[i] = w.g[i]; }
    LB_LANES_SYNC();
    s.flast = s.f;
    LB_PHASE_END(7);
  };
  if (s.stage == S_FG_START || s.stage == S_FG_LNSRCH) save_evaluated_point();

  if (s.stage == S_INIT) {
    s.col = 0; s.head = 0; s.theta = 1.0; s.iupdat = 0; s.updatd = 0;
    s.iback = 0; s.itail = 0; s.ifun = 0; s.iter = 0; s.nfgv = 0; s.nseg = 0;
    s.nfree = n; s.info = 0; s.iword = 0; s.nact = 0; s.nenter = 0; s.ileave = 0;
    s.fold = 0.0; s.dnorm = 0.0; s.sbgnrm = 0.0; s.stp = 0.0; s.xstep = 0.0; s.stpmx = 0.0;
    s.gd = 0.0; s.gdold = 0.0; s.dtd = 0.0; s.wrk = 0;
    // errclb
    if (n <= 0) { finish(s, T_ERROR, M_N_LE_0); return LB_DONE; }
    if (m <= 0) { finish(s, T_ERROR, M_M_LE_0); return LB_DONE; }
    if (opt.factr < 0.0) { finish(s, T_ERROR, M_FACTR_NEG); return LB_DONE; }
    if (VL && coop.nl > 1 && n <= coop.nl) {  // one variable per lane; the FIRST offending variable names the error
      bool bad = false, empty = false;
      if (coop.lane < n) {
        const int nb = nbd[coop.lane];
        bad = nb < 0 || nb > 3;
        empty = nb == 2 && l[coop.lane] > u[coop.lane];
      }
      const unsigned long long mb = lanes_ballot(bad), me = lanes_ballot(empty);
      if (mb | me) {
        const unsigned long long first = (mb | me) & (0ull - (mb | me));  // lowest set bit
        finish(s, T_ERROR, (mb & first) ? M_INVALID_NBD : M_NO_FEASIBLE);
        return LB_DONE;
      }
    } else
    for (int i = 0; i < n; ++i) {
      if (nbd[i] < 0 || nbd[i] > 3) { finish(s, T_ERROR, M_INVALID_NBD); return LB_DONE; }
      if (nbd[i] == 2 && l[i] > u[i]) { finish(s, T_ERROR, M_NO_FEASIBLE); return LB_DONE; }
    }
    // active: project x, classify the variables
    s.prjctd = 0; s.cnstnd = 0; s.boxed = 1;
    if (VL && coop.nl > 1 && n <= coop.nl) {  // one variable per lane (every output depends on variable i alone)
      const int i = coop.lane;
      bool prj = false, notboxed = false, cns = false;
      if (i < n) {
        const int nb = nbd[i];
        if (nb > 0) {
          if (nb <= 2 && w.x[i] <= l[i]) {
            if (w.x[i] < l[i]) { prj = true; w.x[i] = l[i]; }
          } else if (nb >= 2 && w.x[i] >= u[i]) {
            if (w.x[i] > u[i]) { prj = true; w.x[i] = u[i]; }
          }
        }
        notboxed = nb != 2;
        if (nb == 0) {
          w.iwhere[i] = -1;
        } else {
          cns = true;
          w.iwhere[i] = (nb == 2 && u[i] - l[i] <= 0.0) ? 3 : 0;
        }
      }
      s.prjctd = lanes_any(prj) ? 1 : 0;
      s.boxed = lanes_any(notboxed) ? 0 : 1;
      s.cnstnd = lanes_any(cns) ? 1 : 0;
      LB_LANES_SYNC();
    } else {
    for (int i = 0; i < n; ++i) {
      if (nbd[i] > 0) {
        if (nbd[i] <= 2 && w.x[i] <= l[i]) {
          if (w.x[i] < l[i]) { s.prjctd = 1; w.x[i] = l[i]; }
        } else if (nbd[i] >= 2 && w.x[i] >= u[i]) {
          if (w.x[i] > u[i]) { s.prjctd = 1; w.x[i] = u[i]; }
        }
      }
    }
    for (int i = 0; i < n; ++i) {
      if (nbd[i] != 2) s.boxed = 0;
      if (nbd[i] == 0) {
        w.iwhere[i] = -1;
      } else {
        s.cnstnd = 1;
        if (nbd[i] == 2 && u[i] - l[i] <= 0.0) w.iwhere[i] = 3;
        else w.iwhere[i] = 0;
      }
    }
    }
    s.stage = S_FG_START;
    s.task = T_FG;
    ++s.nfev;
    if constexpr (DIRECT) {
      LB_MARK(s, 18);
      fg(s, w);
      LB_MARK(s, 15);
      save_evaluated_point();
    } else {
      return LB_NEED_FG;
    }
  }

  if (s.stage == S_FG_START) {
    s.nfgv = 1;
    s.sbgnrm = projgr(n, l, u, nbd, w.x, w.g, coop);
    if (s.sbgnrm <= opt.pgtol) { finish(s, T_CONVERGENCE, M_PGTOL); return LB_DONE; }
    resume_ls = false;
  }

  // otherwise S_FG_LNSRCH: f and g at the trial point have arrived; resume the line search

  for (;;) {
    // The lane number is made opaque once per pass: every per-lane address below derives from it, and
    // computed from a loop-invariant they are all hoisted out of this loop and kept live around it --
    // 340 spilled registers in the 128-wide restart kernels (profiles/r3/resource_usage.txt).
    if (VL) LB_OPAQUE_LANE(coop.lane);
    if (resume_ls) {
      resume_ls = false;
      first_ls = false;
    } else {
      // (a new iteration: the line search's saved locals are dead -- dcsrch sets every one of them when the next
      // search starts -- and saying so here keeps the compiler from carrying 13 fp64 values through the
      // once-per-iteration routines below)
      s.brackt = 0; s.ls_stage = 0;
      s.ginit = s.gtest = s.gx = s.gy = s.finit = s.fx = s.fy = s.stx = s.sty = s.stmin = s.stmax = 0.0;
      s.width = s.width1 = 0.0;
      // (... and so are the scalars lnsrlb sets when a search starts)
      s.xstep = s.gd = s.gdold = s.stp = s.dtd = s.dnorm = s.stpmx = s.fold = 0.0;
      s.ifun = s.iback = 0;
      s.iword = -1;
      if (!s.cnstnd && s.col > 0) {
        for (int i = coop.lane; i < n; i += coop.nl) w.z[i] = w.x[i];
        LB_LANES_SYNC();
        s.wrk = s.updatd;
        s.nseg = 0;
      } else {
        const IterArgs ia{n, m, LB_UNI(s.col, coop), LB_UNI(s.head, coop), LB_UNI(s.nfree, coop),
                          LB_UNI(s.nenter, coop), LB_UNI(s.ileave, coop), LB_UNI(s.updatd, coop),
                          LB_UNI(s.iupdat, coop), s.theta, s.sbgnrm, coop};
        LB_PHASE_BEGIN(21);
        const int rc = cauchy<VL>(ia, w, l, u, nbd);
        LB_PHASE_END(0);
        s.nseg = rc >> 8;
        if (rc & 0xff) {  // singular triangular system: refresh the memory
          refresh_memory(s);
          continue;
        }
        {
          LB_PHASE_BEGIN(22);
          freev<VL>(s, w, coop);
          LB_PHASE_END(8);
        }
        s.nact = n - s.nfree;
      }
      if (s.nfree != 0 && s.col != 0) {
        const IterArgs ia{n, m, LB_UNI(s.col, coop), LB_UNI(s.head, coop), LB_UNI(s.nfree, coop),
                          LB_UNI(s.nenter, coop), LB_UNI(s.ileave, coop), LB_UNI(s.updatd, coop),
                          LB_UNI(s.iupdat, coop), s.theta, s.sbgnrm, coop};
        if (s.wrk) {
          LB_PHASE_BEGIN(23);
          if (DIRECT && big_stale) {
            zero_big(w, m, coop);
            big_stale = false;
          }
          const int fk = formk(ia, w);
          LB_PHASE_END(1);
          if (fk) { refresh_memory(s); continue; }
        }
        int rc;
        {
          LB_PHASE_BEGIN(24);
          const int cm = cmprlb(s, w, coop);
          LB_PHASE_END(2);
          if (cm) { refresh_memory(s); continue; }
        }
        {
          LB_PHASE_BEGIN(25);
          rc = subsm<VL>(ia, w, l, u, nbd);
          LB_PHASE_END(3);
        }
        s.iword = rc >> 8;
        if (rc & 0xff) { refresh_memory(s); continue; }
      }
      {
        LB_PHASE_BEGIN(26);
        for (int i = coop.lane; i < n; i += coop.nl) w.d[i] = w.z[i] - w.x[i];
        LB_LANES_SYNC();
        LB_PHASE_END(12);
      }
      first_ls = true;
    }

    s.info = 0;
    int ls_rc;
    {
      LB_PHASE_BEGIN(27);
      ls_rc = lnsrlb<VL>(s, w, l, u, nbd, first_ls ? 1 : 0, coop);
      LB_PHASE_END(4);
    }
    if constexpr (TWO) {
      if (ls_rc && n == 2) {
        // ---- the rest of this search in registers (see the header of this function) ----
        const double ftol_ls = 1e-3, gtol_ls = 0.9, xtol_ls = 0.1;   // (lnsrlb's constants)
        double x0 = w.x[0], x1 = w.x[1];
        const double d0 = w.d[0], d1 = w.d[1], t0 = w.t[0], t1 = w.t[1], z0 = w.z[0], z1 = w.z[1];
        double g0 = w.g[0], g1 = w.g[1];
        double xl0 = w.xlast[0], xl1 = w.xlast[1], gl0 = w.glast[0], gl1 = w.glast[1];
        while (s.iback < opt.maxls) {
          // SciPy's ScalarFunction cache: a request at the point evaluated last is served from it
          const bool differs = (x0 != xl0) || (x1 != xl1);
          if (!differs) {
            s.f = s.flast;
            g0 = gl0;
            g1 = gl1;
          } else {
            s.stage = S_FG_LNSRCH;
            s.task = T_FG;
            ++s.nfev;
            LB_MARK(s, 18);
            // (the functor also sees the point evaluated last, its value in s.flast and its gradient: an
            // evaluation that depends on LESS than the fp64 point -- the network reads its float32 image --
            // may answer from them, bore_argmax.hip)
            fg2(s, x0, x1, g0, g1, xl0, xl1, gl0, gl1);
            LB_MARK(s, 15);
            xl0 = x0; xl1 = x1; gl0 = g0; gl1 = g1;
            s.flast = s.f;
            LB_MARK(s, 7);
          }
          // lnsrlb, not the first call of the search
          s.info = 0;
          {
            double sum = 0.0;
            sum += g0 * d0;
            sum += g1 * d1;
            s.gd = sum;
          }
          if (s.ifun == 0) {
            s.gdold = s.gd;
            if (s.gd >= 0.0) {
              s.info = -4;
              ls_rc = 0;
              break;
            }
          }
          LB_MARK(s, 16);
          dcsrch(s, s.f, s.gd, s.stp, ftol_ls, gtol_ls, xtol_ls, 0.0, s.stpmx);
          LB_MARK(s, 17);
          s.xstep = s.stp * s.dnorm;
          if (s.ls_task != LS_CONV && s.ls_task != LS_WARN) {
            ++s.ifun;
            ++s.nfgv;
            s.iback = s.ifun - 1;
            if (s.stp == 1.0) {
              x0 = z0;
              x1 = z1;
            } else {
              x0 = s.stp * d0 + t0;
              x1 = s.stp * d1 + t1;
            }
            ls_rc = 1;
          } else {
            ls_rc = 0;
            break;
          }
          LB_MARK(s, 4);
        }
        // (every lane holds the same values: redundant stores of identical data)
        w.x[0] = x0; w.x[1] = x1;
        w.g[0] = g0; w.g[1] = g1;
        w.xlast[0] = xl0; w.xlast[1] = xl1;
        w.glast[0] = gl0; w.glast[1] = gl1;
        LB_LANES_SYNC();
      }
    }
    if (ls_rc) {
      if (s.iback < opt.maxls) {
        // SciPy's ScalarFunction serves a request at the point it evaluated last from its
        // cache (no call, nfev unchanged); a collapsed bracket asks for such points.
        LB_PHASE_BEGIN(28);
        bool differs = false;
        for (int i = coop.lane; i < n; i += coop.nl) differs = differs || (w.x[i] != w.xlast[i]);
        const bool cached = coop.nl > 1 ? !lanes_any(differs) : !differs;
        LB_PHASE_END(10);
        if (cached) {
          s.f = s.flast;
          for (int i = coop.lane; i < n; i += coop.nl) w.g[i] = w.glast[i];
          LB_LANES_SYNC();
          resume_ls = true;
          continue;
        }
        s.stage = S_FG_LNSRCH;
        s.task = T_FG;
        ++s.nfev;
        LB_MARK(s, 18);
        if constexpr (DIRECT) {
          fg(s, w);
          LB_MARK(s, 15);
          save_evaluated_point();
          resume_ls = true;  // (what a re-entry in stage S_FG_LNSRCH does)
          continue;
        } else {
          return LB_NEED_FG;
        }
      }
      // maxls trial points used up: handled like a failed search (the trial x is dropped)
    }
    if (s.info != 0 || s.iback >= opt.maxls) {
      // restore the previous iterate
      for (int i = coop.lane; i < n; i += coop.nl) { w.x[i] = w.t[i]; w.g[i] = w.r[i]; }
      LB_LANES_SYNC();
      s.f = s.fold;
      if (s.col == 0) {
        // abnormal termination
        if (s.info == 0) { s.info = -9; --s.nfgv; --s.ifun; --s.iback; }
        ++s.iter;
        finish(s, T_ABNORMAL, M_NONE);
        return LB_DONE;
      }
      refresh_memory(s);  // restart from the steepest-descent-like state
      continue;
    }
    // new iterate accepted
    ++s.iter;
    {
      LB_PHASE_BEGIN(29);
      s.sbgnrm = projgr(n, l, u, nbd, w.x, w.g, coop);
      LB_PHASE_END(9);
    }
    // --- what the SciPy driver does on NEW_X ---
    ++s.nit;
    if (s.nit >= opt.maxiter) { finish(s, T_STOP, M_MAXITER); return LB_DONE; }
    if (s.nfev > opt.maxfun) { finish(s, T_STOP, M_MAXFUN); return LB_DONE; }
    // --- convergence tests ---
    if (s.sbgnrm <= opt.pgtol) { finish(s, T_CONVERGENCE, M_PGTOL); return LB_DONE; }
    {
      const double ddum = fmax(fmax(fabs(s.fold), fabs(s.f)), 1.0);
      if ((s.fold - s.f) <= LB_EPSMCH * opt.factr * ddum) {
        if (s.iback >= 10) s.info = -5;
        finish(s, T_CONVERGENCE, M_FACTR);
        return LB_DONE;
      }
    }
    // --- BFGS update: r = g - g_old (y), d = step (s) ---
    LB_PHASE_BEGIN(30);
    for (int i = coop.lane; i < n; i += coop.nl) w.r[i] = w.g[i] - w.r[i];
    LB_LANES_SYNC();
    {
      const double rr = ddot_c<VL>(n, w.r, w.r, coop);
      double dr, ddum;
      if (s.stp == 1.0) {
        dr = s.gd - s.gdold;
        ddum = -s.gdold;
      } else {
        dr = (s.gd - s.gdold) * s.stp;
        for (int i = coop.lane; i < n; i += coop.nl) w.d[i] *= s.stp;
        LB_LANES_SYNC();
        ddum = -s.gdold * s.stp;
      }
      if (dr <= LB_EPSMCH * ddum) {  // curvature too small: skip the update
        s.updatd = 0;
        continue;
      }
      s.updatd = 1;
      ++s.iupdat;
      LB_PHASE_END(11);
      {
        LB_PHASE_BEGIN(31);
        matupd<VL>(s, w, rr, dr, coop);
        LB_PHASE_END(5);
      }
      int ft;
      {
        LB_PHASE_BEGIN(19);
        ft = formt(m, w, LB_UNI(s.col, coop), s.theta, coop);
        LB_PHASE_END(6);
      }
      if (ft) {
        refresh_memory(s);
        continue;
      }
    }
  }
}

}  // namespace lbfgsb
