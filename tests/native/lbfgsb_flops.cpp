// Host build of bore_amd/csrc/lbfgsb.h with every float64 operation COUNTED: `double` is replaced, for this
// translation unit only, by a wrapper whose arithmetic operators bump a counter (adds / subtracts / multiplies /
// compares-free: one each; a division or a square root: one each, counted apart).  Drives the optimiser on a
// caller-supplied objective like lbfgsb_host.cpp; bench.py's restart roofline uses the per-evaluation counts
// (tools/lbfgsb_flops.py).  Test / measurement scaffolding only -- the product is the HIP build of the same header.
#include <math.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>
#include <vector>

namespace counted {
struct Counts {
  unsigned long long addmul, divsqrt;
};
static Counts g;
struct F64 {
  double v;
  F64() = default;
  F64(double x) : v(x) {}
  F64(int x) : v((double)x) {}
  explicit operator int() const { return (int)v; }
  explicit operator float() const { return (float)v; }
  explicit operator long long() const { return (long long)v; }
  double raw() const { return v; }
};
inline F64 operator+(F64 a, F64 b) { ++g.addmul; return F64(a.v + b.v); }
inline F64 operator-(F64 a, F64 b) { ++g.addmul; return F64(a.v - b.v); }
inline F64 operator*(F64 a, F64 b) { ++g.addmul; return F64(a.v * b.v); }
inline F64 operator/(F64 a, F64 b) { ++g.divsqrt; return F64(a.v / b.v); }
inline F64 operator-(F64 a) { return F64(-a.v); }
inline F64 &operator+=(F64 &a, F64 b) { ++g.addmul; a.v += b.v; return a; }
inline F64 &operator-=(F64 &a, F64 b) { ++g.addmul; a.v -= b.v; return a; }
inline F64 &operator*=(F64 &a, F64 b) { ++g.addmul; a.v *= b.v; return a; }
inline F64 &operator/=(F64 &a, F64 b) { ++g.divsqrt; a.v /= b.v; return a; }
inline bool operator<(F64 a, F64 b) { return a.v < b.v; }
inline bool operator>(F64 a, F64 b) { return a.v > b.v; }
inline bool operator<=(F64 a, F64 b) { return a.v <= b.v; }
inline bool operator>=(F64 a, F64 b) { return a.v >= b.v; }
inline bool operator==(F64 a, F64 b) { return a.v == b.v; }
inline bool operator!=(F64 a, F64 b) { return a.v != b.v; }
inline F64 sqrt(F64 a) { ++g.divsqrt; return F64(::sqrt(a.v)); }
inline F64 fabs(F64 a) { return F64(::fabs(a.v)); }
inline F64 fmax(F64 a, F64 b) { return F64(::fmax(a.v, b.v)); }
inline F64 fmin(F64 a, F64 b) { return F64(::fmin(a.v, b.v)); }
}  // namespace counted
using counted::fabs;
using counted::fmax;
using counted::fmin;
using counted::sqrt;

#define double counted::F64
#include "../../bore_amd/csrc/lbfgsb.h"
#undef double

extern "C" {
typedef void (*fg_callback)(int n, const double *x, double *f, double *g);

// out_i = {nit, nfev, status}; out_c = {adds + subtracts + multiplies, divisions + square roots} of the optimiser
// alone (the objective runs in the caller's arithmetic).  Reverse-communication form (every variant of the routine
// performs the same operations on the same values).
int lbfgsb_flops_minimize(int n, int m, const double *x0, const double *l, const double *u, const int *nbd,
                          double factr, double pgtol, int maxiter, int maxfun, int maxls, fg_callback fg,
                          double *x_out, double *f_out, int *out_i, unsigned long long *out_c) {
  using namespace lbfgsb;
  using counted::F64;
  std::vector<F64> dw(dwork_size(n, m), F64(0.0)), cx0(x0, x0 + n), cl(l, l + n), cu(u, u + n);
  std::vector<int> iw(iwork_size(n), 0);
  State s;
  Work w = make_work(dw.data(), iw.data(), n, m);
  Options opt{m, F64(factr), F64(pgtol), maxiter, maxfun, maxls};
  counted::g = counted::Counts{0, 0};
  lbfgsb_init(s, w, n, m, cx0.data(), cl.data(), cu.data(), nbd);
  int rounds = 0;
  std::vector<double> xx(n), gg(n);
  while (lbfgsb_advance(s, w, cl.data(), cu.data(), nbd, opt) == LB_NEED_FG) {
    for (int i = 0; i < n; ++i) xx[i] = w.x[i].raw();
    double f;
    fg(n, xx.data(), &f, gg.data());
    s.f = F64(f);
    for (int i = 0; i < n; ++i) w.g[i] = F64(gg[i]);
    if (++rounds > 10000000) break;
  }
  for (int i = 0; i < n; ++i) x_out[i] = w.x[i].raw();
  *f_out = s.f.raw();
  out_i[0] = s.nit;
  out_i[1] = s.nfev;
  out_i[2] = s.status;
  out_c[0] = counted::g.addmul;
  out_c[1] = counted::g.divsqrt;
  return rounds;
}
}
