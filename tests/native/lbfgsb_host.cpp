// Host build of bore_amd/csrc/lbfgsb.h (the L-BFGS-B state machine that runs inside the
// HIP argmax kernel) so that it can be driven from pytest on a machine without a GPU.
// f/g come from a Python callback.  Test scaffolding only.
#include <cstdlib>
#include <vector>

#include "../../bore_amd/csrc/lbfgsb.h"

extern "C" {

typedef void (*fg_callback)(int n, const double *x, double *f, double *g);

// Returns the number of evaluations asked of the callback.  out_i = {nit, nfev, status, task, msg}.
// form 0: reverse communication (lbfgsb_advance returns for every evaluation); 1: the DIRECT form
// (the routine calls the evaluation); 2: DIRECT with the two-variable line search in registers
// (n == 2; other n fall back to form 1 inside the routine).
int lbfgsb_host_minimize_form(int form, int n, int m, const double *x0, const double *l, const double *u,
                              const int *nbd, double factr, double pgtol, int maxiter, int maxfun,
                              int maxls, fg_callback fg, double *x_out, double *f_out, double *g_out,
                              int *out_i) {
  using namespace lbfgsb;
  std::vector<double> dw(dwork_size(n, m), 0.0);
  std::vector<int> iw(iwork_size(n), 0);
  State s;
  Work w = make_work(dw.data(), iw.data(), n, m);
  Options opt{m, factr, pgtol, maxiter, maxfun, maxls};
  lbfgsb_init(s, w, n, m, x0, l, u, nbd);
  int rounds = 0;
  auto eval = [&](State &st, const Work &wk) {
    double f;
    fg(n, wk.x, &f, wk.g);
    st.f = f;
    ++rounds;
  };
  auto eval2 = [&](State &st, double a, double b, double &ga, double &gb, double, double, double, double) {
    double xx[2] = {a, b}, gg[2], f;
    fg(2, xx, &f, gg);
    st.f = f;
    ga = gg[0];
    gb = gg[1];
    ++rounds;
  };
  if (form == 0) {
    while (lbfgsb_advance(s, w, l, u, nbd, opt) == LB_NEED_FG) {
      eval(s, w);
      if (rounds > 10000000) break;
    }
  } else if (form == 1) {
    lbfgsb_advance<true>(s, w, l, u, nbd, opt, Coop{0, 1}, eval);
  } else {
    lbfgsb_advance<true>(s, w, l, u, nbd, opt, Coop{0, 1}, eval, eval2);
  }
  for (int i = 0; i < n; ++i) {
    x_out[i] = w.x[i];
    g_out[i] = w.g[i];
  }
  *f_out = s.f;
  out_i[0] = s.nit;
  out_i[1] = s.nfev;
  out_i[2] = s.status;
  out_i[3] = s.task;
  out_i[4] = s.msg;
  return rounds;
}

// Returns the number of state-machine round trips.  out_i = {nit, nfev, status, task, msg}.
int lbfgsb_host_minimize(int n, int m, const double *x0, const double *l, const double *u,
                         const int *nbd, double factr, double pgtol, int maxiter, int maxfun,
                         int maxls, fg_callback fg, double *x_out, double *f_out, double *g_out,
                         int *out_i) {
  using namespace lbfgsb;
  std::vector<double> dw(dwork_size(n, m), 0.0);
  std::vector<int> iw(iwork_size(n), 0);
  State s;
  Work w = make_work(dw.data(), iw.data(), n, m);
  Options opt{m, factr, pgtol, maxiter, maxfun, maxls};
  lbfgsb_init(s, w, n, m, x0, l, u, nbd);
  int rounds = 0;
  while (lbfgsb_advance(s, w, l, u, nbd, opt) == LB_NEED_FG) {
    double f;
    fg(n, w.x, &f, w.g);
    s.f = f;
    ++rounds;
    if (rounds > 10000000) break;
  }
  for (int i = 0; i < n; ++i) {
    x_out[i] = w.x[i];
    g_out[i] = w.g[i];
  }
  *f_out = s.f;
  out_i[0] = s.nit;
  out_i[1] = s.nfev;
  out_i[2] = s.status;
  out_i[3] = s.task;
  out_i[4] = s.msg;
  return rounds;
}
}
