// Cycles of one dcsrch call (More'-Thuente step of lbfgsb.h) as one wave runs it: every lane the same scalars, as in
// the restart kernels.  A synthetic line function with float32-rounded values (the network's arithmetic) so that
// the searches take the paths the real ones take: bracketing, interpolation, the noise floor.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off tools/ubench/dcsrch.hip -o /tmp/dcsrch && /tmp/dcsrch
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../../bore_amd/csrc/lbfgsb.h"

__device__ __forceinline__ void phi(double t, double a, double &f, double &g) {
  // a kink at a (the ReLU network along a line): the curvature condition cannot hold across it, the search brackets
  // and shrinks its interval down to the float32 spacing of t
  const float tf = (float)t, af = (float)a, d = tf - af;
  f = (double)(1.0f + 0.5f * fabsf(d) + 0.05f * d * d);
  g = (double)((d < 0.f ? -0.5f : 0.5f) + 0.1f * d);
}

__global__ __launch_bounds__(64) void bench(long long *out, int searches) {
  lbfgsb::State s;
  long long cyc = 0, calls = 0;
  double acc = 0.0;
  for (int k = 0; k < searches; ++k) {
    const double a = 0.35 + 0.01 * (k % 50);
    double f, g, stp = 1.0;
    phi(0.0, a, f, g);
    if (!(g < 0.0)) continue;
    s.ls_task = lbfgsb::LS_START;
    s.stpmx = 1e10;
    int it = 0;
    for (;;) {
      const long long c0 = clock64();
      lbfgsb::dcsrch(s, f, g, stp, 1e-3, 0.9, 0.1, 0.0, 1e10);
      cyc += clock64() - c0;
      ++calls;
      if (s.ls_task != lbfgsb::LS_FG || ++it > 20) break;
      phi(stp, a, f, g);
    }
    acc += stp;
  }
  if (threadIdx.x == 0) {
    out[0] = cyc;
    out[1] = calls;
    out[2] = __double_as_longlong(acc);
  }
}

int main() {
  long long *d, h[3];
  hipMalloc(&d, 24);
  for (int rep = 0; rep < 3; ++rep) {
    bench<<<1, 64>>>(d, 2000);
    hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("dcsrch: %lld calls, %.0f cycles per call (clock64 pair included), checksum %016llx\n", h[1], (double)h[0] / h[1],
           (unsigned long long)h[2]);
  }
  return 0;
}
