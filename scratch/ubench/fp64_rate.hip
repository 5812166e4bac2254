// Microbenchmark: cycles per fp64 / fp32 instruction for one wave alone on a SIMD -- dependent chain
// against four independent chains (gfx950).  hipcc --offload-arch=gfx950 -O3 fp64_rate.hip -o fp64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void k(double *out, long long *cyc, double a, double b) {
  double x0 = a + threadIdx.x, x1 = a * 2, x2 = a * 3, x3 = a * 4;
  float f0 = (float)a, f1 = (float)a * 2, f2 = 3.f, f3 = 4.f, fb = (float)b;
  const long long t0 = clock64();
#pragma unroll 1
  for (int i = 0; i < 256; ++i) {
    if constexpr (MODE == 0) {  // 16 dependent fp64 fma
#pragma unroll
      for (int j = 0; j < 16; ++j) x0 = __builtin_fma(x0, b, a);
    } else if constexpr (MODE == 1) {  // 16 fp64 fma in four independent chains
#pragma unroll
      for (int j = 0; j < 4; ++j) { x0 = __builtin_fma(x0, b, a); x1 = __builtin_fma(x1, b, a); x2 = __builtin_fma(x2, b, a); x3 = __builtin_fma(x3, b, a); }
    } else if constexpr (MODE == 2) {  // 16 dependent fp32 fma
#pragma unroll
      for (int j = 0; j < 16; ++j) f0 = __builtin_fmaf(f0, fb, 1.5f);
    } else if constexpr (MODE == 3) {  // 16 fp32 fma, four chains
#pragma unroll
      for (int j = 0; j < 4; ++j) { f0 = __builtin_fmaf(f0, fb, 1.5f); f1 = __builtin_fmaf(f1, fb, 1.5f); f2 = __builtin_fmaf(f2, fb, 1.5f); f3 = __builtin_fmaf(f3, fb, 1.5f); }
    } else if constexpr (MODE == 4) {  // 4 dependent IEEE fp64 divisions
#pragma unroll
      for (int j = 0; j < 4; ++j) x0 = a / (x0 + b);
    } else if constexpr (MODE == 5) {  // 4 independent IEEE fp64 divisions
      x0 = a / (x0 + b); x1 = a / (x1 + b); x2 = a / (x2 + b); x3 = a / (x3 + b);
    } else if constexpr (MODE == 6) {  // 4 dependent fp64 sqrt
#pragma unroll
      for (int j = 0; j < 4; ++j) x0 = sqrt(x0 + b);
    } else if constexpr (MODE == 7) {  // 16 dependent v_rcp_f64
#pragma unroll
      for (int j = 0; j < 16; ++j) x0 = __builtin_amdgcn_rcp(x0);
    }
  }
  const long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + f0 + f1 + f2 + f3;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  double *out; long long *cyc;
  hipMalloc(&out, 1 << 20); hipMalloc(&cyc, 8 * 4096);
  const char *names[] = {"16 dependent fp64 fma", "16 fp64 fma in 4 chains", "16 dependent fp32 fma", "16 fp32 fma in 4 chains",
                         "4 dependent fp64 divisions", "4 independent fp64 divisions", "4 dependent fp64 sqrt", "16 dependent v_rcp_f64"};
  for (int waves = 1; waves <= 2; ++waves) {   // waves per SIMD: 1 (64 x 4 threads on a CU) or 2
    for (int mode = 0; mode < 8; ++mode) {
      const int threads = 256 * waves, blocks = 256;
      for (int rep = 0; rep < 2; ++rep) {
        switch (mode) {
          case 0: hipLaunchKernelGGL(k<0>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 1: hipLaunchKernelGGL(k<1>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 2: hipLaunchKernelGGL(k<2>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 3: hipLaunchKernelGGL(k<3>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 4: hipLaunchKernelGGL(k<4>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 5: hipLaunchKernelGGL(k<5>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 6: hipLaunchKernelGGL(k<6>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
          case 7: hipLaunchKernelGGL(k<7>, blocks, threads, 0, 0, out, cyc, 1.0000001, 0.9999999); break;
        }
        hipDeviceSynchronize();
      }
      long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double s = 0; for (int i = 0; i < 256; ++i) s += h[i];
      const int per_iter = mode < 4 ? 16 : (mode == 7 ? 16 : 4);
      printf("%d wave(s)/SIMD  %-32s %8.1f cycles per operation\n", waves, names[mode], s / 256 / 256 / per_iter);
    }
  }
  return 0;
}
