// Does a store to a cache line lose to an in-flight L1 fill of the SAME line requested by another wave of the workgroup?
// (Round 4's eight-wave wide fit stored m / v tiles whose last 64 bytes came back stale, with neighbouring tiles -- other
// waves' -- sharing cache lines at the tile boundaries because a model's m / v base is only 4-byte aligned)
// Per workgroup: one fresh 128-byte line X of a large buffer (never touched before: the fill comes from HBM).
//   wave 1 loads the SECOND half of X (bytes 64..127): L1 miss, fill pending;
//   wave 0, `delay` sleeps later, stores NEW to the FIRST half of X (bytes 0..63): write-through;
//   both wait for vmcnt(0), barrier; then wave 1 loads the first half with a plain load.
// stale = the load returned OLD although the store had completed before the barrier.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/l1_fill_race.hip -o tools/ubench/l1_fill_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void race(float *buf, int *stale, int delay, int inv) {
  float *x = buf + (size_t)blockIdx.x * 64;  // 256 B per workgroup: line X = the first 128 B
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float v = 0.f;
  if (wv == 1) {
    if (lane < 16) v = __builtin_nontemporal_load(x + 16 + lane) * 0.f + x[16 + lane];  // (plain load; second half)
  } else {
    for (int i = 0; i < delay; ++i) __builtin_amdgcn_s_sleep(1);
    if (lane < 16) x[lane] = 2.0f;  // NEW
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (inv) asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
  if (wv == 1 && lane < 16) {
    const float w = x[lane];  // plain load of the first half
    if (w != 2.0f) atomicAdd(stale, 1);
    if (v == 123.f) atomicAdd(stale, 1000000);  // (keeps v alive)
  }
}
int main() {
  const int WG = 4096;
  float *buf;
  int *stale;
  hipMalloc((void **)&stale, 4);
  for (int inv = 0; inv < 2; ++inv)
    for (int delay : {0, 1, 2, 4, 8, 16, 32, 64, 128, 256}) {
      hipMalloc((void **)&buf, (size_t)WG * 256);  // fresh memory every trial: cold lines
      std::vector<float> old((size_t)WG * 64, 1.0f);
      hipMemcpy(buf, old.data(), old.size() * 4, hipMemcpyHostToDevice);
      hipMemset(stale, 0, 4);
      hipDeviceSynchronize();
      hipLaunchKernelGGL(race, dim3(WG), dim3(128), 0, 0, buf, stale, delay, inv);
      int s = -1;
      hipMemcpy(&s, stale, 4, hipMemcpyDeviceToHost);
      printf("buffer_inv %d delay %3d sleeps: stale reads of the stored half: %d of %d\n", inv, delay, s, WG * 16);
      hipFree(buf);
    }
  return 0;
}
