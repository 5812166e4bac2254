// Does the link between this device and the host support native atomics (what lbfgsb_body's PUBLISH = 1 flag
// exchange on pinned memory rests on)?  hipcc --offload-arch=gfx950 tools/ubench/host_atomics.hip -o tools/ubench/host_atomics
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void xchg(int *p, int *old) { *old = __hip_atomic_exchange(p, 7, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
int main() {
  int v = -1, dev = 0;
  hipGetDevice(&dev);
  hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeHostNativeAtomicSupported, dev);
  printf("hipDeviceAttributeHostNativeAtomicSupported: rc %d value %d\n", (int)e, v);
  int *h = nullptr, *o = nullptr;
  hipHostMalloc((void **)&h, 8, hipHostMallocMapped);
  h[0] = 3; h[1] = -1;
  hipMalloc((void **)&o, 4);
  hipLaunchKernelGGL(xchg, dim3(1), dim3(1), 0, 0, h, o);
  e = hipDeviceSynchronize();
  int old = -2;
  hipMemcpy(&old, o, 4, hipMemcpyDeviceToHost);
  printf("system-scope exchange on pinned memory: sync rc %d, old value read %d (want 3), host sees %d (want 7)\n", (int)e, old, h[0]);
  return 0;
}
