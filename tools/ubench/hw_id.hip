// Where the waves of co-resident workgroups land: HW_REG_HW_ID (SIMD, CU, SE) and HW_REG_XCC_ID of every wave of a
// launch shaped like the fused loop kernel (256 threads, 53 728 B of LDS: two workgroups per CU).
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/hw_id.hip -o /tmp/hw_id && /tmp/hw_id [workgroups] [lds bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ __launch_bounds__(256) void probe(unsigned *out, int spin) {
  extern __shared__ float smem[];
  const int wv = threadIdx.x >> 6;
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  const unsigned lds = __builtin_amdgcn_s_getreg((31 << 11) | 6);   // HW_REG_LDS_ALLOC
  if ((threadIdx.x & 63) == 0) {
    out[(blockIdx.x * 4 + wv) * 2] = hw;
    out[(blockIdx.x * 4 + wv) * 2 + 1] = (xcc & 0xf) | (lds << 4);
  }
  long long t0 = clock64();
  while (clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);  // stay until the whole grid is resident
  if (threadIdx.x == 0) smem[0] = 1.f;
}
int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 512, lds = argc > 2 ? atoi(argv[2]) : 53728;
  unsigned *d;
  hipMalloc(&d, wgs * 8 * sizeof(unsigned));
  hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  probe<<<wgs, 256, lds>>>(d, 2000000);
  std::vector<unsigned> h(wgs * 8);
  hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, std::vector<int>> per_cu;  // (xcc, se, sh, cu) -> workgroups
  int distinct4 = 0, identity = 0;
  for (int b = 0; b < wgs; ++b) {
    unsigned simds = 0;
    bool ident = true;
    for (int w = 0; w < 4; ++w) {
      const unsigned hw = h[(b * 4 + w) * 2];
      simds |= 1u << ((hw >> 4) & 3);
      ident = ident && ((hw >> 4) & 3) == (unsigned)w;
    }
    distinct4 += simds == 0xf;
    identity += ident;
    const unsigned hw = h[b * 8], xcc = h[b * 8 + 1] & 0xf;
    per_cu[(xcc << 16) | (hw & 0xff00)].push_back(b);
  }
  printf("%d workgroups: waves on four distinct SIMDs in %d, wave i on SIMD i in %d; %zu distinct (xcc, se, sh, cu)\n", wgs,
         distinct4, identity, per_cu.size());
  int shown = 0;
  for (auto &kv : per_cu) {
    if (shown++ >= 6) break;
    printf("  cu key %06x:", kv.first);
    for (int b : kv.second) {
      printf("  wg %d lds_alloc %07x simds", b, h[b * 8 + 1] >> 4);
      for (int w = 0; w < 4; ++w) printf(" %u", (h[(b * 4 + w) * 2] >> 4) & 3);
    }
    printf("\n");
  }
  std::map<size_t, int> hist;
  for (auto &kv : per_cu) hist[kv.second.size()]++;
  for (auto &kv : hist) printf("  CUs with %zu workgroups: %d\n", kv.first, kv.second);
  // same-SIMD collisions of the waves that own rows in the fit (waves 0, 1) between co-resident workgroups
  int pairs = 0, collide = 0;
  for (auto &kv : per_cu)
    for (size_t i = 0; i < kv.second.size(); ++i)
      for (size_t j = i + 1; j < kv.second.size(); ++j) {
        ++pairs;
        unsigned a = 0, b = 0;
        for (int w = 0; w < 2; ++w) {
          a |= 1u << ((h[(kv.second[i] * 4 + w) * 2] >> 4) & 3);
          b |= 1u << ((h[(kv.second[j] * 4 + w) * 2] >> 4) & 3);
        }
        collide += (a & b) != 0;
      }
  printf("  co-resident pairs %d, of which waves 0/1 share a SIMD: %d\n", pairs, collide);
  return 0;
}
