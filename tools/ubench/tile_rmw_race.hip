// Round 4's irreproducible m / v stores, second probe: the access pattern itself.  Eight (or four) waves of a workgroup
// each read-modify-write "tiles" of 256 floats with ONE 16-byte raw buffer load and store per lane (as wide_fused_f32
// does), from a base that is only 4-byte aligned (a model's m / v start at model * P floats, P odd), tiles of
// neighbouring waves sharing cache lines at their boundaries, the NEXT tiles' loads requested AHEAD of a tile's store.
// Afterwards every element is read back with plain loads (after a barrier, as the tile-order conversion at the end of
// the fit does) and compared with old + 1.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/tile_rmw_race.hip -o tools/ubench/tile_rmw_race
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef unsigned u4 __attribute__((ext_vector_type(4)));
template <int NW, int AHEAD>
__global__ void rmw(float *base, int tiles_per_wg, int per_wg_floats, int *bad) {
  float *m = base + (size_t)blockIdx.x * per_wg_floats;  // (per_wg_floats odd: 4-byte aligned bases)
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(m, 0, tiles_per_wg * 1024, 0x00020000);
  constexpr int RING = AHEAD + 2;
  f4u p[RING];
  const int nt = tiles_per_wg / NW;  // tiles of this wave: wv, wv + NW, ..
  auto req = [&](int i) { p[i % RING] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(r, 16 * lane, (wv + NW * i) * 1024, 0)); };
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) if (i < nt) req(i);
#pragma unroll 1
  for (int i0 = 0; i0 < nt; i0 += RING) {
#pragma unroll
    for (int j = 0; j < RING; ++j) {
      const int i = i0 + j;
      if (i >= nt) break;
      if (i + AHEAD < nt) {  // (RING is a compile-time stride: slot indices are constants after unrolling)
        p[(j + AHEAD) % RING] = __builtin_bit_cast(f4u, __builtin_amdgcn_raw_buffer_load_b128(r, 16 * lane, (wv + NW * (i + AHEAD)) * 1024, 0));
      }
      f4u v = p[j];
      for (int k = 0; k < 40; ++k) v = v * 1.0000001f + 0.f;  // (some arithmetic between the load and the store)
      v = p[j] + 1.0f;
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, 16 * lane, (wv + NW * i) * 1024, 0);
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int wrong = 0;
  for (int e = threadIdx.x; e < tiles_per_wg * 256; e += blockDim.x) wrong += m[e] != 2.0f;
  if (wrong) atomicAdd(bad, wrong);
}
template <int NW, int AHEAD>
void run(int tiles) {
  const int WG = 2048, per = tiles * 256 + 1;  // odd stride: the bases walk through every 4-byte misalignment
  float *buf;
  int *bad;
  hipMalloc((void **)&bad, 4);
  long long total_bad = 0;
  for (int trial = 0; trial < 8; ++trial) {
    hipMalloc((void **)&buf, (size_t)WG * per * 4 + 64);  // fresh memory: cold lines
    std::vector<float> old((size_t)WG * per + 16, 1.0f);
    hipMemcpy(buf, old.data(), old.size() * 4, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 4);
    hipDeviceSynchronize();
    hipLaunchKernelGGL((rmw<NW, AHEAD>), dim3(WG), dim3(64 * NW), 0, 0, buf + 1, tiles, per, bad);
    int b = 0;
    hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
    total_bad += b;
    hipFree(buf);
  }
  printf("%d waves, %d tiles ahead, %d tiles per workgroup: %lld wrong elements in %d workgroup runs\n", NW, AHEAD, tiles, total_bad, 8 * WG);
}
int main() {
  run<4, 1>(40); run<4, 3>(40); run<8, 1>(40); run<8, 2>(40); run<8, 3>(40); run<8, 3>(48);
  return 0;
}
