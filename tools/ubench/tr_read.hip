// tools/ubench/tr_read.hip -- what ds_read_b64_tr_b16 delivers (gfx950), checked against the rule the
// kernels rely on: per group of 16 consecutive lanes, lane 4q + p supplies the address of row q,
// columns 4p .. 4p + 3 of a 4-row x 16-column block of 16-bit elements; lane i of the group receives
// column i of the block, row q in element q.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench/tr_read.hip -o tools/ubench/tr_read && tools/ubench/tr_read
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 72;  // elements per row (a multiple of 4: 8-byte aligned rows)
__global__ void k(short *out) {
  __shared__ __attribute__((aligned(16))) short lds[64 * PITCH];
  for (int i = threadIdx.x; i < 64 * PITCH; i += 64) lds[i] = (short)((i / PITCH) * 256 + (i % PITCH));
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15;
  // group g reads the block rows 4g .. 4g + 3 (+ 16 for a second block), columns 16g .. 16g + 15
  short *p = lds + (4 * g + (i >> 2)) * PITCH + 16 * g + 4 * (i & 3);
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16 *)p);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = v[e];
}
int main() {
  short *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int e = 0; e < 4; ++e) {
      const int g = lane >> 4, i = lane & 15;
      const int want = (4 * g + e) * 256 + 16 * g + i;  // row 4g + e, column 16g + i
      if (h[lane * 4 + e] != want) {
        if (bad < 8) printf("lane %d element %d: got row %d col %d, want row %d col %d\n", lane, e, h[lane * 4 + e] >> 8,
                            h[lane * 4 + e] & 255, want >> 8, want & 255);
        ++bad;
      }
    }
  printf(bad ? "ds_read_b64_tr_b16: %d MISMATCHES\n" : "ds_read_b64_tr_b16: lane i of a 16-lane group gets column i, element q = row q: OK (%d)\n", bad);
  return bad != 0;
}
