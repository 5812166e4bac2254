// host_common.h -- host-side plumbing shared by the translation units of libbore_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <mutex>

#include "mlp_layout.h"

inline thread_local char g_bore_err[512] = "";

// batch mode (include/bore_hip.h: bore_set_batch): the calling thread's current batch, if any
inline thread_local bore_batch g_batch_store;
inline thread_local const bore_batch *g_batch = nullptr;

inline int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_bore_err, sizeof(g_bore_err), fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                      \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) return fail(BORE_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// Raise the kernel's dynamic-LDS limit to `bytes` (default cap is 64 KiB).
// The attribute only ever needs to grow: one runtime call per kernel per new high-water mark
// (hipFuncSetAttribute is a driver round trip -- not something for every launch).
template <typename K>
inline int allow_lds(K kernel, size_t bytes) {
  if (bytes > BORE_LDS_BYTES)
    return fail(BORE_E_UNSUPPORTED, "model needs %zu B of LDS per workgroup (> %d)", bytes,
                BORE_LDS_BYTES);
  // the attribute is per device; callers on different host threads (ctypes releases the GIL)
  // share this table
  struct Seen { const void *k; int dev; size_t bytes; };
  static Seen seen[256];
  static int n_seen = 0;
  static std::mutex mu;
  const void *kp = reinterpret_cast<const void *>(kernel);
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < n_seen; ++i)
    if (seen[i].k == kp && seen[i].dev == dev) {
      if (bytes <= seen[i].bytes) return 0;
      HIP_TRY(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      seen[i].bytes = bytes;
      return 0;
    }
  HIP_TRY(hipFuncSetAttribute(kp, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  if (n_seen < 256) seen[n_seen++] = Seen{kp, dev, bytes};
  return 0;
}

// Compute units of the current device (per device, cached).
inline int device_cus() {
  static int cus[64];
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  std::lock_guard<std::mutex> lock(mu);
  if (!cus[dev]) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 1) n = 256;
    cus[dev] = n;
  }
  return cus[dev];
}

// Does the link between the current device and the host do native atomics (PCIe AtomicOps or a coherent fabric)?
// The fused kernels publish a loop's flag to pinned host memory with a system-scope atomic exchange when it does and
// with a fenced release store when it does not (lbfgsb_body, PUBLISH = 1).  BORE_ASYNC_DEBUG=2 takes the
// store path regardless (tests).  Per device, cached.
inline bool device_host_atomics() {
  static int seen[64];  // 0 unknown, 1 no, 2 yes
  static std::mutex mu;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return false;
  std::lock_guard<std::mutex> lock(mu);
  if (!seen[dev]) {
    int v = 0;
    const bool ok = hipDeviceGetAttribute(&v, hipDeviceAttributeHostNativeAtomicSupported, dev) == hipSuccess && v != 0;
    const char *dbg = getenv("BORE_ASYNC_DEBUG");
    seen[dev] = ok && !(dbg && atoi(dbg) == 2) ? 2 : 1;
    if (dbg) fprintf(stderr, "[bore] device %d: host-native atomics %s -> flags by %s\n", dev, ok ? "yes" : "no",
                     seen[dev] == 2 ? "atomic exchange" : "fenced release store");
  }
  return seen[dev] == 2;
}

// Device scratch for the restart kernels whose optimiser keeps its two 2m x 2m matrices outside the
// LDS (lbfgsb.h: make_work's `big`): BIG_SLOTS workgroup slots of `per_slot` doubles each plus one
// lock word per slot (0 = free).  A workgroup takes a free slot when it starts and frees it when its
// last wave is done, so the slots only have to outnumber the workgroups RESIDENT at once (at most one
// per CU for these LDS-filling kernels; launches on several streams share the pool).  Per device,
// grow-only (a request for larger slots allocates a new pool and leaves the old one to launches that
// may still use it), the first call of a size is a blocking hipMalloc + hipMemset.
constexpr int BORE_BIG_SLOTS = 1024;
struct BigPool {
  double *buf = nullptr;
  int *locks = nullptr;
  size_t per_slot = 0;
};
inline int big_pool(size_t per_slot_doubles, BigPool *out) {
  static BigPool pools[64];
  static std::mutex mu;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return fail(BORE_E_UNSUPPORTED, "big_pool: device index %d", dev);
  std::lock_guard<std::mutex> lock(mu);
  BigPool &p = pools[dev];
  if (p.per_slot < per_slot_doubles) {
    BigPool np;
    np.per_slot = per_slot_doubles;
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&np.buf), (size_t)BORE_BIG_SLOTS * per_slot_doubles * 8));
    HIP_TRY(hipMalloc(reinterpret_cast<void **>(&np.locks), (size_t)BORE_BIG_SLOTS * 4));
    HIP_TRY(hipMemset(np.locks, 0, (size_t)BORE_BIG_SLOTS * 4));
    p = np;
  }
  *out = p;
  return 0;
}

// Builds the layout with the largest tile (max_rows, then 16 rows fewer each try: a wave's
// unit of work is a 16-row block) whose theta + tile + `extra_floats` fit the CU's LDS; the
// tile may shrink only when `may_shrink`.
static const char kBf16Shapes[] =
    "compute = bfloat16 is available for the wide static shapes only (16->64-64-64-1, "
    "32->128-128-1; any activations, no l2)";

inline int check_common(const bore_mlp_desc *desc, int n_models, int with_deltas, int max_rows,
                        bool may_shrink, size_t extra_floats, MlpLayout *L) {
  if (n_models < 1) return fail(BORE_E_INVALID, "n_models must be >= 1 (got %d)", n_models);
  for (int tb = max_rows;; tb -= 16) {
    if (tb < 1 || bore_make_layout(desc, with_deltas, tb, L))
      return fail(BORE_E_INVALID, "bad bore_mlp_desc");
    const size_t need = ((size_t)L->P_lds + L->tile_floats + extra_floats) * 4;
    if (need <= BORE_LDS_BYTES) return 0;
    if (!may_shrink || tb <= 16)
      return fail(BORE_E_UNSUPPORTED,
                  "model needs %zu B of LDS per workgroup (> %d) at %d rows per tile", need,
                  BORE_LDS_BYTES, tb);
  }
}
