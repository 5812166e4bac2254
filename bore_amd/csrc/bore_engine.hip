// bore_engine.hip -- native replica engine: many independent BO loops on one GPU
// (BASELINE.json config 4).  Host-side C++ over the library's own C-ABI.
//
// One BO iteration of a loop is  label -> fit -> sample + screen -> L-BFGS-B restarts -> pick
// (README.rst:83-103; bore/plugins/hpbandster/base.py:216-262) and only the objective value of
// the suggestion comes from the host.  The loops of a GPU are split into GROUPS, each stepping on
// its own HIP stream; per iteration a group costs
//     H2D  (x_new, y_new)  [Lg][D+1] fp64        -- the record itself lives on the device
//     6 launches: append, labels, fit, sample + screen, lbfgsb, select
//     D2H  x_best [Lg][D], best [Lg], info [Lg][R][5]
// and one host turnaround (objective callback).  What this file replaces is the Python statement
// of the same loop (bore_amd/engine.py ReplicaEngine, kept as its check: trajectories are
// bit-identical, tested): per group and iteration the interpreter spent ~260 us between the
// result's arrival and the next fit's launch -- on the critical path of a ~2.3 ms iteration.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include <sched.h>

#include "../../include/bore_hip.h"
#include "host_common.h"

namespace {

// numpy.random.RandomState (MT19937) continued from an exported state: the fallback point of a
// loop whose argmax returned None comes from that loop's own stream (engine.py: rs[l].uniform).
struct Mt19937 {
  uint32_t key[624];
  int pos;
  uint32_t next32() {
    if (pos >= 624) {
      int i;
      for (i = 0; i < 624 - 397; ++i) {
        const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
        key[i] = key[i + 397] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
      }
      for (; i < 623; ++i) {
        const uint32_t y = (key[i] & 0x80000000u) | (key[i + 1] & 0x7fffffffu);
        key[i] = key[i + (397 - 624)] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
      }
      const uint32_t y = (key[623] & 0x80000000u) | (key[0] & 0x7fffffffu);
      key[623] = key[396] ^ (y >> 1) ^ (-(int32_t)(y & 1) & 0x9908b0dfu);
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
  }
  double next_double() {  // numpy's random_sample: 53 bits from two draws
    const uint32_t a = next32() >> 5, b = next32() >> 6;
    return (a * 67108864.0 + b) / 9007199254740992.0;
  }
};

struct Group {
  int a = 0, b = 0;  // loops [a, b) of the engine
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr, ev[4] = {nullptr, nullptr, nullptr, nullptr};  // fit / lbfgsb brackets
  int64_t n = 0, cap = 0;          // rows in the device record / its capacity
  int64_t epochs_seen = 0, draws = 0, steps = 0;
  bool inflight = false, has_new = false;
  // device
  double *X_seen = nullptr, *y_seen = nullptr, *y_dense = nullptr;
  float *X32 = nullptr, *z = nullptr;
  double *new_x = nullptr, *new_y = nullptr;
  double *x0 = nullptr, *x = nullptr, *jac = nullptr, *fun = nullptr, *x_best = nullptr;
  int32_t *idx = nullptr, *info = nullptr, *best = nullptr;
  // pinned host
  double *new_x_pin = nullptr, *new_y_pin = nullptr, *x_best_pin = nullptr;
  int32_t *best_pin = nullptr, *info_pin = nullptr;
};

// ---- asynchronous mode: loops advance individually (include/bore_hip.h: bore_batch) --------------
// A WORKER is a stream that runs one launch chain (append, labels, fit, sample + screen, restarts
// + pick) over whatever loops are ready when it becomes free.  A loop's result reaches the host
// through its flag the moment ITS restarts are done -- the chain may still be busy with slower
// loops -- so the loop joins the next free worker's batch instead of waiting for them.
struct Worker {
  hipStream_t stream = nullptr;
  hipEvent_t done = nullptr, ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool busy = false;
  int n = 0;  // slots of the launch in flight
  // staging: ids [L] | its [L] (int32) then x_new [L][D] | y_new [L] (fp64), host and device
  int32_t *h_int = nullptr, *d_int = nullptr;
  double *h_dbl = nullptr, *d_dbl = nullptr;
  // per-slot outputs of the launch (device)
  double *x0 = nullptr, *x = nullptr, *jac = nullptr, *fun = nullptr;
  int32_t *idx = nullptr, *info = nullptr;
  double fit_bytes = 0;
  IterArgs *h_args = nullptr, *d_args = nullptr;  // fused iteration kernel: pinned / device copy
  size_t stage_bytes = 0;                          // (start of the worker's staging block)
};

struct Async {
  int64_t cap = 0;
  double *X_seen = nullptr, *y_seen = nullptr;   // device [L][cap][D], [L][cap]
  float *X32 = nullptr, *z = nullptr;            // device [L][cap][D], [L][cap]
  double *result = nullptr;                      // pinned [L][D + 8]
  int32_t *flag = nullptr;                       // pinned [L]
  int64_t *stamps = nullptr;                     // device [L][4]: phase clock stamps (fused kernel)
  double ns_per_tick = 10.0;                     // of the device wall clock (100 MHz on gfx950)
  std::vector<int32_t> it, target, state;        // per loop: iterations done; 0 ready 1 in flight 2 done
  std::vector<double> x_new, y_new, ready_since; // per loop: the row to append at its next launch
  std::vector<double> seen_at;                   // per loop: when the host took its result
  // residency (fused kernel; include/bore_hip.h bore_batch): one pinned block, host <-> device
  double *ynew = nullptr;                        // pinned [L][D + 1]: the newest row of each loop
  int32_t *yseq = nullptr, *parked = nullptr;    // pinned [L]
  int32_t *abort_flag = nullptr;                 // pinned [1]
  int64_t wait_ticks = 0;                        // how long a workgroup waits for its next row
  std::vector<uint8_t> via_launch;               // per loop: the iteration in flight came with a launch
  std::vector<Worker> workers;
  std::vector<int> scratch_ids, done_ids;
  std::vector<double> cb_x, cb_y;
  bool fused = false;  // one kernel per launch (bore_iter.hip) instead of the five-launch chain
  // work queue (bore_iter.hip: queue_kernel) -- the schedule for more loops than the device holds
  // workgroups: ring of entries (pinned), ticket counter (device), entries written so far, identity index
  bool queue = false;
  QueueEntry *q_ring = nullptr;
  unsigned long long *q_head = nullptr, q_tail = 0;  // q_head: device, 2 words: ticket counter | copy of the tail
  unsigned long long *q_tail_host = nullptr;         // pinned: entries published so far (all ones: exit)
  int q_mask = 0, q_wgs = 0;
  int per_cu = 0;  // loops of this model one CU holds (async_decide_schedule)
  int host_threads = 1;  // of the last work-queue run
  int32_t *d_ident = nullptr;
  std::vector<int> q_owner;  // per ring slot: (loop, iteration) of the entry written there last
  long long n_resident = 0, n_parked = 0;
  int stream_concurrency = 0;  // worker streams the device ran at once in the creation probe
  // diagnostics (BORE_ASYNC_DEBUG): waits between a loop's states
  std::vector<double> launched_at;
  std::vector<double> loop_acc;  // per loop: fit ns, restart ns, evaluations, rounds, launch->result s, iterations
  double sum_wait = 0, sum_flight = 0;
  long long n_wait = 0, n_batches = 0, n_slots = 0;
};

}  // namespace

// Busy-waits `ticks` of the device wall clock: the probe of async_create.
__global__ void bore_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

extern "C" int bore_objective_branin01(const double *x, int64_t n, int32_t D, double *y, void *) {
  if (D != 2 || !x || !y) return 1;
  // (bore_amd/engine.py: branin01 -- numpy evaluates the same expression tree)
  const double pi = 3.141592653589793;
  const double a = 5.1 / (4 * (pi * pi)), b = 5 / pi, c = 10 * (1 - 1 / (8 * pi));
  for (int64_t i = 0; i < n; ++i) {
    const double x1 = 15.0 * x[2 * i] - 5.0, x2 = 15.0 * x[2 * i + 1];
    const double t = x2 - a * (x1 * x1) + b * x1 - 6;
    y[i] = t * t + c * std::cos(x1) + 10;
  }
  return 0;
}

struct bore_engine {
  bore_mlp_desc desc;
  bore_engine_cfg cfg;
  int D = 0, P = 0;
  std::vector<double> low, high;
  bore_objective_fn objective = nullptr;
  void *user = nullptr;
  float *theta = nullptr, *adam_m = nullptr, *adam_v = nullptr;
  int64_t *adam_t = nullptr;
  std::vector<Mt19937> rs;
  std::vector<Group> groups;
  bore_engine_stats st;
  std::vector<double> y_tmp;
  Async *as = nullptr;  // asynchronous mode (cfg.async_loops)
  // set when a run stopped half-way (objective error, watchdog, HIP error): loops are then at
  // different iteration counts and some device-side updates have no host-side record, so the
  // engine refuses further use instead of silently diverging
  bool poisoned = false;
};

namespace {

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

template <typename T>
int dev_alloc(T **p, size_t count) {
  HIP_TRY(hipMalloc(reinterpret_cast<void **>(p), count * sizeof(T)));
  return 0;
}
template <typename T>
int pin_alloc(T **p, size_t count) {
  HIP_TRY(hipHostMalloc(reinterpret_cast<void **>(p), count * sizeof(T), hipHostMallocDefault));
  return 0;
}

// (Re)allocate the record of group g for `cap` rows per loop, keeping its rows.
int alloc_record(bore_engine *e, Group &g, int64_t cap) {
  const size_t Lg = g.b - g.a, D = e->D;
  double *X = nullptr, *y = nullptr, *yd = nullptr;
  float *X32 = nullptr, *z = nullptr;
  int rc;
  if ((rc = dev_alloc(&X, Lg * cap * D)) || (rc = dev_alloc(&y, Lg * cap)) ||
      (rc = dev_alloc(&yd, Lg * cap)) || (rc = dev_alloc(&X32, Lg * cap * D)) ||
      (rc = dev_alloc(&z, Lg * cap)))
    return rc;
  if (g.X_seen) {
    HIP_TRY(hipStreamSynchronize(g.stream));
    HIP_TRY(hipMemcpy2D(X, cap * D * 8, g.X_seen, g.cap * D * 8, g.n * D * 8, Lg,
                        hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy2D(y, cap * 8, g.y_seen, g.cap * 8, g.n * 8, Lg, hipMemcpyDeviceToDevice));
    (void)hipFree(g.X_seen); (void)hipFree(g.y_seen); (void)hipFree(g.y_dense);
    (void)hipFree(g.X32); (void)hipFree(g.z);
  }
  g.X_seen = X; g.y_seen = y; g.y_dense = yd; g.X32 = X32; g.z = z;
  g.cap = cap;
  return 0;
}

// Queue one BO iteration of group g on its stream.
int enqueue(bore_engine *e, Group &g) {
  const double t0 = now_s();
  const int Lg = g.b - g.a, D = e->D, R = e->cfg.num_starts;
  const bore_engine_cfg &c = e->cfg;
  int rc;
  if (g.has_new && g.n + 1 > g.cap && (rc = alloc_record(e, g, 2 * g.cap))) return rc;
  void *sp = g.stream;
  if (g.has_new)  // new_x | new_y are one block on both sides: one copy
    HIP_TRY(hipMemcpyAsync(g.new_x, g.new_x_pin, (size_t)Lg * (D + 1) * 8, hipMemcpyHostToDevice,
                           g.stream));
  if ((rc = bore_append_observations(Lg, D, g.X_seen, g.y_seen, g.n, g.cap,
                                     g.has_new ? g.new_x : nullptr, g.has_new ? g.new_y : nullptr,
                                     g.X32, g.y_dense, sp)))
    return rc;
  if (g.has_new) ++g.n;
  g.has_new = false;
  const int64_t N = g.n;
  if ((rc = bore_labels(Lg, g.y_dense, N, c.gamma, g.z, nullptr, sp))) return rc;
  float *th = e->theta + (size_t)g.a * e->P, *m = e->adam_m + (size_t)g.a * e->P,
        *v = e->adam_v + (size_t)g.a * e->P;
  HIP_TRY(hipEventRecord(g.ev[0], g.stream));
  if ((rc = bore_mlp_fit(&e->desc, Lg, th, m, v, e->adam_t + g.a, g.X32, g.z, N, c.epochs,
                         c.batch_size, nullptr, c.seed, c.loop_id0 + g.a, g.epochs_seen, &c.adam,
                         nullptr, sp)))
    return rc;
  HIP_TRY(hipEventRecord(g.ev[1], g.stream));
  const int64_t steps = (N + c.batch_size - 1) / c.batch_size;
  e->st.fit_bytes += (double)Lg * c.epochs * (4.0 * N * (D + 1) + steps * 24.0 * e->P);
  g.epochs_seen += c.epochs;
  if ((rc = bore_sample_screen_topk(&e->desc, Lg, th, c.seed, c.loop_id0 + g.a, g.draws, c.num_samples,
                                    e->low.data(), e->high.data(), R, g.x0, g.idx, nullptr, sp)))
    return rc;
  ++g.draws;
  HIP_TRY(hipEventRecord(g.ev[2], g.stream));
  if ((rc = bore_lbfgsb_minimize(&e->desc, Lg, th, c.transform, 1, g.x0, R, e->low.data(),
                                 e->high.data(), &c.lbfgsb, g.x, g.fun, g.jac, g.info, sp)))
    return rc;
  HIP_TRY(hipEventRecord(g.ev[3], g.stream));
  if ((rc = bore_select_best(Lg, R, D, g.x, g.fun, g.info, c.deduplicate ? g.X_seen : nullptr, g.n,
                             g.cap, 1e-5, 1e-8, g.x_best, g.best, sp)))
    return rc;
  // x_best | info | best are one block on both sides: one copy
  HIP_TRY(hipMemcpyAsync(g.x_best_pin, g.x_best, (size_t)Lg * (D * 8 + R * 5 * 4 + 4),
                         hipMemcpyDeviceToHost, g.stream));
  HIP_TRY(hipEventRecord(g.done, g.stream));
  g.inflight = true;
  e->st.host_enqueue_s += now_s() - t0;
  return 0;
}

// Host side of a finished iteration: fallback points, objective, stage the new row.
int finalize(bore_engine *e, Group &g) {
  const double t0 = now_s();
  const int Lg = g.b - g.a, D = e->D, R = e->cfg.num_starts;
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, g.ev[0], g.ev[1]));
  e->st.fit_ms += ms;
  e->st.fit_launches += 1;
  HIP_TRY(hipEventElapsedTime(&ms, g.ev[2], g.ev[3]));
  e->st.argmax_ms += ms;
  e->st.argmax_launches += 1;
  // every f/g row reads x and writes value + gradient; every round of a loop
  // re-reads its theta (streaming model)
  int64_t rows = 0, rounds_sum = 0, rounds_max = 0;
  for (int l = 0; l < Lg; ++l) {
    int64_t mx = 0;
    for (int r = 0; r < R; ++r) {
      const int64_t nfev = g.info_pin[((size_t)l * R + r) * 5 + 1];
      rows += nfev;
      mx = nfev > mx ? nfev : mx;
    }
    rounds_sum += mx;
    rounds_max = mx > rounds_max ? mx : rounds_max;
  }
  e->st.n_fg_rows += rows;
  e->st.n_fg_requests += rows;
  e->st.n_rounds += rounds_max;
  e->st.argmax_bytes += rows * 4.0 * (2 * D + 1) + rounds_sum * 4.0 * e->P;
  for (int l = 0; l < Lg; ++l) {
    double *xn = g.new_x_pin + (size_t)l * D;
    if (g.best_pin[l] < 0) {  // reference: fall back to a random point of this loop's stream
      ++e->st.none_results;
      Mt19937 &rs = e->rs[g.a + l];
      for (int d = 0; d < D; ++d) xn[d] = e->low[d] + (e->high[d] - e->low[d]) * rs.next_double();
    } else {
      std::memcpy(xn, g.x_best_pin + (size_t)l * D, (size_t)D * 8);
    }
  }
  g.inflight = false;
  if (e->objective(g.new_x_pin, Lg, D, g.new_y_pin, e->user))
    return fail(BORE_E_CALLBACK, "engine_run: the objective callback failed");
  g.has_new = true;
  ++g.steps;
  e->st.host_finalize_s += now_s() - t0;
  return 0;
}

void free_group(Group &g) {
  void *dev[] = {g.X_seen, g.y_seen, g.y_dense, g.X32, g.z, g.new_x, g.x0,
                 g.x, g.jac, g.fun, g.x_best, g.idx};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  void *pin[] = {g.new_x_pin, g.x_best_pin};
  for (void *p : pin)
    if (p) (void)hipHostFree(p);
  if (g.done) (void)hipEventDestroy(g.done);
  for (hipEvent_t ev : g.ev)
    if (ev) (void)hipEventDestroy(ev);
  if (g.stream) (void)hipStreamDestroy(g.stream);
}

// (Re)allocate the per-loop record buffers of the asynchronous mode for `cap` rows.
int async_alloc(bore_engine *e, int64_t cap) {
  Async &A = *e->as;
  const size_t L = e->cfg.n_loops, D = e->D;
  double *X = nullptr, *y = nullptr;
  float *X32 = nullptr, *z = nullptr;
  int rc;
  if ((rc = dev_alloc(&X, L * cap * D)) || (rc = dev_alloc(&y, L * cap)) ||
      (rc = dev_alloc(&X32, L * cap * D)) || (rc = dev_alloc(&z, L * cap)))
    return rc;
  if (A.X_seen) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy2D(X, cap * D * 8, A.X_seen, A.cap * D * 8, A.cap * D * 8, L, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy2D(y, cap * 8, A.y_seen, A.cap * 8, A.cap * 8, L, hipMemcpyDeviceToDevice));
    HIP_TRY(hipMemcpy2D(X32, cap * D * 4, A.X32, A.cap * D * 4, A.cap * D * 4, L, hipMemcpyDeviceToDevice));
    (void)hipFree(A.X_seen); (void)hipFree(A.y_seen); (void)hipFree(A.X32); (void)hipFree(A.z);
  }
  A.X_seen = X; A.y_seen = y; A.X32 = X32; A.z = z;
  A.cap = cap;
  return 0;
}

// How many loops of this engine's model one CU holds (the fused kernels' registers and the LDS their phases need at
// the current record capacity: iteration_launch asked without launching), and from that the schedule: all loops
// resident, or -- more loops than resident workgroups -- the work queue.  bore_engine_cfg::work_queue = 0 / 1 forces the
// launch-per-batch schedule of round 3 / the queue (tests, A/B).  Called when the engine is created and whenever
// the records have grown (a longer data set in LDS may cost a loop per CU).
static int async_decide_schedule(bore_engine *e) {
  Async &A = *e->as;
  const bore_engine_cfg &c = e->cfg;
  const int L = c.n_loops, D = e->D, R = c.num_starts;
  A.queue = false;
  A.q_wgs = 0;
  if (!A.fused) return 0;
  Worker &w = A.workers[0];
  int per_cu = 0;
  {
    bore_batch bt;
    std::memset(&bt, 0, sizeof(bt));
    bt.ids = w.d_int; bt.its = w.d_int; bt.n_init = c.n_init; bt.deduplicate = c.deduplicate;
    bt.cap = A.cap; bt.X_seen = A.X_seen; bt.result = A.result; bt.flag = A.flag; bt.stamps = A.stamps;
    bore_set_batch(&bt);
    IterArgs probe;
    const int rc = iteration_launch(&e->desc, L, e->theta, e->adam_m, e->adam_v, e->adam_t, A.X_seen, A.y_seen, A.X32,
                                    A.z, w.d_dbl, w.d_dbl + (size_t)L * D, c.gamma, c.epochs, c.batch_size, c.seed,
                                    c.loop_id0, &c.adam, c.num_samples, e->low.data(), e->high.data(), R, c.transform,
                                    &c.lbfgsb, w.x0, w.idx, w.x, w.fun, w.jac, w.info, &probe, w.d_args, w.stage_bytes,
                                    w.stream, 0, nullptr, nullptr, 0, &per_cu);
    bore_set_batch(nullptr);
    if (rc) return rc;
  }
  if (per_cu < 1) return fail(BORE_E_UNSUPPORTED, "engine: the fused kernel does not fit a compute unit with records of %lld rows", (long long)A.cap);
  A.per_cu = per_cu;
  const int forced = c.work_queue;
  const int resident_cap = per_cu * device_cus();
  A.queue = forced < 0 ? L > resident_cap : forced != 0;
  if (A.queue) {
    A.q_wgs = L < resident_cap ? L : resident_cap;
    int capq = 1;
#ifdef BORE_QUEUE_STALL_TEST
    while (capq < L + 64) capq <<= 1;   // (test build: the smallest ring -- every slot comes round every ~L tickets)
#else
    while (capq < 2 * (L + A.q_wgs) + 64) capq <<= 1;   // (outstanding entries <= loops + exit tokens)
#endif
    if (capq - 1 > A.q_mask || !A.q_ring) {
      if (A.q_ring) (void)hipHostFree(A.q_ring);
      A.q_ring = nullptr;
      int rc;
      if ((rc = pin_alloc(&A.q_ring, (size_t)capq))) return rc;
      A.q_mask = capq - 1;
    }
    std::memset(A.q_ring, 0, ((size_t)A.q_mask + 1) * sizeof(QueueEntry));
    if (!A.q_head) {
      int rc;
      if ((rc = dev_alloc(&A.q_head, 2)) || (rc = pin_alloc(&A.q_tail_host, 1)) || (rc = dev_alloc(&A.d_ident, (size_t)L)))
        return rc;
      *A.q_tail_host = 0;
      std::vector<int32_t> ident((size_t)L);
      for (int l = 0; l < L; ++l) ident[(size_t)l] = (int32_t)l;
      HIP_TRY(hipMemcpy(A.d_ident, ident.data(), (size_t)L * 4, hipMemcpyHostToDevice));
    }
  }
  if (getenv("BORE_ASYNC_DEBUG"))
    fprintf(stderr, "[bore] engine: %d loops, records of %lld rows: %d loops per CU -> %s\n", L, (long long)A.cap, per_cu,
            A.queue ? "work queue" : "all resident");
  return 0;
}

int async_create(bore_engine *e, const double *X0, const double *y0) {
  const size_t L = e->cfg.n_loops, D = e->D, R = e->cfg.num_starts, n0 = e->cfg.n_init;
  if (R > 16) return fail(BORE_E_UNSUPPORTED, "engine_create: async_loops needs num_starts <= 16");
  e->as = new (std::nothrow) Async();
  if (!e->as) return fail(BORE_E_HIP, "engine_create: out of memory");
  Async &A = *e->as;
  int rc;
  // (records grow by doubling, async_run; the fit keeps a loop's data set in LDS, sized by this capacity: with 128
  // rows three loops of the 2->16-16-1 model share a CU, with 256 two)
  if ((rc = async_alloc(e, n0 * 2 > 128 ? n0 * 2 : 128))) return rc;
  std::vector<float> x32(L * n0 * D);
  for (size_t i = 0; i < x32.size(); ++i) x32[i] = (float)X0[i];
  HIP_TRY(hipMemcpy2D(A.X_seen, A.cap * D * 8, X0, n0 * D * 8, n0 * D * 8, L, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy2D(A.y_seen, A.cap * 8, y0, n0 * 8, n0 * 8, L, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy2D(A.X32, A.cap * D * 4, x32.data(), n0 * D * 4, n0 * D * 4, L, hipMemcpyHostToDevice));
  if ((rc = pin_alloc(&A.result, L * (D + 8))) || (rc = pin_alloc(&A.flag, L)) ||
      (rc = dev_alloc(&A.stamps, L * 4)))
    return rc;
  std::memset(A.flag, 0, L * 4);
  HIP_TRY(hipMemset(A.stamps, 0, L * 4 * 8));
  {
    int dev = 0, khz = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, dev) == hipSuccess && khz > 0)
      A.ns_per_tick = 1e6 / khz;
  }
  A.seen_at.assign(L, 0.0);
  A.via_launch.assign(L, 1);
  {  // ynew [L][D + 1] fp64 | yseq [L] | parked [L] | abort [1]  (int32), one pinned block
    char *blk = nullptr;
    const size_t bytes = L * (D + 1) * 8 + (2 * L + 1) * 4;
    if ((rc = pin_alloc(&blk, bytes))) return rc;
    std::memset(blk, 0, bytes);
    A.ynew = reinterpret_cast<double *>(blk);
    A.yseq = reinterpret_cast<int32_t *>(blk + L * (D + 1) * 8);
    A.parked = A.yseq + L;
    A.abort_flag = A.parked + L;
    for (size_t l = 0; l < L; ++l) A.parked[l] = -1;
  }
  A.it.assign(L, 0); A.target.assign(L, 0); A.state.assign(L, 2);
  A.x_new.assign(L * D, 0.0); A.y_new.assign(L, 0.0); A.ready_since.assign(L, 0.0);
  A.scratch_ids.reserve(L); A.cb_x.resize(L * D); A.cb_y.resize(L);
  A.launched_at.assign(L, 0.0);
  A.done_ids.assign(L, 0);
  A.fused = iteration_supported(&e->desc);
  // Residency: a loop's workgroup stays on its CU between iterations and waits this long for the
  // objective value before it gives its slot up (bore_engine_cfg::resident_wait_us; 0 = one iteration per
  // launch).  Dropped by the launcher when the device cannot hold all loops at once.
  const double resident_us = e->cfg.resident_wait_us < 0 ? 2000.0 : (double)e->cfg.resident_wait_us;
  A.wait_ticks = A.fused && resident_us > 0 ? (int64_t)(resident_us * 1e3 / A.ns_per_tick) : 0;
  // Worker streams: a dozen independent single-kernel launches in flight when fused (each stream
  // needs its own hardware queue -- streams sharing one serialise, which halves the throughput --
  // so no more than GPU_MAX_HW_QUEUES - 2 of them); four dependent launch chains otherwise.
  int n_workers = A.fused ? 12 : 4;
  if (A.fused) {
    const int queues = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
    if (n_workers > queues - 2) n_workers = queues - 2 > 2 ? queues - 2 : 2;
  }
  if (e->cfg.worker_streams > 0) n_workers = e->cfg.worker_streams;
  A.workers.resize(n_workers > 0 ? n_workers : 1);
  for (Worker &w : A.workers) {
    HIP_TRY(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    for (hipEvent_t &ev : w.ev) HIP_TRY(hipEventCreate(&ev));
    // ONE staging block per side -- IterArgs | x_new [L][D] | y_new [L] | ids [L] | its [L] |
    // targets [L] -- so that a launch is preceded by one upload (each small copy is its own ~30 us
    // blit packet)
    w.stage_bytes = ((sizeof(IterArgs) + 15) & ~(size_t)15) + L * (D + 1) * 8 + 3 * L * 4;
    char *hb = nullptr, *db = nullptr;
    if ((rc = pin_alloc(&hb, w.stage_bytes)) || (rc = dev_alloc(&db, w.stage_bytes))) return rc;
    const size_t o_dbl = (sizeof(IterArgs) + 15) & ~(size_t)15, o_int = o_dbl + L * (D + 1) * 8;
    w.h_args = reinterpret_cast<IterArgs *>(hb); w.d_args = reinterpret_cast<IterArgs *>(db);
    w.h_dbl = reinterpret_cast<double *>(hb + o_dbl); w.d_dbl = reinterpret_cast<double *>(db + o_dbl);
    w.h_int = reinterpret_cast<int32_t *>(hb + o_int); w.d_int = reinterpret_cast<int32_t *>(db + o_int);
    if ((rc = dev_alloc(&w.x0, L * R * D)) || (rc = dev_alloc(&w.x, L * R * D)) ||
        (rc = dev_alloc(&w.jac, L * R * D)) || (rc = dev_alloc(&w.fun, L * R)) ||
        (rc = dev_alloc(&w.idx, L * R)) || (rc = dev_alloc(&w.info, L * R * 5)))
      return rc;
  }
  // Do the worker streams really run side by side?  ROCm multiplexes streams onto
  // GPU_MAX_HW_QUEUES hardware queues, fixed when the runtime initialised (default 4): a caller
  // that touched the GPU before that variable was set gets fewer queues than streams, and streams
  // sharing a queue serialise -- half the throughput of the non-resident schedule, silently.
  // Probe: one 1 ms spinning single-wave kernel per stream, all at once.
  if (A.workers.size() > 1) {
    const long long ticks = (long long)(2e6 / A.ns_per_tick);  // 2 ms
    hipLaunchKernelGGL(bore_spin_kernel, dim3(1), dim3(64), 0, A.workers[0].stream, 1000LL);  // (code load)
    HIP_TRY(hipStreamSynchronize(A.workers[0].stream));
    double t0 = now_s();
    hipLaunchKernelGGL(bore_spin_kernel, dim3(1), dim3(64), 0, A.workers[0].stream, ticks);
    HIP_TRY(hipStreamSynchronize(A.workers[0].stream));
    const double one = now_s() - t0;  // one probe alone (launch + 2 ms + sync)
    t0 = now_s();
    for (Worker &w : A.workers) hipLaunchKernelGGL(bore_spin_kernel, dim3(1), dim3(64), 0, w.stream, ticks);
    for (Worker &w : A.workers) HIP_TRY(hipStreamSynchronize(w.stream));
    const double dt = now_s() - t0;
    // all side by side: about `one` plus a launch each; k streams per queue: about k x `one`
    int conc = (int)A.workers.size();
    if (dt > 1.5 * one) conc = (int)(A.workers.size() * one / dt + 0.5);
    conc = conc < 1 ? 1 : (conc > (int)A.workers.size() ? (int)A.workers.size() : conc);
    A.stream_concurrency = conc;
    // What the probe measured, once, for whoever reads the log.  The advice to raise GPU_MAX_HW_QUEUES is
    // only given when the variable really is below what the streams need; with it at 16 the MI355X boxes
    // of this pool still run ~6 single-wave probes at once (12 probes of 2 ms: 4.2 ms) -- the streams have
    // their queues, the device's dispatchers interleave no more of them.  Root cause not established;
    // the non-resident schedule (more loops than the device holds) is sized for what was measured.
    if (conc < (int)A.workers.size()) {
      const int queues = getenv("GPU_MAX_HW_QUEUES") ? atoi(getenv("GPU_MAX_HW_QUEUES")) : 4;
      if (queues < (int)A.workers.size() + 2)
        fprintf(stderr,
                "bore_engine: %zu worker streams, but the device ran only about %d of them at once (one 2-ms probe "
                "alone %.2f ms, %zu at once %.2f ms): streams share hardware queues.  Set GPU_MAX_HW_QUEUES >= %zu in "
                "the environment BEFORE the first GPU call of the process (it is %d now).\n",
                A.workers.size(), conc, 1e3 * one, A.workers.size(), 1e3 * dt, A.workers.size() + 2, queues);
      else if (getenv("BORE_ASYNC_DEBUG"))
        fprintf(stderr,
                "bore_engine: stream probe: one 2-ms probe alone %.2f ms, %zu at once %.2f ms -> about %d run side by "
                "side (GPU_MAX_HW_QUEUES = %d already covers the %zu streams: not a queue shortage)\n",
                1e3 * one, A.workers.size(), 1e3 * dt, conc, queues, A.workers.size());
    }
  }
  return async_decide_schedule(e);
}

void async_destroy(bore_engine *e) {
  if (!e->as) return;
  Async &A = *e->as;
  for (Worker &w : A.workers) {
    void *dev[] = {w.d_args, w.x0, w.x, w.jac, w.fun, w.idx, w.info};
    if (w.h_args) (void)hipHostFree(w.h_args);
    for (void *p : dev)
      if (p) (void)hipFree(p);
    if (w.done) (void)hipEventDestroy(w.done);
    for (hipEvent_t ev : w.ev)
      if (ev) (void)hipEventDestroy(ev);
    if (w.stream) (void)hipStreamDestroy(w.stream);
  }
  void *dev[] = {A.X_seen, A.y_seen, A.X32, A.z, A.stamps};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  if (A.result) (void)hipHostFree(A.result);
  if (A.flag) (void)hipHostFree(A.flag);
  if (A.ynew) (void)hipHostFree(A.ynew);
  if (A.q_ring) (void)hipHostFree(A.q_ring);
  if (A.q_tail_host) (void)hipHostFree(A.q_tail_host);
  if (A.q_head) (void)hipFree(A.q_head);
  if (A.d_ident) (void)hipFree(A.d_ident);
  delete e->as;
  e->as = nullptr;
}

// One launch of worker w over the loops listed in A.scratch_ids (fused: one kernel whose
// workgroups may go on to later iterations of their loops; otherwise the five-launch chain).
int async_launch(bore_engine *e, Worker &w) {
  const double t0 = now_s();
  Async &A = *e->as;
  const bore_engine_cfg &c = e->cfg;
  const int L = c.n_loops, D = e->D, R = c.num_starts, B = (int)A.scratch_ids.size();
  int max_it = 0;
  double *hx = w.h_dbl, *hy = w.h_dbl + (size_t)L * D;
  for (int b = 0; b < B; ++b) {
    const int l = A.scratch_ids[b], it = A.it[l];
    w.h_int[b] = l;
    w.h_int[L + b] = it;
    w.h_int[2 * L + b] = A.target[l];
    std::memcpy(hx + (size_t)b * D, &A.x_new[(size_t)l * D], (size_t)D * 8);
    hy[b] = A.y_new[l];
    max_it = it > max_it ? it : max_it;
    A.state[l] = 1;
    A.via_launch[l] = 1;
    A.launched_at[l] = t0;
    A.sum_wait += t0 - A.ready_since[l];
    e->st.ready_to_launch_s += t0 - A.ready_since[l];
    ++A.n_wait;
    // (the previous workgroup of this loop, if it parked, wrote `parked` as its last act)
    __atomic_store_n(&A.parked[l], -1, __ATOMIC_RELAXED);
    __atomic_store_n(&A.yseq[l], it, __ATOMIC_RELEASE);
  }
  if (!A.fused)  // (the fused launch uploads the whole block together with its arguments)
    HIP_TRY(hipMemcpyAsync(w.d_dbl, w.h_dbl, w.stage_bytes - ((sizeof(IterArgs) + 15) & ~(size_t)15),
                           hipMemcpyHostToDevice, w.stream));
  const int64_t N_max = c.n_init + max_it;
  bore_batch bt;
  std::memset(&bt, 0, sizeof(bt));
  bt.ids = w.d_int; bt.its = w.d_int + L; bt.n_init = c.n_init; bt.deduplicate = c.deduplicate;
  bt.cap = A.cap; bt.X_seen = A.X_seen; bt.result = A.result; bt.flag = A.flag;
  bt.stamps = A.stamps;
  if (A.fused) {
    bt.targets = w.d_int + 2 * L;
    bt.ynew = A.ynew; bt.yseq = A.yseq; bt.parked = A.parked; bt.abort_flag = A.abort_flag;
    bt.wait_ticks = A.wait_ticks;
    bt.resident_loops = L;
  }
  bore_set_batch(&bt);
  void *sp = w.stream;
  if (A.fused) {
    int rc = hipEventRecord(w.ev[2], w.stream) == hipSuccess ? 0 : fail(BORE_E_HIP, "hipEventRecord");
    if (!rc)
      rc = iteration_launch(&e->desc, B, e->theta, e->adam_m, e->adam_v, e->adam_t, A.X_seen, A.y_seen,
                            A.X32, A.z, w.d_dbl, w.d_dbl + (size_t)L * D, c.gamma, c.epochs,
                            c.batch_size, c.seed, c.loop_id0, &c.adam, c.num_samples, e->low.data(),
                            e->high.data(), R, c.transform, &c.lbfgsb, w.x0, w.idx, w.x, w.fun,
                            w.jac, w.info, w.h_args, w.d_args, w.stage_bytes, sp);
    if (!rc && hipEventRecord(w.ev[3], w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
    if (!rc && hipEventRecord(w.done, w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
    bore_set_batch(nullptr);
    if (rc) return rc;
    w.busy = true;
    w.n = B;
    ++A.n_batches;
    ++e->st.batches;
    A.n_slots += B;
    e->st.host_enqueue_s += now_s() - t0;
    return 0;
  }
  int rc = bore_append_observations(B, D, A.X_seen, A.y_seen, 0, A.cap, w.d_dbl,
                                    w.d_dbl + (size_t)L * D, A.X32, nullptr, sp);
  if (!rc) rc = bore_labels(B, A.y_seen, N_max, c.gamma, A.z, nullptr, sp);
  if (!rc && hipEventRecord(w.ev[0], w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
  if (!rc)
    rc = bore_mlp_fit(&e->desc, B, e->theta, e->adam_m, e->adam_v, e->adam_t, A.X32, A.z, N_max,
                      c.epochs, c.batch_size, nullptr, c.seed, c.loop_id0, 0, &c.adam, nullptr, sp);
  if (!rc && hipEventRecord(w.ev[1], w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
  if (!rc)
    rc = bore_sample_screen_topk(&e->desc, B, e->theta, c.seed, c.loop_id0, 0, c.num_samples,
                                 e->low.data(), e->high.data(), R, w.x0, w.idx, nullptr, sp);
  if (!rc && hipEventRecord(w.ev[2], w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
  if (!rc)
    rc = bore_lbfgsb_minimize(&e->desc, B, e->theta, c.transform, 1, w.x0, R, e->low.data(),
                              e->high.data(), &c.lbfgsb, w.x, w.fun, w.jac, w.info, sp);
  if (!rc && hipEventRecord(w.ev[3], w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
  if (!rc && hipEventRecord(w.done, w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
  bore_set_batch(nullptr);
  if (rc) return rc;
  w.busy = true;
  w.n = B;
  ++A.n_batches;
  ++e->st.batches;
  A.n_slots += B;
  e->st.host_enqueue_s += now_s() - t0;
  return 0;
}

// Stop: workgroups waiting for a row give up (abort flag), everything in flight runs out.  What the
// device did after the host stopped listening has no host-side record: the engine is poisoned.
int async_drain(bore_engine *e) {
  __atomic_store_n(e->as->abort_flag, 1, __ATOMIC_RELEASE);
  for (Worker &w : e->as->workers) {
    (void)hipStreamSynchronize(w.stream);
    w.busy = false;
  }
  __atomic_store_n(e->as->abort_flag, 0, __ATOMIC_RELEASE);
  e->poisoned = true;
  return 0;
}

// The host has seen loop l's result (flag[l] == it[l] + 1): statistics, the suggestion (or the reference's
// random fall-back) into the objective's input block.  Returns the new number of results waiting there.
// (st / sum_flight: where the statistics go -- the engine's own, or a host thread's share of them; xbuf / ids: the
// objective's input block and the list of loops in it -- the engine's, or a host thread's slice of them)
static int take_async_result(bore_engine *e, int l, double t0, int n_done, bore_engine_stats &st, double &sum_flight,
                             double *xbuf, int *ids) {
  Async &A = *e->as;
  const bore_engine_cfg &c = e->cfg;
  const int L = c.n_loops, D = e->D;
  A.state[l] = 3;  // result taken, objective pending
  const double *r = A.result + (size_t)l * (D + 8);
  double *xn = xbuf + (size_t)n_done * D;
  sum_flight += t0 - A.launched_at[l];
  A.seen_at[l] = t0;
  st.launch_to_result_s += t0 - A.launched_at[l];
  st.phase_ns_labels += r[D + 3] * A.ns_per_tick;
  st.phase_ns_fit += r[D + 4] * A.ns_per_tick;
  st.phase_ns_screen += r[D + 5] * A.ns_per_tick;
  st.phase_ns_lbfgsb += r[D + 6] * A.ns_per_tick;
  ++st.phase_iterations;
  if (A.loop_acc.size() == (size_t)L * 6) {
    double *la = &A.loop_acc[(size_t)l * 6];
    la[0] += r[D + 4] * A.ns_per_tick; la[1] += r[D + 6] * A.ns_per_tick;
    la[2] += r[D + 1]; la[3] += r[D + 2]; la[4] += t0 - A.launched_at[l]; la[5] += 1.0;
  }
  if (r[D] < 0.0) {  // reference: fall back to a random point of this loop's stream
    ++st.none_results;
    Mt19937 &rs = e->rs[l];
    for (int d = 0; d < D; ++d) xn[d] = e->low[d] + (e->high[d] - e->low[d]) * rs.next_double();
  } else {
    std::memcpy(xn, r, (size_t)D * 8);
  }
  // (r[D + 7]: evaluations that ran the network; r[D + 1] = nfev also counts the trial points the
  // image shortcut served -- the algorithmic bytes are those of the evaluations that ran)
  st.n_fg_rows += (int64_t)r[D + 7];
  st.n_fg_requests += (int64_t)r[D + 1];
  st.n_rounds += (int64_t)r[D + 2];
  st.argmax_bytes += r[D + 7] * 4.0 * (2 * D + 1) + r[D + 2] * 4.0 * e->P;
  const double N = c.n_init + A.it[l], steps = std::ceil(N / c.batch_size);
  st.fit_bytes += c.epochs * (4.0 * N * (D + 1) + steps * 24.0 * e->P);
  ids[n_done++] = l;
  return n_done;
}
static int take_async_result(bore_engine *e, int l, double t0, int n_done) {
  Async &A = *e->as;
  return take_async_result(e, l, t0, n_done, e->st, A.sum_flight, A.cb_x.data(), A.done_ids.data());
}

// The work-queue schedule (bore_iter.hip: queue_kernel): ONE launch of q_wgs workgroups per run; the
// host appends a (loop, iteration) entry whenever a loop becomes ready -- all of them at the start, then
// each again as soon as its objective value is known -- and ends the workgroups with exit entries.
//
// Host threads (round 5).  One thread serving every loop -- poll the flags, take a result (two lines of pinned
// memory the device has just written: cache misses), the objective, the row and the queue entry -- needs ~1.3 us
// per loop-iteration: ~0.7 M per second, which is what 512 resident workgroups deliver and less than 768 do.  The
// loops are therefore dealt to `T` threads in contiguous shares; a thread owns its loops' host state outright
// (it[], state[], ynew, x_new / y_new, the objective's input block of its share), keeps its own statistics, merged
// at the end, and shares only the queue's tail -- tickets drawn with an atomic add, an entry published by the
// release store of its sequence number, exactly what a waiting workgroup looks for.  A user's objective is called
// by one thread at a time (the library's own Branin is re-entrant).
static int async_host_threads(const bore_engine *e, int L) {
  int cores = (int)std::thread::hardware_concurrency();
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
  // (measured, 4 096 loops on 768 workgroups: 1, 2, 4 and 6 threads give the same 0.70 - 0.74 M it/s -- one thread
  // is not what bounds the schedule; more than one from 8 192 loops on, where a scan of every flag takes long)
  int t = L / 8192 + 1;
  if (t > 4) t = 4;
  if (t > cores - 1) t = cores - 1;  // (the caller's thread is one of them; leave a core to the interpreter's others)
  (void)e;
  return t < 1 ? 1 : t;
}

static int async_run_queue(bore_engine *e, int n_steps) {
  Async &A = *e->as;
  const bore_engine_cfg &c = e->cfg;
  const int L = c.n_loops, D = e->D, R = c.num_starts;
  Worker &w = A.workers[0];
  const size_t ring_n = (size_t)A.q_mask + 1;
  std::memset(A.q_ring, 0, ring_n * sizeof(QueueEntry));
  A.q_tail = 0;
  // (which loop-iteration a slot of the ring held last: a slot is written again only when that entry's result
  // has been seen -- its workgroup read the entry long ago.  With 2 x (loops + workgroups) slots the wait never
  // happens unless a workgroup holding a ticket is descheduled for whole iterations of everybody else; then it
  // keeps that workgroup's entry from being overwritten under it, which would leave it polling for ever.)
  A.q_owner.assign(ring_n * 2, -1);
  HIP_TRY(hipMemsetAsync(A.q_head, 0, 2 * sizeof(unsigned long long), w.stream));
  __atomic_store_n(A.q_tail_host, 0ull, __ATOMIC_RELEASE);
  std::atomic<int> stop{0};  // != 0: a thread has failed (1: callback, 2: no progress for 30 s)
  // An entry is written at the ticket its producer drew and PUBLISHED by the tail counter, which moves over the
  // tickets in order: a producer waits until the tail has reached its ticket (only other host threads' entries,
  // a store each, can be ahead of it) and then moves it one on.
  auto push = [&](int lid, int it) {
    const unsigned long long t = __atomic_fetch_add(&A.q_tail, 1ull, __ATOMIC_RELAXED);
    const size_t slot = (size_t)(t & (unsigned long long)A.q_mask);
    int *own = &A.q_owner[slot * 2];
    const int old_lid = __atomic_load_n(&own[0], __ATOMIC_ACQUIRE), old_it = own[1];
    if (old_lid >= 0) {
      const double t0 = now_s();
      while (__atomic_load_n(&A.flag[old_lid], __ATOMIC_ACQUIRE) < old_it + 1 && !stop.load(std::memory_order_relaxed))
        if (now_s() - t0 > 30.0) { stop.store(2); break; }
    }
    own[1] = it;
    __atomic_store_n(&own[0], lid, __ATOMIC_RELEASE);
    const unsigned long long word = (unsigned long long)(unsigned)lid | ((unsigned long long)(unsigned)it << 32);
    __atomic_store_n(reinterpret_cast<unsigned long long *>(A.q_ring + slot), word, __ATOMIC_RELAXED);
    while (__atomic_load_n(A.q_tail_host, __ATOMIC_RELAXED) != t)
      if (stop.load(std::memory_order_relaxed)) return;
    __atomic_store_n(A.q_tail_host, t + 1, __ATOMIC_RELEASE);
  };
  const double start = now_s();
  for (int l = 0; l < L; ++l) {
    A.target[l] = A.it[l] + n_steps;
    A.state[l] = 1;
    A.launched_at[l] = start;
    // (the row iteration it[l] appends: the previous run's last suggestion and its value)
    double *yn = A.ynew + (size_t)l * (D + 1);
    std::memcpy(yn, &A.x_new[(size_t)l * D], (size_t)D * 8);
    yn[D] = A.y_new[l];
    push(l, A.it[l]);
  }
  __atomic_thread_fence(__ATOMIC_SEQ_CST);
  bore_batch bt;
  std::memset(&bt, 0, sizeof(bt));
  bt.ids = A.d_ident; bt.its = A.d_ident; bt.n_init = c.n_init; bt.deduplicate = c.deduplicate;
  bt.cap = A.cap; bt.X_seen = A.X_seen; bt.result = A.result; bt.flag = A.flag; bt.stamps = A.stamps;
  bt.ynew = A.ynew; bt.yseq = A.yseq; bt.parked = A.parked; bt.abort_flag = A.abort_flag;
  bt.resident_loops = L;
  bore_set_batch(&bt);
  int rc = hipEventRecord(w.ev[2], w.stream) == hipSuccess ? 0 : fail(BORE_E_HIP, "hipEventRecord");
  if (!rc)
    rc = iteration_launch(&e->desc, L, e->theta, e->adam_m, e->adam_v, e->adam_t, A.X_seen, A.y_seen, A.X32,
                          A.z, w.d_dbl, w.d_dbl + (size_t)L * D, c.gamma, c.epochs, c.batch_size, c.seed,
                          c.loop_id0, &c.adam, c.num_samples, e->low.data(), e->high.data(), R, c.transform,
                          &c.lbfgsb, w.x0, w.idx, w.x, w.fun, w.jac, w.info, w.h_args, w.d_args, w.stage_bytes,
                          w.stream, A.q_wgs, A.q_ring, A.q_head, A.q_mask, nullptr, A.q_tail_host);
  if (!rc && hipEventRecord(w.ev[3], w.stream) != hipSuccess) rc = fail(BORE_E_HIP, "hipEventRecord");
  bore_set_batch(nullptr);
  if (rc) {
    e->poisoned = true;
    return rc;
  }
  ++A.n_batches;
  ++e->st.batches;
  A.n_slots += L;

  const int T = async_host_threads(e, L);
  A.host_threads = T;
  const bool own_objective = e->objective == bore_objective_branin01;
  std::mutex objective_mutex;
  struct Share {
    bore_engine_stats st;
    double sum_flight = 0;
    long long n_resident = 0;
  };
  std::vector<Share> shares((size_t)T);
  for (Share &sh : shares) std::memset(&sh.st, 0, sizeof(sh.st));
  auto serve = [&](int ti) {
    const int l0 = (int)((long long)L * ti / T), l1 = (int)((long long)L * (ti + 1) / T);
    Share &sh = shares[(size_t)ti];
    double *xbuf = A.cb_x.data() + (size_t)l0 * D, *ybuf = A.cb_y.data() + l0;
    int *ids = A.done_ids.data() + l0;
    int remaining = l1 - l0, n_done = 0;
    double last_progress = now_s();
    while (remaining && !stop.load(std::memory_order_relaxed)) {
      const double t0 = now_s();
      for (int l = l0; l < l1; ++l) {
        if (A.state[l] != 1) continue;
        if (__atomic_load_n(&A.flag[l], __ATOMIC_ACQUIRE) != A.it[l] + 1) continue;
        n_done = take_async_result(e, l, t0, n_done, sh.st, sh.sum_flight, xbuf, ids);
      }
      if (n_done) {
        int cb;
        if (own_objective) {
          cb = e->objective(xbuf, n_done, D, ybuf, e->user);
        } else {
          std::lock_guard<std::mutex> lock(objective_mutex);
          cb = e->objective(xbuf, n_done, D, ybuf, e->user);
        }
        if (cb) {
          stop.store(1);
          break;
        }
        const double now = now_s();
        for (int k = 0; k < n_done; ++k) {
          const int l = ids[k];
          std::memcpy(&A.x_new[(size_t)l * D], &xbuf[(size_t)k * D], (size_t)D * 8);
          A.y_new[l] = ybuf[k];
          sh.st.result_to_ready_s += now - A.seen_at[l];
          ++A.it[l];
          if (A.it[l] >= A.target[l]) {
            A.state[l] = 2;
            --remaining;
          } else {
            double *yn = A.ynew + (size_t)l * (D + 1);
            std::memcpy(yn, &xbuf[(size_t)k * D], (size_t)D * 8);
            yn[D] = ybuf[k];
            push(l, A.it[l]);   // (the entry's release store orders the row before it)
            A.state[l] = 1;
            A.launched_at[l] = now;
            ++sh.n_resident;
          }
        }
        last_progress = now;
        sh.st.host_finalize_s += now - t0;
        n_done = 0;
      } else if (t0 - last_progress > 30.0) {
        stop.store(2);
      }
    }
  };
  {
    std::vector<std::thread> helpers;
    for (int ti = 1; ti < T; ++ti) helpers.emplace_back(serve, ti);
    serve(0);
    for (std::thread &th : helpers) th.join();
  }
  for (const Share &sh : shares) {  // (sums; host_finalize_s: the busiest thread's, the bound on the host side)
    const bore_engine_stats &p = sh.st;
    bore_engine_stats &q = e->st;
    q.launch_to_result_s += p.launch_to_result_s; q.result_to_ready_s += p.result_to_ready_s;
    q.phase_ns_labels += p.phase_ns_labels; q.phase_ns_fit += p.phase_ns_fit; q.phase_ns_screen += p.phase_ns_screen;
    q.phase_ns_lbfgsb += p.phase_ns_lbfgsb; q.phase_iterations += p.phase_iterations; q.none_results += p.none_results;
    q.n_fg_rows += p.n_fg_rows; q.n_fg_requests += p.n_fg_requests; q.n_rounds += p.n_rounds;
    q.argmax_bytes += p.argmax_bytes; q.fit_bytes += p.fit_bytes;
    A.sum_flight += sh.sum_flight;
    A.n_resident += sh.n_resident;
  }
  {
    double busiest = 0;
    for (const Share &sh : shares) busiest = sh.st.host_finalize_s > busiest ? sh.st.host_finalize_s : busiest;
    e->st.host_finalize_s += busiest;
  }
  if (stop.load()) {
    const int why = stop.load();
    __atomic_store_n(A.q_tail_host, ~0ull, __ATOMIC_RELEASE);
    async_drain(e);
    for (int l = 0; l < L; ++l) A.state[l] = 2;
    return why == 1 ? fail(BORE_E_CALLBACK, "engine_run: the objective callback failed")
                    : fail(BORE_E_HIP, "engine_run: no loop finished for 30 s");
  }
  __atomic_store_n(A.q_tail_host, ~0ull, __ATOMIC_RELEASE);  // (no more work: every workgroup sees it through the copy)
  HIP_TRY(hipStreamSynchronize(w.stream));
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, w.ev[2], w.ev[3]));
  e->st.fit_launches += 1;
  e->st.argmax_ms += ms;
  e->st.argmax_launches += 1;
  e->st.host_enqueue_s += 0.0;
  return 0;
}

int async_run(bore_engine *e, int n_steps) {
  Async &A = *e->as;
  const bore_engine_cfg &c = e->cfg;
  const int L = c.n_loops, D = e->D;
  int max_it = 0;
  for (int l = 0; l < L; ++l) max_it = A.it[l] > max_it ? A.it[l] : max_it;
  int rc;
  if (c.n_init + max_it + n_steps > A.cap) {
    int64_t cap = A.cap;
    while (cap < c.n_init + max_it + n_steps) cap *= 2;
    if ((rc = async_alloc(e, cap)) || (rc = async_decide_schedule(e))) return rc;
  }
  if (A.queue) return async_run_queue(e, n_steps);
  const double start = now_s();
  int remaining = L;
  for (int l = 0; l < L; ++l) {
    A.target[l] = A.it[l] + n_steps;
    A.state[l] = 0;
    A.ready_since[l] = start;
  }
  // Launch policy: a free worker takes the ready loops once there are `min_batch` of them or the
  // oldest has waited `max_wait` (or nothing else is running).  With resident workgroups a loop
  // only needs a launch at the start of a run and after its workgroup gave up waiting: at once.
  const bool resident = A.fused && A.wait_ticks > 0;
  const int min_batch = resident ? 1 : (L >= 16 ? L / 8 : 1);
  const double max_wait = resident ? 0.0 : 150e-6;
  double last_progress = start, first_done = start;
  int n_done = 0;
  // Objective calls are bunched (a call into the interpreter costs tens of microseconds).  Resident
  // workgroups wait for the value on their CUs: call as soon as anything is there -- the results
  // that arrive during a call form the next bunch.
  const int cb_min = resident ? 1 : (L >= 64 ? L / 16 : 1);
  const double cb_wait = resident ? 0.0 : 40e-6;
  while (remaining) {
    // 1. loops on the device: a result has arrived, or the workgroup has left the next iteration
    // to a later launch (parked)
    const double t0 = now_s();
    for (int l = 0; l < L; ++l) {
      if (A.state[l] != 1) continue;
      if (__atomic_load_n(&A.flag[l], __ATOMIC_ACQUIRE) != A.it[l] + 1) {
        if (A.fused && !A.via_launch[l] && __atomic_load_n(&A.parked[l], __ATOMIC_ACQUIRE) == A.it[l]) {
          A.state[l] = 0;  // its row was delivered too late (or nobody waited): needs a launch
          A.ready_since[l] = t0;
          ++A.n_parked;
        }
        continue;
      }
      if (n_done == 0) first_done = t0;
      n_done = take_async_result(e, l, t0, n_done);
    }
    if (n_done && (n_done >= cb_min || t0 - first_done >= cb_wait)) {
      if (e->objective(A.cb_x.data(), n_done, D, A.cb_y.data(), e->user)) {
        async_drain(e);
        for (int l = 0; l < L; ++l) A.state[l] = 2;
        return fail(BORE_E_CALLBACK, "engine_run: the objective callback failed");
      }
      const double now = now_s();
      for (int k = 0; k < n_done; ++k) {
        const int l = A.done_ids[k];
        std::memcpy(&A.x_new[(size_t)l * D], &A.cb_x[(size_t)k * D], (size_t)D * 8);
        A.y_new[l] = A.cb_y[k];
        e->st.result_to_ready_s += now - A.seen_at[l];
        ++A.it[l];
        if (A.it[l] >= A.target[l]) {
          A.state[l] = 2;
          --remaining;
        } else if (A.fused) {
          // hand the row to the loop's workgroup, which may still be waiting for it on its CU;
          // if it has given up (parked[l] == it[l], seen in pass 1) the loop joins a launch
          double *yn = A.ynew + (size_t)l * (D + 1);
          std::memcpy(yn, &A.cb_x[(size_t)k * D], (size_t)D * 8);
          yn[D] = A.cb_y[k];
          __atomic_store_n(&A.yseq[l], A.it[l], __ATOMIC_RELEASE);
          A.state[l] = 1;
          A.via_launch[l] = 0;
          A.launched_at[l] = now;
          ++A.n_resident;
        } else {
          A.state[l] = 0;
          A.ready_since[l] = now;
        }
      }
      last_progress = now;
      e->st.host_finalize_s += now - t0;
      n_done = 0;
    }
    // 2. workers whose launch has run out
    int busy = 0;
    for (Worker &w : A.workers) {
      if (!w.busy) continue;
      const hipError_t q = hipEventQuery(w.done);
      if (q == hipErrorNotReady) {
        ++busy;
        continue;
      }
      if (q != hipSuccess) {
        e->poisoned = true;
        return fail(BORE_E_HIP, "engine_run: %s", hipGetErrorString(q));
      }
      float ms = 0.f;
      if (!A.fused) {
        HIP_TRY(hipEventElapsedTime(&ms, w.ev[0], w.ev[1]));
        e->st.fit_ms += ms;
      }
      e->st.fit_launches += 1;
      HIP_TRY(hipEventElapsedTime(&ms, w.ev[2], w.ev[3]));
      e->st.argmax_ms += ms;
      e->st.argmax_launches += 1;
      w.busy = false;
    }
    // 3. a free worker takes the ready loops
    if (busy < (int)A.workers.size()) {
      A.scratch_ids.clear();
      double oldest = 1e300;
      for (int l = 0; l < L; ++l)
        if (A.state[l] == 0) {
          A.scratch_ids.push_back(l);
          oldest = A.ready_since[l] < oldest ? A.ready_since[l] : oldest;
        }
      const int nr = (int)A.scratch_ids.size();
      if (nr && (nr >= min_batch || busy == 0 || now_s() - oldest >= max_wait)) {
        for (Worker &w : A.workers)
          if (!w.busy) {
            if ((rc = async_launch(e, w))) {
              async_drain(e);
              return rc;
            }
            break;
          }
        last_progress = now_s();
      }
    }
    if (now_s() - last_progress > 30.0) {
      async_drain(e);
      return fail(BORE_E_HIP, "engine_run: no loop finished for 30 s");
    }
  }
  if (getenv("BORE_ASYNC_DEBUG"))
    fprintf(stderr, "[async] %lld launches, %.1f loops/launch; %lld iterations continued on their CU, %lld parked; "
            "ready->launch %.0f us, launch->result %.0f us (means)\n",
            A.n_batches, (double)A.n_slots / (double)(A.n_batches ? A.n_batches : 1), A.n_resident, A.n_parked,
            1e6 * A.sum_wait / (double)(A.n_wait ? A.n_wait : 1), 1e6 * A.sum_flight / (double)(A.n_wait ? A.n_wait : 1));
  // let the launches run out (their loops are done: the workgroups exit after the last iteration)
  for (Worker &w : A.workers)
    if (w.busy) {
      HIP_TRY(hipStreamSynchronize(w.stream));
      float ms = 0.f;
      if (!A.fused) {
        HIP_TRY(hipEventElapsedTime(&ms, w.ev[0], w.ev[1]));
        e->st.fit_ms += ms;
      }
      e->st.fit_launches += 1;
      HIP_TRY(hipEventElapsedTime(&ms, w.ev[2], w.ev[3]));
      e->st.argmax_ms += ms;
      e->st.argmax_launches += 1;
      w.busy = false;
    }
  return 0;
}

}  // namespace

static const char kPoisoned[] =
    "engine: an earlier bore_engine_run stopped half-way (objective error, watchdog or HIP error); "
    "its loops are at different iterations -- create a new engine";

extern "C" void bore_engine_destroy(bore_engine *e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  async_destroy(e);
  for (Group &g : e->groups) free_group(g);
  void *dev[] = {e->theta, e->adam_m, e->adam_v, e->adam_t};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  delete e;
}

extern "C" int bore_engine_create(const bore_mlp_desc *desc, const bore_engine_cfg *cfg,
                                  const float *theta0, const double *X0, const double *y0,
                                  const uint32_t *mt_state, bore_objective_fn objective, void *user,
                                  bore_engine **out) {
  if (!desc || !cfg || !theta0 || !X0 || !y0 || !mt_state || !objective || !out)
    return fail(BORE_E_INVALID, "engine_create: NULL argument");
  const int64_t P = bore_param_count(desc);
  if (P < 0) return fail(BORE_E_INVALID, "engine_create: bad bore_mlp_desc");
  const int D = desc->input_dim, L = cfg->n_loops;
  if (L < 1 || cfg->groups < 1 || cfg->n_init < 1 || cfg->epochs < 1 || cfg->num_starts < 1 ||
      cfg->num_samples < cfg->num_starts || !cfg->low || !cfg->high || D < 1 || D > BORE_DIM_MAX)
    return fail(BORE_E_INVALID, "engine_create: bad configuration");
  bore_engine *e = new (std::nothrow) bore_engine();
  if (!e) return fail(BORE_E_HIP, "engine_create: out of memory");
  e->desc = *desc;
  e->cfg = *cfg;
  e->D = D;
  e->P = (int)P;
  e->low.assign(cfg->low, cfg->low + D);
  e->high.assign(cfg->high, cfg->high + D);
  e->cfg.low = e->low.data();
  e->cfg.high = e->high.data();
  e->objective = objective;
  e->user = user;
  std::memset(&e->st, 0, sizeof(e->st));
  e->rs.resize(L);
  for (int l = 0; l < L; ++l) {
    std::memcpy(e->rs[l].key, mt_state + (size_t)l * 625, 624 * 4);
    e->rs[l].pos = (int)mt_state[(size_t)l * 625 + 624];
  }
  int rc = 0;
#define ENG_TRY(expr)                \
  do {                               \
    if ((rc = (expr))) {             \
      bore_engine_destroy(e);        \
      return rc;                     \
    }                                \
  } while (0)
#define ENG_HIP(expr)                                                          \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      bore_engine_destroy(e);                                                  \
      return fail(BORE_E_HIP, "%s: %s", #expr, hipGetErrorString(e_));         \
    }                                                                          \
  } while (0)
  ENG_TRY(dev_alloc(&e->theta, (size_t)L * P));
  ENG_TRY(dev_alloc(&e->adam_m, (size_t)L * P));
  ENG_TRY(dev_alloc(&e->adam_v, (size_t)L * P));
  ENG_TRY(dev_alloc(&e->adam_t, (size_t)L));
  ENG_HIP(hipMemcpy(e->theta, theta0, (size_t)L * P * 4, hipMemcpyHostToDevice));
  ENG_HIP(hipMemset(e->adam_m, 0, (size_t)L * P * 4));
  ENG_HIP(hipMemset(e->adam_v, 0, (size_t)L * P * 4));
  ENG_HIP(hipMemset(e->adam_t, 0, (size_t)L * 8));
  if (cfg->async_loops) {
    ENG_TRY(async_create(e, X0, y0));
    *out = e;
    return 0;
  }
  const int G = cfg->groups < L ? cfg->groups : L;
  e->groups.resize(G);
  const int R = cfg->num_starts, n0 = cfg->n_init;
  for (int k = 0; k < G; ++k) {
    Group &g = e->groups[k];
    // the same split as numpy.linspace(0, L, G + 1).astype(int) (engine.py)
    const double step = (double)L / G;
    g.a = (int)(k * step);
    g.b = k + 1 == G ? L : (int)((k + 1) * step);
    const size_t Lg = g.b - g.a;
    ENG_HIP(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
    ENG_HIP(hipEventCreateWithFlags(&g.done, hipEventDisableTiming));
    for (hipEvent_t &ev : g.ev) ENG_HIP(hipEventCreate(&ev));
    ENG_TRY(alloc_record(e, g, n0 * 2 > 256 ? n0 * 2 : 256));
    ENG_HIP(hipMemcpy2D(g.X_seen, g.cap * D * 8, X0 + (size_t)g.a * n0 * D, (size_t)n0 * D * 8,
                        (size_t)n0 * D * 8, Lg, hipMemcpyHostToDevice));
    ENG_HIP(hipMemcpy2D(g.y_seen, g.cap * 8, y0 + (size_t)g.a * n0, (size_t)n0 * 8, (size_t)n0 * 8, Lg,
                        hipMemcpyHostToDevice));
    g.n = n0;
    ENG_TRY(dev_alloc(&g.new_x, Lg * (D + 1)));  // new_x [Lg][D] | new_y [Lg]
    g.new_y = g.new_x + Lg * D;
    ENG_TRY(dev_alloc(&g.x0, Lg * R * D));
    ENG_TRY(dev_alloc(&g.x, Lg * R * D));
    ENG_TRY(dev_alloc(&g.jac, Lg * R * D));
    ENG_TRY(dev_alloc(&g.fun, Lg * R));
    // x_best [Lg][D] fp64 | info [Lg][R][5] int32 | best [Lg] int32
    ENG_TRY(dev_alloc(&g.x_best, Lg * D + (Lg * R * 5 + Lg + 1) / 2));
    g.info = reinterpret_cast<int32_t *>(g.x_best + Lg * D);
    g.best = g.info + Lg * R * 5;
    ENG_TRY(dev_alloc(&g.idx, Lg * R));
    ENG_TRY(pin_alloc(&g.new_x_pin, Lg * (D + 1)));
    g.new_y_pin = g.new_x_pin + Lg * D;
    ENG_TRY(pin_alloc(&g.x_best_pin, Lg * D + (Lg * R * 5 + Lg + 1) / 2));
    g.info_pin = reinterpret_cast<int32_t *>(g.x_best_pin + Lg * D);
    g.best_pin = g.info_pin + Lg * R * 5;
  }
#undef ENG_TRY
#undef ENG_HIP
  *out = e;
  return 0;
}

// Advance every loop by n_steps BO iterations.  Groups proceed independently; the host only
// reacts to completion events.
extern "C" int bore_engine_run(bore_engine *e, int n_steps) {
  if (!e || n_steps < 0) return fail(BORE_E_INVALID, "engine_run: bad argument");
  if (e->poisoned) return fail(BORE_E_INVALID, kPoisoned);
  if (n_steps == 0) return 0;
  if (e->as) return async_run(e, n_steps);
  std::vector<int64_t> target(e->groups.size());
  int rc;
  for (size_t k = 0; k < e->groups.size(); ++k) target[k] = e->groups[k].steps + n_steps;
  for (size_t k = 0; k < e->groups.size(); ++k)
    if ((rc = enqueue(e, e->groups[k]))) {
      for (Group &o : e->groups) {
        (void)hipStreamSynchronize(o.stream);
        o.inflight = false;
      }
      return rc;
    }
  size_t remaining = e->groups.size();
  while (remaining) {
    for (size_t k = 0; k < e->groups.size(); ++k) {
      Group &g = e->groups[k];
      if (!g.inflight) continue;
      const hipError_t q = hipEventQuery(g.done);
      if (q == hipErrorNotReady) continue;
      if (q != hipSuccess) {
        e->poisoned = true;
        return fail(BORE_E_HIP, "engine_run: %s", hipGetErrorString(q));
      }
      if ((rc = finalize(e, g)) || (g.steps < target[k] && (rc = enqueue(e, g)))) {
        // stop: let the other groups' launches finish; their iterations are not in the record
        for (Group &o : e->groups) {
          (void)hipStreamSynchronize(o.stream);
          o.inflight = false;
        }
        e->poisoned = true;
        return rc;
      }
      if (g.steps >= target[k]) --remaining;
    }
  }
  return 0;
}

extern "C" int64_t bore_engine_size(const bore_engine *e) {
  if (e && e->as) return e->cfg.n_init + e->as->it[0];
  if (!e || e->groups.empty()) return -1;
  const Group &g = e->groups[0];
  return g.n + (g.has_new ? 1 : 0);
}

// The record of every loop: X host [n_loops][N][D], y host [n_loops][N], N = bore_engine_size().
extern "C" int bore_engine_observations(bore_engine *e, double *X, double *y) {
  if (!e || !X || !y) return fail(BORE_E_INVALID, "engine_observations: NULL argument");
  if (e->poisoned) return fail(BORE_E_INVALID, kPoisoned);
  const int D = e->D;
  const int64_t N = bore_engine_size(e);
  if (e->as) {  // rows on the device: all but the newest, which waits on the host for its launch
    Async &A = *e->as;
    const size_t L = e->cfg.n_loops;
    HIP_TRY(hipDeviceSynchronize());
    const int64_t nd = A.it[0] > 0 ? N - 1 : N;
    HIP_TRY(hipMemcpy2D(X, (size_t)N * D * 8, A.X_seen, A.cap * D * 8, (size_t)nd * D * 8, L,
                        hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy2D(y, (size_t)N * 8, A.y_seen, A.cap * 8, (size_t)nd * 8, L, hipMemcpyDeviceToHost));
    if (A.it[0] > 0)
      for (size_t l = 0; l < L; ++l) {
        std::memcpy(X + (l * N + nd) * D, &A.x_new[l * D], (size_t)D * 8);
        y[l * N + nd] = A.y_new[l];
      }
    return 0;
  }
  for (Group &g : e->groups) {
    const size_t Lg = g.b - g.a;
    HIP_TRY(hipStreamSynchronize(g.stream));
    HIP_TRY(hipMemcpy2D(X + (size_t)g.a * N * D, (size_t)N * D * 8, g.X_seen, g.cap * D * 8,
                        (size_t)g.n * D * 8, Lg, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy2D(y + (size_t)g.a * N, (size_t)N * 8, g.y_seen, g.cap * 8, (size_t)g.n * 8, Lg,
                        hipMemcpyDeviceToHost));
    if (g.has_new)  // the newest row is still staged on the host
      for (size_t l = 0; l < Lg; ++l) {
        std::memcpy(X + ((size_t)(g.a + l) * N + g.n) * D, g.new_x_pin + l * D, (size_t)D * 8);
        y[(size_t)(g.a + l) * N + g.n] = g.new_y_pin[l];
      }
  }
  return 0;
}

// Classifier state of every loop (host buffers; any may be NULL).
extern "C" int bore_engine_state(bore_engine *e, float *theta, float *adam_m, float *adam_v,
                                 int64_t *adam_t) {
  if (!e) return fail(BORE_E_INVALID, "engine_state: NULL engine");
  if (e->poisoned) return fail(BORE_E_INVALID, kPoisoned);
  HIP_TRY(hipDeviceSynchronize());
  const size_t n = (size_t)e->cfg.n_loops * e->P;
  if (theta) HIP_TRY(hipMemcpy(theta, e->theta, n * 4, hipMemcpyDeviceToHost));
  if (adam_m) HIP_TRY(hipMemcpy(adam_m, e->adam_m, n * 4, hipMemcpyDeviceToHost));
  if (adam_v) HIP_TRY(hipMemcpy(adam_v, e->adam_v, n * 4, hipMemcpyDeviceToHost));
  if (adam_t) HIP_TRY(hipMemcpy(adam_t, e->adam_t, (size_t)e->cfg.n_loops * 8, hipMemcpyDeviceToHost));
  return 0;
}

// Diagnostic (tools/loop_tail.py; not part of include/bore_hip.h): per-loop sums since the last reset --
// out[L][6] = fit ns, restart ns, evaluations, rounds of the slowest restart, launch -> result s, iterations.
// reset != 0 also (re)starts the accumulation.  Asynchronous engines only.
extern "C" int bore_debug_engine_loop_stats(bore_engine *e, double *out, int reset) {
  if (!e || !e->as) return fail(BORE_E_INVALID, "loop_stats: an asynchronous engine is needed");
  Async &A = *e->as;
  const size_t n = (size_t)e->cfg.n_loops * 6;
  if (out && A.loop_acc.size() == n) std::memcpy(out, A.loop_acc.data(), n * 8);
  if (reset) A.loop_acc.assign(n, 0.0);
  return 0;
}

extern "C" int bore_engine_get_stats(bore_engine *e, bore_engine_stats *out, int reset) {
  if (!e || !out) return fail(BORE_E_INVALID, "engine_get_stats: NULL argument");
  *out = e->st;
  out->worker_streams = e->as ? (int64_t)e->as->workers.size() : (int64_t)e->groups.size();
  out->stream_concurrency = e->as ? e->as->stream_concurrency : 0;
  out->loops_per_cu = e->as ? e->as->per_cu : 0;
  out->side_by_side_workgroups = e->as && e->as->fused ? (e->as->queue ? e->as->q_wgs : e->cfg.n_loops) : 0;
  out->host_threads = e->as ? e->as->host_threads : 1;
  if (reset) std::memset(&e->st, 0, sizeof(e->st));
  return 0;
}
