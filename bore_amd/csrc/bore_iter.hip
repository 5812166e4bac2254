// bore_iter.hip -- one BO iteration of a loop as ONE kernel (asynchronous replica schedule).
//
// append -> labels -> fit -> sample + screen -> L-BFGS-B restarts -> pick, the bodies of the
// kernels of bore_hip.hip / bore_argmax.hip run back to back by the same workgroup, one workgroup
// per loop of the batch (bore_set_batch).  With separate launches a loop waits for the slowest
// fit of its batch before its restarts start, and a launch chain holds its stream until the
// slowest restart of the batch is done; as one kernel a loop-iteration takes its OWN fit plus its
// OWN restarts, and batches are independent single-kernel packets (more than four concurrently
// running dependent chains do not scale on this GPU).  The phases hand data to each
// other through global memory (labels z, the fitted theta, the screened x0): a workgroup-wide
// barrier plus an agent-scope fence separates them (the fit re-reads nothing it wrote, but the
// screening phase loads theta lines the fit's prologue may have left in the vector L1).
// Static shape 1 (2 -> 16-16-1, the BASELINE config) only; anything else keeps the launch chain.
// Included by bore_all.hip after bore_hip.hip and bore_argmax.hip.

// One entry of the work queue of queue_kernel (pinned host memory, written by the host in ticket order:
// lid and it first, then seq = ticket + 1 with release semantics; lid < 0 = "no more work, exit").
struct QueueEntry {
  int lid, it;
  long long seq;
};

struct IterArgs {
  FitArgs f;
  ScreenArgs s;
  LbfgsbArgs b;
  double *X_seen, *y_seen;  // records, cap-strided per loop
  float *X32, *z;
  const double *x_new, *y_new;  // per slot: the row to append before the launch's first iteration
  long long *stamps;            // [n_loops][4] device-clock stamps of the phase boundaries, or NULL
  double gamma;
  int D;
  // Residency (include/bore_hip.h: bore_batch).  A workgroup may run SEVERAL consecutive iterations
  // of its loop in one launch: after publishing iteration it's suggestion it waits up to
  // wait_ticks of the device clock for the host to deliver the objective value (ynew / yseq,
  // pinned host memory polled over the bus by one lane) and goes on with iteration it + 1 without
  // a launch, an upload or a place in a batch.  If the row does not come in time (a slow
  // objective, wait_ticks = 0, an abort) it says so in parked[] and exits: every wave of the grid
  // ends within wait_ticks of host silence.
  const int *targets;        // per slot: run iterations its[slot] .. targets[slot] - 1 at most
  const double *ynew;        // pinned [n_loops][D + 1]: x | y of each loop's newest observation
  const int *yseq;           // pinned [n_loops]: iterations whose row the host has delivered
  int *parked;               // pinned [n_loops]: the iteration this workgroup left to a later launch
  const int *abort_flag;     // pinned [1]: non-zero = stop waiting
  long long wait_ticks;
  // Work-queue launches (queue_kernel): the ring of entries (pinned), its size - 1 (a power of two),
  // the ticket counter (device memory, zeroed by the host before the launch)
  const QueueEntry *q_ring;
  unsigned long long *q_head;
  int q_mask;
  // Resident launches: loop-iterations finished by the whole grid since the launch (zeroed by the
  // upload of this block; device copy only, touched with agent-scope atomics and never through the
  // constant-address-space view of the block).  A loop behind the grid's mean raises its waves'
  // priority, one ahead lowers it (iteration_kernel).
  int progress;
};

// 64-bit load from host memory that the host may have written since the kernel started
__device__ __forceinline__ double load_host_f64(const double *p) {
  const long long v = __hip_atomic_load(reinterpret_cast<const long long *>(p), __ATOMIC_RELAXED,
                                        __HIP_MEMORY_SCOPE_SYSTEM);
  return __longlong_as_double(v);
}

// One iteration of the loop in `slot`.  Round 2 made this a real (noinline) call from the resident
// kernel's loop: inlined, the compiler hoisted the phases' argument loads and per-lane address
// arithmetic out of the loop and kept them live around it (420 B of scratch per lane) -- but the call
// parks the callee-saved half of the register file in the private segment (568 - 784 B per lane,
// ~300 KB of HBM traffic per loop-iteration, profiles/r2/pmc_traffic.json).  Round 3 inlines it again
// and takes the loop-invariants away instead: the argument block's address is opaque per iteration
// and every phase body derives its lane numbers from an opaque copy of the work-item id
// (BORE_OPAQUE_TID), so what the loop carries is the slot, the iteration count and a dozen lane
// constants of the register network (100 B of scratch, spilled once per BO iteration; same speed:
// profiles/r3/resident_inline_vs_call.txt).
// (the restart phase re-requests the network's operands per evaluation: lbfgsb_body's LEAN)
#ifndef BORE_LAG_PRIO
#define BORE_LAG_PRIO 1  // -DBORE_LAG_PRIO=0: no wave priority for lagging loops, no yielding leaders (profiles/r4/ab_log.txt)
#endif
#ifndef BORE_LAG_YIELD_Q
#define BORE_LAG_YIELD_Q 3  // a leader yields while it is more than Q / 4 iterations ahead of the mean
#endif
template <int SHAPE>
__device__ __forceinline__ void iteration_once(const IterArgs *__restrict__ pa,
                                                         const long long slot, const int it,
                                                         const bool staged) {
  const IterArgs &a = *pa;
  const long long lid = uniform_i64(a.f.ids[slot]), cap = a.f.cap;
  long long *stp = a.stamps ? a.stamps + lid * 4 : nullptr;  // read back by the restart phase's epilogue
  if (stp && threadIdx.x == 0) stp[0] = wall_clock64();
  if (it > 0) {  // append (append_kernel's batch branch)
    const long long row = a.f.n_init + it - 1;
    // (the launch brought the row of its first iteration; later ones come from the host)
    for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
      const double v = staged ? a.x_new[slot * a.D + d] : load_host_f64(a.ynew + lid * (a.D + 1) + d);
      a.X_seen[(lid * cap + row) * a.D + d] = v;
      a.X32[(lid * cap + row) * a.D + d] = (float)v;
    }
    if (threadIdx.x == 0)
      a.y_seen[lid * cap + row] = staged ? a.y_new[slot] : load_host_f64(a.ynew + lid * (a.D + 1) + a.D);
  }
  __threadfence();
  __syncthreads();
  labels_body(a.y_seen, 0, 0.0, a.z, nullptr, a.f.ids, a.f.its, a.f.n_init, cap, a.gamma, slot, it);
  if (stp && threadIdx.x == 0) stp[1] = wall_clock64();
  __threadfence();
  __syncthreads();
  fit_body<SHAPE>(a.f, slot, it);
  if (stp && threadIdx.x == 0) stp[2] = wall_clock64();
  __threadfence();
  __syncthreads();
  screen_body<SHAPE, false>(a.s, slot, it);
  if (stp && threadIdx.x == 0) stp[3] = wall_clock64();
  __threadfence();
  __syncthreads();
  lbfgsb_body<SHAPE, false, true, true, false>(a.b, slot, 0, it);  // (publishes flag[lid] = it + 1)
}

// RESIDENT: the workgroup may go on to later iterations of its loop.  Otherwise ONE iteration -- the
// launch for loops that do not wait on their CU (more loops than the device holds at once,
// wait_ticks = 0): no loop, no scratch at all.
template <int SHAPE, bool RESIDENT>
__global__ __launch_bounds__(BORE_THREADS, 2) void iteration_kernel(const IterArgs *__restrict__ pa) {
  const long long slot = blockIdx.x;
  const int it_first = uniform_i32(pa->f.its[slot]);
  if constexpr (!RESIDENT) {
    typedef const __attribute__((address_space(4))) IterArgs *IterArgsConst1;  // (scalar loads: see below)
    const IterArgs &a = *(const IterArgs *)(IterArgsConst1) reinterpret_cast<unsigned long long>(pa);
    const long long lid = uniform_i64(a.f.ids[slot]), cap = a.f.cap;
    const int it = it_first;
    long long *stp = a.stamps ? a.stamps + lid * 4 : nullptr;
    if (stp && threadIdx.x == 0) stp[0] = wall_clock64();
    if (it > 0) {  // append (append_kernel's batch branch)
      const long long row = a.f.n_init + it - 1;
      for (int d = threadIdx.x; d < a.D; d += blockDim.x) {
        const double v = a.x_new[slot * a.D + d];
        a.X_seen[(lid * cap + row) * a.D + d] = v;
        a.X32[(lid * cap + row) * a.D + d] = (float)v;
      }
      if (threadIdx.x == 0) a.y_seen[lid * cap + row] = a.y_new[slot];
    }
    __threadfence();
    __syncthreads();
    labels_body(a.y_seen, 0, 0.0, a.z, nullptr, a.f.ids, a.f.its, a.f.n_init, cap, a.gamma, slot);
    if (stp && threadIdx.x == 0) stp[1] = wall_clock64();
    __threadfence();
    __syncthreads();
    fit_body<SHAPE>(a.f, slot);
    if (stp && threadIdx.x == 0) stp[2] = wall_clock64();
    __threadfence();
    __syncthreads();
    screen_body<SHAPE, false>(a.s, slot);
    if (stp && threadIdx.x == 0) stp[3] = wall_clock64();
    __threadfence();
    __syncthreads();
    lbfgsb_body<SHAPE, false, true, true, false>(a.b, slot, 0);  // (publishes flag[lid] = it + 1)
    // not the loop's last iteration: leave the next one to a later launch.  (Any wave may say so,
    // and before the others are done: the host reacts to `parked` only after the flag.)
    if (a.targets && threadIdx.x == 0 && it + 1 < a.targets[slot])
      __hip_atomic_store(a.parked + lid, it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  __shared__ __attribute__((aligned(16))) int s_go4[4];  // (16 B: the dynamic LDS keeps its alignment)
  const int target = pa->targets ? uniform_i32(pa->targets[slot]) : it_first + 1;
#if BORE_LAG_PRIO
  // Two loops share a CU (512 loops on 256 CUs) and their waves share the SIMDs' issue slots.  The
  // timed region of K iterations ends with the SLOWEST loop -- loops whose restarts need twice the
  // evaluations, 1.45x the mean loop's time (profiles/r3/loop_tail.txt) -- so a loop that is behind
  // the grid's mean progress runs its waves at priority 3, one that is ahead at 0: the laggard gets
  // the issue slots first (alone on a CU a loop is ~10 % faster), its partner pays with slack it has.
  int *prog;
  {
    unsigned long long pb = reinterpret_cast<unsigned long long>(pa) + offsetof(IterArgs, progress);
    asm volatile("" : "+s"(pb));
    prog = reinterpret_cast<int *>(pb);
  }
#endif
  for (int it = it_first;;) {
#if BORE_LAG_PRIO
    {
      const int total = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int n_wg = (int)gridDim.x;
      const int lead = uniform_i32((it - it_first) * n_wg - total);  // > 0: ahead of the mean, in 1 / n_wg iterations
      if (lead < -(n_wg >> 1)) __builtin_amdgcn_s_setprio(3);
      else if (lead > (n_wg >> 1)) __builtin_amdgcn_s_setprio(0);
      else __builtin_amdgcn_s_setprio(1);
    }
#endif
    // (the argument block's address is made opaque per iteration, so that nothing read through it
    // is hoisted out of the loop and kept live around it)
    // (the block is written by the host before the launch and never by a kernel: read through the
    // constant address space its fields are scalar loads -- uniform values in scalar registers, loops
    // over them scalar loops -- instead of vector loads that a store earlier in the loop might clobber)
    unsigned long long pa_bits = reinterpret_cast<unsigned long long>(pa);
    asm volatile("" : "+s"(pa_bits));
    typedef const __attribute__((address_space(4))) IterArgs *IterArgsConst;
    iteration_once<SHAPE>((const IterArgs *)(IterArgsConst)pa_bits, slot, it, it == it_first);
    __syncthreads();  // the waves leave the restart phase one by one: LDS is reused below
    ++it;
#if BORE_LAG_PRIO
    if (threadIdx.x == 0) __hip_atomic_fetch_add(prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
    if (it >= target) break;
    // ---- the next row: delivered while we were busy, or within wait_ticks, or not our business ----
    if (threadIdx.x == 0) {
      const long long lid = pa->f.ids[slot], wait_ticks = pa->wait_ticks;
      const int *yseq = pa->yseq + lid, *abort_flag = pa->abort_flag;
      int go = 0;
      const long long t0 = wall_clock64();
      for (;;) {
        // (relaxed polls, one acquire once the row is there: see queue_kernel)
        if (__hip_atomic_load(yseq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) >= it) {
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          go = 1;
          break;
        }
        if (wait_ticks <= 0 || wall_clock64() - t0 > wait_ticks) break;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
        __builtin_amdgcn_s_sleep(16);
      }
#if BORE_LAG_PRIO
      // A loop more than 3/4 of an iteration AHEAD of the grid's mean spends its slack here, asleep, and
      // leaves the CU's issue slots to its partner (the region ends with the slowest loop: a leader gains
      // nothing by arriving early): until the mean has come within 3/4 of an iteration, 500 us at most per
      // iteration.  Measured (profiles/r4/ab_log.txt): 489.0 -> 497.2 k it/s at 20 steps, 376.6 -> 387.0 k at
      // 100; thresholds of 1/4 and 1/2 cost at 20 steps, 1 and 2 gain less.  Same trajectories.
      if (go) {
        const long long t1 = wall_clock64();
        const int n_wg = (int)gridDim.x;
        for (;;) {
          const int total = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (4 * ((it - it_first) * n_wg - total) <= BORE_LAG_YIELD_Q * n_wg || wall_clock64() - t1 > 50000) break;
          __builtin_amdgcn_s_sleep(64);
        }
      }
#endif
      // (after this store the workgroup touches nothing of the loop: the host may relaunch it)
      if (!go) __hip_atomic_store(pa->parked + lid, it, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
      s_go4[0] = go;
    }
    __syncthreads();
    const int go = uniform_i32(s_go4[0]);
    __syncthreads();
    if (!go) break;
  }
}

// MORE loops than the device holds workgroups (> 512 for this model): a fixed grid of workgroups stays on
// the device for a whole run and serves WHICHEVER loop-iteration is ready next -- a work queue in pinned
// host memory.  The host appends (loop, iteration) entries in ticket order as loops become ready (their
// newest row is in ynew[loop]); a workgroup draws a ticket from a device-side counter, waits until the
// host has written that entry, runs the iteration, publishes the result through the loop's flag as the
// resident kernel does, and draws again.  No launch, no upload and no batch per loop-iteration (round 3:
// one launch per batch of ready loops, ~13 % below the device's rate at 4 096 loops), and the workgroups
// balance themselves over the loops.  An entry with lid < 0 ends a workgroup.
template <int SHAPE>
__global__ __launch_bounds__(BORE_THREADS, 2) void queue_kernel(const IterArgs *__restrict__ pa) {
  __shared__ __attribute__((aligned(16))) int s_q4[4];  // (16 B: the dynamic LDS keeps its alignment)
  for (;;) {
    if (threadIdx.x == 0) {
      const QueueEntry *ring = pa->q_ring;
      const int *abort_flag = pa->abort_flag;
      const unsigned long long t =
          __hip_atomic_fetch_add(pa->q_head, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const QueueEntry *e = ring + (t & (unsigned long long)pa->q_mask);
      int lid = -1, it = 0;
      for (;;) {
        // (relaxed polls, ONE acquire when the entry is there: an acquire per poll is a cache invalidate per
        // poll, from every waiting workgroup)
        if (__hip_atomic_load(&e->seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == (long long)(t + 1)) {
          __atomic_thread_fence(__ATOMIC_ACQUIRE);
          lid = __hip_atomic_load(&e->lid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          it = __hip_atomic_load(&e->it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;  // (lid stays -1)
        __builtin_amdgcn_s_sleep(16);
      }
      s_q4[0] = lid;
      s_q4[1] = it;
    }
    __syncthreads();
    const int lid = uniform_i32(s_q4[0]), it = uniform_i32(s_q4[1]);
    __syncthreads();
    if (lid < 0) break;
    // (the argument block through the constant address space, its address opaque per iteration: see the
    // resident kernel; ids[] is the identity here, so the loop IS the slot, and the row an iteration
    // appends comes from ynew[loop])
    unsigned long long pa_bits = reinterpret_cast<unsigned long long>(pa);
    asm volatile("" : "+s"(pa_bits));
    typedef const __attribute__((address_space(4))) IterArgs *IterArgsConst;
    iteration_once<SHAPE>((const IterArgs *)(IterArgsConst)pa_bits, (long long)lid, it, false);
    __syncthreads();  // the waves leave the restart phase one by one: LDS is reused by the next iteration
  }
}

// One fused launch for the batch currently set (bore_set_batch): fills *h (pinned host copy of the
// arguments, at the head of a staging block of upload_bytes that also holds x_new / y_new / ids /
// its), uploads the block to d_args and launches.
// Returns BORE_E_UNSUPPORTED when the model is not static shape 1: the caller falls back to the
// launch chain.
static int iteration_supported(const bore_mlp_desc *desc) {
  return desc->compute == BORE_COMPUTE_F32 && bore_kernel_flavour(desc, true) == 1;
}

static int iteration_launch(const bore_mlp_desc *desc, int n_slots, float *theta, float *adam_m,
                            float *adam_v, int64_t *adam_t, double *X_seen, double *y_seen,
                            float *X32, float *z, const double *x_new, const double *y_new,
                            double gamma, int epochs, int batch_size, uint64_t seed,
                            int64_t loop_id0, const bore_adam_cfg *adam, int64_t num_samples,
                            const double *low, const double *high, int num_starts, int transform,
                            const bore_lbfgsb_opts *opts, double *x0, int32_t *idx, double *x,
                            double *fun, double *jac, int32_t *info, IterArgs *h,
                            const IterArgs *d_args, size_t upload_bytes, void *stream,
                            int queue_wgs = 0, const QueueEntry *q_ring = nullptr,
                            unsigned long long *q_head = nullptr, int q_mask = 0) {
  if (!g_batch) return fail(BORE_E_INVALID, "iteration_launch: no batch set");
  const int64_t cap = g_batch->cap;
  size_t lf = 0, ls = 0, lb = 0;
  int sf = 0, ss = 0, sb = 0, blocks = 0;
  int rc = fit_build(desc, n_slots, theta, adam_m, adam_v, adam_t, X32, z, cap, epochs, batch_size,
                     nullptr, seed, loop_id0, 0, adam, nullptr, h->f, lf, sf);
  if (rc) return rc < 0 ? rc : fail(BORE_E_INVALID, "iteration_launch: epochs must be positive");
  const SampleSpec spec{seed, loop_id0, 0, low, high};
  if ((rc = screen_build(desc, n_slots, theta, nullptr, &spec, num_samples, 0, num_starts, x0, idx,
                         nullptr, h->s, ls, ss)))
    return rc;
  if ((rc = lbfgsb_build(desc, n_slots, theta, transform, 1, x0, num_starts, low, high, opts, x, fun,
                         jac, info, h->b, lb, sb, blocks)))
    return rc;
  if (sf != 1 || ss != 1 || sb != 1 || blocks != 1)
    return fail(BORE_E_UNSUPPORTED, "iteration_launch: static shape 1 only");
  h->X_seen = X_seen; h->y_seen = y_seen; h->X32 = X32; h->z = z;
  h->x_new = x_new; h->y_new = y_new;
  h->stamps = reinterpret_cast<long long *>(g_batch->stamps);
  h->b.stamps = h->stamps;
  h->targets = g_batch->targets;
  h->ynew = g_batch->ynew; h->yseq = g_batch->yseq; h->parked = g_batch->parked;
  h->abort_flag = g_batch->abort_flag;
  const bool resident = g_batch->targets && g_batch->ynew && g_batch->yseq && g_batch->parked &&
                        g_batch->abort_flag;
  if (!resident) h->targets = nullptr;  // one iteration per launch
  h->wait_ticks = resident ? g_batch->wait_ticks : 0;
  h->gamma = gamma;
  h->D = desc->input_dim;
  h->progress = 0;
  h->q_ring = q_ring; h->q_head = q_head; h->q_mask = q_mask;
  size_t floats = lf > ls ? lf : ls;
  floats = floats > lb ? floats : lb;
  const size_t labels_floats = 2 * ((size_t)cap + 2);
  floats = floats > labels_floats ? floats : labels_floats;
  if ((rc = allow_lds(iteration_kernel<1, true>, floats * 4)) ||
      (rc = allow_lds(iteration_kernel<1, false>, floats * 4)) || (rc = allow_lds(queue_kernel<1>, floats * 4)))
    return rc;
  if (queue_wgs > 0) {  // the work-queue form: a fixed grid, fed by the host through q_ring
    if (!q_ring || !q_head || !g_batch->ynew || !g_batch->abort_flag)
      return fail(BORE_E_INVALID, "iteration_launch: incomplete work queue");
    h->targets = nullptr;
    h->wait_ticks = 0;
    HIP_TRY(hipMemcpyAsync(const_cast<IterArgs *>(d_args), h, sizeof(IterArgs), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
    hipLaunchKernelGGL((queue_kernel<1>), dim3(queue_wgs), dim3(BORE_THREADS), floats * 4, (hipStream_t)stream,
                       d_args);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (h->wait_ticks > 0) {  // waiting workgroups hold their slots: only when all of them fit at once
    static thread_local size_t cap_bytes = ~(size_t)0;
    static thread_local int cap_wgs = 0, cap_dev = -1;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    if (cap_bytes != floats * 4 || cap_dev != dev) {  // (per device: a thread may drive several)
      int per_cu = 0, cus = 0;
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, iteration_kernel<1, true>, BORE_THREADS,
                                                           floats * 4));
      HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
      cap_bytes = floats * 4;
      cap_dev = dev;
      cap_wgs = per_cu * cus;
    }
    if (g_batch->resident_loops > cap_wgs) h->wait_ticks = 0;
  }
  // h heads the caller's staging block (arguments | per-slot inputs | index lists): one upload
  HIP_TRY(hipMemcpyAsync(const_cast<IterArgs *>(d_args), h, upload_bytes, hipMemcpyHostToDevice,
                         (hipStream_t)stream));
  if (h->wait_ticks > 0)
    hipLaunchKernelGGL((iteration_kernel<1, true>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                       (hipStream_t)stream, d_args);
  else
    hipLaunchKernelGGL((iteration_kernel<1, false>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                       (hipStream_t)stream, d_args);
  HIP_TRY(hipGetLastError());
  return 0;
}
