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
// Static shape 1 (2 -> 16-16-1, the BASELINE config) and, round 6, static shape 5 at its full 16 inputs (16 ->
// 32-32-32-1: the network the reference's only in-repo caller builds, bore/plugins/hpbandster/base.py:23-33 ->
// bore/models.py:16-19); anything else keeps the launch chain.
// Included by bore_all.hip after bore_hip.hip and bore_argmax.hip.

// One entry of the work queue of queue_kernel (pinned host memory, written by the host in ticket order and
// published by the queue's tail counter, IterArgs::q_tail_host): the loop and its iteration, one 64-bit word.
struct QueueEntry {
  int lid, it;
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
  // ... how many entries the host has published (pinned; all ones = no more work, exit) and the device's copy of
  // that number (device memory, zeroed by the host before the launch: the 64-bit word after q_head)
  const unsigned long long *q_tail_host;
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
// a leader yields while it is more than Q / 4 iterations ahead of the mean (iteration_kernel; the A/B of the wave
// priorities and of the yielding: profiles/r4/ab_log.txt -- closed, the switch BORE_LAG_PRIO is gone)
#define BORE_LAG_YIELD_Q 3
// One phase hands its results to the next through global memory (labels z, the fitted theta, the screened x0),
// written and read by waves of the SAME workgroup: the writers' stores are acknowledged (the vector L1 writes
// through to the XCD's L2), barrier, and the CU's L1 drops what it fetched before (buffer_inv sc0: this CU's
// lines alone).  Rounds 2 - 4 put an agent-scope fence here -- on gfx950, whose eight XCDs have an L2 each, that
// is a write-back of the XCD's L2 plus an invalidate of it, four times per loop-iteration from every workgroup.
__device__ __forceinline__ void phase_handoff() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  asm volatile("buffer_inv sc0\n\ts_waitcnt vmcnt(0)" ::: "memory");
}

// LOCAL: everything this loop's state has been through stayed on this XCD (see lbfgsb_body's PUBLISH).
template <int SHAPE, bool LOCAL = false>
__device__ __forceinline__ void iteration_once(const IterArgs *__restrict__ pa,
                                                         const long long slot, const int it,
                                                         const bool staged, const long long t_begin = 0) {
  const IterArgs &a = *pa;
  const long long lid = uniform_i64(a.f.ids[slot]), cap = a.f.cap;
  long long *stp = a.stamps ? a.stamps + lid * 4 : nullptr;  // read back by the restart phase's epilogue
  // (t_begin != 0, -DBORE_QUEUE_STAMP_DRAW: the first phase is charged from the moment the workgroup drew its
  // ticket -- the wait for the queue entry shows up in `labels`)
  if (stp && threadIdx.x == 0) stp[0] = t_begin ? t_begin : wall_clock64();
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
  if constexpr (LOCAL) phase_handoff();
  else {
    __threadfence();
    __syncthreads();
  }
  labels_body(a.y_seen, 0, 0.0, a.z, nullptr, a.f.ids, a.f.its, a.f.n_init, cap, a.gamma, slot, it);
  if (stp && threadIdx.x == 0) stp[1] = wall_clock64();
  if constexpr (LOCAL) phase_handoff();
  else {
    __threadfence();
    __syncthreads();
  }
  fit_body<SHAPE>(a.f, slot, it);
  if (stp && threadIdx.x == 0) stp[2] = wall_clock64();
  if constexpr (LOCAL) phase_handoff();
  else {
    __threadfence();
    __syncthreads();
  }
  screen_body<SHAPE, false>(a.s, slot, it);
  if (stp && threadIdx.x == 0) stp[3] = wall_clock64();
  if constexpr (LOCAL) phase_handoff();
  else {
    __threadfence();
    __syncthreads();
  }
  lbfgsb_body<SHAPE, false, true, true, false, LOCAL ? 1 : 0>(a.b, slot, 0, it);  // (publishes flag[lid] = it + 1)
}

// RESIDENT: the workgroup may go on to later iterations of its loop.  Otherwise ONE iteration -- the
// launch for loops that do not wait on their CU (more loops than the device holds at once,
// wait_ticks = 0): no loop, no scratch at all.
// OCC: loops (workgroups) per CU the kernel is compiled for.  2: the whole register file of two waves per SIMD (256
// registers, no scratch) -- the kernel for up to 2 x CUs loops.  3: 168 registers, the restart phase pays with spills
// to scratch (its per-loop time +7 %), and a CU runs three loops: the kernel for more loops than that (round 5;
// 768 loops 549 -> 606 k it/s).
template <int SHAPE, bool RESIDENT, int OCC = 2>
__global__ __launch_bounds__(BORE_THREADS, OCC) void iteration_kernel(const IterArgs *__restrict__ pa) {
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
  for (int it = it_first;;) {
    {
      const int total = __hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int n_wg = (int)gridDim.x;
      const int lead = uniform_i32((it - it_first) * n_wg - total);  // > 0: ahead of the mean, in 1 / n_wg iterations
      if (lead < -(n_wg >> 1)) __builtin_amdgcn_s_setprio(3);
      else if (lead > (n_wg >> 1)) __builtin_amdgcn_s_setprio(0);
      else __builtin_amdgcn_s_setprio(1);
    }
    // (the argument block's address is made opaque per iteration, so that nothing read through it
    // is hoisted out of the loop and kept live around it)
    // (the block is written by the host before the launch and never by a kernel: read through the
    // constant address space its fields are scalar loads -- uniform values in scalar registers, loops
    // over them scalar loops -- instead of vector loads that a store earlier in the loop might clobber)
    unsigned long long pa_bits = reinterpret_cast<unsigned long long>(pa);
    asm volatile("" : "+s"(pa_bits));
    typedef const __attribute__((address_space(4))) IterArgs *IterArgsConst;
    // (LOCAL: workgroup-scope hand-overs; rounds 2 - 4's agent / system-scope fences lost the A/B, profiles/r5/ab_log.txt)
    iteration_once<SHAPE, true>((const IterArgs *)(IterArgsConst)pa_bits, slot, it, it == it_first);
    __syncthreads();  // the waves leave the restart phase one by one: LDS is reused below
    ++it;
    if (threadIdx.x == 0) __hip_atomic_fetch_add(prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
          // (the row is read with system-scope loads that bypass the caches, load_host_f64: no invalidate --
          // a system-scope acquire here is an invalidate of the XCD's L2 under every other workgroup)
          asm volatile("" ::: "memory");
          go = 1;
          break;
        }
        if (wait_ticks <= 0 || wall_clock64() - t0 > wait_ticks) break;
        if (__hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) break;
        __builtin_amdgcn_s_sleep(16);
      }
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
#ifdef BORE_QUEUE_STAMP_DRAW
// (diagnostics: when each workgroup of the last queue launch started, and where -- HW_ID, XCC_ID)
__device__ long long g_wg_start[4096][2];
extern "C" int bore_debug_wg_starts(long long *out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_start), sizeof(long long) * 4096 * 2);
}
#endif
template <int SHAPE, int OCC = 2>
__global__ __launch_bounds__(BORE_THREADS, OCC) void queue_kernel(const IterArgs *__restrict__ pa) {
  __shared__ __attribute__((aligned(16))) int s_q4[4];  // (16 B: the dynamic LDS keeps its alignment)
#ifdef BORE_QUEUE_STAMP_DRAW
  if (threadIdx.x == 0 && blockIdx.x < 4096) {
    g_wg_start[blockIdx.x][0] = wall_clock64();
    const unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
    const unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));  // HW_REG_XCC_ID
    g_wg_start[blockIdx.x][1] = ((long long)xcc << 32) | hw;
  }
#endif
  for (;;) {
    if (threadIdx.x == 0) {
#ifdef BORE_QUEUE_STAMP_DRAW
      const long long t_draw = wall_clock64();
      s_q4[2] = (int)(t_draw & 0xffffffffll);
      s_q4[3] = (int)(t_draw >> 32);
#endif
      const unsigned long long t =
          __hip_atomic_fetch_add(pa->q_head, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#ifdef BORE_QUEUE_STALL_TEST
      // (test build, tools/queue_stall_test.sh: the workgroup that draws ticket BORE_QUEUE_STALL_TEST sits on it,
      // its entry UNREAD, for ~25 ms -- some ten thousand tickets of everybody else -- while the host's ring, which the
      // same macro shrinks to the smallest legal size, comes round to that slot again and again.  What keeps the entry
      // from being overwritten under it is the host's rule that a slot is written again only once the entry it held
      // has been ANSWERED: bore_engine.hip, async_run_queue's push.)
      if (t == (unsigned long long)(BORE_QUEUE_STALL_TEST))
        for (int z = 0; z < 8000; ++z) __builtin_amdgcn_s_sleep(127);
#endif
      // Waiting for entry t.  Round 4: every waiting workgroup polled the entry's sequence number in pinned memory --
      // a read over the bus every ~2 us from each of them.  With hundreds of workgroups idle (fewer ready loops than
      // workgroups; the end of every run) those reads saturate the link, and everything else that crosses it -- the
      // rows the busy workgroups fetch, their results, their flags -- queues behind them: 1 024 loops on 768
      // workgroups ran at 0.17 M it/s, and every run() ended with ~20 ms of it (profiles/r5/ab_log.txt).  Now the
      // host publishes a tail counter, the device keeps a copy of it, and only the workgroup at the HEAD of the line
      // (ticket == copy: tickets are unique, so there is one) asks the host; it moves the copy on for everybody.  The
      // others watch the copy, and the further back they stand the longer they sleep between looks.
      unsigned long long *tail_dev = pa->q_head + 1;
      const unsigned long long *tail_host = pa->q_tail_host;
      int lid = -1, it = 0;
      for (;;) {
        unsigned long long tail = __hip_atomic_load(tail_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tail == t) {
          const unsigned long long th = __hip_atomic_load(tail_host, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          if (th > t) {
            __hip_atomic_fetch_max(tail_dev, th, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            tail = th;
          }
        }
        if (tail == ~0ull) break;  // (the run is over, or aborted: lid stays -1)
        if (tail > t) {
          // (the entry was written before the tail that covers it, both by the host; read past the caches)
          const long long w = __hip_atomic_load(reinterpret_cast<const long long *>(pa->q_ring + (t & (unsigned long long)pa->q_mask)),
                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          lid = (int)(w & 0xffffffffll);
          it = (int)(w >> 32);
          // (the loop's previous iteration may have run on another XCD: its state went through the system-scope
          // release of lbfgsb_body's PUBLISH = 0; what this XCD's L2 and this CU's L1 hold of it from an earlier
          // visit is dropped here, by one wave, ahead of the barrier that lets the others go)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          break;
        }
        // (entries arrive at ~0.8 per us when the device is full: a workgroup d places back is d us or more away)
        const int back = (int)(t - tail < 64ull ? t - tail : 64ull);
        __builtin_amdgcn_s_sleep(8);
        for (int z = 0; z < back; ++z) __builtin_amdgcn_s_sleep(20);
      }
      s_q4[0] = lid;
      s_q4[1] = it;
    }
    __syncthreads();
    const int lid = uniform_i32(s_q4[0]), it = uniform_i32(s_q4[1]);
#ifdef BORE_QUEUE_STAMP_DRAW
    const long long t_begin = ((long long)s_q4[3] << 32) | (unsigned)s_q4[2];
#else
    const long long t_begin = 0;
#endif
    __syncthreads();
    if (lid < 0) break;
    // (the argument block through the constant address space, its address opaque per iteration: see the
    // resident kernel; ids[] is the identity here, so the loop IS the slot, and the row an iteration
    // appends comes from ynew[loop])
    unsigned long long pa_bits = reinterpret_cast<unsigned long long>(pa);
    asm volatile("" : "+s"(pa_bits));
    typedef const __attribute__((address_space(4))) IterArgs *IterArgsConst;
    iteration_once<SHAPE>((const IterArgs *)(IterArgsConst)pa_bits, (long long)lid, it, false, t_begin);
    __syncthreads();  // the waves leave the restart phase one by one: LDS is reused by the next iteration
  }
}

// One fused launch for the batch currently set (bore_set_batch): fills *h (pinned host copy of the
// arguments, at the head of a staging block of upload_bytes that also holds x_new / y_new / ids /
// its), uploads the block to d_args and launches.
// Returns BORE_E_UNSUPPORTED when the model is not static shape 1: the caller falls back to the
// launch chain.
// The static shape whose fused kernels run the model, or 0 (the caller keeps the launch chain).  Shape 5 only at its
// 16 compiled inputs: with fewer the fit runs zero-padded through a repacked copy (fit_padded), which the fused
// kernel has no phase for.
static int iteration_shape(const bore_mlp_desc *desc) {
  if (desc->compute != BORE_COMPUTE_F32) return 0;
  const int f = bore_kernel_flavour(desc, true);
  if (f == 1) return 1;
  if (f == 5 && desc->input_dim == 16 && bore_flavour_built(5)) return 5;
  return 0;
}
static int iteration_supported(const bore_mlp_desc *desc) { return iteration_shape(desc) != 0; }

// Loops (workgroups) of the fused kernels one CU holds at a time with `lds_bytes` of dynamic LDS each: what the
// runtime's occupancy query says (registers, waves), and no more than the LDS allows at its allocation granularity
// -- 1 280 B on gfx950, which the query does not apply: at 54 240 B it answers 3 where the hardware places 2, and
// a resident grid sized by that answer waits for workgroups that never start (331 k it/s instead of 560 k).
static int iteration_loops_per_cu(int shape, size_t lds_bytes, int *per_cu_out) {
  static thread_local size_t seen_bytes = ~(size_t)0;
  static thread_local int seen_per_cu = 0, seen_dev = -1, seen_shape = 0;
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (seen_bytes != lds_bytes || seen_dev != dev || seen_shape != shape) {  // (per device: a thread may drive several)
    int per_cu = 0, q_per_cu = 0, most = 3;
#if BORE_ON_5
    if (shape == 5) {  // (one build: two loops per CU, the whole register file of two waves per SIMD)
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, iteration_kernel<5, true, 2>, BORE_THREADS, lds_bytes));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q_per_cu, queue_kernel<5, 2>, BORE_THREADS, lds_bytes));
      most = 2;
    } else
#endif
    {
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, iteration_kernel<1, true, 3>, BORE_THREADS, lds_bytes));
      HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&q_per_cu, queue_kernel<1, 3>, BORE_THREADS, lds_bytes));
    }
    if (q_per_cu < per_cu) per_cu = q_per_cu;
    const size_t granule = 1280, static_bytes = 16;  // (s_go4 / s_q4)
    const size_t each = (lds_bytes + static_bytes + granule - 1) / granule * granule;
    const int by_lds = (int)(BORE_LDS_BYTES / each);
    if (by_lds < per_cu) per_cu = by_lds;
    if (per_cu > most) per_cu = most;  // (what the kernels are compiled for)
    seen_bytes = lds_bytes; seen_dev = dev; seen_per_cu = per_cu; seen_shape = shape;
  }
  *per_cu_out = seen_per_cu;
  return 0;
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
                            unsigned long long *q_head = nullptr, int q_mask = 0, int *per_cu_out = nullptr,
                            const unsigned long long *q_tail_host = nullptr) {
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
  const int shape = iteration_shape(desc);
  if (!shape || sf != shape || ss != shape || sb != shape || blocks != 1)
    return fail(BORE_E_UNSUPPORTED, "iteration_launch: static shape 1, or 5 at 16 inputs, one workgroup per loop");
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
  h->q_ring = q_ring; h->q_head = q_head; h->q_mask = q_mask; h->q_tail_host = q_tail_host;
  size_t floats = lf > ls ? lf : ls;
  floats = floats > lb ? floats : lb;
  const size_t labels_floats = 2 * ((size_t)cap + 2);
  floats = floats > labels_floats ? floats : labels_floats;
  if (shape == 1) {
    if ((rc = allow_lds(iteration_kernel<1, true, 2>, floats * 4)) || (rc = allow_lds(iteration_kernel<1, true, 3>, floats * 4)) ||
        (rc = allow_lds(iteration_kernel<1, false, 2>, floats * 4)) || (rc = allow_lds(queue_kernel<1, 2>, floats * 4)) ||
        (rc = allow_lds(queue_kernel<1, 3>, floats * 4)))
      return rc;
  }
#if BORE_ON_5
  if (shape == 5) {
    if ((rc = allow_lds(iteration_kernel<5, true, 2>, floats * 4)) || (rc = allow_lds(iteration_kernel<5, false, 2>, floats * 4)) ||
        (rc = allow_lds(queue_kernel<5, 2>, floats * 4)))
      return rc;
  }
#endif
  if (per_cu_out) {  // (a question, not a launch: how many loops of this model one CU holds)
    *per_cu_out = 0;
    return iteration_loops_per_cu(shape, floats * 4, per_cu_out);
  }
  if (queue_wgs > 0) {  // the work-queue form: a fixed grid, fed by the host through q_ring
    if (!q_ring || !q_head || !q_tail_host || !g_batch->ynew || !g_batch->abort_flag)
      return fail(BORE_E_INVALID, "iteration_launch: incomplete work queue");
    h->targets = nullptr;
    h->wait_ticks = 0;
    HIP_TRY(hipMemcpyAsync(const_cast<IterArgs *>(d_args), h, sizeof(IterArgs), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
    // (no more workgroups than two per CU: the kernel with the whole register file)
#if BORE_ON_5
    if (shape == 5)
      hipLaunchKernelGGL((queue_kernel<5, 2>), dim3(queue_wgs), dim3(BORE_THREADS), floats * 4, (hipStream_t)stream, d_args);
    else
#endif
    if (queue_wgs <= 2 * device_cus())
      hipLaunchKernelGGL((queue_kernel<1, 2>), dim3(queue_wgs), dim3(BORE_THREADS), floats * 4, (hipStream_t)stream, d_args);
    else
      hipLaunchKernelGGL((queue_kernel<1, 3>), dim3(queue_wgs), dim3(BORE_THREADS), floats * 4, (hipStream_t)stream, d_args);
    HIP_TRY(hipGetLastError());
    return 0;
  }
  if (h->wait_ticks > 0) {  // waiting workgroups hold their slots: only when all of them fit at once
    int per_cu = 0;
    if ((rc = iteration_loops_per_cu(shape, floats * 4, &per_cu))) return rc;
    if (getenv("BORE_ASYNC_DEBUG"))
      fprintf(stderr, "[bore] fused kernel: %zu B of LDS per loop (fit %zu, screen %zu, restarts %zu, labels %zu), %d loops per CU\n", floats * 4, lf * 4, ls * 4, lb * 4, labels_floats * 4, per_cu);
    if (g_batch->resident_loops > per_cu * device_cus()) h->wait_ticks = 0;
  }
  // h heads the caller's staging block (arguments | per-slot inputs | index lists): one upload
  HIP_TRY(hipMemcpyAsync(const_cast<IterArgs *>(d_args), h, upload_bytes, hipMemcpyHostToDevice,
                         (hipStream_t)stream));
#if BORE_ON_5
  if (shape == 5) {
    if (h->wait_ticks > 0)
      hipLaunchKernelGGL((iteration_kernel<5, true, 2>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                         (hipStream_t)stream, d_args);
    else
      hipLaunchKernelGGL((iteration_kernel<5, false, 2>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                         (hipStream_t)stream, d_args);
    HIP_TRY(hipGetLastError());
    return 0;
  }
#endif
  if (h->wait_ticks > 0 && g_batch->resident_loops > 2 * device_cus())  // (three loops per CU: see iteration_kernel)
    hipLaunchKernelGGL((iteration_kernel<1, true, 3>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                       (hipStream_t)stream, d_args);
  else if (h->wait_ticks > 0)
    hipLaunchKernelGGL((iteration_kernel<1, true, 2>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                       (hipStream_t)stream, d_args);
  else
    hipLaunchKernelGGL((iteration_kernel<1, false, 2>), dim3(n_slots), dim3(BORE_THREADS), floats * 4,
                       (hipStream_t)stream, d_args);
  HIP_TRY(hipGetLastError());
  return 0;
}
