import numpy as np, heapq
rec = np.load('gpurun_out/nit_nfev.npy')           # [steps, 512, 3, 2]
S, L = rec.shape[0], rec.shape[1]
cost = (52e3*rec[...,0] + 6e3*rec[...,1]).max(axis=2) / 2.4e3 * 1.3   # per-loop lbfgsb us (scaled to match measured)
def fit_us(k):                                      # iteration k -> N = 13 + k
    N = 13 + k
    return (200 * (-(-N // 64))) * 2.5 + 60
def sim(h, Bmin, poll, sync_groups=None):
    # event-driven: returns total time (us) for all loops to finish S iterations
    if sync_groups:
        G = sync_groups; t = np.zeros(G); idx = np.array_split(np.arange(L), G)
        for k in range(S):
            for g in range(G):
                t[g] += fit_us(k) + cost[k, idx[g]].max() + 150
        return t.max()
    it = np.zeros(L, int); ready_t = np.zeros(L)    # loop l is ready for iteration it[l] at ready_t[l]
    now = 0.0; done = 0
    ready = {0: list(range(L))}
    inflight = []                                    # heap of (time, loop)
    host_free = 0.0
    while done < L:
        # choose a batch: iteration with most ready loops
        launched = False
        if ready:
            k = max(ready, key=lambda kk: len(ready[kk]))
            n_in_k_flight = sum(1 for (_, l) in inflight if it[l] + 1 == k + 0)  # loops that will become ready for iteration k
            if len(ready[k]) >= Bmin or not inflight:
                loops = ready.pop(k)
                now = max(now, host_free) + h
                host_free = now
                f = fit_us(k)
                for l in loops:
                    heapq.heappush(inflight, (now + f + cost[k, l] + poll, l))
                launched = True
        if not launched:
            tdone, l = heapq.heappop(inflight)
            now = max(now, tdone)
            it[l] += 1
            if it[l] >= S: done += 1
            else: ready.setdefault(it[l], []).append(l)
            # drain everything else already complete
            while inflight and inflight[0][0] <= now:
                _, l2 = heapq.heappop(inflight)
                it[l2] += 1
                if it[l2] >= S: done += 1
                else: ready.setdefault(it[l2], []).append(l2)
    return now
base = sim(0, 0, 0, sync_groups=4)
print("sync G=4 model: %.1f ms/step -> %.0f it/s" % (base/S/1e3, L*S/(base*1e-6)))
for h in (20, 50, 100, 200):
    for Bmin in (16, 32, 64, 128):
        t = sim(h, Bmin, 20)
        print(f"host {h:4d} us/batch, Bmin {Bmin:4d}: {t/S/1e3:6.2f} ms/step-equivalent -> {L*S/(t*1e-6):9.0f} it/s")
