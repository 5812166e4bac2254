import numpy as np
rec = np.load('gpurun_out/nit_nfev.npy')
S, L = rec.shape[0], rec.shape[1]
cost = (52e3*rec[...,0] + 6e3*rec[...,1]).max(axis=2) / 2.4e3 * 1.3
def fit_us(k):
    N = 13 + k
    return (200 * (-(-N // 64))) * 2.5 + 60
fit_tab = np.array([fit_us(k) for k in range(S + 1)])
def sim(h, Bmin, wait, tick=10.0):
    it = np.zeros(L, int); done_at = np.full(L, -1.0)     # -1: ready now; else completion time
    state = np.zeros(L, int)                                # 0 ready, 1 in flight, 2 finished
    ready_since = np.zeros(L)
    now = 0.0; nb = 0
    while (state != 2).any():
        fl = np.nonzero((state == 1) & (done_at <= now))[0]
        if len(fl):
            it[fl] += 1
            fin = fl[it[fl] >= S]; state[fin] = 2
            go = fl[it[fl] < S]; state[go] = 0; ready_since[go] = now
        r = np.nonzero(state == 0)[0]
        if len(r) and (len(r) >= Bmin or now - ready_since[r].min() >= wait or not (state == 1).any()):
            now += h; nb += 1
            f = fit_tab[it[r]].max()
            done_at[r] = now + f + 40 + cost[it[r], r] + 20
            state[r] = 1
        else:
            now += tick
    return now, nb
for h in (30, 60, 100, 200):
    for Bmin, wait in ((32, 100), (64, 150), (128, 300), (256, 400), (512, 1e9)):
        t, nb = sim(h, Bmin, wait)
        print(f"host {h:4d} us/batch, Bmin {Bmin:4d} wait {wait}: {t/S/1e3:6.2f} ms/step-eq -> {L*S/(t*1e-6):9.0f} it/s, batches/step {nb/S:.1f}, mean batch {L*S/nb:.0f}")
