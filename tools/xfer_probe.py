"""Cost of the small host<->device copies the drop-in model API makes per call (GPU box): pageable .to() / .cpu()
against a persistent pinned staging buffer with asynchronous copies."""
import time, numpy as np, torch
dev = torch.device("cuda", 0)
torch.zeros(1, device=dev)
def t(f, n=300):
    for _ in range(20): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / n
for nbytes in (256, 8192, 65536):
    a = np.random.rand(nbytes // 8)
    d = torch.from_numpy(a).to(dev)
    pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
    out = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    def up_page(): return torch.from_numpy(a).to(dev)
    def up_pin():
        pin.numpy()[...] = a.view(np.uint8)
        o = torch.empty(nbytes, dtype=torch.uint8, device=dev); o.copy_(pin, non_blocking=True); return o
    def down_page(): return d.cpu().numpy()
    def down_pin():
        pin.copy_(d.view(torch.uint8), non_blocking=True); torch.cuda.current_stream().synchronize()
        return pin.numpy().view(np.float64).copy()
    print(f"{nbytes:6d} B: upload pageable {t(up_page):6.1f} us, pinned+async {t(up_pin):6.1f} us; download .cpu() {t(down_page):6.1f} us, pinned+sync {t(down_pin):6.1f} us")
