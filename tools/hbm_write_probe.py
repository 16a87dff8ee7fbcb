"""What HBM gives a write-only / read-only / copy stream of the Jacobian build's size on this box (torch: plumbing only).
usage: python tools/hbm_write_probe.py"""
import torch, time
n = int(2.77e9 // 8)
x = torch.empty(n, dtype=torch.float64, device="cuda"); y = torch.empty(n, dtype=torch.float64, device="cuda")
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
w = t(lambda: x.zero_()); r = t(lambda: x.sum()); c = t(lambda: y.copy_(x))
print({"bytes": n * 8, "write_only_ms": round(w, 4), "write_TBps": round(n * 8 / w / 1e9, 3), "read_only_ms": round(r, 4), "read_TBps": round(n * 8 / r / 1e9, 3),
       "copy_ms": round(c, 4), "copy_TBps_rw": round(2 * n * 8 / c / 1e9, 3)})
