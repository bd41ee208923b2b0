"""What plain streaming kernels reach on this box (torch): read-only reduction, copy, and a 4:1 read:write mix like
pyrDown's (read N bytes, write N/4) — the practical ceilings the pyramid kernel is compared with."""
import torch
dev = torch.device("cuda", 0)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
n = 629 * 1024 * 1024 // 16 * 16
x = torch.randint(0, 255, (n // 4,), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
t = timed(lambda: y.copy_(x)); print(f"copy {n/1e6:.0f} MB: {2*n/t/1e12:.2f} TB/s (read+write)")
t = timed(lambda: x.sum()); print(f"read-only sum {n/1e6:.0f} MB: {n/t/1e12:.2f} TB/s")
x4 = x.view(-1, 4); y4 = torch.empty((x4.shape[0],), dtype=torch.int32, device=dev)
t = timed(lambda: torch.sum(x4, dim=1, out=y4)); print(f"4:1 read:write (row sums of 4 ints): {(n + n/4)/t/1e12:.2f} TB/s")
xf = x.view(torch.float32)
t = timed(lambda: torch.mul(xf, 2.0, out=y.view(torch.float32))); print(f"scale (1:1): {2*n/t/1e12:.2f} TB/s")
