"""Practical HBM ceilings of this box with plain torch kernels: pure write (fill), pure read (sum), copy."""
import torch
dev = torch.device("cuda:0")
n = 1 << 29                       # 2 GiB of fp32
a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
w = t(lambda: a.fill_(1.5)); r = t(lambda: a.sum()); c = t(lambda: b.copy_(a))
print("write (fill 2 GiB): %.2f TB/s   read (sum 2 GiB): %.2f TB/s   copy (2+2 GiB): %.2f TB/s" % (4 * n / w / 1e12, 4 * n / r / 1e12, 8 * n / c / 1e12))
