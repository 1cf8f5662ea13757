"""What a plain device-to-device copy / read / write reaches on this box (context for the per-kernel GB/s figures): python3 tools/ubench/copy_bw.py"""
import torch
n = 1 << 30                                   # 4 GiB of float32 each way
a = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
b = torch.empty_like(a)
def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
s = t(lambda: b.copy_(a));            print(f"copy   (4 GiB read + 4 GiB written): {2 * 4 * n / s / 1e12:.2f} TB/s")
s = t(lambda: a.sum());               print(f"read   (sum of 4 GiB):               {4 * n / s / 1e12:.2f} TB/s")
s = t(lambda: b.fill_(1.0));          print(f"write  (fill of 4 GiB):              {4 * n / s / 1e12:.2f} TB/s")
s = t(lambda: torch.add(a, 1.0, out=b)); print(f"add    (read 4 GiB, write 4 GiB):    {2 * 4 * n / s / 1e12:.2f} TB/s")
c = torch.empty(n // 2, dtype=torch.float32, device="cuda")
s = t(lambda: torch.add(a[: n // 2], a[n // 2:], out=c)); print(f"2 reads : 1 write (4 + 2 GiB):       {6 * n / s / 1e12:.2f} TB/s")
