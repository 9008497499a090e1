import torch, time
n = 822_000_000
a = torch.empty(n, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
a.random_(0, 255)
for _ in range(3): b.copy_(a)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20): b.copy_(a)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("D2D copy of %.2f GB: %.3f ms -> %.2f TB/s (read + write)" % (n / 1e9, ms, 2 * n / ms / 1e9))
