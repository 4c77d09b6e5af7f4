"""What a write-dominant streaming kernel can reach on this GPU: fill (write only), copy (1 read + 1 write), sum (read only) of 1 GiB fp16 tensors
through PyTorch's own elementwise kernels (16-byte accesses), as a yardstick for the write-dominant kernels of the step (transposed-conv forward,
first-layer forward: both at ~2.3 TB/s of written bytes)."""
import torch

n = 1 << 29
a = torch.empty(n, dtype=torch.float16, device="cuda")
b = torch.empty(n, dtype=torch.float16, device="cuda")


def t(f, reps=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 2 / 1e9
print(f"fill  (write {gb:.2f} GB)          : {gb / t(lambda: a.fill_(1.0)) / 1e3:6.2f} TB/s written")
print(f"copy  (read + write {2 * gb:.2f} GB)   : {2 * gb / t(lambda: b.copy_(a)) / 1e3:6.2f} TB/s moved")
print(f"sum   (read {gb:.2f} GB)           : {gb / t(lambda: a.sum()) / 1e3:6.2f} TB/s read")
print(f"add   (2 reads + 1 write {3 * gb:.2f} GB): {3 * gb / t(lambda: torch.add(a, b, out=b)) / 1e3:6.2f} TB/s moved")
