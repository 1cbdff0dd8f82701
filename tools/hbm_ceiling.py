"""What a plain streaming kernel reaches on this box (reference point for the element-wise / 1x1 kernels' HBM fractions):
torch's copy_, add(out=) and a read-only sum over tensors far larger than the 256 MB of Infinity Cache."""
import torch
dev = torch.device("cuda")
n = 1 << 29   # 512 Mi elements
for dt in (torch.float16, torch.float32):
    x = torch.empty(n, dtype=dt, device=dev).normal_()
    y = torch.empty_like(x).normal_()
    z = torch.empty_like(x)
    nb = x.numel() * x.element_size()
    for name, fn, passes in (("copy (1 read + 1 write)", lambda: z.copy_(x), 2), ("add  (2 reads + 1 write)", lambda: torch.add(x, y, out=z), 3),
                             ("sum  (1 read)", lambda: x.sum(), 1)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{str(dt):14s} {name:26s} {nb / 2**20:6.0f} MiB per tensor: {ms * 1e3:8.1f} us  {passes * nb / ms / 1e9:7.2f} TB/s")
    del x, y, z
