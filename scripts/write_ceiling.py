"""Streaming-write ceiling of one MI355X for the maze3d frame batch: how long does it take just to WRITE 12.9 GB
(16 384 frames of 256 x 256 x 3 int32)? torch fill / copy kernels, HIP events."""
import json
import torch

n = 16384 * 256 * 256 * 3
x = torch.empty(n, dtype=torch.int32, device="cuda:0")
y = torch.empty(n // 2, dtype=torch.int32, device="cuda:0")


def timed(f, reps=6):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


ms_fill = timed(lambda: x.fill_(7))
ms_copy = timed(lambda: x[:n // 2].copy_(y))
print(json.dumps({"bytes": 4 * n, "fill_ms": ms_fill, "fill_TBs": 4 * n / ms_fill / 1e9,
                  "copy_half_ms": ms_copy, "copy_TBs_read_plus_write": 4 * n / ms_copy / 1e9}))
