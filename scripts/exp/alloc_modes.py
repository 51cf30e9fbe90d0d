#!/usr/bin/env python3
"""Is the speed of a random gather a property of the ALLOCATION?  A 19-GiB buffer is allocated, gathered from at random
(768-byte rows, torch.index_select), freed (empty_cache: a real hipFree) and allocated again, eight times; a second series
keeps a 10-GiB tensor alive beside it (the index base of the bench).  Prints GB/s per allocation."""
import json, time, torch
dev = torch.device("cuda", 0)
rows, cols = 25_000_000, 192
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.randint(0, rows, (4_000_000,), device=dev, generator=g)
def series(name, n, keep=None):
    out = []
    for i in range(n):
        x = torch.empty((rows, cols), dtype=torch.float32, device=dev)
        x[:1000].zero_()
        for _ in range(2):
            torch.index_select(x, 0, idx)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            y = torch.index_select(x, 0, idx)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        out.append({"ptr": hex(x.data_ptr()), "gather_GBps": round(idx.numel() * cols * 4 / ms / 1e6, 1)})
        del x, y
        torch.cuda.empty_cache()
    print(json.dumps({"series": name, "allocations": out}), flush=True)
series("alone", 8)
base = torch.empty((10_000_000, 256), dtype=torch.float32, device=dev)
series("beside_a_10GiB_tensor", 8)
small = [torch.empty((1 << 28,), dtype=torch.uint8, device=dev) for _ in range(7)]   # 7 x 256 MiB scattered allocations
series("after_scattered_small_allocations", 6)
