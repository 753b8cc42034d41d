"""write / read / copy bandwidth of HBM vs Infinity-Cache-resident buffers (torch elementwise kernels):
python tools/micro/mem_bw.py"""
import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (64, 160, 1024, 4096):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    w = t(lambda: a.fill_(1.0)); r = t(lambda: a.sum()); c = t(lambda: b.copy_(a))
    print(f"{mb:5d} MB: fill {mb/1024/w/1e3*1e3/1e3:.2f} TB/s  sum(read) {mb/1024/r/1e3*1e3/1e3:.2f} TB/s  copy {2*mb/1024/c/1e3*1e3/1e3:.2f} TB/s (read+write)", flush=True)
