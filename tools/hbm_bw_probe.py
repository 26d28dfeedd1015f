import torch, time
x = torch.empty(1<<30, dtype=torch.float32, device='cuda')  # 4 GB
y = torch.empty_like(x)
x.normal_()
def t(fn, n=10):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n
dt = t(lambda: y.copy_(x)); print("copy 4GB r+w: %.2f TB/s" % (8*(1<<30)/dt/1e12))
dt = t(lambda: x.sum()); print("sum 4GB read: %.2f TB/s" % (4*(1<<30)/dt/1e12))
dt = t(lambda: y.fill_(1.0)); print("fill 4GB write: %.2f TB/s" % (4*(1<<30)/dt/1e12))
dt = t(lambda: torch.add(x, 1.0, out=y)); print("add 4GB r+w: %.2f TB/s" % (8*(1<<30)/dt/1e12))
