import torch, time
d=torch.device('cuda:0')
x=torch.empty(1<<30, dtype=torch.float32, device=d)  # 4 GB
y=torch.empty(1<<30, dtype=torch.float32, device=d)
def t(fn,n=5):
    fn(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n
ms=t(lambda: x.zero_()); print("write-only  %.2f TB/s"%(4.295/ms))
ms=t(lambda: y.copy_(x)); print("copy r+w    %.2f TB/s (sum)"%(2*4.295/ms))
ms=t(lambda: x.sum()); print("read-only   %.2f TB/s"%(4.295/ms))
