import torch, time
dev='cuda'
for gb in (0.5, 1.0, 2.0):
    n=int(gb*2**30/2)
    x=torch.randn(n//8, 8, device=dev).to(torch.bfloat16).view(-1)
    y=torch.empty_like(x)
    for name,fn,bytes_ in (("copy", lambda: y.copy_(x), 2*x.numel()*2), ("read-only sum", lambda: x.view(torch.int16).sum(), x.numel()*2), ("fill", lambda: y.zero_(), x.numel()*2), ("add3", lambda: torch.add(x, y, out=y), 3*x.numel()*2)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms=e0.elapsed_time(e1)/10
        print(f"{gb} GB {name:14s} {ms*1e3:8.1f} us  {bytes_/ms/1e9:6.2f} TB/s")
