import torch
a=torch.load('/tmp/s128/new.pt'); b=torch.load('/tmp/s128/old.pt')
for k in a:
    d=(a[k]-b[k]).abs()
    print(k, "max abs diff", d.max().item(), "rel l2", (d.norm()/b[k].norm()).item(), "n diff", int((d>0).sum()), "of", d.numel())
