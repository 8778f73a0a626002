import torch,time
for mb in (6.5,13,26,52,104,416):
    n=int(mb*1e6/4)
    x=torch.randn(n,device='cuda'); y=torch.empty_like(x)
    for _ in range(5): y.copy_(x)
    torch.cuda.synchronize()
    ev=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(20)]
    for a,b in ev:
        a.record(); y.copy_(x); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in ev)[len(ev)//2]
    print(f"copy {mb} MB: {t*1e3:.1f} us  -> {2*mb/1e3/(t/1e3)/1e3:.2f} TB/s (read+write)")
    z=torch.empty_like(x)
    for a,b in ev:
        a.record(); torch.add(x,y,out=z); b.record()
    torch.cuda.synchronize()
    t=sorted(a.elapsed_time(b) for a,b in ev)[len(ev)//2]
    print(f"add  {mb} MB: {t*1e3:.1f} us  -> {3*mb/1e3/(t/1e3)/1e3:.2f} TB/s")
