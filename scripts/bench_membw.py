import torch, time
dev = torch.device('cuda:0')
for mb in (3.7, 14.7, 58.7, 235, 940):
    n = int(mb * 1e6 / 4)
    x = torch.empty(n, device=dev); y = torch.empty(n, device=dev)
    for name, fn in (('fill', lambda: x.fill_(1.0)), ('copy', lambda: y.copy_(x)), ('axpy', lambda: y.add_(x))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        traffic = mb * (1 if name == 'fill' else 2 if name == 'copy' else 3)
        print(f'{mb:7.1f} MB {name}: {us:8.1f} us  -> {traffic / us * 1e-3 * 1e3:.0f} GB/s', flush=True)
