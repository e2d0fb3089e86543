import time, torch
torch.cuda.init(); torch.zeros(1, device='cuda')
for mb in (64, 600):
    n = mb * 1024 * 1024 // 4
    t0 = time.time(); h = torch.empty(n, dtype=torch.float32, pin_memory=True); t1 = time.time()
    h.fill_(1.0); t2 = time.time()
    d = torch.empty(n, dtype=torch.float32, device='cuda'); torch.cuda.synchronize(); t3 = time.time()
    d.copy_(h, non_blocking=True); torch.cuda.synchronize(); t4 = time.time()
    d.copy_(h, non_blocking=True); torch.cuda.synchronize(); t5 = time.time()
    p = torch.empty(n, dtype=torch.float32); p.fill_(1.0); t6 = time.time()
    d.copy_(p); torch.cuda.synchronize(); t7 = time.time()
    h2 = torch.empty(n, dtype=torch.float32, pin_memory=True); t8 = time.time()
    del h2; t9 = time.time()
    h3 = torch.empty(n, dtype=torch.float32, pin_memory=True); t10 = time.time()
    print("%d MB: pin alloc %.1f ms, first touch %.1f ms, H2D pinned %.1f / %.1f ms, pageable H2D %.1f ms, 2nd pin alloc %.1f ms, free %.1f, 3rd (cached?) %.1f ms"
          % (mb, (t1-t0)*1e3, (t2-t1)*1e3, (t4-t3)*1e3, (t5-t4)*1e3, (t7-t6)*1e3, (t8-t7)*1e3, (t9-t8)*1e3, (t10-t9)*1e3), flush=True)
