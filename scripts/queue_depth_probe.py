"""How many launches can the host queue behind a busy GPU before it blocks?  A ~300 ms kernel, then tiny launches, timing each enqueue."""
import time, torch
dev = "cuda"
a = torch.randn(16384, 16384, device=dev)
torch.matmul(a, a); torch.cuda.synchronize()
x = torch.zeros(1024, device=dev)
for trial in range(2):
    t0 = time.perf_counter()
    for _ in range(6): y = torch.matmul(a, a)           # ~ several hundred ms of GPU work
    t_big = time.perf_counter() - t0
    stamps = []
    for i in range(3000):
        t1 = time.perf_counter(); x.add_(1.0); stamps.append(time.perf_counter() - t1)
    torch.cuda.synchronize()
    slow = [i for i, s in enumerate(stamps) if s > 2e-3]
    print(f"trial {trial}: big enqueue {t_big*1e3:.1f} ms; first slow tiny launch at index {slow[0] if slow else None} "
          f"(took {stamps[slow[0]]*1e3 if slow else 0:.1f} ms); median enqueue {sorted(stamps)[len(stamps)//2]*1e6:.1f} us", flush=True)
