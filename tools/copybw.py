import torch, time
a = torch.empty(8 << 30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
for _ in range(2): b.copy_(a)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(5): b.copy_(a)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
print("copy 8GiB: %.2f ms -> %.2f TB/s (read+write)" % (dt*1e3, 2*a.numel()/dt/1e12))
