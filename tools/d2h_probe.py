"""How does a large device-to-host copy into page-locked memory run on this box: copy engine (SDMA) or a blit kernel on
the CUs?  Times a 300 MB copy alone and beside a bandwidth-bound kernel."""
import os, sys, time, torch
n = 300 << 20
d = torch.empty(n, dtype=torch.uint8, device="cuda").fill_(3)
h = torch.empty(n, dtype=torch.uint8, pin_memory=True)
x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def copy_ms(k=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.cuda.stream(s1):
        for _ in range(k): h.copy_(d, non_blocking=True)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
def kern_ms(k=20):
    torch.cuda.synchronize(); t = time.perf_counter()
    with torch.cuda.stream(s2):
        for _ in range(k): x.add_(1)
    torch.cuda.synchronize(); return (time.perf_counter() - t) / k * 1e3
copy_ms(2); kern_ms(2)
c, k = copy_ms(), kern_ms()
torch.cuda.synchronize(); t = time.perf_counter()
with torch.cuda.stream(s1):
    for _ in range(5): h.copy_(d, non_blocking=True)
with torch.cuda.stream(s2):
    for _ in range(60): x.add_(1)
torch.cuda.synchronize(); both = (time.perf_counter() - t) * 1e3
print(f"env {os.environ.get('PROBE_TAG','default')}: copy {c:.2f} ms ({n/c/1e6:.1f} GB/s), kernel {k:.3f} ms, 5 copies + 60 kernels together {both:.2f} ms (serial would be {5*c+60*k:.2f})")
