"""What hipMalloc of a large block costs a fresh process (the CLI's device workspace), whether it matters that the process
before it just gave the same memory back, and whether pieces are cheaper than one block.
Run on the GPU box: python3 tools/alloc_cost.py"""
import subprocess, sys, time
CHILD = r'''
import ctypes, sys, time
hip = ctypes.CDLL("libamdhip64.so")
gb, pieces = float(sys.argv[1]), int(sys.argv[2])
t0 = time.perf_counter()
hip.hipInit(0)
free_b, tot_b = ctypes.c_size_t(), ctypes.c_size_t()
hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(tot_b))
t1 = time.perf_counter()
ps = []
n = int(gb * 1e9 / pieces)
for k in range(pieces):
    p = ctypes.c_void_p()
    rc = hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(n))
    ps.append(p)
hip.hipDeviceSynchronize()
t2 = time.perf_counter()
for p in ps:
    hip.hipMemset(p, 0, ctypes.c_size_t(n))
hip.hipDeviceSynchronize()
t3 = time.perf_counter()
for p in ps:
    hip.hipFree(p)
t4 = time.perf_counter()
print(f"rc {rc} init {1e3*(t1-t0):.0f} ms  hipMalloc {1e3*(t2-t1):.0f} ms  memset {1e3*(t3-t2):.0f} ms  hipFree {1e3*(t4-t3):.0f} ms  free before {free_b.value/1e9:.0f} GB")
'''
plan = [(100, 1, 0), (100, 1, 0), (100, 1, 10), (100, 100, 0), (100, 100, 0), (50, 1, 0), (50, 1, 0), (200, 1, 0)]
for gb, pieces, pause in plan:
    time.sleep(pause)
    t = time.perf_counter()
    r = subprocess.run([sys.executable, "-c", CHILD, str(gb), str(pieces)], capture_output=True, text=True)
    print(f"{gb:5.0f} GB in {pieces:3d} piece(s) after a pause of {pause:2d} s: {r.stdout.strip()} | process wall {time.perf_counter() - t:.2f} s",
          r.stderr.strip()[-200:], flush=True)
