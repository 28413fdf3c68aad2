"""`povu decompose -s` end to end as a child process on a whole-genome-shaped GFA: python tools/cli_sub_e2e.py <segments>"""
import os, shutil, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from povu_amd import hip as H, workloads as W
n = float(sys.argv[1]) if len(sys.argv) > 1 else 2e7
g = W.hprc_whole_genome(n)
povu = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "povu_amd", "bin", "povu")
d = tempfile.mkdtemp(prefix="povu_sub_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    gfa = os.path.join(d, "g.gfa")
    H.write_gfa(g, gfa)
    env = dict(os.environ, POVU_STAGE_COST_TRACE="1")
    for flags in ([], ["-s"], ["-s"]):
        o = os.path.join(d, "out"); shutil.rmtree(o, ignore_errors=True); os.makedirs(o)
        t0 = time.perf_counter()
        r = subprocess.run([povu, "-t", "32", "decompose", "-i", gfa, "-o", o] + flags, capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        size = sum(os.path.getsize(os.path.join(o, f)) for f in os.listdir(o))
        parts = {l.split("stage=")[1].split()[0]: l.split("ms=")[1].split()[0] for l in r.stderr.splitlines() if "stage=" in l and "ms=" in l and "host" in l}
        print(f"segments {g.n_vtx}: povu decompose {' '.join(flags) or '(plain)'}: rc {r.returncode} {dt:.2f} s, {size >> 20} MiB of .pvst", parts, flush=True)
        if r.returncode:
            print(r.stderr[-2000:])
finally:
    shutil.rmtree(d, ignore_errors=True)
