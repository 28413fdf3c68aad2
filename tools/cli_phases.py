"""The CLI end to end on the headline workload, with the loader's phases (POVU_GFA_TIMING) and the CLI's own stage costs.
The parent first runs the workload in a context of its own and closes it, as bench.py does before its end_to_end leg (what a
fresh process pays for its device memory depends on what the device did just before: tools/alloc_cost.py).
Run on the GPU box: python3 tools/cli_phases.py [threads ...]   (POVU_CLI_ENV="A=1 B=2": extra environment of the child)"""
import os, subprocess, sys, time, shutil
sys.path.insert(0, '.')
from povu_amd import workloads as W, hip
scale = float(os.environ.get('POVU_CLI_SCALE', '1e8'))
g = W.hprc_whole_genome(scale)
gfa, out = '/dev/shm/povu_cli_phases.gfa', '/dev/shm/povu_cli_phases_out'
hip.write_gfa(g, gfa)
print('gfa bytes', os.path.getsize(gfa), flush=True)
d = hip.HipDecomposer(0); d.upload(g); f = d.decompose(); del f; d.close()
extra = dict(kv.split('=', 1) for kv in os.environ.get('POVU_CLI_ENV', '').split())
povu = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'povu_amd', 'bin', 'povu')
for th in [int(a) for a in sys.argv[1:]] or [32]:
    for rep in range(2):
        shutil.rmtree(out, ignore_errors=True); os.makedirs(out)
        t = time.perf_counter()
        r = subprocess.run([povu, '-t', str(th), 'decompose', '-i', gfa, '-o', out], capture_output=True, text=True,
                           env=dict(os.environ, POVU_GFA_TIMING='1', POVU_STAGE_COST_TRACE='1', **extra))
        dt = time.perf_counter() - t
        import re
        lines = [' '.join(l.split()) for l in r.stderr.splitlines() if l.startswith('gfa ')]
        lines += [f'{m.group(1)} {float(m.group(2)) / 1e6:.1f} ms' for m in re.finditer(r'contract=host:(\w+) .*?elapsed_ns=(\d+)', r.stderr)]
        print(f'threads {th} run {rep}: wall {dt:.3f} s rc {r.returncode} |', ' | '.join(lines), flush=True)
shutil.rmtree(out, ignore_errors=True); os.remove(gfa)
