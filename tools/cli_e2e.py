"""End-to-end `povu decompose` (parse + upload + decompose + write) on BASELINE config 2 as GFA text."""
import os, subprocess, sys, time, hashlib, json
sys.path.insert(0, '.')
from povu_amd import workloads as W
k = int(sys.argv[1]) if len(sys.argv) > 1 else 333333
threads = sys.argv[2] if len(sys.argv) > 2 else "8"
os.makedirs('/tmp/e2e/out', exist_ok=True)
gfa = '/tmp/e2e/chain.gfa'
if not os.path.exists(gfa):
    t = time.time(); open(gfa, 'w').write(W.chain_of_bubbles(k).to_gfa()); print('wrote gfa', os.path.getsize(gfa), 'bytes in', round(time.time() - t, 1), 's', flush=True)
env = dict(os.environ, POVU_STAGE_COST_TRACE='1')
for rep in range(2):
    t = time.time()
    r = subprocess.run(['povu_amd/bin/povu', '-t', threads, 'decompose', '-i', gfa, '-o', '/tmp/e2e/out'], capture_output=True, text=True, env=env)
    dt = time.time() - t
    print('run', rep, 'wall', round(dt, 3), 's rc', r.returncode, flush=True)
lines = [l for l in r.stderr.splitlines() if 'contract=host' in l or 'contract=hip:total' in l]
print('\n'.join(lines))
print('md5', hashlib.md5(open('/tmp/e2e/out/1.pvst', 'rb').read()).hexdigest())
