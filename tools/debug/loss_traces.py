"""Debug aid: loss trajectories of bench.py runs (fresh child processes): two plain runs and one
with a forced one-rank RCCL process group.  usage: python tools/debug/loss_traces.py [steps] [batch]"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
steps = sys.argv[1] if len(sys.argv) > 1 else '12'
batch = sys.argv[2] if len(sys.argv) > 2 else '2'
base = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'NESIE_FORCE_PG')}
base['NESIE_DETERMINISTIC'] = os.environ.get('NESIE_DETERMINISTIC', '1')
for name, extra, more in (('plain-a', {}, []), ('plain-b', {}, []), ('plain-eager', {}, ['--graph', '0']),
                          ('rccl', dict(NESIE_FORCE_PG='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29541'), []),
                          ('rccl-eager', dict(NESIE_FORCE_PG='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29542'), ['--graph', '0'])):
    out = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', steps, '--warmup', '0',
                          '--batch', batch, '--cpu-baseline', '0', '--parity-gate', '0', '--loss-trace', '1'] + more,
                         env=dict(base, **extra), capture_output=True, text=True, timeout=900)
    lines = [l for l in out.stdout.splitlines() if l.startswith('{"metric"')]
    if out.returncode or not lines:
        print(name, 'FAILED', out.stderr[-1500:])
        continue
    r = json.loads(lines[0])
    print(name, r['config']['collective_backend'], r['config']['collectives_per_step'],
          ' '.join(f'{v:.4f}' for v in r['loss_trace']), flush=True)
