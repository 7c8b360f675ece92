#!/bin/bash
# Do the millisecond steps of bench.py coincide with power / thermal throttling?  The GPU's violation accumulators
# (amd-smi metric --throttle) and clocks / power before and after every bench run.   GPU box only.
cd "$(dirname "$0")/.."
snap() { /opt/rocm/bin/amd-smi metric --throttle --power --clock --json 2>/dev/null | python3 -c "
import json,sys
try:
    d=json.load(sys.stdin)
except Exception as e:
    print('amd-smi gave no JSON:', e); sys.exit(0)
g=d[0] if isinstance(d,list) else d
if 'gpu_data' in g: g=g['gpu_data'][0]
def flat(o,p=''):
    if isinstance(o,dict):
        for k,v in o.items(): yield from flat(v,p+k+'.')
    elif isinstance(o,list):
        for i,v in enumerate(o[:2]): yield from flat(v,p+str(i)+'.')
    else: yield p[:-1],o
keep=[(k,v) for k,v in flat(g) if any(s in k for s in ('accum','violation','socket_power','gfx_0.clk','throttle'))]
print(' '.join('%s=%s'%(k.split('.')[-2]+'.'+k.split('.')[-1] if k.count('.') else k,v) for k,v in keep)[:1500])
"; }
snap
for i in 1 2 3 4 5; do
  timeout -k 10 120 python3 bench.py --no-cpu-baseline --no-extras --grid 0 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('run', d['ms_per_step'], {k:v['ms_min_median_max'][1:] for k,v in d['stages'].items()})"
  snap
done
