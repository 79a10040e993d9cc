#!/bin/bash
# round 4: the GPU suite on the 64-byte light-vertex records + the new stream schedule, then the bench line (no CPU legs)
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06j_pytest.log 2>&1
tail -3 gpurun_out/r06j_pytest.log
timeout 900 python bench.py --cpu-baseline none --steps 20 --warmup 5 > gpurun_out/r06j_bench.log 2> gpurun_out/r06j_bench.err
tail -3 gpurun_out/r06j_bench.err
python3 - <<'PY'
import json
for l in open('gpurun_out/r06j_bench.log'):
    if l.startswith('{'):
        d=json.loads(l)
        print({k:d[k] for k in ('metric','value','ms_per_step','steps','warmup')}, d.get('host_cross_check',{}).get('value'))
        r=d['roofline']; print({k:r[k] for k in list(r)[:14]})
        print('iteration traffic GB', r.get('iteration_traffic',0)/1e9, 'over design', r.get('iteration_traffic_over_design'))
        for k,v in r['per_kernel'].items():
            print(' ',k, v.get('ms'), v.get('ms_serialised_under_profiler'), 'traffic GB', round(v.get('traffic',0)/1e9,2), 'valu', (v.get('valu') or {}).get('frac'), (v.get('valu') or {}).get('lane_util'))
        for c in d.get('configs',[]): print(' ',c.get('name'), c.get('value'), c.get('ms_per_step'), c.get('error'))
PY
