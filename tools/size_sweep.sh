#!/bin/bash
# tools/size_sweep.sh -- bench.py over triangle counts: ns per BVH step (pair or leaf record) as the tree outgrows L2 / Infinity Cache
set -u
: > gpurun_out/size_sweep.log
for t in 20000 200000 1000000 4000000; do
  timeout -k 10 300 python -u bench.py --no-cpu-baseline --steps 2 --warmup 1 --tris $t 2>&1 | grep '^{' >> gpurun_out/size_sweep.log || exit 1
done
python3 - <<'PY'
import json
for l in open('gpurun_out/size_sweep.log'):
    j=json.loads(l); r=j['roofline']; k=j['kernel_ms']
    es=j['extend_walkers']*(r['node_pairs_per_walked_ray']+r['leaf_prims_per_walked_ray'])
    ss=j['shadow_walkers']*(r['shadow_node_pairs_per_walked_ray']+r['shadow_leaf_prims_per_walked_ray'])
    print('tris %8d Mrays/s %.0f ms/step %.1f | ext %.1f ms, %.0fM steps, %.2f ps/step | conn %.1f ms, %.0fM steps, %.2f ps/step | shade %.1f | steps/walker %.0f / %.0f' % (
        j['config']['triangles'], j['value'], j['ms_per_step'], k['extend'], es/1e6, k['extend']*1e9/es, k['connect'], ss/1e6, k['connect']*1e9/ss, k['shade'],
        r['node_pairs_per_walked_ray']+r['leaf_prims_per_walked_ray'], r['shadow_node_pairs_per_walked_ray']+r['shadow_leaf_prims_per_walked_ray']))
PY
