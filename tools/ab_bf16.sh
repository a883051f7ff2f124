#!/bin/bash
# same-box A/B of library variants on the bf16 M-1M bench:  bash tools/ab_bf16.sh <variant.so> [<variant.so> ...]  ("main" = the in-tree library)
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = main ]; then unset MGN_LIB_PATH; else export MGN_LIB_PATH=$v; fi
  python bench.py --dtype bf16 --no-secondary --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('$v', 'step %.4f ms  edge %.4f ms  node-side %.4f ms  frac %.3f' % (d['ms_per_processor_step'], r['avg_launch_ms'], r['node_side']['avg_launch_ms'], r['frac']), d['latents_finite'])"
done; done
