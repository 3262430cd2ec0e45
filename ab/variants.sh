#!/bin/bash
# same-box comparison of launch variants of the ordered kernel chosen by RT_FAST_VARIANT: ABV="0 1 2", ABSPP, ABARGS, ABREPS
set -e
for rep in $(seq 1 ${ABREPS:-2}); do
  for v in ${ABV:-0 1}; do
    RT_FAST_VARIANT=$v timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --spp ${ABSPP:-8} --cpu-seconds 0 --no-extra $ABARGS > gpurun_out/var_$v$rep.json 2> gpurun_out/var_$v$rep.err
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/var_$v$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('variant $v rep $rep', d['value'], round(r['rays_per_s_in_kernel']/1e6,1), r['nodes_per_ray'], r['tris_per_ray'], d['config']['rays_per_step'])"
  done
done
