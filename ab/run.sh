#!/bin/bash
# same-box A/B of library builds ab/lib_<name>.so: ABV="old new" (default), ABSPP, ABARGS, ABREPS
set -e
L=raytracer_amd/lib/libraytracer_amd.so
cp $L ab/lib_keep.so
trap 'cp ab/lib_keep.so '$L EXIT
for rep in $(seq 1 ${ABREPS:-2}); do
  for v in ${ABV:-old new}; do
    cp ab/lib_$v.so $L
    timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --spp ${ABSPP:-8} --cpu-seconds 0 --no-extra $ABARGS > gpurun_out/ab_$v$rep.json 2> gpurun_out/ab_$v$rep.err
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$v$rep.json').read().strip().splitlines()[-1]); r=d['roofline']; print('$v$rep', d['value'], round(r['rays_per_s_in_kernel']/1e6,1), r['nodes_per_ray'], r['tris_per_ray'], d['config']['rays_per_step'])"
  done
done
