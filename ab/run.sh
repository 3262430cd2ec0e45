#!/bin/bash
# same-box A/B of two library builds: ab/lib_old.so and ab/lib_new.so
set -e
L=raytracer_amd/lib/libraytracer_amd.so
for rep in 1 2; do
  for v in old new; do
    cp ab/lib_$v.so $L
    timeout -k 10 200 python3 bench.py --steps 3 --warmup 1 --spp ${ABSPP:-8} --cpu-seconds 0 $ABARGS > gpurun_out/ab_$v$rep.json 2> gpurun_out/ab_$v$rep.err
    python3 -c "
import json,sys
d=json.loads(open('gpurun_out/ab_$v$rep.json').read().strip().splitlines()[-1]); print('$v$rep', d['value'], round(d['roofline']['rays_per_s_in_kernel']/1e6,1))"
  done
done
