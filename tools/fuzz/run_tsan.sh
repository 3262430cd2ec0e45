#!/bin/bash
# Builds the library's host code with ThreadSanitizer (host side only: -Xarch_host)
# together with tools/fuzz/host_threads.cpp and runs it on the CPU.  No GPU needed.
#   bash tools/fuzz/run_tsan.sh [triangles]
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=${FUZZ_OUT:-/tmp/rt_host_fuzz}
mkdir -p "$out"
src="$root/raytracer_amd/csrc"
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
      -fno-slp-vectorize -pthread -Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer -I"$root/include" -I"$src" \
      "$src"/bvh_build.cpp "$src"/bvh_io.cpp "$src"/flatten.cpp "$src"/scene_host.cpp "$src"/obj_loader.cpp \
      "$src"/png_decode.cpp "$src"/backend.cpp "$src"/capi.cpp "$src"/rt_kernels.hip "$root/tools/fuzz/host_threads.cpp" \
      -o "$out/host_threads" 2> "$out/build_tsan.log" || { tail -30 "$out/build_tsan.log"; exit 1; }
export TSAN_OPTIONS=halt_on_error=1:report_signal_unsafe=0
"$out/host_threads" "${1:-150000}"
