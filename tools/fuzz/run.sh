#!/bin/bash
# Builds the library's host code with AddressSanitizer and UndefinedBehaviorSanitizer
# (host side only: -Xarch_host; the device code is compiled as usual and never runs
# here) together with tools/fuzz/host_fuzz.cpp and runs the mutation fuzzer on the
# CPU.  No GPU needed.
#   bash tools/fuzz/run.sh [iterations] [seed]
set -e
root=$(cd "$(dirname "$0")/../.." && pwd)
out=${FUZZ_OUT:-/tmp/rt_host_fuzz}
mkdir -p "$out/scene"
cp "$root/tests/golden/cornell-box/cornell-box.obj" "$root/tests/golden/cornell-box/cornell-box.mtl" "$out/scene/"
python3 - "$out/scene/fuzz_seed.png" <<'EOF'
import struct, sys, zlib
def chunk(t, d): return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
rows = b"".join(b"\x00" + bytes([10 * y, 20, 30, 255, 200, 10 * y, 5, 128]) for y in range(2))
png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(rows)) + chunk(b"IEND", b"")
open(sys.argv[1], "wb").write(png)
EOF
src="$root/raytracer_amd/csrc"
san="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer"
hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
      -fno-slp-vectorize -pthread $san -I"$root/include" -I"$src" \
      "$src"/bvh_build.cpp "$src"/bvh_io.cpp "$src"/flatten.cpp "$src"/scene_host.cpp "$src"/obj_loader.cpp \
      "$src"/png_decode.cpp "$src"/backend.cpp "$src"/capi.cpp "$src"/rt_kernels.hip "$root/tools/fuzz/host_fuzz.cpp" \
      -o "$out/host_fuzz" 2> "$out/build.log" || { tail -30 "$out/build.log"; exit 1; }
# the build environment preloads a library of its own in front of the sanitizer's runtime
export ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0:abort_on_error=0
export UBSAN_OPTIONS=print_stacktrace=1
"$out/host_fuzz" "$out/scene" "${1:-20000}" "${2:-1}"
