#!/bin/bash
# Host code of librt355 and the whole CPU oracle under AddressSanitizer + UBSan (CPU only:
# the device code is compiled as usual and never runs here).
# usage: tests/sanitize/run.sh [thread]     (thread: ThreadSanitizer instead)
set -e
MODE=${1:-address}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=${TMPDIR:-/tmp}/rt355_sanitize
mkdir -p "$OUT"
if [ "$MODE" = thread ]; then
  SAN="-fsanitize=thread -fno-omit-frame-pointer"
  HSAN="-Xarch_host -fsanitize=thread -Xarch_host -fno-omit-frame-pointer"
else
  SAN="-fsanitize=address,undefined -fno-omit-frame-pointer"
  HSAN="-Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer"
fi
for f in orc_api orc_bvh orc_render orc_shapes; do
  [ -f "$ROOT/oracle/$f.c" ] && gcc -std=c11 -O1 -g $SAN -ffp-contract=off -fPIC -pthread -I "$ROOT/include" -I "$ROOT/oracle" -c "$ROOT/oracle/$f.c" -o "$OUT/$f.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Wno-unused-function \
  $HSAN \
  -I "$ROOT/include" -I "$ROOT/raytracer_amd/csrc" -x hip \
  "$ROOT/raytracer_amd/csrc/rt_kernels.hip" "$ROOT/raytracer_amd/csrc/rt_build_gpu.hip" \
  "$ROOT/raytracer_amd/csrc/rt_api.cpp" "$ROOT/raytracer_amd/csrc/rt_bvh.cpp" "$ROOT/tests/sanitize/host_check.cpp" \
  -x none "$OUT"/orc_*.o -lm -o "$OUT/host_check"
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1 RT355_BUILD_THREADS=4 "$OUT/host_check"
