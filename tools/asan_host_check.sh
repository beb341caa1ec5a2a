#!/bin/bash
# Host-side AddressSanitizer build + run of the C ABI's argument handling (GPU ASan / xnack+ are not available on the pool, and
# this needs no GPU).  Usage: bash tools/asan_host_check.sh [out.log]     (about two minutes: every .hip file is rebuilt)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/dev/stdout}
B=${TMPDIR:-/tmp}/cap_asan_build
rm -rf "$B"; mkdir -p "$B"            # never link objects of an earlier run
# the translation units of the product library, from the one list the build uses
SOURCES=$(cd "$ROOT" && python3 -c "from embodied_captioning_amd.build import SOURCES; print(' '.join(s[:-4] for s in SOURCES))")
CLANG=/opt/rocm/lib/llvm/bin/clang
FLAGS="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer"
pids=()
objs=()
for f in $SOURCES; do
  objs+=("$B/$f.o")
  hipcc $FLAGS -c "$ROOT/embodied_captioning_amd/csrc/$f.hip" -o "$B/$f.o" & pids+=($!)
  if [ ${#pids[@]} -ge 4 ]; then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -fsanitize=address -fno-gpu-sanitize "${objs[@]}" -o "$B/libcaptioner_hip_asan.so"
$CLANG -O1 -g -fsanitize=address -fno-omit-frame-pointer "$ROOT/tools/asan_host_check.c" -o "$B/asan_host_check" \
  -L"$B" -lcaptioner_hip_asan -Wl,-rpath,"$B" -L/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib
{
  echo "# $(date -u +%FT%TZ) host-side ASan run of the C ABI (tools/asan_host_check.sh), $(hipcc --version | grep -m1 'HIP version')"
  echo "# translation units: $SOURCES"
  ASAN_OPTIONS=detect_leaks=1:halt_on_error=1:protect_shadow_gap=0 "$B/asan_host_check" 2>&1
  echo "exit code: $?"
} > "$OUT"
