#!/bin/bash
# Build libcaptioner_hip.so of an earlier revision into embodied_captioning_amd/lib/libcaptioner_old.so (git-ignored; it travels to
# the GPU box with the snapshot) for same-process A/B runs: tools/ab_gemm_libs.py.
#   bash tools/build_old_lib.sh <git revision>
set -e
REV=${1:?usage: build_old_lib.sh <git revision>}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d /tmp/oldlib.XXXXXX)
git -C "$ROOT" archive "$REV" embodied_captioning_amd include | tar -x -C "$TMP"
(cd "$TMP" && python3 -m embodied_captioning_amd.build --force > /dev/null)
cp "$TMP/embodied_captioning_amd/lib/libcaptioner_hip.so" "$ROOT/embodied_captioning_amd/lib/libcaptioner_old.so"
rm -rf "$TMP"
echo "built $REV -> embodied_captioning_amd/lib/libcaptioner_old.so"
