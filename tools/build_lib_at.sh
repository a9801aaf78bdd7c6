#!/bin/bash
# tools/build_lib_at.sh <git revision> [name]: builds libdle_mi355x.so from the csrc/ + include/ of another revision into
# tools/kbench/bin/libdle_<name>.so (git-ignored, travels with gpurun).  Run a workload against it with
#   DLE_LIB_PATH=tools/kbench/bin/libdle_<name>.so python bench.py ...
# in the SAME gpurun call as the current library: boxes of the pool differ by +-4 %, two calls do not compare.
set -e
REV=$1; NAME=${2:-prev}
R=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/dle_at_XXXX)
git -C "$R" archive "$REV" deeplearningexamples_amd/csrc include | tar -x -C "$W"
mkdir -p "$W/obj" "$R/tools/kbench/bin"
cd "$W/deeplearningexamples_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-result -DNDEBUG -c $f -o "$W/obj/${f%.hip}.o" 2>/dev/null &
  while [ $(jobs -r | wc -l) -ge 8 ]; do sleep 0.5; done
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$W"/obj/*.o -o "$R/tools/kbench/bin/libdle_$NAME.so"
rm -rf "$W"
ls -la "$R/tools/kbench/bin/libdle_$NAME.so"
