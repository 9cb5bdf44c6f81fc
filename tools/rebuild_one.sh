#!/bin/bash
# Development aid: recompile only the given csrc units (e.g. conv.hip conv_w2d_1.hip) into build/ and build_dev/ and relink both
# libraries, then mark every object current (build.py rebuilds everything when any header is newer than an object).
# Only valid when the edited header is included by nothing but the units named.
set -e
cd "$(dirname "$0")/../aicovergen_amd"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-constant-logical-operand"
pids=()
for u in "$@"; do
  /opt/rocm/bin/hipcc $FLAGS -c csrc/$u -o build/$u.o & pids+=($!)
  /opt/rocm/bin/hipcc $FLAGS -DAICG_DEV_SWITCHES -DAICG_CONV_ABLATION -c csrc/$u -o build_dev/$u.o & pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
(cd build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libaicg_hip.so.tmp *.hip.o && mv libaicg_hip.so.tmp ../libaicg_hip.so) &
(cd build_dev && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libaicg_hip_dev.so.tmp *.hip.o && mv libaicg_hip_dev.so.tmp ../libaicg_hip_dev.so) &
wait
touch build/*.o build_dev/*.o libaicg_hip.so libaicg_hip_dev.so
ls -la libaicg_hip.so libaicg_hip_dev.so
