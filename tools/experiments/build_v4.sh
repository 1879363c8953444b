#!/bin/bash
# build the experimental kernel next to its source (the .so travels to the GPU box with the snapshot)
cd "$(dirname "$0")"
CS=../../neurips21-self-supervised-bug-detection-and-repair_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$CS -shared bl_gemm_x6v4.hip -o libx6v4.so "$@" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -I$CS -shared bl_gemm_x6v5.hip -o libx6v5.so "$@"
