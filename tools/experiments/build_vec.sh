#!/bin/bash
# builds the two vector-unit input-gradient experiments next to their sources (the .so files travel to the GPU box)
cd "$(dirname "$0")"
CS=../../neurips21-self-supervised-bug-detection-and-repair_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$CS -shared bl_routed_dgrad_vec.hip $CS/bl_core.hip -o libvec1.so "$@" && \
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared bl_routed_dgrad_vec2.hip -o libvec2.so "$@"
