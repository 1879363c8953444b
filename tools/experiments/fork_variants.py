#!/usr/bin/env python
"""Variant sources for fork_probe.sh: csrc/bl_mp_layer.hip with the side-stream fork of the routed weight gradient moved behind the routed
input gradient's launch (forkafter_dgrad) or behind the segmented sums (forkafter_sums).  Block moves only; writes tools/experiments/build/."""
import os

R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import subprocess

# (the source as it was when the probe ran: the product has since taken the late fork for layers 256 wide and up, as a run-time choice)
src = subprocess.check_output(["git", "-C", R, "show", "57ba164:neurips21-self-supervised-bug-detection-and-repair_amd/csrc/bl_mp_layer.hip"], text=True)
B = os.path.join(R, "tools/experiments/build")
os.makedirs(B, exist_ok=True)
fork = """    if (two) {
      (void)hipEventRecord(ev->fork2, st);
      (void)hipStreamWaitEvent(side, ev->fork2, 0);
    }
"""
assert src.count(fork) == 1
i0 = src.index(fork)
i1 = src.index("    bl_rows_packed_t g;", i0)          # end of the weight-gradient block
wgrad_block = src[i0 + len(fork):i1]
i2 = src.index("  if (!fused_sums) {", i1)             # the segmented sums
close = src.rindex("  }\n", i1, i2)                    # end of `if (E > 0) {`
open(os.path.join(B, "bl_mp_layer_forkafter_dgrad.hip"), "w").write(src[:i0] + src[i1:close] + fork + wgrad_block + src[close:])
i3 = src.index("  if (two && join_side) {", i2)
a_setup = src[src.rindex("    bl_rows_packed_t a;", 0, i0):i0]
open(os.path.join(B, "bl_mp_layer_forkafter_sums.hip"), "w").write(
    src[:i0] + src[i1:i3] + "  if (E > 0) {\n" + a_setup + fork + wgrad_block + "  }\n" + src[i3:])
