#!/bin/bash
# developer tool: registers / scratch of the built kernels (reads the code objects bundled in csrc/libmpc_hip.so); optional grep pattern on the mangled name
cd "$(dirname "$0")/../../mpc_local_planner_amd/csrc" || exit 1
T=$(mktemp -d); cp libmpc_hip.so $T/l.so; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading l.so > /dev/null 2>&1
for f in l.so.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed_size|\.vgpr_spill_count" | paste - - - - - ; done \
  | sed -E 's/ +/ /g; s/_ZN3mpc19mpc_ipm_wave_kernel(I[a-zA-Z0-9]*EE)Ev[A-Za-z0-9_]*/\1/; s/\.private_segment_fixed_size/scratch/; s/\.(agpr|vgpr)_count/\1/g; s/\.vgpr_spill_count/spill/' | grep "${1:-.}"
rm -rf $T
