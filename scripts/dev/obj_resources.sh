#!/bin/bash
# developer tool: registers / scratch / spills of the kernels of ONE object file (split build or a developer build): obj_resources.sh <file.o>
T=$(mktemp -d); cp "$1" $T/o.o; cd $T
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading o.o > /dev/null 2>&1
for f in o.o.*gfx950; do /opt/rocm/lib/llvm/bin/llvm-readelf --notes "$f" 2>/dev/null | grep -E "^ +\.name:|\.vgpr_count|\.agpr_count|private_segment_fixed_size|\.vgpr_spill_count|\.sgpr_spill_count|group_segment" | paste - - - - - - - ; done \
  | sed -E 's/ +/ /g; s/_ZN3mpc19mpc_ipm_wave_kernel(I[a-zA-Z0-9]*EE)Ev[A-Za-z0-9_]*/\1/; s/\.private_segment_fixed_size/scratch/; s/\.(agpr|vgpr)_count/\1/g; s/\.vgpr_spill_count/vspill/; s/\.sgpr_spill_count/sspill/; s/\.group_segment_fixed_size/lds/'
rm -rf $T
