#!/bin/bash
# register / LDS / spill numbers of the kernels in an object or library: tools/kernel_regs.sh <file> [name filter]
set -e
F=$(readlink -f "$1"); PAT=${2:-.}
D=$(mktemp -d); cp "$F" $D/in.bin; cd $D
/opt/rocm/lib/llvm/bin/llvm-objdump --offloading in.bin > /dev/null
for co in $(ls | grep gfx950); do
  /opt/rocm/lib/llvm/bin/llvm-readelf --notes $co | python3 -c "
import sys, re
txt = sys.stdin.read()
for blk in re.split(r'\n  - \.agpr_count:', txt)[1:]:
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    if not re.search(r'$PAT', name): continue
    g = lambda k: re.search(r'\.' + k + r':\s+(\d+)', blk).group(1)
    print(f\"{name[:90]:90s} agpr {blk.split()[0]:>3s} vgpr {g('vgpr_count'):>3s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} lds {g('group_segment_fixed_size'):>6s} scratch {g('private_segment_fixed_size')}\")
"
done
rm -rf $D
