#!/bin/bash
# Registers, spills and LDS of the device kernels in an object file made by hipcc: tools/kres.sh <file.o> [name filter]
# (unbundles the gfx950 code object from .hip_fatbin and reads its metadata notes)
set -e
T=$(mktemp -d)
L=/opt/rocm/lib/llvm/bin
$L/llvm-objcopy --dump-section .hip_fatbin=$T/fat.bin "$1"
$L/clang-offload-bundler --type=o --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/fat.bin --output=$T/k.co
$L/llvm-readelf --notes $T/k.co | python3 -c "
import re, sys
txt = sys.stdin.read()
for blk in txt.split('- .agpr_count')[1:]:
    g = lambda k: (re.search(r'\.' + k + r':\s*(\S+)', blk) or [None, '?'])[1]
    name = g('name')
    import subprocess
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0].replace('void ', '')
    if len(sys.argv) > 1 and sys.argv[1] not in dem: continue
    print(f\"{dem:60s} vgpr {g('vgpr_count'):>4s} agpr {blk.split()[0]:>3s} sgpr {g('sgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size'):>6s}\")
" "${2:-}"
rm -rf $T
