#!/bin/bash
for t in 128 256 512 1024; do echo "target $t"; DMEL_FBG_TARGET=$t python tools/time_backward_extras.py 2>&1 | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print({k:v for k,v in d.items() if 'xgrad' not in k})"; done
