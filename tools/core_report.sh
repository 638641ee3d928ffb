#!/bin/bash
# post-mortem of a GPU core dump (no GPU access): faulting wave, its pc and the instructions around it
core=$(ls -t gpucore.* 2>/dev/null | head -1)
[ -z "$core" ] && { echo "no gpucore"; exit 0; }
/opt/rocm/bin/rocgdb -batch -ex "info threads" -ex "bt 3" -ex "x/24i \$pc-64" -ex "info registers" python3 "$core" 2>&1 | grep -v "^\[New" | head -${1:-300}
