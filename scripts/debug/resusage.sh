#!/bin/bash
# registers / spills / scratch of every kernel of one translation unit (no link, ~seconds):  scripts/debug/resusage.sh file.hip [flags]
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$f" -o /dev/null --cuda-device-only -Rpass-analysis=kernel-resource-usage "$@" 2>&1 |
  awk '/Function Name:/ {name=$0; sub(/.*Function Name: /,"",name); sub(/ \[-Rpass.*/,"",name)}
       / VGPRs:/ {v=$(NF-1)} / SGPRs:/ {sg=$(NF-1)} /ScratchSize/ {sc=$(NF-1)} /SGPRs Spill/ {ss=$(NF-1)}
       /VGPRs Spill/ {vs=$(NF-1); printf "%-5s %-5s vspill %-4s sspill %-4s scratch %-5s %s\n", v, sg, vs, ss, sc, name}'
