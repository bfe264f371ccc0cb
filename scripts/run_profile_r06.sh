#!/bin/bash
# launches the round-6 profile from a COMMITTED tree and stamps which one (profiles/traffic.json: `_measured_at.head`)
cd "$(dirname "$0")/.."
if [ -n "$(git status --porcelain -- re2nn-seq_amd bench.py scripts include)" ]; then echo "uncommitted changes: commit first"; exit 1; fi
python re2nn-seq_amd/csrc/build.py > /dev/null 2>&1 && python re2nn-seq_amd/csrc/build.py --probes > /dev/null 2>&1 || { echo "build failed"; exit 1; }
mkdir -p gpurun_out; git rev-parse --short=12 HEAD > gpurun_out/prof_r06_head.txt
/usr/local/graft/bin/gpurun --timeout 3300 -- 'bash scripts/gpu_profile_r06.sh' && python scripts/summarize_profile.py r06
