#!/bin/bash
# round 5: the whole GPU suite + the replay of the soak's sensitive geometry
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05suite; rm -rf $O; mkdir -p $O
timeout 3000 python -m pytest tests -m gpu -q --maxfail=15 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -40 $O/pytest.txt | cut -c1-250
PYTHONPATH=. timeout 600 python tests/soak_decomp_shapes.py 1500 1401 > $O/soak_1401.txt 2>&1; tail -3 $O/soak_1401.txt | cut -c1-400
