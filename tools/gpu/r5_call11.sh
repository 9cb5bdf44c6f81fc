#!/bin/bash
# round 5, GPU call 11: conv_g1w as a persistent tile walk (d = 1) against the one-tile-per-workgroup form
cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c11; mkdir -p $O
timeout 600 python -m pytest tests/test_conv_g1w.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -2 $O/tests.log
timeout 600 python tools/kbench_g1w.py 3,6 7 1 > $O/kbench_g1w_pers.txt 2>&1; grep -v amdgpu.ids $O/kbench_g1w_pers.txt
