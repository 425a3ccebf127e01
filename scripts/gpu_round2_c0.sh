#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r2c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round2.py -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest_round2.log 2>&1; echo "round2 tests exit $?" | tee -a $O/pytest_round2.log
tail -15 $O/pytest_round2.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "linear or fixed or whole" > $O/pytest_linear.log 2>&1; echo "linear-family tests exit $?"; tail -4 $O/pytest_linear.log
bash scripts/gpu_round2_c.sh 2>&1 | tail -150
