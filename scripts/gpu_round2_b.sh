#!/bin/bash
# GPU visit B of round 2: the round-2 tests, the whole GPU suite, ASan run, DETEST table, bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r2b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round2.py -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest_round2.log 2>&1; echo "round2 tests exit $?" | tee -a $O/pytest_round2.log
tail -40 $O/pytest_round2.log
bash scripts/gpu_asan.sh 2>&1 | tail -12
timeout 900 python scripts/detest_run.py > $O/detest_table.txt 2>&1; echo "detest exit $?"; grep "Total\|=====" $O/detest_table.txt
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --deselect tests/test_gpu_round2.py > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_gpu.log
tail -30 $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
