#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r2d; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_round2.py -m gpu -q --maxfail=30 -p no:cacheprovider > $O/pytest_round2.log 2>&1; echo "round2 tests exit $?" | tee -a $O/pytest_round2.log
tail -8 $O/pytest_round2.log
timeout 2400 python -m pytest tests -m gpu -q --maxfail=40 -p no:cacheprovider --deselect tests/test_gpu_round2.py > $O/pytest_gpu.log 2>&1; echo "pytest exit $?" | tee -a $O/pytest_gpu.log
tail -12 $O/pytest_gpu.log
for i in 1 2; do timeout 300 python bench.py --config 5 --steps 20 --warmup 2 2>> $O/bench.err | cut -c1-1200; done
timeout 300 python bench.py --no-cpu-baseline --steps 10 2>> $O/bench.err | cut -c1-300
