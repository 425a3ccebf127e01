#!/bin/bash
# GPU visit A of round 2: new tests first, then the whole GPU suite, bench (config 4 + others), 1-rank dist bench.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp
O=gpurun_out/r2a; rm -rf $O; mkdir -p $O
lscpu | grep -E "Model name|^CPU\(s\)|Socket|Thread" > $O/env.log; rocm-smi --showproductname 2>&1 | head -12 >> $O/env.log
timeout 900 python -m pytest tests/test_gpu_round2.py -m gpu -q -x --maxfail=30 -p no:cacheprovider > $O/pytest_round2.log 2>&1; echo "round2 tests exit $?" | tee -a $O/pytest_round2.log
tail -25 $O/pytest_round2.log
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "one_gpu or two_processes" > $O/pytest_dist.log 2>&1; echo "dist tests exit $?" | tee -a $O/pytest_dist.log
tail -25 $O/pytest_dist.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench exit $?"; cat $O/bench.json; tail -3 $O/bench.err
for c in 1 2 3 5; do timeout 300 python bench.py --config $c --steps 20 --warmup 2 >> $O/bench_configs.jsonl 2>> $O/bench.err; done; cat $O/bench_configs.jsonl | cut -c1-600
BENCH_FORCE_DIST=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_dist1.json 2>> $O/bench.err; echo "dist1 exit $?"; cut -c1-900 $O/bench_dist1.json
timeout 300 python bench.py --gpus 2 > $O/bench_gpus2.json 2> $O/bench_gpus2.err; echo "gpus2 exit $? (2 = refused, as it must on a 1-GPU box)"; tail -2 $O/bench_gpus2.err
