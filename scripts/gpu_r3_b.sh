#!/bin/bash
# Round 3, GPU visit B: full GPU suite on the lean config-4 pass, bench, in-kernel timeline of the attempt pass
cd "${GRAFT_REPO_ROOT:-/root/repo}"; mkdir -p gpurun_out/r3b; export PYTHONPATH=$PWD; export TMPDIR=/tmp; O=gpurun_out/r3b
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest_gpu.log
python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json
TFDIFFEQ_AMD_LIB=$PWD/tfdiffeq_amd/_variants/libmi_ode_trace.so python bench.py --no-cpu-baseline --steps 1 --warmup 0 2>&1 | grep "\[trace\]" | cut -c1-140 > $O/config4_timeline.txt; head -16 $O/config4_timeline.txt
