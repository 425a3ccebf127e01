#!/bin/bash
# Round-6 GPU session (one gpurun call): scripts/gpu_round6.sh [tests] [bench] [published] [mlp64] [prof] [adjoint] [wide] [summaries]
# Everything lands in gpurun_out/r06/; rocprofv3 runs are bounded by `timeout` and write csv.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
cd $R
want() { [[ " $ARGS " == *" $1 "* ]]; }
ARGS="${*:-tests bench published mlp64 prof adjoint}"
if want tests; then
  timeout 1500 python -m pytest tests -q -m gpu -rfs 2>&1 | tail -25 > $O/r06_pytest_gpu_summary.txt
  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v Warn | tail -2 >> $O/r06_pytest_gpu_summary.txt
  tail -3 $O/r06_pytest_gpu_summary.txt
fi
if want bench; then
  timeout 600 python bench.py > $O/r06_bench_config4.json 2> $O/r06_bench_config4.err; tail -c 600 $O/r06_bench_config4.json
  timeout 300 python bench.py --config 5 --no-cpu-baseline > $O/r06_bench_config5.json 2>/dev/null
fi
if want published; then
  timeout 900 python bench.py --config published --published-all --steps 5 > $O/r06_published.jsonl 2> $O/r06_published.err; wc -l $O/r06_published.jsonl
fi
if want mlp64; then
  timeout 600 python scripts/bench_mlp64.py 2>&1 | grep -v Warn > $O/r06_mlp_f64.txt; cat $O/r06_mlp_f64.txt
fi
if want prof; then
  cat > /tmp/p64.py <<PY
import os, sys, torch
sys.path.insert(0, "$R")
from tfdiffeq_amd import models, odeint
dev = torch.device('cuda:0'); torch.manual_seed(0)
blk = models.ODEBlock(models.ODEFunc(64, 128, non_linearity='tanh'), tol=1e-3).to(dev).double()
for batch in (4096, 32768):
    x = torch.randn(batch, 64, dtype=torch.float64, device=dev)
    with torch.no_grad():
        for _ in range(200): blk(x)
torch.cuda.synchronize()
PY
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_mlp64 -o r -- python /tmp/p64.py > $O/prof_mlp64.log 2>&1)
  f=$(find $O/prof_mlp64 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/r06_mlp_f64_kernel_stats.csv && head -5 $O/r06_mlp_f64_kernel_stats.csv | cut -c1-220
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mlp64 -o r -- python /tmp/p64.py > $O/pmc_mlp64.log 2>&1)
  f=$(find $O/pmc_mlp64 -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" > $O/r06_mlp_f64_sq_pmc.txt <<PY
import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r['Kernel_Name'][:90]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    if 'mlp64' in k:
        print(k, {c: '%.4g avg over %d dispatches (max %.4g)' % (sum(v) / len(v), len(v), max(v)) for c, v in d.items()})
PY
  cat $O/r06_mlp_f64_sq_pmc.txt 2>/dev/null
  rm -rf $O/prof_mlp64 $O/pmc_mlp64
fi
if want adjoint; then
  # the fused MLP adjoint (k_adjoint_mlp) and config 5's forward kernel again (the review of round 5: no profile of the adjoint kernel since round 4)
  for td in 0; do
    D=$O/prof_adj$td; rm -rf $D
    (cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && ADJ_TD=$td timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $R/scripts/adjoint_train_step.py fused 10 > $O/r06_adjoint_td${td}_train_step.txt 2>&1)
    f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -12 "$f" > $O/r06_adjoint_td${td}_kernel_stats.csv
    grep ms_per $O/r06_adjoint_td${td}_train_step.txt | cut -c1-250
    rm -rf $D
  done
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c5 -o r -- python $R/bench.py --config 5 --steps 200 --warmup 5 --no-cpu-baseline > $O/prof_c5.log 2>&1)
  f=$(find $O/prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" > $O/r06_c5_kernel_stats.csv && head -3 $O/r06_c5_kernel_stats.csv | cut -c1-200
  rm -rf $O/prof_c5
  (cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_c4 -o r -- python $R/bench.py --steps 100 --warmup 5 --no-cpu-baseline > $O/prof_c4.log 2>&1)
  f=$(find $O/prof_c4 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -6 "$f" > $O/r06_whole_kernel_stats.csv && head -3 $O/r06_whole_kernel_stats.csv | cut -c1-200
  rm -rf $O/prof_c4
fi
if want wide; then
  # linear right-hand side at dims 129 .. 256 (the 256-wide tile kernels, W streamed): tests, times, kernel trace
  timeout 600 python -m pytest tests/test_gpu_linear_wide.py -q -x 2>&1 | tail -3 > $O/r06_linear_wide_tests.txt; cat $O/r06_linear_wide_tests.txt
  timeout 300 python scripts/bench_linear_wide.py bench valu 2>&1 | grep -v Warn > $O/r06_linear_wide.txt; cat $O/r06_linear_wide.txt
  (cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_wide -o r -- python $R/scripts/bench_linear_wide.py bench > $O/prof_wide.log 2>&1)
  f=$(find $O/prof_wide -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" > $O/r06_linear_wide_kernel_stats.csv && head -5 $O/r06_linear_wide_kernel_stats.csv | cut -c1-220
  rm -rf $O/prof_wide
fi
if want summaries; then
  # the committed rocprofv3 passes bench.py cites (profiles/r06_{whole,c5}_summary.json): kernel trace + FETCH_SIZE / WRITE_SIZE passes
  run() { TAG=$1; shift; bash scripts/gpu_prof.sh $TAG "$@" > gpurun_out/prof_$TAG.out 2>&1; python scripts/pmc_summary.py gpurun_out $TAG $O/r06_$TAG --no-raw | tail -4; rm -rf gpurun_out/prof_$TAG gpurun_out/pmc_${TAG}_*; }
  run whole
  run c5 --config 5
fi
