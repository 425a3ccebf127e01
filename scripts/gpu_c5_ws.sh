#!/bin/bash
# Config 5, the review's item 4 (round 5): the weights-streamed variant of k_persist_mlp (csrc/mi_ode_mlp.h, WS; MI_ODE_MLP_STREAM=1: two
# workgroups per CU at 128 registers, =2: one workgroup per CU) against the product kernel: bench line, rocprofv3 kernel stats, SQ counters.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r06
mkdir -p $O
OUT=$O/r06_c5_ws_variant.txt
: > $OUT
cd $R
for v in 0 2 1; do
  echo "== MI_ODE_MLP_STREAM=$v ($( [ $v = 0 ] && echo 'product: weight slices resident in registers, one workgroup per CU' || ( [ $v = 2 ] && echo 'weights streamed, one workgroup per CU' || echo 'weights streamed, two workgroups per CU (128 registers)')))" >> $OUT
  MI_ODE_MLP_STREAM=$v timeout 200 python bench.py --config 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench.py --config 5: ms_per_step %.4f  kernel (HIP events) %.4f ms  frac of fp32 MFMA peak %.4f  attempts %s launches %s' % (d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['attempts_per_step'], d['config']['kernel_launches']))" >> $OUT
  D=/tmp/c5ws_$v; rm -rf $D
  (cd /tmp && export TMPDIR=/tmp && MI_ODE_MLP_STREAM=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o r -- python $R/bench.py --config 5 --steps 200 --warmup 5 --no-cpu-baseline > /dev/null 2>&1)
  f=$(find $D -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && sed -n 2,3p "$f" | cut -c1-160 >> $OUT
  for C in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU"; do
    P=/tmp/c5pmc_$v; rm -rf $P
    (cd /tmp && export TMPDIR=/tmp && MI_ODE_MLP_STREAM=$v timeout 300 rocprofv3 --pmc $C --output-format csv -d $P -o r -- python $R/bench.py --config 5 --steps 20 --warmup 2 --no-cpu-baseline > /dev/null 2>&1)
    g=$(find $P -name "*counter_collection.csv" | head -1)
    [ -n "$g" ] && python - "$g" >> $OUT <<'PY'
import collections, csv, sys
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_persist_mlp' in r['Kernel_Name']:
        acc[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in acc.items():
    print('  %-22s %.5g per launch (avg of %d)' % (c, sum(v) / len(v), len(v)))
PY
  done
done
cat $OUT
