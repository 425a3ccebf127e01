#!/bin/bash
# The N > 1 path of bench.py on the one-GPU test box: a 1-rank RCCL group, then 2, 4 and 8 ranks sharing device 0 over a gloo group
# (BENCH_SHARE_GPU=1), weak and strong.  The shards are small enough for every rank's persistent grid to be co-resident (8 x 25
# workgroups); what is exercised is the plumbing - sharding, transports, per-rank logs, the survey - not a scaling curve.
export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
show='import json,sys; d=json.loads(sys.stdin.read()); c=d["config"]; print(d["n_gpus"], "ranks:", round(d["ms_per_step"],3), "ms/step", int(d["value"]), "elements/s; records_transport:", c["records_transport"], "| rccl_ranks", c["rccl_ranks"], "| backend", c["process_group_backend"]); print(" per rank:", json.dumps([{k: (round(v,3) if isinstance(v,float) else v) for k,v in r.items() if k in ("rank","rows","attempts","launches","ms_per_step","handoff_us","cross_rank")} for r in (c.get("per_rank") or [])])[:1600]); print(" survey:", json.dumps([{k: (round(v,3) if isinstance(v,float) else v) for k,v in e.items() if k in ("requested","ran","ms_per_step","launches","error")} for e in (c.get("transport_survey") or [])])[:1600])'
echo "== 1-rank nccl group (BENCH_FORCE_DIST)"; BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/r05_dist1.err | python -c "$show"; tail -2 gpurun_out/r05_dist1.err
for n in 2 4 8; do
for sc in weak strong; do
rows=$([ $sc = weak ] && echo 400 || echo $((400 * n)))
echo "== $n ranks sharing the GPU, gloo, $sc ($rows rows per rank / in total)"
BENCH_SHARE_GPU=1 BENCH_NO_SMI_LOOP=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29570 + n)) bench.py --gpus $n --batch $rows --scaling $sc --steps 3 --warmup 1 2> gpurun_out/r05_dist${n}_$sc.err | python -c "$show"; grep -v "amdgpu.ids\|^$" gpurun_out/r05_dist${n}_$sc.err | tail -2
done
done
