export PYTHONPATH=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0
echo "== 1-rank nccl group (BENCH_FORCE_DIST)"; BENCH_FORCE_DIST=1 timeout 300 python bench.py --no-cpu-baseline --steps 3 --warmup 1 2> gpurun_out/r04_dist1.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['ms_per_step'], c['cross_rank']); print(json.dumps(c.get('per_rank'))[:900]); print(json.dumps(c.get('transport_survey'))[:2500])"; tail -3 gpurun_out/r04_dist1.err
for sc in weak strong; do
echo "== 2 ranks sharing the GPU, gloo, $sc"; BENCH_SHARE_GPU=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --batch $([ $sc = weak ] && echo 800 || echo 1600) --scaling $sc --steps 3 --warmup 1 2> gpurun_out/r04_dist2_$sc.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']; print(d['n_gpus'], d['ms_per_step'], d['value'], c['cross_rank']); print(json.dumps(c.get('per_rank'))[:1500]); print(json.dumps(c.get('transport_survey'))[:3000])"; tail -3 gpurun_out/r04_dist2_$sc.err
done
echo "== config 4 with the CPU baseline"; python bench.py 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step']); print(json.dumps(d['cpu_baseline']['torch_threads'])); print(d['cpu_baseline']['sample'])"
timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k cubic 2>&1 | tail -2
