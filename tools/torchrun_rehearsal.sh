W2V2_DIST_BACKEND=gloo W2V2_SHARE_GPU=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 6 --warmup 2 > gpurun_out/torchrun2.json 2> gpurun_out/torchrun2.err
echo rc=$?
tail -c 300 gpurun_out/torchrun2.err
python - <<'PY'
import json
d=json.loads([l for l in open("gpurun_out/torchrun2.json") if l.startswith("{")][-1])
print(d["n_gpus"], d["value"], d["ms_per_step"], d["rccl"]["backend"], [r["allreduce_ok"] for r in d["rccl"]["ranks"]], d["ddp"]["step_time_ratio_vs_solo"])
PY
