#!/usr/bin/env python3
"""First command of an N-GPU lease: N ranks, one per GPU, through BOTH gradient reducers.

  1. torch.distributed "nccl" (= RCCL) process group and a C-ABI communicator (w2v2_comm_* over librccl.so, id through
     a TCPStore) side by side; every rank reports its device -- the run fails unless N distinct devices answer;
  2. the same random f32 buffer (16 MiB) summed by `dist.all_reduce` and by `w2v2_allreduce_async`: bitwise equal at
     N = 2 (a two-term sum has one order), within 1e-6 relative beyond (ring order may differ);
  3. two fp16 training steps of the tiny model from identical replicas with each reducer (dropout, per-rank LayerDrop
     and masks): replicas bit-identical across ranks under either reducer, and the two reducers agree with each other
     (bitwise at N = 2);
  4. the start-up broadcast of both reducers carries rank 0's parameters AND its host step counters.

Prints one JSON line on rank 0; exit code != 0 on any failure.   python tools/ddp_selfcheck.py --gpus 2
(ref: config/trainer/trainer.yaml:6-12 -- PL `accelerator: ddp`)"""
import argparse
import json
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def _free_port():
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def worker(rank, world, port, port2, q):
    import numpy as np
    import torch
    import torch.distributed as dist
    from w2v2_speaker_amd.comm import CAbiBucketAllReducer, RcclComm
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import BucketAllReducer, SpeakerTrainer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    comm = RcclComm.from_store(rank, world, rank, port=port2)
    res = {"rank": rank, "device": torch.cuda.get_device_name(rank), "pci": torch.cuda.get_device_properties(rank).pci_bus_id
           if hasattr(torch.cuda.get_device_properties(rank), "pci_bus_id") else rank}
    # 2. the two collectives on the same data
    g = torch.Generator(device="cpu").manual_seed(100 + rank)
    x = (torch.randn(1 << 22, generator=g) * 3).to(dev)
    a, b = x.clone(), x.clone()
    dist.all_reduce(a)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_(b, side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    res["allreduce_bitwise_equal"] = bool(torch.equal(a, b))
    res["allreduce_rel_diff"] = float((a - b).norm() / a.norm())
    # 3./4. training steps with each reducer
    finals = {}
    for kind in ("torch", "cabi"):
        cfg = W2V2Config.tiny()
        st = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=10)
        st.init_weights(seed=3 + 17 * rank)                    # replicas start DIFFERENT: the broadcast must fix it
        if rank == 0:
            st.set_step_counts(5, 5)
        st.scaler[0] = 1024.0
        reg = Wav2Vec2RegularisationConfig(attention_dropout=0.1, feat_proj_dropout=0.1, hidden_dropout=0.1, layerdrop=0.3,
                                           mask_time_prob=0.05, mask_time_length=2)
        plan = Plan(st, 2, 4000, train=True, reg=reg, seed=7 + rank)
        red = BucketAllReducer(st) if kind == "torch" else CAbiBucketAllReducer(st, comm)
        tr = SpeakerTrainer(st, plan, OneCycle(max_lr=1e-3, total_steps=20), layerdrop_seed=1234 + rank, mask_seed=7 + rank,
                            reducer=red)
        if rank == 0:
            tr.step = 9
        tr.broadcast_state(0)
        res[kind + "_host_counters_ok"] = (tr.step, st.step_head, st.step_body) == (9, 5, 5)
        gw = torch.Generator(device="cpu").manual_seed(11)
        wav = torch.randn(2 * world, 4000, generator=gw)
        wav = ((wav - wav.mean(1, keepdim=True)) / (wav.std(1, keepdim=True) + 1e-5)).to(dev)[2 * rank:2 * rank + 2]
        label = torch.randint(0, 10, (2 * world,), generator=gw).to(dev)[2 * rank:2 * rank + 2]
        for _ in range(2):
            tr.train_step(wav, label)
        torch.cuda.synchronize()
        mine = st.flat[:st.n_train].clone()
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        res[kind + "_replicas_identical"] = all(bool(torch.equal(allp[0], p)) for p in allp[1:])
        res[kind + "_finite"] = bool(torch.isfinite(mine).all())
        finals[kind] = mine
    res["reducers_bitwise_equal"] = bool(torch.equal(finals["torch"], finals["cabi"]))
    res["reducers_rel_diff"] = float((finals["torch"] - finals["cabi"]).norm() / finals["torch"].norm())
    q.put(res)
    dist.barrier()
    comm.destroy()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=2)
    a = ap.parse_args()
    import torch
    import torch.multiprocessing as mp
    n_dev = torch.cuda.device_count()                       # (counting devices does not initialise the GPU)
    if n_dev < a.gpus:
        print(json.dumps({"ok": False, "error": f"{a.gpus} ranks asked, {n_dev} devices visible"}))
        raise SystemExit(2)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port, port2 = _free_port(), _free_port()
    procs = [ctx.Process(target=worker, args=(r, a.gpus, port, port2, q)) for r in range(a.gpus)]
    for p in procs:
        p.start()
    res = []
    try:
        for _ in procs:
            res.append(q.get(timeout=600))
    except Exception as ex:
        for p in procs:
            p.terminate()
        print(json.dumps({"ok": False, "error": f"a rank died or hung: {ex!r}", "ranks": res}))
        raise SystemExit(3)
    for p in procs:
        p.join(timeout=120)
    res.sort(key=lambda r: r["rank"])
    two = a.gpus == 2
    ok = (len(res) == a.gpus and len({str(r["pci"]) for r in res}) == a.gpus
          and all(r["allreduce_rel_diff"] < 1e-6 and (r["allreduce_bitwise_equal"] or not two) for r in res)
          and all(r[k + "_replicas_identical"] and r[k + "_finite"] and r[k + "_host_counters_ok"]
                  for r in res for k in ("torch", "cabi"))
          and all(r["reducers_rel_diff"] < 1e-5 and (r["reducers_bitwise_equal"] or not two) for r in res)
          and all((p.exitcode or 0) == 0 for p in procs))
    print(json.dumps({"ok": ok, "n_gpus": a.gpus, "ranks": res}))
    raise SystemExit(0 if ok else 1)


if __name__ == "__main__":
    main()
