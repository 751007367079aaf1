"""Data-parallel step of the REAL engine on the GPU: two processes share the one device of the test box and
all-reduce over gloo (RCCL refuses two ranks on one device; the reducer, its side stream and the bucket events are the
same code path), then the result is compared with ONE process stepping on the joint batch.

ref: PL ``accelerator: ddp`` (config/trainer/trainer.yaml:6-12): the gradient of the mean loss over the global batch is
the average of the ranks' mean-loss gradients, so after one Adam step every replica holds the parameters of the
single-process run."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(dev, batch, seed_rank=None):
    from oracle import w2v2_oracle as O
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.params import ParamStore
    cfg = W2V2Config.tiny()
    st = ParamStore(cfg, dev, torch.float32, head="aam", num_speakers=10)
    st.init_weights(seed=3)
    reg = Wav2Vec2RegularisationConfig(attention_dropout=0.0, feat_proj_dropout=0.0, hidden_dropout=0.0, layerdrop=0.0,
                                       mask_time_prob=0.0)
    plan = Plan(st, batch, 4000, train=True, reg=reg)
    wav, label = O.synth_batch(4, 4000, 10, seed=11)          # the joint batch; rank r takes rows 2r, 2r+1
    return st, plan, OneCycle(max_lr=1e-3, total_steps=10), wav.to(dev), label.to(dev)


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    st, plan, sched, wav, label = _setup(dev, 2)
    tr = SpeakerTrainer(st, plan, sched)
    for _ in range(2):
        loss, _ = tr.train_step(wav[2 * rank:2 * rank + 2], label[2 * rank:2 * rank + 2])
    torch.cuda.synchronize()
    q.put((rank, st.flat[:st.n_train].cpu().numpy(), float(loss)))     # by value (no shared-memory handle)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_matches_single_process_on_joint_batch():
    import torch.multiprocessing as mp
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    st, plan, sched, wav, label = _setup(torch.device("cuda", 0), 4)
    tr = SpeakerTrainer(st, plan, sched)
    for _ in range(2):
        tr.train_step(wav, label)
    torch.cuda.synchronize()
    ref = st.flat[:st.n_train].cpu()
    p0, p1 = torch.from_numpy(res[0][1]), torch.from_numpy(res[1][1])
    assert torch.equal(p0, p1), "replicas diverged"
    moved = float((ref - _fresh_params()).norm())
    err = float((p0 - ref).norm())
    print(f"two-rank vs joint batch: |dp| = {moved:.3e}, |p_ddp - p_joint| = {err:.3e}")
    assert moved > 0 and err < 5e-4 * moved        # Adam normalises the update: compare against the step length


def _fresh_params():
    st, *_ = _setup(torch.device("cuda", 0), 1)
    return st.flat[:st.n_train].cpu()


# ------------------------------------------------------------------------------------------------------------------
# The benchmarked mode: fp16 activations under the dynamic loss scale, dropout, per-rank LayerDrop draws and
# SpecAugment masks.  An overflow that happens on ONE rank only must make BOTH ranks skip the step and halve the scale
# (the non-finite values travel through the SUM all-reduce into every replica's last bucket, which is what
# w2v2_grad_scaler_check scans), and the replicas must stay bit-identical throughout.
def _worker_fp16(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import w2v2_oracle as O
    from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
    from w2v2_speaker_amd.engine import Plan
    from w2v2_speaker_amd.optim.schedule import OneCycle
    from w2v2_speaker_amd.params import ParamStore
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    cfg = W2V2Config.tiny()
    st = ParamStore(cfg, dev, torch.float16, head="aam", num_speakers=10)
    st.init_weights(seed=3)
    st.scaler[0] = 1024.0
    reg = Wav2Vec2RegularisationConfig(attention_dropout=0.1, feat_proj_dropout=0.1, hidden_dropout=0.1, layerdrop=0.4,
                                       mask_time_prob=0.05, mask_time_length=2)
    plan = Plan(st, 2, 4000, train=True, reg=reg, seed=7 + rank)
    tr = SpeakerTrainer(st, plan, OneCycle(max_lr=1e-3, total_steps=10), layerdrop_seed=1234 + rank, mask_seed=7 + rank)
    wav, label = O.synth_batch(4, 4000, 10, seed=11)
    wav, label = wav.to(dev)[2 * rank:2 * rank + 2], label.to(dev)[2 * rank:2 * rank + 2]
    head_fb = plan.head_forward_backward
    record = []
    skips = []
    for step in range(5):
        if step == 2 and rank == 1:
            def poisoned(lbl):                     # an overflow on THIS rank only: inf in d(loss)/d(embedding)
                out = head_fb(lbl)
                plan.demb[0, 0] = float("inf")
                return out
            plan.head_forward_backward = poisoned
        else:
            plan.head_forward_backward = head_fb
        tr.train_step(wav, label)
        skips.append(tuple(plan._skip))
        torch.cuda.synchronize()
        record.append((float(st.scaler[0]), int(st.scaler[3])))
    q.put((rank, st.flat[:st.n_train].cpu().numpy(), record, skips))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_fp16_step_with_overflow_on_one_rank_skips_on_both():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker_fp16, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=300) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    (_, p0, rec0, sk0), (_, p1, rec1, sk1) = res
    p0, p1 = torch.from_numpy(p0), torch.from_numpy(p1)
    assert torch.isfinite(p0).all() and torch.equal(p0, p1), "replicas diverged or went non-finite"
    assert rec0 == rec1, (rec0, rec1)                          # identical scale / skip history on both ranks
    assert rec0[1] == (1024.0, 0) and rec0[2] == (512.0, 1) and rec0[4] == (512.0, 1), rec0
    assert sk0 != sk1, "the ranks were meant to draw different LayerDrop patterns"


def test_c_abi_collective_entry_points_on_rccl():
    """include/w2v2_hip.h "collective": w2v2_comm_unique_id / w2v2_comm_init / w2v2_allreduce_async / w2v2_comm_destroy
    resolve librccl.so at run time and run a real RCCL communicator.  The test box has ONE GPU (RCCL refuses two ranks
    per device), so this is the world-size-1 communicator: the all-reduce must leave the buffer unchanged, on a side
    stream, and the reducer built on it must drive a training step exactly like no reducer at all."""
    from w2v2_speaker_amd.comm import CAbiBucketAllReducer, RcclComm
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    uid = RcclComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = RcclComm(0, 1, 0, uid)
    x = torch.randn(1 << 20, device="cuda")
    ref = x.clone()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    comm.all_reduce_(x, side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    # w2v2_broadcast_async (start-up broadcast of the replica state): root 0 of a one-rank communicator, any dtype
    h = torch.randn(4097, device="cuda").to(torch.float16)
    href = h.clone()
    comm.broadcast_(h, 0, side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert torch.equal(h, href)
    with pytest.raises(RuntimeError):
        comm.broadcast_(h, 3)                                  # root outside the communicator
    dev = torch.device("cuda", 0)
    outs = []
    for use in (False, True):
        st, plan, sched, wav, label = _setup(dev, 4)
        red = CAbiBucketAllReducer(st, comm) if use else None
        tr = SpeakerTrainer(st, plan, sched, reducer=red)
        if use:
            red.world = 2                 # exercise the issue / wait path (a 1-rank SUM is the identity) ...
            tr.world = 1                  # ... without the 1/world gradient scaling
            before = st.flat.clone()
            red.broadcast_parameters(0)   # one rank: every state tensor comes back unchanged, operand copies rebuilt
            assert torch.equal(st.flat, before)
        for _ in range(2):
            tr.train_step(wav, label)
        torch.cuda.synchronize()
        outs.append(st.flat[:st.n_train].clone())
    assert torch.equal(outs[0], outs[1])
    comm.destroy()


@pytest.mark.parametrize("world", [2, 8])
def test_c_abi_reducer_world_n_path_on_a_loopback_communicator(world):
    """the world > 1 path of CAbiBucketAllReducer without hand-set fields.  w2v2_comm_init_loopback stands
    for rank 0 of a ``world``-rank job whose peers hold identical buffers (SUM = x world, broadcast = identity): the reducer
    reports that world, issues every bucket on its side stream, checks the rank-agreement signature, broadcasts the state
    and the host counters, the trainer scales by 1 / world -- and two training steps end bit-identical to a run with no
    reducer (x world and x 1/world are exact in f32)."""
    from w2v2_speaker_amd.comm import CAbiBucketAllReducer, RcclComm
    from w2v2_speaker_amd.trainer import SpeakerTrainer
    comm = RcclComm.loopback(world, 0)
    x = torch.randn(1 << 16, device="cuda")
    ref = x * world
    comm.all_reduce_(x)
    torch.cuda.synchronize()
    assert torch.equal(x, ref)
    dev = torch.device("cuda", 0)
    outs, issued = [], []
    for use in (False, True):
        st, plan, sched, wav, label = _setup(dev, 4)
        red = CAbiBucketAllReducer(st, comm) if use else None
        tr = SpeakerTrainer(st, plan, sched, reducer=red)
        if use:
            assert red.world == world and tr.world == world
            tr.step = 3
            st.set_step_counts(5, 4)
            got = tr.broadcast_state(0, [17])                  # signature all-reduce + state + host counters
            assert got == [17] and tr.step == 3 and (st.step_head, st.step_body) == (5, 4)
            tr.step = 0
            st.set_step_counts(0, 0)
            orig = red.bucket_ready
            red.bucket_ready = lambda name: (issued.append(name), orig(name))[1]
        for _ in range(2):
            tr.train_step(wav, label)
        torch.cuda.synchronize()
        outs.append(st.flat[:st.n_train].clone())
    assert issued[:1] == ["head"] and "projection" in issued
    assert torch.isfinite(outs[0]).all() and torch.equal(outs[0], outs[1])
    comm.destroy()
