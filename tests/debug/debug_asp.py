import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import w2v2_oracle as O
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.params import ParamStore
from test_parity_gpu import _asp_weights, _no_reg
cfg, ocfg = W2V2Config.tiny(), O.OracleConfig.tiny()
B, N, C = 8, 4000, 10
res = {}
for dtype in (torch.float32, torch.bfloat16):
    st = ParamStore(cfg, "cuda", dtype, head="aam", num_speakers=C, attentive_pool=True)
    sd = O.make_state_dict(ocfg, 20211)
    sd["loss_fn.fc_weights"] = O.synth_tensor("loss_fn.fc_weights", (C, st.embed_dim), 20211)
    asd, aod = _asp_weights(st)
    sd.update(asd)
    st.load_state_dict(sd)
    wav, label = O.synth_batch(B, N, C, seed=11)
    tr = Plan(st, B, N, train=True, reg=_no_reg(), pooling="attentive")
    st.zero_grad()
    emb = tr.embed(wav.cuda())
    loss, _ = tr.head_forward_backward(label.cuda())
    a = tr._asp_cur
    demb = tr.demb.clone()
    a.backward(tr.demb)
    torch.cuda.synchronize()
    res[dtype] = {k: getattr(a, k).float().cpu().clone() for k in ("ctx", "cb", "a_pre", "h", "s", "emb", "ds", "dh", "da")}
    res[dtype]["demb"] = demb.float().cpu()
    res[dtype]["G"] = tr.G.float().cpu().clone()
    from w2v2_speaker_amd.asp import ASP_PREFIX as AP
    gw = st.g(AP + "tdnn.conv.conv.weight").view(128, -1).float().cpu().clone()
    Cc = gw.shape[1] // 3
    res[dtype]["dW1x"], res[dtype]["dW1m"], res[dtype]["dW1s"] = gw[:, :Cc], gw[:, Cc:2*Cc], gw[:, 2*Cc:]
    res[dtype]["db1"] = st.g(AP + "tdnn.conv.conv.bias").float().cpu().clone()
    res[dtype]["dW2"] = st.g(AP + "conv.conv.weight").float().cpu().clone()
    res[dtype]["x"] = a.x.float().cpu().clone()
def rl(a, b): return float((a - b).norm() / (b.norm() + 1e-30))
for k in res[torch.float32]:
    print(f"{k:6s} bf16 vs f32 rel-L2 {rl(res[torch.bfloat16][k], res[torch.float32][k]):.3e}   |f32| {float(res[torch.float32][k].norm()):.3e}")
# recompute da in f64 from the bf16 run's own inputs
from w2v2_speaker_amd.asp import ASP_PREFIX
r = res[torch.bfloat16]
gam = st.p(ASP_PREFIX + "tdnn.norm.norm.weight").double().cpu(); bet = st.p(ASP_PREFIX + "tdnn.norm.norm.bias").double().cpu()
ap, dh = r["a_pre"].double(), r["dh"].double()
rl_ = ap.clamp_min(0)
mu, var = rl_.mean(0), rl_.var(0, unbiased=False)
rstd = (var + 1e-5).rsqrt()
rh = (rl_ - mu) * rstd
y = torch.tanh(rh * gam + bet)
dz = dh * (1 - y * y)
dr = gam * rstd * (dz - dz.mean(0) - rh * (dz * rh).mean(0))
da_ref = dr * (ap > 0)
print("da kernel vs f64 recompute from same bf16 inputs:", rl(r["da"].double(), da_ref))
mr = a.mean_rstd.double().cpu()
print("mean err", float((mr[:, 0] - mu).abs().max()), "rstd rel err", float(((mr[:, 1] - rstd) / rstd).abs().max()))
r32 = res[torch.float32]
ap32, dh32 = r32["a_pre"].double(), r32["dh"].double()
print("relu mask mismatches", int(((ap > 0) != (ap32 > 0)).sum()), "of", ap.numel())
