"""Debug: where does utterance i's forward first differ between a B=66 plan and a B=1 plan?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from oracle import w2v2_oracle as O
from w2v2_speaker_amd.config import W2V2Config
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.params import ParamStore

dev = "cuda"
dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 66
cfg, ocfg = W2V2Config(), O.OracleConfig.base()
st = ParamStore(cfg, dev, dtype, head=None)
st.load_state_dict(O.make_state_dict(ocfg, 20211))
wav, _ = O.synth_batch(B, 48000, 10, seed=1)
wav = wav.to(dev)
big = Plan(st, B, 48000, train=True)          # train=True keeps every layer's buffers
big.train_flag = False
from w2v2_speaker_amd.config import Wav2Vec2RegularisationConfig
noreg = Wav2Vec2RegularisationConfig(attention_dropout=0, feat_proj_dropout=0, hidden_dropout=0, layerdrop=0, mask_time_prob=0)
big = Plan(st, B, 48000, train=True, reg=noreg)
one = Plan(st, 1, 48000, train=True, reg=noreg)
big.embed(wav)
torch.cuda.synchronize()
for i in (0, B // 2, B - 1):
    one.embed(wav[i:i + 1])
    torch.cuda.synchronize()
    T, H = big.T, cfg.hidden_size
    def cmp(name, a, b):
        a, b = a.float(), b.float()
        d = (a - b).abs().max().item()
        print(f"  utt {i:2d} {name:12s} maxabs diff {d:.3e}  equal={torch.equal(a, b)}")
    for li in range(7):
        cmp(f"conv{li}", big.conv[li][i], one.conv[li][0])
    cmp("ln_feat", big.ln_feat.view(B, T, -1)[i], one.ln_feat.view(1, T, -1)[0])
    cmp("h0", big.h0.view(B, T, H)[i], one.h0.view(1, T, H)[0])
    cmp("pos(s0)", big.pos.view(B, T, H)[i], one.pos.view(1, T, H)[0])
    cmp("X0", big.X[0].view(B, T, H)[i], one.X[0].view(1, T, H)[0])
    lb, lo = big.lb[0], one.lb[0]
    cmp("qkv0", lb.qkv.view(B, T, -1)[i], lo.qkv.view(1, T, -1)[0])
    cmp("ctx0", lb.ctx.view(B, T, -1)[i], lo.ctx.view(1, T, -1)[0])
    cmp("s1_0", lb.a.view(B, T, -1)[i], lo.a.view(1, T, -1)[0])
    cmp("x1_0", lb.x1.view(B, T, -1)[i], lo.x1.view(1, T, -1)[0])
    cmp("h_0", lb.h.view(B, T, -1)[i], lo.h.view(1, T, -1)[0])
    cmp("X1", big.X[1].view(B, T, H)[i], one.X[1].view(1, T, H)[0])
    cmp("X12", big.X[12].view(B, T, H)[i], one.X[12].view(1, T, H)[0])
    cmp("emb", big.emb[i], one.emb[0])
