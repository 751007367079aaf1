import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.optim.schedule import Constant
from w2v2_speaker_amd.params import ParamStore
from w2v2_speaker_amd.trainer import SpeakerTrainer
from oracle import w2v2_oracle as O
DEV = "cuda"
cfg = W2V2Config.tiny()
reg = Wav2Vec2RegularisationConfig(activation_dropout=0.0, attention_dropout=0.0, feat_proj_dropout=0.0,
                                   hidden_dropout=0.0, layerdrop=0.0, mask_time_prob=0.0)
wav, label = O.synth_batch(3, 4000, 10, seed=11)
wav, label = wav.to(DEV), label.to(DEV)
grads = []
for selective in (False, True, False, True):
    st = ParamStore(cfg, DEV, torch.float16, head="aam", num_speakers=10)
    st.init_weights(5)
    st.scaler[0] = 256.0
    plan = Plan(st, 3, 4000, train=True, reg=reg)
    tr = SpeakerTrainer(st, plan, Constant(0.0))
    if not selective:
        st.zero_grad = (lambda orig: (lambda skip_layers=None: orig(None)))(st.zero_grad)
    for name, off in st.offsets.items():
        if off < st.n_train:
            st.grad[off:off + int(np.prod(st.shapes[name]))] = float("nan")
    tr.train_step(wav, label, skip_layers=(0,))
    tr.train_step(wav, label, skip_layers=(1,))
    torch.cuda.synchronize()
    grads.append(st.grad.clone())
for a, b, nm in ((0, 1, "full vs selective"), (0, 2, "full vs full"), (1, 3, "sel vs sel")):
    print("==", nm, torch.equal(grads[a], grads[b]))
    for name, off in st.offsets.items():
        if off < st.n_train:
            n = int(np.prod(st.shapes[name]))
            x, y = grads[a][off:off + n], grads[b][off:off + n]
            if not torch.equal(x, y):
                print("   ", name, float((x - y).abs().max()), float(x.abs().max()))
