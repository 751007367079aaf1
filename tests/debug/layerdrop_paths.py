"""Debug helper: 12 training steps with LayerDrop 0.5 (every held / skipped / paired combination of the backward)
printing the loss trajectory; run under W2V2_NO_LN_FOLD=1 / W2V2_NO_DEFER=1 / W2V2_NO_WGRAD_PAIRS=1 to check that the
deferred folds, deferred stores and pair launches do not change a single bit."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from w2v2_speaker_amd.config import W2V2Config, Wav2Vec2RegularisationConfig
from w2v2_speaker_amd.engine import Plan
from w2v2_speaker_amd.optim.schedule import OneCycle
from w2v2_speaker_amd.params import ParamStore
from w2v2_speaker_amd.trainer import SpeakerTrainer

dev = torch.device("cuda", 0)
cfg = W2V2Config.from_huggingface_id("facebook/wav2vec2-base")
store = ParamStore(cfg, dev, torch.bfloat16, head="aam", num_speakers=5994, freeze_cnn=True, embed_dim=2 * cfg.hidden_size)
store.init_weights(seed=20211)
reg = Wav2Vec2RegularisationConfig(layerdrop=0.5)
plan = Plan(store, 8, 48000, train=True, reg=reg, seed=7)
tr = SpeakerTrainer(store, plan, OneCycle(max_lr=5e-5, total_steps=20), layerdrop_seed=3, mask_seed=7)
g = torch.Generator().manual_seed(1)
wav = torch.randn(8, 48000, generator=g).to(dev)
label = torch.randint(0, 5994, (8,), generator=g).to(dev)
out = []
for _ in range(12):
    loss, _ = tr.train_step(wav, label)
    out.append(float(loss))
print(" ".join(f"{v:.6f}" for v in out), "| grad checksum", float(store.grad.double().abs().sum()))
