"""Synthetic speaker-verification trial set for the "eval EER" half of the metric ("eval EER on a fixed
synthetic trial list").  VoxCeleb is not available offline, so the trial list the reference scores
(ref: src/evaluation/speaker/speaker_recognition_evaluator.py:46-115 on `veri_test2.txt` pairs) is replaced by a
structured one that any host can regenerate bit for bit from its seed (numpy PCG64, no torch RNG):

  * ``n_speakers`` speakers, each a fixed unit-variance noise waveform v_s[n] (the speaker's "template");
  * utterance u of speaker s = ``mix`` * v_s + sqrt(1 - mix^2) * fresh white noise, normalised per utterance exactly as
    the reference's ``InputNormalizer2D`` does (ref: src/data/preprocess/input_normalisation.py:54-67).  With random
    (untrained) weights the network is no speaker model: what makes two utterances score high is that their waveforms
    are close, so ``mix`` sets the difficulty -- 0.9 puts the reference's own EER at 0.11 (0.97: 0, 0.7: 0.42), i.e.
    target and non-target scores overlap and the EER reacts to embedding errors.  Only IEEE adds / multiplies /
    one sqrt in a fixed order (float64, rounded once to f32): every host regenerates the same samples;
  * trials = ALL pairs of distinct utterances, target iff same speaker (8 x 4 utterances: 496 trials, 48 targets).

The same waveforms go through the reference (tests/golden/make_goldens.py `eer` -> tests/golden/g12_eer.npz holds ITS
embeddings, scores and EER) and through the HIP path (tests/test_parity_gpu.py, bench.py `eer` field)."""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np

TRIAL_SET_DEFAULT = dict(n_speakers=8, utts_per_speaker=4, n_samples=48000, mix=0.9, seed=52001)


def synth_trial_set(n_speakers: int = 8, utts_per_speaker: int = 4, n_samples: int = 48000, mix: float = 0.9,
                    seed: int = 52001) -> Tuple[np.ndarray, np.ndarray, List[str], List[Tuple[int, int, int]]]:
    """-> (wav [S*U, n_samples] f32 normalised, speaker [S*U] int64, keys, trials [(same, i, j)] over all i < j)."""
    g = np.random.Generator(np.random.PCG64(seed))
    voices = g.standard_normal((n_speakers, n_samples))
    wav = np.empty((n_speakers * utts_per_speaker, n_samples), dtype=np.float32)
    spk = np.empty(n_speakers * utts_per_speaker, dtype=np.int64)
    keys = []
    a, b = float(mix), float(np.sqrt(1.0 - mix * mix))
    for s in range(n_speakers):
        for u in range(utts_per_speaker):
            x = a * voices[s] + b * g.standard_normal(n_samples)
            mean = x.sum() / n_samples
            var = ((x - mean) ** 2).sum() / (n_samples - 1)    # unbiased, like torch.std_mean in the reference
            i = s * utts_per_speaker + u
            wav[i] = ((x - mean) / (np.sqrt(var) + 1e-5)).astype(np.float32)
            spk[i] = s
            keys.append(f"id{s:05d}/synth/{u:05d}")
    n = len(keys)
    trials = [(int(spk[i] == spk[j]), i, j) for i in range(n) for j in range(i + 1, n)]
    return wav, spk, keys, trials


def score_trials(emb, trials) -> Tuple[List[int], List[float]]:
    """Cosine score of every trial mapped to [0, 1] as the reference's evaluator does before the EER
    (ref: speaker_recognition_evaluator.py:81 ``clip((s + 1) / 2, 0, 1)``) -> (ground truth, scores)."""
    e = np.asarray(emb, dtype=np.float64)
    en = e / np.maximum(np.linalg.norm(e, axis=1, keepdims=True), 1e-8)
    gt = [t[0] for t in trials]
    sc = [float(np.clip((en[i] @ en[j] + 1.0) / 2.0, 0.0, 1.0)) for _, i, j in trials]
    return gt, sc


# --------------------------------------------------------------------------- synthetic weights (no checkpoint offline)
def _name_seed(name: str, seed: int) -> int:
    h = 1469598103934665603
    for ch in (name + f"#{seed}").encode():
        h = ((h ^ ch) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def synth_weight(name: str, shape: Sequence[int], seed: int) -> np.ndarray:
    """Deterministic name-keyed PCG64 parameter (f32): the golden script (tests/golden/make_goldens.py, which loads these
    weights into the REFERENCE's model), the CPU oracle and the GPU box regenerate identical weights from (name, seed),
    so no 378 MB checkpoint is ever committed.  ``name`` = the HF state-dict key without the
    ``wav2vec.model.`` prefix, or ``loss_fn.fc_weights`` / ``fc_list.*``."""
    g = np.random.Generator(np.random.PCG64(_name_seed(name, seed)))
    shape = tuple(shape)
    leaf = name.rsplit(".", 1)[-1]
    if name == "masked_spec_embed":
        a = g.random(shape)                                     # HF:1253 uniform_()
    elif name.endswith("parametrizations.weight.original0"):
        a = 1.0 + 0.25 * g.standard_normal(shape)               # weight-norm gain g
    elif "layer_norm" in name and leaf == "weight":
        a = 1.0 + 0.1 * g.standard_normal(shape)
    elif leaf == "bias":
        a = 0.1 * g.standard_normal(shape)
    elif name.startswith("feature_extractor") and leaf == "weight":
        fan_in = shape[1] * shape[2]
        a = math.sqrt(2.0 / fan_in) * 1.3 * g.standard_normal(shape)   # keep variance through GELU
    elif name.endswith("original1"):
        fan_in = shape[1] * shape[2]
        a = math.sqrt(1.0 / fan_in) * g.standard_normal(shape)
    elif len(shape) == 2:
        a = math.sqrt(1.0 / shape[1]) * g.standard_normal(shape)
    else:
        a = g.standard_normal(shape)
    return np.ascontiguousarray(a, dtype=np.float32)


def outlier_family(sd: Dict[str, np.ndarray], seed: int, prefix: str = "") -> Dict[str, np.ndarray]:
    """A HEAVY-TAILED variant of a synthetic state dict.  Every golden up to g16 draws its weights from one
    well-conditioned family (LayerNorm gains 1 +- 0.1, 1/sqrt(fan-in) matrices); the pretrained wav2vec2-base is known for a
    handful of residual channels that carry activations tens of times larger than the rest and for large pre-projection
    activations.  Imitated here, deterministically from ``seed``:
      * six hidden channels: the gain of EVERY encoder LayerNorm (prologue, both norms of each block) x 20 on them,
      * four entries of every FFN-1 bias + 8,
      * the convolutions 1, 3, 5 of the feature extractor x 3 (27x larger features in front of the projection's LayerNorm).
    In place on a copy; ``sd`` keys are HF names with an optional ``prefix`` (as ParamStore.shapes has them)."""
    out = {k: np.array(v, copy=True) for k, v in sd.items()}
    g = np.random.Generator(np.random.PCG64(_name_seed("outlier_family", seed)))
    hidden = out[prefix + "encoder.layer_norm.weight"].shape[0]
    chans = np.sort(g.choice(hidden, size=min(6, hidden), replace=False))
    for k in out:
        n = k[len(prefix):] if prefix and k.startswith(prefix) else k
        if n.startswith("encoder.") and n.endswith("layer_norm.weight"):
            out[k][chans] *= 20.0
        elif n.endswith("feed_forward.intermediate_dense.bias"):
            idx = g.choice(out[k].shape[0], size=min(4, out[k].shape[0]), replace=False)
            out[k][idx] += 8.0
        elif n in ("feature_extractor.conv_layers.1.conv.weight", "feature_extractor.conv_layers.3.conv.weight",
                   "feature_extractor.conv_layers.5.conv.weight"):
            out[k] *= 3.0
    return out


def synth_state_dict(shapes: Dict[str, Sequence[int]], seed: int, prefix: str = "wav2vec.model.") -> Dict[str, np.ndarray]:
    """``shapes`` as ``ParamStore.shapes`` has them (reference-style keys): every parameter drawn from ``synth_weight``
    under its HF name (``prefix`` stripped) -- the weights the reference goldens were generated with."""
    out = {}
    for n, shp in shapes.items():
        key = n[len(prefix):] if n.startswith(prefix) else n
        out[n] = synth_weight(key, shp, seed)
    return out
