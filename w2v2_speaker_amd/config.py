"""Shape / regularisation configuration of the wav2vec2 speaker path.

Mirrors the keys the reference reads: HF ``Wav2Vec2Config`` fields used by ``facebook/wav2vec2-base``
(/ ``-large``) and ``Wav2Vec2RegularisationConfig`` (ref: src/models/wav2vec2.py:83-94,
config/network/wav2vec2_fc.yaml:5-73)."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Tuple


@dataclass
class Wav2Vec2RegularisationConfig:
    """ref: src/models/wav2vec2.py:83-94 (same field names and defaults)."""
    gradient_checkpointing: bool = False
    activation_dropout: float = 0.0
    attention_dropout: float = 0.1
    feat_proj_dropout: float = 0.1
    hidden_dropout: float = 0.1
    layerdrop: float = 0.05
    mask_feature_length: int = 10
    mask_feature_prob: float = 0.0
    mask_time_length: int = 10
    mask_time_prob: float = 0.05


@dataclass
class W2V2Config:
    """HF Wav2Vec2Config subset (group-norm CNN, post-LN encoder = facebook/wav2vec2-base/-large)."""
    conv_dim: Tuple[int, ...] = (512,) * 7
    conv_kernel: Tuple[int, ...] = (10, 3, 3, 3, 3, 2, 2)
    conv_stride: Tuple[int, ...] = (5, 2, 2, 2, 2, 2, 2)
    hidden_size: int = 768
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    intermediate_size: int = 3072
    num_conv_pos_embeddings: int = 128
    num_conv_pos_embedding_groups: int = 16
    layer_norm_eps: float = 1e-5
    mask_time_min_masks: int = 2
    # The "-lv60" / xlsr family: pre-LN encoder (HF:611-654,729-802: LayerNorm BEFORE attention / FFN, one
    # LayerNorm after the last block), a LayerNorm over the channels after EVERY convolution instead of the layer-0
    # GroupNorm (HF:275-299), convolutions with bias.  The reference loads any HF id (ref: src/models/wav2vec2.py:25-55).
    do_stable_layer_norm: bool = False
    feat_extract_norm: str = "group"            # "group" | "layer"
    conv_bias: bool = False

    # checkpoint families by id substring.  From the published HF configs (NOT verifiable offline):
    # post-LN / group-norm: wav2vec2-base*, wav2vec2-large, wav2vec2-large-960h;
    # pre-LN / layer-norm convolutions with bias: *-lv60*, *xlsr*, *xls-r*, *-robust*, *voxpopuli* large models
    _STABLE_MARKS = ("lv60", "xlsr", "xls-r", "xls_r", "robust", "voxpopuli")
    _POST_LN_LARGE = ("wav2vec2-large", "wav2vec2-large-960h")

    @staticmethod
    def from_huggingface_id(hf_id: str) -> "W2V2Config":
        """The reference only inspects the substrings "base" / "large" for the WIDTH (ref: src/models/wav2vec2.py:112-117)
        and lets ``from_pretrained`` bring the rest of the architecture.  Offline there is no config.json to read, so the
        norm placement comes from the id: ids this table cannot classify RAISE instead of silently getting the post-LN
        geometry (pass an explicit W2V2Config for them)."""
        name = hf_id.rsplit("/", 1)[-1].lower()
        stable = any(m in name for m in W2V2Config._STABLE_MARKS)
        if "base" in hf_id:
            if stable:
                raise ValueError(f"{hf_id}: a base-width checkpoint of a pre-LN family is not in the table; pass a W2V2Config")
            return W2V2Config()
        if "large" in hf_id:
            large = dict(hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
            if stable:
                return W2V2Config(do_stable_layer_norm=True, feat_extract_norm="layer", conv_bias=True, **large)
            if name in W2V2Config._POST_LN_LARGE:
                return W2V2Config(**large)
            raise ValueError(f"{hf_id}: cannot tell whether this 'large' checkpoint is post-LN (wav2vec2-large, -large-960h) or "
                             "pre-LN (-lv60, xlsr, robust, voxpopuli); pass a W2V2Config with do_stable_layer_norm / "
                             "feat_extract_norm / conv_bias set")
        raise ValueError("cannot determine num features")

    @staticmethod
    def tiny() -> "W2V2Config":
        return W2V2Config(conv_dim=(32,) * 7, hidden_size=64, num_hidden_layers=2, num_attention_heads=4,
                          intermediate_size=128, num_conv_pos_embeddings=16, num_conv_pos_embedding_groups=4)

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    def conv_lengths(self, n_samples: int):
        out, length = [], n_samples
        for k, s in zip(self.conv_kernel, self.conv_stride):
            length = (length - k) // s + 1           # HF:997-1016
            out.append(length)
        return out

    def num_frames(self, n_samples: int) -> int:
        return self.conv_lengths(n_samples)[-1]

    # algorithmic FLOPs (2*MAC) per utterance -- BASELINE.md section 2 / SURVEY 8(d)
    def flops_per_utt(self, n_samples: int, head_classes: int = 5994, pooled: int = 2):
        L = self.conv_lengths(n_samples)
        T, H, I = L[-1], self.hidden_size, self.intermediate_size
        conv, cin = 0.0, 1
        for c, k, l in zip(self.conv_dim, self.conv_kernel, L):
            conv += 2.0 * l * c * cin * k
            cin = c
        proj = 2.0 * T * self.conv_dim[-1] * H
        pos = 2.0 * T * H * (H // self.num_conv_pos_embedding_groups) * self.num_conv_pos_embeddings
        layer = 2.0 * T * H * H * 4 + 2.0 * T * T * H * 2 + 2.0 * T * H * I * 2
        head = 2.0 * pooled * H * head_classes
        fwd = conv + proj + pos + layer * self.num_hidden_layers + head
        return {"conv": conv, "proj": proj, "pos": pos, "layer": layer, "head": head, "fwd": fwd,
                "train_frozen_cnn": fwd + 2.0 * (fwd - conv), "train_full": 3.0 * fwd}
