"""Log-mel filterbank front-end of the ECAPA / x-vector pipelines (ref: src/data/preprocess/audio_features.py:62-78
``FilterBank`` -> speechbrain 0.5.x ``Fbank(n_mels=40)``; config/data/pipeline/xvector_pipeline.yaml: selector ->
filterbank -> normalizer(channel-wise)).

Host-side data preprocessing, as in the reference (its DataLoader workers run it on the CPU); not part of the GPU
path.  speechbrain is not available here: this follows the published defaults of ``speechbrain.lobes.features.Fbank``
(STFT 25 ms Hamming window / 10 ms hop / n_fft 400, centred with constant padding; power spectrum; 40 triangular
filters equally spaced on the mel scale 2595 log10(1 + f/700) between 0 and 8000 Hz; 10 log10(max(., 1e-10)) clipped
to 80 dB below the maximum) -- PARITY UNPINNED."""
from __future__ import annotations

import math
from typing import List, Union

import torch

from .pipeline import SpeakerClassificationDataSample


class Fbank:
    def __init__(self, n_mels: int = 40, sample_rate: int = 16000, n_fft: int = 400, win_ms: float = 25.0,
                 hop_ms: float = 10.0, f_min: float = 0.0, f_max: float = 8000.0, amin: float = 1e-10,
                 top_db: float = 80.0):
        self.n_fft = n_fft
        self.win = int(round(sample_rate / 1000.0 * win_ms))
        self.hop = int(round(sample_rate / 1000.0 * hop_ms))
        self.window = torch.hamming_window(self.win)
        self.amin, self.top_db = amin, top_db
        n_stft = n_fft // 2 + 1
        to_mel = lambda hz: 2595.0 * math.log10(1.0 + hz / 700.0)
        mel = torch.linspace(to_mel(f_min), to_mel(f_max), n_mels + 2)
        hz = 700.0 * (10.0 ** (mel / 2595.0) - 1.0)
        band = hz[1:] - hz[:-1]                       # [n_mels + 1]
        f_central = hz[1:-1]                          # [n_mels]
        freqs = torch.linspace(0, sample_rate // 2, n_stft)
        # triangular filters: rising slope over band[m], falling slope over band[m+1]
        slope = (freqs[None, :] - f_central[:, None])
        left = slope / band[:-1, None] + 1.0
        right = -slope / band[1:, None] + 1.0
        self.fbank = torch.clamp(torch.minimum(left, right), min=0.0).t().contiguous()      # [n_stft, n_mels]

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        """[1, N] or [N] waveform -> [num_frames, n_mels]."""
        x = wav.reshape(-1).float()
        spec = torch.stft(x, self.n_fft, hop_length=self.hop, win_length=self.win, window=self.window, center=True,
                          pad_mode="constant", normalized=False, onesided=True, return_complex=True)
        power = spec.real ** 2 + spec.imag ** 2       # [n_stft, frames]
        mel = power.t() @ self.fbank                  # [frames, n_mels]
        db = 10.0 * torch.log10(torch.clamp(mel, min=self.amin))
        return torch.maximum(db, db.max() - self.top_db)


class FilterBank:
    """Preprocessor wrapper (ref: audio_features.py:62-78)."""

    def __init__(self, n_mels: int = 40):
        self.fb = Fbank(n_mels=n_mels)

    def process(self, sample: SpeakerClassificationDataSample
                ) -> Union[SpeakerClassificationDataSample, List[SpeakerClassificationDataSample]]:
        sample.network_input = self.fb(sample.network_input)
        return sample
