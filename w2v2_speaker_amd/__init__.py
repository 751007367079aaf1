"""w2v2_speaker_amd -- MI355X-native wav2vec2 speaker-recognition training path.

Hand-written gfx950 HIP kernels behind the C ABI of ``include/w2v2_hip.h`` (``libw2v2hip.so``), driven
by a static-plan engine (``engine.Plan``) over a flat parameter arena (``params.ParamStore``), with the
reference's module surface mirrored under ``models/``, ``layers/``, ``optim/`` and
``lightning_modules/``.  There is no CPU / eager fallback anywhere in this package.
"""
from .config import W2V2Config, Wav2Vec2RegularisationConfig  # noqa: F401

__version__ = "0.1.0"
