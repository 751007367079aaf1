"""w2v2_speaker_amd -- MI355X-native wav2vec2 speaker-recognition training path.

Hand-written gfx950 HIP kernels behind the C ABI of ``include/w2v2_hip.h`` (``libw2v2hip.so``), driven
by a static-plan engine (``engine.Plan``) over a flat parameter arena (``params.ParamStore``), with the
reference's module surface mirrored under ``models/``, ``layers/``, ``optim/`` and
``lightning_modules/``.  There is no CPU / eager fallback anywhere in this package.
"""
import os as _os

# Kernel arguments in device memory instead of host-coherent memory: the command processor fetches them faster, and a
# step is ~254 dependent launches of 10-100 us kernels -- measured -3.1 % step time (11.76 -> 11.39 ms, three ABAB rounds in
# one call).  The HIP runtime reads the flag when it initialises (first GPU call of the process), so importing this package
# before touching the GPU is enough; an explicit setting in the environment wins.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

from .config import W2V2Config, Wav2Vec2RegularisationConfig  # noqa: F401

__version__ = "0.1.0"
