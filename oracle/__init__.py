"""CPU oracle for the wav2vec2 speaker-recognition hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported by the
product package ``w2v2_speaker_amd``; the only legal importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``, and
there only as the checker / the timed CPU baseline, never as the thing shipped.
"""
