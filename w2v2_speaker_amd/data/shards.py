"""Shard files of the reference (ref: src/data/modules/speaker/voxceleb.py:1690-1962, written there with
``webdataset.TarWriter``): a tar (optionally .tar.gz) whose members come in pairs

    <key>.wav.pyd     pickle of a torch tensor [1, num_frames] f32   (webdataset "pyd" = pickle.dumps)
    <key>.meta.json   {"speaker_id", "youtube_id", "utterance_id", "speaker_id_idx", "num_frames", "sampling_rate"}

with key = "<speaker_id>/<youtube_id>/<utterance_id>" (ID_SEPARATOR "/"; webdataset takes everything before the first
"." of the member's base name as the key), and a folder-level ``meta.json`` with the statistics the data module
reads (``num_speakers``, ...).  webdataset itself is not available offline; the format only needs ``tarfile``."""
from __future__ import annotations

import glob
import io
import json
import os
import pickle
import tarfile
from typing import Dict, Iterable, Iterator, List, Sequence, Tuple

import torch


def _split_member(name: str) -> Tuple[str, str]:
    """webdataset key / extension split: directory part + base name up to its first '.'."""
    d, base = os.path.split(name)
    i = base.find(".")
    if i < 0:
        return name, ""
    return (d + "/" if d else "") + base[:i], base[i + 1:]


def iter_shard(path: str) -> Iterator[Dict]:
    """Yield {"__key__", "wav.pyd": tensor, "meta.json": dict} per sample, in tar order
    (what ``wds.WebDataset(...).decode()`` hands to ``_pipe_to_classification_data_sample``, ref :562-583)."""
    with tarfile.open(path, "r:*") as tar:
        cur_key, cur = None, {}
        for m in tar:
            if not m.isfile():
                continue
            key, ext = _split_member(m.name)
            if cur_key is not None and key != cur_key:
                if "wav.pyd" in cur and "meta.json" in cur:
                    yield cur
                cur = {}
            cur_key = key
            cur["__key__"] = key
            data = tar.extractfile(m).read()
            if ext == "wav.pyd":
                cur[ext] = pickle.loads(data)
            elif ext == "meta.json":
                cur[ext] = json.loads(data.decode("utf-8"))
        if cur_key is not None and "wav.pyd" in cur and "meta.json" in cur:
            yield cur


def find_shards(folder: str, pattern: str = "train_shard_*.tar*") -> List[str]:
    """ref: voxceleb.py ``_find_shard_paths``: sorted shard paths of a folder."""
    return sorted(glob.glob(os.path.join(folder, pattern)))


def read_meta(folder: str) -> Dict:
    with open(os.path.join(folder, "meta.json")) as f:
        return json.load(f)


def write_shards(samples: Iterable[Tuple[str, int, torch.Tensor]], folder: str, prefix: str = "train_shard",
                 samples_per_shard: int = 5000, sampling_rate: int = 16000, compress: bool = False) -> List[str]:
    """Write (key, speaker_id_idx, waveform [1, N] or [N]) samples in the reference's shard format and the folder
    ``meta.json`` (ref :1766-1795).  Used for synthetic data and tests; the reference's own writer converts the
    VoxCeleb m4a/wav tree."""
    os.makedirs(folder, exist_ok=True)
    paths: List[str] = []
    speakers, n_total, tar, count = set(), 0, None, 0

    def open_next():
        name = f"{prefix}_{len(paths):06d}.tar" + (".gz" if compress else "")
        p = os.path.join(folder, name)
        paths.append(p)
        return tarfile.open(p, "w:gz" if compress else "w")

    def add(tar, name: str, payload: bytes):
        info = tarfile.TarInfo(name)
        info.size = len(payload)
        tar.addfile(info, io.BytesIO(payload))

    for key, spk_idx, wav in samples:
        if tar is None or count == samples_per_shard:
            if tar is not None:
                tar.close()
            tar, count = open_next(), 0
        wav = torch.as_tensor(wav, dtype=torch.float32)
        if wav.dim() == 1:
            wav = wav[None]
        parts = key.split("/")
        meta = {"speaker_id": parts[0], "youtube_id": parts[1] if len(parts) > 1 else "",
                "utterance_id": parts[2] if len(parts) > 2 else "", "speaker_id_idx": int(spk_idx),
                "num_frames": int(wav.shape[-1]), "sampling_rate": sampling_rate}
        add(tar, key + ".wav.pyd", pickle.dumps(wav))
        add(tar, key + ".meta.json", json.dumps(meta).encode("utf-8"))
        speakers.add(int(spk_idx))
        count += 1
        n_total += 1
    if tar is not None:
        tar.close()
    with open(os.path.join(folder, "meta.json"), "w") as f:
        json.dump({"num_samples": n_total, "num_speakers": len(speakers), "num_shards": len(paths)}, f)
    return paths
