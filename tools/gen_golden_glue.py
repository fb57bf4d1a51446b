#!/usr/bin/env python
"""Golden vectors for the caller-side glue, produced by the REFERENCE's own dataload.py (build container only).

dataload.py imports lmdb / decord / nltk / pytorch_transformers / cv2 / skimage / torchvision, none of which exist here; none of
them takes part in the rules being pinned (caption tables, encode / decode, the `speed` frame sub-sampling, padding, collate), so
they are registered as empty stub modules in sys.modules BY THIS HARNESS (the reference is untouched), the LMDB / video readers are
replaced by in-memory stand-ins whose frame t is the constant t (so the sampled indices can be read off the output), and
nltk.word_tokenize -- only ever applied to captions that are already space-separated -- by str.split.

    python tools/gen_golden_glue.py      # writes tests/golden/glue_dataload.json
"""
import json
import os
import random
import sys
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def import_dataload():
    for name in ("lmdb", "decord", "nltk", "pytorch_transformers", "cv2", "skimage", "skimage.transform", "torchvision"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["decord"].VideoReader = object
    sys.modules["nltk"].download = lambda *a, **k: None
    sys.modules["nltk"].word_tokenize = lambda s: s.split()
    saved = list(sys.path)
    sys.path[:] = [REF] + [p for p in saved if os.path.abspath(p or os.getcwd()) != ROOT and p != REF]
    for name in list(sys.modules):
        if name == "utils" or name.startswith("utils."):
            del sys.modules[name]
    try:
        import dataload
    finally:
        sys.path[:] = saved
    assert dataload.__file__.startswith(REF)
    return dataload


def main():
    dl = import_dataload()
    out = {}
    T_RAW = 20

    class FakeLmdb:
        def __init__(self, path):
            self.caps = ["the digit 3 is moving up then down .", "the digit 0 is moving left then right and the digit 7 is moving up .",
                         "the digit 9 is moving down .", "the digit 5 is bouncing around here and there ."]

        def __len__(self):
            return len(self.caps)

        def __getitem__(self, i):
            return (np.arange(T_RAW, dtype=np.uint8)[:, None, None, None] * np.ones((T_RAW, 1, 2, 2), np.uint8)), self.caps[i]
    dl.LmdbReader = FakeLmdb
    for L, ss in ((16, [1.0, 2.0]), (8, [1.0, 3.0]), (24, [1.0, 1.5])):
        ds = dl.MovingMnistLMDB("x/", "train", frames_length=L, sample_speed=ss)
        if "mnist_vocab" not in out:
            out["mnist_vocab"] = ds.vocab
            out["mnist_padding_idx"] = int(ds.padding_idx)
        items, recs = [], []
        for i in range(len(ds)):
            random.seed(100 * L + i)
            it = ds[i]
            items.append(it)
            recs.append({"caption": ds.reader.caps[i], "rng_seed": 100 * L + i, "speed": float(it["speed"]), "text": it["text"].tolist(),
                         "decoded": ds.decode(it["text"].numpy()), "frame_values_x255": [int(round((v + 0.5) * 255)) for v in it["images"][:, 0, 0, 0].tolist()],
                         "images_shape": list(it["images"].shape), "images_dtype": str(it["images"].dtype)})
        b = ds.collate_fn(items)
        out[f"mnist_L{L}"] = {"frames_length": L, "sample_speed": ss, "raw_frames": T_RAW, "items": recs, "collate_keys": sorted(b.keys()),
                              "collate_text": b["text"].tolist(), "collate_images_shape": list(b["images"].shape),
                              "collate_speed_shape": list(b["speed"].shape)}

    # CATER: annotation json + a stand-in VideoReader (frame t = constant t), identity transform
    class FakeVid:
        def __init__(self, path):
            self.n = 301

        def __len__(self):
            return self.n

        def get_batch(self, idx):
            arr = np.array(idx, np.int64)[:, None, None, None] * np.ones((len(idx), 2, 2, 3), np.int64)
            return types.SimpleNamespace(asnumpy=lambda: arr)
    dl.VideoReader = FakeVid
    dl.Image = types.SimpleNamespace(fromarray=lambda a: a)
    caps = {"caterv1": ["the cone is sliding to ( 2 , -1 ) .", "the snitch is picked up and placed to the first quadrant ."],
            "caterv2": ["the large metal gold sphere is rotating while the small rubber red cube is sliding to ( 1 , 3 ) ."]}
    for dset, cc in caps.items():
        with tempfile.TemporaryDirectory() as td:
            json.dump({str(i): {"video": f"videos/v{i}.avi", "caption": c} for i, c in enumerate(cc)}, open(os.path.join(td, "train_explicit.json"), "w"))
            tf = lambda frames: torch.tensor(np.stack(frames), dtype=torch.float).permute(3, 0, 1, 2)     # [C, T, H, W] like ClipToTensor
            ds = dl.CATER(dset, td, "train", frames_length=32, sample_speed=[3.0, 6.0], image_transform=tf)
            out[dset + "_vocab"] = ds.vocab
            recs, items = [], []
            for i in range(len(ds)):
                random.seed(7 + i)
                it = ds[i]
                items.append(it)
                recs.append({"caption": cc[i], "rng_seed": 7 + i, "speed": float(it["speed"]), "text": it["text"].tolist(), "video_id": it["video_id"],
                             "decoded": ds.decode(it["text"].numpy()), "frame_index": [int(v) for v in it["images"][:, 0, 0, 0].tolist()],
                             "images_shape": list(it["images"].shape)})
            b = ds.collate_fn(items)
            out[dset] = {"frames_length": 32, "sample_speed": [3.0, 6.0], "raw_frames": 301, "min_interval": 3.0, "items": recs,
                         "collate_keys": sorted(b.keys()), "collate_text": b["text"].tolist()}
    path = os.path.join(os.environ.get("MAGE_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden"), "glue_dataload.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
