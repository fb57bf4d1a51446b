"""Caller-side glue of the hot path: what the reference's train / sample scripts do AROUND the model.

* the batch contract producer (dataload.py:183-271): caption -> token ids with the dataset's word table, the frame
  sub-sampling by ``speed``, the [-0.5, 0.5] normalisation, last-frame padding, and the collate that right-pads captions;
* the checkpoint format (main_mage.py:189-199, 210-228): ``{'epoch', 'state_dict', 'optimizer'}``, ``module.``-prefixed keys
  of DistributedDataParallel checkpoints stripped on load;
* the gif writer (main_mage.py:250-257).

No LMDB / decord readers here (out of scope): ``SyntheticMovingMnist`` feeds the same contract from the
synthetic clip generator, which is what the tests and the training smoke run use.  Pure host-side code: no arithmetic of the
model lives here.
"""
from __future__ import annotations

import os
import random
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import synth

__all__ = ["MNIST_VOCAB", "CATER_V1_VOCAB", "CATER_V2_VOCAB", "encode_caption", "decode_caption", "sample_indices", "sample_clip", "collate",
           "SyntheticMovingMnist", "make_checkpoint", "save_checkpoint", "load_checkpoint_into", "save_gifs"]

# word tables (data, dataload.py:199-203, 300-312): ids are part of the checkpoint contract (text_encoder.token_embedding rows)
_MNIST_WORDS = ["[PAD]", "[CLS]", "[SEP]", "0", "1", "2", "3", "4", "5", "6", "7", "8", "9", "the", "digit", "and", "is", "are",
                "bouncing", "moving", "here", "there", "around", "jumping", "up", "down", "left", "right", "then", "."]
_CATER1_WORDS = ["[PAD]", "[CLS]", "[SEP]", "the", "cone", "snitch", "is", "sliding", "picked", "placed", "containing", "rotating",
                 "and", "to", "up", "(", ")", "1", "2", "3", "-1", "-2", "-3", ",", ".", "first", "second", "third", "fourth", "quadrant"]
_CATER2_WORDS = ["[PAD]", "[CLS]", "[SEP]", "the", "cone", "snitch", "is", "sliding", "picked", "placed", "containing", "and", "to",
                 "up", "sphere", "cylinder", "cube", "small", "medium", "large", "metal", "rubber", "gold", "gray", "red", "blue",
                 "green", "brown", "purple", "cyan", "yellow", "(", ")", "1", "2", "3", "-1", "-2", "-3", ",", ".", "rotating", "while",
                 "contained", "still", "first", "second", "third", "fourth", "quadrant"]
MNIST_VOCAB = {w: i for i, w in enumerate(_MNIST_WORDS)}
CATER_V1_VOCAB = {w: i for i, w in enumerate(_CATER1_WORDS)}
CATER_V2_VOCAB = {w: i for i, w in enumerate(_CATER2_WORDS)}


def encode_caption(text: str, vocab: Dict[str, int] = MNIST_VOCAB) -> torch.Tensor:
    """'the digit 3 is moving up then down .' -> int64 [CLS] w1 .. wn [SEP] (dataload.py:216-224).  Unknown words raise KeyError."""
    return torch.tensor([vocab["[CLS]"]] + [vocab[w] for w in text.split()] + [vocab["[SEP]"]], dtype=torch.long)


def decode_caption(tokens, vocab: Dict[str, int] = MNIST_VOCAB) -> str:
    rev = {i: w for w, i in vocab.items()}
    return "".join(" " + rev[int(t)] for t in tokens)                      # leading space, like dataload.py:228-236


def sample_indices(n_raw: int, sample_speed: Sequence[float], speed: float, min_interval: float = 1.0) -> np.ndarray:
    """The `speed` frame sub-sampling rule (dataload.py:245-248 MovingMnist, min interval 1.0; :361-363 CATER, min interval 3.0): keep
    round(T / interval) evenly spaced frames with interval = max(min_interval, speed * (s_max - s_min) + s_min)."""
    interval = max(float(min_interval), speed * (sample_speed[-1] - sample_speed[0]) + sample_speed[0])
    return np.floor(np.linspace(0, n_raw - 1, round(n_raw / interval), endpoint=True)).astype(np.int32)


def sample_clip(images_raw: np.ndarray, frames_length: int, sample_speed: Sequence[float], speed: float, min_interval: float = 1.0) -> torch.Tensor:
    """uint8 [T, C, H, W] -> float32 [frames_length, C, H, W] in [-0.5, 0.5] (dataload.py:243-259): the sample_indices frames,
    truncated to frames_length, normalised, padded by repeating the last frame."""
    idx = sample_indices(images_raw.shape[0], sample_speed, speed, min_interval)
    clip = torch.tensor(images_raw[idx][:frames_length] / 255.0 - 0.5, dtype=torch.float)
    if clip.shape[0] < frames_length:
        clip = torch.cat([clip, clip[-1].unsqueeze(0).repeat(frames_length - clip.shape[0], 1, 1, 1)], 0)
    return clip


def collate(items: List[Dict[str, torch.Tensor]], padding_idx: int = 0) -> Dict[str, torch.Tensor]:
    """dataload.py:263-271: images stacked, captions right-padded with padding_idx, speeds stacked."""
    text = torch.nn.utils.rnn.pad_sequence([d["text"] for d in items], batch_first=True, padding_value=padding_idx)
    return {"images": torch.stack([d["images"] for d in items], 0), "text": text, "speed": torch.stack([d["speed"] for d in items], 0)}


_MOTIONS = ["up then down", "left then right", "down then up", "right then left"]        # data/mnist_caption_single.py:30


class SyntheticMovingMnist(torch.utils.data.Dataset):
    """The MovingMnistLMDB item contract (dataload.py:240-261) without an LMDB: item i is a deterministic synthetic clip
    (mage_amd.utils.synth sprites bouncing along one axis) with the caption 'the digit D is moving M .', sub-sampled by a random
    speed exactly as the reference does."""

    def __init__(self, n_items: int, frames_length: int, sample_speed: Sequence[float] = (1.0, 2.0), raw_frames: int = 20, seed: int = 0):
        self.n, self.frames_length, self.sample_speed, self.raw_frames, self.seed = n_items, frames_length, list(sample_speed), raw_frames, seed
        self.vocab, self.padding_idx = MNIST_VOCAB, MNIST_VOCAB["[PAD]"]

    def __len__(self):
        return self.n

    def raw(self, idx: int):
        g = synth.rng_for(self.seed, f"glue_item/{idx}")
        digit, motion = int(g.integers(0, 10)), int(g.integers(0, 4))
        clip = synth.synth_batch_mnist(1, self.raw_frames, seed=self.seed * 100003 + idx)["images"][0]          # [T,1,64,64] in [-0.5,0.5]
        raw = ((clip + 0.5) * 255.0).round().clamp(0, 255).to(torch.uint8).numpy()
        return raw, f"the digit {digit} is moving {_MOTIONS[motion]} ."

    def __getitem__(self, idx: int):
        raw, caption = self.raw(idx)
        speed = random.random()                                              # dataload.py:246
        return {"images": sample_clip(raw, self.frames_length, self.sample_speed, speed), "text": encode_caption(caption, self.vocab),
                "speed": torch.tensor(speed, dtype=torch.float)}

    def collate_fn(self, data):
        return collate(data, self.padding_idx)


# ----------------------------------------------------------------------------------------------------------------- checkpoints
def make_checkpoint(epoch: int, model: torch.nn.Module, optimizer) -> dict:
    """main_mage.py:189-193.  The reference calls this inside its rank-0 branch (:186).  With ``mage_amd.optim.FlatAdam`` on more than one
    rank the optimizer state is SHARDED: every rank must call ``optimizer.consolidate_state_dict()`` (a collective) before rank 0 builds
    the checkpoint here; ``optimizer.state_dict()`` itself is local and raises if that gather was skipped (it never hangs in RCCL)."""
    return {"epoch": epoch, "state_dict": model.state_dict(), "optimizer": optimizer.state_dict()}


def save_checkpoint(state: dict, is_best: bool, filename: str = "work_dirs/checkpoint.pth") -> Optional[str]:
    """main_mage.py:195-199, quirk included: only the best checkpoint is written, as <dir>/model_best.pth."""
    d = os.path.dirname(filename)
    if d and not os.path.isdir(d):
        os.makedirs(d)
    if is_best:
        path = os.path.join(d, "model_best.pth")
        torch.save(state, path)
        return path
    return None


def load_checkpoint_into(model: torch.nn.Module, checkpoint, map_location=None) -> dict:
    """main_mage.py:210-226: accepts a path or a loaded dict; strips the 'module.' prefix DistributedDataParallel adds."""
    ck = torch.load(checkpoint, map_location=map_location, weights_only=False) if isinstance(checkpoint, (str, os.PathLike)) else checkpoint
    sd = ck["state_dict"]
    if next(iter(sd)).startswith("module."):
        sd = OrderedDict((k[7:], v) for k, v in sd.items())
    model.load_state_dict(sd)
    return ck


# ----------------------------------------------------------------------------------------------------------------- gifs
def save_gifs(tgr: torch.Tensor, video_id: str, test_model: str, fps: int = 3) -> str:
    """main_mage.py:250-257: frames [L, C, H, W] in [-1, 1] -> <dir(test_model)>/videos/<video_id>.gif at 3 fps (PIL instead of
    imageio, which is not installed here)."""
    from PIL import Image
    imgs = ((tgr.detach().float().cpu() + 1) * 0.5 * 255.0).numpy().astype(np.uint8).transpose(0, 2, 3, 1)
    save_path = os.path.join(os.path.dirname(test_model), "videos")
    os.makedirs(save_path, exist_ok=True)
    frames = [Image.fromarray(f[..., 0] if f.shape[-1] == 1 else f) for f in imgs]
    out = os.path.join(save_path, video_id + ".gif")
    frames[0].save(out, save_all=True, append_images=frames[1:], duration=int(1000 / fps), loop=0)
    return out
