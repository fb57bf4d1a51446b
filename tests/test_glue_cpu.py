"""Caller-side glue: batch contract producer, checkpoint format, gif writer."""
import os

import numpy as np
import torch

from mage_amd.utils import glue, synth
from mage_amd.utils.util import instantiate_from_config


def test_caption_tables_and_encoding():
    assert glue.MNIST_VOCAB["[PAD]"] == 0 and glue.MNIST_VOCAB["."] == 29 and len(glue.MNIST_VOCAB) == 30       # vocab_size 30
    assert glue.CATER_V1_VOCAB["quadrant"] == 29 and glue.CATER_V2_VOCAB["quadrant"] == 49 and len(glue.CATER_V2_VOCAB) == 50
    t = glue.encode_caption("the digit 3 is moving up then down .")
    assert t.dtype == torch.long and t.tolist() == [1, 13, 14, 6, 16, 19, 24, 28, 25, 29, 2]                     # 11 tokens: cfg1/cfg2 captions
    assert glue.decode_caption(t[1:-1]) == " the digit 3 is moving up then down ."
    two = glue.encode_caption("the digit 1 is moving left then right and the digit 7 is moving up .")
    assert len(two) in (16, 18, 20)                                                                              # the cfg3 lengths


def test_sample_clip_follows_the_reference_rule():
    raw = (np.arange(20)[:, None, None, None] * np.ones((20, 1, 4, 4))).astype(np.uint8)
    c = glue.sample_clip(raw, 16, [1.0, 2.0], speed=0.0)                      # interval 1: the first 16 frames
    assert c.shape == (16, 1, 4, 4) and torch.allclose(c[:, 0, 0, 0], torch.arange(16) / 255.0 - 0.5)
    c = glue.sample_clip(raw, 16, [1.0, 2.0], speed=1.0)                      # interval 2: 10 evenly spaced frames, padded with the last
    want = np.floor(np.linspace(0, 19, 10)).astype(np.int32)
    assert torch.allclose(c[:10, 0, 0, 0], torch.tensor(want / 255.0 - 0.5, dtype=torch.float))
    assert torch.equal(c[10:], c[9:10].repeat(6, 1, 1, 1))


def test_dataset_and_collate_produce_the_batch_contract():
    ds = glue.SyntheticMovingMnist(5, frames_length=8)
    items = [ds[i] for i in range(5)]
    b = ds.collate_fn(items)
    assert b["images"].shape == (5, 8, 1, 64, 64) and b["images"].dtype == torch.float32
    assert b["images"].min() >= -0.5 and b["images"].max() <= 0.5
    assert b["text"].dtype == torch.int64 and b["text"].shape[0] == 5 and (b["text"][:, 0] == 1).all()
    assert b["speed"].shape == (5,) and set(b.keys()) == {"images", "text", "speed"}
    ragged = glue.collate([{"images": items[0]["images"], "text": torch.tensor([1, 5, 2]), "speed": items[0]["speed"]},
                           {"images": items[1]["images"], "text": torch.tensor([1, 5, 6, 7, 2]), "speed": items[1]["speed"]}])
    assert ragged["text"].tolist() == [[1, 5, 2, 0, 0], [1, 5, 6, 7, 2]]


def test_checkpoint_round_trip_with_ddp_prefix(tmp_path):
    cfg = synth.mnist_model_config(frames_length=4, width=64, layers=2, vq_dim=32, K=32)
    m = instantiate_from_config(cfg)
    synth.fill_state_dict(m, 3)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-6)
    ck = glue.make_checkpoint(7, m, opt)
    assert set(ck) == {"epoch", "state_dict", "optimizer"}
    assert glue.save_checkpoint(ck, False, str(tmp_path / "run" / "iteration_10.pth")) is None and os.path.isdir(tmp_path / "run")
    path = glue.save_checkpoint(ck, True, str(tmp_path / "run" / "iteration_10.pth"))
    assert path.endswith("model_best.pth")
    m2 = instantiate_from_config(cfg)
    got = glue.load_checkpoint_into(m2, path, map_location="cpu")
    assert got["epoch"] == 7 and all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
    # a checkpoint written from a DistributedDataParallel-wrapped model
    ddp = {"epoch": 1, "state_dict": {"module." + k: v for k, v in m.state_dict().items()}, "optimizer": {}}
    m3 = instantiate_from_config(cfg)
    glue.load_checkpoint_into(m3, ddp)
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m3.state_dict().values()))


def test_gif_writer(tmp_path):
    from PIL import Image
    clip = torch.rand(6, 1, 64, 64) * 2 - 1
    out = glue.save_gifs(clip, "clip0-0.5000", str(tmp_path / "ckpt" / "model_best.pth"))
    assert out == str(tmp_path / "ckpt" / "videos" / "clip0-0.5000.gif")
    im = Image.open(out)
    assert im.n_frames == 6 and im.size == (64, 64) and im.info["duration"] in (330, 333)          # GIF delays are stored in 1/100 s
    rgb = glue.save_gifs(torch.rand(3, 3, 32, 32) * 2 - 1, "rgb", str(tmp_path / "ckpt" / "model_best.pth"))
    assert Image.open(rgb).n_frames == 3
