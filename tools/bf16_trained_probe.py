#!/usr/bin/env python
"""bf16 / f16 / f16x3 tokens on TRAINED weights against the CPU oracle (bench.py's `gpu_tokens_vs_oracle_trained_weights` leg, stand-alone
and with other steps / clips).  usage: bf16_trained_probe.py [train_steps=600] [clips=4] [threads=16] [style=strokes] [stage1_steps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 600
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 4
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 16
torch.cuda.set_device(0)
from mage_amd import _lib  # noqa: E402
_lib.load()
style = sys.argv[4] if len(sys.argv) > 4 else "strokes"
s1 = int(sys.argv[5]) if len(sys.argv) > 5 else None
print(json.dumps(bench.trained_token_agreement(torch.device("cuda:0"), 16, steps, 64, clips, threads, style=style, s1_steps=s1), indent=1))
