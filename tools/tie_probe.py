import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from mage_amd.utils import synth
from tests.helpers import build_mage, cpu_sd
B, L = 1, 9
cfg = synth.cater_model_config(frames_length=L, width=64, layers=3, vq_dim=32, K=64)
m = build_mage(cfg, 41, "cuda:0")
batch = synth.synth_batch_cater(B, L, seed=41, text_len=9)
tok = m.first_stage_encode(batch["images"].to("cuda:0")).reshape(B, L, 16, 16).cpu()
sd = {k: v.double() for k, v in cpu_sd(m).items()}
v = sd["visual_token_embedding.weight"][tok].permute(0, 4, 1, 2, 3).contiguous()
for i in range(4):
    p = f"conv3d.{i}."
    t1 = F.group_norm(F.conv3d(v, sd[p + "conv1.weight"], None, stride=(2, 1, 1), padding=1), 16, sd[p + "bn1.weight"], sd[p + "bn1.bias"])
    o = F.conv3d(F.relu(t1), sd[p + "conv2.weight"], None, padding=1)
    o = F.group_norm(o, 16, sd[p + "bn2.weight"], sd[p + "bn2.bias"])
    r = F.group_norm(F.conv3d(v, sd[p + "downsample.0.weight"], None, stride=(2, 1, 1), padding=1), 16, sd[p + "downsample.1.weight"], sd[p + "downsample.1.bias"])
    t2 = o + r
    for nm, t in (("t1", t1), ("t2", t2)):
        a = t.abs()
        print(f"block {i} {nm}: min |t| {a.min().item():.3e}, count |t|<1e-5: {(a < 1e-5).sum().item()}, <1e-4: {(a < 1e-4).sum().item()} of {a.numel()}, scale {a.mean().item():.2f}")
    v = F.relu(t2)
