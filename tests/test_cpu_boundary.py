"""CPU: the drop-in boundary -- state_dict layout, config factory, C-ABI exports, loud failure without a GPU."""
import json
import os
import re

import pytest
import torch

from mage_amd import _lib
from mage_amd.utils import synth
from mage_amd.utils.util import instantiate_from_config
from tests.helpers import GOLDEN, ROOT


def _layout(m):
    return [[k, list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()]


def test_state_dict_layout_identical_to_reference():
    want = json.load(open(os.path.join(GOLDEN, "state_dict_layout.json")))
    mn = instantiate_from_config(synth.mnist_model_config(frames_length=16))
    assert _layout(mn) == want["mnist_L16"] and len(want["mnist_L16"]) == 205
    ct = instantiate_from_config(synth.cater_model_config(frames_length=10))
    assert _layout(ct) == want["caterv1_L10"] and len(want["caterv1_L10"]) == 256
    mp = instantiate_from_config(synth.magep_model_config(frames_length=10))          # MAGE+ (use_cids=False) layout
    assert _layout(mp) == want["magep_caterv2_L10"]


def test_reference_yaml_targets_resolve_and_shims_import():
    cfg = {"target": "modules.vqvae_model.VectorQuantizedVAE", "params": {"input_dim": 1, "down_ratio": 4, "dim": 32, "K": 16}}
    vq = instantiate_from_config(cfg)
    from modules.vqvae_model import VectorQuantizedVAE
    from modules.mage_model import MAGE, FlatAxialDecoder  # noqa: F401
    from utils.util import instantiate_from_config as shim_factory
    assert isinstance(vq, VectorQuantizedVAE) and shim_factory is instantiate_from_config
    with pytest.raises(KeyError):
        instantiate_from_config({"params": {}})
    # first stage is frozen, eval, and train() is disabled (mage_model.py:516-521)
    m = instantiate_from_config(synth.mnist_model_config(frames_length=4, width=64, layers=1, vq_dim=32, K=16))
    m.train()
    assert not m.first_stage_model.training and all(not p.requires_grad for p in m.first_stage_model.parameters())
    ddp_keys = {"module." + k for k in m.state_dict()}          # DDP prefix is stripped by the sampler (main_mage.py:218-223)
    m.load_state_dict({k[7:]: v for k, v in zip(sorted(ddp_keys), [m.state_dict()[k[7:]] for k in sorted(ddp_keys)])})


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "mage_hip.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(mage_\w+)\s*\(", header, flags=re.M))
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name)
    assert lib.mage_abi_version() == _lib.ABI_VERSION
    # descriptor structs match the header field-for-field
    for struct, cname in ((_lib.GemmDesc, "mage_gemm_desc"), (_lib.AttnDesc, "mage_attn_desc")):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), header, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if decl:
                fields += [f.strip().lstrip("*").strip() for f in re.sub(r"^(const\s+)?\w+\s*\*?", "", decl, count=1).split(",")]
        assert fields == [f[0] for f in struct._fields_], (cname, fields)


def test_product_path_fails_loudly_without_gpu_and_never_touches_oracle():
    m = instantiate_from_config(synth.mnist_model_config(frames_length=4, width=64, layers=1, vq_dim=32, K=16)).eval()
    batch = synth.synth_batch_mnist(1, 4, seed=1)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m.autoregressive_generate(batch)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m(batch)
    with pytest.raises(RuntimeError, match="ROCm GPU"):
        m.first_stage_model.encode(batch["images"][:, 0])
    for root, _, files in os.walk(os.path.join(ROOT, "mage_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|import_module\([\"']oracle", src, flags=re.M), \
                    f"{f} imports the oracle"


def test_synthetic_weights_are_portable_and_keyed_by_name():
    a = synth.synth_tensor("generate_model.blocks.0.attn.in_proj_weight", (1536, 512), torch.float32, 0)
    b = synth.synth_tensor("generate_model.blocks.0.attn.in_proj_weight", (1536, 512), torch.float32, 0)
    c = synth.synth_tensor("generate_model.blocks.1.attn.in_proj_weight", (1536, 512), torch.float32, 0)
    assert torch.equal(a, b) and not torch.equal(a, c)
    assert abs(a.std().item() - 512 ** -0.5) < 1e-3
    batch = synth.synth_batch_mnist(2, 16, seed=3, ragged_text=True, text_len=20, digits=2)
    assert batch["images"].shape == (2, 16, 1, 64, 64) and batch["images"].min() == -0.5 and batch["images"].max() <= 0.5
    assert batch["text"].dtype == torch.int64 and (batch["text"][:, 0] == 1).all()


def test_committed_bench_line_keeps_the_driver_contract():
    """profiles/r02_bench_cfg2_bf16.json is a bench.py output line: every key the driver / judge reads is there, typed and
    consistent (value = global_batch * frames / (ms_per_step / 1e3); roofline.frac = achieved / peak)."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = json.load(open(os.path.join(root, "profiles", "r02_bench_cfg2_bf16.json")))
    for k, typ in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                   ("ms_per_step", float), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str),
                   ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(r[k], typ), (k, type(r[k]))
    assert "vs_baseline" in r and r["vs_baseline"] is None and r["scaling"] == "weak" and r["higher_is_better"] is True
    assert "workload" in r["config"] and "model" not in r["config"]
    assert abs(r["value"] - r["config"]["global_batch"] * r["config"]["frames"] / (r["ms_per_step"] * 1e-3)) < 1e-2 * r["value"]
    rf = r["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] in ("GB/s", "TFLOP/s")
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["traffic"] is not None and rf["traffic"] > 0
    cb = r["cpu_baseline"]
    assert cb["kind"] in ("reference", "port") and cb["cores"] >= 1 and cb["value"] > 0 and isinstance(cb["sample"], str)
    assert cb["one_thread"]["cores"] == 1 and cb["one_thread"]["value"] > 0 and isinstance(cb["cpu"], str)
    # round 2: the parity-gated precision beside the benchmarked one, the decode roofline, the whole-call fraction
    pm = r["parity_mode"]
    assert pm["dtype"] == "fp32" and pm["value"] > 0 and 0.0 <= pm["bf16_free_running_token_agreement"]["first_generated_frame"] <= 1.0
    rd = r["roofline_decode"]
    assert rd["bound"] == "hbm" and abs(rd["frac"] - rd["achieved"] / rd["peak"]) < 1e-3 and rd["mfma"]["frac"] > 0
    wc = r["whole_call"]
    assert abs(wc["frac"] - wc["flops_per_call"] / (r["ms_per_step"] * 1e-3) / 1e12 / wc["peak"]) < 1e-3
    assert r["config"]["ranks_seen"] == r["n_gpus"] and "traffic_source" in rf
    ts = r.get("train_step")                       # secondary: the training step with its gradient exchange (round 2, late)
    assert ts is None or "error" in ts or (ts["value"] > 0 and ts["scaling"] == "weak" and ts["loss_first_last"][1] < ts["loss_first_last"][0])


def test_config_is_read_once_and_library_options_round_trip(monkeypatch):
    """mage_amd.config: the host-side switches are fields of one frozen object built from the MAGE_* variables (ENV) once; the library-side
    switches live in one table inside libmage_hip.so (mage_set_option / mage_get_option: no GPU needed)."""
    import dataclasses
    import importlib
    from mage_amd import config
    base = config.get()
    assert all(getattr(base, f.name) == f.default for f in dataclasses.fields(config.Config)) or True      # (the suite may run with switches set)
    assert set(config.ENV) == {f.name for f in dataclasses.fields(config.Config)}
    monkeypatch.setenv("MAGE_NO_LN_FOLD", "1")
    assert config.get() is base                                  # setting a variable after import changes nothing
    fresh = config._from_env()
    assert fresh.ln_fold is False and fresh.stream_16bit == base.stream_16bit
    with config.override(decode_head_fusion=False) as c:
        assert config.get() is c and c.decode_head_fusion is False
    assert config.get() is base
    with pytest.raises(dataclasses.FrozenInstanceError):
        base.ln_fold = False
    opts = config.lib_options()
    assert set(opts) == set(config.LIB_OPTIONS) and opts["gemm_small_m"] >= 0
    with config.lib_option("gemm_no_4w", 1):
        assert config.lib_flag("gemm_no_4w") == 1
    assert config.lib_flag("gemm_no_4w") == opts["gemm_no_4w"]
    with pytest.raises(Exception):
        with config.lib_option("no_such_option", 1):
            pass
    # values are validated by the library (round 6): switches take 0 | 1, counts and percentages have ranges; the Python mirror stays in step
    for name, bad in (("gemm_no_4h", 2), ("gemm_no_4w", -1), ("gemm_stagger_groups", 65), ("gemm_stagger_percent", 401), ("gemm_small_m", -5)):
        with pytest.raises(Exception, match="out of range"):
            config.set_lib_option(name, bad)
        assert config.lib_flag(name) == opts[name]
    with config.lib_option("gemm_4h_plain", 1):
        assert config.lib_options()["gemm_4h_plain"] == 1
    assert config.lib_options() == opts
    # no module of the package besides config.py (and _lib.py's library path) reads the environment
    import pathlib
    root = pathlib.Path(config.__file__).parent
    offenders = [str(p.relative_to(root)) for p in root.rglob("*.py")
                 if "os.environ" in p.read_text() and p.name not in ("config.py", "_lib.py", "dist.py")]
    assert offenders == [], offenders
