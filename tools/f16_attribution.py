"""Where the single-pass f16 mode's logit error comes from, and what each promotion would buy (VERDICT r5 "next round" item 2).

Runs on the GPU box.  The cfg2 model is trained in-tree (bench.train_strokes_model: the 'strokes' task, seeded), then

  A. EMULATION (tool-side torch ops in fp32, roundings to IEEE half inserted one class at a time): the decoder pass of
     /root/reference/modules/mage_model.py:374-390 teacher-forced on the reference-identical token sequences, max / rms |d logit| against the
     un-rounded pass -- which rounding of the f16 data flow carries the error: the 16-bit residual stream, the GEMM
     operands per class (QKV, out_proj, c_fc, c_proj, head), the 16-bit q|k|v, attention-output and hidden rows.
  B. THE HIP PATH itself in the variants that exist as switches: bf16, f16, f16 with the fp32 residual stream (config stream_16bit=False),
     f16 with fp32 stream and standalone LayerNorm (ln_fold=False), f16x3: teacher-forced error against the f16x3 logits (which sit 3e-5 from the
     CPU oracle's, tests + bench), free-running clips identical to the f16x3 sequences on 64 held-out clips, ms per cfg2 call.

The CPU oracle is run on the first `--oracle-clips` clips to tie the f16x3 reference to it.  Output: one JSON object (stdout and --out).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

import bench
from mage_amd import config
from mage_amd.utils import synth
from oracle import mage_oracle as O


def emu_decoder(sd, p, motion, imgs, r):
    """O.flat_axial_decoder with the f16 mode's roundings switched on by the flags in r (all off = the exact pass)."""
    def h(t, flag):
        return t.half().float() if r.get(flag) else t

    def ln_linear(x_pre, x_st, lnp, w, b, wflag, aflag):
        """Linear(LN(x)) as the folded form computes it: statistics of the fp32 sums x_pre, operand rows x_st (the stored stream), W' = gamma W."""
        g, bt = sd[lnp + ".weight"], sd[lnp + ".bias"]
        mean = x_pre.mean(-1, keepdim=True)
        rstd = torch.rsqrt(x_pre.var(-1, unbiased=False, keepdim=True) + 1e-5)
        wq = h(w * g[None, :], wflag)
        a = h(x_st, aflag)
        return rstd * (F.linear(a, wq) - mean * wq.sum(1)) + (F.linear(bt, w) + b)

    x = torch.cat([F.linear(motion, sd[p + "context_linear.weight"], sd[p + "context_linear.bias"]).unsqueeze(1),
                   F.linear(imgs, sd[p + "in_linear.weight"], sd[p + "in_linear.bias"])], 1)
    x_pre = x + sd[p + "T_positional_embedding"]
    x = h(x_pre, "stream")
    if r.get("stream"):
        x_pre = x                                                   # the fill's statistics are taken from the stored rows (mage_row_stats)
    i = 0
    n_blocks = 0
    while (p + f"blocks.{n_blocks}.ln_1.weight") in sd:
        n_blocks += 1
    for i in range(n_blocks):
        bp = p + f"blocks.{i}"
        axis, causal = i % 3 + 1, i % 3 == 0
        C = x.shape[-1]
        mv = lambda t: t.movedim(axis, -2)
        rows, rows_pre = mv(x), mv(x_pre)
        lead = rows.shape[:-2]
        rows, rows_pre = rows.reshape(-1, rows.shape[-2], C), rows_pre.reshape(-1, rows.shape[-2], C)
        A = rows.shape[1]
        nh, hd = C // 32, 32
        qkv = h(ln_linear(rows_pre, rows, bp + ".ln_1", sd[bp + ".attn.in_proj_weight"], sd[bp + ".attn.in_proj_bias"], "w_qkv", "a_qkv"), "o_qkv")
        q, k, v = qkv.split(C, dim=-1)
        R = q.shape[0]
        q = q.view(R, A, nh, hd).transpose(1, 2)
        k = k.view(R, A, nh, hd).transpose(1, 2)
        v = v.view(R, A, nh, hd).transpose(1, 2)
        s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
        if causal:
            s = s + torch.full((A, A), float("-inf"), device=s.device).triu_(1)
        o = h((torch.softmax(s, dim=-1) @ v).transpose(1, 2).reshape(R, A, C), "o_ao")
        pre1 = rows + F.linear(h(o, "a_out"), h(sd[bp + ".attn.out_proj.weight"], "w_out"), sd[bp + ".attn.out_proj.bias"])
        st1 = h(pre1, "stream")
        hdn = h(O.quick_gelu(ln_linear(pre1, st1, bp + ".ln_2", sd[bp + ".mlp.c_fc.weight"], sd[bp + ".mlp.c_fc.bias"], "w_cfc", "a_cfc")), "o_hdn")
        pre2 = st1 + F.linear(hdn, h(sd[bp + ".mlp.c_proj.weight"], "w_cproj"), sd[bp + ".mlp.c_proj.bias"])
        st2 = h(pre2, "head_in" if i == n_blocks - 1 else "stream")
        back = lambda t: t.view(*lead, A, C).movedim(-2, axis).contiguous()
        x, x_pre = back(st2), back(pre2)
    return F.linear(x[:, 1:], h(sd[p + "out.weight"], "w_head"), sd[p + "out.bias"])


EMU_FLAGS = ["stream", "head_in", "a_qkv", "a_cfc", "a_out", "w_qkv", "w_out", "w_cfc", "w_cproj", "w_head", "o_qkv", "o_ao", "o_hdn"]
# with a 16-bit stream the LayerNorm consumers read the stored rows as they are: a_qkv / a_cfc are roundings only when the stream is fp32
F16_MODE = {k: True for k in EMU_FLAGS}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=300)
    ap.add_argument("--clips", type=int, default=64)
    ap.add_argument("--oracle-clips", type=int, default=4)
    ap.add_argument("--tf-clips", type=int, default=16, help="clips of the teacher-forced comparisons (one decoder pass over them)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    L, B = 16, 64
    t0 = time.time()
    tm, sd, tinfo = bench.train_strokes_model(dev, L, a.train_steps, B)
    res = {"trained_weights_sha256": tinfo["trained_weights_sha256"], "train_seconds": tinfo["train_seconds"], "loss_trajectory": tinfo["loss_trajectory"]}
    held = synth.synth_batch_mnist(a.clips, L, seed=5000, style="strokes")
    hb = {k: v.to(dev) for k, v in held.items()}
    R = tm.image_resolution
    tm.ar_mode = "full"

    # ---- reference: f16x3 free-running tokens (and the CPU oracle's on the first clips)
    tm.set_precision("f16x3")
    tm.autoregressive_generate(hb)
    ref_tok = tm.last_tokens.clone()                                    # [clips, L-1, R, R]
    if a.oracle_clips > 0:
        sub = {k: v[:a.oracle_clips] for k, v in held.items()}
        with torch.no_grad():
            _, o_tok, _, o_trace = O.mage_generate(sd, sub, L, return_trace=True)
        res["f16x3_tokens_equal_cpu_oracle"] = {"clips": a.oracle_clips, "identical": bool((ref_tok[:a.oracle_clips].cpu() == o_tok).all())}
    n_tf = min(a.tf_clips, a.clips)
    tfb = {k: v[:n_tf] for k, v in hb.items()}

    def hip_logits(prec, **over):
        with config.override(**over):
            tm.set_precision(prec)
            dt_ = tm._dt()
            tok0 = tm.first_stage_encode(tfb["images"][:, 0:1])[:, 0].reshape(n_tf, -1)
            ctx = torch.cat([tok0[:, None, :], ref_tok[:n_tf].reshape(n_tf, L - 1, -1)[:, :L - 2]], 1).contiguous()
            ma = tm._motion_anchor(tok0.contiguous(), tfb, None)
            feats = tm._frame_source(ctx, dt_)
            lg = tm.generate_model._run(ma if dt_ == torch.float32 else ma.to(dt_), feats, B=n_tf, hh=R, ww=R)
            return lg.view(n_tf, L - 1, R, R, -1).float()

    ref_lg = hip_logits("f16x3")
    top2 = ref_lg.topk(2, dim=-1)[0]
    margin = (top2[..., 0] - top2[..., 1])
    res["reference_margin_quantiles"] = {q: float(torch.quantile(margin.flatten().double().cpu(), torch.tensor([v], dtype=torch.float64)))
                                         for q, v in (("0.1%", 0.001), ("1%", 0.01), ("10%", 0.1), ("50%", 0.5))}
    if a.oracle_clips > 0:
        res["f16x3_teacher_forced_max_err_vs_cpu_oracle"] = float((ref_lg[:a.oracle_clips].cpu() - o_trace.reshape(ref_lg[:a.oracle_clips].shape)).abs().max())

    def err_stats(lg):
        d = (lg - ref_lg).abs()
        flips = (lg.argmax(-1) != ref_lg.argmax(-1))
        return {"max": float(d.max()), "rms": float(d.pow(2).mean().sqrt()), "p99.9": float(torch.quantile(d.flatten()[::97].double().cpu(), 0.999)),
                "decisions_flipped": int(flips.sum()), "of": int(flips.numel())}

    # ---- A. emulation
    sdg = {k: v.to(dev) for k, v in sd.items() if k.startswith("generate_model.")}
    with torch.no_grad():
        # the decoder's inputs from the CPU oracle (convolutions stay off the tool-side GPU ops), the decoder pass itself as torch matmuls on the GPU
        cpu_b = {k: v[:n_tf] for k, v in held.items()}
        tok0 = O.vqvae_encode(sd, "first_stage_model.", cpu_b["images"][:, 0])
        ma = O.motion_anchor(sd, tok0, cpu_b["text"], cpu_b.get("speed")).to(dev)
        ctx = torch.cat([tok0[:, None], ref_tok[:n_tf, :L - 2].cpu()], 1)
        feats = O._frame_features(sd, ctx).to(dev)
        emu = lambda flags: emu_decoder(sdg, "generate_model.", ma, feats, flags)
        exact = emu({})
        res["emulation_exact_vs_f16x3_hip"] = err_stats(exact)
        table = {}
        for f in EMU_FLAGS:
            table["only_" + f] = {k: round(v, 6) if isinstance(v, float) else v for k, v in _vs(exact, emu({f: True})).items()}
        groups = {"all (the f16 mode)": dict(F16_MODE),
                  "all but the 16-bit stream (fp32 stream: stream + head_in off)": {k: True for k in EMU_FLAGS if k not in ("stream", "head_in")},
                  "stream + head_in only": {"stream": True, "head_in": True},
                  "operands only (a_*, w_*)": {k: True for k in EMU_FLAGS if k[:2] in ("a_", "w_")},
                  "weights only (w_*)": {k: True for k in EMU_FLAGS if k[:2] == "w_"},
                  "stored rows only (o_qkv, o_ao, o_hdn)": {"o_qkv": True, "o_ao": True, "o_hdn": True},
                  "all but o_hdn": {k: True for k in EMU_FLAGS if k != "o_hdn"},
                  "all but o_qkv": {k: True for k in EMU_FLAGS if k != "o_qkv"},
                  "all but o_ao": {k: True for k in EMU_FLAGS if k != "o_ao"},
                  "all but weights": {k: True for k in EMU_FLAGS if k[:2] != "w_"}}
        for name, flags in groups.items():
            table[name] = {k: round(v, 6) if isinstance(v, float) else v for k, v in _vs(exact, emu(flags)).items()}
        res["emulation_vs_exact_emulation"] = table
    del sdg
    torch.cuda.empty_cache()

    # ---- B. the HIP path's variants
    variants = [("bf16", "bf16", {}), ("f16", "f16", {}), ("f16 + fp32 residual stream", "f16", dict(stream_16bit=False)),
                ("f16 + fp32 stream + standalone LayerNorm", "f16", dict(stream_16bit=False, ln_fold=False)),
                ("bf16 + fp32 residual stream", "bf16", dict(stream_16bit=False)), ("f16x3", "f16x3", {})]
    hip = {}
    big = {k: v.to(dev) for k, v in synth.synth_batch_mnist(B, L, seed=7000, style="strokes").items()}
    for name, prec, over in variants:
        try:
            e = err_stats(hip_logits(prec, **over))
            with config.override(**over):
                tm.set_precision(prec)
                tm.autoregressive_generate(hb)
                eq = tm.last_tokens == ref_tok
                ident = int(eq.flatten(1).all(1).sum())
                tm.autoregressive_generate(big)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(3):
                    tm.autoregressive_generate(big)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t1) / 3 * 1e3
            hip[name] = {"teacher_forced_vs_f16x3": e, "clips_identical_to_f16x3": f"{ident} of {a.clips}", "token_agreement": round(float(eq.float().mean()), 5),
                         "ms_per_cfg2_call_full_loop": round(ms, 2), "frames_per_s": round(B * L / ms * 1e3, 1)}
        except Exception as ex:                                          # a variant a build refuses: say so, keep the table
            hip[name] = {"error": f"{type(ex).__name__}: {ex}"[:300]}
    res["hip_variants"] = hip
    res["seconds"] = round(time.time() - t0, 1)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        open(a.out, "w").write(txt + "\n")


def _vs(exact, lg):
    d = (lg - exact).abs()
    flips = lg.argmax(-1) != exact.argmax(-1)
    return {"max": float(d.max()), "rms": float(d.pow(2).mean().sqrt()), "decisions_flipped": int(flips.sum())}


if __name__ == "__main__":
    main()
