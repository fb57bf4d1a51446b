"""Thin tensor-level wrappers over the C ABI (include/mage_hip.h).

PyTorch is plumbing here: it owns device memory and the stream; every arithmetic
op below is one libmage_hip.so call on ``torch.cuda.current_stream()``.
All wrappers raise if a tensor is not on a CUDA (ROCm) device -- there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import ACT_GELU_ERF, ACT_NONE, ACT_QUICKGELU, ACT_QUICKGELU_GRAD, ACT_RELU, ACT_TANH, BF16, BF16X3, F16, F16X3, F32, AttnDesc, GemmDesc

__all__ = ["gemm", "layernorm", "attention", "embedding", "table_conv", "split_rows", "vq_prepare", "vq_nearest", "argmax", "cross_entropy",
           "conv_in", "conv_out", "convt_fold_tanh", "row_affine", "groupnorm_silu", "groupnorm_act", "reparam_kl", "mse", "check_device_errors", "graph_events_supported", "transpose", "row_sum", "sum_partials", "layernorm_bwd", "dropout_add_layernorm", "act", "act_bwd", "cross_entropy_bwd", "embedding_bwd", "group_rowsum", "attention_bwd", "dropout", "adam", "bn_train_stats", "bn_apply", "bn_backward", "convt_unfold_tanh_bwd", "maxpool2", "upsample2", "relu", "cast", "adain", "add_scaled_rowvec",
           "split", "split_empty", "split_dtype", "PROFILE", "F32", "BF16", "F16", "BF16X3", "F16X3", "ACT_NONE", "ACT_RELU", "ACT_QUICKGELU", "ACT_GELU_ERF", "ACT_TANH", "tdtype", "code"]


def code(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


def tdtype(c: int) -> torch.dtype:
    return {F32: torch.float32, BF16: torch.bfloat16, F16: torch.float16}[c]


# ---- split-precision tensors (MAGE_BF16X3 / MAGE_F16X3, include/mage_hip.h): a logical fp32 [rows, C] matrix kept as two 16-bit pieces
# per element, per row as 64-column slabs [hi(64) | lo(64)].  To torch it is a [rows, 2C] tensor of bfloat16 / float16 (plumbing:
# the pieces are only ever read by mage_gemm); the kind travels beside it.
def split_dtype(kind: int) -> torch.dtype:
    return torch.bfloat16 if kind == BF16X3 else torch.float16


def split_empty(rows: int, C_: int, kind: int, device, zero: bool = False) -> torch.Tensor:
    assert C_ % 64 == 0 and kind in (BF16X3, F16X3)
    f = torch.zeros if zero else torch.empty
    return f(rows, 2 * C_, device=device, dtype=split_dtype(kind))


def split(x: torch.Tensor, kind: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 rows [rows, C] (row stride x.stride(0)) -> split rows [rows, 2C] (mage_split)."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1
    rows, C_ = x.shape
    if out is None:
        out = split_empty(rows, C_, kind, x.device)
    assert out.dtype == split_dtype(kind) and out.is_contiguous() and out.shape[-1] == 2 * C_
    _lib.check(l.mage_split(x.data_ptr(), x.stride(0), out.data_ptr(), 2 * C_, rows, C_, kind, s), l)
    return out


def _dev(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("mage_amd ops need tensors on a ROCm GPU (cuda device); there is no CPU fallback")
    idx = t.device.index or 0
    if idx != torch.cuda.current_device():
        # the C ABI launches on the CURRENT device (zero page, launch attributes, the kernels themselves): a tensor of another
        # device would be addressed from the wrong GPU
        raise RuntimeError(f"mage_amd ops: tensor on cuda:{idx} but the current device is cuda:{torch.cuda.current_device()}; "
                           f"wrap the call in torch.cuda.device({idx})")
    return _lib.lib(idx), torch.cuda.current_stream(t.device).cuda_stream


_GRAPH_EVENTS = {}


def graph_events_supported(device) -> bool:
    """Can timing events be captured into a HIP graph as event-record nodes (torch.cuda.Event(external=True)) and read back
    after a replay?  Probed once per device with a two-node graph."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _GRAPH_EVENTS:
        ok = False
        import warnings
        try:
            with torch.cuda.device(idx), warnings.catch_warnings():
                warnings.simplefilter("ignore")          # a runtime without event-record nodes leaves an empty graph behind: expected, not news
                x = torch.zeros(1 << 20, device=dev)
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                a = torch.cuda.Event(enable_timing=True, external=True)
                b = torch.cuda.Event(enable_timing=True, external=True)
                with torch.cuda.graph(g):
                    a.record()
                    x.add_(1.0)
                    b.record()
                for _ in range(2):
                    g.replay()
                    torch.cuda.synchronize()
                    ms = a.elapsed_time(b)
                ok = 0.0 < ms < 100.0
        except Exception:
            ok = False
        _GRAPH_EVENTS[idx] = ok
    return _GRAPH_EVENTS[idx]


def check_device_errors(device) -> None:
    """Raise (ValueError) if a kernel met an out-of-range id / target since the last check; synchronises the current stream
    of `device` (include/mage_hip.h: mage_check_device_errors)."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with torch.cuda.device(idx):
        l = _lib.lib(idx)
        _lib.check(l.mage_check_device_errors(torch.cuda.current_stream(idx).cuda_stream), l)


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _lib_opts(l) -> dict:
    """The library's kernel-selection options that the profile keys mirror (mage_get_option; read only while profiling)."""
    out = {}
    v = C.c_int32(0)
    for name in ("gemm_no_narrow", "gemm_no_narrow_few", "gemm_no_8phase", "gemm_no_taps8", "gemm4_train_forms", "gemm_no_4w", "gemm_no_4h", "gemm_4h_plain",
                 "conv_no_tile"):
        _lib.check(l.mage_get_option(name.encode(), C.byref(v)), l)
        out[name] = int(v.value)
    return out


class _Profile:
    """Optional per-launch timing with HIP events recorded on the launch stream (torch.cuda.Event on the
    current stream IS a hipEvent on the stream the kernels are enqueued on).  Used by bench.py for the
    roofline of the dominant kernel; off by default (zero overhead).

    Eager launches: begin()/end() record a pair of events per instrumented launch.  Graph replay (MAGE.use_graph): the pairs
    are recorded ONCE, while the graph is captured, as `external` events (event-record nodes of the graph); every replay
    re-records them, and absorb() adds their elapsed times after the replay has been synchronised."""

    def __init__(self):
        self.enabled = False
        self.records = []
        self.only = None             # None: every instrumented launch; else the set of keys to time (others run un-bracketed)
        self.external = False        # True while a HIP graph is being captured
        self.acc = {}

    def reset(self, enabled: bool = False, only=None):
        self.enabled, self.records, self.only = enabled, [], (set(only) if only is not None else None)
        self.acc = {}

    def clear(self):
        """Drop what was measured so far, keep the mode (so that graphs captured for this mode stay valid)."""
        self.records, self.acc = [], {}

    def mode_key(self):
        """What a captured graph has to match: off / every launch / a set of keys."""
        if not self.enabled:
            return "off"
        return "all" if self.only is None else tuple(sorted(self.only))

    def wants(self, key: str) -> bool:
        return self.enabled and (self.only is None or key in self.only)

    def begin(self):
        e = torch.cuda.Event(enable_timing=True, external=True) if self.external else torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, key: str, start, flops: float = 0.0, nbytes: float = 0.0):
        e = torch.cuda.Event(enable_timing=True, external=True) if self.external else torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.append((key, start, e, flops, nbytes))

    def capture_begin(self):
        """Start collecting the event pairs of a graph capture; returns the state to hand back to capture_end."""
        saved = (self.records, self.external)
        self.records, self.external = [], True
        return saved

    def capture_end(self, saved):
        recs = self.records
        self.records, self.external = saved
        return recs

    def _add(self, out, recs):
        for key, s, e, fl, nb in recs:
            d = out.setdefault(key, {"ms": 0.0, "calls": 0, "flops": 0.0, "bytes": 0.0})
            d["ms"] += s.elapsed_time(e)
            d["calls"] += 1
            d["flops"] += fl
            d["bytes"] += nb

    def absorb(self, recs):
        """Add the elapsed times of a replayed graph's event pairs (the replay must have completed)."""
        if self.enabled and recs:
            self._add(self.acc, recs)

    def summary(self):
        torch.cuda.synchronize()
        out = {k: dict(v) for k, v in self.acc.items()}
        self._add(out, self.records)
        return out


PROFILE = _Profile()


def gemm(a: torch.Tensor, w: torch.Tensor, y: torch.Tensor, *, M: int, N: int, K: int, lda: int, ldy: int,
         out_h: int = 1, out_w: Optional[int] = None, in_h: Optional[int] = None, in_w: Optional[int] = None,
         a_img_stride: Optional[int] = None, a_off: int = 0, taps_h: int = 1, taps_w: int = 1, cin: Optional[int] = None,
         stride: int = 1, dy0: int = 0, dx0: int = 0, dys: int = 1, dxs: int = 1,
         y_img_stride: Optional[int] = None, y_mul_y: Optional[int] = None, y_mul_x: int = 1, y_off: int = 0,
         bias=None, scale=None, shift=None, act: int = ACT_NONE, rowadd=None, rowadd_div: int = 1, rowadd_mod: int = 1,
         residual=None, ldr: int = 0, post_relu: bool = False, ldw: int = 0, n_split: int = 1, a_split_stride: int = 0,
         w_split_stride: int = 0, y_split_stride: int = 0, y2=None, ldy2: int = 0, ln_part=None, ln_stats=None,
         ln_colsum=None, res_half: bool = False, a_half: bool = False, split_kind: int = 0, y_split: bool = False,
         ln_eps: float = 0.0, head_w=None, head_phases: int = 0, a_relu: bool = False) -> torch.Tensor:
    """Y = epilogue(A (*) W^T); see mage_gemm in include/mage_hip.h for the geometry fields.
    split_kind BF16X3 / F16X3: a and w are split-precision tensors (lda / ldw in 16-bit elements); y_split: so is y (ldy likewise).
    head_w (bf16 [16, N], padded-taps form with N == 256, bias, ReLU): y (fp32, ldy >= 16) receives the narrow Linear head_w on the
    bf16-rounded rows relu(acc + bias), which are not stored (mage_gemm_desc::head_w).
    a_relu (16-bit plain GEMM, N <= 128): the product over relu(a) (mage_gemm_desc::a_relu)."""
    l, s = _dev(a)
    if split_kind and a_relu:
        raise ValueError("ops.gemm: split-precision operands (split_kind) do not take a_relu")
    if split_kind:
        # the split-precision form takes a subset of the arguments (launch_spl / try_taps8 in csrc/gemm.hip): refuse the rest loudly
        # instead of dropping them
        unsupported = dict(scale=scale, shift=shift, y2=y2, ln_part=ln_part, ln_stats=ln_stats, ln_colsum=ln_colsum, head_w=head_w)
        bad = [k for k, v_ in unsupported.items() if v_ is not None]
        bad += [k for k, v_ in dict(post_relu=post_relu, res_half=res_half, a_half=a_half).items() if v_]
        bad += [k for k, v_ in dict(n_split=n_split).items() if v_ not in (0, 1)]
        bad += [k for k, v_ in dict(a_split_stride=a_split_stride, w_split_stride=w_split_stride, y_split_stride=y_split_stride, ldy2=ldy2,
                                    stride=stride - 1, dy0=dy0, dx0=dx0, dys=dys - 1, dxs=dxs - 1).items() if v_]
        if ln_eps:
            bad.append("ln_eps")
        if bad:
            raise ValueError(f"ops.gemm: split-precision operands (split_kind) do not take {', '.join(bad)}")
        if residual is not None and residual.dtype != torch.float32:
            raise ValueError(f"ops.gemm: the split-precision x + Linear(.) form adds an fp32 residual, got {residual.dtype}")
        return _gemm_split(l, s, a, w, y, M=M, N=N, K=K, lda=lda, ldy=ldy, out_h=out_h, out_w=out_w, in_h=in_h, in_w=in_w,
                           a_img_stride=a_img_stride, a_off=a_off, taps_h=taps_h, taps_w=taps_w, cin=cin, y_img_stride=y_img_stride,
                           y_mul_y=y_mul_y, y_mul_x=y_mul_x, y_off=y_off, bias=bias, act=act, rowadd=rowadd, rowadd_div=rowadd_div,
                           rowadd_mod=rowadd_mod, residual=residual, ldr=ldr, ldw=ldw, split_kind=split_kind, y_split=y_split)
    out_w = M if out_w is None else out_w
    in_h = out_h if in_h is None else in_h
    in_w = out_w if in_w is None else in_w
    d = GemmDesc()
    d.dtype, d.M, d.N, d.K = code(a), M, N, K
    assert w.dtype == a.dtype, (w.dtype, a.dtype)
    d.A, d.W, d.Y = a.data_ptr(), w.data_ptr(), y.data_ptr()
    d.lda, d.ldy, d.y_dtype = lda, ldy, code(y)
    d.out_h, d.out_w, d.in_h, d.in_w = out_h, out_w, in_h, in_w
    d.a_img_stride = in_h * in_w if a_img_stride is None else a_img_stride
    d.a_off = a_off
    d.taps_h, d.taps_w, d.cin, d.stride = taps_h, taps_w, (K // (taps_h * taps_w) if cin is None else cin), stride
    d.dy0, d.dx0, d.dys, d.dxs = dy0, dx0, dys, dxs
    d.y_img_stride = out_h * out_w if y_img_stride is None else y_img_stride
    d.y_mul_y = out_w if y_mul_y is None else y_mul_y
    d.y_mul_x, d.y_off = y_mul_x, y_off
    d.bias, d.scale, d.shift = _p(bias), _p(scale), _p(shift)
    d.act = act
    d.rowadd, d.rowadd_div, d.rowadd_mod = _p(rowadd), rowadd_div, rowadd_mod
    d.residual, d.ldr, d.res_dtype = _p(residual), ldr, (code(residual) if residual is not None else 0)
    d.post_relu = int(post_relu)
    d.ldw, d.n_split = ldw, n_split
    d.a_split_stride, d.w_split_stride, d.y_split_stride = a_split_stride, w_split_stride, y_split_stride
    d.y2, d.ldy2, d.ln_part, d.ln_stats, d.ln_colsum = _p(y2), ldy2, _p(ln_part), _p(ln_stats), _p(ln_colsum)
    d.ln_eps = float(ln_eps)
    if ln_part is not None:                                 # slice-major partial sums [n_slices, rows, 2] (mage_gemm_desc::ln_part_rows)
        assert ln_part.dtype == torch.float32 and ln_part.dim() == 3 and ln_part.shape[2] == 2 and ln_part.is_contiguous(), tuple(ln_part.shape)
        d.ln_part_rows = ln_part.shape[1]
    if head_w is not None:
        assert head_w.dtype == torch.bfloat16 and head_w.is_contiguous() and tuple(head_w.shape) == (16, N // (head_phases or 1)) \
            and y.dtype == torch.float32, (head_w.dtype, tuple(head_w.shape), y.dtype)
    d.head_w = _p(head_w)
    d.head_phases = head_phases if head_w is not None else 0
    d.res_half = int(res_half)
    d.a_half = int(a_half)
    d.a_relu = int(a_relu)
    ln = 2 if (ln_stats is not None or ln_colsum is not None) else (1 if (y2 is not None or ln_part is not None) else 0)      # LN_CONSUME / LN_PRODUCE
    h16 = d.dtype in (BF16, F16)                                                             # 16-bit operands: the same kernels, bf16 or f16 MFMA
    hf = "true" if d.dtype == F16 else "false"                                               # HF, the kernels' last template argument
    rb = residual is not None and residual.dtype == a.dtype and h16                          # 16-bit residual stream (RB in csrc/gemm_impl.h)
    if y2 is not None and ln_part is None and ln_stats is None and ln_colsum is None and act == ACT_QUICKGELU:
        ln = 3                                                                                  # LN_DUAL: pre-activation + activated rows
    act_k = act
    if act == ACT_QUICKGELU_GRAD:                                                               # y = acc * QuickGELU'(y2): LN_GELUBWD, act none
        ln, act_k = 4, ACT_NONE
    if PROFILE.enabled:
        # key = the kernel instantiation mage_gemm dispatches to (mirrors launch_act in csrc/gemm_impl.h, incl. the library's options), so
        # that the per-kernel averages line up with rocprofv3's per-symbol statistics
        lo = _lib_opts(l)
        gather = taps_h * taps_w > 1 or stride != 1 or dy0 != 0 or dx0 != 0 or d.in_h != d.out_h or d.in_w != d.out_w or a_half
        n_cu = torch.cuda.get_device_properties(a.device).multi_processor_count & ~7
        mt = 8 if (h16 and ((M + 255) // 256) * ((N + 255) // 256) * max(n_split, 1) >= 2 * n_cu) else 4
        if scale is None and rowadd is None and residual is None and not post_relu:
            ek = 0
        elif (not gather and act == ACT_NONE and residual is not None and (residual.dtype == torch.float32 or rb) and scale is None
              and rowadd is None and not post_relu and out_h == 1 and out_w >= M and residual.data_ptr() % 16 == 0 and not res_half):
            ek = 1
        else:
            ek = 2
        sp = "true" if n_split > 1 else "false"
        nw = 4
        if (ek != 1 and ln == 0 and N <= 128 and n_split == 1 and ((M + 255) // 256) * ((N + 63) // 64) >= n_cu
                and not lo["gemm_no_narrow"]):
            mt, nw = 2, 1                                   # the narrow 256 x 64 tile (launch_ek in csrc/gemm.hip)
        if (ek == 1 and h16 and not gather and act == ACT_NONE and n_split == 1 and N % 64 == 0
                and ((M + 127) // 128) * ((N + 255) // 256) < n_cu and not lo["gemm_no_narrow"]
                and not lo["gemm_no_narrow_few"]):
            mt, nw = 2, 1                                   # few rows: x + Linear(.) of the incremental loop on the narrow tile
        rbs = ", true" if (rb and ek == 1) else ", false"       # rocprofv3 prints every template argument: the keys match its symbols
        key = f"gemm_kernel<{d.dtype}, {'true' if gather else 'false'}, {act_k}, {mt}, {ek}, {sp}, {ln}, {nw}, 0{rbs}>"
        a_rows = ((M + out_h * out_w - 1) // (out_h * out_w)) * d.a_img_stride + a_off + 1
        if (h16 and not gather and mt == 8 and ek != 2 and K % 64 == 0 and a_rows * lda * 2 < 2 ** 32
                and N * K * 2 < 2 ** 32 and not lo["gemm_no_8phase"]):
            key = f"gemm8_kernel<{act_k}, {ek}, {sp}, false, {ln}, 0{rbs}, {hf}>"    # the 8-phase ping-pong variant (launch_tile in csrc/gemm_impl.h)
        # padded-taps convolutions and row-table Linears on the 8-phase kernel (try_taps8 in csrc/gemm.hip)
        ntaps = taps_h * taps_w
        table, plain = rowadd is not None and residual is None, rowadd is None and residual is None
        if (h16 and n_split == 1 and (ntaps > 1 or table) and stride == 1 and dys == 1 and dxs == 1 and dy0 == 0 and dx0 == 0
                and d.in_h >= out_h + taps_h - 1 and d.in_w >= out_w + taps_w - 1 and d.cin % 64 == 0 and K % 64 == 0 and scale is None
                and not post_relu and N % 256 == 0 and M % 256 == 0
                and not lo["gemm_no_8phase"] and not lo["gemm_no_taps8"]):
            if table and act == ACT_NONE:
                key = f"gemm8_kernel<0, 1, false, true, 0, 0, false, {hf}>"
            elif (plain or (head_w is not None and rowadd is None)) and act in (ACT_NONE, ACT_RELU) and d.dtype == BF16:
                key = f"gemm8_kernel<{act}, 0, false, true, {5 if head_w is not None else 0}, 0, false, false>"
            elif (rowadd is None and residual is not None and residual.dtype == torch.bfloat16 and y.dtype == torch.bfloat16 and act == ACT_NONE
                  and d.dtype == BF16 and bias is not None and y_mul_x == 1):
                key = "gemm8_kernel<0, 0, false, true, 0, 0, true, false>"       # the convolution adds a residual tensor in its epilogue
        # 64 -> 64 channel 3x3 convolutions over whole 16 x 16 pixel tiles (mage_conv3x3_c64_try in csrc/conv_tile.hip)
        tile_conv = (d.dtype == BF16 and y.dtype == torch.bfloat16 and N == 64 and d.cin == 64 and taps_h == 3 and taps_w == 3 and stride == 1 and dy0 == -1
                     and dx0 == -1 and dys == 1 and dxs == 1 and d.in_h == out_h and d.in_w == out_w and out_h % 16 == 0 and out_w % 16 == 0 and ek == 0
                     and act in (ACT_NONE, ACT_RELU) and y_mul_x == 1 and a_off == 0 and n_split == 1 and M % (out_h * out_w) == 0
                     and (M // 256) >= n_cu and not lo["conv_no_tile"])
        if tile_conv:
            key = "conv3x3_c64_kernel"
        # the one-wave-per-SIMD kernel (mage_gemm4_try in csrc/gemm4.hip): QKV / c_fc at full-loop sizes
        if (h16 and not gather and n_split <= 1 and M % 256 == 0 and N % 256 == 0 and K % 128 == 0 and 256 <= K <= 1024
                and out_h == 1 and out_w >= M and y_mul_x == 1 and ek == 0 and not res_half and ln_part is None
                and (bias is not None or (N <= 4096 and ln_stats is None)) and (ln_stats is None) == (ln_colsum is None)
                and (act in (ACT_NONE, ACT_QUICKGELU) if y2 is None else (ln in (3, 4) and y.dtype == torch.bfloat16 and ldy2 % 8 == 0
                                                                          and bool(lo["gemm4_train_forms"])))
                and lda % 8 == 0 and ldy % 8 == 0 and not lo["gemm_no_4w"]):
            nt4 = (M // 256) * (N // 256)
            if nt4 >= 4 * n_cu:
                key = f"gemm4_kernel<{act_k}, 0, {ln}, false, {hf}>"
            # its split-half form (mage_gemm4h_try in csrc/gemm4h.hip): K = 512, 16-bit rows out; at >= 4 tiles per CU the QuickGELU forms
            # (c_fc), from 3/4 tile per CU up to there (the incremental loop's step) every form
            if (K == 512 and y.dtype == a.dtype and y2 is None and ln in (0, 2) and not lo["gemm_no_4h"] and nt4 * 4 >= 3 * n_cu
                    and (nt4 < 4 * n_cu or act == ACT_QUICKGELU or lo["gemm_4h_plain"]) and (ln_stats is None or ln_stats.data_ptr() % 16 == 0)):
                key = f"gemm4h_kernel<{act_k}, {ln}, {hf}>"
        if PROFILE.wants(key):
            ev = PROFILE.begin()
            _lib.check(l.mage_gemm(C.byref(d), s), l)
            # algorithmic HBM bytes of this launch (SURVEY 8d's convention: every operand once): A + W + Y (+ fp32 residual, + the bf16
            # copy and the LayerNorm partial sums of the producer form)
            es, ys = a.element_size(), y.element_size()
            nb = float(M) * K * es + float(N) * K * es + float(M) * N * ys
            if key == "conv3x3_c64_kernel":                  # the input once (a quarter of it with a_half), not once per tap
                nb = float(M) * 64 * es * (0.25 if a_half else 1.0) + float(N) * K * es + float(M) * N * ys
            if residual is not None:
                nb += float(M) * N * residual.element_size()
            if y2 is not None:
                nb += float(M) * N * 2
            if ln_part is not None:
                nb += float(M) * (N // 64) * 8
            fl = 2.0 * M * N * K
            if head_w is not None:                           # the rows stay on the CU; 16 fp32 values per row (and phase) leave it
                nb += float(M) * 16 * 4 * (head_phases or 1) - float(M) * N * ys
                fl += 2.0 * M * N * 16
            PROFILE.end(key, ev, fl, nb)
            return y
    _lib.check(l.mage_gemm(C.byref(d), s), l)
    return y


def gemm_is_small(a: torch.Tensor, M: int, N: int, K: int) -> bool:
    """True if a bf16 plain GEMM of this size runs on the few-rows kernel on a's device (mage_gemm_is_small): its LayerNorm-consuming form
    then takes (mean, rstd) straight from the producer's partial sums (ln_part + ln_eps) and the mage_ln_stats launch can be skipped."""
    l, _ = _dev(a)
    r = l.mage_gemm_is_small(M, N, K)
    if r < 0:
        _lib.check(r, l)
    return r == 1


def _gemm_split(l, s, a, w, y, *, M, N, K, lda, ldy, out_h, out_w, in_h, in_w, a_img_stride, a_off, taps_h, taps_w, cin, y_img_stride, y_mul_y,
                y_mul_x, y_off, bias, act, rowadd, rowadd_div, rowadd_mod, residual, ldr, ldw, split_kind, y_split):
    """The split-precision form of mage_gemm (3 MFMA products per K slab, fp32-class result): plain rows or the padded-taps form."""
    sd = split_dtype(split_kind)
    assert a.dtype == sd and w.dtype == sd and (y.dtype == sd if y_split else y.dtype == torch.float32), (a.dtype, w.dtype, y.dtype)
    out_w = M if out_w is None else out_w
    in_h = out_h if in_h is None else in_h
    in_w = out_w if in_w is None else in_w
    d = GemmDesc()
    d.dtype, d.M, d.N, d.K = split_kind, M, N, K
    d.A, d.W, d.Y = a.data_ptr(), w.data_ptr(), y.data_ptr()
    d.lda, d.ldy, d.y_dtype = lda, ldy, (split_kind if y_split else F32)
    d.out_h, d.out_w, d.in_h, d.in_w = out_h, out_w, in_h, in_w
    d.a_img_stride = in_h * in_w if a_img_stride is None else a_img_stride
    d.a_off = a_off
    d.taps_h, d.taps_w, d.cin, d.stride = taps_h, taps_w, (K // (taps_h * taps_w) if cin is None else cin), 1
    d.dy0, d.dx0, d.dys, d.dxs = 0, 0, 1, 1
    d.y_img_stride = out_h * out_w if y_img_stride is None else y_img_stride
    d.y_mul_y = out_w if y_mul_y is None else y_mul_y
    d.y_mul_x, d.y_off = y_mul_x, y_off
    d.bias, d.act = _p(bias), act
    d.rowadd, d.rowadd_div, d.rowadd_mod = _p(rowadd), rowadd_div, rowadd_mod
    d.residual, d.ldr, d.res_dtype = _p(residual), ldr, F32
    d.ldw, d.n_split = ldw, 1
    if PROFILE.enabled:
        key = f"gemm_split<{split_kind}, {act}, {1 if residual is not None else 0}, {taps_h * taps_w}>"
        if PROFILE.wants(key):
            ev = PROFILE.begin()
            _lib.check(l.mage_gemm(C.byref(d), s), l)
            nb = 4.0 * M * K + 4.0 * N * K + float(M) * N * 4 * (2 if residual is not None else 1)
            PROFILE.end(key, ev, 2.0 * M * N * K, nb)
            return y
    _lib.check(l.mage_gemm(C.byref(d), s), l)
    return y


def row_stats(x: torch.Tensor, eps: float, stats: torch.Tensor) -> torch.Tensor:
    """(mean, rstd) per row of bf16 rows x [rows, C] (mage_row_stats)."""
    l, s = _dev(x)
    assert x.dtype in (torch.bfloat16, torch.float16) and x.dim() == 2 and x.stride(1) == 1 and stats.is_contiguous() and stats.numel() >= 2 * x.shape[0]
    ev = PROFILE.begin() if PROFILE.wants("layernorm") else None
    _lib.check(l.mage_row_stats(x.data_ptr(), code(x), x.shape[0], x.shape[1], x.stride(0), float(eps), stats.data_ptr(), s), l)
    if ev is not None:
        PROFILE.end("layernorm", ev, 0.0, float(x.numel()) * 2)
    return stats


def ln_stats(part: torch.Tensor, C_: int, eps: float, stats: torch.Tensor) -> torch.Tensor:
    """(mean, rstd) per row from a producer GEMM's partial sums ``part[n_slices, rows, 2]`` (slice-major; mage_ln_stats)."""
    l, s = _dev(part)
    n_slices, rows = part.shape[0], part.shape[1]
    assert part.dtype == torch.float32 and part.is_contiguous() and stats.is_contiguous() and stats.numel() >= 2 * rows
    if PROFILE.wants("ln_stats_kernel"):
        ev = PROFILE.begin()
        _lib.check(l.mage_ln_stats(part.data_ptr(), rows, n_slices, C_, eps, stats.data_ptr(), s), l)
        PROFILE.end("ln_stats_kernel", ev, 0.0)
        return stats
    _lib.check(l.mage_ln_stats(part.data_ptr(), rows, n_slices, C_, eps, stats.data_ptr(), s), l)
    return stats


def layernorm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, y: torch.Tensor, eps: float, split_kind: int = 0) -> torch.Tensor:
    """y = LayerNorm(x); split_kind BF16X3 / F16X3: y is a split-precision tensor [rows, 2C]."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
    Cc = x.shape[-1]
    assert not split_kind or (y.dtype == split_dtype(split_kind) and y.numel() == 2 * x.numel())
    ev = PROFILE.begin() if PROFILE.wants("layernorm") else None
    _lib.check(l.mage_layernorm(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), split_kind or code(y),
                                x.numel() // Cc, Cc, float(eps), s), l)
    if ev is not None:
        PROFILE.end("layernorm", ev, 0.0, float(x.numel()) * (4 + y.element_size()))
    return y


def attention(q, k, v, out, *, ldq, ldk, ldv, ldo, n_seq, inner, nq, nk, n_head, q_outer_stride, q_axis_stride,
              kv_outer_stride, kv_axis_stride, causal=False, kv_len=None, kv_len_div=1, scale=None, out_split: int = 0,
              drop_p: float = 0.0, drop_seed: int = 0, split_kind: int = 0, o_outer_stride: int = 0, o_axis_stride: int = 0):
    """out_split BF16X3 / F16X3 (fp32 q, k, v): out is a split-precision tensor, ldo in 16-bit elements (2 * logical width).
    split_kind F16X3: q, k, v are split-precision tensors too (ld* in 16-bit elements; views start at 2 * the logical column)."""
    l, s = _dev(q)
    d = AttnDesc()
    d.dtype = split_kind or code(q)
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, ldo
    d.n_seq, d.inner, d.nq, d.nk, d.n_head = n_seq, inner, nq, nk, n_head
    d.q_outer_stride, d.q_axis_stride = q_outer_stride, q_axis_stride
    d.kv_outer_stride, d.kv_axis_stride = kv_outer_stride, kv_axis_stride
    d.causal = int(causal)
    d.kv_len, d.kv_len_div = _p(kv_len), kv_len_div
    d.scale = float(32 ** -0.5 if scale is None else scale)
    d.out_split = out_split
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed) & (2 ** 64 - 1)       # dropout on the probabilities (fp32 kernels; mage_hip.h)
    d.o_outer_stride, d.o_axis_stride = o_outer_stride, o_axis_stride            # out's own row map (0, 0: q's)
    ev = PROFILE.begin() if PROFILE.wants("attention") else None
    _lib.check(l.mage_attention(C.byref(d), s), l)
    if ev is not None:
        es = q.element_size()
        PROFILE.end("attention", ev, 4.0 * n_seq * nq * nk * n_head * 32, float(n_seq) * n_head * 32 * es * (2 * nq + 2 * nk))
    return out


def embedding(ids: torch.Tensor, table: torch.Tensor, out: torch.Tensor, *, relu: bool = False, group: Optional[int] = None,
              group_stride: Optional[int] = None, off: int = 0, inner: int = 0, inner_stride: int = 0, split_kind: int = 0) -> torch.Tensor:
    l, s = _dev(table)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and table.dtype == torch.float32 and table.is_contiguous()
    n = ids.numel()
    group = n if group is None else group
    group_stride = group if group_stride is None else group_stride
    _lib.check(l.mage_embedding(ids.data_ptr(), table.data_ptr(), out.data_ptr(), split_kind or code(out), n, table.shape[1],
                                table.shape[0], int(relu), group, group_stride, off, inner, inner_stride, s), l)
    return out


def table_conv(ids: torch.Tensor, table: torch.Tensor, y: torch.Tensor, *, n_img: int, H: int, W: int, taps: int = 3, pos=None, bias=None,
               relu: bool = False, rowadd=None, split_kind: int = 0,
               rowadd_div: int = 1, rowadd_mod: int = 1, ldy: Optional[int] = None, group: Optional[int] = None,
               y_group_stride: Optional[int] = None, y_off: int = 0) -> torch.Tensor:
    """y rows = pos + sum of table[tap][ids[neighbour]] + rowadd: a k x k convolution of embedding rows as a table sum (mage_table_conv)."""
    l, s = _dev(table)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == n_img * H * W
    assert table.dim() == 3 and table.shape[0] == taps * taps and table.is_contiguous()
    Cc = table.shape[2]
    group = n_img * H * W if group is None else group
    y_group_stride = group if y_group_stride is None else y_group_stride
    ev = PROFILE.begin() if PROFILE.wants("table_conv") else None
    _lib.check(l.mage_table_conv(ids.data_ptr(), n_img, H, W, taps, taps, table.data_ptr(), code(table), table.shape[1], Cc, _p(pos), _p(bias),
                                 int(relu), _p(rowadd), rowadd_div, rowadd_mod, y.data_ptr(), split_kind or code(y), Cc if ldy is None else ldy, group,
                                 y_group_stride, y_off, s), l)
    if ev is not None:
        PROFILE.end("table_conv", ev, 0.0, float(n_img) * H * W * Cc * (taps * taps * table.element_size() + 4))
    return y


def resblock_table(ids: torch.Tensor, table: torch.Tensor, codebook: torch.Tensor, w1: torch.Tensor, y: torch.Tensor, *, n_img: int, H: int, W: int,
                   bias3: torch.Tensor, b1: torch.Tensor, scale1=None, shift1=None, post_relu: bool = False, ldy: int, y_img_stride: int,
                   y_row_pitch: int, y_off: int = 0) -> torch.Tensor:
    """The first ResBlock of the f4 VQ-VAE decoder on codebook rows in one launch (mage_resblock_table): y rows (bf16, the interior of a
    zero-padded frame buffer) = relu(x + BN(conv1x1(relu(table sum)))), x = relu(codebook[ids])."""
    l, s = _dev(table)
    Cc = table.shape[2]
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.numel() == n_img * H * W
    assert table.dtype == torch.bfloat16 and table.dim() == 3 and table.shape[0] == 9 and table.is_contiguous()
    assert codebook.dtype == torch.float32 and codebook.is_contiguous() and tuple(codebook.shape) == (table.shape[1], Cc)
    assert w1.dtype == torch.bfloat16 and w1.is_contiguous() and tuple(w1.shape) == (Cc, Cc) and y.dtype == torch.bfloat16
    ev = PROFILE.begin() if PROFILE.wants("resblock_table") else None
    _lib.check(l.mage_resblock_table(ids.data_ptr(), n_img, H, W, table.data_ptr(), table.shape[1], Cc, bias3.data_ptr(), codebook.data_ptr(),
                                     w1.data_ptr(), b1.data_ptr(), _p(scale1), _p(shift1), int(post_relu), y.data_ptr(), ldy, y_img_stride,
                                     y_row_pitch, y_off, s), l)
    if ev is not None:
        npix = float(n_img) * H * W
        PROFILE.end("resblock_table", ev, 2.0 * npix * Cc * Cc, npix * (8 + 2 * Cc))       # HBM: the ids in, the bf16 rows out
    return y


def resblock_rows(t: torch.Tensor, w1: torch.Tensor, residual: torch.Tensor, y: torch.Tensor, *, n_img: int, H: int, W: int, b1: torch.Tensor,
                  scale1=None, shift1=None, post_relu: bool = False, lda: int, ldr: int, ldy: int, img_stride: int, row_pitch: int,
                  off: int = 0) -> torch.Tensor:
    """y = relu(x + BN(conv1x1(t))) on bf16 rows, x and y in (padded) frame buffers (mage_resblock_rows): the tail of a ResBlock."""
    l, s = _dev(t)
    Cc = w1.shape[0]
    assert t.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and residual.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
    assert w1.is_contiguous() and tuple(w1.shape) == (Cc, Cc)
    ev = PROFILE.begin() if PROFILE.wants("resblock_rows") else None
    _lib.check(l.mage_resblock_rows(t.data_ptr(), lda, w1.data_ptr(), b1.data_ptr(), _p(scale1), _p(shift1), residual.data_ptr(), ldr,
                                    int(post_relu), y.data_ptr(), ldy, n_img, H, W, Cc, img_stride, row_pitch, off, s), l)
    if ev is not None:
        npix = float(n_img) * H * W
        PROFILE.end("resblock_rows", ev, 2.0 * npix * Cc * Cc, npix * 3 * 2 * Cc)
    return y


def vq_prepare(codebook: torch.Tensor):
    l, s = _dev(codebook)
    K, D = codebook.shape
    cbt = torch.empty(D, K, device=codebook.device, dtype=torch.float32)
    c2 = torch.empty(K, device=codebook.device, dtype=torch.float32)
    _lib.check(l.mage_vq_prepare(codebook.contiguous().data_ptr(), K, D, cbt.data_ptr(), c2.data_ptr(), s), l)
    return cbt, c2


def vq_nearest(z: torch.Tensor, cbt: torch.Tensor, c2: torch.Tensor, want_margin: bool = False):
    l, s = _dev(z)
    assert z.dtype == torch.float32 and z.is_contiguous()
    D, K = cbt.shape
    M = z.numel() // D
    idx = torch.empty(M, device=z.device, dtype=torch.int64)
    margin = torch.empty(M, device=z.device, dtype=torch.float32) if want_margin else None
    _lib.check(l.mage_vq_nearest(z.data_ptr(), cbt.data_ptr(), c2.data_ptr(), M, D, K, idx.data_ptr(), _p(margin), s), l)
    return (idx, margin) if want_margin else idx


def argmax(logits: torch.Tensor, out: torch.Tensor, *, rows: int, K: int, ld: Optional[int] = None, group: Optional[int] = None,
           in_group_stride: Optional[int] = None, in_off: int = 0, out_group_stride: Optional[int] = None, out_off: int = 0,
           margin=None) -> torch.Tensor:
    l, s = _dev(logits)
    assert logits.dtype == torch.float32 and out.dtype == torch.int64
    group = rows if group is None else group
    in_group_stride = group if in_group_stride is None else in_group_stride
    out_group_stride = group if out_group_stride is None else out_group_stride
    _lib.check(l.mage_argmax(logits.data_ptr(), rows, K, K if ld is None else ld, group, in_group_stride, in_off,
                             out.data_ptr(), out_group_stride, out_off, _p(margin), s), l)
    return out


def cross_entropy(logits: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    l, s = _dev(logits)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and target.dtype == torch.int64
    K = logits.shape[-1]
    rows = logits.numel() // K
    row_loss = torch.empty(rows, device=logits.device, dtype=torch.float32)
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    _lib.check(l.mage_cross_entropy(logits.data_ptr(), target.contiguous().data_ptr(), rows, K, row_loss.data_ptr(),
                                    loss.data_ptr(), s), l)
    return loss[0]


def conv_in(x, weight_t, bias, scale, shift, y, *, cin, H, W, cout, kh, kw, stride, pad, act=ACT_NONE, split_kind: int = 0, s2d: bool = False):
    """split_kind: y is a split-precision tensor; s2d: offset space-to-depth rows [N, OH/2+1, OW/2+1, 4*cout] (mage_hip.h)."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    _lib.check(l.mage_conv_in(x.data_ptr(), weight_t.data_ptr(), _p(bias), _p(scale), _p(shift), y.data_ptr(), split_kind or code(y),
                              x.shape[0], cin, H, W, cout, kh, kw, stride, pad, act, int(s2d), s), l)
    return y


def split_rows(x: torch.Tensor, y: torch.Tensor, kind: int, *, relu: bool = False, group: Optional[int] = None,
               group_stride: Optional[int] = None, off: int = 0, inner: int = 0, inner_stride: int = 0, relu_writeback: bool = False):
    """fp32 rows x [rows, C] -> split rows of y under mage_embedding's row map (+ ReLU; relu_writeback: x itself becomes relu(x))."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and y.dtype == split_dtype(kind)
    rows, C_ = x.shape
    group = rows if group is None else group
    group_stride = group if group_stride is None else group_stride
    _lib.check(l.mage_split_rows(x.data_ptr(), x.stride(0), y.data_ptr(), rows, C_, kind, int(relu), group, group_stride, off, inner, inner_stride,
                                 x.data_ptr() if relu_writeback else None, s), l)
    return y


def conv_out(x, weight_t, bias, y, *, N, IH, IW, cin, cout, transposed: bool):
    l, s = _dev(x)
    _lib.check(l.mage_conv_out(x.data_ptr(), code(x), weight_t.data_ptr(), _p(bias), y.data_ptr(), N, IH, IW, cin, cout,
                               int(transposed), s), l)
    return y



def convt_fold_tanh(taps, bias, y, *, N, IH, IW, cout):
    """ConvTranspose2d(., cout, 4, 2, 1) + tanh from per-input-pixel tap products [N*IH*IW, 16*cout] (see mage_hip.h)."""
    l, s = _dev(taps)
    assert taps.dtype == torch.float32 and taps.is_contiguous() and taps.numel() == N * IH * IW * 16 * cout
    _lib.check(l.mage_convt_fold_tanh(taps.data_ptr(), _p(bias), y.data_ptr(), N, IH, IW, cout, s), l)
    return y

def maxpool2(x, y, *, N, H, W, Cc, relu=False):
    l, s = _dev(x)
    _lib.check(l.mage_maxpool2(x.data_ptr(), y.data_ptr(), code(x), N, H, W, Cc, int(relu), s), l)
    return y


def upsample2(x, y, *, N, H, W, Cc):
    l, s = _dev(x)
    _lib.check(l.mage_upsample2(x.data_ptr(), y.data_ptr(), code(x), N, H, W, Cc, s), l)
    return y


def relu(x, y):
    l, s = _dev(x)
    _lib.check(l.mage_relu(x.data_ptr(), y.data_ptr(), code(x), x.numel(), s), l)
    return y


def cast(x, y):
    l, s = _dev(x)
    _lib.check(l.mage_cast(x.data_ptr(), code(x), y.data_ptr(), code(y), x.numel(), s), l)
    return y


def adain(x, gamma, beta, out, *, B, P, Cc, eps=1e-5):
    l, s = _dev(x)
    _lib.check(l.mage_adain(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), out.data_ptr(), B, P, Cc, float(eps), s), l)
    return out


def add_scaled_rowvec(x, svec, vec, *, B, P, Cc):
    l, s = _dev(x)
    _lib.check(l.mage_add_scaled_rowvec(x.data_ptr(), svec.data_ptr(), vec.data_ptr(), B, P, Cc, s), l)
    return x


def row_affine(x, rs=None, table=None, *, div: int = 1, mod: int = 1):
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    Cc = x.shape[-1]
    _lib.check(l.mage_row_affine(x.data_ptr(), _p(rs), _p(table), x.numel() // Cc, Cc, div, mod, s), l)
    return x


def caption_mask(ids: torch.Tensor, padding_idx: int):
    """(kv_len int32 [B], keep fp32 [B*S]) of int64 captions [B, S] (mage_caption_mask)."""
    l, s = _dev(ids)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and ids.dim() == 2
    B, S = ids.shape
    kv_len = torch.empty(B, device=ids.device, dtype=torch.int32)
    keep = torch.empty(B * S, device=ids.device, dtype=torch.float32)
    _lib.check(l.mage_caption_mask(ids.data_ptr(), B, S, int(padding_idx), kv_len.data_ptr(), keep.data_ptr(), s), l)
    return kv_len, keep


def assemble_video(first: torch.Tensor, video: torch.Tensor) -> torch.Tensor:
    """[B, L, C, H, W] = frame 0 of `first` ([B, >=1, C, H, W], any batch stride) followed by `video` [B, L-1, C, H, W] (mage_model.py:691),
    as two strided block copies on the stream (mage_copy2d)."""
    l, s = _dev(video)
    B, Lm1 = video.shape[0], video.shape[1]
    frame = video[0, 0].numel()
    assert first.dtype == video.dtype and video.is_contiguous() and first[0, 0].is_contiguous() and first[0, 0].numel() == frame
    out = torch.empty(B, Lm1 + 1, *video.shape[2:], device=video.device, dtype=video.dtype)
    es = video.element_size()
    pitch = (Lm1 + 1) * frame * es
    _lib.check(l.mage_copy2d(out.data_ptr(), pitch, first.data_ptr(), max(first.stride(0), frame) * es, frame * es, B, s), l)
    _lib.check(l.mage_copy2d(out.data_ptr() + frame * es, pitch, video.data_ptr(), Lm1 * frame * es, Lm1 * frame * es, B, s), l)
    return out


def groupnorm_silu(x, gamma, beta, y, *, n_samples, rows_per_sample, sample_stride_rows, row_off, groups, eps=1e-5):
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous()
    Cc = x.shape[-1]
    stats = torch.empty(n_samples, groups, 2, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_groupnorm_silu(x.data_ptr(), sample_stride_rows, row_off, n_samples, rows_per_sample, Cc, groups,
                                     gamma.data_ptr(), beta.data_ptr(), float(eps), stats.data_ptr(), y.data_ptr(), code(y), s), l)
    return y


def groupnorm_act(x, gamma, beta, y, *, n_samples, rows_per_sample, sample_stride_rows, row_off, groups, eps=1e-5, act=0,
                  residual=None, y_sample_stride_rows=None, y_row_off=0, stats=None):
    """y rows (b*y_sample_stride_rows + y_row_off + r) = act(GroupNorm(x rows (b*sample_stride_rows + row_off + r)) + residual);
    act 0 none / 1 ReLU / 2 SiLU (see mage_hip.h)."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and y.is_contiguous()
    assert residual is None or (residual.dtype == torch.float32 and residual.is_contiguous())
    Cc = x.shape[-1]
    if stats is None:                                   # (mean, rstd) per (sample, group): kept by the training path for the backward
        stats = torch.empty(n_samples, groups, 2, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_groupnorm_act(x.data_ptr(), sample_stride_rows, row_off, n_samples, rows_per_sample, Cc, groups,
                                    gamma.data_ptr(), beta.data_ptr(), float(eps), stats.data_ptr(), _p(residual), act, y.data_ptr(),
                                    code(y), rows_per_sample if y_sample_stride_rows is None else y_sample_stride_rows, y_row_off, s), l)
    return y


def reparam_kl(mu, logvar, eps, out, kl_sum):
    """out = eps * exp(0.5 logvar) + mu; kl_sum[b] = sum(1 + logvar - mu^2 - exp(logvar)) over sample b.  All [B, n] fp32."""
    l, s = _dev(mu)
    B = mu.shape[0]
    n = mu.numel() // B
    for t_ in (mu, logvar, eps, out):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == B * n
    _lib.check(l.mage_reparam_kl(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), out.data_ptr(), kl_sum.data_ptr(), B, n, s), l)
    return out


def groupnorm_bwd(x, gamma, beta, stats, dy, dx, *, n_samples, rows_per_sample, sample_stride_rows, row_off, groups, act=0, residual=None,
                  dy_sample_stride_rows=None, dy_row_off=0, want_dres=False):
    """Backward of groupnorm_act (same row maps): writes dx (x's row map; other rows untouched), returns (dgamma [C], dbeta [C], dres|None)."""
    l, s = _dev(x)
    Cc = x.shape[-1]
    for t_ in (x, dy, dx, stats):
        assert t_.dtype == torch.float32 and t_.is_contiguous()
    dev = x.device
    red = torch.empty(n_samples, groups, 2, device=dev, dtype=torch.float32)
    dgp = torch.empty(n_samples, Cc, device=dev, dtype=torch.float32)
    dbp = torch.empty(n_samples, Cc, device=dev, dtype=torch.float32)
    dres = torch.empty(n_samples * rows_per_sample, Cc, device=dev, dtype=torch.float32) if want_dres else None
    _lib.check(l.mage_groupnorm_bwd(x.data_ptr(), sample_stride_rows, row_off, n_samples, rows_per_sample, Cc, groups, stats.data_ptr(),
                                    gamma.data_ptr(), beta.data_ptr(), _p(residual), act, dy.data_ptr(),
                                    rows_per_sample if dy_sample_stride_rows is None else dy_sample_stride_rows, dy_row_off,
                                    red.data_ptr(), dx.data_ptr(), _p(dres), dgp.data_ptr(), dbp.data_ptr(), s), l)
    if n_samples == 1:
        return dgp[0], dbp[0], dres
    dg = sum_partials(dgp, torch.empty(Cc, device=dev, dtype=torch.float32), stride=Cc, n_part=n_samples, n=Cc)
    db = sum_partials(dbp, torch.empty(Cc, device=dev, dtype=torch.float32), stride=Cc, n_part=n_samples, n=Cc)
    return dg, db, dres


def adain_bwd(x, gamma_map, dout, *, B, P, Cc, eps=1e-5):
    """Backward of adain: (dx, dgamma_map); dbeta_map is dout itself."""
    l, s = _dev(x)
    dx, dg = torch.empty_like(x), torch.empty_like(x)
    _lib.check(l.mage_adain_bwd(x.data_ptr(), gamma_map.data_ptr(), dout.data_ptr(), dx.data_ptr(), dg.data_ptr(), B, P, Cc, float(eps), s), l)
    return dx, dg


def reparam_kl_bwd(mu, logvar, eps, dz, coef):
    """(dmu, dlogvar) of z = eps exp(logvar/2) + mu and the KL term; coef = 1-element fp32 device tensor dL/dkl / B."""
    l, s = _dev(mu)
    dmu, dlv = torch.empty_like(mu), torch.empty_like(mu)
    _lib.check(l.mage_reparam_kl_bwd(mu.data_ptr(), logvar.data_ptr(), eps.data_ptr(), dz.data_ptr(), coef.data_ptr(), dmu.data_ptr(),
                                     dlv.data_ptr(), mu.numel(), s), l)
    return dmu, dlv


def transpose_colsum(x, y, *, M: int, Mp: int, C: int, ldx: int, ldy: int, out_w: Optional[int] = None, img_stride: Optional[int] = None,
                     a_off: int = 0):
    """y[c, m] = x[arow(m), c] (bf16, zero for M <= m < Mp) and the column sums of the gathered rows in the same pass: returns
    db [C] fp32 (mage_transpose_colsum + mage_sum_partials)."""
    l, s = _dev(x)
    assert x.dtype == torch.bfloat16 and y.dtype == torch.bfloat16
    out_w = M if out_w is None else out_w
    n_part = ((Mp + 63) // 64 + 15) // 16
    part = torch.empty(n_part, C, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_transpose_colsum(x.data_ptr(), ldx, y.data_ptr(), ldy, M, Mp, C, out_w, out_w if img_stride is None else img_stride,
                                       a_off, part.data_ptr(), n_part, s), l)
    if n_part == 1:
        return part[0]
    return sum_partials(part, torch.empty(C, device=x.device, dtype=torch.float32), stride=C, n_part=n_part, n=C)


def maxpool2_bwd(x, dy, *, N, H, W, Cc):
    """dx [N*H*W, C] of y = maxpool2(x): dy [N*(H/2)*(W/2), C] routed to the first maximum of each window."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(x)
    _lib.check(l.mage_maxpool2_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), N, H, W, Cc, s), l)
    return dx


def upsample2_bwd(dy, *, N, H, W, Cc):
    """dx [N*H*W, C] of y = upsample2(x) (nearest): the sum of each 2x2 block of dy [N*2H*2W, C]."""
    l, s = _dev(dy)
    assert dy.dtype == torch.float32 and dy.is_contiguous()
    dx = torch.empty(N * H * W, Cc, device=dy.device, dtype=torch.float32)
    _lib.check(l.mage_upsample2_bwd(dy.data_ptr(), dx.data_ptr(), N, H, W, Cc, s), l)
    return dx


def mse_bwd(a, b, gout, *, rows, cols, lda, ldb):
    """d mse / da * gout as [rows, lda] fp32 (zeros in the padding columns)."""
    l, s = _dev(a)
    da = torch.empty(rows, lda, device=a.device, dtype=torch.float32)
    _lib.check(l.mage_mse_bwd(a.data_ptr(), lda, b.data_ptr(), ldb, rows, cols, gout.data_ptr(), da.data_ptr(), lda, s), l)
    return da


def mse(a, b, *, rows, cols, lda, ldb):
    """mean((a[:, :cols] - b[:, :cols])^2) over `rows` rows with row strides lda / ldb (fp32) -> 0-dim fp32 tensor."""
    l, s = _dev(a)
    assert a.dtype == torch.float32 and b.dtype == torch.float32
    out = torch.empty(1, device=a.device, dtype=torch.float32)
    ws = torch.empty(256, device=a.device, dtype=torch.float64)
    _lib.check(l.mage_mse(a.data_ptr(), lda, b.data_ptr(), ldb, rows, cols, ws.data_ptr(), out.data_ptr(), s), l)
    return out[0]


# ----------------------------------------------------------------------------------------------------------------- training path
def transpose(x, y, *, M: int, Mp: int, C: int, ldx: int, ldy: int, y_row0: int = 0, out_h: int = 1, out_w: Optional[int] = None,
              in_h: Optional[int] = None, in_w: Optional[int] = None, img_stride: Optional[int] = None, a_off: int = 0, dy: int = 0,
              dx: int = 0, stride: int = 1):
    """y[(c + y_row0), m] = x[arow(m), c] (zero for M <= m < Mp and outside the plane); see mage_transpose in mage_hip.h."""
    l, s = _dev(x)
    assert x.dtype == y.dtype
    out_w = M if out_w is None else out_w
    in_h = out_h if in_h is None else in_h
    in_w = out_w if in_w is None else in_w
    img_stride = in_h * in_w if img_stride is None else img_stride
    _lib.check(l.mage_transpose(x.data_ptr(), code(x), ldx, y.data_ptr(), ldy, y_row0, M, Mp, C, out_h, out_w, in_h, in_w, img_stride,
                                a_off, dy, dx, stride, s), l)
    return y


def gemm_tn(dy, x, *, T: int, N: int, K: int, ld_dy: int, ld_x: int, want_bias: bool = True):
    """dW [N, K] = dy[:T, :N]^T x[:T, :K] (fp32) and db [N] = column sums of dy, from ROW-MAJOR bf16 dy / x (no transposed copies):
    mage_gemm_tn over token slices + mage_sum_partials (+ mage_colsum)."""
    l, s = _dev(dy)
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and N % 256 == 0 and K % 256 == 0
    n_cu = torch.cuda.get_device_properties(dy.device).multi_processor_count & ~7
    tiles = (N // 256) * (K // 256)
    # ONE workgroup per CU, never a partial second round (measured at cfg2, TFLOP/s for in_proj / out_proj / c_fc / c_proj: one round
    # 596 / 631 / 734 / 719, two 574 / 535 / 689 / 675, four 502 / 418 / 648 / 629: long K loops amortise the ring start-up and the 256 KB
    # of fp32 partials each workgroup writes); >= 512 tokens per slice
    S = int(max(1, min(256, n_cu // tiles, (T + 511) // 512)))
    tps = ((T + S - 1) // S + 63) // 64 * 64
    S = (T + tps - 1) // tps
    part = torch.empty(S, N, K, device=dy.device, dtype=torch.float32)
    dbp = torch.empty(S, N, device=dy.device, dtype=torch.float32) if want_bias else None
    ev = PROFILE.begin() if PROFILE.wants("gemm_tn") else None
    _lib.check(l.mage_gemm_tn(dy.data_ptr(), ld_dy, x.data_ptr(), ld_x, T, N, K, S, tps, part.data_ptr(), _p(dbp), s), l)
    if ev is not None:
        PROFILE.end("gemm_tn", ev, 2.0 * T * N * K)
    dW = part[0] if S == 1 else sum_partials(part, torch.empty(N, K, device=dy.device, dtype=torch.float32), stride=N * K, n_part=S, n=N * K)
    db = None
    if want_bias:           # the kernel's per-slice column sums of dy (from the fragments it holds anyway), added in a fixed order
        db = dbp[0] if S == 1 else sum_partials(dbp, torch.empty(N, device=dy.device, dtype=torch.float32), stride=N, n_part=S, n=N)
    return dW, db


def colsum(x: torch.Tensor, *, T: int, C_: int, ld: int) -> torch.Tensor:
    """Column sums (fp32 [C]) of bf16 rows x[:T, :C] (mage_colsum + mage_sum_partials): a bias gradient on its own."""
    l, s = _dev(x)
    n_part = int(max(1, min(1024, T // 128)))
    cp = torch.empty(n_part, C_, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_colsum(x.data_ptr(), ld, T, C_, cp.data_ptr(), n_part, s), l)
    return cp[0] if n_part == 1 else sum_partials(cp, torch.empty(C_, device=x.device, dtype=torch.float32), stride=C_, n_part=n_part, n=C_)


def row_sum(x, out, *, ld: int, n: int, rows: int):
    l, s = _dev(x)
    n_chunk = int(max(1, min(64, n // 2048, 16384 // max(rows, 1))))      # enough waves to fill the chip, >= 2048 columns each
    if n_chunk == 1:
        _lib.check(l.mage_row_sum(x.data_ptr(), code(x), ld, n, rows, out.data_ptr(), 1, s), l)
        return out
    part = torch.empty(n_chunk, rows, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_row_sum(x.data_ptr(), code(x), ld, n, rows, part.data_ptr(), n_chunk, s), l)
    return sum_partials(part, out, stride=rows, n_part=n_chunk, n=rows)


def sum_partials(part, out, *, stride: int, n_part: int, n: int, accumulate: bool = False):
    l, s = _dev(part)
    assert part.dtype == torch.float32 and out.dtype == torch.float32
    _lib.check(l.mage_sum_partials(part.data_ptr(), stride, n_part, n, out.data_ptr(), int(accumulate), s), l)
    return out


def layernorm_bwd(x, gamma, dy, dx, *, eps: float, accumulate: bool, dx_bf16=None, p: float = 0.0, seed: int = 0):
    """dx (+)= dLN/dx; returns (dgamma, dbeta) fp32 [C].  dx_bf16: the updated dx once more as bf16 through dropout(., p, seed)'s mask."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and dx.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous()
    assert dx_bf16 is None or (dx_bf16.dtype == torch.bfloat16 and dx_bf16.is_contiguous() and dx_bf16.numel() == dx.numel())
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    n_part = int(min(1024, (rows + 3) // 4))
    part = torch.empty(n_part, 2, Cc, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_layernorm_bwd(x.data_ptr(), gamma.data_ptr(), dy.data_ptr(), code(dy), dx.data_ptr(), part.data_ptr(), n_part, rows,
                                    Cc, float(eps), int(accumulate), dx_bf16.data_ptr() if dx_bf16 is not None else None, float(p),
                                    int(seed) & (2 ** 64 - 1), s), l)
    gb = torch.empty(2, Cc, device=x.device, dtype=torch.float32)
    sum_partials(part, gb, stride=2 * Cc, n_part=n_part, n=2 * Cc)
    return gb[0], gb[1]


def act(x, y, kind: int):
    l, s = _dev(x)
    assert x.dtype == y.dtype and x.is_contiguous() and y.is_contiguous()
    _lib.check(l.mage_act(x.data_ptr(), y.data_ptr(), code(x), x.numel(), kind, s), l)
    return y


def act_bwd(x, dy, dx, kind: int):
    l, s = _dev(x)
    assert x.dtype == dy.dtype == dx.dtype and x.is_contiguous() and dy.is_contiguous() and dx.is_contiguous()
    _lib.check(l.mage_act_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), code(x), x.numel(), kind, s), l)
    return dx


def cross_entropy_bwd(logits, target, grad_out, dlogits):
    l, s = _dev(logits)
    assert logits.dtype == torch.float32 and logits.is_contiguous() and target.dtype == torch.int64 and grad_out.dtype == torch.float32
    K = logits.shape[-1]
    _lib.check(l.mage_cross_entropy_bwd(logits.data_ptr(), target.contiguous().data_ptr(), logits.numel() // K, K, grad_out.data_ptr(),
                                        dlogits.data_ptr(), code(dlogits), s), l)
    return dlogits


def embedding_bwd(ids, dout, dtable, *, padding_idx: int = -1, group: Optional[int] = None, group_stride: Optional[int] = None, off: int = 0):
    l, s = _dev(dout)
    assert ids.dtype == torch.int64 and ids.is_contiguous() and dtable.dtype == torch.float32 and dtable.is_contiguous()
    n = ids.numel()
    group = n if group is None else group
    group_stride = group if group_stride is None else group_stride
    scratch = None
    if dtable.shape[0] <= 512 and dtable.shape[1] % 64 == 0:
        # small table (the visual token table, the codebook, the caption vocabulary): per-chunk partial tables summed in chunk order, rows added
        # in ascending order inside a chunk -- no atomics between waves, a fixed-order fp32 sum: bit-identical gradients from run to run
        n_chunk = min(64, (n + 4095) // 4096)
        scratch = torch.empty(n_chunk * dtable.numel(), device=dtable.device, dtype=torch.float32)
    _lib.check(l.mage_embedding_bwd(ids.data_ptr(), dout.data_ptr(), code(dout), dtable.data_ptr(), n, dtable.shape[1], dtable.shape[0],
                                    padding_idx, group, group_stride, off, _p(scratch), 0 if scratch is None else scratch.numel(), s), l)
    return dtable


def group_rowsum(x, out, *, rows: int, C: int, div: int, mod: int, row_scale=None, row_scale_div: int = 1):
    l, s = _dev(x)
    assert out.dtype == torch.float32
    per_group = rows // max(mod, 1)
    n_chunk = int(max(1, min(256, per_group // 64, 2048 // max(mod, 1))))
    if n_chunk == 1:
        _lib.check(l.mage_group_rowsum(x.data_ptr(), code(x), rows, C, div, mod, _p(row_scale), row_scale_div, out.data_ptr(), 1, s), l)
        return out
    part = torch.empty(n_chunk, mod, C, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_group_rowsum(x.data_ptr(), code(x), rows, C, div, mod, _p(row_scale), row_scale_div, part.data_ptr(), n_chunk, s), l)
    return sum_partials(part, out, stride=mod * C, n_part=n_chunk, n=mod * C)


def attention_bwd(q, k, v, dout, dq, dk, dv, *, ldq, ldk, ldv, ldo, ld_dq, ld_dk, ld_dv, n_seq, inner, nq, nk, n_head, q_outer_stride,
                  q_axis_stride, kv_outer_stride, kv_axis_stride, causal=False, kv_len=None, kv_len_div=1, scale=None,
                  drop_p: float = 0.0, drop_seed: int = 0):
    l, s = _dev(q)
    d = AttnDesc()
    d.dtype = code(q)
    d.q, d.k, d.v, d.out = q.data_ptr(), k.data_ptr(), v.data_ptr(), dout.data_ptr()
    d.ldq, d.ldk, d.ldv, d.ldo = ldq, ldk, ldv, ldo
    d.n_seq, d.inner, d.nq, d.nk, d.n_head = n_seq, inner, nq, nk, n_head
    d.q_outer_stride, d.q_axis_stride = q_outer_stride, q_axis_stride
    d.kv_outer_stride, d.kv_axis_stride = kv_outer_stride, kv_axis_stride
    d.causal = int(causal)
    d.kv_len, d.kv_len_div = _p(kv_len), kv_len_div
    d.scale = float(32 ** -0.5 if scale is None else scale)
    d.drop_p, d.drop_seed = float(drop_p), int(drop_seed) & (2 ** 64 - 1)
    _lib.check(l.mage_attention_bwd(C.byref(d), dout.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), ld_dq, ld_dk, ld_dv, s), l)


def dropout(x, y, p: float, seed: int, accumulate: bool = False):
    l, s = _dev(x)
    assert x.is_contiguous() and y.is_contiguous() and x.numel() == y.numel()
    _lib.check(l.mage_dropout(x.data_ptr(), code(x), y.data_ptr(), code(y), x.numel(), float(p), int(seed) & (2 ** 64 - 1), int(accumulate), s), l)
    return y


def dropout_add(x, r, y, p: float, seed: int, y_bf16=None):
    """y = r + dropout(x) (the mask of dropout(x, ., p, seed)); r, y fp32; y_bf16: the same rows once more as bf16."""
    l, s = _dev(x)
    assert x.is_contiguous() and r.is_contiguous() and y.is_contiguous() and r.dtype == y.dtype == torch.float32 and x.numel() == y.numel() == r.numel()
    assert y_bf16 is None or (y_bf16.dtype == torch.bfloat16 and y_bf16.is_contiguous() and y_bf16.numel() == y.numel())
    _lib.check(l.mage_dropout_add(x.data_ptr(), code(x), r.data_ptr(), y.data_ptr(), y_bf16.data_ptr() if y_bf16 is not None else None,
                                  x.numel(), float(p), int(seed) & (2 ** 64 - 1), s), l)
    return y


def dropout_add_layernorm(x, r, y, gamma, beta, yn, eps: float, p: float, seed: int):
    """y = r + dropout(x) (fp32) and yn = LayerNorm(y) in one pass; returns (y, yn)."""
    l, s = _dev(x)
    Cc = r.shape[-1]
    assert x.is_contiguous() and r.is_contiguous() and y.is_contiguous() and yn.is_contiguous() and r.dtype == y.dtype == torch.float32
    assert x.numel() == y.numel() == r.numel() == yn.numel() and gamma.numel() == beta.numel() == Cc
    _lib.check(l.mage_dropout_add_layernorm(x.data_ptr(), code(x), r.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), yn.data_ptr(),
                                            code(yn), r.numel() // Cc, Cc, float(eps), float(p), int(seed) & (2 ** 64 - 1), s), l)
    return y, yn


def adam(p, g, m, v, *, lr: float, beta1: float, beta2: float, eps: float, step: int, grad_scale: float = 1.0):
    l, s = _dev(p)
    for t_ in (p, g, m, v):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == p.numel()
    _lib.check(l.mage_adam(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1), float(beta2),
                           float(eps), int(step), float(grad_scale), s), l)
    return p


def _bn_reduce(mode, x, dy, mask, mean, rstd, nout):
    l, s = _dev(x)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    n_part = int(min(1024, max(1, rows // 64)))
    part = torch.empty(n_part, nout, Cc, device=x.device, dtype=torch.float32)
    _lib.check(l.mage_bn_colreduce(mode, x.data_ptr(), _p(dy), _p(mask), _p(mean), _p(rstd), rows, Cc, part.data_ptr(), n_part, s), l)
    out = torch.empty(nout, Cc, device=x.device, dtype=torch.float32)
    return sum_partials(part, out, stride=nout * Cc, n_part=n_part, n=nout * Cc), rows


def bn_train_stats(x, eps: float):
    """Batch statistics of channels-last rows x [rows, C] fp32 (two passes): (mean [C], biased var [C], rstd [C])."""
    assert x.dtype == torch.float32 and x.is_contiguous()
    s, rows = _bn_reduce(0, x, None, None, None, None, 1)
    mean = (s[0] / rows).contiguous()
    q, _ = _bn_reduce(1, x, None, None, mean, None, 1)
    var = (q[0] / rows).contiguous()
    return mean, var, torch.rsqrt(var + eps).contiguous()        # [C]-sized vector bookkeeping


def bn_apply(x, mean, rstd, gamma, beta, y, relu: bool, residual=None):
    l, s = _dev(x)
    Cc = x.shape[-1]
    _lib.check(l.mage_bn_apply(x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(residual), y.data_ptr(),
                               code(y), x.numel() // Cc, Cc, int(relu), s), l)
    return y


def bn_backward(x, dy, mean, rstd, gamma, dx, mask=None):
    """dx (into `dx`), dgamma, dbeta of training-mode BatchNorm; mask = the post-ReLU output when a ReLU follows the norm."""
    l, s = _dev(x)
    assert x.dtype == torch.float32 and dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    Cc = x.shape[-1]
    sums, rows = _bn_reduce(2, x, dy, mask, mean, rstd, 2)
    _lib.check(l.mage_bn_bwd_apply(x.data_ptr(), dy.data_ptr(), _p(mask), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(), sums.data_ptr(),
                                   dx.data_ptr(), rows, Cc, s), l)
    return sums[1], sums[0]                                       # dgamma = sum g xhat, dbeta = sum g


def convt_unfold_tanh_bwd(grad_y, y, dtaps, *, N, IH, IW, cout):
    l, s = _dev(grad_y)
    assert grad_y.dtype == torch.float32 and grad_y.is_contiguous() and (y is None or (y.dtype == torch.float32 and y.is_contiguous()))
    _lib.check(l.mage_convt_unfold_tanh_bwd(grad_y.data_ptr(), _p(y), dtaps.data_ptr(), N, IH, IW, cout, s), l)
    return dtaps
