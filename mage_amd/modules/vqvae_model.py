"""MI355X-native VQ-VAE first stage (drop-in for the reference's modules/vqvae_model.py).

Same public surface -- ``VectorQuantizedVAE(input_dim, down_ratio, dim, K=512, ckpt_path=None,
ignore_keys=[])``, ``.encode(x) -> int64 [N,h,w]``, ``.decode(latents) -> f32 [N,C,H,W]``,
``.forward(x) -> (x_tilde, z_e_x, z_q_x)`` -- and the same state_dict keys/shapes
(vqvae_model.py:168-248), but nothing below runs a torch op on the data:
``nn.Conv2d`` / ``nn.BatchNorm2d`` objects are only parameter containers that give the reference's
key names; the arithmetic is libmage_hip.so (channels-last activations, BatchNorm(eval) + bias +
ReLU + residual fused into the implicit-GEMM epilogues, ConvTranspose as 4 sub-pixel GEMMs).

encode / decode are the inference entry points (this is how MAGE uses the model: frozen + eval, mage_model.py:516-521).
``forward`` in training mode (stage-1 training, train_vqvae.py:13-35: BatchNorm on batch statistics, the straight-through
quantiser vqvae_model.py:34-65, ``loss.backward()``) runs modules/vqvae_train.py on the same kernels (f4 stack).
"""
from __future__ import annotations


import contextlib
import itertools
import threading
from itertools import chain
from typing import Dict, List, Optional

import torch
from torch import nn

from .. import config, ops

__all__ = ["VectorQuantizedVAE", "VQEmbedding", "ResBlock", "EncoderBlock", "DecoderBlock", "weights_init"]


def weights_init(m: nn.Module) -> None:
    """xavier-uniform conv weights, zero bias (reference vqvae_model.py:77-84)."""
    if "Conv" in m.__class__.__name__ and hasattr(m, "weight"):
        nn.init.xavier_uniform_(m.weight.data)
        if getattr(m, "bias", None) is not None:
            m.bias.data.zero_()


def _no_torch_forward(self, *a, **k):
    raise RuntimeError(f"{type(self).__name__} is a parameter container; the arithmetic runs in libmage_hip.so "
                       "through VectorQuantizedVAE.encode/decode/forward")


class VQEmbedding(nn.Module):
    """Codebook holder: ``embedding.weight`` [K, D] ~ U(-1/K, 1/K) (vqvae_model.py:87-91)."""

    def __init__(self, K: int, D: int):
        super().__init__()
        self.embedding = nn.Embedding(K, D)
        self.embedding.weight.data.uniform_(-1.0 / K, 1.0 / K)

    forward = _no_torch_forward


class ResBlock(nn.Module):
    """Keys block.{1,2,4,5}.* (vqvae_model.py:111-124)."""

    def __init__(self, dim: int):
        super().__init__()
        self.block = nn.Sequential(nn.ReLU(True), nn.Conv2d(dim, dim, 3, 1, 1), nn.BatchNorm2d(dim), nn.ReLU(True),
                                   nn.Conv2d(dim, dim, 1), nn.BatchNorm2d(dim))

    forward = _no_torch_forward


def _bottleneck(dim_in: int, dim_out: int, first_k: int, last_k: int) -> nn.Sequential:
    hid = dim_out // 4
    ks = [first_k, 3, 3, last_k]
    chans = [dim_in, hid, hid, hid, dim_out]
    layers: List[nn.Module] = []
    for i, k in enumerate(ks):
        layers += [nn.ReLU(), nn.Conv2d(chans[i], chans[i + 1], k, 1, k // 2)]
    return nn.Sequential(*layers)


class EncoderBlock(nn.Module):
    """3x3,3x3,3x3,1x1 bottleneck + id path (vqvae_model.py:126-145)."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hid = dim_in, dim_out, dim_out // 4
        self.id_path = nn.Conv2d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()
        self.block = _bottleneck(dim_in, dim_out, 3, 1)

    forward = _no_torch_forward


class DecoderBlock(nn.Module):
    """1x1,3x3,3x3,3x3 bottleneck + id path (vqvae_model.py:147-166)."""

    def __init__(self, dim_in: int, dim_out: int):
        super().__init__()
        self.dim_in, self.dim_out, self.dim_hid = dim_in, dim_out, dim_out // 4
        self.id_path = nn.Conv2d(dim_in, dim_out, 1) if dim_in != dim_out else nn.Identity()
        self.block = _bottleneck(dim_in, dim_out, 1, 3)

    forward = _no_torch_forward


_WEIGHTS_EPOCH = 0


def bump_weights_epoch() -> None:
    """Tell every derived cache that parameter storage was written behind torch's back (a kernel updating the flat parameter
    arena: mage_amd.optim.FlatAdam.step): tensor version counters do not see that."""
    global _WEIGHTS_EPOCH
    _WEIGHTS_EPOCH += 1


# weights_frozen() regions are PER THREAD: a thread that has left its own region must not inherit another thread's promise (nn.DataParallel
# replicas, inference in one thread and an optimizer step in another); each region gets a process-unique token
_FROZEN = threading.local()
_FROZEN_TOKENS = itertools.count(1)


def _frozen_token() -> int:
    """0 outside a weights_frozen() region of THIS thread, else the region's token."""
    return getattr(_FROZEN, "token", 0) if getattr(_FROZEN, "depth", 0) else 0


@contextlib.contextmanager
def weights_frozen():
    """The caller promises that no parameter or buffer changes inside the block (one inference call: MAGE.autoregressive_generate).  Every
    derived cache then validates its signature ONCE per block instead of at every fetch -- the walk over all parameters and buffers
    (data_ptr, version, device: ~0.2 ms per fetch, ~40 fetches per incremental call) was half of the host's enqueue time of an
    incremental call, enough to make the call host-bound on a slow box.  The promise is the calling thread's; bump_weights_epoch()
    (FlatAdam.step) is honoured inside a region too: the fast path compares the epoch."""
    depth = getattr(_FROZEN, "depth", 0)
    if depth == 0:
        _FROZEN.token = next(_FROZEN_TOKENS)
    _FROZEN.depth = depth + 1
    try:
        yield
    finally:
        _FROZEN.depth -= 1


class _Derived:
    """Device-side derived caches (channels-last / transposed / bf16 weight copies, folded BN vectors).
    Rebuilt whenever a parameter or buffer is replaced or modified in place (load_state_dict, .to(), an optimizer step)."""

    def __init__(self, module: nn.Module):
        self._m = module
        self._sig = None
        self._store: Dict[str, torch.Tensor] = {}
        self._checked = (-1, -1)       # (weights_frozen() region token, weights epoch) this cache was last validated in
        self.gen = 0                   # bumped at every rebuild: captured HIP graphs that reference the old copies are stale

    def get(self, builder) -> Dict[str, torch.Tensor]:
        tok = _frozen_token()
        if tok and self._checked == (tok, _WEIGHTS_EPOCH):
            return self._store
        sig = (_WEIGHTS_EPOCH,) + tuple((t.data_ptr(), t._version, t.device) for t in chain(self._m.parameters(), self._m.buffers()))
        if sig != self._sig:
            with torch.no_grad():
                self._store = builder()
            self._sig = sig
            self.gen += 1
        if tok:
            self._checked = (tok, _WEIGHTS_EPOCH)
        return self._store


def _bn_vectors(bn: nn.BatchNorm2d):
    """eval BatchNorm as y = x*alpha + beta  (alpha = w/sqrt(var+eps), beta = b - mean*alpha)."""
    alpha = bn.weight.float() / torch.sqrt(bn.running_var.float() + bn.eps)
    return alpha.contiguous(), (bn.bias.float() - bn.running_mean.float() * alpha).contiguous()


def _conv_w(conv: nn.Conv2d) -> torch.Tensor:
    """[Cout, Cin, kh, kw] -> GEMM weight [Cout, kh*kw*Cin] (ci fastest)."""
    w = conv.weight.float()
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


class VectorQuantizedVAE(nn.Module):
    def __init__(self, input_dim, down_ratio, dim, K=512, ckpt_path=None, ignore_keys=[]):
        super().__init__()
        self.input_dim, self.down_ratio, self.dim, self.K = input_dim, down_ratio, dim, K
        if down_ratio == 4:                                           # vqvae_model.py:171-190
            self.encoder = nn.Sequential(nn.Conv2d(input_dim, dim, 4, 2, 1), nn.BatchNorm2d(dim), nn.ReLU(True),
                                         nn.Conv2d(dim, dim, 4, 2, 1), ResBlock(dim), ResBlock(dim))
            self.decoder = nn.Sequential(ResBlock(dim), ResBlock(dim), nn.ReLU(True), nn.ConvTranspose2d(dim, dim, 4, 2, 1),
                                         nn.BatchNorm2d(dim), nn.ReLU(True), nn.ConvTranspose2d(dim, input_dim, 4, 2, 1),
                                         nn.Tanh())
            self.codebook = VQEmbedding(K, dim)
        elif down_ratio == 8:                                         # vqvae_model.py:191-215
            self.encoder = nn.Sequential(nn.Conv2d(input_dim, dim, 7, padding=3), EncoderBlock(dim, dim), nn.MaxPool2d(2),
                                         EncoderBlock(dim, dim), nn.MaxPool2d(2), EncoderBlock(dim, 2 * dim),
                                         nn.MaxPool2d(2), EncoderBlock(2 * dim, 4 * dim), nn.ReLU())
            self.decoder = nn.Sequential(DecoderBlock(4 * dim, 2 * dim), nn.Upsample(scale_factor=2, mode="nearest"),
                                         DecoderBlock(2 * dim, dim), nn.Upsample(scale_factor=2, mode="nearest"),
                                         DecoderBlock(dim, dim), nn.Upsample(scale_factor=2, mode="nearest"),
                                         DecoderBlock(dim, dim), nn.ReLU(), nn.Conv2d(dim, input_dim, 1), nn.Tanh())
            self.codebook = VQEmbedding(K, 4 * dim)
        else:
            raise ValueError(f"down_ratio must be 4 or 8, got {down_ratio}")
        self.apply(weights_init)
        self.decode_dtype = torch.float32          # torch.bfloat16 = MFMA-bf16 performance mode for decode
        self.decode_split = 0                      # ops.F16X3: the f4 decode stack on split-precision operands (set_precision('f16x3'))
        self.encode_split = False                  # True: the f4 encoder's convolutions on f16x3 split operands (z_e within 7e-7 of the exact-fp32
                                                   # path, 2.2x faster); every precision except 'fp32', whose encoder stays the exact-fp32 MFMA chain
        self.decode_chunk = 1024                   # frames per decode launch group (bounds workspace: 0.8 GB at dim 256)
        self._pad_bufs = {}                        # zero-padded frame buffers of the bf16 decode, keyed by (frames, grid, device, stream)
        self._derived = _Derived(self)
        if ckpt_path is not None:
            self.init_from_ckpt(ckpt_path, ignore_keys=ignore_keys)

    # ------------------------------------------------------------------ checkpoint (vqvae_model.py:222-231)
    def init_from_ckpt(self, path, ignore_keys=list()):
        sd = torch.load(path, map_location="cpu")
        for k in list(sd.keys()):
            if any(k.startswith(ik) for ik in ignore_keys):
                print("Deleting key {} from state_dict.".format(k))
                del sd[k]
        self.load_state_dict(sd, strict=False)
        print(f"Restored from {path}")

    def set_precision(self, precision: str) -> "VectorQuantizedVAE":
        """'fp32' (parity: exact-fp32 MFMA), 'bf16' (decode stack on bf16 MFMA; encode + VQ stay fp32-class so token indices stay
        bit-exact), or 'f16x3' / 'bf16x3' (the fast parity modes: the f4 decode stack on split-precision operands, fp32-class frames)."""
        # 'f16' (the decoder stack's single-pass half mode): frames come from the bf16 decode stack -- tokens -> pixels is not on the token
        # path, and the fused decode kernels (resblock_table, the sub-pixel head) are bf16
        self.decode_dtype = {"fp32": torch.float32, "bf16": torch.bfloat16, "f16": torch.bfloat16, "f16x3": torch.float32,
                             "bf16x3": torch.float32}[precision]
        self.decode_split = ops.F16X3 if precision in ("f16x3", "bf16x3") else 0     # f16 pieces for both: the encoder's kind
        # 'fp32' is the exact parity mode end to end: its encoder keeps the exact-fp32 MFMA chain, so that a near-tie token cannot flip
        # against the reference because of the 7e-7 the split operands move z_e by; the other modes take the split encoder
        self.encode_split = precision != "fp32"
        return self

    # ------------------------------------------------------------------ derived weights
    def _build(self) -> Dict[str, torch.Tensor]:
        d: Dict[str, torch.Tensor] = {}
        cb = self.codebook.embedding.weight.float().contiguous()
        d["cb"] = cb
        d["cbt"], d["c2"] = ops.vq_prepare(cb)

        def both(name, w):
            d[name + ".f32"] = w
            d[name + ".bf16"] = w.to(torch.bfloat16)

        def res(prefix, rb: ResBlock):
            both(prefix + ".w3", _conv_w(rb.block[1]))
            d[prefix + ".b3"] = rb.block[1].bias.float().contiguous()
            d[prefix + ".s3"], d[prefix + ".t3"] = _bn_vectors(rb.block[2])
            both(prefix + ".w1", _conv_w(rb.block[4]))
            d[prefix + ".b1"] = rb.block[4].bias.float().contiguous()
            d[prefix + ".s1"], d[prefix + ".t1"] = _bn_vectors(rb.block[5])

        def bott(prefix, blk):
            if isinstance(blk.id_path, nn.Conv2d):
                both(prefix + ".wid", _conv_w(blk.id_path))
                d[prefix + ".bid"] = blk.id_path.bias.float().contiguous()
            for j in (1, 3, 5, 7):
                both(f"{prefix}.w{j}", _conv_w(blk.block[j]))
                d[f"{prefix}.b{j}"] = blk.block[j].bias.float().contiguous()

        enc, dec = self.encoder, self.decoder
        if self.down_ratio == 4:
            d["e0.wt"] = enc[0].weight.float().permute(1, 2, 3, 0).contiguous()       # [cin,kh,kw,cout]
            d["e0.b"] = enc[0].bias.float().contiguous()
            d["e0.s"], d["e0.t"] = _bn_vectors(enc[1])
            both("e3.w", _conv_w(enc[3]))
            d["e3.b"] = enc[3].bias.float().contiguous()
            res("e4", enc[4]); res("e5", enc[5]); res("d0", dec[0]); res("d1", dec[1])
            # ConvTranspose2d(dim, dim, 4, 2, 1) as 4 sub-pixel 2x2 convolutions: output pixel (2*oy+py, 2*ox+px) reads
            # input (oy+dy, ox+dx) through kernel tap ky = py + 1 - 2*dy (dy = -ky2 for py = 0, 1 - ky2 for py = 1).
            wt = dec[3].weight.float()                                                  # [cin, cout, 4, 4]
            for py in range(2):
                for px in range(2):
                    kys = [py + 1 - 2 * (py - k2) for k2 in range(2)]
                    kxs = [px + 1 - 2 * (px - k2) for k2 in range(2)]
                    sub = wt[:, :, kys][:, :, :, kxs]                                   # [cin, cout, 2, 2]
                    both(f"d3.w{py}{px}", sub.permute(1, 2, 3, 0).reshape(wt.shape[1], -1).contiguous())
            d["d3.b"] = dec[3].bias.float().contiguous()
            d["d3.s"], d["d3.t"] = _bn_vectors(dec[4])
            # bf16 decode on the 8-phase kernel's padded-taps form (_decode_chunk): eval BatchNorm folded into the weights and the
            # bias (W' = alpha W per output channel, b' = alpha b + beta), inputs in zero-padded (h+2) x (w+2) frame buffers.  The
            # sub-pixel taps in forward window order: tap a' = 1 - a of the window that starts at padded (y + py, x + px).
            for rp in ("d0", "d1"):
                d[rp + ".w3f.bf16"] = (d[rp + ".w3.f32"] * d[rp + ".s3"][:, None]).to(torch.bfloat16)
                d[rp + ".b3f"] = (d[rp + ".b3"] * d[rp + ".s3"] + d[rp + ".t3"]).contiguous()
            co = wt.shape[1]
            for py in range(2):
                for px in range(2):
                    wsub = d[f"d3.w{py}{px}.f32"].view(co, 2, 2, -1).flip(1, 2).reshape(co, -1)
                    d[f"d3.w{py}{px}f.bf16"] = (wsub * d["d3.s"][:, None]).to(torch.bfloat16)
            d["d3.bf"] = (d["d3.b"] * d["d3.s"] + d["d3.t"]).contiguous()
            # the four phases stacked (py, px) = (0,0) (0,1) (1,0) (1,1): one launch of the padded-taps GEMM (mage_gemm_desc::head_phases)
            d["d3.wallf.bf16"] = torch.cat([d[f"d3.w{py}{px}f.bf16"] for py in range(2) for px in range(2)], 0).contiguous()
            d["d3.bf4"] = d["d3.bf"].repeat(4).contiguous()
            d["d6.wt"] = dec[6].weight.float().permute(2, 3, 1, 0).contiguous()         # [4,4,cout,cin]
            # the same taps as GEMM rows [(ky*4+kx)*cout + co, cin], padded to a multiple of 8 rows (mage_gemm: N % 8 == 0)
            taps = d["d6.wt"].reshape(16 * self.input_dim, -1)
            both("d6.w16", taps)
            d["d6.b"] = dec[6].bias.float().contiguous()
        else:
            w8 = torch.zeros(enc[0].weight.shape[0], 7, 7, 8, device=enc[0].weight.device)       # [cout, ky, kx, ci padded to 8]
            w8[..., :enc[0].weight.shape[1]] = enc[0].weight.float().permute(0, 2, 3, 1)
            d["e0.w8"] = w8.reshape(w8.shape[0], -1).contiguous()
            d["e0.wt"] = enc[0].weight.float().permute(1, 2, 3, 0).contiguous()
            d["e0.b"] = enc[0].bias.float().contiguous()
            for i in (1, 3, 5, 7):
                bott(f"e{i}", enc[i])
            for i in (0, 2, 4, 6):
                bott(f"d{i}", dec[i])
            d["d8.wt"] = dec[8].weight.float().reshape(dec[8].weight.shape[0], -1).contiguous()   # [cout, cin]
            d["d8.b"] = dec[8].bias.float().contiguous()
            # the 1x1 RGB head taken on the tiles of the last block's closing convolution (mage_gemm_desc::head_w): 16 head rows, the first
            # input_dim of them the head's weights; the first four sums of a row ([row][4] fp32) then go through tanh(. + bias) by an identity 1x1
            h16 = torch.zeros(16, d["d8.wt"].shape[1], device=d["d8.wt"].device)
            h16[:self.input_dim] = d["d8.wt"]
            d["d8.w16.bf16"] = h16.to(torch.bfloat16).contiguous()
            d["d8.eye"] = torch.eye(self.input_dim, 4, device=d["d8.wt"].device).contiguous()      # the head's first 4 sums -> RGB (input_dim <= 4)
        return d

    def _weights(self) -> Dict[str, torch.Tensor]:
        return self._derived.get(self._build)

    def _bn_training(self) -> bool:
        return any(isinstance(m, nn.BatchNorm2d) and m.training for m in self.modules())

    def _check_input(self, x: torch.Tensor) -> None:
        if not x.is_cuda:
            raise RuntimeError("VectorQuantizedVAE runs on libmage_hip.so kernels: move the model and inputs to a ROCm GPU "
                               "(there is no CPU fallback)")
        if self._bn_training():
            raise NotImplementedError("encode() / decode() use the folded eval-mode BatchNorm; in training mode call forward() (batch "
                                      "statistics, modules/vqvae_train.py) or .eval() first (MAGE freezes the first stage: mage_model.py:516-521)")

    # ------------------------------------------------------------------ kernels: conv helpers
    @staticmethod
    def _conv(a, w, y, *, n_img, H, W, cin, cout, k, stride=1, pad=None, OH=None, OW=None, **epi):
        """Conv2d(cin, cout, k, stride, pad) on channels-last [n_img, H, W, cin] as one implicit GEMM."""
        pad = k // 2 if pad is None else pad
        OH = H if OH is None else OH
        OW = W if OW is None else OW
        return ops.gemm(a, w, y, M=n_img * OH * OW, N=cout, K=k * k * cin, lda=cin, ldy=cout, out_h=OH, out_w=OW, in_h=H,
                        in_w=W, taps_h=k, taps_w=k, cin=cin, stride=stride, dy0=-pad, dx0=-pad, **epi)

    def _resblock(self, w, p, r, dt, n_img, post_relu, H=16, W=16):
        """r = relu(x) already (in-place-ReLU quirk, vqvae_model.py:113): out = r + BN(conv1(relu(BN(conv3(r)))))."""
        dim, s = self.dim, "." + ("f32" if dt == torch.float32 else "bf16")
        t = torch.empty_like(r)
        self._conv(r, w[p + ".w3" + s], t, n_img=n_img, H=H, W=W, cin=dim, cout=dim, k=3, bias=w[p + ".b3"],
                   scale=w[p + ".s3"], shift=w[p + ".t3"], act=ops.ACT_RELU)
        out = torch.empty_like(r)
        self._conv(t, w[p + ".w1" + s], out, n_img=n_img, H=H, W=W, cin=dim, cout=dim, k=1, bias=w[p + ".b1"],
                   scale=w[p + ".s1"], shift=w[p + ".t1"], residual=r, ldr=dim, post_relu=post_relu)
        return out

    def _bottleneck(self, w, p, x, dt, n_img, H, W, cin, cout, first_k, last_k, post_relu, up_first=False, head=None):
        """up_first (decoder blocks behind an nn.Upsample, first_k = 1): x is the LOW-resolution input [n_img, H/2, W/2, cin].  A 1x1
        convolution and a ReLU act per pixel, so they commute with nearest-neighbour upsampling: the block's first convolution and its
        identity path run on a quarter of the pixels and nothing is upsampled at all: the second convolution gathers the first one's
        output at half resolution (mage_gemm a_half), the block's last convolution reads the identity path there (res_half) -- the
        same arithmetic per output pixel, bit-identical results.
        head (bf16 [16, cout], the decoder's last block): the block's closing 3x3 convolution runs in the padded-taps form on the 8-phase kernel
        (its input is written into a zero-padded frame buffer by the convolution before it) with the identity path, the ReLU that follows the
        block and the 1x1 head taken on the tile: returns the head's first four sums [n_img*H*W, 4] fp32; the block's output is never stored."""
        s = "." + ("f32" if dt == torch.float32 else "bf16")
        hid = cout // 4
        dev = x.device
        Hi, Wi = (H // 2, W // 2) if up_first else (H, W)
        # the block's leading ReLU: only its first convolution reads relu(x) (the identity path reads x).  A 1x1 first convolution with a
        # narrow hidden width takes it on its operand fragments (mage_gemm_desc::a_relu: the 256 x 64 tile) -- relu(x) is never stored
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count & ~7
        fold_relu = (dt == torch.bfloat16 and first_k == 1 and hid <= 128 and config.get().decode_relu_fold and not config.lib_flag("gemm_no_narrow")
                     and ((n_img * Hi * Wi + 255) // 256) * ((hid + 63) // 64) >= n_cu)
        xr = x if fold_relu else ops.relu(x, torch.empty_like(x))
        relu_kw = dict(a_relu=True) if fold_relu else {}
        if (p + ".wid" + s) in w:
            idp = torch.empty(n_img * Hi * Wi, cout, device=dev, dtype=dt)
            self._conv(x, w[p + ".wid" + s], idp, n_img=n_img, H=Hi, W=Wi, cin=cin, cout=cout, k=1, bias=w[p + ".bid"])
        else:
            idp = x
        ks = [first_k, 3, 3, last_k]
        chans = [cin, hid, hid, hid, cout]
        h = xr
        j0 = 0
        if up_first:
            assert first_k == 1
            h1 = torch.empty(n_img * Hi * Wi, hid, device=dev, dtype=dt)
            self._conv(xr, w[f"{p}.w1{s}"], h1, n_img=n_img, H=Hi, W=Wi, cin=cin, cout=hid, k=1, bias=w[f"{p}.b1"], act=ops.ACT_RELU, **relu_kw)
            h = h1                                  # stays at low resolution too: the next convolution gathers it there (a_half);
            j0 = 1                                  # the identity path is read there by the last convolution (res_half)
        Pw, PP = W + 2, (H + 2) * (W + 2)
        # the closing 3x3 convolution (hid -> cout, + identity path) in the padded-taps form on the 8-phase kernel, the identity rows fetched in
        # its epilogue (at half resolution behind an Upsample); with 64 hidden channels a K slab is one tap, so the sums keep the gather kernel's order
        taps_close = (head is None and last_k == 3 and not post_relu and dt == torch.bfloat16 and hid % 64 == 0 and cout % 256 == 0
                      and (n_img * H * W) % 256 == 0 and (n_img * PP + PP) * hid * 2 < 2 ** 32 and config.get().decode_taps8
                      and not config.lib_flag("gemm_no_8phase") and not config.lib_flag("gemm_no_taps8"))
        for j in range(j0, 3):
            half = dict(a_half=True, a_img_stride=Hi * Wi) if (up_first and j == 1) else {}
            if (head is not None or taps_close) and j == 2:             # the closing convolution's input: the interior of a zero-padded frame buffer
                key = ("f8tail", n_img, H, W, hid, str(dev), torch.cuda.current_stream(dev).cuda_stream)
                nh = self._pad_bufs.get(key)
                if nh is None:
                    if len(self._pad_bufs) > 6:
                        self._pad_bufs.clear()
                    nh = self._pad_bufs[key] = torch.zeros(n_img * PP + 1, hid, device=dev, dtype=dt)
                half = dict(half, y_img_stride=PP, y_mul_y=Pw, y_off=Pw + 1)
            else:
                nh = torch.empty(n_img * H * W, chans[j + 1], device=dev, dtype=dt)
            self._conv(h, w[f"{p}.w{2 * j + 1}{s}"], nh, n_img=n_img, H=H, W=W, cin=chans[j], cout=chans[j + 1], k=ks[j],
                       bias=w[f"{p}.b{2 * j + 1}"], act=ops.ACT_RELU, **half, **(relu_kw if j == 0 else {}))
            h = nh
        if head is not None:
            sums = torch.empty(n_img * H * W, 4, device=dev, dtype=torch.float32)            # (ldy == 4: the head's first four outputs only)
            ops.gemm(h, w[p + ".w7" + s], sums, M=n_img * H * W, N=cout, K=9 * hid, lda=hid, ldy=4, out_h=H, out_w=W, in_h=H + 2, in_w=Pw,
                     a_img_stride=PP, taps_h=3, taps_w=3, cin=hid, stride=1, dy0=0, dx0=0, bias=w[p + ".b7"], act=ops.ACT_RELU,
                     residual=idp, ldr=cout, res_half=up_first, head_w=head)
            return sums
        out = torch.empty(n_img * H * W, cout, device=dev, dtype=dt)
        if taps_close:
            ops.gemm(h, w[p + ".w7" + s], out, M=n_img * H * W, N=cout, K=9 * hid, lda=hid, ldy=cout, out_h=H, out_w=W, in_h=H + 2, in_w=Pw,
                     a_img_stride=PP, taps_h=3, taps_w=3, cin=hid, stride=1, dy0=0, dx0=0, bias=w[p + ".b7"], residual=idp, ldr=cout, res_half=up_first)
            return out
        self._conv(h, w[p + ".w7" + s], out, n_img=n_img, H=H, W=W, cin=hid, cout=cout, k=ks[3], bias=w[p + ".b7"],
                   residual=idp, ldr=cout, post_relu=post_relu, res_half=up_first)
        return out

    # ------------------------------------------------------------------ encoder (always fp32: bit-exact tokens)
    @torch.no_grad()
    def _encode_features(self, x: torch.Tensor) -> torch.Tensor:
        """z_e as channels-last rows [N*h*w, D] fp32."""
        self._check_input(x)
        w = self._weights()
        x = x.float().contiguous()
        N, dev, dim, f = x.shape[0], x.device, self.dim, torch.float32
        if self.down_ratio == 4:
            H, W = x.shape[2], x.shape[3]
            if (dim % 256 == 0 and H % 4 == 0 and W % 4 == 0 and (N * (H // 4) * (W // 4)) % 256 == 0 and self.input_dim <= 4
                    and config.get().encode_split and self.encode_split):
                return self._encode_f4_split(w, x, N, H, W)
            h0 = torch.empty(N * (H // 2) * (W // 2), dim, device=dev, dtype=f)
            ops.conv_in(x, w["e0.wt"], w["e0.b"], w["e0.s"], w["e0.t"], h0, cin=self.input_dim, H=H, W=W, cout=dim, kh=4,
                        kw=4, stride=2, pad=1, act=ops.ACT_RELU)
            h1 = torch.empty(N * (H // 4) * (W // 4), dim, device=dev, dtype=f)
            # relu folded: the only consumers of conv3's output are ResBlock e4's skip and body, both behind its in-place ReLU
            self._conv(h0, w["e3.w.f32"], h1, n_img=N, H=H // 2, W=W // 2, cin=dim, cout=dim, k=4, stride=2, pad=1,
                       OH=H // 4, OW=W // 4, bias=w["e3.b"], act=ops.ACT_RELU)
            if H % 4 or W % 4:
                raise ValueError(f"f4 VQ-VAE input {H}x{W} must be a multiple of 4 in both dimensions")
            h2 = self._resblock(w, "e4", h1, f, N, post_relu=True, H=H // 4, W=W // 4)
            return self._resblock(w, "e5", h2, f, N, post_relu=False, H=H // 4, W=W // 4)
        H, W = x.shape[2], x.shape[3]
        h = self._stem7(w, x)
        chans = [(dim, dim), (dim, dim), (dim, 2 * dim), (2 * dim, 4 * dim)]
        for bi, (ci, co) in zip((1, 3, 5, 7), chans):
            last = bi == 7
            h = self._bottleneck(w, f"e{bi}", h, f, N, H, W, ci, co, 3, 1, post_relu=last)   # trailing nn.ReLU (:201)
            if not last:
                p = torch.empty(N * (H // 2) * (W // 2), co, device=dev, dtype=f)
                ops.maxpool2(h, p, N=N, H=H, W=W, Cc=co)
                h, H, W = p, H // 2, W // 2
        return h

    def _enc_split_weights(self, w):
        """f16x3 operands of the f4 encoder's GEMM-shaped convolutions (built once per weights): the 4x4 / stride-2 convolution as a
        2x2 window over offset space-to-depth blocks (w2[co, by, bx, dy, dx, ci] = W[co, ci, 2by+dy, 2bx+dx]), the ResBlocks' 3x3 and 1x1
        convolutions with their eval BatchNorm folded in (W' = alpha W, b' = alpha b + beta)."""
        if "e3.ws" not in w:
            sk, dim = ops.F16X3, self.dim
            W4 = self.encoder[3].weight.float()                                            # [cout, cin, 4, 4]
            w2 = W4.view(dim, dim, 2, 2, 2, 2).permute(0, 2, 4, 3, 5, 1).reshape(dim, 16 * dim).contiguous()   # [co, by, bx, dy, dx, ci]
            w["e3.ws"] = ops.split(w2, sk)
            for p in ("e4", "e5"):
                w[p + ".w3s"] = ops.split((w[p + ".w3.f32"] * w[p + ".s3"][:, None]).contiguous(), sk)
                w[p + ".b3f"] = (w[p + ".b3"] * w[p + ".s3"] + w[p + ".t3"]).contiguous()
                w[p + ".w1s"] = ops.split((w[p + ".w1.f32"] * w[p + ".s1"][:, None]).contiguous(), sk)
                w[p + ".b1f"] = (w[p + ".b1"] * w[p + ".s1"] + w[p + ".t1"]).contiguous()
        return w

    def _encode_f4_split(self, w, x: torch.Tensor, N: int, H: int, W: int) -> torch.Tensor:
        """The f4 encoder (vqvae_model.py:172-179) with its four GEMM-shaped convolutions on split-precision operands (MAGE_F16X3:
        three f16 MFMA products per K slab, fp32 accumulation -- fp32-class results, measured 5e-6 per GEMM against the exact-fp32
        chain's 1e-5) in the padded-taps form of the 8-phase kernel, 2.5x the exact-fp32 MFMA rate:
          stem (direct kernel) -> split rows in the offset space-to-depth layout -> Conv 4x4/s2 as a 2x2 window over 4*dim channels
          -> ReLU (the ResBlock's in-place one) -> per ResBlock: rows into a zero-padded split frame buffer, 3x3 (+BN, ReLU) -> split
          rows, 1x1 (+BN) + the fp32 skip.
        The quantiser that follows stays on the fp64 matrix cores; the golden token gates are unchanged."""
        sk, dim, dev = ops.F16X3, self.dim, x.device
        self._enc_split_weights(w)
        h, wd = H // 4, W // 4
        hw, BH, BW = h * wd, H // 4 + 1, W // 4 + 1
        PPb, Pw = BH * BW, wd + 2
        PP = (h + 2) * Pw
        key = ("enc", N, H, W, str(dev), torch.cuda.current_stream(dev).cuda_stream)
        bufs = self._pad_bufs.get(key)
        if bufs is None:
            if len(self._pad_bufs) > 4:
                self._pad_bufs.clear()
            bufs = self._pad_bufs[key] = (ops.split_empty(N * PPb + 1, 4 * dim, sk, dev, zero=True), ops.split_empty(N * PP + 1, dim, sk, dev, zero=True))
        y1, pad = bufs
        ops.conv_in(x, w["e0.wt"], w["e0.b"], w["e0.s"], w["e0.t"], y1, cin=self.input_dim, H=H, W=W, cout=dim, kh=4, kw=4, stride=2, pad=1,
                    act=ops.ACT_RELU, split_kind=sk, s2d=True)
        r = torch.empty(N * hw, dim, device=dev, dtype=torch.float32)
        ops.gemm(y1, w["e3.ws"], r, M=N * hw, N=dim, K=16 * dim, lda=8 * dim, ldy=dim, out_h=h, out_w=wd, in_h=BH, in_w=BW, a_img_stride=PPb,
                 taps_h=2, taps_w=2, cin=4 * dim, bias=w["e3.b"], act=ops.ACT_RELU, split_kind=sk)
        t = ops.split_empty(N * hw, dim, sk, dev)
        for i, p in enumerate(("e4", "e5")):
            # r = relu(block input): the skip path (fp32) and, split into the padded frame buffer, the 3x3 convolution's input
            ops.split_rows(r, pad, sk, relu=i > 0, relu_writeback=i > 0, group=hw, group_stride=PP, off=Pw + 1, inner=wd, inner_stride=Pw)
            ops.gemm(pad, w[p + ".w3s"], t, M=N * hw, N=dim, K=9 * dim, lda=2 * dim, ldy=2 * dim, out_h=h, out_w=wd, in_h=h + 2, in_w=Pw,
                     a_img_stride=PP, taps_h=3, taps_w=3, cin=dim, bias=w[p + ".b3f"], act=ops.ACT_RELU, split_kind=sk, y_split=True)
            z = torch.empty(N * hw, dim, device=dev, dtype=torch.float32)
            ops.gemm(t, w[p + ".w1s"], z, M=N * hw, N=dim, K=dim, lda=2 * dim, ldy=dim, bias=w[p + ".b1f"], residual=r, ldr=dim, split_kind=sk)
            r = z
        return r

    def _stem7(self, w, x: torch.Tensor) -> torch.Tensor:
        """The f8 stem Conv2d(C, dim, 7, padding=3) on full-resolution frames as an implicit GEMM over the image laid out as
        channels-last rows padded to 8 channels (layout plumbing; K = 49 * 8): the per-pixel direct kernel ran at 14 TFLOP/s
        (13.9 ms for 160 frames of 128x128), the fp32 MFMA GEMM at ~110."""
        N, Cin, H, W = x.shape
        dim = self.dim
        h = torch.empty(N * H * W, dim, device=x.device, dtype=torch.float32)
        if Cin > 8 or "e0.w8" not in w:
            return ops.conv_in(x, w["e0.wt"], w["e0.b"], None, None, h, cin=Cin, H=H, W=W, cout=dim, kh=7, kw=7, stride=1, pad=3)
        xr = torch.zeros(N * H * W, 8, device=x.device, dtype=torch.float32)
        xr[:, :Cin] = x.permute(0, 2, 3, 1).reshape(N * H * W, Cin)
        return self._conv(xr, w["e0.w8"], h, n_img=N, H=H, W=W, cin=8, cout=dim, k=7, bias=w["e0.b"])

    @torch.no_grad()
    def encode(self, x: torch.Tensor) -> torch.Tensor:
        """vqvae_model.py:233-237 -> int64 [N, h, w]."""
        z = self._encode_features(x)
        w = self._weights()
        hw = x.shape[2] // self.down_ratio
        return ops.vq_nearest(z, w["cbt"], w["c2"]).view(x.shape[0], hw, x.shape[3] // self.down_ratio)

    # ------------------------------------------------------------------ decoder
    @torch.no_grad()
    def decode(self, latents: torch.Tensor = None) -> torch.Tensor:
        """vqvae_model.py:239-242: int64 [N, h, w] -> fp32 [N, C, H, W] in (-1, 1).  Ids outside [0, K) raise (ValueError), as the
        reference's nn.Embedding does (IndexError)."""
        out = self._decode_nocheck(latents)
        ops.check_device_errors(latents.device)
        return out

    @torch.no_grad()
    def _decode_nocheck(self, latents: torch.Tensor) -> torch.Tensor:
        """decode without the closing device-error check (which synchronises the stream): for callers that check once at the
        end of their own call (MAGE.autoregressive_generate)."""
        self._check_input(latents)
        N = latents.shape[0]
        out = torch.empty(N, self.input_dim, latents.shape[1] * self.down_ratio, latents.shape[2] * self.down_ratio,
                          device=latents.device, dtype=torch.float32)
        ids = latents.to(torch.int64).contiguous()
        for s in range(0, N, self.decode_chunk):
            e = min(N, s + self.decode_chunk)
            self._decode_chunk(ids[s:e], out[s:e])
        return out

    def _d0_table(self, w):
        """T[tap][code] = (BatchNorm-folded) W3_tap relu(codebook[code]) of the decoder's first ResBlock (vqvae_model.py:111-124,180),
        bf16 [9, K, dim]; built once per weights on the fp32 MFMA kernel.  None when switched off (MAGE_NO_DECODE_TABLE=1)."""
        if not config.get().decode_table:             # switched off for this call: not cached (a config.override block must not stick)
            return None
        if "d0.tab" not in w:
            if 9 * self.K * self.dim * 2 > (64 << 20):
                w["d0.tab"] = None
            else:
                dim, Kc, dev = self.dim, self.K, w["cb"].device
                rcb = torch.relu(w["cb"]).contiguous()                                              # [K, dim] fp32 (derived-cache bookkeeping)
                w3 = (w["d0.w3.f32"] * w["d0.s3"][:, None]).view(dim, 9, dim)
                T = torch.empty(9, Kc, dim, device=dev, dtype=torch.float32)
                for tap in range(9):
                    ops.gemm(rcb, w3[:, tap].contiguous(), T[tap], M=Kc, N=dim, K=dim, lda=dim, ldy=dim)
                w["d0.tab"] = T.to(torch.bfloat16)
        return w["d0.tab"]

    def _dec_split_weights(self, w):
        """f16x3 operands of the f4 decoder (built once per weights): the first ResBlock's 3x3 convolution as an fp32 table over
        relu(codebook) (_d0_table's fp32 twin), the second one's 3x3 and both 1x1 convolutions with their BatchNorm folded in, the four
        sub-pixel 2x2 convolutions of ConvTranspose2d(dim, dim, 4, 2, 1) + BatchNorm in forward window order."""
        if "d0.tab32" not in w:
            sk, dim, Kc = ops.F16X3, self.dim, self.K
            rcb = torch.relu(w["cb"]).contiguous()
            w3 = (w["d0.w3.f32"] * w["d0.s3"][:, None]).view(dim, 9, dim)
            T = torch.empty(9, Kc, dim, device=rcb.device, dtype=torch.float32)
            for tap in range(9):
                ops.gemm(rcb, w3[:, tap].contiguous(), T[tap], M=Kc, N=dim, K=dim, lda=dim, ldy=dim)
            w["d0.tab32"] = T
            w["d1.w3s"] = ops.split((w["d1.w3.f32"] * w["d1.s3"][:, None]).contiguous(), sk)
            for p in ("d0", "d1"):
                w[p + ".w1s"] = ops.split((w[p + ".w1.f32"] * w[p + ".s1"][:, None]).contiguous(), sk)
                w[p + ".b1f"] = (w[p + ".b1"] * w[p + ".s1"] + w[p + ".t1"]).contiguous()
            for py in range(2):
                for px in range(2):
                    wsub = w[f"d3.w{py}{px}.f32"].view(dim, 2, 2, -1).flip(1, 2).reshape(dim, -1)
                    w[f"d3.w{py}{px}s"] = ops.split((wsub * w["d3.s"][:, None]).contiguous(), sk)
        return w

    def _decode_chunk_split(self, w, ids: torch.Tensor, out: torch.Tensor) -> None:
        """The f4 decoder (vqvae_model.py:180-189) in the fast parity modes: every 256-channel convolution on f16x3 operands (fp32-class
        frames at ~2.5x the exact-fp32 gather kernels' speed), the C-channel head on the fp32 narrow-tile GEMM as before."""
        sk, dim, dev = ops.F16X3, self.dim, ids.device
        self._dec_split_weights(w)
        N, h, wd = ids.shape
        hw, Pw = h * wd, wd + 2
        PP = (h + 2) * Pw
        key = ("dec_s", N, h, wd, str(dev), torch.cuda.current_stream(dev).cuda_stream)
        pad = self._pad_bufs.get(key)
        if pad is None:
            if len(self._pad_bufs) > 6:
                self._pad_bufs.clear()
            pad = self._pad_bufs[key] = ops.split_empty(N * PP + 1, dim, sk, dev, zero=True)
        inner = dict(group=hw, group_stride=PP, off=Pw + 1, inner=wd, inner_stride=Pw)
        flat = ids.reshape(-1)
        r = ops.embedding(flat, w["cb"], torch.empty(N * hw, dim, device=dev, dtype=torch.float32), relu=True)        # the skip path of d0
        t = ops.split_empty(N * hw, dim, sk, dev)
        ops.table_conv(flat, w["d0.tab32"], t, n_img=N, H=h, W=wd, bias=w["d0.b3f"], relu=True, split_kind=sk)       # conv3x3 + BN + ReLU of d0
        z = torch.empty(N * hw, dim, device=dev, dtype=torch.float32)
        ops.gemm(t, w["d0.w1s"], z, M=N * hw, N=dim, K=dim, lda=2 * dim, ldy=dim, bias=w["d0.b1f"], residual=r, ldr=dim, split_kind=sk)
        ops.split_rows(z, pad, sk, relu=True, relu_writeback=True, **inner)                                           # d1's in-place ReLU
        win = dict(out_h=h, out_w=wd, in_h=h + 2, in_w=Pw, a_img_stride=PP, cin=dim)
        ops.gemm(pad, w["d1.w3s"], t, M=N * hw, N=dim, K=9 * dim, lda=2 * dim, ldy=2 * dim, taps_h=3, taps_w=3, bias=w["d1.b3f"],
                 act=ops.ACT_RELU, split_kind=sk, y_split=True, **win)
        z2 = torch.empty(N * hw, dim, device=dev, dtype=torch.float32)
        ops.gemm(t, w["d1.w1s"], z2, M=N * hw, N=dim, K=dim, lda=2 * dim, ldy=dim, bias=w["d1.b1f"], residual=z, ldr=dim, split_kind=sk)
        ops.split_rows(z2, pad, sk, relu=True, **inner)                                                               # decoder[2] ReLU
        up = torch.empty(N * 4 * hw, dim, device=dev, dtype=torch.float32)
        for py in range(2):
            for px in range(2):
                ops.gemm(pad, w[f"d3.w{py}{px}s"], up, M=N * hw, N=dim, K=4 * dim, lda=2 * dim, ldy=dim, taps_h=2, taps_w=2,
                         a_off=py * Pw + px, y_img_stride=4 * hw, y_mul_y=4 * wd, y_mul_x=2, y_off=py * 2 * wd + px, bias=w["d3.bf"],
                         act=ops.ACT_RELU, split_kind=sk, **win)
        nt = 16 * self.input_dim
        taps = torch.empty(N * 4 * hw, nt, device=dev, dtype=torch.float32)
        ops.gemm(up, w["d6.w16.f32"], taps, M=N * 4 * hw, N=nt, K=dim, lda=dim, ldy=nt)
        ops.convt_fold_tanh(taps, w["d6.b"], out, N=N, IH=2 * h, IW=2 * wd, cout=self.input_dim)

    def _decode_chunk(self, ids: torch.Tensor, out: torch.Tensor) -> None:
        w = self._weights()
        dt = self.decode_dtype
        if (self.down_ratio == 4 and getattr(self, "decode_split", 0) and self.dim % 256 == 0 and (ids.shape[0] * ids.shape[1] * ids.shape[2]) % 256 == 0
                and config.get().decode_split):
            return self._decode_chunk_split(w, ids, out)
        s = "." + ("f32" if dt == torch.float32 else "bf16")
        N, dev, dim = ids.shape[0], ids.device, self.dim
        h, wd = ids.shape[1], ids.shape[2]
        if self.down_ratio == 4 and dt == torch.bfloat16 and dim % 256 == 0 and (N * h * wd) % 256 == 0 and config.get().decode_taps8:
            # the two 3x3 convolutions and the four sub-pixel convolutions on the 8-phase GEMM kernel (padded-taps form): every
            # activation that a windowed layer reads lives in a zero-padded frame buffer, written there by its producer
            hw, Pw = h * wd, wd + 2
            PP = (h + 2) * Pw                                         # rows per padded image
            # the first ResBlock reads relu(codebook[ids]): its 3x3 convolution is a table sum, and with 16-wide frames and dim == 256 the
            # whole block is ONE launch (mage_resblock_table): neither the embedded frames nor t ever reach HBM
            fused0 = (self._d0_table(w) is not None and wd == 16 and h % 2 == 0 and dim == 256
                      and config.get().decode_resblock_fusion)
            key = (N, h, wd, str(dev), torch.cuda.current_stream(dev).cuda_stream, fused0)
            pads = self._pad_bufs.get(key)
            if pads is None:
                if len(self._pad_bufs) > 4:
                    self._pad_bufs.clear()
                # (pads[0] holds the embedded frames: nobody writes or reads it when the first block is the fused kernel)
                pads = self._pad_bufs[key] = [None if (fused0 and i == 0) else torch.zeros(N * PP + 1, dim, device=dev, dtype=dt) for i in range(3)]
            inner = dict(out_h=h, out_w=wd, y_img_stride=PP, y_mul_y=Pw, y_off=Pw + 1)      # a producer's rows inside the padding
            win = dict(out_h=h, out_w=wd, in_h=h + 2, in_w=Pw, a_img_stride=PP, cin=dim, stride=1, dy0=0, dx0=0)
            if not fused0:
                ops.embedding(ids, w["cb"], pads[0], relu=True, group=hw, group_stride=PP, off=Pw + 1, inner=wd, inner_stride=Pw)
            t = torch.empty(N * hw, dim, device=dev, dtype=dt)
            for i, rp in enumerate(("d0", "d1")):
                if i == 0 and fused0:
                    ops.resblock_table(ids.reshape(-1), w["d0.tab"], w["cb"], w["d0.w1.bf16"], pads[1], n_img=N, H=h, W=wd, bias3=w["d0.b3f"],
                                       b1=w["d0.b1"], scale1=w["d0.s1"], shift1=w["d0.t1"], post_relu=True, ldy=dim, y_img_stride=PP,
                                       y_row_pitch=Pw, y_off=Pw + 1)
                    continue
                if i == 0 and self._d0_table(w) is not None:
                    # the first 3x3 convolution reads relu(codebook[ids]): K distinct input vectors -> a table sum (mage_table_conv;
                    # 9 x K x dim bf16 = 2.4 MB: resident in every XCD's L2), a quarter of the stack's matrix-core FLOPs not spent
                    ops.table_conv(ids.reshape(-1), w["d0.tab"], t, n_img=N, H=h, W=wd, bias=w["d0.b3f"], relu=True)
                else:
                    ops.gemm(pads[i], w[rp + ".w3f.bf16"], t, M=N * hw, N=dim, K=9 * dim, lda=dim, ldy=dim, taps_h=3, taps_w=3, bias=w[rp + ".b3f"],
                             act=ops.ACT_RELU, **win)
                if dim == 256 and (N * hw) % 64 == 0 and config.get().decode_resblock_fusion:
                    # the block's tail as an HBM-bound row kernel (mage_resblock_rows: whole rows in and out, W1 in registers): same bits
                    ops.resblock_rows(t, w[rp + ".w1.bf16"], pads[i], pads[i + 1], n_img=N, H=h, W=wd, b1=w[rp + ".b1"], scale1=w[rp + ".s1"],
                                      shift1=w[rp + ".t1"], post_relu=True, lda=dim, ldr=dim, ldy=dim, img_stride=PP, row_pitch=Pw, off=Pw + 1)
                    continue
                ops.gemm(t, w[rp + ".w1.bf16"], pads[i + 1], M=N * hw, N=dim, K=dim, lda=dim, ldy=dim, bias=w[rp + ".b1"], scale=w[rp + ".s1"],
                         shift=w[rp + ".t1"], residual=pads[i], ldr=dim, post_relu=True, **inner)                # decoder[2] ReLU folded
            nt = 16 * self.input_dim
            taps = torch.empty(N * 4 * hw, nt, device=dev, dtype=torch.float32)
            # one output channel and dim == 256 (one column tile of the GEMM holds whole rows): the last transposed convolution's 4 x 4 taps
            # are taken on the sub-pixel GEMMs' tiles before they leave the CU (mage_gemm_desc::head_w) -- the 4x-resolution activation
            # `up` (0.5 GB per 960 frames, written once and read once) and the head GEMM's launch are gone
            # (head_w is a fusion of the 8-phase padded-taps kernel: with that kernel switched off in the library the unfused launches run)
            head = (nt == 16 and dim == 256 and (N * PP + PP) * dim * 2 < 2 ** 32 and config.get().decode_head_fusion
                    and not config.lib_flag("gemm_no_8phase") and not config.lib_flag("gemm_no_taps8"))
            up = None if head else torch.empty(N * 4 * hw, dim, device=dev, dtype=dt)
            if head and config.get().decode_phase_merge:
                # ... and the four sub-pixel launches are one: a frame's four phases are neighbouring tiles (same bits as four launches)
                ops.gemm(pads[2], w["d3.wallf.bf16"], taps, M=N * hw, N=4 * dim, K=4 * dim, lda=dim, ldy=nt, taps_h=2, taps_w=2, a_off=0,
                         y_img_stride=4 * hw, y_mul_y=4 * wd, y_mul_x=2, y_off=0, bias=w["d3.bf4"], act=ops.ACT_RELU, head_w=w["d6.w16" + s],
                         head_phases=4, **win)
                ops.convt_fold_tanh(taps, w["d6.b"], out, N=N, IH=2 * h, IW=2 * wd, cout=self.input_dim)
                return
            for py in range(2):
                for px in range(2):
                    ops.gemm(pads[2], w[f"d3.w{py}{px}f.bf16"], taps if head else up, M=N * hw, N=dim, K=4 * dim, lda=dim, ldy=nt if head else dim,
                             taps_h=2, taps_w=2, a_off=py * Pw + px, y_img_stride=4 * hw, y_mul_y=4 * wd, y_mul_x=2, y_off=py * 2 * wd + px,
                             bias=w["d3.bf"], act=ops.ACT_RELU, head_w=w["d6.w16" + s] if head else None, **win)
            if not head:
                ops.gemm(up, w["d6.w16" + s], taps, M=N * 4 * hw, N=nt, K=dim, lda=dim, ldy=nt)
            ops.convt_fold_tanh(taps, w["d6.b"], out, N=N, IH=2 * h, IW=2 * wd, cout=self.input_dim)
            return
        if self.down_ratio == 4:
            hw = h * wd
            r = ops.embedding(ids, w["cb"], torch.empty(N * hw, dim, device=dev, dtype=dt), relu=True)
            r = self._resblock(w, "d0", r, dt, N, post_relu=True, H=h, W=wd)
            r = self._resblock(w, "d1", r, dt, N, post_relu=True, H=h, W=wd)          # decoder[2] ReLU folded
            up = torch.empty(N * 4 * hw, dim, device=dev, dtype=dt)
            for py in range(2):
                for px in range(2):
                    ops.gemm(r, w[f"d3.w{py}{px}{s}"], up, M=N * hw, N=dim, K=4 * dim, lda=dim, ldy=dim, out_h=h, out_w=wd,
                             in_h=h, in_w=wd, taps_h=2, taps_w=2, cin=dim, stride=1, dy0=py, dx0=px, dys=-1, dxs=-1,
                             y_img_stride=4 * hw, y_mul_y=4 * wd, y_mul_x=2, y_off=py * 2 * wd + px, bias=w["d3.b"], scale=w["d3.s"],
                             shift=w["d3.t"], act=ops.ACT_RELU)
            # ConvTranspose2d(dim, C, 4, 2, 1) + Tanh as GEMM + fold: `up` (the widest activation of the stack) is read once
            nt = 16 * self.input_dim
            taps = torch.empty(N * 4 * hw, nt, device=dev, dtype=torch.float32)
            ops.gemm(up, w["d6.w16" + s], taps, M=N * 4 * hw, N=nt, K=dim, lda=dim, ldy=nt)
            ops.convt_fold_tanh(taps, w["d6.b"], out, N=N, IH=2 * h, IW=2 * wd, cout=self.input_dim)
            return
        H, W = h, wd
        x = ops.embedding(ids, w["cb"], torch.empty(N * H * W, 4 * dim, device=dev, dtype=dt))
        chans = [(4 * dim, 2 * dim), (2 * dim, dim), (dim, dim), (dim, dim)]
        for bi, (ci, co) in zip((0, 2, 4, 6), chans):
            last = bi == 6
            # the last block's closing convolution + identity path + decoder[7] ReLU + the 1x1 head (decoder[8]) in one launch: the
            # block's [N, H, W, dim] output (8 MB per 128 x 128 frame) is neither written nor read
            fuse_tail = (last and dt == torch.bfloat16 and co == 256 and (co // 4) % 64 == 0 and (N * H * W) % 256 == 0 and self.input_dim <= 4
                         and (N * (H + 2) * (W + 2) + (H + 2) * (W + 2)) * (co // 4) * 2 < 2 ** 32 and config.get().decode_head_fusion
                         and not config.lib_flag("gemm_no_8phase") and not config.lib_flag("gemm_no_taps8"))
            # the nn.Upsample in front of blocks 2, 4, 6 is folded into the block (see _bottleneck up_first)
            x = self._bottleneck(w, f"d{bi}", x, dt, N, H, W, ci, co, 1, 3, post_relu=last, up_first=bi != 0,      # decoder[7] ReLU folded
                                 head=w["d8.w16.bf16"] if fuse_tail else None)
            if not last:
                H, W = H * 2, W * 2
        if fuse_tail:
            ops.conv_out(x, w["d8.eye"], w["d8.b"], out, N=N, IH=H, IW=W, cin=4, cout=self.input_dim, transposed=False)   # tanh(sums + bias) -> NCHW
        else:
            ops.conv_out(x, w["d8.wt"], w["d8.b"], out, N=N, IH=H, IW=W, cin=dim, cout=self.input_dim, transposed=False)

    # ------------------------------------------------------------------ forward (values only)
    def forward(self, x: torch.Tensor):
        """vqvae_model.py:244-248: (x_tilde, z_e_x, z_q_x), NCHW.  eval(): values from the inference kernels.  train(): BatchNorm on
        batch statistics; in grad mode the three outputs hang off one autograd node (vqvae_train.VQVAEForwardFn) whose backward is
        the straight-through estimator + every convolution / BatchNorm gradient on the HIP kernels (train_vqvae.py:13-35; both the
        f4 and the f8 stack)."""
        f8_graph = (self.down_ratio == 8 and self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()))
        if self._bn_training() or f8_graph:                       # the f8 stack has no BatchNorm: train() + grad mode asks for the graph
            from . import vqvae_train
            if not x.is_cuda:
                raise RuntimeError("VectorQuantizedVAE runs on libmage_hip.so kernels: move the model and inputs to a ROCm GPU")
            with torch.cuda.device(x.device):
                if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
                    names = [n for n, p in self.named_parameters() if p.requires_grad]
                    byname = dict(self.named_parameters())
                    return vqvae_train.VQVAEForwardFn.apply(self, x, names, *[byname[n] for n in names])
                with torch.no_grad():
                    x_tilde, z_e, zq, t = vqvae_train.vq_train_forward(self, x)
                N, hh, ww, D = t["N"], t["h"], t["wd"], self.dim
                return x_tilde, z_e.view(N, hh, ww, D).permute(0, 3, 1, 2), zq.view(N, hh, ww, D).permute(0, 3, 1, 2)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()) and not getattr(self, "_warned_eval_grad", False):
            import warnings
            self._warned_eval_grad = True
            warnings.warn("VectorQuantizedVAE.forward in eval() returns VALUES from the inference kernels (no autograd graph): a "
                          "loss.backward() on them fails.  Call model.train() for the training graph (train_vqvae.py:13-35), or wrap "
                          "evaluation in torch.no_grad() as the reference's test loop does (train_vqvae.py:38-55).", stacklevel=2)
        with torch.no_grad():
            return self._forward_eval(x)

    def _forward_eval(self, x: torch.Tensor):
        z = self._encode_features(x)
        w = self._weights()
        N = x.shape[0]
        hh, ww = x.shape[2] // self.down_ratio, x.shape[3] // self.down_ratio
        ids = ops.vq_nearest(z, w["cbt"], w["c2"]).view(N, hh, ww)
        D = z.shape[1]
        z_q = ops.embedding(ids, w["cb"], torch.empty(N * hh * ww, D, device=x.device, dtype=torch.float32))
        x_tilde = self._decode_nocheck(ids)
        return x_tilde, z.view(N, hh, ww, D).permute(0, 3, 1, 2), z_q.view(N, hh, ww, D).permute(0, 3, 1, 2)
