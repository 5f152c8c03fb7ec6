"""Denoiser protocol + the FFDNet denoisers of the plug-and-play prior
(reference dprox/proxfn/pnp/denoisers/base.py:5-25, wrapper.py:25-48, models/network_ffdnet.py:27-68).

``FFDNet`` here is not an ``nn.Sequential`` of cuDNN/MIOpen convolutions: its forward is the
hand-written gfx950 kernel stack behind ``dpx_ffdnet_forward`` (exact-fp32 MFMA implicit GEMM,
pixel-(un)shuffle, sigma map, bias and ReLU fused).  Checkpoints in the reference's format
(``model.{0,2,...}.weight/bias``) load unchanged.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from ... import _backend as be
from ... import _ops as ops


class Denoiser(nn.Module):
    def denoise(self, input: torch.Tensor, sigma: torch.Tensor):
        """input: [N,C,H,W]; sigma: 0-d or [N]"""
        sigma = sigma.view(-1, 1, 1, 1)
        out = self._denoise(input, sigma)
        try:
            be.check_f16_range("denoise")       # (no-op inside a solve: checked once at its end)
        except be.F16RangeError:
            if not be.f16_fallback(self.modules(), "denoise"):
                raise
            out = self._denoise(input, sigma)    # split-bf16: fp32's range
        return out

    def _denoise(self, x, sigma):
        raise NotImplementedError


class Denoiser2D(Denoiser):
    """applies a single-channel denoiser band by band"""

    def denoise(self, input: torch.Tensor, sigma: torch.Tensor):
        sigma = sigma.view(-1, 1, 1, 1)
        run = lambda: torch.cat([self._denoise(band.contiguous(), sigma) for band in input.split(1, dim=1)], dim=1)
        out = run()
        try:
            be.check_f16_range("denoise")
        except be.F16RangeError:
            if not be.f16_fallback(self.modules(), "denoise"):
                raise
            out = run()
        return out


class RefKeyed(nn.Module):
    """Base of the hand-written networks: parameters are plain ``nn.Parameter``s of THIS module, registered under attribute
    names without dots, and saved / loaded under the reference model's dotted state-dict keys (``model.0.weight``,
    ``m_down1.0.res.0.weight``, ``inc.conv.conv-0.conv2d.weight`` ...) through nn.Module's own recursive protocol
    (``_save_to_state_dict`` / ``_load_from_state_dict``): a parent module's ``state_dict()`` / ``load_state_dict()`` and the
    reference's checkpoints both work unchanged."""

    def __init__(self):
        super().__init__()
        self._ref_attr = {}                   # reference key -> attribute name (insertion order = reference order)

    def add_ref_param(self, key, shape, requires_grad=True):
        attr = "p__" + key.replace(".", "__")
        self.register_parameter(attr, nn.Parameter(torch.zeros(*shape), requires_grad=requires_grad))
        self._ref_attr[key] = attr

    def ref_param(self, key):
        return getattr(self, self._ref_attr[key])

    @property
    def ref_keys(self):
        return list(self._ref_attr)

    def _weights_changed(self):
        """drop the packed-weight caches"""
        self._packed = None
        self._packed_T = None

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        for key, attr in self._ref_attr.items():
            p = getattr(self, attr)
            destination[prefix + key] = p if keep_vars else p.detach()

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        for key, attr in self._ref_attr.items():
            full = prefix + key
            if full not in state_dict:
                missing_keys.append(full)
                continue
            p, v = getattr(self, attr), torch.as_tensor(state_dict[full])
            if tuple(v.shape) != tuple(p.shape):
                error_msgs.append(f"size mismatch for {full}: checkpoint {tuple(v.shape)} vs model {tuple(p.shape)}")
                continue
            with torch.no_grad():
                p.copy_(v)
        if strict:
            unexpected_keys.extend(k for k in state_dict if k.startswith(prefix) and k[len(prefix):] not in self._ref_attr)
        self._weights_changed()

    def _apply(self, fn, *a, **k):
        self._weights_changed()
        return super()._apply(fn, *a, **k)

    def _weights_version(self):
        return tuple(getattr(self, a)._version for a in self._ref_attr.values())


class FFDNet(RefKeyed):
    """FFDNet(in_nc, out_nc, nc, nb): conv3x3(in_nc*4+1 -> nc) + (nb-2) x conv3x3(nc -> nc) + conv3x3(nc -> out_nc*4)"""

    def __init__(self, in_nc=1, out_nc=1, nc=64, nb=15, act_mode="R"):
        super().__init__()
        assert "R" in act_mode, "only the ReLU FFDNet variants are built for the HIP path"
        assert in_nc == out_nc
        self.in_nc, self.out_nc, self.nc, self.nb = in_nc, out_nc, nc, nb
        chans = [in_nc * 4 + 1] + [nc] * (nb - 1) + [out_nc * 4]
        for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:])):        # reference layout: Conv2d at the even indices of `model`
            self.add_ref_param(f"model.{2 * i}.weight", (co, ci, 3, 3))
            self.add_ref_param(f"model.{2 * i}.bias", (co,))
        self._packed = None
        # arithmetic of the (non-differentiable) forward pass:
        #   "bf16x3" -- fp32 accuracy on the bf16 matrix cores (weights / activations split into three bf16 terms, six products,
        #               fp32 accumulation: dpx_conv_bf16.hip); any operand range;
        #   "f32"    -- the f32-input matrix instruction (bitwise an fmaf chain); also what the differentiable path uses;
        #   "f16x2"  -- fp32 accuracy on the f16 matrix cores: x = hi + lo / 2^11 with two binary16 terms, three products (half the
        #               matrix work of "bf16x3"; 1e-7 from float64 on this stack, like fp32 itself).  Operands must stay inside the
        #               binary16 range (|x| < 6e4) -- a solve / denoise() call that met one that did not raises (be.check_f16_range).
        #               The default wherever the layer widths are multiples of 16;
        #   "f16x2w" -- "f16x2" with every layer behind the first one as Winograd F(2x2, 3x3): 16 instead of 36 matrix products per 2 x 2
        #               outputs (dpx_conv_wino_dev.h; the transformed weights are computed in float64, the transforms are fp32 additions);
        #               the same range rule (the trap watches the transformed inputs: |x| < 1.5e4 is always safe).  Under autograd the
        #               forward pass runs as "f16x2";
        #   "bf16"   -- plain bf16 operands, fp32 accumulation (bf16 training / inference mode, ~3e-3 relative).
        self.compute_mode = os.environ.get("DPX_FFDNET_MODE", "f16x2" if nc % 16 == 0 else "f32")
        # trainable weights: False = forward / backward-data on the split kernels as with frozen weights (weight gradients: the f32-input GEMM
        # on planar copies of their planes); True = everything on the f32-input kernels (the round-3 path, A/B)
        self.train_f32 = bool(os.environ.get("DPX_FFDNET_TRAIN_F32"))
        # what happens when a "f16x2" forward meets an operand outside the binary16 range (a checkpoint with a large dynamic range):
        # "bf16x3" -- the enclosing solve() / denoise() is re-run on the split-bf16 arithmetic and the network keeps that mode
        # (a RuntimeWarning says so); "raise" -- be.F16RangeError
        self.f16_fallback = os.environ.get("DPX_F16_FALLBACK", "bf16x3")
        # arithmetic of the backward pass on the split kernels (autograd through the network): "auto" -- split-f16 on gradients scaled by a
        # power of two (dpx_ffdnet_backward_bf16, mode 3: half the matrix work) when the forward pass runs split-f16, else split-bf16;
        # "f16x2" / "bf16x3" force one.  A scaled gradient that leaves the binary16 range on its way through the stack (2^12 of headroom)
        # makes that backward pass raise be.F16RangeError and the network fall back to "bf16x3" (be.note_f16_backward)
        self.backward_mode = os.environ.get("DPX_FFDNET_BWD_MODE", "auto")
        self._packed_bf16 = None

    @property
    def weights(self):
        return [self.ref_param(f"model.{2 * i}.weight") for i in range(self.nb)]

    @property
    def biases(self):
        return [self.ref_param(f"model.{2 * i}.bias") for i in range(self.nb)]

    def _weights_changed(self):
        super()._weights_changed()
        self._packed_bf16 = None
        self._packed_T_bf16 = None

    def packed_bf16(self, mode):
        """weights pre-split for the bf16 matrix cores (mode 6: three exact bf16 planes; mode 1: one rounded plane)"""
        dev = self.weights[0].device
        key = (self._weights_version(), str(dev), mode)
        if self._packed_bf16 is None or self._packed_bf16[0] != key:
            L = be.lib()
            blob = torch.empty(L.query("dpx_ffdnet_bf16_packed_bytes", self.in_nc, self.nc, self.nb), dtype=torch.uint8, device=dev)
            ws = [w.detach().float().contiguous() for w in self.weights]
            bs = [b.detach().float().contiguous() for b in self.biases]
            pw = (ctypes.c_void_p * self.nb)(*[w.data_ptr() for w in ws])
            pb = (ctypes.c_void_p * self.nb)(*[b.data_ptr() for b in bs])
            L.call("dpx_ffdnet_bf16_pack", be.ptr(blob), pw, pb, self.in_nc, self.nc, self.nb, mode, be.stream())
            self._packed_bf16 = (key, blob)
        return self._packed_bf16[1]

    @staticmethod
    def f16_overflowed(reset=True):
        """True if a "f16x2" forward pass since the last reset met an operand outside the binary16 range (its result is then invalid:
        use "bf16x3").  Synchronises the device."""
        return bool(be.lib().query("dpx_ffdnet_f16_overflow", int(bool(reset))))

    def load_reference_state_dict(self, sd):
        self.load_state_dict(sd, strict=True)
        return self

    def load_layers(self, layers):
        """layers: [(weight[co,ci,3,3], bias[co])] numpy/torch"""
        assert len(layers) == self.nb
        for (w, b), pw, pb in zip(layers, self.weights, self.biases):
            pw.data.copy_(torch.as_tensor(w))
            pb.data.copy_(torch.as_tensor(b))
        self._weights_changed()
        return self

    def reference_state_dict(self):
        return {k: v.detach().cpu() for k, v in self.state_dict().items()}

    def packed_T(self):
        """flipped / transposed weights of the backward-data convolutions (dpx_ffdnet_pack_transposed), cached per weight version"""
        dev = self.weights[0].device
        key = (self._weights_version(), str(dev))
        if getattr(self, "_packed_T", None) is None or self._packed_T[0] != key:
            L = be.lib()
            blob = torch.empty(max(L.query("dpx_ffdnet_packed_transposed_bytes", self.in_nc, self.nc, self.nb), 16), dtype=torch.uint8, device=dev)
            ws = [w.detach().float().contiguous() for w in self.weights]
            pw = (ctypes.c_void_p * self.nb)(*[w.data_ptr() for w in ws])
            L.call("dpx_ffdnet_pack_transposed", be.ptr(blob), pw, self.in_nc, self.nc, self.nb, be.stream())
            self._packed_T = (key, blob)
        return self._packed_T[1]

    def backward_mode_id(self):
        """`mode` of dpx_ffdnet_backward_bf16[_w] for this network's next backward pass: 3 (split-f16, scaled gradients) or 6 (split-bf16)"""
        bm = self.backward_mode
        if bm == "auto":
            bm = "f16x2" if self.compute_mode in be.F16_MODES else "bf16x3"
        if bm not in ("f16x2", "bf16x3"):
            raise ValueError(f"FFDNet.backward_mode: 'auto', 'f16x2' or 'bf16x3', got {self.backward_mode!r}")
        return 3 if bm == "f16x2" else 6

    def packed_T_bf16(self, mode=6):
        """the backward-data layers' weights for the split kernels (dpx_ffdnet_bf16_pack_transposed; mode 6: split-bf16 planes, 3: split-f16),
        cached per weight version and mode"""
        dev = self.weights[0].device
        key = (self._weights_version(), str(dev), mode)
        if getattr(self, "_packed_T_bf16", None) is None or self._packed_T_bf16[0] != key:
            L = be.lib()
            blob = torch.empty(L.query("dpx_ffdnet_bf16_packed_transposed_bytes", self.in_nc, self.nc, self.nb), dtype=torch.uint8, device=dev)
            ws = [w.detach().float().contiguous() for w in self.weights]
            pw = (ctypes.c_void_p * self.nb)(*[w.data_ptr() for w in ws])
            L.call("dpx_ffdnet_bf16_pack_transposed", be.ptr(blob), pw, self.in_nc, self.nc, self.nb, mode, be.stream())
            self._packed_T_bf16 = (key, blob)
        return self._packed_T_bf16[1]

    def packed(self):
        dev = self.weights[0].device
        if self._packed is not None and getattr(self, "_packed_version", None) != self._weights_version():
            self._packed = None                           # weights were updated in place (optimizer step)
        if self._packed is None or self._packed.device != dev:
            L = be.lib()
            blob = torch.empty(max(L.query("dpx_ffdnet_packed_bytes", self.in_nc, self.nc, self.nb), 16), dtype=torch.uint8, device=dev)
            ws = [w.detach().float().contiguous() for w in self.weights]
            bs = [b.detach().float().contiguous() for b in self.biases]
            pw = (ctypes.c_void_p * self.nb)(*[w.data_ptr() for w in ws])
            pb = (ctypes.c_void_p * self.nb)(*[b.data_ptr() for b in bs])
            L.call("dpx_ffdnet_pack", be.ptr(blob), pw, pb, self.in_nc, self.nc, self.nb, be.stream())
            self._packed = blob
            self._packed_version = self._weights_version()
        return self._packed

    def forward(self, x, sigma):
        be.require(x, what="FFDNet input")
        B, C, H, W = x.shape
        assert C == self.in_nc, f"FFDNet built for {self.in_nc} channels, got {C}"
        # (inference inside torch.no_grad() -- every plug-and-play iteration -- must not pay for a walk over the 30 parameters: the host is
        #  the critical path between the x-update and the first convolution launch of config 4's shard)
        train_w = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if torch.is_grad_enabled() and (train_w or x.requires_grad or (isinstance(sigma, torch.Tensor) and sigma.requires_grad)):
            sig_t = sigma if isinstance(sigma, torch.Tensor) else torch.as_tensor(sigma, dtype=torch.float32)
            sig_t = sig_t.to(device=x.device, dtype=torch.float32).reshape(-1)
            sig_t = sig_t.expand(B).contiguous() if sig_t.numel() == 1 else sig_t.contiguous()
            params = (self.weights + self.biases) if train_w else []
            if self.compute_mode in ("bf16x3", "f16x2", "f16x2w") and self.nc % 16 == 0 and not (train_w and self.train_f32):
                # forward and backward-data on the split kernels -- fp32-accurate, 2 - 3x the f32-input matrix instruction -- with frozen
                # weights (unrolled plug-and-play training of schedules / upstream parameters) and with trainable ones (the weight-gradient
                # GEMM is the f32-input kernel on planar copies of the split path's planes; `train_f32`: everything on the f32-input
                # kernels, the round-3 path, for A/B)
                self.last_train_path = "split"
                return _FFDNetSplitFn.apply(self, x, sig_t, *params)
            self.last_train_path = "f32"
            return _FFDNetFn.apply(self, x, sig_t, *params)
        sig = ops.as_batch_vec(sigma, B, x.device)
        L = be.lib()
        y = torch.empty_like(x)
        if self.compute_mode in ("bf16x3", "bf16", "f16x2", "f16x2w"):
            mode = be.FFDNET_MODES[self.compute_mode]
            if mode in (3, 4):
                be.note_f16_launch()
            ws = ops.workspace("ffdnet_bf16", L.query("dpx_ffdnet_bf16_ws_bytes", B, self.in_nc, self.nc, H, W), x.device)
            L.call("dpx_ffdnet_forward_bf16", be.ptr(x), be.ptr(y), be.ptr(sig), be.ptr(self.packed_bf16(mode)), self.in_nc, self.nc,
                   self.nb, mode, B, H, W, be.ptr(ws), be.stream())
            return y
        ws = ops.workspace("ffdnet", L.query("dpx_ffdnet_ws_bytes", B, self.in_nc, self.nc, H, W), x.device)
        L.call("dpx_ffdnet_forward", be.ptr(x), be.ptr(y), be.ptr(sig), be.ptr(self.packed()), self.in_nc, self.nc,
               self.nb, B, H, W, be.ptr(ws), be.stream())
        return y


class _FFDNetSplitFn(torch.autograd.Function):
    """FFDNet under autograd on the split kernels (frozen weights, or -- parameters passed -- trainable ones: the weight / bias
    gradients then come from dpx_ffdnet_backward_bf16_w): the forward pass keeps every layer's output (C8 layout),
    the backward pass is the same convolution kernel on flipped / transposed split weights with the ReLU masks in its epilogue
    (dpx_ffdnet_forward_bf16_save / dpx_ffdnet_backward_bf16).  The backward pass runs in `net.backward_mode`: split-bf16, or -- with a
    split-f16 forward -- split-f16 on gradients scaled by a power of two (gradients of a mean loss sit far below the binary16 range; the
    pass is linear in them); an operand outside the binary16 range is caught by the range trap in either direction."""

    @staticmethod
    def forward(ctx, net, x, sig, *params):
        B, C, H, W = x.shape
        L = be.lib()
        x = x.contiguous()
        y = torch.empty_like(x)
        mode = {"bf16x3": 6, "f16x2": 3, "f16x2w": 3}[net.compute_mode]
        if mode == 3:
            be.note_f16_launch()
        acts = torch.empty(L.query("dpx_ffdnet_bf16_acts_bytes", B, net.in_nc, net.nc, net.nb, H, W), dtype=torch.uint8, device=x.device)
        L.call("dpx_ffdnet_forward_bf16_save", be.ptr(x), be.ptr(y), be.ptr(sig), be.ptr(net.packed_bf16(mode)), net.in_nc, net.nc, net.nb, mode,
               B, H, W, be.ptr(acts), be.stream())
        ctx.net, ctx.shape = net, (B, C, H, W)
        ctx.save_for_backward(acts)
        return y

    @staticmethod
    def backward(ctx, gy):
        net = ctx.net
        B, C, H, W = ctx.shape
        (acts,) = ctx.saved_tensors
        L = be.lib()
        gy = gy.contiguous()
        gx = torch.empty_like(gy) if ctx.needs_input_grad[1] else None
        gs = torch.empty(B, dtype=torch.float32, device=gy.device) if ctx.needs_input_grad[2] else None
        nb = net.nb
        need = ctx.needs_input_grad[3:]
        bmode = net.backward_mode_id()
        if bmode == 3 and (any(need) or gx is not None or gs is not None):
            be.note_f16_backward(net)                      # (checked once, when this backward pass of the autograd engine ends)
        if any(need):                                      # weights / biases were passed and are trained: fill their gradients
            gws = [torch.empty_like(net.weights[i], dtype=torch.float32) if (need[i] or need[nb + i]) else None for i in range(nb)]
            gbs = [torch.empty_like(net.biases[i], dtype=torch.float32) if gws[i] is not None else None for i in range(nb)]
            pw = (ctypes.c_void_p * nb)(*[None if t is None else t.data_ptr() for t in gws])
            pb = (ctypes.c_void_p * nb)(*[None if t is None else t.data_ptr() for t in gbs])
            ws = ops.workspace("ffdnet_bf16_bwd_w", L.query("dpx_ffdnet_bf16_bwd_w_ws_bytes", B, net.in_nc, net.nc, H, W), gy.device)
            L.call("dpx_ffdnet_backward_bf16_w", be.ptr(gy), be.ptr(gx), be.ptr(gs), pw, pb, be.ptr(net.packed_T_bf16(bmode)), be.ptr(acts), net.in_nc,
                   net.nc, nb, bmode, B, H, W, be.ptr(ws), be.stream())
            return (None, gx, gs, *gws, *gbs)
        if gx is None and gs is None:
            return (None, None, None) + (None,) * len(need)
        ws = ops.workspace("ffdnet_bf16_bwd", L.query("dpx_ffdnet_bf16_bwd_ws_bytes", B, net.in_nc, net.nc, H, W), gy.device)
        L.call("dpx_ffdnet_backward_bf16", be.ptr(gy), be.ptr(gx), be.ptr(gs), be.ptr(net.packed_T_bf16(bmode)), be.ptr(acts), net.in_nc, net.nc, net.nb,
               bmode, B, H, W, be.ptr(ws), be.stream())
        return (None, gx, gs) + (None,) * len(need)


class _FFDNetFn(torch.autograd.Function):
    """FFDNet forward that keeps the layer outputs + hand-written backward-data pass (dpx_ffdnet_backward)"""

    @staticmethod
    def forward(ctx, net, x, sig, *params):
        B, C, H, W = x.shape
        L = be.lib()
        x = x.contiguous()
        y = torch.empty_like(x)
        acts = torch.empty(L.query("dpx_ffdnet_acts_bytes", B, net.in_nc, net.nc, net.nb, H, W), dtype=torch.uint8, device=x.device)
        L.call("dpx_ffdnet_forward_save", be.ptr(x), be.ptr(y), be.ptr(sig), be.ptr(net.packed()), net.in_nc, net.nc, net.nb,
               B, H, W, be.ptr(acts), be.stream())
        ctx.net, ctx.shape = net, (B, C, H, W)
        ctx.save_for_backward(acts)
        return y

    @staticmethod
    def backward(ctx, gy):
        net = ctx.net
        B, C, H, W = ctx.shape
        (acts,) = ctx.saved_tensors
        L = be.lib()
        gy = gy.contiguous()
        gx = torch.empty_like(gy) if ctx.needs_input_grad[1] else None
        gs = torch.empty(B, dtype=torch.float32, device=gy.device) if ctx.needs_input_grad[2] else None
        nb = net.nb
        gws, gbs, pw, pb = [], [], None, None
        if len(ctx.needs_input_grad) > 3:                 # weights / biases were passed: fill their gradients
            need = ctx.needs_input_grad[3:]
            gws = [torch.empty_like(net.weights[i], dtype=torch.float32) if (need[i] or need[nb + i]) else None for i in range(nb)]
            gbs = [torch.empty_like(net.biases[i], dtype=torch.float32) if gws[i] is not None else None for i in range(nb)]
            pw = (ctypes.c_void_p * nb)(*[None if t is None else t.data_ptr() for t in gws])
            pb = (ctypes.c_void_p * nb)(*[None if t is None else t.data_ptr() for t in gbs])
        if gx is None and gs is None and pw is None:
            return (None, None, None) + (None,) * (2 * nb if gws else 0)
        ws = ops.workspace("ffdnet_bwd", L.query("dpx_ffdnet_bwd_ws_bytes", B, net.in_nc, net.nc, H, W), gy.device)
        L.call("dpx_ffdnet_backward", be.ptr(gy), be.ptr(gx), be.ptr(gs), pw, pb, be.ptr(net.packed_T()), be.ptr(acts), net.in_nc,
               net.nc, nb, B, H, W, be.ptr(ws), be.stream())
        return (None, gx, gs, *gws, *gbs)


def _load_checkpoint(model, model_path):
    if model_path is None:
        return model
    if isinstance(model_path, (list, tuple)):
        return model.load_layers(model_path)
    if isinstance(model_path, dict):
        return model.load_reference_state_dict(model_path)
    return model.load_reference_state_dict(torch.load(model_path, map_location="cpu"))


class FFDNetDenoiser(Denoiser2D):
    """gray FFDNet (nc=64, nb=15) applied per band -- wrapper.py:25-35"""

    def __init__(self, model_path=None):
        super().__init__()
        self.model = _load_checkpoint(FFDNet(in_nc=1, out_nc=1, nc=64, nb=15, act_mode="R"), model_path)

    def _denoise(self, x, sigma):
        return self.model(x, sigma)


class FFDNetColorDenoiser(Denoiser):
    """color FFDNet (nc=96, nb=12) -- wrapper.py:38-48"""

    def __init__(self, model_path=None):
        super().__init__()
        self.model = _load_checkpoint(FFDNet(in_nc=3, out_nc=3, nc=96, nb=12, act_mode="R"), model_path)

    def _denoise(self, x, sigma):
        return self.model(x, sigma)


class Augment(nn.Module):
    """``deep_prior(..., x8=True)`` (reference denoisers/composite.py:6-46): call k of the wrapped denoiser sees the image
    under dihedral transform k mod 8 (rotations by 90 degrees / a flip of the rows) and its output is mapped back, so that
    consecutive iterations average out the denoiser's orientation bias.  Pure tensor plumbing around ``denoise``."""

    # mode -> (quarter turns, flip rows afterwards); the inverse of 3 is 5 and vice versa, the other modes undo themselves
    MODES = ((0, False), (1, True), (0, True), (3, False), (2, True), (1, False), (2, False), (3, True))
    INVERSE = (0, 1, 2, 5, 4, 3, 6, 7)

    def __init__(self, base_denoiser):
        super().__init__()
        self.base_denoiser = base_denoiser
        self.iter = 0

    def reset(self):
        self.iter = 0

    @classmethod
    def augment(cls, img, mode=0):
        turns, flip = cls.MODES[mode]
        out = img.rot90(turns, [2, 3]) if turns else img
        return out.flip([2]) if flip else out

    def denoise(self, x, sigma):
        mode = self.iter % 8
        y = self.base_denoiser.denoise(self.augment(x, mode).contiguous(), sigma)
        self.iter += 1
        return self.augment(y, self.INVERSE[mode]).contiguous()


class _ConvFn(torch.autograd.Function):
    """one dpx_conv2d layer (optional fused ReLU / residual) with a hand-written backward: the transposed convolution is the
    same kernel on flipped / transposed weights (packed once per weight version), the ReLU mask comes from the saved
    output, and -- when ``weight`` (the layer's weights in kernel form [cout, cin, taps], an autograd view of the
    parameter) is given -- the weight gradient is the pixels-as-K GEMM ``dpx_conv2d_wgrad`` on the saved input"""

    @staticmethod
    def forward(ctx, x, res, weight, fwd, bwd, relu, dilation=1, bias=None, slope=0.0):
        blob, cout, taps = fwd
        x = x.contiguous()
        if slope:                                            # LeakyReLU(slope) epilogue (U-Net); relu is implied
            y, relu = ops.conv2d_leaky(x, blob, cout, slope), True
        else:
            y = ops.conv2d(x, blob, cout, taps, relu=relu, res=None if res is None else res.contiguous(), dilation=dilation)
        ctx.bwd, ctx.relu, ctx.has_res, ctx.taps, ctx.dilation, ctx.slope = bwd, relu, res is not None, taps, dilation, slope
        train_w = (weight is not None and weight.requires_grad) or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(y if relu else x.new_empty(0), x if train_w else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gp = g
        y, x = ctx.saved_tensors
        if ctx.slope:
            gp = ops.leaky_relu_bwd(y, g, ctx.slope)                                                                    # g * (y > 0 ? 1 : slope)
        elif ctx.relu:
            gp, _ = ops.prox_bwd(be.PROX_NONNEG, y, g, torch.zeros((), device=g.device), 1.0, None, want_dlam=False)   # g * [y > 0]
        need = ctx.needs_input_grad                          # as many entries as apply() got arguments (6 .. 9)
        need_b = len(need) > 7 and need[7]
        gw = gb = None
        if need[2] or need_b:
            gw, gb = ops.conv2d_wgrad(gp, x, ctx.taps, dilation=ctx.dilation, want_bias=need_b)
            gw = gw if need[2] else None
        blob_t, cin, taps = ctx.bwd
        gx = None
        if ctx.needs_input_grad[0]:
            if gp.shape[1] % 2:
                gp = torch.cat([gp, torch.zeros_like(gp[:, :1])], dim=1).contiguous()
            gx = ops.conv2d(gp, blob_t, cin, taps, dilation=ctx.dilation)
        return (gx, (g if ctx.has_res else None), gw, None, None, None, None, gb, None)[:len(need)]


class _S2DFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.space_to_depth(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.depth_to_space(g.contiguous())


class _D2SFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return ops.depth_to_space(x.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.space_to_depth(g.contiguous())


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.lincomb([(1.0, a.contiguous()), (1.0, b.contiguous())])

    @staticmethod
    def backward(ctx, g):
        return g, g


class _SubFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return ops.lincomb([(1.0, a.contiguous()), (-1.0, b.contiguous())])

    @staticmethod
    def backward(ctx, g):
        return g, ops.lincomb([(-1.0, g.contiguous())])


class UNetRes(RefKeyed):
    """DRUNet body (reference models/network_unet.py:67-117): head conv, 3 x (nb ResBlocks + 2x2 stride-2 conv), nb ResBlocks,
    3 x (2x2 stride-2 transposed conv + nb ResBlocks), tail conv; no biases.  Every convolution runs on the fp32-MFMA
    kernel behind ``dpx_conv2d`` (ReLU / residual add fused into the epilogue); the strided / transposed 2x2 convolutions
    are 1x1 convolutions around ``dpx_space_to_depth`` / ``dpx_depth_to_space``.  Parameters keep the reference's
    state-dict names, so its checkpoints load unchanged.  Differentiable w.r.t. its input (every layer's backward is the same
    kernel on transposed weights) and, after ``requires_grad_(True)`` / ``deep_prior(trainable=True)``, w.r.t. its weights
    (``dpx_conv2d_wgrad``), see ``_ConvFn``."""

    def __init__(self, in_nc=1, out_nc=1, nc=(64, 128, 256, 512), nb=4, act_mode="R", downsample_mode="strideconv", upsample_mode="convtranspose"):
        super().__init__()
        assert act_mode == "R" and downsample_mode == "strideconv" and upsample_mode == "convtranspose", \
            "only the DRUNet configuration (ReLU, strideconv, convtranspose) is built for the HIP path"
        assert in_nc % 2 == 0, "dpx_conv2d needs an even number of input channels (DRUNet: image + sigma map = 4 or 2)"
        self.in_nc, self.out_nc, self.nc, self.nb = in_nc, out_nc, tuple(nc), nb
        shapes = {"m_head.weight": (nc[0], in_nc, 3, 3), "m_tail.weight": (out_nc, nc[0], 3, 3)}
        for lvl in range(3):
            for i in range(nb):
                for k in (0, 2):
                    shapes[f"m_down{lvl + 1}.{i}.res.{k}.weight"] = (nc[lvl], nc[lvl], 3, 3)
            shapes[f"m_down{lvl + 1}.{nb}.weight"] = (nc[lvl + 1], nc[lvl], 2, 2)
        for i in range(nb):
            for k in (0, 2):
                shapes[f"m_body.{i}.res.{k}.weight"] = (nc[3], nc[3], 3, 3)
        for lvl in (3, 2, 1):
            shapes[f"m_up{lvl}.0.weight"] = (nc[lvl], nc[lvl - 1], 2, 2)
            for i in range(nb):
                for k in (0, 2):
                    shapes[f"m_up{lvl}.{i + 1}.res.{k}.weight"] = (nc[lvl - 1], nc[lvl - 1], 3, 3)
        self._diff = False
        for key, shp in shapes.items():                      # the reference's state-dict keys (frozen until trained: requires_grad_())
            self.add_ref_param(key, shp, requires_grad=False)
        self._packed = None

    @property
    def _names(self):
        return self.ref_keys

    def _w(self, name):
        return self.ref_param(name).detach().float()

    def _check_version(self):
        """the packed blobs follow the parameters (an optimizer step bumps their version counters)"""
        ver = sum(self._weights_version())
        if ver != getattr(self, "_pack_version", None):
            self._packed, self._packed_T, self._pack_version = None, None, ver

    def _kernel_form(self, name):
        """the parameter as the [cout, cin, taps] tensor dpx_conv2d sees (an autograd view: gradients flow back to it)"""
        w = self.ref_param(name)
        if w.shape[-1] == 3:
            return w.reshape(w.shape[0], w.shape[1], 9)
        if name.startswith("m_down"):
            return w.reshape(w.shape[0], w.shape[1] * 4, 1)
        return w.permute(1, 2, 3, 0).reshape(w.shape[1] * 4, w.shape[0], 1)

    def packed(self):
        if self._packed is None:
            pk = {}
            for n in self._names:
                w = self._w(n)
                if w.shape[-1] == 3:
                    pk[n] = (ops.conv_pack(w.reshape(w.shape[0], w.shape[1], 9).contiguous(), None, 9), int(w.shape[0]), 9)
                elif n.startswith("m_down"):                 # Conv2d 2x2 / stride 2 = space_to_depth + 1x1 over (ci, dy, dx)
                    co, ci = int(w.shape[0]), int(w.shape[1])
                    pk[n] = (ops.conv_pack(w.reshape(co, ci * 4, 1).contiguous(), None, 1), co, 1)
                else:                                        # ConvTranspose2d [ci, co, 2, 2] = 1x1 to (co, dy, dx) + depth_to_space
                    ci, co = int(w.shape[0]), int(w.shape[1])
                    pk[n] = (ops.conv_pack(w.permute(1, 2, 3, 0).reshape(co * 4, ci, 1).contiguous(), None, 1), co * 4, 1)
            self._packed = pk
        return self._packed

    def packed_T(self):
        """weights of the backward-data convolutions: 3x3 flipped and channel-transposed; the 1x1 forms transposed"""
        if getattr(self, "_packed_T", None) is None:
            pk = {}
            for n in self._names:
                w = self._w(n)
                if w.shape[-1] == 3:
                    wt = w.flip(-1, -2).permute(1, 0, 2, 3).reshape(w.shape[1], w.shape[0], 9)
                    if wt.shape[1] % 2:                          # odd forward cout (tail conv): the gradient gets a zero channel
                        wt = torch.cat([wt, torch.zeros_like(wt[:, :1])], dim=1)
                    pk[n] = (ops.conv_pack(wt.contiguous(), None, 9), int(w.shape[1]), 9)
                elif n.startswith("m_down"):
                    co, ci = int(w.shape[0]), int(w.shape[1])
                    pk[n] = (ops.conv_pack(w.reshape(co, ci * 4).t().reshape(ci * 4, co, 1).contiguous(), None, 1), ci * 4, 1)
                else:
                    ci, co = int(w.shape[0]), int(w.shape[1])
                    pk[n] = (ops.conv_pack(w.permute(1, 2, 3, 0).reshape(co * 4, ci).t().reshape(ci, co * 4, 1).contiguous(), None, 1), ci, 1)
            self._packed_T = pk
        return self._packed_T

    def _conv(self, x, name, relu=False, res=None):
        if self._diff:
            w = self._kernel_form(name) if self.ref_param(name).requires_grad else None
            return _ConvFn.apply(x, res, w, self.packed()[name], self.packed_T()[name], relu)
        blob, cout, taps = self.packed()[name]
        return ops.conv2d(x, blob, cout, taps, relu=relu, res=res)

    def _res(self, x, prefix, first):
        for i in range(self.nb):
            t = self._conv(x, f"{prefix}.{first + i}.res.0.weight", relu=True)
            x = self._conv(t, f"{prefix}.{first + i}.res.2.weight", res=x)
        return x

    def forward(self, x0):
        be.require(x0, what="UNetRes input")
        self._check_version()
        self._diff = torch.is_grad_enabled() and (x0.requires_grad or any(p.requires_grad for p in self.parameters()))
        if self._diff:
            s2d, d2s, add = _S2DFn.apply, _D2SFn.apply, _AddFn.apply
        else:
            s2d, d2s, add = ops.space_to_depth, ops.depth_to_space, lambda a, b: ops.lincomb([(1.0, a), (1.0, b)])
        nb = self.nb
        x1 = self._conv(x0.contiguous(), "m_head.weight")
        x2 = self._conv(s2d(self._res(x1, "m_down1", 0)), f"m_down1.{nb}.weight")
        x3 = self._conv(s2d(self._res(x2, "m_down2", 0)), f"m_down2.{nb}.weight")
        x4 = self._conv(s2d(self._res(x3, "m_down3", 0)), f"m_down3.{nb}.weight")
        x = self._res(x4, "m_body", 0)
        x = self._res(d2s(self._conv(add(x, x4), "m_up3.0.weight")), "m_up3", 1)
        x = self._res(d2s(self._conv(add(x, x3), "m_up2.0.weight")), "m_up2", 1)
        x = self._res(d2s(self._conv(add(x, x2), "m_up1.0.weight")), "m_up1", 1)
        return self._conv(add(x, x1), "m_tail.weight")


class DRUNetDenoiser(Denoiser):
    """reference denoisers/wrapper.py:89-146: sigma map as an extra channel; images up to 256x256 are replicate-padded to a
    multiple of 16 and denoised in one pass, larger ones as four overlapping quadrants (recursively).  The tensor
    assembly (concat / pad / slices) is PyTorch memory plumbing, every convolution is HIP."""

    def __init__(self, n_channels, model_path=None):
        super().__init__()
        self.model = UNetRes(in_nc=n_channels + 1, out_nc=n_channels, nc=(64, 128, 256, 512), nb=4)
        if model_path is not None:
            sd = model_path if isinstance(model_path, dict) else torch.load(model_path, map_location="cpu")
            self.model.load_state_dict(sd, strict=True)

    def _denoise(self, x, sigma):
        if sigma.shape[0] != x.shape[0]:
            sigma = sigma.repeat(x.shape[0], 1, 1, 1)
        L = torch.cat((x, sigma.to(x.device, x.dtype).repeat(1, 1, x.shape[2], x.shape[3])), dim=1)
        return self._run(L)

    # Tiling policy of wrapper.py:112-146, expressed per axis: an axis of length n is covered by two overlapping windows of
    # k = (n // 2 // refield + 1) * refield samples, [0, k) and [n - k, n); the first contributes the output samples [0, n // 2),
    # the second [n // 2, n).  A plane is the product of its two axes' windows (four overlapping tiles); tiles are denoised
    # directly while the plane has at most 4 * min_size^2 pixels and recursively otherwise.  Planes of at most min_size^2
    # pixels are replicate-padded to a multiple of `modulo` (the three 2x down-samplings) and denoised in one pass.
    @staticmethod
    def _axis_windows(n, refield):
        k = (n // 2 // refield + 1) * refield
        half = n // 2
        return ((slice(0, k), slice(0, half), slice(0, half)),                 # (input window, output range, range inside the tile)
                (slice(n - k, n), slice(half, n), slice(k - (n - half), k)))

    def _run(self, L, refield=32, min_size=256, modulo=16):
        h, w = L.shape[-2:]
        if h * w <= min_size * min_size:
            pad_h, pad_w = -h % modulo, -w % modulo
            padded = torch.nn.functional.pad(L, (0, pad_w, 0, pad_h), mode="replicate") if (pad_h or pad_w) else L
            return self.model(padded.contiguous())[..., :h, :w]
        direct = h * w <= 4 * min_size * min_size
        out = None
        for win_h, dst_h, src_h in self._axis_windows(h, refield):
            for win_w, dst_w, src_w in self._axis_windows(w, refield):
                tile = L[..., win_h, win_w].contiguous()
                den = self.model(tile) if direct else self._run(tile, refield, min_size, modulo)
                if out is None:
                    out = torch.empty(den.shape[0], den.shape[1], h, w, dtype=L.dtype, device=L.device)
                out[..., dst_h, dst_w] = den[..., src_h, src_w]
        return out


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        ctx.save_for_backward(x)
        return ops.maxpool2(x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return ops.maxpool2_bwd(x, g.contiguous())


class _ConcatUpFn(torch.autograd.Function):
    """cat([skip, zero-pad(bilinear x2 (low))], dim=1) as one pair of kernels, and its adjoint"""

    @staticmethod
    def forward(ctx, skip, low):
        ctx.c2, ctx.low_hw = int(skip.shape[1]), (int(low.shape[2]), int(low.shape[3]))
        return ops.concat_skip_upsampled(skip.contiguous(), low.contiguous())

    @staticmethod
    def backward(ctx, g):
        return ops.concat_skip_upsampled_bwd(g.contiguous(), ctx.c2, ctx.low_hw)


class UNet(RefKeyed):
    """The U-Net denoiser body (reference models/unet/unet.py:34-135): inc / down1..4 / up1..4 are ConvBlocks of three biased 3x3
    convolutions each followed by LeakyReLU(0.2) -- one ``dpx_conv2d_leaky`` launch per layer on the fp32 matrix-core kernel --
    with MaxPool2d(2) on the way down (``dpx_maxpool2``) and, on the way up, bilinear x2 up-sampling (align_corners=True),
    zero-padding to the skip tensor's size and channel concatenation fused into ``dpx_upsample2_into`` + ``dpx_copy_channels``;
    outc is a 1x1 convolution and the network returns ``input[:, :out_channels] + residual``.  Reference state-dict keys.
    Differentiable w.r.t. its input and (``requires_grad_(True)``) its weights / biases: every layer's backward is the same
    kernel on transposed weights (``_ConvFn``), the pooling / up-sampling adjoints are ``dpx_maxpool2_bwd`` / ``dpx_upsample2_into_bwd``."""

    WIDTHS = (32, 64, 128, 256, 512)
    SLOPE = 0.2

    def __init__(self, in_channels, out_channels, requires_grad=False):
        super().__init__()
        assert in_channels % 2 == 0, "dpx_conv2d needs an even number of input channels (UNetDenoiser: image + noise map = 2)"
        self.in_channels, self.out_channels = in_channels, out_channels
        w = self.WIDTHS
        self.blocks = {"inc.conv": (in_channels, w[0])}
        for k in range(4):
            self.blocks[f"down{k + 1}.mpconv.1"] = (w[k], w[k + 1])
        for k in range(4):
            self.blocks[f"up{k + 1}.conv"] = (w[4 - k] + w[3 - k], w[3 - k])
        for prefix, (ci, co) in self.blocks.items():
            for i in range(3):
                self.add_ref_param(f"{prefix}.conv-{i}.conv2d.weight", (co, ci if i == 0 else co, 3, 3), requires_grad)
                self.add_ref_param(f"{prefix}.conv-{i}.conv2d.bias", (co,), requires_grad)
        self.add_ref_param("outc.conv.weight", (out_channels, w[0], 1, 1), requires_grad)
        self.add_ref_param("outc.conv.bias", (out_channels,), requires_grad)
        self._packed = self._packed_T = None

    def _layers(self):
        """(weight key, bias key) of every convolution, 3x3 layers first"""
        return [(f"{p}.conv-{i}.conv2d.weight", f"{p}.conv-{i}.conv2d.bias") for p in self.blocks for i in range(3)] + [("outc.conv.weight", "outc.conv.bias")]

    def _check_version(self):
        ver = sum(self._weights_version())
        if ver != getattr(self, "_pack_version", None):
            self._packed, self._packed_T, self._pack_version = None, None, ver

    def packed(self):
        if self._packed is None:
            pk = {}
            for wk, bk in self._layers():
                w, b = self.ref_param(wk).detach().float(), self.ref_param(bk).detach().float().contiguous()
                taps = int(w.shape[-1]) ** 2
                pk[wk] = (ops.conv_pack(w.reshape(w.shape[0], w.shape[1], taps).contiguous(), b, taps), int(w.shape[0]), taps)
            self._packed = pk
        return self._packed

    def packed_T(self):
        """backward-data layers: 3x3 weights flipped and channel-transposed, the 1x1 layer transposed; no bias; an odd number of
        gradient channels (the 1-channel output) is padded with a zero channel"""
        if self._packed_T is None:
            pk = {}
            for wk, _ in self._layers():
                w = self.ref_param(wk).detach().float()
                taps = int(w.shape[-1]) ** 2
                wt = (w.flip(-1, -2) if taps == 9 else w).permute(1, 0, 2, 3).reshape(w.shape[1], w.shape[0], taps)
                if wt.shape[1] % 2:
                    wt = torch.cat([wt, torch.zeros_like(wt[:, :1])], dim=1)
                pk[wk] = (ops.conv_pack(wt.contiguous(), None, taps), int(w.shape[1]), taps)
            self._packed_T = pk
        return self._packed_T

    def _conv(self, x, wk, bk, slope):
        blob, cout, taps = self.packed()[wk]
        if not self._diff:
            return ops.conv2d_leaky(x, blob, cout, slope) if slope else ops.conv2d(x, blob, cout, taps)
        w, b = self.ref_param(wk), self.ref_param(bk)
        wkf = w.reshape(w.shape[0], w.shape[1], taps) if w.requires_grad else None       # kernel form, an autograd view of the parameter
        return _ConvFn.apply(x, None, wkf, (blob, cout, taps), self.packed_T()[wk], bool(slope), 1, b if b.requires_grad else None, slope)

    def _block(self, x, prefix):
        for i in range(3):
            x = self._conv(x, f"{prefix}.conv-{i}.conv2d.weight", f"{prefix}.conv-{i}.conv2d.bias", self.SLOPE)
        return x

    def forward(self, x):
        be.require(x, what="UNet input")
        assert x.shape[1] == self.in_channels, f"UNet built for {self.in_channels} input channels, got {x.shape[1]}"
        assert min(x.shape[-2:]) >= 16, "four 2x poolings need planes of at least 16 x 16"
        self._check_version()
        self._diff = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self.parameters()))
        pool = _MaxPoolFn.apply if self._diff else ops.maxpool2
        cat_up = _ConcatUpFn.apply if self._diff else ops.concat_skip_upsampled
        x = x.contiguous()
        skips = [self._block(x, "inc.conv")]
        for k in range(4):
            skips.append(self._block(pool(skips[-1]), f"down{k + 1}.mpconv.1"))
        t = skips[4]
        for k in range(4):
            t = self._block(cat_up(skips[3 - k], t), f"up{k + 1}.conv")
        res = self._conv(t, "outc.conv.weight", "outc.conv.bias", 0.0)
        head = x[:, :self.out_channels].contiguous()
        if self._diff:
            return _AddFn.apply(head, res)
        return ops.lincomb([(1.0, head), (1.0, res)])


class UNetDenoiser(Denoiser2D):
    """reference denoisers/wrapper.py:206-221: single-band U-Net applied band by band; the noise level enters as a constant second
    channel; the output is clamped to [0, 1]"""

    def __init__(self, model_path=None):
        super().__init__()
        self.model = UNet(2, 1)
        if model_path is not None:
            sd = model_path if isinstance(model_path, dict) else torch.load(model_path, map_location="cpu")
            self.model.load_state_dict(sd, strict=True)

    def _denoise(self, x, sigma):
        noise_map = torch.ones_like(x) * sigma.to(x.device, x.dtype).view(-1, 1, 1, 1)
        out = self.model(torch.cat([x, noise_map], dim=1).contiguous())
        return torch.clamp(out, 0, 1)


class IRCNN(RefKeyed):
    """IRCNN body (reference models/network_dncnn.py:74-113): x - net(x) with seven biased 3x3 convolutions of dilation
    1,2,3,4,3,2,1 -- ``dpx_conv2d`` with the dilation-templated staging tile.  Reference state-dict names.  Differentiable
    w.r.t. its input and (``requires_grad_(True)``) its weights / biases through ``_ConvFn``."""

    DIL = (1, 2, 3, 4, 3, 2, 1)

    def __init__(self, in_nc=1, out_nc=1, nc=64):
        super().__init__()
        chans = [in_nc] + [nc] * 6 + [out_nc]
        self.in_nc, self.out_nc, self.nc = in_nc, out_nc, nc
        for i, (ci, co) in enumerate(zip(chans[:-1], chans[1:])):
            self.add_ref_param(f"model.{2 * i}.weight", (co, ci, 3, 3), requires_grad=False)
            self.add_ref_param(f"model.{2 * i}.bias", (co,), requires_grad=False)
        self._packed = None

    @property
    def weights(self):
        return [self.ref_param(f"model.{2 * i}.weight") for i in range(7)]

    @property
    def biases(self):
        return [self.ref_param(f"model.{2 * i}.bias") for i in range(7)]

    def packed(self):
        if self._packed is None:
            pk = []
            for w, b in zip(self.weights, self.biases):
                w = w.detach().float()
                if w.shape[1] % 2:                           # odd input channel count (gray input): zero weights for the pad channel
                    w = torch.cat([w, torch.zeros_like(w[:, :1])], dim=1)
                pk.append((ops.conv_pack(w.reshape(w.shape[0], w.shape[1], 9).contiguous(), b.detach().float().contiguous(), 9), int(w.shape[0])))
            self._packed = pk
        return self._packed

    def packed_T(self):
        """backward-data layers: weights flipped and channel-transposed (same dilation), no bias; the transposed layer's input
        (the forward layer's output gradient) is padded to an even channel count"""
        if getattr(self, "_packed_T", None) is None:
            pk = []
            for w in self.weights:
                w = w.detach().float()
                cin_pad = w.shape[1] + (w.shape[1] % 2)                   # the forward layer saw a zero pad channel
                wt = w.flip(-1, -2).permute(1, 0, 2, 3).reshape(w.shape[1], w.shape[0], 9)
                if wt.shape[0] != cin_pad:
                    wt = torch.cat([wt, torch.zeros_like(wt[:1])], dim=0)
                if wt.shape[1] % 2:
                    wt = torch.cat([wt, torch.zeros_like(wt[:, :1])], dim=1)
                pk.append((ops.conv_pack(wt.contiguous(), None, 9), int(cin_pad)))
            self._packed_T = pk
        return self._packed_T

    def _check_version(self):
        ver = sum(p._version for p in self.parameters())
        if ver != getattr(self, "_pack_version", None):
            self._packed, self._packed_T, self._pack_version = None, None, ver

    def forward(self, x):
        be.require(x, what="IRCNN input")
        self._check_version()
        train_w = any(p.requires_grad for p in self.parameters())
        diff = torch.is_grad_enabled() and (x.requires_grad or train_w)
        n = x
        if n.shape[1] % 2:
            n = torch.cat([n, torch.zeros_like(n[:, :1])], dim=1).contiguous()
        for i, ((blob, cout), d) in enumerate(zip(self.packed(), self.DIL)):
            if diff:
                w, b = self.weights[i], self.biases[i]
                wk = None
                if w.requires_grad:                          # kernel form [cout, cin(+pad), 9] as an autograd view of the parameter
                    wk = w.reshape(w.shape[0], w.shape[1], 9)
                    if w.shape[1] % 2:
                        wk = torch.cat([wk, torch.zeros_like(wk[:, :1])], dim=1)
                blob_t, cin_t = self.packed_T()[i]
                n = _ConvFn.apply(n, None, wk, (blob, cout, 9), (blob_t, cin_t, 9), i < 6, d, b if b.requires_grad else None)
            else:
                n = ops.conv2d(n, blob, cout, 9, relu=i < 6, dilation=d)
            if i < 6 and n.shape[1] % 2:
                n = torch.cat([n, torch.zeros_like(n[:, :1])], dim=1).contiguous()
        if diff:
            return _SubFn.apply(x, n)
        return ops.lincomb([(1.0, x.contiguous()), (-1.0, n)])


class IRCNNDenoiser(Denoiser2D):
    """reference denoisers/wrapper.py:68-86: 25 IRCNN models, one per noise-level bin ceil(sigma * 255 / 2) - 1; applied band
    by band.  ``model_path``: a path to (or the dict of) the 25 state dicts keyed "0".."24"."""

    def __init__(self, n_channels, model_path):
        super().__init__()
        self.model = IRCNN(in_nc=n_channels, out_nc=n_channels, nc=64)
        self.model25 = model_path if isinstance(model_path, dict) else torch.load(model_path, map_location="cpu")
        self.former_idx = None

    def _denoise(self, x, sigma):
        current_idx = int(np.ceil(sigma.reshape(-1)[:1].cpu().numpy() * 255. / 2.)[0] - 1)      # float32 arithmetic, like the reference
        if current_idx != self.former_idx:
            dev = x.device
            self.model.load_state_dict(self.model25[str(current_idx)], strict=True)
            self.model = self.model.to(dev)
        self.former_idx = current_idx
        return self.model(x)
