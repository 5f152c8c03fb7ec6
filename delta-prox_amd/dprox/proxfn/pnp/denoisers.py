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
        return self._denoise(input, sigma)

    def _denoise(self, x, sigma):
        raise NotImplementedError


class Denoiser2D(Denoiser):
    """applies a single-channel denoiser band by band"""

    def denoise(self, input: torch.Tensor, sigma: torch.Tensor):
        sigma = sigma.view(-1, 1, 1, 1)
        return torch.cat([self._denoise(band.contiguous(), sigma) for band in input.split(1, dim=1)], dim=1)


class FFDNet(nn.Module):
    """FFDNet(in_nc, out_nc, nc, nb): conv3x3(in_nc*4+1 -> nc) + (nb-2) x conv3x3(nc -> nc) + conv3x3(nc -> out_nc*4)"""

    def __init__(self, in_nc=1, out_nc=1, nc=64, nb=15, act_mode="R"):
        super().__init__()
        assert "R" in act_mode, "only the ReLU FFDNet variants are built for the HIP path"
        assert in_nc == out_nc
        self.in_nc, self.out_nc, self.nc, self.nb = in_nc, out_nc, nc, nb
        chans = [in_nc * 4 + 1] + [nc] * (nb - 1) + [out_nc * 4]
        self.weights = nn.ParameterList([nn.Parameter(torch.zeros(co, ci, 3, 3)) for ci, co in zip(chans[:-1], chans[1:])])
        self.biases = nn.ParameterList([nn.Parameter(torch.zeros(co)) for co in chans[1:]])
        self._packed = None

    # reference checkpoint layout: Conv2d modules at the even indices of `model`
    def load_reference_state_dict(self, sd):
        for i in range(self.nb):
            self.weights[i].data.copy_(torch.as_tensor(sd[f"model.{2 * i}.weight"]))
            self.biases[i].data.copy_(torch.as_tensor(sd[f"model.{2 * i}.bias"]))
        self._packed = None
        return self

    def load_layers(self, layers):
        """layers: [(weight[co,ci,3,3], bias[co])] numpy/torch"""
        assert len(layers) == self.nb
        for i, (w, b) in enumerate(layers):
            self.weights[i].data.copy_(torch.as_tensor(w))
            self.biases[i].data.copy_(torch.as_tensor(b))
        self._packed = None
        return self

    def reference_state_dict(self):
        sd = {}
        for i in range(self.nb):
            sd[f"model.{2 * i}.weight"] = self.weights[i].detach().cpu()
            sd[f"model.{2 * i}.bias"] = self.biases[i].detach().cpu()
        return sd

    def _apply(self, fn, *a, **k):
        self._packed = None
        return super()._apply(fn, *a, **k)

    def _weights_version(self):
        return tuple(p._version for p in list(self.weights) + list(self.biases))

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
        train_w = any(p.requires_grad for p in self.parameters())
        if torch.is_grad_enabled() and (train_w or x.requires_grad or (isinstance(sigma, torch.Tensor) and sigma.requires_grad)):
            sig_t = sigma if isinstance(sigma, torch.Tensor) else torch.as_tensor(sigma, dtype=torch.float32)
            sig_t = sig_t.to(device=x.device, dtype=torch.float32).reshape(-1)
            sig_t = sig_t.expand(B).contiguous() if sig_t.numel() == 1 else sig_t.contiguous()
            params = (list(self.weights) + list(self.biases)) if train_w else []
            return _FFDNetFn.apply(self, x, sig_t, *params)
        sig = ops.as_batch_vec(sigma, B, x.device)
        L = be.lib()
        y = torch.empty_like(x)
        ws = ops.workspace("ffdnet", L.query("dpx_ffdnet_ws_bytes", B, self.in_nc, self.nc, H, W), x.device)
        L.call("dpx_ffdnet_forward", be.ptr(x), be.ptr(y), be.ptr(sig), be.ptr(self.packed()), self.in_nc, self.nc,
               self.nb, B, H, W, be.ptr(ws), be.stream())
        return y


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
    def forward(ctx, x, res, weight, fwd, bwd, relu, dilation=1, bias=None):
        blob, cout, taps = fwd
        x = x.contiguous()
        y = ops.conv2d(x, blob, cout, taps, relu=relu, res=None if res is None else res.contiguous(), dilation=dilation)
        ctx.bwd, ctx.relu, ctx.has_res, ctx.taps, ctx.dilation = bwd, relu, res is not None, taps, dilation
        train_w = (weight is not None and weight.requires_grad) or (bias is not None and bias.requires_grad)
        ctx.save_for_backward(y if relu else x.new_empty(0), x if train_w else x.new_empty(0))
        return y

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gp = g
        y, x = ctx.saved_tensors
        if ctx.relu:
            gp, _ = ops.prox_bwd(be.PROX_NONNEG, y, g, torch.zeros((), device=g.device), 1.0, None, want_dlam=False)   # g * [y > 0]
        need = ctx.needs_input_grad                          # as many entries as apply() got arguments (6 or 8)
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
        return (gx, (g if ctx.has_res else None), gw, None, None, None, None, gb)[:len(need)]


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


class UNetRes(nn.Module):
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
        self._names = list(shapes)
        self._diff = False
        self.params = nn.ParameterDict({n.replace(".", "/"): nn.Parameter(torch.zeros(*shp), requires_grad=False) for n, shp in shapes.items()})
        self._packed = None

    def load_state_dict(self, sd, strict=True):              # reference key names
        missing = [n for n in self._names if n not in sd]
        extra = [k for k in sd if k not in self._names]
        if strict and (missing or extra):
            raise RuntimeError(f"UNetRes.load_state_dict: missing {missing[:3]}..., unexpected {extra[:3]}...")
        for n in self._names:
            if n in sd:
                self.params[n.replace(".", "/")].data.copy_(torch.as_tensor(sd[n]))
        self._packed = None
        self._packed_T = None
        return self

    def state_dict(self, *a, **k):
        return {n: self.params[n.replace(".", "/")].detach().cpu() for n in self._names}

    def _apply(self, fn, *a, **k):
        self._packed = None
        self._packed_T = None
        return super()._apply(fn, *a, **k)

    def _w(self, name):
        return self.params[name.replace(".", "/")].detach().float()

    def _check_version(self):
        """the packed blobs follow the parameters (an optimizer step bumps their version counters)"""
        ver = sum(p._version for p in self.params.values())
        if ver != getattr(self, "_pack_version", None):
            self._packed, self._packed_T, self._pack_version = None, None, ver

    def _kernel_form(self, name):
        """the parameter as the [cout, cin, taps] tensor dpx_conv2d sees (an autograd view: gradients flow back to it)"""
        w = self.params[name.replace(".", "/")]
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
            w = self._kernel_form(name) if self.params[name.replace(".", "/")].requires_grad else None
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
        self._diff = torch.is_grad_enabled() and (x0.requires_grad or any(p.requires_grad for p in self.params.values()))
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

    def _run(self, L, refield=32, min_size=256, modulo=16):
        h, w = L.shape[-2:]
        if h * w <= min_size ** 2:
            Lp = torch.nn.functional.pad(L, (0, int(np.ceil(w / modulo) * modulo - w), 0, int(np.ceil(h / modulo) * modulo - h)), mode="replicate")
            return self.model(Lp.contiguous())[..., :h, :w]
        top, bottom = slice(0, (h // 2 // refield + 1) * refield), slice(h - (h // 2 // refield + 1) * refield, h)
        left, right = slice(0, (w // 2 // refield + 1) * refield), slice(w - (w // 2 // refield + 1) * refield, w)
        Ls = [L[..., top, left], L[..., top, right], L[..., bottom, left], L[..., bottom, right]]
        if h * w <= 4 * (min_size ** 2):
            Es = [self.model(q.contiguous()) for q in Ls]
        else:
            Es = [self._run(q, refield, min_size, modulo) for q in Ls]
        b, c = Es[0].shape[:2]
        E = torch.zeros(b, c, h, w, dtype=L.dtype, device=L.device)
        E[..., :h // 2, :w // 2] = Es[0][..., :h // 2, :w // 2]
        E[..., :h // 2, w // 2:] = Es[1][..., :h // 2, (-w + w // 2):]
        E[..., h // 2:, :w // 2] = Es[2][..., (-h + h // 2):, :w // 2]
        E[..., h // 2:, w // 2:] = Es[3][..., (-h + h // 2):, (-w + w // 2):]
        return E


class IRCNN(nn.Module):
    """IRCNN body (reference models/network_dncnn.py:74-113): x - net(x) with seven biased 3x3 convolutions of dilation
    1,2,3,4,3,2,1 -- ``dpx_conv2d`` with the dilation-templated staging tile.  Reference state-dict names.  Differentiable
    w.r.t. its input and (``requires_grad_(True)``) its weights / biases through ``_ConvFn``."""

    DIL = (1, 2, 3, 4, 3, 2, 1)

    def __init__(self, in_nc=1, out_nc=1, nc=64):
        super().__init__()
        chans = [in_nc] + [nc] * 6 + [out_nc]
        self.in_nc, self.out_nc, self.nc = in_nc, out_nc, nc
        self.weights = nn.ParameterList([nn.Parameter(torch.zeros(co, ci, 3, 3), requires_grad=False) for ci, co in zip(chans[:-1], chans[1:])])
        self.biases = nn.ParameterList([nn.Parameter(torch.zeros(co), requires_grad=False) for co in chans[1:]])
        self._packed = None

    def load_state_dict(self, sd, strict=True):
        for i in range(7):
            self.weights[i].data.copy_(torch.as_tensor(sd[f"model.{2 * i}.weight"]))
            self.biases[i].data.copy_(torch.as_tensor(sd[f"model.{2 * i}.bias"]))
        self._packed = self._packed_T = None
        return self

    def _apply(self, fn, *a, **k):
        self._packed = self._packed_T = None
        return super()._apply(fn, *a, **k)

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
