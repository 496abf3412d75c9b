"""Generator backbones of the multi-prompt configs.

Generator3D's synthesis network runs on the HIP path (`backend = "hip"`, the default; csrc/conv3d.hip): every 3x3x3 modulated convolution
— forward, input gradient, weight gradient — is a split-fp16 MFMA kernel (each fp32 operand = two fp16 planes, three products, fp32
accumulation), the layer tail (noise + bias + leaky-ReLU + clamp) rides in the convolution's epilogue or in the trilinear upsampling
kernel, and the volumes stay channel-last [N, D, H, W, C] from the constant input to the voxel sampler that consumes the result.  What
remains library code are the per-sample style / demodulation arithmetic on the weights (a few MB), the mapping network and the
1x1x1 toRGB projections (plain matrix products on the channel-last view).  `backend = "library"` is the torch-op restatement (depth-sliced
conv2d through MIOpen): the A/B arm of tools/ and what the CPU golden test runs; it is never selected silently.

  Generator3D            custom/amortized/extern/stylegan_3dconv_modules.py:85-344   (StyleGAN2-style 3-D synthesis: mapping
                         network -> modulated 3x3x3 convolutions 4^3 ... 128^3, trilinear upsampling, skip "toRGB" volumes)
  TriplaneTransformer    custom/amortized/extern/triplane_transformer_modules.py:9-187 (LRM-style transformer over 3 x 32^2
                         learned tokens with text cross-attention -> ConvTranspose2d -> [N, 3, 32, 64, 64] planes)
Parameter names and shapes are the reference's, so its checkpoints load with load_state_dict.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib


# ---- StyleGAN-3D --------------------------------------------------------------------------------------------------------
class FullyConnectedLayer(nn.Module):
    """equalised-learning-rate linear layer (:36-53): y = act(x (W g_w)^T + b g_b) * act_gain"""

    def __init__(self, in_features, out_features, bias=True, activation="linear", lr_multiplier=1, bias_init=0):
        super().__init__()
        self.lrelu = activation == "lrelu"
        self.weight = nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = nn.Parameter(torch.full([out_features], float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        b = self.bias * self.bias_gain if self.bias_gain != 1 else self.bias
        y = torch.addmm(b.unsqueeze(0), x, (self.weight * self.weight_gain).t())
        return F.leaky_relu(y, 0.2) * math.sqrt(2) if self.lrelu else y


def _conv3d_depth_sliced(x, w, groups):
    """3x3x3 'same' convolution of x [1, G*Cin, D, H, W] with w [G*Cout, Cin, 3, 3, 3] as three 2-D convolutions over the depth
    slices (depth becomes the batch axis): y[:, :, d] = sum_kd conv2d(x[:, :, d + kd - 1], w[:, :, kd]).  Same arithmetic as
    F.conv3d up to fp32 summation order; MIOpen's fp32 conv3d backward is 18x slower than its conv2d backward on MI355X
    (64->64 at 128^3: 342 ms vs 19 ms forward+backward), and this generator is trained every step."""
    D = x.shape[2]
    xp = F.pad(x, (0, 0, 0, 0, 1, 1))
    y = None
    for kd in range(3):
        t = F.conv2d(xp[0, :, kd:kd + D].permute(1, 0, 2, 3), w[:, :, kd], padding=1, groups=groups)   # [D, G*Cout, H, W]
        y = t if y is None else y + t
    return y.permute(1, 0, 2, 3).unsqueeze(0)


def modulated_conv3d(x, weight, styles, padding=0, demodulate=True):
    """per-sample modulated (and demodulated) convolution as ONE grouped convolution (:64-82)"""
    n = x.shape[0]
    cout, cin = weight.shape[:2]
    w = weight.unsqueeze(0) * styles.reshape(n, 1, cin, 1, 1, 1)
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4, 5]) + 1e-8).rsqrt().reshape(n, cout, 1, 1, 1, 1)
    xg, wg = x.reshape(1, n * cin, *x.shape[2:]), w.reshape(n * cout, cin, *weight.shape[2:])
    if tuple(weight.shape[2:]) == (3, 3, 3) and padding == 1:
        y = _conv3d_depth_sliced(xg, wg, n)
    elif tuple(weight.shape[2:]) == (1, 1, 1) and padding == 0:      # toRGB: a per-sample matrix product (GEMM forward and backward)
        return torch.bmm(w.reshape(n, cout, cin), x.reshape(n, cin, -1)).reshape(n, cout, *x.shape[2:])
    else:
        y = F.conv3d(xg, wg, padding=padding, groups=n)
    return y.reshape(n, cout, *y.shape[2:])


_UP_MATS = {}


def _upsample_matrix(r: int, device, dtype):
    """[2r, r] matrix of 1-D linear interpolation with align_corners=True (source coordinate a * (r-1) / (2r-1))"""
    key = (r, str(device), dtype)
    if key not in _UP_MATS:
        src = torch.arange(2 * r, dtype=torch.float64) * ((r - 1) / (2 * r - 1))
        i0 = src.floor().clamp(max=r - 1).long()
        i1 = (i0 + 1).clamp(max=r - 1)
        f = (src - i0).to(torch.float64)
        m = torch.zeros(2 * r, r, dtype=torch.float64)
        m.scatter_add_(1, i0[:, None], (1 - f)[:, None])
        m.scatter_add_(1, i1[:, None], f[:, None])
        _UP_MATS[key] = m.to(device=device, dtype=dtype)
    return _UP_MATS[key]


def _upsample2(x):
    """F.interpolate(x, scale_factor=2, mode='trilinear', align_corners=True) (SmoothUpsample, :56-62) as three separable 1-D
    interpolations written as matrix products: the backward pass is three GEMMs instead of the atomic scatter of
    upsample_trilinear3d_backward (66 ms per step at 128^3 on MI355X)."""
    if x.shape[2] != x.shape[3] or x.shape[3] != x.shape[4]:
        return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)
    m = _upsample_matrix(x.shape[2], x.device, x.dtype)
    x = torch.matmul(x, m.t())                                  # W axis:  [..., W] x [W, 2W]
    x = torch.matmul(m, x)                                      # H axis:  [2H, H] x [..., H, 2W]
    return torch.einsum("ad,ncdhw->ncahw", m, x)               # D axis


# ---- HIP path: autograd nodes over the C ABI (channel-last fp32 volumes) ------------------------------------------------------------
def _ops():
    from . import ops
    return ops


class _Conv3dFn(torch.autograd.Function):
    """y = act(conv3d(x, w) + noise * ns + bias) with per-sample weights w [N, Cout, Cin, 3, 3, 3]; x, y channel-last [N, D, H, W, C]
    (include/asd_hip.h: asd_conv3d_fwd / _dgrad / _wgrad, asd_layer_act_bwd).  ax: the word in which x's producer left max|x| (or None);
    returns (y, ay) with ay the same for y — the split scales then cost no pass over the volumes."""

    @staticmethod
    def forward(ctx, x, ax, w, bias, noise, ns, act, gain, clamp):
        ops = _ops()
        if ax is None:
            ax = ops.absmax(x)
        ay = ops.new_amax(x.device)
        y = ops.conv3d_fwd(x, w, bias, noise, ns, act, gain, clamp, amax_x=ax, amax_out=ay)
        ctx.save_for_backward(x, ax, w, y if act else None, noise)
        ctx.act, ctx.gain, ctx.clamp, ctx.has_bias, ctx.has_noise = act, gain, clamp, bias is not None, noise is not None
        ctx.mark_non_differentiable(ay)
        return y, ay

    @staticmethod
    def backward(ctx, dy, _):
        x, ax, w, y, noise = ctx.saved_tensors
        ops = _ops()
        d_bias = d_ns = None
        if ctx.act:
            az = ops.new_amax(x.device)
            dz, d_bias, d_rows = ops.layer_act_bwd(dy, y, ctx.gain, ctx.clamp, want_bias=ctx.has_bias, want_rowsum=ctx.has_noise, amax_out=az)
            if ctx.has_noise:
                d_ns = torch.dot(d_rows, noise.reshape(-1)).reshape(1)
        else:
            dz = dy.contiguous()
            az = ops.absmax(dz)
        dx = ops.conv3d_dgrad(dz, w, x.shape[4], amax_dy=az) if ctx.needs_input_grad[0] else None
        dw = ops.conv3d_wgrad(x, dz, amax_x=ax, amax_dy=az) if ctx.needs_input_grad[2] else None
        return dx, None, dw, d_bias, None, d_ns, None, None, None


class _UpsampleFn(torch.autograd.Function):
    """y = act(trilinear_2x(x) + noise * ns + bias) + add on channel-last volumes (asd_upsample3d_fwd / _bwd; the activation's gradient is
    read off y, or with an added volume off the branch bits the forward kernel records); returns (y, ay) like _Conv3dFn"""

    @staticmethod
    def forward(ctx, x, bias, noise, ns, act, gain, clamp, add):
        ops = _ops()
        ay = ops.new_amax(x.device)
        # with an added volume (the block's const_bias) the activation's branch cannot be read off y - add exactly (a value clamped at
        # +-256 gain comes back as 255.9999, one below ulp(add) / 2 loses its sign): the kernel records two bits per element instead
        mask = (torch.empty(x.shape[0] * 8 * x.shape[1] ** 3 * x.shape[4] // 4, dtype=torch.uint8, device=x.device)
                if act and add is not None else None)
        y = ops.upsample3d_fwd(x, bias, noise, ns, act, gain, clamp, add, amax_out=ay, act_mask=mask)
        ctx.save_for_backward(y if act and mask is None else None, noise, mask)
        ctx.act, ctx.gain, ctx.clamp, ctx.has_bias, ctx.has_noise, ctx.has_add = act, gain, clamp, bias is not None, noise is not None, add is not None
        ctx.mark_non_differentiable(ay)
        return y, ay

    @staticmethod
    def backward(ctx, dy, _):
        y, noise, mask = ctx.saved_tensors
        ops = _ops()
        d_bias = d_ns = None
        if ctx.act:
            dz, d_bias, d_rows = ops.layer_act_bwd(dy, y if mask is None else dy, ctx.gain, ctx.clamp, want_bias=ctx.has_bias, want_rowsum=ctx.has_noise,
                                                   act_mask=mask)
            if ctx.has_noise:
                d_ns = torch.dot(d_rows, noise.reshape(-1)).reshape(1)
        else:
            dz = dy.contiguous()
        dx = ops.upsample3d_bwd(dz) if ctx.needs_input_grad[0] else None
        return dx, d_bias, None, d_ns, None, None, None, (dy if ctx.has_add else None)


class _ToRGBFn(torch.autograd.Function):
    """y = x w^T + bias (+ add) with per-sample w [N, 32, Cin] on channel-last rows, exact fp32 (asd_torgb_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, x, w, bias, add):
        ctx.save_for_backward(x, w)
        ctx.has_add = add is not None
        return _ops().torgb_fwd(x, w, bias, add)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dx, dw, db = _ops().torgb_bwd(x, dy, w, need_dx=ctx.needs_input_grad[0])
        return dx, dw, db, (dy if ctx.has_add else None)


_FUSED_MODULATION = os.environ.get("ASD_FUSED_MODULATION", "1") != "0"      # =0: the tensor-op form (same-box A/B, tools/)


class _ModWeightsFn(torch.autograd.Function):
    """the per-sample weights of modulated_conv3d (stylegan_3dconv_modules.py:64-82): weight * styles * gain, demodulated — one launch
    forward, two backward (asd_modulated_weights_fwd / _bwd) instead of six / a dozen tensor ops over [N, Cout, Cin, 27] floats"""

    @staticmethod
    def forward(ctx, weight, styles, gain, demodulate):
        wm, dcoef = _ops().modulated_weights_fwd(weight, styles, gain, demodulate)
        ctx.save_for_backward(weight, styles, wm, dcoef)
        ctx.gain, ctx.demodulate = gain, demodulate
        return wm

    @staticmethod
    def backward(ctx, d_wm):
        weight, styles, wm, dcoef = ctx.saved_tensors
        d_weight, d_styles = _ops().modulated_weights_bwd(d_wm, wm, weight, styles, dcoef, ctx.gain, ctx.demodulate)
        return d_weight, d_styles, None, None


def _conv3d_cl(x, ax, w, bias=None, noise=None, ns=None, act=False, gain=1.0, clamp=0.0):
    """the convolution node -> (y, ay); volumes below 16 x 16 in-plane (the 4^3 and 8^3 levels: 0.1 % of the generator's flops) are zero-padded
    to the kernel's 16 x 16 patch and cropped, with their layer tail as tensor ops on the cropped volume"""
    H, W = x.shape[2], x.shape[3]
    if H % 16 == 0 and W % 16 == 0:
        return _Conv3dFn.apply(x, ax, w, bias, noise, ns, act, gain, clamp)
    y = _Conv3dFn.apply(F.pad(x, (0, 0, 0, -W % 16, 0, -H % 16)), ax, w, None, None, None, False, 1.0, 0.0)[0][:, :, :H, :W]
    if noise is not None:
        y = y + (noise * ns)[..., None]
    if bias is not None:
        y = y + bias
    return (torch.clamp(F.leaky_relu(y, 0.2) * gain, -clamp, clamp) if act else y), None


class SynthesisLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, upsample=False):
        super().__init__()
        self.resolution, self.upsample, self.padding = resolution, upsample, kernel_size // 2
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size, kernel_size]))
        self.register_buffer("noise_const", torch.randn([resolution, resolution, resolution]))
        self.noise_strength = nn.Parameter(torch.zeros([1]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode, gain=1):
        x = modulated_conv3d(x, self.weight, self.affine(w), padding=self.padding)
        if self.upsample:
            x = _upsample2(x)
        if noise_mode == "random":
            r = self.resolution
            x = x + torch.randn([x.shape[0], 1, r, r, r], device=x.device) * self.noise_strength
        elif noise_mode == "const":
            x = x + self.noise_const * self.noise_strength
        else:
            raise TypeError(f"noise_mode {noise_mode!r}: the reference adds `None` here; only 'random' and 'const' are usable")
        act_gain = math.sqrt(2) * gain
        return torch.clamp(F.leaky_relu(x + self.bias[None, :, None, None, None], 0.2) * act_gain, -256 * gain, 256 * gain)

    def forward_cl(self, x, ax, w, noise_mode, gain=1, add=None):
        """the same layer on a channel-last volume x [N, D, H, W, Cin] through the HIP nodes -> (y, ay); ax / ay: max|.| words (or None);
        add: a volume added to the result (the block's const_bias, which the reference adds right after this layer)"""
        n, cin = x.shape[0], x.shape[4]
        if _FUSED_MODULATION:
            wm = _ModWeightsFn.apply(self.weight, self.affine(w), 1.0, True)
        else:
            wm = self.weight.unsqueeze(0) * self.affine(w).reshape(n, 1, cin, 1, 1, 1)
            wm = wm * (wm.square().sum(dim=[2, 3, 4, 5]) + 1e-8).rsqrt().reshape(n, -1, 1, 1, 1, 1)
        r = self.resolution
        if noise_mode == "random":
            noise = torch.randn([n, 1, r, r, r], device=x.device).reshape(n, r, r, r)       # (the reference's draw: same shape, same stream position)
        elif noise_mode == "const":
            noise = self.noise_const.expand(n, r, r, r).contiguous()
        else:
            raise TypeError(f"noise_mode {noise_mode!r}: the reference adds `None` here; only 'random' and 'const' are usable")
        act_gain, clamp = math.sqrt(2) * gain, 256.0 * gain
        if self.upsample:
            y0, _ = _conv3d_cl(x, ax, wm)
            return _UpsampleFn.apply(y0, self.bias, noise, self.noise_strength, True, act_gain, clamp, add)
        assert add is None
        return _conv3d_cl(x, ax, wm, self.bias, noise, self.noise_strength, True, act_gain, clamp)


class ToRGBLayer(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1):
        super().__init__()
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size, kernel_size]))
        self.bias = nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / math.sqrt(in_channels) * (kernel_size ** 3)

    def forward(self, x, w):
        return modulated_conv3d(x, self.weight, self.affine(w) * self.weight_gain, demodulate=False) + self.bias[None, :, None, None, None]

    def forward_cl(self, x, w):
        """channel-last: the 1x1x1 modulated convolution (no demodulation) is a per-sample matrix product on the [voxels, C] view"""
        n, cin = x.shape[0], x.shape[4]
        if _FUSED_MODULATION:
            wm = _ModWeightsFn.apply(self.weight.reshape(-1, cin), self.affine(w), self.weight_gain, False)      # [N, Cout, Cin]
        else:
            wm = self.weight.reshape(1, -1, cin) * (self.affine(w) * self.weight_gain).reshape(n, 1, cin)
        if wm.shape[1] == 32 and cin % 64 == 0:
            return _ToRGBFn.apply(x, wm, self.bias, None)
        y = torch.baddbmm(self.bias.reshape(1, 1, -1), x.reshape(n, -1, cin), wm.transpose(1, 2))       # other widths: library product
        return y.reshape(*x.shape[:4], -1)


class SynthesisPrologue(nn.Module):
    def __init__(self, out_channels, w_dim, resolution, img_channels):
        super().__init__()
        self.const = nn.Parameter(torch.randn([out_channels, resolution, resolution, resolution]))
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution)
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim)
        self.num_ws = 2

    def forward(self, ws, noise_mode="random"):
        x = self.const.unsqueeze(0).repeat([ws.shape[0], 1, 1, 1, 1])
        x = self.conv1(x, ws[:, 0], noise_mode=noise_mode)
        return x, self.torgb(x, ws[:, 1])

    def forward_cl(self, ws, noise_mode="random"):
        x = self.const.permute(1, 2, 3, 0).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1, 1])
        x, ax = self.conv1.forward_cl(x, None, ws[:, 0], noise_mode=noise_mode)
        return x, ax, self.torgb.forward_cl(x, ws[:, 1])


class SynthesisBlock(nn.Module):
    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, use_const_bias=False):
        super().__init__()
        self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, upsample=True)
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution)
        self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim)
        self.num_ws = 3
        self.const_bias = (nn.Parameter(torch.randn([out_channels, resolution, resolution, resolution]) / math.sqrt(out_channels))
                           if use_const_bias else None)

    def forward(self, x, img, ws, noise_mode="random"):
        x = self.conv0(x, ws[:, 0], noise_mode=noise_mode)
        if self.const_bias is not None:
            x = x + self.const_bias
        x = self.conv1(x, ws[:, 1], noise_mode=noise_mode)
        return x, _upsample2(img) + self.torgb(x, ws[:, 2])

    def forward_cl(self, x, ax, img, ws, noise_mode="random"):
        cb = None if self.const_bias is None else self.const_bias.permute(1, 2, 3, 0).contiguous().unsqueeze(0).expand(x.shape[0], -1, -1, -1, -1)
        x, ax = self.conv0.forward_cl(x, ax, ws[:, 0], noise_mode=noise_mode, add=cb)       # (+ const_bias inside the upsampling kernel)
        x, ax = self.conv1.forward_cl(x, ax, ws[:, 1], noise_mode=noise_mode)
        return x, ax, _UpsampleFn.apply(img, None, None, None, False, 1.0, 0.0, self.torgb.forward_cl(x, ws[:, 2]))[0]


class SynthesisNetwork3D(nn.Module):
    CHANNELS = {4: 512, 8: 512, 16: 512, 32: 256, 64: 128, 128: 64, 256: 32}

    def __init__(self, w_dim, img_resolution, img_channels, channel_multiplier=1, bias_resolution=64, backend="hip"):
        super().__init__()
        if backend not in ("hip", "library"):
            raise ValueError(f"unknown generator backend {backend!r}")
        self.backend = backend
        log2 = int(math.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, log2 + 1)]
        ch = {r: (c if r <= 16 else c * channel_multiplier) for r, c in self.CHANNELS.items()}
        self.blocks = nn.ModuleList()          # (registration order = the reference's state-dict key order)
        self.biases = nn.ParameterList()
        self.first_block = SynthesisPrologue(ch[4], w_dim=w_dim, resolution=4, img_channels=img_channels)
        for r in self.block_resolutions[1:]:
            self.blocks.append(SynthesisBlock(ch[r // 2], ch[r], w_dim=w_dim, resolution=r, img_channels=img_channels,
                                              use_const_bias=r <= bias_resolution))
        self.num_ws = self.first_block.num_ws + sum(b.num_ws for b in self.blocks)

    def forward(self, ws, noise_mode="random"):
        # style slices overlap by one (the toRGB style of a block is the first conv style of the next), as in StyleGAN2 (:161)
        if self.backend == "hip":
            if not ws.is_cuda:
                raise RuntimeError("Generator3D(backend='hip') needs device tensors: the HIP path has no CPU fallback "
                                   "(backend='library' is the explicit torch-op restatement)")
            x, ax, img = self.first_block.forward_cl(ws[:, 0:2], noise_mode=noise_mode)
            for i, blk in enumerate(self.blocks):
                x, ax, img = blk.forward_cl(x, ax, img, ws[:, 2 * (i + 1) + 1: 2 * (i + 1) + 4], noise_mode)
            return img.permute(0, 4, 1, 2, 3)          # [N, C, D, H, W] view of the channel-last volume (what the voxel sampler reads as is)
        x, img = self.first_block(ws[:, 0:2], noise_mode=noise_mode)
        for i, blk in enumerate(self.blocks):
            x, img = blk(x, img, ws[:, 2 * (i + 1) + 1: 2 * (i + 1) + 4], noise_mode)
        return img


class MappingNetwork(nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, lr_multiplier=0.01, w_avg_beta=0.995):
        super().__init__()
        self.c_dim, self.num_ws, self.num_layers = c_dim, num_ws, num_layers
        feats = [z_dim] + [w_dim] * num_layers
        self.layers = nn.ModuleList([FullyConnectedLayer(feats[i], feats[i + 1], activation="lrelu",
                                                         lr_multiplier=lr_multiplier if c_dim == 0 else 1) for i in range(num_layers)])
        self.embed = FullyConnectedLayer(c_dim + feats[-1], w_dim) if c_dim > 0 else None
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer("w_avg", torch.zeros([w_dim]))

    def forward(self, z, c=None, truncation_psi=1, truncation_cutoff=None):
        x = z * (z.square().mean(dim=1, keepdim=True) + 1e-8).rsqrt()
        for layer in self.layers:
            x = layer(x)
        if self.c_dim > 0:
            x = self.embed(torch.cat((x, c), dim=1))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            raise NotImplementedError("Truncation is not implemented")
        return x


class Generator3D(nn.Module):
    def __init__(self, z_dim, w_dim, num_layers, img_resolution, img_channels, c_dim=0, channel_multiplier=1, bias_resolution=64,
                 backend="hip", **unused):
        super().__init__()
        self.z_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, w_dim, img_resolution, img_channels
        self.synthesis = SynthesisNetwork3D(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels,
                                            channel_multiplier=channel_multiplier, bias_resolution=bias_resolution, backend=backend)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, num_layers=num_layers)

    def forward(self, z, c=None, truncation_psi=1, truncation_cutoff=None, noise_mode="random", report_stats=False):
        if report_stats:
            blocks = [self.synthesis.first_block.conv1] + [b.conv0 for b in self.synthesis.blocks]
            return {f"res_{4 * 2 ** i}": torch.norm(l.affine.weight, p=2).item() for i, l in enumerate(blocks)}
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff)
        return {"image": self.synthesis(ws, noise_mode=noise_mode)}


# ---- Triplane transformer ---------------------------------------------------------------------------------------------
class Attention(nn.Module):
    """diffusers.models.attention_processor.Attention as constructed at triplane_transformer_modules.py:45-53 (un-vendored,
    diffusers < 0.20): to_q / to_k / to_v without bias, to_out = [Linear(+bias), Dropout], softmax(q k^T / sqrt(d)) v."""

    def __init__(self, query_dim, heads, dim_head, cross_attention_dim=None, dropout=0.0, bias=False):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        ctx = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(ctx, inner, bias=bias)
        self.to_v = nn.Linear(ctx, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None):
        ctx = hidden_states if encoder_hidden_states is None else encoder_hidden_states
        n, lq, _ = hidden_states.shape
        split = lambda t: t.view(n, -1, self.heads, t.shape[-1] // self.heads).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(self.to_q(hidden_states)), split(self.to_k(ctx)), split(self.to_v(ctx)))
        return self.to_out[1](self.to_out[0](o.transpose(1, 2).reshape(n, lq, -1)))


class ConditionModulationBlock(nn.Module):
    """cross-attention to the text tokens, self-attention, MLP (pre-LayerNorm residual blocks; :34-72)"""

    def __init__(self, inner_dim, cond_dim, num_heads, eps, mlp_ratio=4.0, attn_drop=0.0, attn_bias=False, mlp_drop=0.0):
        super().__init__()
        self.norm1 = nn.LayerNorm(inner_dim, eps)
        self.cross_attn = Attention(inner_dim, num_heads, inner_dim // num_heads, cross_attention_dim=cond_dim, dropout=attn_drop, bias=attn_bias)
        self.norm2 = nn.LayerNorm(inner_dim, eps)
        self.self_attn = Attention(inner_dim, num_heads, inner_dim // num_heads, cross_attention_dim=inner_dim, dropout=attn_drop, bias=attn_bias)
        self.norm3 = nn.LayerNorm(inner_dim, eps)
        hid = int(inner_dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.Linear(inner_dim, hid), nn.GELU(), nn.Dropout(mlp_drop), nn.Linear(hid, inner_dim), nn.Dropout(mlp_drop))

    def forward(self, x, cond):
        x = x + self.cross_attn(self.norm1(x), cond)
        x = x + self.self_attn(self.norm2(x))
        return x + self.mlp(self.norm3(x))


class ConditionModulationBlockwoCrossAttn(nn.Module):
    """the condition is one extra token in front of the sequence (:74-112)"""

    def __init__(self, inner_dim, cond_dim, num_heads, eps, mlp_ratio=4.0, attn_drop=0.0, attn_bias=False, mlp_drop=0.0):
        super().__init__()
        self.norm2 = nn.LayerNorm(inner_dim, eps)
        self.self_attn = Attention(inner_dim, num_heads, inner_dim // num_heads, cross_attention_dim=inner_dim, dropout=attn_drop, bias=attn_bias)
        self.norm3 = nn.LayerNorm(inner_dim, eps)
        hid = int(inner_dim * mlp_ratio)
        self.mlp = nn.Sequential(nn.GELU(), nn.Linear(inner_dim, hid), nn.GELU(), nn.Linear(hid, inner_dim), nn.Dropout(mlp_drop))

    def forward(self, x, cond):
        x = torch.cat([cond, x], dim=1)
        x = x + self.self_attn(self.norm2(x))
        x = x + self.mlp(self.norm3(x))
        return x[:, 1:, :]


class _TritxBuffers:
    """Persistent buffers + HIP graphs of one (module, batch size, text length): the generator's forward is ~330 launches and its backward
    ~800, all enqueued by two C calls — the step was host-bound on them (4.5 ms of idle GPU per 56 ms step, profiles/r05_c5_*).  Every
    pointer a call sees is fixed here (text embedding, planes, the saved activations, the output gradient, ONE flat gradient buffer whose
    slices are the per-parameter gradients), so both calls are captured once and replayed.  `busy` marks saved activations that a backward
    pass still needs: a second forward before that backward takes the uncaptured path with buffers of its own."""

    def __init__(self, module, desc, table, packed, params, n, tokens, dev, own_workspace=False):
        L = _lib.lib()
        self.key = (n, tokens, packed.data_ptr()) + tuple(p.data_ptr() for p in params)
        self.desc, self.table, self.packed, self.n = desc, table, packed, n
        self.te = torch.zeros((n, tokens, desc.cond_dim), device=dev, dtype=torch.float32)
        self.planes = torch.empty((n, 3, 2 * desc.low_res, 2 * desc.low_res, desc.out_channels), device=dev, dtype=torch.float32)
        self.save = torch.empty(L.asd_tritx_save_floats(C.byref(desc), _lib.i32(n)), device=dev, dtype=torch.float32)
        self.d_cl = torch.empty_like(self.planes)
        # (the backward pass reads scratch its forward left in the workspace: a pass that runs between the two gets a workspace of its own)
        self.ws = (torch.empty(L.asd_tritx_workspace_floats(C.byref(desc)), device=dev, dtype=torch.float32) if own_workspace
                   else module._tritx_workspace(desc, dev))
        # the flat gradient buffer: [vector-shaped gradients (the kernels ACCUMULATE into them: zeroed per pass) | matrices]
        nl, D, Dc, Fh = desc.n_layers, desc.dim, desc.cond_dim, desc.hidden
        n_small = nl * (9 * D + Fh) + 2 * D
        sizes = []          # (parameter index or None, numel) of the matrix region, in allocation order
        for l in range(nl):
            P = params[20 * l:20 * l + 20]
            sizes += [("kv", l, 2 * D * Dc), ("qkv", l, 3 * D * D)] + [(i, l, P[i].numel()) for i in (2, 5, 12, 16, 18)]
        sizes += [(i, nl, params[20 * nl + i].numel()) for i in (0, 3)]
        total = n_small + sum(sz for _, _, sz in sizes)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.n_small = n_small
        self.offsets = [None] * len(params)           # (offset, shape) of every parameter's gradient inside `flat`
        cut = [0]

        def vec(i, p):
            self.offsets[i] = (cut[0], tuple(p.shape))
            cut[0] += p.numel()
        mat = {}
        off = n_small
        for name, l, sz in sizes:
            mat[(name, l)] = off
            off += sz
        ptr_offs = []
        for l in range(nl):
            P = params[20 * l:20 * l + 20]
            base = 20 * l
            for i in (0, 1, 6, 7, 8, 13, 14, 15, 17, 19):
                vec(base + i, P[i])
            for i in (2, 5, 12, 16, 18):
                self.offsets[base + i] = (mat[(i, l)], tuple(P[i].shape))
            kv, qkv = mat[("kv", l)], mat[("qkv", l)]
            self.offsets[base + 3], self.offsets[base + 4] = (kv, (D, Dc)), (kv + D * Dc, (D, Dc))
            self.offsets[base + 9], self.offsets[base + 10], self.offsets[base + 11] = (qkv, (D, D)), (qkv + D * D, (D, D)), (qkv + 2 * D * D, (D, D))
            o = lambda i: self.offsets[base + i][0]
            ptr_offs += [o(0), o(1), o(2), kv, o(5), o(6), o(7), o(8), qkv, o(12), o(13), o(14), o(15), o(16), o(17), o(18), o(19)]
        base = 20 * nl
        for i in range(4):
            if i in (1, 2):
                vec(base + i, params[base + i])
            else:
                self.offsets[base + i] = (mat[(i, nl)], tuple(params[base + i].shape))
            ptr_offs.append(self.offsets[base + i][0])
        assert cut[0] == n_small and off == total
        self.gtable = (C.c_void_p * len(ptr_offs))(*[self.flat.data_ptr() + 4 * o for o in ptr_offs])
        self.bdesc = _lib.TritxDesc.from_buffer_copy(desc)
        self.bdesc.grads_prezeroed = 1
        self.fwd_graph = self.bwd_graph = None
        self.fwd_runs = self.bwd_runs = 0
        self.busy = False
        self.use_graph = os.environ.get("ASD_TRITX_GRAPH", "1") != "0"

    def views(self, flat, params):
        return [flat[o:o + math.prod(sh)].view(sh) if p.requires_grad else None for (o, sh), p in zip(self.offsets, params)]

    def _fwd(self):
        _lib.check(_lib.lib().asd_tritx_fwd(C.byref(self.desc), self.table, _lib.ptr(self.packed), _lib.ptr(self.te), _lib.i32(self.n), _lib.ptr(self.planes),
                                            _lib.ptr(self.save), _lib.ptr(self.ws), _lib.stream()))

    def _bwd(self):
        self.flat[:self.n_small].zero_()
        _lib.check(_lib.lib().asd_tritx_bwd(C.byref(self.bdesc), self.table, _lib.ptr(self.packed), _lib.ptr(self.te), _lib.i32(self.n), _lib.ptr(self.d_cl),
                                            _lib.ptr(self.save), self.gtable, _lib.ptr(self.ws), _lib.stream()))

    def _run(self, which):
        """first call eager (lazy kernel attributes), second call captured, then replays"""
        fn, graph, runs = (self._fwd, self.fwd_graph, self.fwd_runs) if which == "fwd" else (self._bwd, self.bwd_graph, self.bwd_runs)
        if graph is not None:
            graph.replay()
            return
        if self.use_graph and runs >= 1 and not torch.cuda.is_current_stream_capturing():
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                fn()
            g.replay()
            if which == "fwd":
                self.fwd_graph = g
            else:
                self.bwd_graph = g
        else:
            fn()
        if which == "fwd":
            self.fwd_runs += 1
        else:
            self.bwd_runs += 1


class _TritxFn(torch.autograd.Function):
    """the whole generator as ONE autograd node on the HIP path (include/asd_hip.h: asd_tritx_pack / _fwd / _bwd, csrc/tritx.hip): every
    Linear / attention product on the fp16 matrix pipe at fp32-class accuracy (operands split into two fp16 planes), LayerNorm / GELU /
    residuals in fp32; nothing in between is a Python-issued launch.  Both passes replay HIP graphs over persistent buffers (_TritxBuffers)."""

    @staticmethod
    def forward(ctx, module, text_embed, *params):
        desc, table, packed = module._tritx_state(params)
        n, dev = text_embed.shape[0], text_embed.device
        need_grad = any(ctx.needs_input_grad)          # (grad mode is off INSIDE a Function's forward: torch.is_grad_enabled() says nothing here)
        bufs = module._tritx_buffers(desc, table, packed, params, n, text_embed.shape[1], dev)
        if bufs.busy:
            # An earlier forward still waits for its backward pass, which reads the saved activations, the text embedding and scratch its
            # forward left in the workspace: this pass gets buffers of its own (rare: a validation render between a training forward and
            # its backward), and no graphs
            bufs = _TritxBuffers(module, desc, table, packed, params, n, text_embed.shape[1], dev, own_workspace=True)
            bufs.use_graph = False
        bufs.te.copy_(text_embed)
        bufs._run("fwd")
        bufs.busy = need_grad
        ctx.module, ctx.bufs, ctx.params = module, bufs, params     # (params held for their pointers: the table refers to them)
        ctx.set_materialize_grads(False)
        return bufs.planes.clone().permute(0, 1, 4, 2, 3)     # the reference's [N, 3, C, H, W] as a view of channel-last planes (a copy: the buffer is reused)

    @staticmethod
    def backward(ctx, d_planes):
        params, bufs, module = ctx.params, ctx.bufs, ctx.module
        bufs.busy = False
        if d_planes is None:
            return (None, None) + (None,) * len(params)
        bufs.d_cl.copy_(d_planes.permute(0, 1, 3, 4, 2))
        bufs._run("bwd")
        # Gradient accumulation over micro-batches (accumulate_grad_batches = 8 in the shipped YAML): autograd would add 244 tensors one launch
        # at a time.  The gradients leave as views of ONE copy of the flat buffer; while every parameter's .grad still IS its view of the
        # copy handed out by an earlier micro-batch, the sum is one add of the flat buffers and nothing is returned.  (Single process only: the
        # data-parallel exchange launches its collectives from autograd's accumulation hooks.)
        acc = getattr(module, "_tritx_acc", None)
        from . import dist as asd_dist
        if (acc is not None and acc[0] is bufs.offsets and not asd_dist.is_distributed()
                and all((not p.requires_grad) or (p.grad is not None and p.grad.data_ptr() == acc[1].data_ptr() + 4 * o) for p, (o, _) in zip(params, bufs.offsets))):
            acc[1].add_(bufs.flat)
            return (None, None) + (None,) * len(params)
        out = bufs.flat.clone()
        module._tritx_acc = (bufs.offsets, out)
        return (None, None) + tuple(bufs.views(out, params))


class TriplaneTransformer(nn.Module):
    def __init__(self, inner_dim, condition_dim, triplane_low_res, triplane_high_res, triplane_dim, num_layers, num_heads, local_text,
                 mlp_ratio=4.0, eps=1e-6, backend="hip", **unused):
        super().__init__()
        self.backend, self.eps, self.num_heads = backend, eps, num_heads
        self.triplane_low_res, self.triplane_high_res, self.triplane_dim = triplane_low_res, triplane_high_res, triplane_dim
        self.pos_embed = nn.Parameter(torch.randn(1, 3 * triplane_low_res ** 2, inner_dim) * (1.0 / inner_dim) ** 0.5)
        self.needs_local_text = local_text
        blk = ConditionModulationBlock if local_text else ConditionModulationBlockwoCrossAttn
        self.layers = nn.ModuleList([blk(inner_dim=inner_dim, cond_dim=condition_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, eps=eps)
                                     for _ in range(num_layers)])
        self.norm = nn.LayerNorm(inner_dim, eps=eps)
        self.deconv = nn.ConvTranspose2d(inner_dim, triplane_dim, kernel_size=2, stride=2, padding=0, bias=False)
        if not local_text:
            self.proj = nn.Linear(condition_dim, inner_dim)

    # ---- HIP path (the shipped configuration: local_text = True, head dim 48) ----------------------------------------------------------------
    _LAYER_KEYS = ("norm1.weight", "norm1.bias", "cross_attn.to_q.weight", "cross_attn.to_k.weight", "cross_attn.to_v.weight", "cross_attn.to_out.0.weight",
                   "cross_attn.to_out.0.bias", "norm2.weight", "norm2.bias", "self_attn.to_q.weight", "self_attn.to_k.weight", "self_attn.to_v.weight",
                   "self_attn.to_out.0.weight", "self_attn.to_out.0.bias", "norm3.weight", "norm3.bias", "mlp.0.weight", "mlp.0.bias", "mlp.3.weight", "mlp.3.bias")

    def _hip_ok(self, text_embed) -> bool:
        d = self.pos_embed.shape[-1]
        return (self.backend == "hip" and self.needs_local_text and text_embed.is_cuda and text_embed.dim() == 3 and d == self.num_heads * 48
                and d % 64 == 0 and d <= 1024 and self.triplane_high_res == 2 * self.triplane_low_res and (4 * self.triplane_dim) % 64 == 0
                and os.environ.get("ASD_TRITX", "1") != "0")

    def _hip_params(self):
        ps = []
        for layer in self.layers:
            named = dict(layer.named_parameters())
            ps += [named[k] for k in self._LAYER_KEYS]
        return ps + [self.pos_embed, self.norm.weight, self.norm.bias, self.deconv.weight]

    def _tritx_state(self, params):
        """(descriptor, pointer table, packed operand planes); the planes are rebuilt when a weight has changed (optimizer step, checkpoint load)"""
        # the fused optimizers write parameters through raw pointers and bump `_version` themselves (optimizers._Table.launch)
        key = (self._cond_tokens,) + tuple((p.data_ptr(), p._version) for p in params)
        st = getattr(self, "_tritx_cache", None)
        if st is not None and st[0] == key:
            return st[1], st[2], st[3]
        layer0 = self.layers[0]
        desc = _lib.TritxDesc(n_layers=len(self.layers), dim=self.pos_embed.shape[-1], heads=self.num_heads, cond_dim=layer0.cross_attn.to_k.weight.shape[1],
                              cond_tokens=self._cond_tokens, hidden=layer0.mlp[0].weight.shape[0], low_res=self.triplane_low_res,
                              out_channels=self.triplane_dim, eps=float(self.eps))
        for p in params:
            if not (p.is_contiguous() and p.dtype == torch.float32):
                raise _lib.AsdError("TriplaneTransformer (HIP): parameters must be contiguous fp32")
        table = (C.c_void_p * len(params))(*[p.data_ptr() for p in params])
        n = _lib.lib().asd_tritx_packed_floats(C.byref(desc))
        if n < 0:
            raise _lib.AsdError(_lib.lib().asd_last_error().decode())
        packed = st[3] if st is not None and st[3].numel() == n else torch.empty(n, device=params[0].device, dtype=torch.float32)
        _lib.check(_lib.lib().asd_tritx_pack(C.byref(desc), table, _lib.ptr(packed), _lib.stream()))
        self._tritx_cache = (key, desc, table, packed)
        return desc, table, packed

    def _tritx_buffers(self, desc, table, packed, params, n, tokens, dev):
        key = (n, tokens, packed.data_ptr()) + tuple(p.data_ptr() for p in params)
        cache = self.__dict__.setdefault("_tritx_bufs", {})
        b = cache.get((n, tokens))
        if b is None or b.key != key:
            b = cache[(n, tokens)] = _TritxBuffers(self, desc, table, packed, params, n, tokens, dev)
        return b

    def _tritx_workspace(self, desc, dev):
        n = _lib.lib().asd_tritx_workspace_floats(C.byref(desc))
        ws = getattr(self, "_tritx_ws", None)
        if ws is None or ws.numel() < n or ws.device != dev:
            ws = self._tritx_ws = torch.empty(n, device=dev, dtype=torch.float32)
        return ws

    def forward(self, text_embed):
        if self._hip_ok(text_embed):
            self._cond_tokens = text_embed.shape[1]
            return _TritxFn.apply(self, text_embed, *self._hip_params())
        if self.backend == "hip" and text_embed.is_cuda and os.environ.get("ASD_TRITX", "1") != "0":
            # the HIP generator is instantiated for the shipped shape family (local text tokens, head dim 48, width % 64 == 0 and <= 1024,
            # 2x deconvolution).  Anything else — reduced test models, the global-text variant — is NOT silently run on library kernels
            # under the name of the HIP path: ask for the torch-op restatement explicitly
            raise NotImplementedError("TriplaneTransformer(backend='hip'): this configuration is outside the HIP generator's shape family (local_text, head dim 48, "
                                      "width % 64 == 0 and <= 1024, triplane_high_res == 2 * triplane_low_res, 4 * triplane_dim % 64 == 0); "
                                      "construct it with backend='library' for the library-op restatement")
        N, Hh = text_embed.shape[0], self.triplane_low_res
        if not self.needs_local_text:
            text_embed = self.proj(text_embed).unsqueeze(1)
        x = self.pos_embed.repeat(N, 1, 1)
        for layer in self.layers:
            x = layer(x, text_embed)
        x = self.norm(x).view(N, 3, Hh, Hh, -1).permute(1, 0, 4, 2, 3).contiguous().view(3 * N, -1, Hh, Hh)   # [3N, D, H, W]
        x = self.deconv(x)
        x = x.view(3, N, *x.shape[-3:]).permute(1, 0, 2, 3, 4).contiguous()                                       # [N, 3, D', H', W']
        assert self.triplane_high_res == x.shape[-2], f"Output triplane resolution does not match with expected: {x.shape[-2]} vs {self.triplane_high_res}"
        assert self.triplane_dim == x.shape[-3], f"Output triplane dimension does not match with expected: {x.shape[-3]} vs {self.triplane_dim}"
        return x
