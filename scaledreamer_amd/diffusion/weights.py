"""Architecture description (parameter names + shapes) and seeded random weights for the frozen diffusion
prior of the ASD step.

Names follow the LDM / Stability state-dict layout of the reference's vendored model
(extern/mvdream/ldm/modules/diffusionmodules/openaimodel.py:422-808 `UNetModel`, :811-1213
`MultiViewUNetModel`; extern/mvdream/ldm/modules/diffusionmodules/model.py:452-543 VAE `Encoder`;
extern/mvdream/ldm/models/autoencoder.py:32 `quant_conv`), so a real SD-2.1-base / MVDream checkpoint in
that layout loads unchanged.  No pretrained weights exist offline: benchmarks and parity tests use
`gen_params` (a seeded, name-keyed rule, so any single tensor can be regenerated on its own).
"""
from __future__ import annotations

import zlib
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch


@dataclass
class UNetConfig:
    """extern/mvdream/configs/sd-v2-base.yaml:13-27 (SD-2.1-base UNet = the same minus camera_dim)."""
    in_channels: int = 4
    out_channels: int = 4
    model_channels: int = 320
    attention_resolutions: Tuple[int, ...] = (4, 2, 1)
    num_res_blocks: int = 2
    channel_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_head_channels: int = 64
    transformer_depth: int = 1
    context_dim: int = 1024
    camera_dim: Optional[int] = None   # 16 for MVDream

    @property
    def time_embed_dim(self) -> int:
        return self.model_channels * 4


@dataclass
class VAEConfig:
    """first_stage_config.ddconfig of sd-v2-base.yaml:33-50"""
    in_channels: int = 3
    ch: int = 128
    ch_mult: Tuple[int, ...] = (1, 2, 4, 4)
    num_res_blocks: int = 2
    z_channels: int = 4
    embed_dim: int = 4
    scale_factor: float = 0.18215


Shapes = Dict[str, Tuple[int, ...]]


def _conv(s: Shapes, name: str, cin: int, cout: int, k: int):
    s[name + ".weight"] = (cout, cin, k, k)
    s[name + ".bias"] = (cout,)


def _lin(s: Shapes, name: str, cin: int, cout: int, bias: bool = True):
    s[name + ".weight"] = (cout, cin)
    if bias:
        s[name + ".bias"] = (cout,)


def _norm(s: Shapes, name: str, c: int):
    s[name + ".weight"] = (c,)
    s[name + ".bias"] = (c,)


def _resblock(s: Shapes, p: str, cin: int, cout: int, emb: int):
    _norm(s, p + ".in_layers.0", cin)
    _conv(s, p + ".in_layers.2", cin, cout, 3)
    _lin(s, p + ".emb_layers.1", emb, cout)
    _norm(s, p + ".out_layers.0", cout)
    _conv(s, p + ".out_layers.3", cout, cout, 3)
    if cin != cout:
        _conv(s, p + ".skip_connection", cin, cout, 1)


def _transformer(s: Shapes, p: str, c: int, ctx: int, depth: int):
    _norm(s, p + ".norm", c)
    _lin(s, p + ".proj_in", c, c)
    for d in range(depth):
        b = f"{p}.transformer_blocks.{d}"
        for attn, kv in (("attn1", c), ("attn2", ctx)):
            _lin(s, f"{b}.{attn}.to_q", c, c, bias=False)
            _lin(s, f"{b}.{attn}.to_k", kv, c, bias=False)
            _lin(s, f"{b}.{attn}.to_v", kv, c, bias=False)
            _lin(s, f"{b}.{attn}.to_out.0", c, c)
        _lin(s, f"{b}.ff.net.0.proj", c, 8 * c)
        _lin(s, f"{b}.ff.net.2", 4 * c, c)
        for n in ("norm1", "norm2", "norm3"):
            _norm(s, f"{b}.{n}", c)
    _lin(s, p + ".proj_out", c, c)


@dataclass
class UNetBlock:
    """One entry of input_blocks / middle_block / output_blocks: an ordered list of (kind, prefix, cin, cout)."""
    layers: List[Tuple[str, str, int, int]] = field(default_factory=list)


def unet_layout(cfg: UNetConfig):
    """(shapes, input_blocks, middle_block, output_blocks) — the construction loop of openaimodel.py:563-752."""
    s: Shapes = {}
    mc, emb = cfg.model_channels, cfg.time_embed_dim
    _lin(s, "time_embed.0", mc, emb)
    _lin(s, "time_embed.2", emb, emb)
    if cfg.camera_dim is not None:
        _lin(s, "camera_embed.0", cfg.camera_dim, emb)
        _lin(s, "camera_embed.2", emb, emb)
    inputs: List[UNetBlock] = []
    _conv(s, "input_blocks.0.0", cfg.in_channels, mc, 3)
    inputs.append(UNetBlock([("conv", "input_blocks.0.0", cfg.in_channels, mc)]))
    chans, ch, ds = [mc], mc, 1
    for level, mult in enumerate(cfg.channel_mult):
        for _ in range(cfg.num_res_blocks):
            i = len(inputs)
            blk = UNetBlock()
            _resblock(s, f"input_blocks.{i}.0", ch, mult * mc, emb)
            blk.layers.append(("res", f"input_blocks.{i}.0", ch, mult * mc))
            ch = mult * mc
            if ds in cfg.attention_resolutions:
                _transformer(s, f"input_blocks.{i}.1", ch, cfg.context_dim, cfg.transformer_depth)
                blk.layers.append(("attn", f"input_blocks.{i}.1", ch, ch))
            inputs.append(blk)
            chans.append(ch)
        if level != len(cfg.channel_mult) - 1:
            i = len(inputs)
            _conv(s, f"input_blocks.{i}.0.op", ch, ch, 3)
            inputs.append(UNetBlock([("down", f"input_blocks.{i}.0.op", ch, ch)]))
            chans.append(ch)
            ds *= 2
    middle = UNetBlock()
    _resblock(s, "middle_block.0", ch, ch, emb)
    _transformer(s, "middle_block.1", ch, cfg.context_dim, cfg.transformer_depth)
    _resblock(s, "middle_block.2", ch, ch, emb)
    middle.layers += [("res", "middle_block.0", ch, ch), ("attn", "middle_block.1", ch, ch), ("res", "middle_block.2", ch, ch)]
    outputs: List[UNetBlock] = []
    for level, mult in list(enumerate(cfg.channel_mult))[::-1]:
        for i in range(cfg.num_res_blocks + 1):
            ich = chans.pop()
            j = len(outputs)
            blk = UNetBlock()
            _resblock(s, f"output_blocks.{j}.0", ch + ich, mc * mult, emb)
            blk.layers.append(("res", f"output_blocks.{j}.0", ch + ich, mc * mult))
            ch = mc * mult
            n = 1
            if ds in cfg.attention_resolutions:
                _transformer(s, f"output_blocks.{j}.{n}", ch, cfg.context_dim, cfg.transformer_depth)
                blk.layers.append(("attn", f"output_blocks.{j}.{n}", ch, ch))
                n += 1
            if level and i == cfg.num_res_blocks:
                _conv(s, f"output_blocks.{j}.{n}.conv", ch, ch, 3)
                blk.layers.append(("up", f"output_blocks.{j}.{n}.conv", ch, ch))
                ds //= 2
            outputs.append(blk)
    _norm(s, "out.0", ch)
    _conv(s, "out.2", mc, cfg.out_channels, 3)
    return s, inputs, middle, outputs


def vae_encoder_layout(cfg: VAEConfig):
    """(shapes, plan) of Encoder (model.py:452-543) + quant_conv; plan = ordered (kind, prefix, cin, cout)."""
    s: Shapes = {}
    plan: List[Tuple[str, str, int, int]] = []
    _conv(s, "encoder.conv_in", cfg.in_channels, cfg.ch, 3)
    plan.append(("conv", "encoder.conv_in", cfg.in_channels, cfg.ch))
    in_mult = (1,) + tuple(cfg.ch_mult)
    block_in = cfg.ch
    for lvl in range(len(cfg.ch_mult)):
        block_in, block_out = cfg.ch * in_mult[lvl], cfg.ch * cfg.ch_mult[lvl]
        for b in range(cfg.num_res_blocks):
            p = f"encoder.down.{lvl}.block.{b}"
            _vae_res(s, p, block_in, block_out)
            plan.append(("res", p, block_in, block_out))
            block_in = block_out
        if lvl != len(cfg.ch_mult) - 1:
            _conv(s, f"encoder.down.{lvl}.downsample.conv", block_in, block_in, 3)
            plan.append(("down", f"encoder.down.{lvl}.downsample.conv", block_in, block_in))
    _vae_res(s, "encoder.mid.block_1", block_in, block_in)
    plan.append(("res", "encoder.mid.block_1", block_in, block_in))
    p = "encoder.mid.attn_1"
    _norm(s, p + ".norm", block_in)
    for n in ("q", "k", "v", "proj_out"):
        _conv(s, f"{p}.{n}", block_in, block_in, 1)
    plan.append(("attn", p, block_in, block_in))
    _vae_res(s, "encoder.mid.block_2", block_in, block_in)
    plan.append(("res", "encoder.mid.block_2", block_in, block_in))
    _norm(s, "encoder.norm_out", block_in)
    _conv(s, "encoder.conv_out", block_in, 2 * cfg.z_channels, 3)
    plan.append(("out", "encoder", block_in, 2 * cfg.z_channels))
    _conv(s, "quant_conv", 2 * cfg.z_channels, 2 * cfg.embed_dim, 1)
    plan.append(("quant", "quant_conv", 2 * cfg.z_channels, 2 * cfg.embed_dim))
    return s, plan


def _vae_res(s: Shapes, p: str, cin: int, cout: int):
    _norm(s, p + ".norm1", cin)
    _conv(s, p + ".conv1", cin, cout, 3)
    _norm(s, p + ".norm2", cout)
    _conv(s, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(s, p + ".nin_shortcut", cin, cout, 1)


def _is_norm(name: str) -> bool:
    base = name.rsplit(".", 1)[0]
    leaf = base.rsplit(".", 1)[-1]
    return (leaf.startswith("norm") or base.endswith("in_layers.0") or base.endswith("out_layers.0") or base == "out.0")


def gen_param(name: str, shape: Tuple[int, ...], seed: int, dtype=torch.float32) -> torch.Tensor:
    """Seeded, name-keyed random value of one parameter.
      norm scale  : 1 + 0.1 N(0,1)        norm shift / bias : 0.05 N(0,1)
      weight      : N(0,1) / sqrt(fan_in)  (keeps activations O(1) through ~60 layers; zero-initialised
                    modules of the reference are randomised too, otherwise the output is identically 0)"""
    g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    x = torch.randn(shape, generator=g, dtype=torch.float32)
    if name.endswith(".bias"):
        x = x * 0.05
    elif _is_norm(name):
        x = 1.0 + 0.1 * x
    else:
        fan_in = 1
        for d in shape[1:]:
            fan_in *= d
        x = x / float(fan_in) ** 0.5
    return x.to(dtype)


def gen_params(shapes: Shapes, seed: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    return {n: gen_param(n, shp, seed, dtype) for n, shp in shapes.items()}


def count_params(shapes: Shapes) -> int:
    total = 0
    for shp in shapes.values():
        n = 1
        for d in shp:
            n *= d
        total += n
    return total


# ---- packing into the weight tables the C schedules publish (include/asd_hip.h: asd_unet_weight_info / asd_vae_enc_weight_info) ----
def _pack_conv3x3(w: torch.Tensor, cin_pad: Optional[int] = None) -> torch.Tensor:
    """PyTorch [Cout, Cin, 3, 3] -> [Cout, 9*Cin'] with k = (ky, kx, cin), Cin' = Cin padded to a multiple of 32."""
    cout, cin = w.shape[:2]
    cp = cin_pad or ((cin + 31) // 32 * 32)
    wp = torch.zeros((cout, 3, 3, cp), dtype=w.dtype, device=w.device)
    wp[..., :cin] = w.permute(0, 2, 3, 1)
    return wp.reshape(cout, 9 * cp)


def _pack_upsample_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """Upsample.conv (openaimodel.py:99-131: nearest 2x, then 3x3 / pad 1) in its parity form: output pixel (2y + a, 2x + b) only
    sees the 2 x 2 input pixels (y - 1 + a + dy, x - 1 + b + dx), dy, dx in {0, 1}, with the taps that land on the same pixel summed:
    rows a = 0: {ky = 0} -> dy = 0, {1, 2} -> dy = 1;  a = 1: {0, 1} -> dy = 0, {2} -> dy = 1 (columns alike).
    PyTorch [Cout, Cin, 3, 3] -> [4 * Cout, 4 * Cin'] = [parity (a, b)][cout][(dy, dx, cin)], summed in fp32."""
    cout, cin = w.shape[:2]
    cp = (cin + 31) // 32 * 32
    taps = {0: ((0,), (1, 2)), 1: ((0, 1), (2,))}
    wf = w.float()
    out = torch.zeros((2, 2, cout, 2, 2, cp), dtype=torch.float32, device=w.device)
    for a in (0, 1):
        for b in (0, 1):
            for dy in (0, 1):
                for dx in (0, 1):
                    acc = 0
                    for ky in taps[a][dy]:
                        for kx in taps[b][dx]:
                            acc = acc + wf[:, :, ky, kx]
                    out[a, b, :, dy, dx, :cin] = acc
    return out.reshape(4 * cout, 4 * cp).to(w.dtype)


def _geglu_rows(c: int) -> torch.Tensor:
    """GEGLU.proj rows [value (c) | gate (c)] (attention.py:49-56) -> interleaved in 32-row groups [16 value | 16 gate]: the layout
    the fused epilogue of asd_gemm_f16 (act = 2) reads."""
    assert c % 16 == 0
    return torch.stack([torch.arange(c).view(-1, 16), torch.arange(c, 2 * c).view(-1, 16)], dim=1).reshape(-1)


LN_FOLD_MIN_C = 1024      # transformer blocks at least this wide fold their LayerNorms into the consuming GEMMs (csrc/net.hip: ASD_LN_FOLD_MIN_C)


def _ln_fold(w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor):
    """LayerNorm folded into the Linear that consumes it (asd_gemm_args.ln_mode, csrc/gemm_tile.h: ln_fold):
        LN(x) W^T = rstd * (x W'^T - mean * s) + c,   W' = gamma (.) W,  s = rowsum(W'),  c = W beta.
    W' is rounded to fp16 here and s is summed from the ROUNDED values (the kernel's accumulator holds exactly mean * s plus the
    centred part, the subtraction must cancel the same numbers).  Returns (W' fp16, {s, c} fp32 [2, N] viewed as fp16 [4 N] — the weight
    table of csrc/net.hip carries fp16 matrices only)."""
    wf = w.float()
    w2 = (wf * gamma.float()[None, :]).half()
    sc = torch.stack([w2.float().sum(1), wf @ beta.float()]).contiguous()
    return w2, sc.view(torch.float16).reshape(1, -1)


def pack_unet(p: Dict[str, torch.Tensor], cfg: UNetConfig) -> Dict[str, torch.Tensor]:
    """name-keyed LDM UNet state dict -> the packed fp32/fp16-agnostic matrices of the C table (names as csrc/net.hip: unet_build)."""
    _, inputs, middle, outputs = unet_layout(cfg)
    out: Dict[str, torch.Tensor] = {}
    emb_w, emb_b, ck, cv = [], [], [], []
    up_convs = {n + ".weight" for blk in outputs for kind, n, _, _ in blk.layers if kind == "up"}
    for name, t in p.items():
        if ".emb_layers.1." in name or any(k in name for k in (".attn1.to_q.", ".attn1.to_k.", ".attn2.to_k.", ".attn2.to_v.")):
            continue
        if name.endswith(".weight") and t.ndim == 4:
            if name in up_convs:
                out[name] = _pack_upsample_conv3x3(t)
            else:
                out[name] = _pack_conv3x3(t) if t.shape[-1] == 3 else t.reshape(t.shape[0], t.shape[1])   # 1x1 conv == Linear on NHWC
        else:
            out[name] = t
    for blk in list(inputs) + [middle] + list(outputs):            # the order csrc/net.hip assigns the column offsets in
        for kind, name, cin, cout in blk.layers:
            if kind == "res":
                emb_w.append(p[name + ".emb_layers.1.weight"])
                emb_b.append(p[name + ".emb_layers.1.bias"])
            elif kind == "attn":
                for d in range(cfg.transformer_depth):
                    b = f"{name}.transformer_blocks.{d}"
                    ck.append(p[b + ".attn2.to_k.weight"])
                    cv.append(p[b + ".attn2.to_v.weight"])
                    perm = _geglu_rows(4 * cout).to(p[b + ".ff.net.0.proj.weight"].device)
                    out[b + ".ff.net.0.proj.bias"] = p[b + ".ff.net.0.proj.bias"][perm]
                    if cout < LN_FOLD_MIN_C:      # the wide levels keep their LayerNorm kernels (csrc/net.hip: u_attn)
                        out[b + ".attn1.to_qk.weight"] = torch.cat([p[b + ".attn1.to_q.weight"], p[b + ".attn1.to_k.weight"]], 0)
                        out[b + ".ff.net.0.proj.weight"] = p[b + ".ff.net.0.proj.weight"][perm]
                        continue
                    # norm1 -> (q | k, v), norm2 -> cross-attention q, norm3 -> GEGLU projection: folded into those weights
                    for dst, wmat, nm in ((b + ".attn1.to_qk", torch.cat([p[b + ".attn1.to_q.weight"], p[b + ".attn1.to_k.weight"]], 0), ".norm1"),
                                          (b + ".attn1.to_v", p[b + ".attn1.to_v.weight"], ".norm1"),
                                          (b + ".attn2.to_q", p[b + ".attn2.to_q.weight"], ".norm2"),
                                          (b + ".ff.net.0.proj", p[b + ".ff.net.0.proj.weight"][perm], ".norm3")):
                        out[dst + ".weight"], out[dst + ".ln_sc"] = _ln_fold(wmat, p[b + nm + ".weight"], p[b + nm + ".bias"])
                    for nm in (".norm1", ".norm2", ".norm3"):          # no longer read: the table of csrc/net.hip does not list them
                        out.pop(b + nm + ".weight", None), out.pop(b + nm + ".bias", None)
    out["emb_all.weight"], out["emb_all.bias"] = torch.cat(emb_w, 0), torch.cat(emb_b, 0)
    out["ctx_k_all.weight"], out["ctx_v_all.weight"] = torch.cat(ck, 0), torch.cat(cv, 0)
    return out


def _pack_stride2_dgrad_conv3x3(w: torch.Tensor) -> torch.Tensor:
    """Input gradient of the encoder's Downsample convolution (model.py:80-85: pad (0,1,0,1), then 3x3 / stride 2 / pad 0) in the parity
    form of the upsampling-convolution kernel (asd_gemm_args.upsample = 3): dX[Y, X] = sum_{ky, kx} dY[(Y - ky) / 2, (X - kx) / 2] W[ky, kx]
    over the taps with even numerators, so input row Y = 2y + a only sees the low-resolution rows y - 1 + a + ty, ty in {0, 1}:
    a = 0: ty = 0 <- ky = 2, ty = 1 <- ky = 0;  a = 1: ty = 0 <- ky = 1, ty = 1 <- nothing (columns alike).  9 of the 16 tap slots are
    live: 4/9 of the multiply-adds of the nine-tap gather form (asd_gemm_args.upsample = 2), whose other taps multiply zero pages.
    Forward weight [Cout, Cin, 3, 3] -> [4 * Cin', 4 * Cout'] = [parity (a, b)][cin][(ty, tx, cout)]."""
    cout, cin = w.shape[:2]
    cin_p, cout_p = (cin + 31) // 32 * 32, (cout + 31) // 32 * 32
    taps = {0: (2, 0), 1: (1, None)}
    out = torch.zeros((2, 2, cin_p, 2, 2, cout_p), dtype=torch.float32, device=w.device)
    for a in (0, 1):
        for b in (0, 1):
            for ty in (0, 1):
                for tx in (0, 1):
                    ky, kx = taps[a][ty], taps[b][tx]
                    if ky is not None and kx is not None:
                        out[a, b, :cin, ty, tx, :cout] = w[:, :, ky, kx].float().t()
    return out.reshape(4 * cin_p, 4 * cout_p)


def pack_vae_encoder(p: Dict[str, torch.Tensor], cfg: VAEConfig) -> Dict[str, torch.Tensor]:
    """LDM first-stage encoder state dict -> the C table of csrc/net.hip: vae_build.  Every convolution gets its forward matrix and
    the matrix of its input gradient (roles of the channel axes swapped, taps flipped for stride 1; the stride-2 gradient in the parity
    form of _pack_stride2_dgrad_conv3x3); conv_out and quant_conv are both linear and are composed into one convolution (exact)."""
    _, plan = vae_encoder_layout(cfg)
    out: Dict[str, torch.Tensor] = {}

    def conv(name, weight, bias, stride=1):
        wt = weight.float()
        out[name + ".fwd"] = _pack_conv3x3(wt)
        cin = wt.shape[1]
        cin_p = (cin + 31) // 32 * 32
        wb = wt.permute(1, 0, 2, 3)
        if stride == 1:
            wb = wb.flip(2, 3)
        if cin_p != cin:                                  # gradient w.r.t. the zero-padded input channels
            wb = torch.cat([wb, wb.new_zeros(cin_p - cin, *wb.shape[1:])], 0)
        out[name + ".bwd"] = _pack_conv3x3(wb) if stride == 1 else _pack_stride2_dgrad_conv3x3(wt)
        out[name + ".bias"] = bias.float()

    for kind, name, cin, cout in plan:
        if kind == "conv":
            conv(name, p[name + ".weight"], p[name + ".bias"])
        elif kind == "res":
            for n in ("norm1", "norm2"):
                out[f"{name}.{n}.weight"], out[f"{name}.{n}.bias"] = p[f"{name}.{n}.weight"], p[f"{name}.{n}.bias"]
            conv(name + ".conv1", p[name + ".conv1.weight"], p[name + ".conv1.bias"])
            conv(name + ".conv2", p[name + ".conv2.weight"], p[name + ".conv2.bias"])
            if name + ".nin_shortcut.weight" in p:
                ws = p[name + ".nin_shortcut.weight"].reshape(cout, cin)
                out[name + ".nin.w"], out[name + ".nin.wt"], out[name + ".nin.b"] = ws, ws.t(), p[name + ".nin_shortcut.bias"]
        elif kind == "down":
            conv(name, p[name + ".weight"], p[name + ".bias"], stride=2)
        elif kind == "attn":
            out[name + ".norm.weight"], out[name + ".norm.bias"] = p[name + ".norm.weight"], p[name + ".norm.bias"]
            for n in ("q", "k", "v", "proj_out"):
                wm = p[f"{name}.{n}.weight"].reshape(cout, cin)
                out[f"{name}.{n}.w"], out[f"{name}.{n}.wt"], out[f"{name}.{n}.b"] = wm, wm.t(), p[f"{name}.{n}.bias"]
        elif kind == "out":
            out[name + ".norm_out.weight"], out[name + ".norm_out.bias"] = p[name + ".norm_out.weight"], p[name + ".norm_out.bias"]
            wq = p["quant_conv.weight"].float().reshape(p["quant_conv.weight"].shape[0], -1)
            wc = p[name + ".conv_out.weight"].float()
            conv(name + ".conv_out_quant", torch.einsum("om,mikl->oikl", wq, wc),
                 wq @ p[name + ".conv_out.bias"].float() + p["quant_conv.bias"].float())
    return out
