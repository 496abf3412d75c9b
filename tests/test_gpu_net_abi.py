"""The guidance forward driven through the bare C ABI (include/asd_hip.h: asd_unet_*, asd_vae_enc_*) with nothing but ctypes and
device buffers — what a non-Python host would do: create -> read the weight table -> bind -> workspace -> fwd (-> bwd)."""
import ctypes as C
import os
import zlib

import numpy as np
import pytest
import torch

from golden_util import GOLDEN_DIR

pytestmark = pytest.mark.gpu


def rnd(name, shape, seed=0):
    g = torch.Generator().manual_seed((seed * 1_000_003 + zlib.crc32(name.encode())) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g)


def _rel(got, ref):
    got, ref = got.double().cpu(), ref.double().cpu()
    return float((got - ref).norm() / ref.norm()), float((got - ref).abs().max() / ref.abs().max())


def _bind(lib, kind, h, packed):
    from scaledreamer_amd._lib import WeightInfo, check, i32

    n = getattr(lib, f"asd_{kind}_num_weights")(h)
    info, keep, ptrs = WeightInfo(), [], (C.c_void_p * n)()
    for i in range(n):
        check(getattr(lib, f"asd_{kind}_weight_info")(h, i32(i), C.byref(info)))
        t = packed[info.name.decode()].to(device="cuda", dtype=torch.float16).contiguous()
        assert t.numel() == info.rows * info.cols
        keep.append(t)
        ptrs[i] = t.data_ptr()
    check(getattr(lib, f"asd_{kind}_bind_weights")(h, ptrs, i32(n)))
    return keep


def test_asd_unet_fwd_through_ctypes_matches_the_reference_golden():
    from scaledreamer_amd._lib import check, i32, lib
    from scaledreamer_amd.diffusion import weights as W
    from scaledreamer_amd.diffusion.engine import unet_desc

    l = lib()
    g = dict(np.load(os.path.join(GOLDEN_DIR, "diffusion_unet_small.npz")))       # reference UNetModel, reduced width (head_dim 32 -> use the 64 one)
    cfg = W.UNetConfig(model_channels=128, num_head_channels=64, context_dim=128)
    from oracle import diffusion_ref as D

    layout = W.unet_layout(cfg)
    p = W.gen_params(layout[0], seed=21)
    B, hw, n_ctx = 3, 32, 77
    x, ctx, t = rnd("in.x", (B, 4, hw, hw), 21), rnd("in.context", (B, n_ctx, 128), 21), torch.tensor([815.0, 20.0, 999.0])
    with torch.no_grad():
        ref = D.unet_forward(p, layout, cfg, x, t, ctx)
    h = C.c_void_p()
    check(l.asd_unet_create(C.byref(unet_desc(cfg)), C.byref(h)))
    keep = _bind(l, "unet", h, W.pack_unet(p, cfg))
    ctx_stride = (n_ctx + 7) // 8 * 8
    xin = torch.zeros(B, hw, hw, 32, device="cuda", dtype=torch.float16)
    xin[..., :4] = x.permute(0, 2, 3, 1)
    cin = torch.zeros(B, ctx_stride, 128, device="cuda", dtype=torch.float16)
    cin[:, :n_ctx] = ctx
    tin, eps = t.cuda(), torch.empty(B, hw, hw, 4, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tune in (1, 0):       # first call times the GEMM shapes that have no plan, second runs on the recorded plans
        nb = l.asd_unet_workspace_bytes(h, i32(B), i32(hw), i32(hw), i32(n_ctx), i32(1), i32(tune))
        assert nb > 0
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        check(l.asd_unet_fwd(h, C.c_void_p(xin.data_ptr()), C.c_void_p(tin.data_ptr()), C.c_void_p(cin.data_ptr()), None, i32(B), i32(hw), i32(hw),
                             i32(n_ctx), i32(1), C.c_void_p(ws.data_ptr()), C.c_int64(nb), C.c_void_p(eps.data_ptr()), i32(tune), st))
        l2, mx = _rel(eps.permute(0, 3, 1, 2), ref)
        assert l2 < 1e-2 and mx < 1e-2, (tune, l2, mx)
    # error behaviour: too small a workspace, camera passed to a UNet without camera conditioning
    assert l.asd_unet_fwd(h, C.c_void_p(xin.data_ptr()), C.c_void_p(tin.data_ptr()), C.c_void_p(cin.data_ptr()), None, i32(B), i32(hw), i32(hw), i32(n_ctx),
                          i32(1), C.c_void_p(ws.data_ptr()), C.c_int64(1024), C.c_void_p(eps.data_ptr()), i32(0), st) != 0
    assert b"too small" in l.asd_last_error()
    assert l.asd_unet_fwd(h, C.c_void_p(xin.data_ptr()), C.c_void_p(tin.data_ptr()), C.c_void_p(cin.data_ptr()), C.c_void_p(cin.data_ptr()), i32(B), i32(hw),
                          i32(hw), i32(n_ctx), i32(1), C.c_void_p(ws.data_ptr()), C.c_int64(nb), C.c_void_p(eps.data_ptr()), i32(0), st) != 0
    torch.cuda.synchronize()
    l.asd_unet_destroy(h)
    del keep, g


def test_asd_vae_enc_fwd_bwd_through_ctypes_matches_the_reference_golden():
    from scaledreamer_amd._lib import check, i32, lib
    from scaledreamer_amd.diffusion import weights as W
    from scaledreamer_amd.diffusion.vae_hip import vae_desc

    l = lib()
    g = dict(np.load(os.path.join(GOLDEN_DIR, "diffusion_vae_small.npz")))
    ch, nrb, zc = (int(v) for v in g["cfg"])
    cfg = W.VAEConfig(ch=ch, num_res_blocks=nrb, z_channels=zc, ch_mult=tuple(int(v) for v in g["ch_mult"]))
    seed, B, res = int(g["seed"]), int(g["batch"]), int(g["res"])
    h = C.c_void_p()
    check(l.asd_vae_enc_create(C.byref(vae_desc(cfg)), C.byref(h)))
    keep = _bind(l, "vae_enc", h, W.pack_vae_encoder(W.gen_params(W.vae_encoder_layout(cfg)[0], seed), cfg))
    img = torch.tanh(rnd("in.img", (B, 3, res, res), seed))
    x = torch.zeros(B, res, res, 32, device="cuda", dtype=torch.float16)
    x[..., :3] = img.permute(0, 2, 3, 1)
    gm = rnd("in.gmoments", (B, 8, res // 8, res // 8), seed).permute(0, 2, 3, 1).contiguous().cuda()
    m = torch.empty(B, res // 8, res // 8, 8, device="cuda")
    dx = torch.empty(B, res, res, 32, device="cuda", dtype=torch.float16)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tune in (1, 0):
        nb = l.asd_vae_enc_workspace_bytes(h, i32(B), i32(res), i32(res), i32(tune))
        ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
        assert l.asd_vae_enc_bwd(h, C.c_void_p(gm.data_ptr()), i32(B), i32(res), i32(res), C.c_void_p(ws.data_ptr()), C.c_int64(nb),
                                 C.c_void_p(dx.data_ptr()), i32(tune), st) != 0          # no forward has run on this workspace yet
        check(l.asd_vae_enc_fwd(h, C.c_void_p(x.data_ptr()), i32(B), i32(res), i32(res), C.c_void_p(ws.data_ptr()), C.c_int64(nb),
                                C.c_void_p(m.data_ptr()), i32(tune), st))
        check(l.asd_vae_enc_bwd(h, C.c_void_p(gm.data_ptr()), i32(B), i32(res), i32(res), C.c_void_p(ws.data_ptr()), C.c_int64(nb),
                                C.c_void_p(dx.data_ptr()), i32(tune), st))
        l2, mx = _rel(m.permute(0, 3, 1, 2), torch.from_numpy(g["moments"]))
        assert l2 < 1e-2 and mx < 1e-2, (tune, l2, mx)
        l2, mx = _rel(dx[..., :3].permute(0, 3, 1, 2), torch.from_numpy(g["grad_x_sub"]))
        assert l2 < 1e-2 and mx < 1e-2, (tune, l2, mx)
        assert float(dx[..., 3:].abs().max()) == 0.0                                      # gradient of the zero padding channels
    torch.cuda.synchronize()
    l.asd_vae_enc_destroy(h)
    del keep
