"""GPU parity of the diffusion kernels (called through the C ABI) against a plain PyTorch fp32 reference of the
same op on the same fp16-rounded inputs.  Tolerance: fp16 output rounding (rel 2^-10) + fp32 accumulation order;
north_star allows 1e-2 rel on the UNet eps, the single ops are held to ~2e-3."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _rand(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half().cuda()


def _close(got, want, tol=3e-3):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert err <= tol * max(ref, 1.0), f"max abs err {err} vs ref max {ref}"


@pytest.mark.parametrize("M,N,K", [(20480, 320, 320), (5120, 640, 1280), (385, 1280, 1024), (320, 1280, 5120), (5, 1280, 320),
                                   (1280, 2560, 320), (4096, 128, 64), (130, 68, 96)])
def test_gemm_bias_rowbias_residual_silu(M, N, K):
    from scaledreamer_amd.diffusion import hip_ops as H

    a, w = _rand(M, K, seed=1), _rand(N, K, scale=K ** -0.5, seed=2)
    bias, res = _rand(N, seed=3), _rand(M, N, seed=4)
    groups = 5 if M % 5 == 0 else 1
    rb = _rand(groups, N, seed=5)
    ref = a.float() @ w.float().T + bias.float() + rb.float().repeat_interleave(M // groups, 0)
    ref = F.silu(ref) + res.float()
    got = H.gemm(a, w, bias=bias, row_bias=rb, rows_per_group=M // groups, residual=res, act=1)
    _close(got, ref)
    for sk in (1, 3):
        _close(H.gemm(a, w, split_k=sk), a.float() @ w.float().T)
    _close(H.gemm(a, w, out_f32=True), a.float() @ w.float().T, tol=1e-3)


def test_gemm_strided_views():
    from scaledreamer_amd.diffusion import hip_ops as H

    big = _rand(1024, 960, seed=6)
    a = big[:, 320:640]                      # lda = 960
    w = _rand(192, 320, scale=0.05, seed=7)
    outbuf = torch.zeros(1024, 576, dtype=torch.float16, device="cuda")
    H.gemm(a, w, out=outbuf[:, 192:384])
    _close(outbuf[:, 192:384], a.float() @ w.float().T)
    assert float(outbuf[:, :192].abs().max()) == 0 and float(outbuf[:, 384:].abs().max()) == 0


@pytest.mark.parametrize("B,H_,W_,Cin,Cout,stride,pad,up", [
    (5, 64, 64, 320, 320, 1, 1, False), (2, 32, 32, 640, 1280, 1, 1, False), (5, 8, 8, 2560, 1280, 1, 1, False),
    (2, 64, 64, 320, 320, 2, 1, False), (2, 16, 16, 1280, 1280, 1, 1, True), (1, 64, 64, 4, 320, 1, 1, False),
    (1, 64, 64, 320, 4, 1, 1, False), (1, 33, 47, 64, 96, 2, 0, False), (1, 7, 5, 32, 64, 1, 1, False)])
def test_conv3x3_variants(B, H_, W_, Cin, Cout, stride, pad, up):
    from scaledreamer_amd.diffusion import hip_ops as H

    x = _rand(B, Cin, H_, W_, seed=8)
    w = _rand(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=9)
    bias = _rand(Cout, seed=10)
    xin = F.interpolate(x.float(), scale_factor=2, mode="nearest") if up else x.float()
    if pad == 0 and stride == 2:  # VAE Downsample: asymmetric (0,1,0,1) zero pad (model.py:80-85)
        xin = F.pad(xin, (0, 1, 0, 1))
        ref = F.conv2d(xin, w.float(), bias.float(), stride=2, padding=0)
    else:
        ref = F.conv2d(xin, w.float(), bias.float(), stride=stride, padding=pad)
    cp = (Cin + 31) // 32 * 32
    xn = torch.zeros(B, H_, W_, cp, dtype=torch.float16, device="cuda")
    xn[..., :Cin] = x.permute(0, 2, 3, 1)
    wp = H.pack_conv3x3_weight(w)
    out_hw = (ref.shape[2], ref.shape[3])
    got = H.conv3x3(xn, wp, bias=bias, stride=stride, pad=pad, upsample=up, out_hw=out_hw)
    _close(got.permute(0, 3, 1, 2), ref)


@pytest.mark.parametrize("B,hw,cin,cout", [(5, 32, 640, 640), (5, 16, 1280, 1280), (5, 8, 1280, 1280), (2, 16, 96, 64), (1, 8, 40, 320), (3, 24, 64, 128)])
@pytest.mark.parametrize("tile,sk", [(0, 1), (12, 1), (2, 3), (16, 1), (1, 2)])
def test_upsample_conv_parity_form(B, hw, cin, cout, tile, sk):
    """Upsample.conv (nearest 2x, then 3x3) as four 2x2 convolutions over the low-resolution image with pre-summed taps (asd_gemm_args
    upsample = 3) against F.interpolate + F.conv2d, and against the nine-tap fused form (upsample = 1) of the same kernel; tiles whose
    rows do not divide a parity's rows are replaced by the library (the result must not depend on the request)"""
    from scaledreamer_amd.diffusion import hip_ops as H

    x = _rand(B, cin, hw, hw, seed=21)
    w = _rand(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=22)
    bias, res = _rand(cout, seed=23), _rand(B, 2 * hw, 2 * hw, cout, seed=24)
    ref = F.conv2d(F.interpolate(x.float(), scale_factor=2, mode="nearest"), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1) + res.float()
    cp = (cin + 31) // 32 * 32
    xn = torch.zeros(B, hw, hw, cp, dtype=torch.float16, device="cuda")
    xn[..., :cin] = x.permute(0, 2, 3, 1)
    if sk > 1 and (4 * cp) // sk < 256:
        pytest.skip("split deeper than the reduction")
    got = H.conv3x3(xn, H.pack_upsample_conv3x3_weight(w), bias=bias, residual=res.view(-1, cout), upsample=3, tile_cfg=tile + 1, split_k=sk)
    _close(got, ref)
    nine = H.conv3x3(xn, H.pack_conv3x3_weight(w), bias=bias, residual=res.view(-1, cout), upsample=1, split_k=1)
    assert float((got.float() - nine.float()).abs().max()) <= 4e-3 * float(ref.abs().max())


@pytest.mark.parametrize("tile", list(range(25)) + [26, 27, 28])      # 25 (weight streaming, 8x8 images only) has its own test below
@pytest.mark.parametrize("B,HW,Cin,Cout,sk", [(2, 32, 128, 320, 1), (1, 16, 320, 256, 2), (3, 48, 64, 640, 1), (2, 64, 96, 128, 1), (1, 32, 160, 640, 5),
                                              (1, 64, 32, 128, 1)])
def test_every_tile_configuration_computes_the_same_convolution(tile, B, HW, Cin, Cout, sk):
    """all GEMM tile shapes (implicit GEMM 128x64 ... 256x320, 320x128) and the LDS-window kernels (16x16-pixel patches x 64 /
    128 channels) against F.conv2d, with bias + residual, with and without split-K"""
    import ctypes as C

    from scaledreamer_amd._lib import lib
    from scaledreamer_amd.diffusion import hip_ops as H

    bn = H.TILE_BN[tile]
    if bn != 64 and Cout % bn != 0:
        pytest.skip("tile does not divide N")
    if tile in H.WINDOW_TILES and ((sk > 1 and sk > Cin // 64) or Cin % (32 if tile in H.PP_TILES else 64)):
        pytest.skip("more splits than channel chunks / channels not a multiple of 64 (ping-pong kernel: 32, e.g. the VAE's padded RGB input)")
    if tile in H.PP_TILES and (Cout % bn or HW % (H.TILE_BM[tile] // 16)):
        pytest.skip("ping-pong window kernel: whole N tiles and whole patches only")
    x = _rand(B, Cin, HW, HW, seed=8)
    w = _rand(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=9)
    bias, res = _rand(Cout, seed=10), _rand(B, HW, HW, Cout, seed=11)
    ref = F.conv2d(x.float(), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1) + res.float()
    xn, wp = x.permute(0, 2, 3, 1).contiguous(), H.pack_conv3x3_weight(w)
    lib().asd_gemm_force_tile(C.c_int32(tile))
    try:
        got = H.conv3x3(xn, wp, bias=bias, residual=res.view(-1, Cout), split_k=sk)
    finally:
        lib().asd_gemm_force_tile(C.c_int32(-1))
    _close(got, ref)


@pytest.mark.parametrize("B,Cin,Cout,sk", [(5, 1280, 1280, 10), (5, 2560, 1280, 10), (5, 1280, 1280, 5), (3, 256, 128, 2), (1, 64, 64, 2), (4, 320, 192, 5),
                                           (2, 128, 64, 4), (5, 640, 1280, 20)])
def test_weight_streaming_convolution_of_the_8x8_level(B, Cin, Cout, sk):
    """tile configuration 25 (csrc/gemm_ws.hip: all M <= 320 rows x 64 channels x one channel slice per block, raw 10 x 10 activation
    windows in LDS) against F.conv2d in fp32 and, bit for bit in the slabs' sum order aside, against the implicit-GEMM tile on the same
    split; with bias + time-embedding row bias + residual through the split-K reduction, and with the producer-applied GroupNorm"""
    from scaledreamer_amd.diffusion import hip_ops as H

    x = _rand(B, Cin, 8, 8, seed=8)
    w = _rand(Cout, Cin, 3, 3, scale=(9 * Cin) ** -0.5, seed=9)
    bias, res, temb = _rand(Cout, seed=10), _rand(B, 8, 8, Cout, seed=11), _rand(B, Cout, seed=12)
    ref = F.conv2d(x.float(), w.float(), bias.float(), padding=1).permute(0, 2, 3, 1) + res.float() + temb.float()[:, None, None, :]
    xn, wp = x.permute(0, 2, 3, 1).contiguous(), H.pack_conv3x3_weight(w)
    kw = dict(bias=bias, residual=res.view(-1, Cout), row_bias=temb, rows_per_group=64, split_k=sk)
    got = H.conv3x3(xn, wp, tile_cfg=H.WS_TILE + 1, **kw)
    _close(got, ref)
    other = H.conv3x3(xn, wp, tile_cfg=13, **kw)
    assert float((got.float() - other.float()).abs().max()) <= 2e-3 * float(ref.abs().max())
    if Cout % 128 == 0:
        gamma, beta = (_rand(Cout, seed=5) * 0.1 + 1).half(), (_rand(Cout, seed=6) * 0.1).half()
        c, y, st = H.conv3x3(xn, wp, tile_cfg=H.WS_TILE + 1, gn_rows=64, gn_apply=dict(gamma=gamma, beta=beta, eps=1e-5, silu=True), **kw)
        assert y is not None and torch.equal(c.reshape(-1), got.reshape(-1))
        want = H.groupnorm(c.view(B, 64, Cout), gamma, beta, 1e-5, True)
        assert float((y.view_as(want).float() - want.float()).abs().max()) <= 2e-3


@pytest.mark.parametrize("B,HW,C1,C2,silu", [(5, 4096, 320, 0, True), (5, 64, 1280, 1280, True), (2, 1024, 640, 320, False),
                                             (3, 256, 1920, 0, True), (1, 77, 32, 0, False)])
def test_groupnorm(B, HW, C1, C2, silu):
    from scaledreamer_amd.diffusion import hip_ops as H

    x1 = _rand(B, HW, C1, scale=2.0, seed=11) + 0.5
    x2 = _rand(B, HW, C2, scale=0.5, seed=12) if C2 else None
    gam, bet = _rand(C1 + C2, seed=13) * 0.1 + 1, _rand(C1 + C2, seed=14) * 0.1
    x = torch.cat([x1, x2], -1) if C2 else x1
    ref = F.group_norm(x.float().permute(0, 2, 1), 32, gam.float(), bet.float(), 1e-5).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    _close(H.groupnorm(x1, gam, bet, 1e-5, silu, x2), ref, tol=4e-3)


def test_layernorm_geglu_silu_concat_timestep():
    from scaledreamer_amd.diffusion import hip_ops as H

    for rows, c in [(20480, 320), (1280, 1280), (77, 640)]:
        x, g, b = _rand(rows, c, scale=3, seed=15), _rand(c, seed=16) * 0.1 + 1, _rand(c, seed=17) * 0.1
        _close(H.layernorm(x, g, b), F.layer_norm(x.float(), (c,), g.float(), b.float()))
    h = _rand(1000, 2 * 1280, seed=18)
    a, gate = h.float().chunk(2, -1)
    _close(H.geglu(h), a * F.gelu(gate))
    x = _rand(5, 1280, seed=19)
    _close(H.silu(x), F.silu(x.float()))
    a, b = _rand(300, 640, seed=20), _rand(300, 320, seed=21)
    assert torch.equal(H.concat(a, b), torch.cat([a, b], -1))
    t = torch.tensor([1.0, 500.0, 815.0, 999.0, 904.0], device="cuda")
    half = 160
    freqs = torch.exp(-np.log(10000) * torch.arange(half, dtype=torch.float32, device="cuda") / half)
    ref = torch.cat([torch.cos(t[:, None] * freqs), torch.sin(t[:, None] * freqs)], -1)
    _close(H.timestep_embedding(t, 320), ref, tol=2e-3)


@pytest.mark.parametrize("B,heads,lq,lk,lk_stride", [(5, 5, 4096, 4096, 4096), (2, 10, 1024, 1024, 1024), (5, 20, 64, 64, 64),
                                                     (5, 5, 4096, 77, 80), (3, 20, 256, 77, 80), (1, 2, 100, 50, 56)])
def test_attention(B, heads, lq, lk, lk_stride):
    from scaledreamer_amd.diffusion import hip_ops as H

    C_ = heads * 64
    q = _rand(B * lq, C_, seed=22)
    k = _rand(B * lk_stride, C_, seed=23)
    v = _rand(B * lk_stride, C_, seed=24)
    vT = v.t().contiguous()
    got = H.attention(q, k, vT, B, heads, lq, lk, lk_stride)
    qf = q.float().view(B, lq, heads, 64).transpose(1, 2)
    kf = k.float().view(B, lk_stride, heads, 64)[:, :lk].transpose(1, 2)
    vf = v.float().view(B, lk_stride, heads, 64)[:, :lk].transpose(1, 2)
    ref = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(B * lq, C_)
    _close(got, ref, tol=3e-3)


@pytest.mark.parametrize("lq,lk,wide", [(4096, 4096, True), (512, 1024, False)])
def test_attention_running_maximum_that_grows_late_and_by_every_amount(lq, lk, wide):
    """The kernel defers the running maximum (no accumulator rescale while no query's tile maximum exceeds its running maximum by more than
    2^8 in the exponent's log2 domain).  The branch is data dependent, so it is forced: chosen keys are made to score far above (spikes of
    +3 ... +60 in the logit, planted at key tiles 1, 5 and the last one), just below and just above the deferral, against individual queries
    and against whole query blocks — every path (defer, grow, grow after defer, spike in the last tile) against torch's fp32 attention."""
    from scaledreamer_amd.diffusion import hip_ops as H

    B, heads = 2, 3
    C_ = heads * 64
    q = _rand(B * lq, C_, seed=31).float()
    k = _rand(B * lk, C_, seed=32).float()
    v = _rand(B * lk, C_, seed=33)
    kv = k.view(B, lk, heads, 64)
    qv = q.view(B, lq, heads, 64)
    g = torch.Generator().manual_seed(5)
    for key, boost in ((70, 3.0), (64 * 5 + 3, 9.0), (64 * 5 + 40, 20.0), (lk - 2, 60.0), (lk - 1, 7.9), (lk // 2, 8.1)):
        # make key `key` point along a few queries: q . k / 8 grows by about `boost` for them
        rows = torch.randint(0, lq, (5,), generator=g).tolist() + list(range(128, 128 + 64))
        for b in range(B):
            for h in range(heads):
                qq = qv[b, rows, h].mean(0)
                kv[b, key, h] = kv[b, key, h] + qq / qq.norm() ** 2 * 8.0 * boost * (1.0 if (b + h) % 2 == 0 else 0.5)
    qh, kh = q.half(), k.half()
    got = H.attention(qh, kh, v.t().contiguous(), B, heads, lq, lk, lk)
    qf = qh.float().view(B, lq, heads, 64).transpose(1, 2)
    kf = kh.float().view(B, lk, heads, 64).transpose(1, 2)
    vf = v.float().view(B, lk, heads, 64).transpose(1, 2)
    ref = F.scaled_dot_product_attention(qf, kf, vf).transpose(1, 2).reshape(B * lq, C_)
    _close(got, ref, tol=3e-3)


@pytest.mark.parametrize("rows,cols", [(64, 4096), (33, 1024), (7, 8192), (5, 264)])
def test_softmax_and_its_gradient(rows, cols):
    from scaledreamer_amd.diffusion import hip_ops as H

    x = _rand(rows, cols, scale=6.0, seed=1)
    scale = 512 ** -0.5
    ref = torch.softmax(x.float() * scale, dim=-1)
    p = H.softmax(x, scale)
    _close(p, ref, tol=1e-3)
    assert abs(p.float().sum(-1) - 1).max().item() < 5e-3
    # row-strided input (a column slice of a wider matrix)
    wide = _rand(rows, cols + 16, scale=6.0, seed=2)
    _close(H.softmax(wide[:, 8:8 + cols], scale), torch.softmax(wide[:, 8:8 + cols].float() * scale, -1), tol=1e-3)
    dp = _rand(rows, cols, seed=3)
    pf = p.float()
    ref_ds = scale * pf * (dp.float() - (dp.float() * pf).sum(-1, keepdim=True))
    ds = H.softmax_bwd(p, dp, scale)
    err = (ds.float() - ref_ds).abs().max().item()
    assert err <= 2e-3 * ref_ds.abs().max().item() + 1e-6


class _AttnFn(torch.autograd.Function):
    """the leaf-op sequence csrc/net.hip enqueues for the VAE mid-block attention (AttnBlock, model.py:195-224): per image
    S = Q K^T -> P = softmax(S / sqrt(C)) -> O = P V, and the four GEMMs + softmax-gradient kernel of its input gradients."""

    @staticmethod
    def forward(ctx, q, k, v, B):
        from scaledreamer_amd.diffusion import hip_ops as H

        L, C_ = q.shape[0] // B, q.shape[1]
        scale = float(C_) ** -0.5
        o = torch.empty_like(q)
        ps = []
        for b in range(B):
            r = slice(b * L, (b + 1) * L)
            p = H.softmax(H.gemm(q[r], k[r]), scale)
            H.gemm(p, H.transpose(v[r]), out=o[r])
            ps.append(p)
        ctx.save_for_backward(q, k, v, *ps)
        ctx.B, ctx.scale = B, scale
        return o

    @staticmethod
    def backward(ctx, do):
        from scaledreamer_amd.diffusion import hip_ops as H

        q, k, v, *ps = ctx.saved_tensors
        B, scale = ctx.B, ctx.scale
        L = q.shape[0] // B
        do = do.contiguous()
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for b in range(B):
            r = slice(b * L, (b + 1) * L)
            p = ps[b]
            H.gemm(H.transpose(p), H.transpose(do[r]), out=dv[r])
            ds = H.softmax_bwd(p, H.gemm(do[r], v[r]), scale)
            H.gemm(ds, H.transpose(k[r]), out=dq[r])
            H.gemm(H.transpose(ds), H.transpose(q[r]), out=dk[r])
        return dq, dk, dv, None


def test_vae_single_head_attention_matches_sdpa_forward_and_backward():
    B, L, C_ = 2, 1024, 512
    q, k, v = (_rand(B * L, C_, seed=s).requires_grad_(True) for s in (1, 2, 3))
    o = _AttnFn.apply(q, k, v, B)
    go = _rand(B * L, C_, seed=4)
    o.backward(go)
    qf, kf, vf = (t.detach().float().view(B, 1, L, C_).requires_grad_(True) for t in (q, k, v))
    of = F.scaled_dot_product_attention(qf, kf, vf)
    of.backward(go.float().view(B, 1, L, C_))
    _close(o, of.reshape(B * L, C_), tol=4e-3)
    for got, want in ((q.grad, qf.grad), (k.grad, kf.grad), (v.grad, vf.grad)):
        want = want.reshape(B * L, C_)
        err = (got.float() - want).abs().max().item()
        assert err <= 1e-2 * want.abs().max().item(), f"{err} vs {want.abs().max().item()}"


@pytest.mark.parametrize("M,C_,K", [(5120, 2560, 640), (333, 1280, 320), (1280, 5120, 1280)])
def test_gemm_with_fused_geglu_epilogue(M, C_, K):
    from scaledreamer_amd.diffusion import hip_ops as H

    a, w, b = _rand(M, K, seed=1), _rand(2 * C_, K, scale=K ** -0.5, seed=2), _rand(2 * C_, seed=3)
    h = a.float() @ w.float().T + b.float()
    ref = h[:, :C_] * F.gelu(h[:, C_:])
    wp, bp = H.pack_geglu_weight(w, b)
    _close(H.gemm(a, wp, bias=bp, act=2), ref)
    _close(H.geglu(H.gemm(a, w, bias=b)), ref)


@pytest.mark.parametrize("B,hw,cin,cout,tile", [(2, 64, 128, 128, 11), (2, 64, 128, 128, 14), (1, 64, 64, 320, 10), (1, 64, 64, 320, 13), (1, 32, 128, 256, 8),
                                                 (1, 32, 128, 256, 9), (3, 16, 64, 640, 2), (3, 16, 64, 640, 0), (2, 32, 96, 320, 12), (2, 32, 64, 320, 4),
                                                 (1, 64, 128, 256, 5), (5, 8, 64, 320, 12), (5, 8, 64, 320, 16), (2, 16, 128, 256, 19), (3, 16, 64, 640, 15), (5, 8, 128, 1280, 12),
                                                 (5, 16, 128, 1280, 12), (2, 64, 128, 128, 20), (1, 32, 128, 256, 21), (2, 32, 64, 320, 22), (1, 64, 64, 128, 23), (2, 16, 64, 320, 24)])
def test_groupnorm_statistics_from_the_producers_epilogue(B, hw, cin, cout, tile):
    """asd_gemm_args.gn_partials: the conv / GEMM that stores a tensor also leaves its per-group sums; GroupNorm from those records ==
    GroupNorm with its own statistics pass (same kernel afterwards; the sums are accumulated in another order)"""
    from scaledreamer_amd.diffusion import hip_ops as H

    x = _rand(B, hw, hw, cin, seed=1)
    w = H.pack_conv3x3_weight(_rand(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=2))
    bias, res = _rand(cout, seed=3), _rand(B * hw * hw, cout, seed=4)
    gamma, beta = (_rand(cout, seed=5) * 0.1 + 1).half(), (_rand(cout, seed=6) * 0.1).half()
    y, rec, nrec = H.conv3x3(x, w, bias=bias, residual=res, tile_cfg=tile + 1, split_k=1, gn_rows=hw * hw)
    bm = H.TILE_BM[tile]
    if tile in H.WINDOW_TILES or (hw * hw) % bm == 0:
        assert nrec > 0, "this plan can produce the records"
    if nrec == 0:
        pytest.skip("tile rows do not divide the rows of a batch element")
    assert torch.equal(y, H.conv3x3(x, w, bias=bias, residual=res, tile_cfg=tile + 1, split_k=1).view_as(y))   # the stored tensor is unchanged
    yv = y.view(B, hw * hw, cout)
    want, wstats = H.groupnorm(yv, gamma, beta, 1e-5, True, return_stats=True)
    got, gstats = H.groupnorm_apply(yv, gamma, beta, 1e-5, True, rec)
    ref = yv.float()
    sums = torch.stack([ref.view(B, -1, 32, cout // 32).sum(dim=(1, 3)), (ref ** 2).view(B, -1, 32, cout // 32).sum(dim=(1, 3))], -1).reshape(B * 64)
    torch.testing.assert_close(gstats, sums, rtol=2e-4, atol=1e-2)
    torch.testing.assert_close(gstats, wstats, rtol=2e-4, atol=1e-2)
    assert float((got.float() - want.float()).abs().max()) <= 2e-3
    # a plain GEMM producer (transformer proj_out) and a split-K launch (no records: the consumer falls back)
    a, wl = _rand(B * hw * hw, 64, seed=7), _rand(cout, 64, scale=0.125, seed=8)
    yl, rl, nl = H.gemm(a, wl, bias=bias, gn_rows=hw * hw)
    if nl:
        g2, _ = H.groupnorm_apply(yl.view(B, hw * hw, cout), gamma, beta, 1e-6, False, rl)
        assert float((g2.float() - H.groupnorm(yl.view(B, hw * hw, cout), gamma, beta, 1e-6, False).float()).abs().max()) <= 2e-3
    # a split-K launch: the records come from the split-K epilogue (64 x 64 blocks) when a batch element is small enough for them
    # to be folded by the apply kernel (<= 96 records), otherwise there are none and the consumer runs its own statistics pass
    y2, r2, n2 = H.conv3x3(x, w, bias=bias, residual=res, tile_cfg=1, split_k=2, gn_rows=hw * hw)
    want_n2 = (hw * hw // 64) * (cout // 64) if (hw * hw) % 64 == 0 and cout % 64 == 0 else 0
    if want_n2 > 96:
        want_n2 = 0
    assert n2 == want_n2 and (r2 is None) == (n2 == 0)
    if n2:
        y2v = y2.view(B, hw * hw, cout)
        assert torch.equal(y2v, H.conv3x3(x, w, bias=bias, residual=res, tile_cfg=1, split_k=2).view_as(y2v))
        w2, ws2 = H.groupnorm(y2v, gamma, beta, 1e-5, True, return_stats=True)
        g2, gs2 = H.groupnorm_apply(y2v, gamma, beta, 1e-5, True, r2)
        torch.testing.assert_close(gs2, ws2, rtol=2e-4, atol=1e-2)
        assert float((g2.float() - w2.float()).abs().max()) <= 2e-3


@pytest.mark.parametrize("B,hw,cin,cout,split,silu,conv", [(5, 8, 128, 1280, 4, True, True), (5, 16, 64, 1280, 2, True, True), (5, 16, 64, 640, 3, False, True),
                                                           (2, 32, 128, 640, 2, True, True), (3, 8, 64, 128, 2, True, True), (5, 16, 320, 1280, 5, False, False),
                                                           (1, 8, 64, 2560, 2, True, True)])
def test_groupnorm_applied_by_the_split_k_reduction(B, hw, cin, cout, split, silu, conv):
    """asd_gemm_args.gn_apply: a split-K launch whose only consumer is GroupNorm(32)(+SiLU) lets its reduction kernel own whole
    (batch element, group) blocks — it stores C, takes the statistics of the stored values and writes the normalised tensor in the same
    launch.  C must be bit-identical to the plain split-K launch, y must equal GroupNorm of that C (the stand-alone kernels and the
    torch fp32 op), the statistics the sums of the stored values."""
    from scaledreamer_amd.diffusion import hip_ops as H

    rows = hw * hw
    bias, res = _rand(cout, seed=3), _rand(B * rows, cout, seed=4)
    temb = _rand(B, cout, seed=9)
    gamma, beta = (_rand(cout, seed=5) * 0.1 + 1).half(), (_rand(cout, seed=6) * 0.1).half()
    eps = 1e-5 if silu else 1e-6
    spec = dict(gamma=gamma, beta=beta, eps=eps, silu=silu)
    if conv:
        x = _rand(B, hw, hw, cin, seed=1)
        w = H.pack_conv3x3_weight(_rand(cout, cin, 3, 3, scale=(9 * cin) ** -0.5, seed=2))
        kw = dict(bias=bias, residual=res, row_bias=temb, rows_per_group=rows, tile_cfg=13, split_k=split)
        c, y, st = H.conv3x3(x, w, gn_rows=rows, gn_apply=spec, **kw)
        plain = H.conv3x3(x, w, **kw).view(B * rows, cout)
    else:
        a, w = _rand(B * rows, cin, seed=1), _rand(cout, cin, scale=cin ** -0.5, seed=2)
        kw = dict(bias=bias, residual=res, tile_cfg=13, split_k=split)
        c, y, st = H.gemm(a, w, gn_rows=rows, gn_apply=spec, **kw)
        plain = H.gemm(a, w, **kw)
    cg = cout // 32
    if cg % 4 or (rows * (cg // 4) + 1023) // 1024 > 3:
        assert y is None
        pytest.skip("group width / rows outside the fused reduction's range: the caller runs its own GroupNorm")
    assert y is not None and st is not None
    c = c.view(B * rows, cout)
    assert torch.equal(c, plain)
    cv = c.view(B, rows, cout)
    want, wstats = H.groupnorm(cv, gamma, beta, eps, silu, return_stats=True)
    torch.testing.assert_close(st.view(-1), wstats.view(-1), rtol=2e-4, atol=1e-2)
    assert float((y.view_as(want).float() - want.float()).abs().max()) <= 2e-3
    ref = F.group_norm(cv.float().permute(0, 2, 1), 32, gamma.float(), beta.float(), eps).permute(0, 2, 1)
    if silu:
        ref = F.silu(ref)
    _close(y.view(B, rows, cout), ref, tol=4e-3)


@pytest.mark.parametrize("B,hw,c,tile,silu", [(2, 64, 128, 11, True), (1, 64, 128, 14, True), (2, 32, 256, 8, True), (1, 32, 128, 1, False),
                                              (3, 16, 320, 2, True), (2, 32, 64, 0, False), (2, 64, 128, 20, True), (1, 32, 256, 21, True), (1, 32, 128, 23, False)])
def test_groupnorm_backward_reductions_from_the_dgrad_epilogue(B, hw, c, tile, silu):
    """asd_gemm_args.gn_bwd_x: the launch that produces dy (the gradient reaching GroupNorm(x)[+SiLU]) also leaves that layer's two
    reductions {sum g, sum g*xhat}; the apply-only backward on those records == the backward with its own reduction pass"""
    from scaledreamer_amd.diffusion import hip_ops as H

    x = _rand(B, hw * hw, c, seed=1)
    gamma, beta = (_rand(c, seed=5) * 0.1 + 1).half(), (_rand(c, seed=6) * 0.1).half()
    _, stats = H.groupnorm(x, gamma, beta, 1e-6, silu, return_stats=True)
    up = _rand(B, hw, hw, c, seed=2)                                    # gradient w.r.t. the conv output
    w = H.pack_conv3x3_weight(_rand(c, c, 3, 3, scale=(9 * c) ** -0.5, seed=3))
    dx_add = _rand(B, hw * hw, c, seed=4)
    spec = dict(x=x, fstats=stats, gamma=gamma, beta=beta, eps=1e-6, silu=silu)
    dy, rec, nrec = H.conv3x3(up, w, tile_cfg=tile + 1, split_k=1, gn_rows=hw * hw, gn_bwd=spec)
    if tile in H.WINDOW_TILES or (hw * hw) % H.TILE_BM[tile] == 0:
        assert nrec > 0, "this plan can produce the records"
    if nrec == 0:
        pytest.skip("tile rows do not divide the rows of a batch element")
    assert torch.equal(dy, H.conv3x3(up, w, tile_cfg=tile + 1, split_k=1).view_as(dy))
    dyv = dy.view(B, hw * hw, c)
    want = H.groupnorm_bwd(x, dyv, gamma, beta, 1e-6, silu, stats, dx_add=dx_add)
    got = H.groupnorm_bwd_apply(x, dyv, gamma, beta, 1e-6, silu, stats, rec, dx_add=dx_add)
    # fp32 restatement of the two sums
    xf, cg, n = x.float().view(B, -1, 32, c // 32), c // 32, hw * hw * (c // 32)
    mean = xf.mean(dim=(1, 3), keepdim=True)
    rstd = (xf.var(dim=(1, 3), unbiased=False, keepdim=True) + 1e-6).rsqrt()
    xh = (xf - mean) * rstd
    gm, bt = gamma.float().view(1, 1, 32, cg), beta.float().view(1, 1, 32, cg)
    z = xh * gm + bt
    g = dyv.float().view(B, -1, 32, cg) * gm
    if silu:
        sg = torch.sigmoid(z)
        g = g * sg * (1 + z * (1 - sg))
    sums = torch.stack([g.sum(dim=(1, 3)), (g * xh).sum(dim=(1, 3))], -1)                        # [B, 32, 2]
    recsum = rec.view(B, nrec, 32, 2).sum(1) if rec.shape[-1] == 64 else None
    scale = float(g.abs().sum(dim=(1, 3)).max())
    assert float((recsum - sums).abs().max()) <= 2e-3 * scale + 1e-3
    ref = (g - (sums[..., 0].view(B, 1, 32, 1) + xh * sums[..., 1].view(B, 1, 32, 1)) / n) * rstd
    ref = ref.reshape(B, hw * hw, c) + dx_add.float()
    assert float((got.float() - ref).abs().max()) <= 4e-3 * max(1.0, float(ref.abs().max()))
    assert float((got.float() - want.float()).abs().max()) <= 2e-3 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("M,C_,N,tile", [(20480, 320, 640, 2), (5120, 640, 640, 12), (1280, 1280, 1280, 16), (320, 1280, 2560, 17), (333, 320, 320, 15),
                                          (1280, 1280, 2560, 19), (4096, 640, 1280, 5), (777, 320, 640, 0), (2048, 320, 320, 3)])
def test_layernorm_folded_into_the_consuming_gemm(M, C_, N, tile):
    """asd_gemm_args.ln_mode: LN(x) W^T = rstd * (x (gamma . W)^T - mean * rowsum(gamma . W)) + W beta with the row statistics reduced
    from the A fragments in the main loop (mode 1, every tile family incl. the k-group and ring variants) or read from the statistics
    a mode-1 launch left (mode 2: the V^T = W_v LN(x)^T form) — against torch's LayerNorm + matmul in fp32"""
    from scaledreamer_amd.diffusion import hip_ops as H
    from scaledreamer_amd.diffusion.weights import _ln_fold

    if N % H.TILE_BN[tile] and H.TILE_BN[tile] != 64:
        pytest.skip("tile does not divide N")
    x = (_rand(M, C_, seed=1).float() * 1.5 + _rand(M, 1, seed=2).float() * 2.0 + 0.7).half()      # rows with their own offsets and scales
    gamma, beta = (_rand(C_, seed=3).float() * 0.2 + 1.0), _rand(C_, seed=4).float() * 0.3
    w = _rand(N, C_, scale=C_ ** -0.5, seed=5)
    bias = _rand(N, seed=6)
    y = F.layer_norm(x.float(), (C_,), gamma.cuda(), beta.cuda(), 1e-5)
    ref = y @ w.float().t() + bias.float()
    w2, sc = _ln_fold(w, gamma.cuda(), beta.cuda())
    sc32 = sc.view(torch.float32).view(2, N).contiguous()
    stats = torch.zeros(M, 2, device="cuda")
    got = H.gemm(x, w2.contiguous(), bias=bias, tile_cfg=tile + 1, split_k=1, ln=dict(mode=1, sc=sc32, stats=stats))
    _close(got, ref)
    torch.testing.assert_close(stats[:, 0], x.float().mean(1), rtol=1e-3, atol=2e-3)
    torch.testing.assert_close(stats[:, 1], (x.float().var(1, unbiased=False) + 1e-5).rsqrt(), rtol=2e-3, atol=1e-4)
    # mode 2: the same product transposed — the normalised rows are the W operand, their statistics come from the launch above
    if M % 4 == 0:
        wv = _rand(256, C_, scale=C_ ** -0.5, seed=7)
        wv2, scv = _ln_fold(wv, gamma.cuda(), beta.cuda())
        gotT = H.gemm(wv2.contiguous(), x, split_k=1, ln=dict(mode=2, sc=scv.view(torch.float32).view(2, 256).contiguous(), stats=stats))
        _close(gotT, wv.float() @ y.t())
    # the GEGLU form (ff.net.0.proj behind norm3)
    if N % 32 == 0:      # (every tile of this list has whole 32-column groups per wave)
        wp, bp = H.pack_geglu_weight(w, bias)
        wp2, scp = _ln_fold(wp, gamma.cuda(), beta.cuda())
        gg = H.gemm(x, wp2.contiguous(), bias=bp, act=2, tile_cfg=tile + 1, ln=dict(mode=1, sc=scp.view(torch.float32).view(2, N).contiguous()))
        val, gate = ref[:, :N // 2], ref[:, N // 2:]
        _close(gg, val * F.gelu(gate))


@pytest.mark.parametrize("ratio,tol", [(16, 6e-4), (40, 8e-4), (100, 3e-3)])
def test_layernorm_fold_on_rows_with_a_large_mean(ratio, tol):
    """the folded LayerNorm takes the variance as E[x^2] - mean^2 from one-pass fp32 sums and subtracts mean * rowsum(W) from the product: both
    cancel when |mean| >> std.  Rows at mean / std = 16, 40, 100 (fp16 inputs) against torch's two-pass LayerNorm + matmul in fp32: the fold
    stays at the unfolded path's fp16 error (3.5e-4 of the output range) up to 40 and is 1.8e-3 at 100 (measured; bounds with margin)"""
    from scaledreamer_amd.diffusion import hip_ops as H
    from scaledreamer_amd.diffusion.weights import _ln_fold

    M = C_ = N = 1280
    x = (_rand(M, C_, seed=1).float() * 0.5 + 0.5 * ratio).half()
    gamma, beta = (_rand(C_, seed=3).float() * 0.2 + 1.0), _rand(C_, seed=4).float() * 0.3
    w, bias = _rand(N, C_, scale=C_ ** -0.5, seed=5), _rand(N, seed=6)
    ref = F.layer_norm(x.float(), (C_,), gamma.cuda(), beta.cuda(), 1e-5) @ w.float().t() + bias.float()
    w2, sc = _ln_fold(w, gamma.cuda(), beta.cuda())
    got = H.gemm(x, w2.contiguous(), bias=bias, split_k=1, ln=dict(mode=1, sc=sc.view(torch.float32).view(2, N).contiguous()))
    assert float((got.float() - ref).abs().max() / ref.abs().max()) < tol
