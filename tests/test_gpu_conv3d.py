"""GPU parity of the generator backbone's HIP path (csrc/conv3d.hip), through the C ABI: the split-fp16 3x3x3 convolution
— forward, input gradient, weight gradient — against float64 F.conv3d (and next to the fp32 library path's own error), the layer tail and
the trilinear upsampling against torch autograd, and the whole Generator3D against the golden of the reference's own `Generator`."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _cl(x):          # [N, C, D, H, W] -> channel-last [N, D, H, W, C]
    return x.permute(0, 2, 3, 4, 1).contiguous()


def _cf(x):
    return x.permute(0, 4, 1, 2, 3).contiguous()


def _relerr(a, b):
    return float((a.detach().double() - b.detach().double()).abs().max() / b.detach().double().abs().max())


def _rel_l2(a, b):
    """gradients through the leaky-ReLU / clamp: an output within rounding distance of the kink may take the other branch than the float64
    reference (the forward values agree to 1e-6, not to the sign of a 1e-7), which moves single elements by 80 % — compare in the L2 norm"""
    return float((a.detach().double() - b.detach().double()).norm() / b.detach().double().norm())


@pytest.mark.parametrize("N,D,H,W,cin,cout", [(1, 6, 32, 32, 64, 64), (2, 5, 16, 32, 128, 64), (1, 4, 16, 16, 64, 128), (1, 3, 16, 16, 192, 256)])
def test_conv3d_forward_input_gradient_weight_gradient_vs_float64(N, D, H, W, cin, cout):
    from scaledreamer_amd import ops

    g = torch.Generator().manual_seed(N * 1000 + cin + cout)
    x = torch.randn(N, cin, D, H, W, generator=g) * torch.rand(N, cin, 1, 1, 1, generator=g) * 3          # uneven channel magnitudes
    w = torch.randn(N, cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    dy = torch.randn(N, cout, D, H, W, generator=g) * 1e-4                                                  # gradients are small numbers
    x64, w64, dy64 = x.double().requires_grad_(True), w.double().requires_grad_(True), dy.double()
    y64 = torch.cat([F.conv3d(x64[n:n + 1], w64[n], padding=1) for n in range(N)])
    y64.backward(dy64)
    xd, wd, dyd = _cl(x).cuda(), w.cuda(), _cl(dy).cuda()
    y = ops.conv3d_fwd(xd, wd)
    dx = ops.conv3d_dgrad(dyd, wd, cin)
    dw = ops.conv3d_wgrad(xd, dyd)
    torch.cuda.synchronize()
    e_y, e_dx, e_dw = _relerr(_cf(y.cpu()), y64.detach()), _relerr(_cf(dx.cpu()), x64.grad), _relerr(dw.cpu(), w64.grad)
    # the fp32 library path on the same problem (what the reference runs): its error against float64 is the yardstick
    y32 = torch.cat([F.conv3d(x[n:n + 1].cuda(), w[n].cuda(), padding=1) for n in range(N)]).cpu()
    e_lib = _relerr(y32, y64.detach())
    print(f"conv3d {cin}->{cout} @ {D}x{H}x{W}: rel. error vs float64  fwd {e_y:.2e}  dgrad {e_dx:.2e}  wgrad {e_dw:.2e}   (library fp32 fwd {e_lib:.2e})")
    # split-fp16: 2^-22 per operand + fp32 accumulation over K = 27 cin terms
    assert e_y < 3e-6 and e_dx < 3e-6 and e_dw < 3e-6
    assert e_y < max(4 * e_lib, 1e-6)


@pytest.mark.parametrize("N,cout,cin,k,gain,demod", [(2, 64, 128, 3, 1.0, True), (1, 512, 512, 3, 1.0, True), (3, 32, 64, 1, 0.125, False), (8, 7, 5, 3, 1.0, True)])
def test_modulated_weights_and_their_gradients_vs_float64(N, cout, cin, k, gain, demod):
    """asd_modulated_weights_fwd / _bwd against the tensor-op form of the reference (stylegan_3dconv_modules.py:64-82: w = weight * styles;
    dcoefs = (w.square().sum(dim=[2,3,4,5]) + 1e-8).rsqrt(); w = w * dcoefs) evaluated in float64"""
    from scaledreamer_amd.generators import _ModWeightsFn

    g = torch.Generator().manual_seed(cout + cin)
    weight = torch.randn(cout, cin, k, k, k, generator=g)
    styles = torch.randn(N, cin, generator=g) + 1.0
    d_wm = torch.randn(N, cout, cin, k, k, k, generator=g)
    w64, s64 = weight.double().requires_grad_(True), styles.double().requires_grad_(True)
    ref = w64.unsqueeze(0) * (s64 * gain).reshape(N, 1, cin, 1, 1, 1)
    if demod:
        ref = ref * (ref.square().sum(dim=[2, 3, 4, 5]) + 1e-8).rsqrt().reshape(N, -1, 1, 1, 1, 1)
    ref.backward(d_wm.double())
    wd, sd = weight.cuda().requires_grad_(True), styles.cuda().requires_grad_(True)
    wm = _ModWeightsFn.apply(wd, sd, gain, demod)
    wm.backward(d_wm.cuda())
    torch.cuda.synchronize()
    e = _relerr(wm.cpu(), ref), _relerr(wd.grad.cpu(), w64.grad), _relerr(sd.grad.cpu(), s64.grad)
    print(f"modulated weights {cin}->{cout} k={k} N={N}: rel. error vs float64  wm {e[0]:.2e}  d_weight {e[1]:.2e}  d_styles {e[2]:.2e}")
    assert max(e) < 2e-6


def test_conv3d_one_outlier_voxel_far_above_the_rest():
    """the split-fp16 operands carry ONE power-of-two scale per tensor (csrc/conv3d.hip: x s = hi + lo): an element 2^20 times the rest pushes
    every other element's low half into fp16's subnormal range (19 instead of 22 bits).  The outputs the outlier does not reach — every
    weight-gradient entry of the other input channels, every output voxel outside its 3x3x3 neighbourhood — are compared on their own
    scale against float64; what the outlier reaches, on the scale of the whole tensor."""
    from scaledreamer_amd import ops

    g = torch.Generator().manual_seed(11)
    N, cin, cout, D, H, W = 1, 64, 64, 4, 16, 16
    x = torch.randn(N, cin, D, H, W, generator=g)
    w = torch.randn(N, cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5
    dy = torch.randn(N, cout, D, H, W, generator=g) * 1e-4
    oc, od, oh, ow = 5, 2, 7, 9
    x[0, oc, od, oh, ow] = float(2 ** 20)
    x64, w64, dy64 = x.double().requires_grad_(True), w.double().requires_grad_(True), dy.double()
    y64 = F.conv3d(x64, w64[0], padding=1)
    y64.backward(dy64)
    xd, wd, dyd = _cl(x).cuda(), w.cuda(), _cl(dy).cuda()
    y = _cf(ops.conv3d_fwd(xd, wd).cpu())
    dw = ops.conv3d_wgrad(xd, dyd).cpu()
    torch.cuda.synchronize()
    y32 = F.conv3d(x.cuda(), w[0].cuda(), padding=1).cpu()
    far = torch.ones(D, H, W, dtype=torch.bool)
    far[od - 1:od + 2, oh - 1:oh + 2, ow - 1:ow + 2] = False
    others = [c for c in range(cin) if c != oc]
    e_y_all, e_dw_all = _relerr(y, y64.detach()), _relerr(dw, w64.grad)
    e_y_far, e_y_far_lib = _relerr(y[0][:, far], y64.detach()[0][:, far]), _relerr(y32[0][:, far], y64.detach()[0][:, far])
    e_dw_far = _relerr(dw[0][:, others], w64.grad[0][:, others])
    print(f"conv3d with one voxel 2^20 x the rest: fwd {e_y_all:.2e} (whole tensor) {e_y_far:.2e} (voxels it does not reach; library fp32 {e_y_far_lib:.2e})"
          f"  wgrad {e_dw_all:.2e} (whole) {e_dw_far:.2e} (other input channels)")
    assert e_y_all < 3e-6 and e_dw_all < 3e-6               # on the scale of the tensor: as without the outlier
    # on their own scale: 19-bit operands (2^-19 = 1.9e-6 per element, errors of K = 27 cin terms adding at random); measured 1.4e-6 / 1.2e-6
    assert e_y_far < 6e-6 and e_dw_far < 6e-6


def test_conv3d_fused_layer_tail_and_its_gradient():
    from scaledreamer_amd import ops
    from scaledreamer_amd.generators import _Conv3dFn

    g = torch.Generator().manual_seed(7)
    N, D, H, W, cin, cout = 1, 4, 16, 16, 64, 64
    x = (torch.randn(N, D, H, W, cin, generator=g)).cuda().requires_grad_(True)
    w = (torch.randn(N, cout, cin, 3, 3, 3, generator=g) / (27 * cin) ** 0.5 * 150).cuda().requires_grad_(True)     # large: some outputs reach the clamp
    bias = torch.randn(cout, generator=g).cuda().requires_grad_(True)
    noise = torch.randn(N, D, H, W, generator=g).cuda()
    ns = torch.tensor([0.7]).cuda().requires_grad_(True)
    gain, clamp = 2 ** 0.5, 256.0
    y, ay = _Conv3dFn.apply(x, None, w, bias, noise, ns, True, gain, clamp)
    assert float(ay.view(torch.float32)) == float(y.detach().abs().max())          # the max|y| word the next layer's split reads
    gy = torch.randn(y.shape, generator=g).cuda()
    y.backward(gy)
    got = [t.grad.clone() for t in (x, w, bias, ns)]
    for t in (x, w, bias, ns):
        t.grad = None
    x64, w64, b64, ns64 = (t.detach().double().requires_grad_(True) for t in (x, w, bias, ns))
    z = F.conv3d(x64.permute(0, 4, 1, 2, 3), w64[0], padding=1).permute(0, 2, 3, 4, 1) + (noise.double() * ns64)[..., None] + b64
    ref = torch.clamp(F.leaky_relu(z, 0.2) * gain, -clamp, clamp)
    assert float((ref.abs() >= clamp).float().mean()) > 1e-4, "the test must exercise the clamp"
    ref.backward(gy.double())
    assert _relerr(y, ref.detach()) < 3e-6
    for a, b, name in zip(got, (x64.grad, w64.grad, b64.grad, ns64.grad), ("dx", "dw", "d_bias", "d_noise_strength")):
        assert _rel_l2(a, b) < 1e-4, name


@pytest.mark.parametrize("r,C", [(4, 512), (8, 64), (16, 32)])
def test_upsample_matches_trilinear_align_corners_and_its_transpose(r, C):
    from scaledreamer_amd.generators import _UpsampleFn

    g = torch.Generator().manual_seed(r)
    x = torch.randn(2, r, r, r, C, generator=g).cuda().requires_grad_(True)
    bias = torch.randn(C, generator=g).cuda().requires_grad_(True)
    noise = torch.randn(2, 2 * r, 2 * r, 2 * r, generator=g).cuda()
    ns = torch.tensor([0.3]).cuda().requires_grad_(True)
    add = torch.randn(2, 2 * r, 2 * r, 2 * r, C, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(2, 2 * r, 2 * r, 2 * r, C, generator=g).cuda()

    def ref(x_, bias_, ns_, add_, act, with_add):
        up = F.interpolate(x_.permute(0, 4, 1, 2, 3), scale_factor=2, mode="trilinear", align_corners=True).permute(0, 2, 3, 4, 1)
        if act:
            up = torch.clamp(F.leaky_relu(up + (noise * ns_)[..., None] + bias_, 0.2) * 1.4142135, -256, 256)
        return up + add_ if with_add else up

    for act, with_add in ((True, False), (False, True), (True, True)):          # layer tail / skip volume / layer tail + const_bias
        y, ay = _UpsampleFn.apply(x, bias if act else None, noise if act else None, ns if act else None, act, 1.4142135, 256.0, add if with_add else None)
        assert float(ay.view(torch.float32)) == float(y.detach().abs().max())
        y.backward(gy)
        got = {k: t.grad.clone() for k, t in (("x", x), ("bias", bias), ("ns", ns), ("add", add)) if t.grad is not None}
        for t in (x, bias, ns, add):
            t.grad = None
        yr = ref(x, bias, ns, add, act, with_add)
        yr.backward(gy)
        assert _relerr(y, yr.detach()) < 2e-6
        for k, t in (("x", x), ("bias", bias), ("ns", ns), ("add", add)):
            if t.grad is not None:
                assert (_rel_l2 if act else _relerr)(got[k], t.grad) < (5e-4 if act else 2e-5), (act, with_add, k)      # (without `add`, act' is read off y: the fp32 reference may take the other branch within rounding distance of the kink)
            t.grad = None


def test_upsample_with_added_volume_takes_the_activation_branch_from_recorded_bits():
    """y = act(.) + const_bias: (y - const_bias) does not give the activation back exactly — a value clamped at +-256 gain returns as
    255.9999 (gradient 0 in the reference, slope 1 read off the difference), one smaller than ulp(const_bias) / 2 loses its sign (slope
    0.2 vs 1).  The forward kernel records the branch of every element; the gradient equals torch autograd's on exactly those cases."""
    from scaledreamer_amd.generators import _UpsampleFn

    r, C, gain, clamp = 4, 8, 2 ** 0.5, 256.0
    # constant volumes upsample to themselves: channel 0 far above the clamp, 1 far below -clamp, 2 tiny positive, 3 tiny negative,
    # 4 ordinary positive, 5 ordinary negative, 6 just inside the clamp, 7 just outside
    vals = torch.tensor([1.0e4, -1.0e4, 1.0e-9, -1.0e-9, 0.7, -0.7, 255.9 / gain, 256.1 / gain])
    x = torch.zeros(1, r, r, r, C) + vals
    bias = torch.zeros(C)
    add = torch.full((1, 2 * r, 2 * r, 2 * r, C), 1234.567)
    dy = torch.randn(1, 2 * r, 2 * r, 2 * r, C, generator=torch.Generator().manual_seed(2))
    xr = x.clone().requires_grad_(True)
    up = F.interpolate(xr.permute(0, 4, 1, 2, 3), scale_factor=2, mode="trilinear", align_corners=True).permute(0, 2, 3, 4, 1)
    ref = torch.clamp(F.leaky_relu(up + bias, 0.2) * gain, -clamp, clamp) + add
    ref.backward(dy)
    xd = x.cuda().requires_grad_(True)
    y, _ = _UpsampleFn.apply(xd, bias.cuda(), None, None, True, gain, clamp, add.cuda())
    y.backward(dy.cuda())
    torch.cuda.synchronize()
    torch.testing.assert_close(y.detach().cpu(), ref.detach(), rtol=1e-6, atol=0)
    per_ch_ref, per_ch = xr.grad.sum(dim=(0, 1, 2, 3)), xd.grad.cpu().sum(dim=(0, 1, 2, 3))
    assert float(per_ch_ref[0].abs()) == 0.0 and float(per_ch_ref[7].abs()) == 0.0 and float(per_ch_ref[2].abs()) > 0.0      # the cases are in there
    torch.testing.assert_close(xd.grad.cpu(), xr.grad, rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(per_ch, per_ch_ref, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("cin,rows_shape", [(64, (2, 5, 16, 16)), (512, (1, 4, 4, 4)), (128, (1, 3, 8, 40))])
def test_torgb_matches_float64(cin, rows_shape):
    from scaledreamer_amd.generators import _ToRGBFn

    g = torch.Generator().manual_seed(cin)
    N = rows_shape[0]
    x = torch.randn(*rows_shape, cin, generator=g).cuda().requires_grad_(True)
    w = (torch.randn(N, 32, cin, generator=g) / cin ** 0.5).cuda().requires_grad_(True)
    b = torch.randn(32, generator=g).cuda().requires_grad_(True)
    add = torch.randn(*rows_shape, 32, generator=g).cuda().requires_grad_(True)
    gy = torch.randn(*rows_shape, 32, generator=g).cuda()
    y = _ToRGBFn.apply(x, w, b, add)
    y.backward(gy)
    got = [t.grad.clone() for t in (x, w, b, add)]
    x64, w64, b64, a64 = (t.detach().double().requires_grad_(True) for t in (x, w, b, add))
    ref = torch.einsum("n...i,noi->n...o", x64, w64) + b64 + a64
    ref.backward(gy.double())
    assert _relerr(y, ref.detach()) < 2e-6
    for a_, r_, name in zip(got, (x64.grad, w64.grad, b64.grad, a64.grad), ("dx", "dw", "d_bias", "d_add")):
        assert _relerr(a_, r_) < 5e-6, name


def test_generator3d_hip_backend_matches_reference_golden():
    """Generator3D(backend="hip") — every 3x3x3 convolution forward / input gradient / weight gradient through csrc/conv3d.hip, the
    4^3 and 8^3 levels through the padded-patch path — against the golden of the reference's own `Generator` (forward sub-sampled image,
    its norm, parameter gradients)"""
    from test_goldens_amortized_cpu import check_generator3d_golden

    check_generator3d_golden("cuda", "hip", tol=2.0)


def test_conv3d_full_size_layer_sampled_against_float64():
    """the generator's largest layer (64 -> 64 channels at 128^3: 464 GFLOP per pass) at full size: 512 random outputs of the forward and the input
    gradient and 8 x 27 entries of the weight gradient (each a sum over all 2 M voxels) against float64"""
    from scaledreamer_amd import ops

    g = torch.Generator(device="cuda").manual_seed(11)
    R, Cc = 128, 64
    x = torch.randn(1, R, R, R, Cc, device="cuda", generator=g).clamp_(-3, 3) * 5
    w = torch.randn(1, Cc, Cc, 3, 3, 3, device="cuda", generator=g) / (27 * Cc) ** 0.5
    dy = torch.randn(1, R, R, R, Cc, device="cuda", generator=g) * 1e-5
    y = ops.conv3d_fwd(x, w)
    dx = ops.conv3d_dgrad(dy, w, Cc)
    dw = ops.conv3d_wgrad(x, dy)
    xp = F.pad(x[0], (0, 0, 1, 1, 1, 1, 1, 1)).double()            # [R+2, R+2, R+2, C]
    dyp = F.pad(dy[0], (0, 0, 1, 1, 1, 1, 1, 1)).double()
    w64 = w[0].double()                                            # [co, ci, kd, ky, kx]
    idx = torch.randint(0, R, (512, 3), device="cuda", generator=g)
    idx[:8] = torch.tensor([[0, 0, 0], [R - 1, R - 1, R - 1], [0, R - 1, 5], [7, 0, R - 1], [R - 1, 3, 0], [64, 64, 64], [1, 1, 1], [0, 64, 0]], device="cuda")
    scale_y, scale_dx = float(y.abs().max()), float(dx.abs().max())
    for (d, h, ww) in idx.tolist():
        win = xp[d:d + 3, h:h + 3, ww:ww + 3]                      # [3,3,3,ci]
        ref = torch.einsum("dhwi,oidhw->o", win, w64)
        assert float((y[0, d, h, ww].double() - ref).abs().max()) < 3e-6 * scale_y
        gwin = dyp[d:d + 3, h:h + 3, ww:ww + 3].flip(0, 1, 2)      # dx[v] = sum_tap dy[v - off(tap)] w[:, :, tap]
        refd = torch.einsum("dhwo,oidhw->i", gwin, w64)
        assert float((dx[0, d, h, ww].double() - refd).abs().max()) < 3e-6 * scale_dx
    scale_dw = float(dw.abs().max())
    for co, ci in [(0, 0), (63, 63), (5, 40), (40, 5), (17, 17), (1, 62), (33, 2), (60, 31)]:
        gy = dy[0, ..., co].double()
        for kd in range(3):
            for ky in range(3):
                for kx in range(3):
                    ref = float((gy * xp[kd:kd + R, ky:ky + R, kx:kx + R, ci]).sum())
                    assert abs(float(dw[0, co, ci, kd, ky, kx]) - ref) < 3e-6 * scale_dw, (co, ci, kd, ky, kx)
