"""CPU tests that pin the oracle (oracle/asd_oracle.c) — known answers, closed forms, gradient checks.

The reference has no tests or golden vectors for this arithmetic; these are the
known-answer tests SURVEY.md §8c prescribes.  tests/test_goldens_cpu.py adds the vectors produced by the
reference's own Python glue.
"""
import numpy as np
import pytest


def _weights(rng, scale=0.3):
    return (rng.normal(0, scale, (64, 32)).astype(np.float32), rng.normal(0, scale, (1, 64)).astype(np.float32),
            rng.normal(0, scale, (64, 32)).astype(np.float32), rng.normal(0, scale, (3, 64)).astype(np.float32))


def test_param_counts(oracle):
    # SURVEY.md §8c known-answer #1 / Appendix D.5
    assert oracle.grid_meta().n_params == 12_599_920
    assert oracle.grid_meta(4, 2, 19, 4, 4.0).n_params == 1_581_184
    assert oracle.grid_meta(16, 2, 19, 16, 1.0).n_params == 131_072
    m = oracle.grid_meta()
    assert list(m.resolution)[:5] == [16, 24, 34, 49, 71] and m.resolution[15] == 4096
    assert list(m.size)[:5] == [4096, 13824, 39304, 117656, 357912]
    assert all(s == 524288 for s in list(m.size)[5:16])
    assert list(m.dense) == [1] * 5 + [0] * 11
    assert abs(m.scale[15] - 4095.0) < 1e-2  # "max resolution 4096" (asd_sd_nerf.yaml:53)


def test_dense_level_reproduces_linear_field(oracle):
    # known-answer #2: params = linear function of the grid coordinate -> trilinear interpolation is exact
    m = oracle.grid_meta(1, 2, 19, 16, 1.0)  # one dense level, res 16, scale 15
    res = m.resolution[0]
    params = np.zeros((m.size[0], 2), np.float32)
    ix, iy, iz = np.meshgrid(np.arange(res), np.arange(res), np.arange(res), indexing="ij")
    idx = (ix + iy * res + iz * res * res).reshape(-1)
    a = np.array([0.25, -0.5, 0.125], np.float32)
    params[idx, 0] = (a[0] * ix + a[1] * iy + a[2] * iz).reshape(-1)
    params[idx, 1] = 1.0
    rng = np.random.default_rng(1)
    x = rng.uniform(0.0, 0.96, (200, 3)).astype(np.float32)
    out = oracle.hashgrid_fwd(m, params.reshape(-1), x)
    pos = x * m.scale[0] + 0.5
    np.testing.assert_allclose(out[:, 0], pos @ a, rtol=0, atol=2e-5)
    np.testing.assert_allclose(out[:, 1], 1.0, atol=1e-6)


def test_hashed_index_of_111(oracle):
    # known-answer #3: the entry of integer corner (1,1,1) on a hashed level
    m = oracle.grid_meta(1, 2, 4, 64, 1.0)  # res 64 -> 262144 cells > 2^4 -> hashed, table of 16
    assert m.dense[0] == 0 and m.size[0] == 16
    want = (1 ^ 2654435761 ^ 805459861) % 16
    params = np.zeros((16, 2), np.float32)
    params[want] = [3.0, 7.0]
    # x such that pos = x*scale+0.5 = 1.0 exactly -> cell (1,1,1), weights (0,0,0) -> only corner (1,1,1)
    x = np.full((1, 3), 0.5 / m.scale[0], np.float32)
    out = oracle.hashgrid_fwd(m, params.reshape(-1), x)
    np.testing.assert_allclose(out[0], [3.0, 7.0], atol=1e-5)


def test_hashgrid_bwd_is_adjoint_of_fwd(oracle):
    m = oracle.grid_meta(6, 2, 12, 4, 1.7)
    rng = np.random.default_rng(2)
    p = rng.normal(size=m.n_params).astype(np.float32)
    x = rng.uniform(0, 1, (64, 3)).astype(np.float32)
    dout = rng.normal(size=(64, 12)).astype(np.float32)
    lhs = float((oracle.hashgrid_fwd(m, p, x).astype(np.float64) * dout).sum())
    rhs = float((oracle.hashgrid_bwd(m, x, dout).astype(np.float64) * p).sum())
    assert abs(lhs - rhs) < 1e-3 * max(1.0, abs(lhs))


def _torch_field(oracle, m, c, grid, w1d, w2d, w1f, w2f, pts, with_normal):
    """float64 torch restatement of implicit_volume.py:109-207 on top of the oracle's encoder."""
    import torch

    def enc_of(p):
        x01 = ((p + c.radius) / (2 * c.radius)).astype(np.float32)
        return torch.tensor(oracle.hashgrid_fwd(m, grid, x01), dtype=torch.float64, requires_grad=True), x01

    W = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in (w1d, w2d, w1f, w2f)]

    def density(e, p):
        raw = (torch.relu(e @ W[0].T) @ W[1].T)[:, 0]
        r = torch.tensor(np.linalg.norm(p.astype(np.float64), axis=1))
        return torch.nn.functional.softplus(raw + c.blob_scale * (1 - r / c.blob_std))

    e0, x0 = enc_of(pts)
    sigma = density(e0, pts)
    feats = torch.relu(e0 @ W[2].T) @ W[3].T
    encs, xs, normal = [e0], [x0], None
    if with_normal:
        cols = []
        for k in range(3):
            q = pts.copy()
            q[:, k] += np.float32(c.fd_eps)
            q = np.clip(q, -c.radius, c.radius).astype(np.float32)
            ek, xk = enc_of(q)
            encs.append(ek)
            xs.append(xk)
            cols.append(-(density(ek, q) - sigma) / np.float32(c.fd_eps))
        normal = torch.nn.functional.normalize(torch.stack(cols, -1), dim=-1)
    return sigma, feats, normal, W, encs, xs


@pytest.mark.parametrize("with_normal", [False, True])
def test_field_fwd_bwd_match_torch_autograd(oracle, with_normal):
    m, c = oracle.grid_meta(), oracle.field_cfg()
    rng = np.random.default_rng(3)
    grid = rng.uniform(-0.1, 0.1, m.n_params).astype(np.float32)
    w1d, w2d, w1f, w2f = _weights(rng)
    pts = rng.uniform(-0.6, 0.6, (40, 3)).astype(np.float32)
    pts[0] = [0.995, -0.2, 0.3]  # finite-difference offset gets clamped to the radius
    ds = rng.normal(size=40).astype(np.float32)
    df = rng.normal(size=(40, 3)).astype(np.float32)
    dn = rng.normal(size=(40, 3)).astype(np.float32) if with_normal else None
    import torch

    sigma, feats, normal, W, encs, xs = _torch_field(oracle, m, c, grid, w1d, w2d, w1f, w2f, pts, with_normal)
    s, f, n, enc = oracle.field_fwd(m, c, grid, w1d, w2d, w1f, w2f, pts, want_normal=with_normal)
    np.testing.assert_allclose(s, sigma.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(f, feats.detach().numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(enc, encs[0].detach().numpy(), rtol=0, atol=0)
    loss = (sigma * torch.tensor(ds, dtype=torch.float64)).sum() + (feats * torch.tensor(df, dtype=torch.float64)).sum()
    if with_normal:
        np.testing.assert_allclose(n, normal.detach().numpy(), rtol=0, atol=2e-4)  # fp32 cancellation in s_k - s
        loss = loss + (normal * torch.tensor(dn, dtype=torch.float64)).sum()
    loss.backward()
    g = oracle.field_bwd(m, c, grid, w1d, w2d, w1f, w2f, pts, ds, df, dn)
    tol = dict(rtol=2e-3, atol=2e-3) if with_normal else dict(rtol=1e-4, atol=1e-5)
    for k in range(4):
        np.testing.assert_allclose(g[1 + k], W[k].grad.numpy(), **tol)
    dgrid = sum(oracle.hashgrid_bwd(m, x, e.grad.numpy().astype(np.float32)) for e, x in zip(encs, xs))
    np.testing.assert_allclose(g[0], dgrid, **tol)


def test_envmap_fwd_bwd_match_torch_autograd(oracle):
    import torch

    m = oracle.grid_meta(4, 2, 19, 4, 4.0)
    rng = np.random.default_rng(4)
    grid = rng.uniform(-0.5, 0.5, m.n_params).astype(np.float32)
    w0 = rng.normal(0, 0.5, (16, 8)).astype(np.float32)
    w1 = rng.normal(0, 0.5, (16, 16)).astype(np.float32)
    w2 = rng.normal(0, 0.5, (3, 16)).astype(np.float32)
    d = rng.normal(size=(30, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    dc = rng.normal(size=(30, 3)).astype(np.float32)
    x01 = ((d + 1) / 2).astype(np.float32)
    enc = torch.tensor(oracle.hashgrid_fwd(m, grid, x01), dtype=torch.float64, requires_grad=True)
    W = [torch.tensor(w, dtype=torch.float64, requires_grad=True) for w in (w0, w1, w2)]
    # neural_environment_map_background.py:52-55: Linear-ReLU-Linear-ReLU-Linear, sigmoid
    col = torch.sigmoid(torch.relu(torch.relu(enc @ W[0].T) @ W[1].T) @ W[2].T)
    np.testing.assert_allclose(oracle.envmap_fwd(m, grid, w0, w1, w2, d), col.detach().numpy(), atol=1e-6)
    (col * torch.tensor(dc, dtype=torch.float64)).sum().backward()
    g = oracle.envmap_bwd(m, grid, w0, w1, w2, d, dc)
    for k in range(3):
        np.testing.assert_allclose(g[1 + k], W[k].grad.numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(g[0], oracle.hashgrid_bwd(m, x01, enc.grad.numpy().astype(np.float32)), rtol=1e-4,
                               atol=1e-6)


def test_composite_closed_form_constant_sigma(oracle):
    # constant sigma => T_i = exp(-sigma t_i), sum w = 1 - T_end
    n, sigma, dt = 100, 3.0, 0.01
    t0 = (0.5 + dt * np.arange(n)).astype(np.float32)
    t1 = (t0 + dt).astype(np.float32)
    out = oracle.composite_fwd(np.full(n, sigma, np.float32), t0, t1, np.ones((n, 3), np.float32), [0], [n],
                               np.zeros((1, 3), np.float32))
    T = np.exp(-sigma * dt * np.arange(n))
    np.testing.assert_allclose(out["weights"], T * (1 - np.exp(-sigma * dt)), rtol=2e-5)
    np.testing.assert_allclose(out["opacity"][0], 1 - np.exp(-sigma * dt * n), rtol=1e-5)
    np.testing.assert_allclose(out["rgb_fg"][0], out["opacity"][0], rtol=1e-6)
    # background shows through the remaining transmittance
    out2 = oracle.composite_fwd(np.full(n, sigma, np.float32), t0, t1, np.zeros((n, 3), np.float32), [0], [n],
                                np.ones((1, 3), np.float32))
    np.testing.assert_allclose(out2["comp_rgb"][0], np.exp(-sigma * dt * n), rtol=1e-4)


@pytest.mark.parametrize("mode", [0, 1])
def test_composite_bwd_matches_finite_differences(oracle, mode):
    rng = np.random.default_rng(5)
    counts = np.array([0, 7, 70, 1, 130], np.int32)
    offs = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.int32)
    n = int(counts.sum())
    t0 = np.concatenate([np.sort(rng.uniform(0.2, 2.0, c)) for c in counts]).astype(np.float32)
    t1 = (t0 + 0.01).astype(np.float32)
    sig = (rng.uniform(0, 40, n) if mode == 0 else rng.uniform(0, 0.3, n)).astype(np.float32)
    rgb = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    bg = rng.uniform(0, 1, (5, 3)).astype(np.float32)
    ups = dict(d_comp_rgb=rng.normal(size=(5, 3)), d_rgb_fg=rng.normal(size=(5, 3)), d_opacity=rng.normal(size=5),
               d_depth=rng.normal(size=5), d_z_var=rng.normal(size=5), d_weights=rng.normal(size=n))
    ups = {k: v.astype(np.float32) for k, v in ups.items()}

    def loss(sig_, rgb_, bg_):
        o = oracle.composite_fwd(sig_, t0, t1, rgb_, offs, counts, bg_, mode=mode)
        f64 = lambda a: a.astype(np.float64)
        return float((f64(o["comp_rgb"]) * ups["d_comp_rgb"]).sum() + (f64(o["rgb_fg"]) * ups["d_rgb_fg"]).sum()
                     + (f64(o["opacity"]) * ups["d_opacity"]).sum() + (f64(o["depth"]) * ups["d_depth"]).sum()
                     + (f64(o["z_var"]) * ups["d_z_var"]).sum() + (f64(o["weights"]) * ups["d_weights"]).sum())

    fwd = oracle.composite_fwd(sig, t0, t1, rgb, offs, counts, bg, mode=mode)
    d_sigma, d_rgb, d_bg = oracle.composite_bwd(sig, t0, t1, rgb, offs, counts, bg, fwd, mode=mode, **ups)
    eps = 1e-2 if mode == 0 else 1e-3
    for i in [0, 3, 6, 10, 50, 76, 77, 100, n - 1]:
        sp, sm = sig.copy(), sig.copy()
        sp[i] += eps
        sm[i] -= eps
        fd = (loss(sp, rgb, bg) - loss(sm, rgb, bg)) / (2 * eps)
        assert abs(fd - d_sigma[i]) < 2e-2 * max(1.0, abs(fd)), (i, fd, d_sigma[i])
    for i, k in [(2, 0), (40, 1), (n - 1, 2)]:
        rp, rm = rgb.copy(), rgb.copy()
        rp[i, k] += 1e-2
        rm[i, k] -= 1e-2
        fd = (loss(sig, rp, bg) - loss(sig, rm, bg)) / 2e-2
        assert abs(fd - d_rgb[i, k]) < 1e-2 * max(1.0, abs(fd))
    bp, bm = bg.copy(), bg.copy()
    bp[2, 1] += 1e-2
    bm[2, 1] -= 1e-2
    assert abs((loss(sig, rgb, bp) - loss(sig, rgb, bm)) / 2e-2 - d_bg[2, 1]) < 1e-2


def _rays(rng, n, radius=1.0):
    o = rng.normal(size=(n, 3))
    o = (o / np.linalg.norm(o, axis=1, keepdims=True) * rng.uniform(1.0, 1.5, (n, 1))).astype(np.float32)
    target = rng.uniform(-0.5, 0.5, (n, 3)).astype(np.float32)
    d = target - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("stratified", [False, True])
def test_march_invariants(oracle, stratified):
    # SURVEY.md §8c marcher invariants
    rng = np.random.default_rng(6)
    c = oracle.march_cfg(num_samples_per_ray=128)
    ix, iy, iz = np.meshgrid(*[np.arange(32)] * 3, indexing="ij")
    centre = (np.stack([ix, iy, iz], -1) + 0.5) / 32 * 2 - 1
    binaries = (np.linalg.norm(centre, axis=-1) < 0.5)
    bits = oracle.pack_bits(binaries)
    o, d = _rays(rng, 300)
    d[0] = -o[0] / np.linalg.norm(o[0])              # straight through the centre
    o[1] = [3, 3, 3]; d[1] = [1, 0, 0]               # misses the box
    o[2] = [0.01, 0.02, -2.0]; d[2] = [0, 0, 1]      # axis-aligned (two zero direction components)
    jit = rng.uniform(0, 1, 300).astype(np.float32) if stratified else None
    count, offset, ray_idx, t0, t1, pts = oracle.march(c, o, d, bits, jit)
    assert count[1] == 0 and count[0] > 0 and count[2] > 0
    assert (np.diff(ray_idx) >= 0).all()
    np.testing.assert_allclose(t1 - t0, c.step, rtol=0, atol=2e-6)
    mid = o[ray_idx] + d[ray_idx] * ((t0 + t1) * 0.5)[:, None]
    np.testing.assert_allclose(mid, pts, atol=1e-5)
    assert (np.abs(mid) <= 1.0 + 1e-5).all()
    cell = np.clip(np.floor((mid + 1) / 2 * 32).astype(int), 0, 31)
    assert binaries[cell[:, 0], cell[:, 1], cell[:, 2]].all()
    for r in range(300):
        tt = t0[offset[r]:offset[r] + count[r]]
        assert (np.diff(tt) > 0).all()
    # every occupied lattice point was emitted: the central ray crosses the r=0.5 ball => ~1.0/step samples
    assert abs(count[0] - 1.0 / c.step) <= 12
    if not stratified:  # deterministic
        again = oracle.march(c, o, d, bits, None)
        assert all(np.array_equal(a, b) for a, b in zip(again, (count, offset, ray_idx, t0, t1, pts)))


def test_prune_semantics(oracle):
    t0 = (0.01 * np.arange(10)).astype(np.float32)
    t1 = (t0 + 0.01).astype(np.float32)
    sigma = np.array([0, 200, 200, 200, 200, 200, 200, 0.1, 200, 200], np.float32)
    keep, kept = oracle.prune(sigma, t0, t1, [0], [10], early_stop_eps=1e-4, alpha_thre=0.01)
    T = np.exp(-np.concatenate([[0], np.cumsum(sigma * 0.01)[:-1]]))
    alpha = 1 - np.exp(-sigma * 0.01)
    np.testing.assert_array_equal(keep.astype(bool), (T >= 1e-4) & (alpha >= 0.01))
    assert kept[0] == keep.sum() and keep[0] == 0 and keep[1] == 1 and keep[9] == 0


def test_occgrid_update(oracle):
    rng = np.random.default_rng(7)
    occs = rng.uniform(0, 0.02, 32768).astype(np.float32)
    idx = rng.permutation(32768)[:8192].astype(np.int32)
    new = rng.uniform(0, 0.05, 8192).astype(np.float32)
    o2, bits, binaries = oracle.occgrid_update(occs, idx, new, 0.95, 0.01)
    want = occs.copy()
    want[idx] = np.maximum(want[idx] * np.float32(0.95), new)
    np.testing.assert_array_equal(o2, want)
    thre = min(want.astype(np.float64).mean(), 0.01)
    np.testing.assert_array_equal(binaries.astype(bool), want > np.float32(thre))
    np.testing.assert_array_equal(oracle.pack_bits(binaries), bits)
