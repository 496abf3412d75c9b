"""Pins the oracle against outputs of the REFERENCE ITSELF run in the build container.

tests/golden/*.npz were produced by the reference's own NeRFVolumeRenderer / ImplicitVolume / NoMaterial /
NeuralEnvironmentMapBackground code (tests/golden/make_goldens.py).  Here the oracle-only composition
(oracle/ref_renderer.py) must reproduce every key of that output dictionary and the parameter gradients.
"""
import numpy as np
import pytest
import torch

from golden_util import RENDERER_GOLDENS, load_renderer_golden, reference_loss_torch


@pytest.mark.parametrize("name", RENDERER_GOLDENS)
def test_oracle_renderer_matches_reference_glue(name):
    from oracle import ref_renderer as R

    g = load_renderer_golden(name)
    out, ctx = R.forward(g)
    n = g["out_weights"].shape[0]
    assert out["weights"].shape[0] == n, "sample count differs from the reference run"
    np.testing.assert_array_equal(out["ray_indices"], g["out_ray_indices"])
    for k in ["t_points", "t_intervals", "points", "t_dirs"]:
        np.testing.assert_array_equal(out[k], g["out_" + k], err_msg=k)
    for k in ["density", "features", "weights"]:
        np.testing.assert_allclose(out[k], g["out_" + k], rtol=2e-5, atol=2e-6, err_msg=k)
    np.testing.assert_allclose(out["normal"], g["out_normal"], rtol=0, atol=5e-4)
    hw = int(g["h"]) * int(g["w"])
    for k in ["comp_rgb", "comp_rgb_fg", "comp_rgb_bg", "opacity", "depth", "z_variance"]:
        np.testing.assert_allclose(out[k].reshape(hw, -1), g["out_" + k].reshape(hw, -1), rtol=2e-5, atol=2e-6, err_msg=k)

    # gradients: torch autograd gives d loss / d outputs, the oracle back-propagates them to the parameters
    t = {k: torch.tensor(out[k], requires_grad=out[k].dtype == np.float32) for k in
         ["comp_rgb", "depth", "opacity", "z_variance", "weights", "normal", "t_dirs"]}
    loss = reference_loss_torch(t, g)
    assert abs(loss.item() - float(g["loss"])) < 1e-4 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    gr = lambda k: None if t[k].grad is None else t[k].grad.numpy()
    grads = R.backward(g, ctx, d_comp_rgb=gr("comp_rgb"), d_opacity=gr("opacity"), d_depth=gr("depth"),
                       d_z_var=gr("z_variance"), d_normal=gr("normal"))
    for k in ["w1d", "w2d", "w1f", "w2f", "bw0", "bw1", "bw2"]:
        ref = g["g_" + k]
        scale = max(float(np.abs(ref).max()), 1e-6)
        np.testing.assert_allclose(grads[k] / scale, ref / scale, rtol=0, atol=2e-3, err_msg=k)
    idx, val = g["g_grid_idx"], g["g_grid_val"]
    np.testing.assert_allclose(grads["grid"][idx] / np.abs(val).max(), val / np.abs(val).max(), rtol=0, atol=2e-3)
    assert abs(np.linalg.norm(grads["grid"].astype(np.float64)) / float(g["g_grid_l2"]) - 1) < 2e-3
    assert int((grads["grid"] != 0).sum()) == int(g["g_grid_nnz"])
    idx, val = g["g_bgrid_idx"], g["g_bgrid_val"]
    np.testing.assert_allclose(grads["bgrid"][idx] / np.abs(val).max(), val / np.abs(val).max(), rtol=0, atol=2e-3)
