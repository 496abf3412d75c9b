"""Import harness for the reference's in-tree Python.

Runs ONLY in the build container (it needs /root/reference).  It registers stub modules for the packages
the reference imports but that are not installed (jaxtyping, typeguard, omegaconf, igl, pytorch_lightning)
and injects the build's OWN CPU restatement (oracle/) as `tinycudann` / `nerfacc`, so the reference's
renderer / geometry / background / guidance glue executes unchanged on CPU.  Nothing here is shipped or
imported by the product, and no reference source is copied: the reference modules are imported in place.
"""
from __future__ import annotations

import contextlib
import dataclasses
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REFERENCE = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle as O  # noqa: E402


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Sub:
    def __class_getitem__(cls, item):
        return cls


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


def _to_attr(o):
    if isinstance(o, dict):
        return _AttrDict({k: _to_attr(v) for k, v in o.items()})
    return o


# ---- oracle-backed tinycudann --------------------------------------------------------------------
class _OracleHashGrid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, meta):
        ctx.meta = meta
        ctx.save_for_backward(x)
        return torch.from_numpy(O.hashgrid_fwd(meta, params.detach().numpy(), x.detach().numpy()))

    @staticmethod
    def backward(ctx, dout):
        (x,) = ctx.saved_tensors
        return None, torch.from_numpy(O.hashgrid_bwd(ctx.meta, x.numpy(), dout.contiguous().numpy())), None


class OracleEncoding(nn.Module):
    def __init__(self, n_input_dims, encoding_config, dtype=torch.float32):
        super().__init__()
        c = dict(encoding_config)
        assert c.get("otype", "HashGrid") == "HashGrid"
        self.meta = O.grid_meta(c["n_levels"], c["n_features_per_level"], c["log2_hashmap_size"], c["base_resolution"],
                                c["per_level_scale"])
        self.n_input_dims = n_input_dims
        self.n_output_dims = self.meta.n_levels * 2
        self.params = nn.Parameter(torch.zeros(self.meta.n_params))

    def forward(self, x):
        return _OracleHashGrid.apply(x.contiguous().float(), self.params, self.meta)


# ---- oracle-backed nerfacc -----------------------------------------------------------------------
class _OracleWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sigmas, t0, t1, offset, count):
        n, nr = sigmas.shape[0], count.shape[0]
        z3, bg = np.zeros((n, 3), np.float32), np.zeros((nr, 3), np.float32)
        out = O.composite_fwd(sigmas.detach().numpy(), t0.numpy(), t1.numpy(), z3, offset, count, bg)
        ctx.stuff = (sigmas.detach().numpy(), t0.numpy(), t1.numpy(), z3, offset, count, bg, out)
        return torch.from_numpy(out["weights"])

    @staticmethod
    def backward(ctx, dw):
        s, t0, t1, z3, offset, count, bg, out = ctx.stuff
        d_sigma, _, _ = O.composite_bwd(s, t0, t1, z3, offset, count, bg, out, d_weights=dw.contiguous().numpy())
        return torch.from_numpy(d_sigma), None, None, None, None


def _packed(ray_indices, n_rays):
    count = np.bincount(ray_indices.numpy(), minlength=n_rays).astype(np.int32)
    offset = np.concatenate([[0], np.cumsum(count)[:-1]]).astype(np.int32)
    return offset, count


def render_weight_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None):
    offset, count = _packed(ray_indices, n_rays)
    w = _OracleWeights.apply(sigmas, t_starts.contiguous(), t_ends.contiguous(), offset, count)
    with torch.no_grad():
        alphas = 1.0 - torch.exp(-sigmas * (t_ends - t_starts))
        trans = w / alphas.clamp_min(1e-10)
    return w, trans, alphas


def accumulate_along_rays(weights, values=None, ray_indices=None, n_rays=None):
    src = weights[..., None] if values is None else weights[..., None] * values
    return torch.zeros((n_rays, src.shape[-1]), dtype=src.dtype).index_add_(0, ray_indices, src)


class OracleOccGridEstimator(nn.Module):
    def __init__(self, roi_aabb, resolution=32, levels=1):
        super().__init__()
        self.register_buffer("aabbs", torch.as_tensor(roi_aabb, dtype=torch.float32).reshape(1, 6).clone())
        self.register_buffer("occs", torch.zeros(resolution**3))
        self.register_buffer("binaries", torch.zeros((1, resolution, resolution, resolution), dtype=torch.bool))
        self.res = resolution
        self.jitter = None  # injected by the golden script ("identical inputs")

    def sampling(self, rays_o, rays_d, sigma_fn=None, alpha_fn=None, near_plane=0.0, far_plane=1e10, t_min=None,
                 t_max=None, render_step_size=1e-3, early_stop_eps=1e-4, alpha_thre=0.0, stratified=False, cone_angle=0.0):
        c = O.MarchCfg()
        aabb = self.aabbs[0].tolist()
        for i in range(6):
            c.aabb[i] = aabb[i]
        c.resolution, c.near_plane, c.far_plane, c.step = self.res, near_plane, far_plane, render_step_size
        diag = sum((aabb[3 + i] - aabb[i]) ** 2 for i in range(3)) ** 0.5
        c.max_steps = int(diag / float(render_step_size)) + 3
        bits = O.pack_bits(self.binaries.numpy())
        jit = self.jitter if stratified else None
        count, offset, ray_idx, t0, t1, pts = O.march(c, rays_o.numpy(), rays_d.numpy(), bits, jit)
        ray_idx_t, t0_t, t1_t = torch.from_numpy(ray_idx).long(), torch.from_numpy(t0), torch.from_numpy(t1)
        if sigma_fn is not None and (early_stop_eps > 0 or alpha_thre > 0):
            alpha_thre = min(alpha_thre, float(self.occs.mean().item()))
            sigmas = sigma_fn(t0_t, t1_t, ray_idx_t) if t0.shape[0] else torch.zeros(0)
            keep, kept = O.prune(sigmas.detach().numpy(), t0, t1, offset, count, early_stop_eps, alpha_thre)
            m = torch.from_numpy(keep.astype(bool))
            return ray_idx_t[m], t0_t[m], t1_t[m]
        return ray_idx_t, t0_t, t1_t

    def update_every_n_steps(self, step, occ_eval_fn, occ_thre=1e-2, ema_decay=0.95, warmup_steps=256, n=16):
        raise NotImplementedError("goldens set occs/binaries directly")


def install():
    """Register the stubs and make `threestudio` / `extern` importable from /root/reference."""
    if "threestudio" in sys.modules:
        return
    _mod("jaxtyping", **{k: _Sub for k in ["Bool", "Complex", "Float", "Inexact", "Int", "Integer", "Num", "Shaped", "UInt"]})
    _mod("typeguard", typechecked=lambda f=None, **k: f)

    class OmegaConf:
        @staticmethod
        def register_new_resolver(*a, **k):
            pass

        @staticmethod
        def structured(obj):
            return obj

        @staticmethod
        def to_container(c, resolve=True):
            return dict(c) if isinstance(c, dict) else c

        @staticmethod
        def load(path):
            raise NotImplementedError

    _mod("omegaconf", OmegaConf=OmegaConf, DictConfig=dict)
    _mod("omegaconf.listconfig", ListConfig=list)
    _mod("igl", fast_winding_number_for_meshes=None, point_mesh_squared_distance=None, read_obj=None)
    _mod("pytorch_lightning")
    _mod("pytorch_lightning.utilities")
    ident = lambda f: f
    _mod("pytorch_lightning.utilities.rank_zero", rank_zero_only=ident, rank_zero_info=print, rank_zero_debug=lambda *a, **k: None)
    _mod("tinycudann", Encoding=OracleEncoding, free_temporary_memory=lambda: None)
    na = _mod("nerfacc", OccGridEstimator=OracleOccGridEstimator, render_weight_from_density=render_weight_from_density,
              accumulate_along_rays=accumulate_along_rays, PropNetEstimator=None)
    _mod("nerfacc.data_specs", RayIntervals=None)
    _mod("nerfacc.estimators")
    _mod("nerfacc.estimators.base", AbstractEstimator=nn.Module)
    _mod("nerfacc.pdf", importance_sampling=None, searchsorted=None)
    _mod("nerfacc.volrend", render_transmittance_from_density=None)
    na.__path__ = []

    registry = {}

    def register(name):
        def deco(cls):
            registry[name] = cls
            return cls
        return deco

    ts = _mod("threestudio", register=register, find=lambda n: registry[n], info=print, warn=print, debug=lambda *a, **k: None,
              __modules__=registry)
    ts.__path__ = [os.path.join(REFERENCE, "threestudio")]
    ex = _mod("extern")
    ex.__path__ = [os.path.join(REFERENCE, "extern")]
    exm = _mod("extern.mvdream")  # bypass extern/mvdream/__init__.py (model_zoo needs real OmegaConf)
    exm.__path__ = [os.path.join(REFERENCE, "extern", "mvdream")]
    # synthetic package objects: the reference's package __init__ files import every sibling eagerly
    # (exporters -> cv2, guidance -> diffusers ...); we want single modules imported in place instead
    for pkg in ["models", "models.geometry", "models.renderers", "models.background", "models.materials",
                "models.guidance", "models.prompt_processors", "data"]:
        m = _mod("threestudio." + pkg)
        m.__path__ = [os.path.join(REFERENCE, "threestudio", *pkg.split("."))]
    # light stand-ins for heavy sub-modules the geometry base pulls in but the hot path never calls
    _mod("threestudio.models.isosurface", IsosurfaceHelper=object, MarchingCubeCPUHelper=object, MarchingTetrahedraHelper=object)
    _mod("threestudio.models.mesh", Mesh=object)
    _mod("threestudio.systems")
    sys.modules["threestudio.systems"].__path__ = []
    _mod("threestudio.systems.utils", parse_optimizer=None, parse_scheduler_to_instance=None)

    torch.cuda.device = lambda *a, **k: contextlib.nullcontext()
    import threestudio.utils.config as tcfg
    import threestudio.utils.misc as tmisc

    tmisc.get_device = lambda: torch.device("cpu")
    tmisc.get_rank = lambda: 0

    def parse_structured(fields, cfg=None):
        obj = fields(**dict(cfg or {}))
        for f in dataclasses.fields(obj):  # free-form sub-configs get attribute access like a DictConfig
            setattr(obj, f.name, _to_attr(getattr(obj, f.name)))
        return obj

    tcfg.parse_structured = parse_structured
    tcfg.config_to_primitive = lambda c, resolve=True: c
    import threestudio.utils.base as tbase

    tbase.parse_structured = parse_structured
    tbase.get_device = lambda: torch.device("cpu")
    import threestudio.models.networks as tnet

    tnet.config_to_primitive = lambda c, resolve=True: c
    tnet.get_rank = lambda: 0


# ---- amortized path (custom/amortized): oracle-backed nerfacc.pdf / volrend / render_weight_from_alpha ----------
class _OracleAlphaWeights(torch.autograd.Function):
    @staticmethod
    def forward(ctx, alphas, offset, count):
        n, nr = alphas.shape[0], count.shape[0]
        z, z3, bg = np.zeros(n, np.float32), np.zeros((n, 3), np.float32), np.zeros((nr, 3), np.float32)
        out = O.composite_fwd(alphas.detach().numpy(), z, z, z3, offset, count, bg, mode=1)
        ctx.stuff = (alphas.detach().numpy(), z, z3, offset, count, bg, out)
        return torch.from_numpy(out["weights"])

    @staticmethod
    def backward(ctx, dw):
        a, z, z3, offset, count, bg, out = ctx.stuff
        d_alpha, _, _ = O.composite_bwd(a, z, z, z3, offset, count, bg, out, d_weights=dw.contiguous().numpy(), mode=1)
        return torch.from_numpy(d_alpha), None, None


def render_weight_from_alpha(alphas, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None):
    offset, count = _packed(ray_indices, n_rays)
    w = _OracleAlphaWeights.apply(alphas, offset, count)
    with torch.no_grad():
        trans = w / alphas.clamp_min(1e-10)
    return w, trans


@dataclasses.dataclass
class RayIntervals:
    vals: torch.Tensor


IMPORTANCE_JITTER = []   # golden scripts push one [n_rays] jitter array per importance_sampling call (stratified)
IMPORTANCE_LOG = []


def importance_sampling(intervals, cdfs, n_intervals_per_ray, stratified=False):
    jit = IMPORTANCE_JITTER.pop(0) if stratified else None
    out = O.importance_resample(intervals.vals.numpy(), cdfs.detach().numpy(), int(n_intervals_per_ray), jit)
    IMPORTANCE_LOG.append(dict(vals=intervals.vals.numpy().copy(), cdfs=cdfs.detach().numpy().copy(), n=int(n_intervals_per_ray),
                               jitter=None if jit is None else jit.copy(), out=out.copy()))
    t = torch.from_numpy(out)
    return RayIntervals(vals=t), RayIntervals(vals=0.5 * (t[:, 1:] + t[:, :-1]))


def render_transmittance_from_density(t_starts, t_ends, sigmas, packed_info=None, ray_indices=None, n_rays=None, prefix_trans=None):
    sd = sigmas * (t_ends - t_starts)
    alphas = 1.0 - torch.exp(-sd)
    trans = torch.exp(-(torch.cumsum(sd, dim=-1) - sd))   # exclusive cumulative sum
    return trans, alphas


def install_amortized():
    """extra stubs for custom/amortized (Hyper-iNGP, VolSDF renderer, importance estimator, hyper background)."""
    install()
    sys.modules["omegaconf"].ListConfig = list
    _mod("omegaconf.dictconfig", DictConfig=dict)
    na = sys.modules["nerfacc"]
    na.render_weight_from_alpha = render_weight_from_alpha
    sys.modules["nerfacc.data_specs"].RayIntervals = RayIntervals

    class AbstractEstimator(nn.Module):
        @property
        def device(self):
            return torch.device("cpu")
    sys.modules["nerfacc.estimators.base"].AbstractEstimator = AbstractEstimator
    sys.modules["nerfacc.pdf"].importance_sampling = importance_sampling
    sys.modules["nerfacc.pdf"].searchsorted = None
    sys.modules["nerfacc.volrend"].render_transmittance_from_density = render_transmittance_from_density
    cu = _mod("custom")
    cu.__path__ = [os.path.join(REFERENCE, "custom")]
    for pkg in ["amortized", "amortized.models", "amortized.models.geometry", "amortized.models.renderers",
                "amortized.models.background", "amortized.extern"]:
        m = _mod("custom." + pkg)
        m.__path__ = [os.path.join(REFERENCE, "custom", *pkg.split("."))]
    import threestudio.utils.misc as tmisc

    tmisc.broadcast = lambda t, src=0: t
    import threestudio as ts

    ts.error = print
