"""`stable-diffusion-asynchronous-score-distillation-guidance`
(threestudio/models/guidance/stable_diffusion_asd_guidance.py:24-440) with the diffusion prior behind a small
backend interface, plus the prompt-side helper the guidance consumes
(threestudio/models/prompt_processors/base.py:38-167 `PromptProcessorOutput`).

The reference loads diffusers' StableDiffusionPipeline (fp16).  Here the frozen UNet / VAE encoder are a
`DiffusionBackend`:  `unet(latents[N,4,h,w], t[N], context[N,77,1024]) -> eps`  (no grad) and
`encode(images[B,3,H,W] in [-1,1]) -> moments[B,8,H/8,W/8]`  (differentiable w.r.t. the images).  The
innermost seams are the reference's own: forward_unet (:319-331) and encode_images (:171-178).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

from .base import BaseObject
from .config import C
from .registry import info, register


def perpendicular_component(x: torch.Tensor, y: torch.Tensor) -> torch.Tensor:
    """threestudio/utils/ops.py:501-511"""
    eps = torch.ones_like(x[:, 0, 0, 0]) * 1e-6
    return x - (torch.mul(x, y).sum(dim=[1, 2, 3]) / torch.maximum(torch.mul(y, y).sum(dim=[1, 2, 3]), eps)).view(-1, 1, 1, 1) * y


def shift_azimuth_deg(azimuth: torch.Tensor) -> torch.Tensor:
    return (azimuth + 180) % 360 - 180


def shifted_expotional_decay(a, b, c, r):
    return a * torch.exp(-b * r) + c


def ddpm_alphas_cumprod(n: int = 1000, beta_start: float = 0.00085, beta_end: float = 0.012) -> torch.Tensor:
    """scaled-linear schedule of SD (in-tree: extern/mvdream/ldm/interface.py:48-76)."""
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float64) ** 2
    return torch.cumprod(1.0 - betas, dim=0).float()


_RESIZE_MATS = {}


def _resize_matrix(n_in: int, n_out: int, device) -> torch.Tensor:
    """[n_out, n_in] matrix of F.interpolate(mode='bilinear', align_corners=False) along one axis, taken from torch itself."""
    key = (n_in, n_out, str(device))
    if key not in _RESIZE_MATS:
        eye = torch.eye(n_in, device=device, dtype=torch.float32).view(1, n_in, n_in, 1)
        _RESIZE_MATS[key] = F.interpolate(eye, (n_out, 1), mode="bilinear", align_corners=False)[0, :, :, 0].t().contiguous()
    return _RESIZE_MATS[key]


class _BilinearResize(torch.autograd.Function):
    """F.interpolate(x, size, mode='bilinear', align_corners=False) with the same forward kernel and a backward written as the two
    small matmuls  d_x = R_h^T d_y R_w  (bilinear resampling is separable): ATen's upsample_bilinear2d_backward takes 112 us for a
    64x64 -> 512x512 image on MI355X, the matmuls ~20 us (profiles/r01_final3_kernel_stats_top70.csv)."""

    @staticmethod
    def forward(ctx, x, size):
        ctx.in_hw = (x.shape[-2], x.shape[-1])
        ctx.size = size
        return F.interpolate(x, size, mode="bilinear", align_corners=False)

    @staticmethod
    def backward(ctx, dy):
        rh = _resize_matrix(ctx.in_hw[0], ctx.size[0], dy.device)
        rw = _resize_matrix(ctx.in_hw[1], ctx.size[1], dy.device)
        dx = torch.matmul(torch.matmul(rh.t(), dy.float()), rw)
        return dx.to(dy.dtype), None


def resize_bilinear(x: torch.Tensor, size) -> torch.Tensor:
    if x.is_cuda and x.requires_grad and x.dtype == torch.float32:
        return _BilinearResize.apply(x, tuple(size))
    return F.interpolate(x, size, mode="bilinear", align_corners=False)


@dataclass
class PromptUtils:
    """Output layout of the prompt processors: view-dependent embeddings [4,77,1024] in the order
    side / front / back / overhead (prompt_processors/base.py:100-104) + the Perp-Neg interpolation rules."""
    text_embeddings_vd: torch.Tensor
    uncond_text_embeddings_vd: torch.Tensor
    text_embeddings: Optional[torch.Tensor] = None
    uncond_text_embeddings: Optional[torch.Tensor] = None
    use_perp_neg: bool = True
    overhead_threshold: float = 60.0
    front_threshold: float = 45.0
    back_threshold: float = 45.0
    perp_neg_f_sb: Tuple[float, float, float] = (1, 0.5, -0.606)
    perp_neg_f_fsb: Tuple[float, float, float] = (1, 0.5, +0.967)
    perp_neg_f_fs: Tuple[float, float, float] = (4, 0.5, -2.426)
    perp_neg_f_sf: Tuple[float, float, float] = (4, 0.5, -2.426)

    @staticmethod
    def synthetic(seed: int = 1234, device="cpu", **kw) -> "PromptUtils":
        """ "synthetic random prompts" (BASELINE.json): N(0,1) embeddings, one uncond row."""
        g = torch.Generator().manual_seed(seed)
        vd = torch.randn(4, 77, 1024, generator=g)
        un = torch.randn(1, 77, 1024, generator=g).expand(4, -1, -1).contiguous()
        return PromptUtils(vd.to(device), un.to(device), vd[0].to(device), un[0].to(device), **kw)

    def direction_idx(self, elevation, azimuth, camera_distances):
        """later directions override earlier ones: side < front < back < overhead (base.py:262-294)."""
        idx = torch.zeros_like(elevation, dtype=torch.long)
        azi = shift_azimuth_deg(azimuth)
        idx = torch.where((azi > -self.front_threshold) & (azi < self.front_threshold), torch.ones_like(idx), idx)
        idx = torch.where((azi > 180 - self.back_threshold) | (azi < -180 + self.back_threshold), torch.full_like(idx, 2), idx)
        return torch.where(elevation > self.overhead_threshold, torch.full_like(idx, 3), idx)

    def get_text_embeddings(self, elevation, azimuth, camera_distances, view_dependent_prompting: bool = True):
        batch_size = elevation.shape[0]
        if view_dependent_prompting:
            idx = self.direction_idx(elevation, azimuth, camera_distances)
            text, uncond = self.text_embeddings_vd[idx], self.uncond_text_embeddings_vd[idx]
        else:
            text = self.text_embeddings.expand(batch_size, -1, -1)
            uncond = self.uncond_text_embeddings.expand(batch_size, -1, -1)
        return torch.cat([text, uncond], dim=0)  # (cond, uncond): the reference's order

    def get_text_embeddings_perp_neg(self, elevation, azimuth, camera_distances, view_dependent_prompting: bool = True):
        """prompt_processors/base.py:82-167.  The reference walks the batch in Python and branches on device scalars (one host
        sync per `if`); here the same selection is written with torch.where so that nothing between the VAE forward and the
        UNet launch waits for the GPU.  Output order as the reference: [pos (B), uncond (B), neg (2B: n1_0, n2_0, n1_1, ...)]."""
        assert view_dependent_prompting, "Perp-Neg only works with view-dependent prompting"
        batch_size = elevation.shape[0]
        idx = self.direction_idx(elevation, azimuth, camera_distances)
        side, front, back, overhead = (self.text_embeddings_vd[i] for i in range(4))
        a = torch.abs(shift_azimuth_deg(azimuth)).to(side.dtype)
        over = (idx == 3).view(-1, 1, 1)
        is_front = (a < 90).view(-1, 1, 1)
        r_f = (1 - a / 90).view(-1, 1, 1)             # front <-> side
        r_b = (2.0 - a / 90).view(-1, 1, 1)           # side <-> back
        uncond = self.uncond_text_embeddings_vd[idx]
        pos = torch.where(over, overhead.expand(batch_size, -1, -1),
                          torch.where(is_front, r_f * front + (1 - r_f) * side, r_b * side + (1 - r_b) * back))
        neg1 = torch.where(over, uncond, torch.where(is_front, front.expand(batch_size, -1, -1), side.expand(batch_size, -1, -1)))
        neg2 = torch.where(over, uncond, torch.where(is_front, side.expand(batch_size, -1, -1), front.expand(batch_size, -1, -1)))
        rf, rb = r_f.view(-1), r_b.view(-1)
        w1 = torch.where(a < 90, -shifted_expotional_decay(*self.perp_neg_f_fs, rf), -shifted_expotional_decay(*self.perp_neg_f_sb, rb))
        w2 = torch.where(a < 90, -shifted_expotional_decay(*self.perp_neg_f_sf, 1 - rf), -shifted_expotional_decay(*self.perp_neg_f_fsb, rb))
        zero = torch.zeros_like(w1)
        weights = torch.stack([torch.where(idx == 3, zero, w1), torch.where(idx == 3, zero, w2)], dim=1)
        neg = torch.stack([neg1, neg2], dim=1).reshape(2 * batch_size, *neg1.shape[1:])
        return torch.cat([pos, uncond, neg], dim=0), weights.to(torch.float32)


class DiffusionBackend:
    """Frozen SD-2.1 prior.  The product implementation is scaledreamer_amd.diffusion.engine (hand-written HIP); tests inject
    stand-ins through `configure(backend=...)`."""
    scaling_factor: float = 0.18215

    def unet(self, latents: torch.Tensor, t: torch.Tensor, context: torch.Tensor, camera: Optional[torch.Tensor] = None,
             num_frames: int = 1) -> torch.Tensor:
        raise NotImplementedError

    def encode(self, images: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError


_BACKEND_FACTORY: Dict[str, Callable[..., DiffusionBackend]] = {}


def register_backend(name: str):
    def deco(fn):
        _BACKEND_FACTORY[name] = fn
        return fn
    return deco


@register("stable-diffusion-asynchronous-score-distillation-guidance")
class SDTimestepShiftedScoreDistillationGuidance(BaseObject):
    @dataclass
    class Config(BaseObject.Config):
        pretrained_model_name_or_path: str = "stabilityai/stable-diffusion-2-1-base"
        enable_memory_efficient_attention: bool = False
        enable_sequential_cpu_offload: bool = False
        enable_attention_slicing: bool = False
        enable_channels_last_format: bool = True
        guidance_scale: float = 7.5
        grad_clip: Optional[Any] = None
        half_precision_weights: bool = True
        min_step_percent: Any = 0.02
        max_step_percent: Any = 0.98
        weighting_strategy: str = "sds"
        plus_ratio: float = 0.1
        plus_random: bool = False
        view_dependent_prompting: bool = True
        guidance_perp_neg: float = 0.0
        # build-specific: which DiffusionBackend executes the frozen prior, and the seed of the random-init
        # weights used when no checkpoint exists at pretrained_model_name_or_path (none exists offline)
        backend: str = "hip"
        weights_seed: int = 1
        allow_random_weights: bool = False     # without it a missing checkpoint is an error, never a silent random prior

    cfg: Config

    def configure(self, backend: Optional[DiffusionBackend] = None) -> None:
        info("Loading Stable Diffusion ...")
        self.weights_dtype = torch.float16 if self.cfg.half_precision_weights else torch.float32
        if backend is None:
            if self.cfg.backend not in _BACKEND_FACTORY:
                from .diffusion import engine  # noqa: F401  (registers "hip" / "hip-mvdream"; raises if libasd_hip.so is missing)
            backend = _BACKEND_FACTORY[self.cfg.backend](self.cfg, self.device, self.weights_dtype)
        self.backend = backend
        self.num_train_timesteps = 1000
        min_p = self.cfg.min_step_percent if isinstance(self.cfg.min_step_percent, (int, float)) else 0.02
        max_p = self.cfg.max_step_percent if isinstance(self.cfg.max_step_percent, (int, float)) else 0.98
        self.set_min_max_steps(min_p, max_p)
        self.alphas = ddpm_alphas_cumprod(self.num_train_timesteps).to(self.device)
        self.grad_clip_val: Optional[float] = None
        self.use_perp_neg = self.cfg.guidance_perp_neg != 0
        # RNG injection points: tests replace these for "identical inputs"
        self.noise_fn = torch.randn_like
        self.timestep_fn = lambda lo, hi, n, device: torch.randint(lo, hi, [n], dtype=torch.long, device=device)
        self.rand_fn = lambda shape, device: torch.rand(*shape, device=device)
        self.posterior_noise_fn = torch.randn_like
        info(f"Loaded Stable Diffusion! ({getattr(self.backend, 'weights_source', 'caller-supplied backend')})")

    def set_min_max_steps(self, min_step_percent=0.02, max_step_percent=0.98):
        self.min_step = int(self.num_train_timesteps * min_step_percent)
        self.max_step = int(self.num_train_timesteps * max_step_percent)

    # -- seams ------------------------------------------------------------------------------------
    def forward_unet(self, latents, t, encoder_hidden_states):
        input_dtype = latents.dtype
        return self.backend.unet(latents, t, encoder_hidden_states).to(input_dtype)

    def encode_images(self, imgs: torch.Tensor) -> torch.Tensor:
        input_dtype = imgs.dtype
        imgs = imgs * 2.0 - 1.0
        moments = self.backend.encode(imgs)
        mean, logvar = torch.chunk(moments.float(), 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        latents = (mean + torch.exp(0.5 * logvar) * self.posterior_noise_fn(mean)) * self.backend.scaling_factor
        return latents.to(input_dtype)

    def get_latents(self, rgb_BCHW: torch.Tensor, rgb_as_latents: bool = False) -> torch.Tensor:
        if rgb_as_latents:
            return F.interpolate(rgb_BCHW, (64, 64), mode="bilinear", align_corners=False)
        rgb_BCHW_512 = resize_bilinear(rgb_BCHW, (512, 512))
        return self.encode_images(rgb_BCHW_512)

    def add_noise(self, latents, noise, t):
        a = self.alphas.to(latents.device)[t].view(-1, 1, 1, 1)
        return a.sqrt() * latents + (1 - a).sqrt() * noise

    def get_t_plus(self, t: torch.Tensor) -> torch.Tensor:
        assert self.cfg.plus_ratio >= 0.0
        t_plus = self.cfg.plus_ratio * (t - self.min_step)
        t_plus = t_plus.clamp(torch.zeros_like(t), self.num_train_timesteps - t - 1)
        if self.cfg.plus_random:
            t_plus = t_plus * self.rand_fn(t.shape, t.device)
        t_plus = t + t_plus.to(torch.long)
        return torch.clamp(t_plus, 1, max=self.num_train_timesteps - 1)

    def get_eps(self, latents_noisy, latents_noisy_second, t, t_plus, prompt_utils, elevation, azimuth, camera_distances):
        batch_size = latents_noisy.shape[0]
        if self.use_perp_neg:
            assert prompt_utils.use_perp_neg
            text_embeddings, neg_w = prompt_utils.get_text_embeddings_perp_neg(
                elevation, azimuth, camera_distances, self.cfg.view_dependent_prompting)
            neg_w = neg_w * -1 * self.cfg.guidance_perp_neg
            vd, uncond = text_embeddings[0:batch_size], text_embeddings[batch_size:2 * batch_size]
            vd_neg = text_embeddings[2 * batch_size:4 * batch_size]
        else:
            text_embeddings = prompt_utils.get_text_embeddings(elevation, azimuth, camera_distances,
                                                               self.cfg.view_dependent_prompting)
            neg_w, vd_neg = None, None
            vd, uncond = text_embeddings[0:batch_size], text_embeddings[batch_size:2 * batch_size]
        parts = [vd, uncond] + ([vd_neg] if self.use_perp_neg else []) + [vd]
        text_embeddings = torch.cat(parts, dim=0).to(latents_noisy.device)
        num_repeats = text_embeddings.shape[0] // batch_size - 1
        input_t = torch.cat([t] * num_repeats + [t_plus], dim=0)
        input_latents = torch.cat([latents_noisy] * num_repeats + [latents_noisy_second], dim=0)
        with torch.no_grad():
            noise_pred = self.forward_unet(input_latents, input_t, encoder_hidden_states=text_embeddings)
        B = batch_size
        text, unc = noise_pred[0:B], noise_pred[B:2 * B]
        second = noise_pred[4 * B:5 * B] if self.use_perp_neg else noise_pred[2 * B:3 * B]
        eps_pos = text - unc
        if neg_w is not None:
            neg = noise_pred[2 * B:4 * B]
            accum = 0
            n_neg = neg_w.shape[-1]
            for i in range(n_neg):
                eps_neg = neg[i::n_neg] - unc
                accum = accum + neg_w[:, i].view(-1, *[1] * (eps_neg.ndim - 1)).to(eps_neg) * perpendicular_component(eps_neg, eps_pos)
            noise_pred_p = (eps_pos + accum) * self.cfg.guidance_scale + unc
        else:
            noise_pred_p = eps_pos * self.cfg.guidance_scale + unc
        return noise_pred_p, second

    def __call__(self, rgb: torch.Tensor, prompt_utils, elevation, azimuth, camera_distances, rgb_as_latents=False,
                 guidance_eval=False, **kwargs) -> Dict[str, Any]:
        batch_size = rgb.shape[0]
        rgb_BCHW = rgb.permute(0, 3, 1, 2)
        latents = self.get_latents(rgb_BCHW, rgb_as_latents=rgb_as_latents)
        noise = self.noise_fn(latents)  # shared by both timesteps
        assert self.min_step is not None and self.max_step is not None
        with torch.no_grad():
            t = self.timestep_fn(self.min_step, self.max_step + 1, batch_size, latents.device)
            latents_noisy = self.add_noise(latents, noise, t)
            t_plus = self.get_t_plus(t)
            latents_noisy_second = self.add_noise(latents, noise, t_plus)
            noise_pred, noise_pred_second = self.get_eps(latents_noisy, latents_noisy_second, t, t_plus, prompt_utils,
                                                         elevation, azimuth, camera_distances)
        alphas = self.alphas.to(latents.device)
        if self.cfg.weighting_strategy == "sds":
            w = (1 - alphas[t]).view(-1, 1, 1, 1)
        elif self.cfg.weighting_strategy == "uniform":
            w = 1
        elif self.cfg.weighting_strategy == "fantasia3d":
            w = (alphas[t] ** 0.5 * (1 - alphas[t])).view(-1, 1, 1, 1)
        else:
            raise ValueError(f"Unknown weighting strategy: {self.cfg.weighting_strategy}")
        grad = torch.nan_to_num(w * (noise_pred - noise_pred_second))
        if self.grad_clip_val is not None:
            grad = grad.clamp(-self.grad_clip_val, self.grad_clip_val)
        target = (latents - grad).detach()  # d(loss)/d(latents) = grad
        loss_sds = 0.5 * F.mse_loss(latents, target, reduction="sum") / batch_size
        return {"loss_asd": loss_sds, "grad_norm": grad.norm(), "min_step": self.min_step, "max_step": self.max_step}

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        if self.cfg.grad_clip is not None:
            self.grad_clip_val = C(self.cfg.grad_clip, epoch, global_step)
        self.set_min_max_steps(min_step_percent=C(self.cfg.min_step_percent, epoch, global_step),
                               max_step_percent=C(self.cfg.max_step_percent, epoch, global_step))


def normalize_camera(camera_matrix: torch.Tensor) -> torch.Tensor:
    """extern/mvdream/camera_utils.py:45-57: camera location projected onto the unit sphere, flattened to [B,16]."""
    cam = camera_matrix.reshape(-1, 4, 4).clone()
    tr = cam[:, :3, 3]
    cam[:, :3, 3] = tr / (torch.norm(tr, dim=1, keepdim=True) + 1e-8)
    return cam.reshape(-1, 16)


@register("mvdream-asynchronous-score-distillation-guidance")
class MVDreamTimestepShiftedScoreDistillationGuidance(BaseObject):
    """threestudio/models/guidance/mvdream_asd_guidance.py:26-304: 4-view groups share one timestep, the UNet batch is
    [x_t | x_t | x_t+] with contexts [cond | uncond | cond], camera-conditioned with cross-view self-attention, plain CFG
    (no Perp-Neg), VAE at 256x256 -> 32x32 latents."""

    @dataclass
    class Config(BaseObject.Config):
        model_name: str = "sd-v2.1-base-4view"
        ckpt_path: Optional[str] = None
        grad_clip: Optional[Any] = None
        half_precision_weights: bool = True
        guidance_scale: float = 7.5
        n_view: int = 4
        min_step_percent: Any = 0.02
        max_step_percent: Any = 0.98
        weighting_strategy: str = "sds"
        plus_ratio: float = 0.1
        plus_random: bool = False
        camera_condition_type: str = "rotation"
        view_dependent_prompting: bool = False
        backend: str = "hip-mvdream"
        weights_seed: int = 1
        allow_random_weights: bool = False

    cfg: Config

    def configure(self, backend: Optional[DiffusionBackend] = None) -> None:
        info("Loading Multiview Diffusion ...")
        if backend is None:
            if self.cfg.backend not in _BACKEND_FACTORY:
                from .diffusion import engine  # noqa: F401
            # The reference builds this model in fp32 (mvdream_asd_guidance.py:40,67: `half_precision_weights` is declared but never
            # applied).  The HIP engine computes in fp16 operands with fp32 accumulation for both priors; the deviation from the
            # fp32 reference is bounded at full width by tests/test_gpu_unet_engine.py::test_full_mvdream_unet_b12_matches_reference_golden
            # (< 1e-2, north_star's tolerance) and reported as such by bench.py (`dtype`).
            backend = _BACKEND_FACTORY[self.cfg.backend](self.cfg, self.device, torch.float16)
        self.backend = backend
        self.weights_dtype = torch.float16
        self.num_train_timesteps = 1000
        self.alphas = ddpm_alphas_cumprod(self.num_train_timesteps).to(self.device)
        min_p = self.cfg.min_step_percent if isinstance(self.cfg.min_step_percent, (int, float)) else 0.02
        max_p = self.cfg.max_step_percent if isinstance(self.cfg.max_step_percent, (int, float)) else 0.98
        self.min_step, self.max_step = int(self.num_train_timesteps * min_p), int(self.num_train_timesteps * max_p)
        self.grad_clip_val: Optional[float] = None
        self.noise_fn = torch.randn_like
        self.timestep_fn = lambda lo, hi, n, device: torch.randint(lo, hi, [n], dtype=torch.long, device=device)
        self.rand_fn = lambda shape, device: torch.rand(*shape, device=device)
        self.posterior_noise_fn = torch.randn_like

    def get_camera_cond(self, camera: torch.Tensor, fovy=None) -> torch.Tensor:
        if self.cfg.camera_condition_type != "rotation":
            raise NotImplementedError(f"Unknown camera_condition_type={self.cfg.camera_condition_type}")
        return normalize_camera(camera).flatten(start_dim=1)

    def encode_images(self, imgs: torch.Tensor) -> torch.Tensor:
        imgs = imgs * 2.0 - 1.0
        moments = self.backend.encode(imgs)
        mean, logvar = torch.chunk(moments.float(), 2, dim=1)
        logvar = torch.clamp(logvar, -30.0, 20.0)
        return (mean + torch.exp(0.5 * logvar) * self.posterior_noise_fn(mean)) * self.backend.scaling_factor

    def get_latents(self, rgb_BCHW: torch.Tensor, rgb_as_latents: bool = False) -> torch.Tensor:
        if rgb_as_latents:
            return F.interpolate(rgb_BCHW, size=(32, 32), mode="bilinear", align_corners=False)
        return self.encode_images(resize_bilinear(rgb_BCHW, (256, 256)))

    def q_sample(self, x, t, noise):
        a = self.alphas.to(x.device)[t].view(-1, 1, 1, 1)
        return a.sqrt() * x + (1 - a).sqrt() * noise

    def get_t_plus(self, t: torch.Tensor) -> torch.Tensor:
        assert self.cfg.plus_ratio >= 0.0
        t_plus = self.cfg.plus_ratio * (t - self.min_step)
        t_plus = t_plus.clamp(torch.zeros_like(t), self.num_train_timesteps - t - 1)
        if self.cfg.plus_random:
            t_plus = t_plus * self.rand_fn(t.shape, t.device)
        return torch.clamp(t + t_plus.to(torch.long), 1, max=self.num_train_timesteps - 1)

    def __call__(self, rgb: torch.Tensor, prompt_utils, elevation, azimuth, camera_distances, c2w, rgb_as_latents: bool = False,
                 fovy=None, input_is_latent=False, **kwargs) -> Dict[str, Any]:
        camera = c2w
        batch_size = rgb.shape[0]
        latents = self.get_latents(rgb.permute(0, 3, 1, 2), rgb_as_latents=rgb_as_latents)
        noise = self.noise_fn(latents)
        text_embeddings = prompt_utils.get_text_embeddings(elevation, azimuth, camera_distances, self.cfg.view_dependent_prompting)
        tb = text_embeddings.shape[0] // 2
        vd = text_embeddings[0:tb].repeat(batch_size // tb, 1, 1)
        uncond = text_embeddings[tb:2 * tb].repeat(batch_size // tb, 1, 1)
        text_embeddings = torch.cat([vd, uncond, vd], dim=0)
        with torch.no_grad():
            _t = self.timestep_fn(self.min_step, self.max_step + 1, 1, latents.device)   # one t shared by the group
            t = _t.repeat(batch_size)
            latents_noisy = self.q_sample(latents, t, noise)
            t_plus = self.get_t_plus(_t).repeat(batch_size)
            latents_noisy_second = self.q_sample(latents, t_plus, noise)
            x_in = torch.cat([latents_noisy, latents_noisy, latents_noisy_second], dim=0)
            t_in = torch.cat([t, t, t_plus], dim=0)
            if camera is not None:
                cam = self.get_camera_cond(camera, fovy).repeat(3, 1).to(text_embeddings)
                noise_pred = self.backend.unet(x_in, t_in, text_embeddings, camera=cam, num_frames=self.cfg.n_view)
            else:
                noise_pred = self.backend.unet(x_in, t_in, text_embeddings)
            noise_pred = noise_pred.to(latents.dtype)
        text, unc, second = noise_pred.chunk(3)
        first = unc + self.cfg.guidance_scale * (text - unc)
        alphas = self.alphas.to(latents.device)
        if self.cfg.weighting_strategy == "sds":
            w = (1 - alphas[t]).view(-1, 1, 1, 1)
        elif self.cfg.weighting_strategy == "uniform":
            w = 1
        elif self.cfg.weighting_strategy == "fantasia3d":
            w = (alphas[t] ** 0.5 * (1 - alphas[t])).view(-1, 1, 1, 1)
        else:
            raise ValueError(f"Unknown weighting strategy: {self.cfg.weighting_strategy}")
        grad = torch.nan_to_num((first - second) * w)
        if self.grad_clip_val is not None:
            grad = torch.clamp(grad, -self.grad_clip_val, self.grad_clip_val)
        target = (latents - grad).detach()
        loss = 0.5 * F.mse_loss(latents, target, reduction="sum") / batch_size
        return {"loss_asd": loss, "grad_norm": grad.norm(), "min_step": self.min_step, "max_step": self.max_step}

    def update_step(self, epoch: int, global_step: int, on_load_weights: bool = False):
        self.min_step = int(self.num_train_timesteps * C(self.cfg.min_step_percent, epoch, global_step))
        self.max_step = int(self.num_train_timesteps * C(self.cfg.max_step_percent, epoch, global_step))
