"""TEST INFRASTRUCTURE ONLY — never imported by the product path.

Makes the reference's own model code (`/root/reference/models_video/*.py`) importable, read-only
and unmodified, in a container that lacks its third-party dependencies.  Used by
`oracle/make_golden.py` (in the build container, where /root/reference exists) to
  (a) pin the CPU restatement `oracle/uav_oracle.py` against the reference's own arithmetic and
  (b) generate the golden vectors committed under tests/golden/.

What is restated here is third-party GLUE that is absent from /root/reference (pinned versions
from /root/reference/requirements.txt): diffusers==0.16.0 (ConfigMixin / ModelMixin /
Timesteps / TimestepEmbedding / DDPMScheduler.add_noise / randn_tensor / DiffusionPipeline
plumbing), rotary-embedding-torch==0.2.3 (RotaryEmbedding), plus empty torchvision / imageio /
cv2 modules.  The diffusers classes `AttentionBlock`, `FeedForward`, `GEGLU` are NOT restated:
they are exec'd from the reference's own vendored spec copy
(models_video/diffusers_attention.py:249-381 and :735-858).
None of these packages can be diffed offline, so agreement with *released weights* rests on this
restatement; oracle<->engine parity with synthetic weights is self-consistent.
"""
import enum
import inspect
import json
import math
import os
import sys
import types
from collections import OrderedDict
from dataclasses import fields, is_dataclass

import torch
import torch.nn as nn
import torch.nn.functional as F

REFERENCE_ROOT = os.environ.get("UAV_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "models_video"))


class _AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kw):
        if not hasattr(self, "_internal_dict"):
            object.__setattr__(self, "_internal_dict", _AttrDict())
        self._internal_dict.update(kw)

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kwargs):
        if isinstance(config, (str, os.PathLike)):
            with open(config) as fh:
                config = json.load(fh)
        sig = inspect.signature(cls.__init__).parameters
        init = {k: v for k, v in dict(config).items() if k in sig and not k.startswith("_")}
        init.update({k: v for k, v in kwargs.items() if k in sig})
        return cls(**init)


def register_to_config(init):
    def wrapper(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self" and not k.startswith("_")}
        ConfigMixin.register_to_config(self, **cfg)
        init(self, *args, **kwargs)
    wrapper.__wrapped__ = init
    wrapper.__signature__ = inspect.signature(init)
    return wrapper


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype

    @property
    def device(self):
        return next(self.parameters()).device


class BaseOutput(OrderedDict):
    """dataclass-backed ordered dict with attribute + index access (diffusers.utils.BaseOutput)."""

    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def __getitem__(self, k):
        if isinstance(k, str):
            return dict(self.items())[k]
        return self.to_tuple()[k]

    def __setattr__(self, name, value):
        if name in self.keys() and value is not None:
            super().__setitem__(name, value)
        super().__setattr__(name, value)

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


# ---- diffusers.models.embeddings (restated from diffusers 0.16.0) ------------------------------
def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1, scale=1,
                           max_period=10000):
    half_dim = embedding_dim // 2
    exponent = -math.log(max_period) * torch.arange(start=0, end=half_dim, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half_dim - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half_dim:], emb[:, :half_dim]], dim=-1)
    if embedding_dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, timesteps):
        return get_timestep_embedding(timesteps, self.num_channels, flip_sin_to_cos=self.flip_sin_to_cos,
                                      downscale_freq_shift=self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim, act_fn="silu", out_dim=None):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, out_dim if out_dim is not None else time_embed_dim)

    def forward(self, sample):
        return self.linear_2(self.act(self.linear_1(sample)))


# ---- rotary-embedding-torch 0.2.3 (restated) ---------------------------------------------------
class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)
        self.dim = dim

    @staticmethod
    def _rotate_half(x):
        x = x.reshape(*x.shape[:-1], -1, 2)
        x1, x2 = x.unbind(dim=-1)
        return torch.stack((-x2, x1), dim=-1).reshape(*x.shape[:-2], -1)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        seq_len = t.shape[seq_dim]
        pos = torch.arange(seq_len, device=t.device).type_as(self.freqs)
        freqs = torch.einsum("..., f -> ... f", pos, self.freqs)
        freqs = freqs.repeat_interleave(2, dim=-1)                     # (n, dim)
        rot_dim = freqs.shape[-1]
        t_left, t_mid, t_right = t[..., :0], t[..., :rot_dim], t[..., rot_dim:]
        freqs = freqs.to(t)
        t_mid = (t_mid * freqs.cos()) + (self._rotate_half(t_mid) * freqs.sin())
        return torch.cat((t_left, t_mid, t_right), dim=-1)


# ---- schedulers / utils -------------------------------------------------------------------------
def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.randn_tensor: a CPU generator draws on CPU, then moves to `device`."""
    rand_device = device
    if generator is not None:
        gen_device = generator.device.type if not isinstance(generator, list) else generator[0].device.type
        if gen_device == "cpu":
            rand_device = "cpu"
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(device)


class SchedulerMixin:
    pass


class KarrasDiffusionSchedulers(enum.Enum):
    DDIMScheduler = 1
    DDPMScheduler = 2


class DDPMScheduler(SchedulerMixin, ConfigMixin):
    """Only what the pipeline uses: `add_noise` (same math as scheduling_ddim.py:524-545)."""

    @register_to_config
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear"):
        if beta_schedule == "linear":
            self.betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            self.betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        else:
            raise NotImplementedError(beta_schedule)
        self.alphas_cumprod = torch.cumprod(1.0 - self.betas, dim=0)

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        while len(sa.shape) < len(original_samples.shape):
            sa = sa.unsqueeze(-1)
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while len(sb.shape) < len(original_samples.shape):
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise


class DiffusionPipeline(ConfigMixin):
    def register_modules(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    @property
    def device(self):
        for m in (getattr(self, "unet", None), getattr(self, "vae", None)):
            if isinstance(m, nn.Module):
                return next(m.parameters()).device
        return torch.device("cpu")

    def to(self, device):
        return self


class StableDiffusionPipelineOutput(BaseOutput):
    def __init__(self, images=None, nsfw_content_detected=None):
        super().__init__()
        self.images = images
        self.nsfw_content_detected = nsfw_content_detected
        self["images"] = images


class _Logger:
    def __getattr__(self, k):
        return lambda *a, **kw: None


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_INSTALLED = False


def install():
    """Register the stub modules and put the reference on sys.path.  Idempotent."""
    global _INSTALLED
    if _INSTALLED:
        return
    if not reference_available():
        raise RuntimeError(f"{REFERENCE_ROOT} is not available: the reference can only be imported in the build "
                           "container; on the GPU box use tests/golden/ and oracle/uav_oracle.py")
    # transformers must be imported before a spec-less torchvision stub exists
    try:
        from transformers import CLIPImageProcessor, CLIPTextModel, CLIPTokenizer  # noqa: F401
    except Exception:
        pass
    logging = types.SimpleNamespace(get_logger=lambda name=None: _Logger())
    _mod("diffusers")
    _mod("diffusers.configuration_utils", ConfigMixin=ConfigMixin, register_to_config=register_to_config)
    _mod("diffusers.models")
    _mod("diffusers.models.modeling_utils", ModelMixin=ModelMixin)
    _mod("diffusers.models.embeddings", Timesteps=Timesteps, TimestepEmbedding=TimestepEmbedding,
         ImagePositionalEmbeddings=type("ImagePositionalEmbeddings", (nn.Module,), {}))
    _mod("diffusers.models.attention_processor", Attention=type("Attention", (nn.Module,), {}))
    _mod("diffusers.utils", BaseOutput=BaseOutput, logging=logging, randn_tensor=randn_tensor,
         apply_forward_hook=lambda f: f, deprecate=lambda *a, **k: None,
         is_accelerate_available=lambda: False, is_accelerate_version=lambda *a: False)
    _mod("diffusers.utils.import_utils", is_xformers_available=lambda: False)
    _mod("diffusers.schedulers", DDPMScheduler=DDPMScheduler)
    _mod("diffusers.schedulers.scheduling_utils", SchedulerMixin=SchedulerMixin,
         KarrasDiffusionSchedulers=KarrasDiffusionSchedulers)
    _mod("diffusers.loaders", TextualInversionLoaderMixin=type(
        "TextualInversionLoaderMixin", (), {"maybe_convert_prompt": lambda self, prompt, tokenizer: prompt}))
    _mod("diffusers.pipelines")
    _mod("diffusers.pipelines.pipeline_utils", DiffusionPipeline=DiffusionPipeline)
    _mod("diffusers.pipelines.stable_diffusion", StableDiffusionPipelineOutput=StableDiffusionPipelineOutput)
    _mod("rotary_embedding_torch", RotaryEmbedding=RotaryEmbedding)
    _mod("torchvision")
    _mod("torchvision.ops", deform_conv2d=None)
    _mod("imageio")
    cv2 = _mod("cv2", setNumThreads=lambda n: None)
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    # diffusers.models.attention: exec the reference's own vendored spec of AttentionBlock / FeedForward / GEGLU
    att = _mod("diffusers.models.attention")
    att.__dict__.update(dict(torch=torch, nn=nn, F=F, math=math, Optional=__import__("typing").Optional,
                             is_xformers_available=lambda: False, xformers=None))
    src = open(os.path.join(REFERENCE_ROOT, "models_video", "diffusers_attention.py")).read().split("\n")
    code = "\n".join(src[248:381]) + "\n" + "\n".join(src[734:858])
    exec(compile(code, "diffusers_attention_spec", "exec"), att.__dict__)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    _INSTALLED = True


def import_reference():
    """Returns a namespace with the reference classes used by the hot path."""
    install()
    # the drop-in package in upscale-a-video_amd/ is also called `models_video`: make sure the
    # reference's package is the one imported here.
    for k in [k for k in sys.modules if k == "models_video" or k.startswith("models_video.")]:
        if REFERENCE_ROOT not in (getattr(sys.modules[k], "__file__", "") or ""):
            del sys.modules[k]
    pkg = types.ModuleType("models_video")
    pkg.__path__ = [os.path.join(REFERENCE_ROOT, "models_video")]
    sys.modules["models_video"] = pkg
    import importlib
    ns = types.SimpleNamespace()
    ns.unet_video = importlib.import_module("models_video.unet_video")
    ns.attention = importlib.import_module("models_video.attention")
    ns.resnet = importlib.import_module("models_video.resnet")
    ns.temporal_module = importlib.import_module("models_video.temporal_module")
    ns.vae = importlib.import_module("models_video.autoencoder_kl_cond_video")
    ns.scheduling_ddim = importlib.import_module("models_video.scheduling_ddim")
    ns.propagation = importlib.import_module("models_video.propagation_module")
    ns.raft_bi = importlib.import_module("models_video.RAFT.raft_bi")
    ns.raft = importlib.import_module("models_video.RAFT.raft")
    pkg.AutoencoderKLVideo = ns.vae.AutoencoderKLVideo
    pkg.UNetVideoModel = ns.unet_video.UNetVideoModel
    pkg.Propagation = ns.propagation.Propagation
    ns.pipeline = importlib.import_module("models_video.pipeline_upscale_a_video")
    if "torchvision.transforms" not in sys.modules:        # color_correction.py imports two names it never uses
        _mod("torchvision.transforms", ToTensor=None, ToPILImage=None)
    ns.color = importlib.import_module("models_video.color_correction")
    return ns
