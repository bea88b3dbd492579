"""MI355X-native DDIMScheduler (drop-in for the reference's `models_video/scheduling_ddim.py`:
class :79, set_timesteps :237-259, step_v0 :383-433, step_vt :436-520, add_noise :524-545).

The reference evaluates the DDIM update as ~10 tiny ATen launches per step with fp32 CPU scalars
multiplied into fp16 tensors, plus two device->host syncs (indexing the CPU `alphas_cumprod` with a
device timestep, :404,:459).  Here every coefficient is a host float, the update is ONE fused
kernel per call (`uav_cfg_ddim_v0` also folds the classifier-free-guidance combine,
pipeline_upscale_a_video.py:643-645), computed in fp32 and rounded once, and `timesteps` stay on
the host so the loop never synchronises.
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

from uav import ops

from ._compat import BaseOutput, ConfigMixin, register_to_config


@dataclass
class DDIMSchedulerOutput(BaseOutput):
    prev_sample: Optional[torch.FloatTensor] = None
    pred_original_sample: Optional[torch.FloatTensor] = None


def _betas(schedule, beta_start, beta_end, n):
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(f"{schedule} is not implemented")


class DDIMScheduler(ConfigMixin):
    config_name = "scheduler_config.json"
    order = 1

    @register_to_config
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "linear", trained_betas=None, clip_sample: bool = True,
                 set_alpha_to_one: bool = True, steps_offset: int = 0, prediction_type: str = "epsilon",
                 thresholding: bool = False, dynamic_thresholding_ratio: float = 0.995, clip_sample_range: float = 1.0,
                 sample_max_value: float = 1.0):
        if thresholding:
            raise NotImplementedError("dynamic thresholding is unsuitable for latent diffusion and not built")
        if trained_betas is not None:
            self.betas = torch.tensor(trained_betas, dtype=torch.float32)
        else:
            self.betas = _betas(beta_schedule, beta_start, beta_end, num_train_timesteps)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError(f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                             f"`self.config.train_timesteps`: {self.config.num_train_timesteps}")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        # kept on the HOST on purpose (see module docstring); `device` is accepted for API parity
        self.timesteps = torch.from_numpy(ts) + self.config.steps_offset

    # ---- host-side coefficients -----------------------------------------------------------------
    def _alphas(self, timestep):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the scheduler")
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_prev = float(self.alphas_cumprod[prev]) if prev >= 0 else float(self.final_alpha_cumprod)
        return a_t, a_prev

    def v0_coefficients(self, timestep):
        """x0 = cs*sample + ce*model_output (scheduling_ddim.py:408-420)."""
        a_t, _ = self._alphas(timestep)
        b_t = 1.0 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            return 1.0 / a_t ** 0.5, -(b_t ** 0.5) / a_t ** 0.5
        if pt == "sample":
            return 0.0, 1.0
        if pt == "v_prediction":
            return a_t ** 0.5, -(b_t ** 0.5)
        raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")

    def vt_coefficients(self, timestep):
        """prev = c0*x0 + cd*(em*model + es*sample + e0*x0), eta = 0 (scheduling_ddim.py:464-500)."""
        a_t, a_prev = self._alphas(timestep)
        b_t = 1.0 - a_t
        pt = self.config.prediction_type
        if pt == "epsilon":
            em, es, e0 = 1.0, 0.0, 0.0
        elif pt == "sample":
            em, es, e0 = 0.0, 1.0 / b_t ** 0.5, -(a_t ** 0.5) / b_t ** 0.5
        elif pt == "v_prediction":
            em, es, e0 = a_t ** 0.5, b_t ** 0.5, 0.0
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")
        return a_prev ** 0.5, (1.0 - a_prev) ** 0.5, em, es, e0

    # ---- fused device steps ---------------------------------------------------------------------
    def cfg_step_v0(self, eps_uncond, eps_text, guidance_scale, timestep, sample):
        """CFG combine + step_v0 in one kernel.  Returns (guided model output, x0)."""
        cs, ce = self.v0_coefficients(timestep)
        return ops.cfg_ddim_v0(eps_uncond, eps_text, sample, guidance=float(guidance_scale), coef_sample=cs, coef_eps=ce,
                               clip=bool(self.config.clip_sample), clip_range=float(self.config.clip_sample_range))

    def step_v0(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
                generator=None, variance_noise=None, return_dict: bool = True):
        _, x0 = self.cfg_step_v0(_c(model_output, sample), None, 1.0, timestep, _c(sample))
        x0 = x0.to(sample.dtype)
        return DDIMSchedulerOutput(pred_original_sample=x0) if return_dict else (x0,)

    def step_vt(self, v0, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
                generator=None, variance_noise=None, return_dict: bool = True):
        if eta != 0.0 or use_clipped_model_output:
            raise NotImplementedError("the pipeline always steps with eta = 0")
        c0, cd, em, es, e0 = self.vt_coefficients(timestep)
        prev = ops.ddim_vt(_c(v0, sample), _c(model_output, sample), _c(sample), coef_x0=c0, coef_dir=cd, eps_from_model=em,
                           eps_from_sample=es, eps_from_x0=e0, clip=bool(self.config.clip_sample),
                           clip_range=float(self.config.clip_sample_range)).to(sample.dtype)
        return DDIMSchedulerOutput(prev_sample=prev) if return_dict else (prev,)

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True):
        x0 = self.step_v0(model_output, timestep, sample).pred_original_sample
        prev = self.step_vt(x0, model_output, timestep, sample, eta).prev_sample
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0) if return_dict else (prev,)

    def add_noise(self, original_samples, noise, timesteps):
        a = float(self.alphas_cumprod[int(torch.as_tensor(timesteps).reshape(-1)[0])])
        return ops.axpby(_c(original_samples), _c(noise, original_samples), a ** 0.5, (1.0 - a) ** 0.5).to(original_samples.dtype)

    def __len__(self):
        return self.config.num_train_timesteps


def _h(t):
    return t.contiguous() if t.dtype == torch.float16 else t.half().contiguous()


def _c(t, like=None):
    """Scheduler operand: fp32 tensors stay fp32 (high-precision latents, UNet stream_dtype = float32), everything else
    is fp16 like the reference's half pipeline; `like` forces the dtype of the step's sample."""
    dt = (like.dtype if like is not None else t.dtype)
    dt = torch.float32 if dt == torch.float32 else torch.float16
    return t.to(dt).contiguous()


class DDPMScheduler(ConfigMixin):
    """`low_res_scheduler` of the pipeline: only `add_noise` is used (pipeline :548; math
    scheduling_ddim.py:524-545).  Defaults = SD-x4-upscaler low_res_scheduler config."""
    config_name = "scheduler_config.json"

    @register_to_config
    def __init__(self, num_train_timesteps: int = 1000, beta_start: float = 0.0001, beta_end: float = 0.02,
                 beta_schedule: str = "scaled_linear", **_unused):
        self.alphas_cumprod = torch.cumprod(1.0 - _betas(beta_schedule, beta_start, beta_end, num_train_timesteps), dim=0)

    def add_noise(self, original_samples, noise, timesteps):
        a = float(self.alphas_cumprod[int(torch.as_tensor(timesteps).reshape(-1)[0])])
        return ops.axpby(_c(original_samples), _c(noise, original_samples), a ** 0.5, (1.0 - a) ** 0.5).to(original_samples.dtype)
