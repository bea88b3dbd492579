"""MI355X-native VideoUpscalePipeline (drop-in for the reference's
`models_video/pipeline_upscale_a_video.py`: class :61, _encode_prompt :177-321, check_inputs
:356-418, prepare_latents_3d :421-432, __call__ :436-716).

Call contract kept: `pipeline(prompt, image=(1,3,T,h,w) in [-1,1], flows_bi=[fwd, bwd] | None,
generator, num_inference_steps, guidance_scale, noise_level, negative_prompt, propagation_steps)`
-> `.images` (1,3,T,4h,4w) fp32 in [-1,1]; `return_dict=False` -> `(images, latents)`; the same
ValueError / TypeError conditions; RNG draw order (LR noise, then latents) preserved.

What differs inside (results unchanged up to fp rounding):
  * the loop is sync-free: timesteps and scheduler coefficients stay on the host, CFG + DDIM
    step_v0 is one fused kernel, step_vt another (reference: ~10 ATen launches + 2 host syncs +
    2 `empty_cache()` per step, :612-659);
  * sliding temporal windows (:619-635): when the reference's range(0,T,6) re-anchors the tail
    window onto the previous one (T=32: start 30 -> [24,32) again) the UNet is NOT evaluated a
    second time — its output is reused and the running 0.5/0.5 blend is applied again, which is
    what the reference computes (the second blend is not an identity on the 2 overlap frames);
  * the VAE decodes in fp16-in / fp32-accumulate MFMA kernels instead of fp32 ATen (:668-702) and
    clamps in the layout-conversion kernel.
Multi-GPU: `UAV_RANKS` style clip/chunk sharding lives in bench.py / uav.dist — a pipeline object
always drives ONE GPU, like the reference.
"""
import os
from dataclasses import dataclass
from typing import List, Optional

import torch

from uav import dist as D
from uav import engine as E
from uav import ops
from uav import streams

from ._compat import BaseOutput, ConfigMixin
from .scheduling_ddim import DDIMScheduler, DDPMScheduler


@dataclass
class StableDiffusionPipelineOutput(BaseOutput):
    images: torch.Tensor = None
    nsfw_content_detected: Optional[List[bool]] = None


def randn_tensor(shape, generator=None, device=None, dtype=None):
    """diffusers.utils.randn_tensor: a CPU generator draws on the CPU, then the tensor moves."""
    rand_device = device
    if generator is not None and generator.device.type == "cpu":
        rand_device = "cpu"
    return torch.randn(shape, generator=generator, device=rand_device, dtype=dtype).to(device)


def window_schedule(t_total, short_seq=8, overlap_seq=2):
    """(start, end) windows in the reference's visiting order (:601-629), duplicates included."""
    if t_total <= short_seq:
        return [(0, t_total)]
    out = []
    for start in range(0, t_total, short_seq - overlap_seq):
        end = min(t_total, start + short_seq)
        if end - start < short_seq:
            start = end - short_seq
        out.append((start, end))
    return out


class VideoUpscalePipeline(ConfigMixin):
    config_name = "model_index.json"
    native_text_encoder = True         # from_pretrained converts the CLIP text model to uav.clip_text.UavCLIPTextModel

    def __init__(self, text_encoder=None, tokenizer=None, low_res_scheduler=None, scheduler: DDIMScheduler = None,
                 vae=None, unet=None, propagator=None, max_noise_level: int = 350):
        super().__init__()
        self.register_modules(vae=vae, text_encoder=text_encoder, tokenizer=tokenizer, unet=unet,
                              low_res_scheduler=low_res_scheduler, scheduler=scheduler, propagator=propagator)
        self.register_to_config(max_noise_level=max_noise_level)
        self._device = torch.device("cpu")
        # under classifier-free guidance both UNet batch entries see identical latents / low_res / timestep, so the
        # text-independent head of the UNet is computed once (UNetVideoModel.forward cfg_shared_input; bit-identical)
        self.cfg_shared_input = True
        # one LONG clip over several GPUs (BASELINE config 4): deal the temporal windows of each DDIM step and the decode
        # chunks over the ranks of the default process group; results are bit-identical to the single-GPU call
        self.shard_windows = False
        # ... at (window x guidance branch) granularity: every UNet evaluation of a DDIM step is split into its
        # unconditional and its text-conditioned half (two batch-1 calls, the CFG-shared head is given up: +2.7 % FLOP), so
        # T = 32 gives 10 units per step instead of 5 (speed-up bound 3.3x instead of 2.5x on 4 ranks, 5x instead of 2.5x on
        # 8) and even one 8-frame clip splits over 2 GPUs.  Only read when `shard_windows` is set; the result is
        # bit-identical for every world size (the unit decomposition does not depend on it).
        self.shard_cfg = False
        # ONE clip on ONE GPU, its independent units on `overlap_streams` HIP streams (uav/streams.py), issued by this one host
        # thread: the temporal windows of a DDIM step and the 3-frame decode chunks of a clip LONGER than 8 frames (two or more
        # unique windows) — same bits as the serial order, +2.5 % on a 32-frame clip (run 24).  An 8-frame clip has
        # one window and stays serial: its decode chunks on two streams measured neutral, and `overlap_split_cfg` (its two
        # guidance branches as separate batch-1 units, the decomposition of `shard_cfg`) measured 2.4 % SLOWER — half-size
        # launches, CFG-shared head given up — so that one is opt-in.  0 / 1 = always serial.
        self.overlap_streams = int(os.environ.get("UAV_OVERLAP_STREAMS", "2"))
        self.overlap_split_cfg = os.environ.get("UAV_OVERLAP_SPLIT_CFG", "0") == "1"
        self.overlap_min_free_fraction = 0.35        # serial fallback below this share of free device memory
        self.last_overlap_mode = "serial"            # what the last call's window / chunk loop ran as (bench.py reports it)
        self.latents_trace = None          # test hook: set to a list to collect the latents after every DDIM step
        self.cache_prompt_embeds = True
        self._prompt_cache = {}

    def register_modules(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    @classmethod
    def from_pretrained(cls, path, torch_dtype=None, **kwargs):
        """Loads low_res_scheduler / text_encoder / tokenizer from a diffusers-layout directory
        (inference_upscale_a_video.py:101); vae / unet / scheduler are assigned afterwards by the
        caller exactly like the reference CLI does (:104-121)."""
        import os
        from transformers import CLIPTextModel, CLIPTokenizer
        tok = CLIPTokenizer.from_pretrained(os.path.join(path, "tokenizer"))
        te = CLIPTextModel.from_pretrained(os.path.join(path, "text_encoder"), torch_dtype=torch_dtype)
        if cls.native_text_encoder:
            # same weights, same arithmetic, on the HIP kernels (uav/clip_text.py) instead of eager transformers ops
            from uav.clip_text import UavCLIPTextModel
            te = UavCLIPTextModel.from_hf(te)
        lrs_dir = os.path.join(path, "low_res_scheduler")
        lrs = DDPMScheduler.from_config(lrs_dir) if os.path.isdir(lrs_dir) else DDPMScheduler()
        return cls(text_encoder=te, tokenizer=tok, low_res_scheduler=lrs)

    def to(self, device):
        for name in ("vae", "text_encoder", "unet", "propagator"):
            m = getattr(self, name, None)
            if m is not None and hasattr(m, "to"):
                setattr(self, name, m.to(device))
        self._device = torch.device(device)
        return self

    @property
    def device(self):
        return self._device

    @property
    def _execution_device(self):
        return self._device

    # ------------------------------------------------------------------------------------------
    def _encode_prompt(self, prompt, device, num_images_per_prompt, do_classifier_free_guidance, negative_prompt=None,
                       prompt_embeds=None, negative_prompt_embeds=None):
        """Same token / embedding plumbing as the reference (:177-321); the text encoder itself is an
        external module (CLIP in the release, a stand-in in benchmarks) and not part of this engine."""
        if prompt is not None and isinstance(prompt, str):
            batch_size = 1
        elif prompt is not None and isinstance(prompt, list):
            batch_size = len(prompt)
        else:
            batch_size = prompt_embeds.shape[0]

        def encode(text, max_length):
            inputs = self.tokenizer(text, padding="max_length", max_length=max_length, truncation=True, return_tensors="pt")
            mask = None
            cfg = getattr(self.text_encoder, "config", None)
            if cfg is not None and getattr(cfg, "use_attention_mask", False):
                mask = inputs.attention_mask.to(device)
            return self.text_encoder(inputs.input_ids.to(device), attention_mask=mask)[0]

        if prompt_embeds is None:
            prompt_embeds = encode(prompt, self.tokenizer.model_max_length)
        prompt_embeds = prompt_embeds.to(dtype=self.text_encoder.dtype, device=device)
        bs_embed, seq_len, _ = prompt_embeds.shape
        prompt_embeds = prompt_embeds.repeat(1, num_images_per_prompt, 1).view(bs_embed * num_images_per_prompt, seq_len, -1)
        if do_classifier_free_guidance and negative_prompt_embeds is None:
            if negative_prompt is None:
                uncond_tokens = [""] * batch_size
            elif type(prompt) is not type(negative_prompt):
                raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got {type(negative_prompt)} !="
                                f" {type(prompt)}.")
            elif isinstance(negative_prompt, str):
                uncond_tokens = [negative_prompt]
            elif batch_size != len(negative_prompt):
                raise ValueError(f"`negative_prompt`: {negative_prompt} has batch size {len(negative_prompt)}, but `prompt`:"
                                 f" {prompt} has batch size {batch_size}.")
            else:
                uncond_tokens = negative_prompt
            negative_prompt_embeds = encode(uncond_tokens, prompt_embeds.shape[1])
        if do_classifier_free_guidance:
            seq_len = negative_prompt_embeds.shape[1]
            negative_prompt_embeds = negative_prompt_embeds.to(dtype=self.text_encoder.dtype, device=device)
            negative_prompt_embeds = negative_prompt_embeds.repeat(1, num_images_per_prompt, 1)
            negative_prompt_embeds = negative_prompt_embeds.view(batch_size * num_images_per_prompt, seq_len, -1)
            prompt_embeds = torch.cat([negative_prompt_embeds, prompt_embeds])
        return prompt_embeds

    def _cached_prompt_embeds(self, prompt, device, num_images_per_prompt, do_cfg, negative_prompt=None,
                              prompt_embeds=None, negative_prompt_embeds=None):
        """`_encode_prompt` with a small cache for string prompts: the CLI calls the pipeline once per
        tile with the same prompt pair, and the reference runs the CLIP text encoder twice per call.  The key holds the
        strings, the encoder / tokenizer objects and the current stream (an entry made on one stream is not handed to
        another stream); `cache_prompt_embeds = False` restores one encode per call.  The returned tensor is the same
        object on a hit, so the UNet's per-prompt text K/V caches hit as well."""
        dev = torch.device(device)
        cacheable = (self.cache_prompt_embeds and isinstance(prompt, str) and prompt_embeds is None
                     and negative_prompt_embeds is None and (negative_prompt is None or isinstance(negative_prompt, str)))
        if not cacheable:
            return self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt,
                                       prompt_embeds=prompt_embeds, negative_prompt_embeds=negative_prompt_embeds)
        stream = torch.cuda.current_stream(dev).cuda_stream if dev.type == "cuda" else 0
        key = (prompt, negative_prompt, bool(do_cfg), int(num_images_per_prompt), str(dev), stream)
        hit = self._prompt_cache.get(key)
        # an entry is valid only for the encoder / tokenizer OBJECTS that made it (held weakly: `id()` alone can be
        # recycled after garbage collection) and for the encoder's current parameter values
        enc_stamp = E._stamp(list(self.text_encoder.parameters())[:4]) if hasattr(self.text_encoder, "parameters") else ()
        if hit is not None and (hit[1]() is not self.text_encoder or hit[2]() is not self.tokenizer or hit[3] != enc_stamp):
            hit = None
        if hit is None:
            import weakref
            emb = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt)
            if len(self._prompt_cache) >= 8:
                self._prompt_cache.clear()
            hit = (emb, weakref.ref(self.text_encoder), weakref.ref(self.tokenizer), enc_stamp)
            self._prompt_cache[key] = hit
        return hit[0]

    def _stream_set(self, device, n_windows, do_cfg):
        """The side streams this call issues its independent units on (uav/streams.py), or None = everything on the caller's
        stream: GPU only, not together with the multi-GPU sharding, and only when there is something to overlap — two or more
        unique temporal windows, or the guidance branches when `overlap_split_cfg` asks for them.  Every side stream owns a
        caching-allocator pool (blocks freed there are not reusable by the caller's stream) and two units are live at once, so
        the overlap is skipped when less than `overlap_min_free_fraction` of the device memory is free."""
        self.last_overlap_mode = "serial"             # every early return below leaves the truth behind, not the previous call's value
        if self.overlap_streams <= 1 or self.shard_windows or torch.device(device).type != "cuda":
            return None
        try:
            free, total = torch.cuda.mem_get_info(device)
            # blocks the caching allocator holds but does not use (incl. the side-stream pools of an earlier call) are free for
            # this purpose: counting them as used made the mode depend on call order
            free += torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        except (RuntimeError, AssertionError):       # no HIP runtime (host-logic tests with stand-in streams): nothing to gate on
            free, total = 1, 1
        if free < self.overlap_min_free_fraction * total:
            self.last_overlap_mode = "serial (memory gate: %.0f %% of the device free)" % (100.0 * free / max(total, 1))
            return None
        if n_windows > 1 or (self.overlap_split_cfg and do_cfg):
            self.last_overlap_mode = "%d streams" % self.overlap_streams
            return streams.stream_set(device, self.overlap_streams)
        return None

    def check_inputs(self, prompt, image, noise_level, negative_prompt=None, prompt_embeds=None,
                     negative_prompt_embeds=None):
        if prompt is not None and prompt_embeds is not None:
            raise ValueError(f"Cannot forward both `prompt`: {prompt} and `prompt_embeds`: {prompt_embeds}. Please make sure to"
                             " only forward one of the two.")
        elif prompt is None and prompt_embeds is None:
            raise ValueError("Provide either `prompt` or `prompt_embeds`. Cannot leave both `prompt` and `prompt_embeds` undefined.")
        elif prompt is not None and (not isinstance(prompt, str) and not isinstance(prompt, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if negative_prompt is not None and negative_prompt_embeds is not None:
            raise ValueError("Cannot forward both `negative_prompt` and `negative_prompt_embeds`.")
        if prompt_embeds is not None and negative_prompt_embeds is not None and prompt_embeds.shape != negative_prompt_embeds.shape:
            raise ValueError("`prompt_embeds` and `negative_prompt_embeds` must have the same shape when passed directly")
        if not isinstance(image, (torch.Tensor, list)):
            raise ValueError(f"`image` has to be of type `torch.Tensor` or `list` but is {type(image)}")
        if isinstance(image, (list, torch.Tensor)):
            # the reference evaluates len(prompt) here and crashes for prompt=None (:401-405); kept: a
            # missing prompt with prompt_embeds is reported as the same TypeError
            batch_size = 1 if isinstance(prompt, str) else len(prompt)
            image_batch_size = len(image) if isinstance(image, list) else image.shape[0]
            if batch_size != image_batch_size:
                raise ValueError(f"`prompt` has batch size {batch_size} and `image` has batch size {image_batch_size}."
                                 " Please make sure that passed `prompt` matches the batch size of `image`.")
        if noise_level > self.config.max_noise_level:
            raise ValueError(f"`noise_level` has to be <= {self.config.max_noise_level} but is {noise_level}")

    def prepare_latents_3d(self, batch_size, num_channels_latents, seq_len, height, width, dtype, device, generator,
                           latents=None):
        shape = (batch_size, num_channels_latents, seq_len, height, width)
        if latents is None:
            latents = randn_tensor(shape, generator=generator, device=device, dtype=dtype)
        else:
            if latents.shape != shape:
                raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
            latents = latents.to(device)
        return latents * self.scheduler.init_noise_sigma

    def decode_latents_vsr(self, latents, img, w_lr):
        """latents / scaling_factor -> vae.decode -> clamp(-1,1) fp32 (reference :350-354); the
        division is folded into the layout-conversion kernel, the clamp into the output one."""
        sf = self.vae.config.scaling_factor
        if hasattr(self.vae, "decode_rows"):
            from uav import engine as E
            y, g2 = self.vae.decode_rows(latents, img, float(w_lr), latent_scale=1.0 / sf)
            return E.from_rows(y, g2, self.vae.config.out_channels, out_dtype=torch.float32, clamp=(-1.0, 1.0))
        return self.vae.decode(latents / sf, img, w_lr).sample.clamp(-1, 1).float()

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, prompt=None, image=None, flows_bi: Optional[list] = None, num_inference_steps: int = 75,
                 guidance_scale: float = 9.0, noise_level: int = 20, denoise_level: Optional[int] = None,
                 negative_prompt=None, num_images_per_prompt: Optional[int] = 1, eta: float = 0.0, generator=None,
                 latents=None, prompt_embeds=None, negative_prompt_embeds=None, propagation_steps: list = [],
                 w_lr: float = 1, return_dict: bool = True, progress: bool = False):
        self.check_inputs(prompt, image, noise_level, negative_prompt, prompt_embeds, negative_prompt_embeds)
        if image is None:
            raise ValueError("`image` input cannot be undefined.")
        if eta != 0.0:
            raise NotImplementedError("eta != 0 is never used by the CLI")
        batch_size = 1 if isinstance(prompt, str) else len(prompt) if prompt is not None else prompt_embeds.shape[0]
        if batch_size != 1 or num_images_per_prompt != 1:
            raise NotImplementedError("the CLI upscales one clip per call")
        device = self._execution_device
        if device.type == "cuda" and device.index is not None and device.index != torch.cuda.current_device():
            # launches go to the current device's stream (uav/ops.py:_stream): make the pipeline's GPU current
            with torch.cuda.device(device):
                return self.__call__(prompt, image, flows_bi, num_inference_steps, guidance_scale, noise_level, denoise_level,
                                     negative_prompt, num_images_per_prompt, eta, generator, latents, prompt_embeds,
                                     negative_prompt_embeds, propagation_steps, w_lr, return_dict, progress)
        do_cfg = guidance_scale > 1.0

        prompt_embeds = self._cached_prompt_embeds(prompt, device, num_images_per_prompt, do_cfg, negative_prompt,
                                                   prompt_embeds, negative_prompt_embeds)
        draw_dtype = prompt_embeds.dtype          # the reference draws both noises in prompt_embeds.dtype (:547,:573)
        prompt_embeds = prompt_embeds.to(torch.float16).contiguous()
        # Latent precision between DDIM steps.  fp16 (default, the reference's `.half()` arithmetic: latents, UNet output,
        # CFG and both scheduler halves are fp16 tensors there too), or fp32 when the UNet runs on an fp32 residual
        # stream (UNetVideoModel.stream_dtype): the UNet's fp32 output rows, the guided output (CFG multiplies the
        # rounding error of its two inputs by ~2*guidance), x0 and the latents then stay fp32 from step to step.
        lat_dtype = torch.float32 if getattr(self.unet, "stream_f32", lambda: False)() else torch.float16
        if self.__dict__.get("_logged_mode") != lat_dtype:          # once per mode: users of the released CLI expect `.half()` arithmetic
            self.__dict__["_logged_mode"] = lat_dtype
            import logging
            logging.getLogger("uav").info(
                "VideoUpscalePipeline: UNet residual stream / latents in %s (%s); set unet.stream_dtype = torch.float16 or "
                "UAV_UNET_STREAM=f16 before building the UNet for the reference's all-fp16 `.half()` arithmetic",
                "fp32" if lat_dtype == torch.float32 else "fp16",
                "default: fp16 MFMA operands on fp32 rows, inside 1e-3 of the reference's fp32 run" if lat_dtype == torch.float32
                else "the reference CLI's precision mix")

        # 4/5. LR frames: fp32 copy for the VAE conditioning, fp16 + noise for the UNet (:542-551)
        image_dec = image.clone().to(dtype=torch.float32, device=device)
        image = image.to(dtype=lat_dtype, device=device)
        noise = randn_tensor(image.shape, generator=generator, device=device, dtype=draw_dtype).to(lat_dtype)
        image = self.low_res_scheduler.add_noise(image, noise, torch.tensor([noise_level]))
        level = torch.tensor([noise_level if denoise_level is None else denoise_level], dtype=torch.long)
        # reference quirk kept: the single-window branch (T <= 8) conditions the UNet on `noise_level` even when a
        # `denoise_level` was passed (:638), only the sliding-window branch uses `denoise_level` (:628)
        level_single = torch.tensor([noise_level], dtype=torch.long)
        if do_cfg:
            image = torch.cat([image] * 2)

        self.scheduler.set_timesteps(num_inference_steps, device=device)
        timesteps = [int(t) for t in self.scheduler.timesteps]
        t_total, height, width = image.shape[2:]
        num_channels_latents = self.vae.config.latent_channels
        latents = self.prepare_latents_3d(1, num_channels_latents, t_total, height, width, draw_dtype, device,
                                          generator, latents).to(lat_dtype).contiguous()
        if num_channels_latents + image.shape[1] != self.unet.config.in_channels:
            raise ValueError(f"Incorrect configuration settings! The config of `pipeline.unet`: {self.unet.config} expects"
                             f" {self.unet.config.in_channels} but received `num_channels_latents`: {num_channels_latents} +"
                             f" `num_channels_image`: {image.shape[1]}")

        wins = window_schedule(t_total)
        overlap = self._stream_set(device, len(set(wins)), do_cfg)
        # per-branch text rows as stable objects: the UNet's text K/V caches are keyed on tensor identity
        # (and kept across calls for the same prompt tensor, so those caches also hit on the next clip / tile)
        pe_branch = None
        if do_cfg:
            hit = self.__dict__.get("_pe_branch")
            if hit is None or hit[0] is not prompt_embeds or hit[1] != prompt_embeds._version:
                hit = (prompt_embeds, prompt_embeds._version,
                       E.publish([prompt_embeds[b:b + 1].contiguous() for b in range(prompt_embeds.shape[0])]))
                self.__dict__["_pe_branch"] = hit
            pe_branch = E.acquire(hit[2])
        if flows_bi is not None and len(propagation_steps) > 0:
            # reference :651: `flows_bi[k].to(latents)` — the flows follow the latent dtype
            flows_f = flows_bi[0].to(device=device, dtype=lat_dtype)
            flows_b = flows_bi[1].to(device=device, dtype=lat_dtype)

        for i, t in enumerate(timesteps):
            lin = torch.cat([latents] * 2) if do_cfg else latents
            uniq = [w for k, w in enumerate(wins) if w not in wins[:k]]
            split = do_cfg and ((self.shard_windows and self.shard_cfg) or (overlap is not None and self.overlap_split_cfg and len(uniq) < len(overlap.streams)))

            def eval_unit(u):
                """One UNet evaluation: window (s, e) with both guidance branches (b is None) or one of them."""
                (s_, e_), b = u
                bs = slice(None) if b is None else slice(b, b + 1)
                return self.unet(lin[bs, :, s_:e_].contiguous(), t, image[bs, :, s_:e_].contiguous(),
                                 encoder_hidden_states=prompt_embeds if b is None else pe_branch[b],
                                 class_labels=level if len(wins) > 1 else level_single,
                                 cfg_shared_input=do_cfg and self.cfg_shared_input and b is None).sample.contiguous()
            # the windows of one step are independent UNet evaluations: with `shard_windows` and an initialised process
            # group the units are dealt over the ranks and all-gathered (uav/dist.py:sharded_map); the blend below is
            # replayed identically on every rank.  A duplicate tail window is evaluated once.
            units = [(w, b) for w in uniq for b in ((0, 1) if split else (None,))]
            like = ((1 if split else lin.shape[0], latents.shape[1], uniq[0][1] - uniq[0][0]) + tuple(lin.shape[3:]), lat_dtype, device)
            if self.shard_windows:
                res = D.sharded_map(units, eval_unit, like=like)
            else:
                res = overlap.map(units, eval_unit) if overlap is not None else [eval_unit(u) for u in units]
            outs = {w: (torch.cat([res[2 * k], res[2 * k + 1]]) if split else res[k]) for k, w in enumerate(uniq)}
            if len(wins) > 1:
                eps = None
                written = [False] * t_total
                for (s, e) in wins:
                    o = outs[(s, e)]
                    if eps is None:
                        eps = torch.empty((o.shape[0], o.shape[1], t_total) + tuple(o.shape[3:]), dtype=o.dtype, device=device)
                    for k, idx in enumerate(range(s, e)):
                        if not written[idx]:
                            eps[:, :, idx] = o[:, :, k]
                            written[idx] = True
                        else:                                                     # running 0.5/0.5 blend (:634)
                            eps[:, :, idx] = ops.axpby(eps[:, :, idx].contiguous(), o[:, :, k].contiguous(), 0.5, 0.5)
            else:
                eps = outs[wins[0]]
            eps = eps.contiguous()
            if do_cfg:
                guided, x0 = self.scheduler.cfg_step_v0(eps[0:1], eps[1:2], guidance_scale, t, latents)
            else:
                guided, x0 = self.scheduler.cfg_step_v0(eps, None, 1.0, t, latents)
            if flows_bi is not None and i in propagation_steps:
                # fp16 latents: the kernel replays the reference's fp16 warp (bit-exact against it on half tensors);
                # fp32 latents: fp32 values on fp32 grids, held to the reference's fp32 run (no rounding of the trajectory)
                x0 = self.propagator(x0, flows_f, flows_b, interpolation="nearest", mode="fuse",
                                     fuse_scale=0.5, alpha1=0.001, alpha2=0.05).to(lat_dtype).contiguous()
            latents = self.scheduler.step_vt(x0, guided, t, latents).prev_sample
            if self.latents_trace is not None:        # test hook: per-step latents (error-vs-step curves)
                self.latents_trace.append(latents.clone())

        latents_out = latents.float()
        # 11. decode in chunks of 3 frames on the GLOBAL frame index (:685-702)
        short_seq = 3
        if t_total > short_seq:
            starts = list(range(0, t_total, short_seq))

            def decode_chunk(s):
                e = min(t_total, s + short_seq)
                y = self.decode_latents_vsr(latents[:, :, s:e].contiguous(), image_dec[:, :, s:e].contiguous(), w_lr)
                if self.shard_windows and e - s < short_seq:      # ragged last chunk: pad to a common shape for the gather
                    y = torch.cat([y, y.new_zeros(y.shape[:2] + (short_seq - (e - s),) + y.shape[3:])], dim=2)
                return y.contiguous()
            like = ((1, image_dec.shape[1], short_seq, 4 * height, 4 * width), torch.float32, device)
            if self.shard_windows:
                chunks = D.sharded_map(starts, decode_chunk, like=like)
            else:
                chunks = overlap.map(starts, decode_chunk) if overlap is not None else [decode_chunk(s) for s in starts]
            out = torch.cat(chunks, dim=2)[:, :, :t_total]
        else:
            out = self.decode_latents_vsr(latents, image_dec, w_lr)
        if not return_dict:
            return (out, latents_out)
        return StableDiffusionPipelineOutput(images=out, nsfw_content_detected=None)
