"""Model hyper-parameters of the released Upscale-A-Video checkpoints, in the diffusers JSON schema the
reference consumes with `X.from_config(...)` (values: /root/reference/configs/unet_video_config.json,
vae_3d_config.json, vae_video_config.json; scheduler values: upstream SD-x4-upscaler
scheduler_config.json, which is absent from the reference tree — SURVEY.md §8c)."""

UNET_VIDEO = {
    "_class_name": "UNetVideoModel", "act_fn": "silu", "attention_head_dim": 8,
    "block_out_channels": [256, 512, 512, 1024], "center_input_sample": False, "cross_attention_dim": 1024,
    "down_block_types": ["DownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D", "CrossAttnDownBlock3D"],
    "downsample_padding": 1, "dual_cross_attention": False, "flip_sin_to_cos": True, "freq_shift": 0, "in_channels": 7,
    "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05, "norm_num_groups": 32,
    "num_class_embeds": 1000, "only_cross_attention": [True, True, True, False], "out_channels": 4, "sample_size": 128,
    "up_block_types": ["CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "CrossAttnUpBlock3D", "UpBlock3D"],
    "use_linear_projection": True, "down_temporal_idx": [0, 1, 2, 3], "mid_temporal": True,
    "up_temporal_idx": [0, 1, 2, 3], "temporal_module_config": {"attention_block_types": ["", ""]},
}

VAE_3D = {
    "_class_name": "AutoencoderKL3D", "act_fn": "silu", "block_out_channels": [128, 256, 512],
    "down_block_types": ["DownEncoderBlock3D"] * 3, "in_channels": 3, "latent_channels": 4, "layers_per_block": 2,
    "norm_num_groups": 32, "out_channels": 3, "sample_size": 256, "up_block_types": ["UpDecoderBlock3D"] * 3,
    "scaling_factor": 0.08333,
}

VAE_VIDEO = dict(VAE_3D, _class_name="AutoencoderKLVideo", up_block_types=["UpDecoderBlock3D_plus"] * 3,
                 condition_img=True, condition_channels=128, use_temporal_block=True)

DDIM = dict(num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
            clip_sample=False, set_alpha_to_one=False, steps_offset=1, prediction_type="v_prediction")
LOW_RES_DDPM = dict(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="scaled_linear")

# analytic FLOP model (validated against torch.utils.flop_counter): TFLOP
TFLOP_UNET_320 = 176.99          # one UNetVideoModel.forward, CFG batch 2, T=8, 320x320
TFLOP_VAE3D_320 = 299.05         # vae_3d decode of 8 frames at 320x320 -> 1280x1280
TFLOP_CLIP_C2 = 30 * TFLOP_UNET_320 + TFLOP_VAE3D_320      # 5608.7 per 8-frame clip (config 2)
