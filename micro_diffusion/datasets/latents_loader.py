from micro_diffusion_amd.data import SyntheticLatents, build_streaming_latents_dataloader  # noqa: F401
