from micro_diffusion_amd.dit import DiT, MicroDiT_Tiny, MicroDiT_Tiny_2, MicroDiT_XL_2  # noqa: F401
