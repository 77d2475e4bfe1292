from micro_diffusion_amd.model import DATA_TYPES, DistLoss, text_encoder_embedding_format  # noqa: F401
