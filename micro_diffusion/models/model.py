from micro_diffusion_amd.model import *  # noqa: F401,F403
from micro_diffusion_amd.model import LatentDiffusion, create_latent_diffusion, DistLoss, text_encoder_embedding_format  # noqa: F401
