from micro_diffusion_amd.data import (LatentsLoader, StreamingLatentsDataset, SyntheticLatents,  # noqa: F401
                                      build_streaming_latents_dataloader)
