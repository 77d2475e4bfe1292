"""Stand-in for timm's PatchEmbed as used at reference dit.py:312-314: Conv2d(k=s=patch) -> flatten(2) ->
transpose(1, 2); no norm.  (timm is unpinned in the reference's setup.py:13 — semantics per SURVEY.md C.9.)"""
import torch.nn as nn


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x):
        assert x.shape[-2] == self.img_size[0] and x.shape[-1] == self.img_size[1]
        return self.norm(self.proj(x).flatten(2).transpose(1, 2))
