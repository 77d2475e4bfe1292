import torch.nn as nn


class ComposerModel(nn.Module):
    pass
