import torch


class Metric(torch.nn.Module):
    def __init__(self, **kwargs):
        super().__init__()

    def add_state(self, name, default, dist_reduce_fx=None):
        setattr(self, name, default)
