import torch


class Raft_Large_Weights:
    DEFAULT = None

    @staticmethod
    def transforms():
        return lambda a, b: (a, b)


class _ZeroFlow(torch.nn.Module):
    def forward(self, a, b, *args, **kwargs):
        return [torch.zeros(a.shape[0], 2, a.shape[-2], a.shape[-1])]


def raft_large(*args, **kwargs):
    return _ZeroFlow()
