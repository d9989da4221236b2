"""Minimal stand-in for torch_geometric.data.Data (not installed here): an attribute bag whose tensors move together."""


class Data(object):
    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def to(self, device):
        import torch
        for k, v in list(self.__dict__.items()):
            if isinstance(v, torch.Tensor):
                setattr(self, k, v.to(device))
        return self

    @property
    def num_nodes(self):
        return int(self.x.shape[0]) if getattr(self, "x", None) is not None else int(len(self.y))
