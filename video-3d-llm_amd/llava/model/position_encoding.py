"""llava.model.position_encoding.PositionEmbeddingSine3D on HIP (reference: position_encoding.py:5-49)."""
import torch
import torch.nn as nn

from v3d import ops


class PositionEmbeddingSine3D(nn.Module):
    """Same constructor and call contract as the reference module; forward launches v3d_sin3d_pe.
    n_points > 1 (sample9 / minmax ablations) is out of scope: the shipped config is n_points = 1."""

    def __init__(self, embedding_size, temperature=10000, n_points=1):
        super().__init__()
        if n_points != 1:
            raise NotImplementedError("only n_points == 1 (avg-discrete-sin3d) is on the accelerated path")
        self.embedding_size = embedding_size
        self.temperature = temperature
        self.n_points = n_points
        # the reference recomputes this on every call (position_encoding.py:24-25); same torch expression, once
        self._dim_t = ops.reference_dim_t(embedding_size // 3, temperature)
        self._dim_t_dev = {}

    def forward(self, x):
        key = x.device
        if key not in self._dim_t_dev:
            self._dim_t_dev[key] = self._dim_t.to(x.device)
        return ops.sin3d_pe(x, self.embedding_size, dim_t=self._dim_t_dev[key])
