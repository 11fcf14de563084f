"""TEST INFRASTRUCTURE ONLY - see the package docstring.  Constructible, not callable."""
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, *args, **kwargs):
        super().__init__()
        self.dim = dim

    def forward(self, *args, **kwargs):
        raise NotImplementedError("x_transformers stand-in: the conformer condition encoder is not on the generate path")

    forward_from_seq_len = forward


def apply_rotary_pos_emb(*args, **kwargs):
    raise NotImplementedError("x_transformers stand-in: the conformer condition encoder is not on the generate path")
