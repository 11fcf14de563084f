from torch import nn


class ResidualVectorQuantize(nn.Module):
    def __init__(self, *a, **k):
        super().__init__()
