"""Host-side mirror of the reference's field classes (models/nerf.py).

`render_rays` never calls these modules' forward(): it reads their parameters
(by the reference's state_dict names, so reference checkpoints load unchanged)
and attributes (`typ`, `encode_appearance`, `encode_transient`, `beta_min`,
`in_channels_*`, reference models/nerf.py:104-119) and evaluates the field inside
the fused HIP kernel.  forward() exists so that code which calls a module
directly (reference models/nerf.py:153-212, 19-32) keeps working: it runs the same
HIP kernels through the C ABI (`nfl_field_forward`, `nfl_posenc`), is inference-only
and raises on CPU tensors.
"""
import math

import torch
from torch import nn


class PosEmbedding(nn.Module):
    """x -> [x, sin(2^0 x), cos(2^0 x), ..., sin(2^(N-1) x), cos(2^(N-1) x)]
    (reference models/nerf.py:6-32; log-spaced frequencies only)."""

    def __init__(self, max_logscale, N_freqs, logscale=True):
        super().__init__()
        if not logscale:
            raise NotImplementedError("linear frequency spacing is unused by the reference's configs")
        if max_logscale != N_freqs - 1:
            raise ValueError("frequencies must be 2^0 .. 2^(N_freqs-1)")
        self.N_freqs = N_freqs
        self.freqs = 2 ** torch.linspace(0, max_logscale, N_freqs)   # attribute, not a buffer (as in the reference)

    def forward(self, x):
        from . import rendering                 # HIP kernel nfl_posenc; device tensors only
        return rendering.posenc(x, self.N_freqs)


class BarfPosEmbedding(PosEmbedding):
    """Coarse-to-fine (BARF) variant used with learnable poses (reference models/nerf.py:35-77,
    created by train.py:42-44 as BarfPosEmbedding(N-1, N, 4, 8)).  The per-frequency weight is a
    literal restatement of the reference, including its quirks: alpha = N_freqs / epoch inside
    (epoch_start, epoch_end], N_freqs after, 0 before, and alpha is compared with the frequency
    VALUE 2^k rather than with its index."""

    def __init__(self, max_logscale, N_freqs, epoch_start, epoch_end, logscale=True):
        super().__init__(max_logscale, N_freqs, logscale)
        self.epoch_start, self.epoch_end = epoch_start, epoch_end

    def barf_weight(self, freq, epoch):
        freq = float(freq)
        if self.epoch_start < epoch <= self.epoch_end:
            alpha = self.N_freqs / epoch
        elif epoch > self.epoch_end:
            alpha = float(self.N_freqs)
        else:
            alpha = 0.0
        if alpha < freq:
            return 0.0
        if alpha - freq < 1:
            return float((1 - torch.cos(torch.tensor((alpha - freq) * math.pi, dtype=torch.float32))) / 2)
        return 1.0

    def weights(self, epoch):
        return torch.tensor([self.barf_weight(f, epoch) for f in self.freqs], dtype=torch.float32)

    def forward(self, x, epoch):
        from . import rendering
        return rendering.posenc(x, self.N_freqs, self.weights(epoch))


class NeRF(nn.Module):
    """Parameter container with the reference's layer names and shapes
    (models/nerf.py:81-151): 8x256 trunk with the encoded position re-read at
    layer 5, sigma / rgb heads, optional appearance input and transient head."""

    def __init__(self, typ, D=8, W=256, skips=(4,), in_channels_xyz=63, in_channels_dir=27,
                 encode_appearance=False, in_channels_a=48, encode_transient=False, in_channels_t=16,
                 beta_min=0.03, refine_pose=False):
        super().__init__()
        if D != 8 or W != 256 or tuple(skips) != (4,):
            raise NotImplementedError("the HIP renderer is built for D=8, W=256, skips=[4]")
        self.typ, self.D, self.W, self.skips = typ, D, W, list(skips)
        self.in_channels_xyz, self.in_channels_dir = in_channels_xyz, in_channels_dir
        self.refine_pose = bool(refine_pose)      # render_rays then expects BarfPosEmbedding embeddings + current_epoch
        self.encode_appearance = False if typ == "coarse" else encode_appearance
        self.in_channels_a = in_channels_a if encode_appearance else 0
        self.encode_transient = False if typ == "coarse" else encode_transient
        self.in_channels_t = in_channels_t
        self.beta_min = beta_min

        for i in range(D):
            fan_in = in_channels_xyz if i == 0 else (W + in_channels_xyz if i in self.skips else W)
            setattr(self, f"xyz_encoding_{i + 1}", nn.Sequential(nn.Linear(fan_in, W), nn.ReLU(True)))
        self.xyz_encoding_final = nn.Linear(W, W)
        self.dir_encoding = nn.Sequential(nn.Linear(W + in_channels_dir + self.in_channels_a, W // 2), nn.ReLU(True))
        self.static_sigma = nn.Sequential(nn.Linear(W, 1), nn.Softplus())
        self.static_rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
        if self.encode_transient:
            self.transient_encoding = nn.Sequential(
                nn.Linear(W + in_channels_t, W // 2), nn.ReLU(True),
                nn.Linear(W // 2, W // 2), nn.ReLU(True),
                nn.Linear(W // 2, W // 2), nn.ReLU(True),
                nn.Linear(W // 2, W // 2), nn.ReLU(True))
            self.transient_sigma = nn.Sequential(nn.Linear(W // 2, 1), nn.Softplus())
            self.transient_rgb = nn.Sequential(nn.Linear(W // 2, 3), nn.Sigmoid())
            self.transient_beta = nn.Sequential(nn.Linear(W // 2, 1), nn.Softplus())

    def forward(self, x, sigma_only=False, output_transient=True):
        """(B, C) encoded inputs -> sigma | [rgb, sigma] | [rgb, sigma, rgb_t, sigma_t, beta]
        (reference models/nerf.py:153-212), computed by the fused HIP field kernel."""
        from . import rendering
        return rendering.field_forward(self, x, sigma_only=sigma_only, output_transient=output_transient)
