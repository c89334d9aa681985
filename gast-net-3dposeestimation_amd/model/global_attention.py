"""Global graph attention (reference model/global_attention.py).

Same constructor signatures, parameter names/shapes and initialisers as the reference (global_attention.py:12-50,
86-101, 134-146).  Inside SpatioTemporalModel the forward arithmetic (global_attention.py:52-82, 103-130) is part of the fused
HIP plan (gast_hip/engine.py: G1 + ATT + G3): g/theta/phi are columns of one GEMM, the additive score
f_ij = LeakyReLU(w_theta.theta_i + w_phi.phi_j) is rank-1 so theta/phi fold into two C-vectors per head, and the
(BT, 2Ci, J, J) concat tensor of the reference is never built.  Called on their own, the modules run the same kernels as a
small forward-only plan (gast_hip/modules.py).

What the `state_dict` contract and the seed-for-seed initialisation test (tests/test_host_contract.py) need is kept:
the registration ORDER of the sub-modules (it fixes both the key order and the order of the RNG draws) and their
initialisers.  Activation helper modules of the reference (parameter-free) are not re-created.
"""
from __future__ import absolute_import, division

import torch
from torch import nn


def _pointwise(c_in, c_out):
    """1x1 Conv1d with bias (reference global_attention.py:30-35)."""
    return nn.Conv1d(c_in, c_out, kernel_size=1, stride=1, padding=0)


class GlobalGraph(nn.Module):
    """One attention head: g / theta / phi projections, the learnable (J, J) offset C_k and the 2Ci -> 1 score projection."""

    def __init__(self, adj, in_channels, inter_channels=None):
        super().__init__()
        assert inter_channels is not None and inter_channels > 0
        self.adj, self.in_channels, self.inter_channels = adj, in_channels, inter_channels
        # the value projection keeps the full width when the head is half as wide as its input (reference :25-28)
        self.g_channels = in_channels if inter_channels == in_channels // 2 else inter_channels
        # registration order = state_dict order = RNG order: g, theta, phi, C_k, concat_project
        for name, width in (('g', self.g_channels), ('theta', inter_channels), ('phi', inter_channels)):
            setattr(self, name, _pointwise(in_channels, width))
        self.C_k = nn.Parameter(torch.zeros(adj.shape, dtype=torch.float))
        self.concat_project = nn.Sequential(nn.Conv2d(2 * inter_channels, 1, 1, 1, 0, bias=False))
        # initialisers (reference :44-50): score projection first, then g, theta, phi (Kaiming weights, zero biases)
        nn.init.kaiming_normal_(self.concat_project[0].weight)
        for proj in (self.g, self.theta, self.phi):
            nn.init.kaiming_normal_(proj.weight)
            nn.init.constant_(proj.bias, 0)

    def forward(self, x):
        """x: (B*T, C, J) -> (B*T, g_channels, J)   (reference :52-82)"""
        from gast_hip.modules import global_graph_forward
        return global_graph_forward(self, x)


class MultiGlobalGraph(nn.Module):
    """in_channels // inter_channels heads + the C -> C mixing convolution with its BatchNorm (reference :86-101)."""

    def __init__(self, adj, in_channels, inter_channels, dropout=None):
        super().__init__()
        self.num_non_local = in_channels // inter_channels
        heads = [GlobalGraph(adj, in_channels, inter_channels) for _ in range(self.num_non_local)]
        self.attentions = nn.ModuleList(heads)
        self.cat_conv = nn.Conv2d(in_channels, in_channels, 1, bias=False)
        self.cat_bn = nn.BatchNorm2d(in_channels, momentum=0.1)
        self.dropout = None if dropout is None else nn.Dropout(dropout)

    def forward(self, x):
        """x: (B, T, J, C) -> (B, T, J, C)   (reference :103-130)"""
        from gast_hip.modules import multi_global_forward
        return multi_global_forward(self, x)


class SingleGlobalGraph(nn.Module):
    """One full-width head + BatchNorm (reference global_attention.py:133-173); unreachable from gast_net.py (:17 is commented out
    there) but part of the namespace `from model.gast_net import *` exports."""

    def __init__(self, adj, in_channels, output_channels, dropout=None):
        super().__init__()
        self.attentions = GlobalGraph(adj, in_channels, output_channels // 2)
        self.bn = nn.BatchNorm2d(in_channels, momentum=0.1)
        self.dropout = None if dropout is None else nn.Dropout(dropout)

    def forward(self, x):
        """x: (B, T, J, C) -> (B, T, J, C)   (reference :148-173)"""
        from gast_hip.modules import single_global_forward
        return single_global_forward(self, x)
