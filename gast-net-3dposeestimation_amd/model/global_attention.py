"""Global graph attention: parameter containers (reference model/global_attention.py).

Same constructor signatures, parameter names/shapes and initialisers as the reference (global_attention.py:12-50,
86-101, 134-146).  The forward arithmetic (global_attention.py:52-82, 103-130) is part of the fused HIP plan
(gast_hip/engine.py: G1 + ATT + G3): g/theta/phi are columns of one GEMM, the additive score
f_ij = LeakyReLU(w_theta.theta_i + w_phi.phi_j) is rank-1 so theta/phi fold into two C-vectors per head, and the
(BT, 2Ci, J, J) concat tensor of the reference is never built.
"""
from __future__ import absolute_import, division

import torch
from torch import nn


class GlobalGraph(nn.Module):
    """Global graph attention layer (one head)."""

    def __init__(self, adj, in_channels, inter_channels=None):
        super(GlobalGraph, self).__init__()
        self.adj = adj
        self.in_channels = in_channels
        self.inter_channels = inter_channels
        self.softmax = nn.Softmax(dim=-1)
        self.relu = nn.ReLU(inplace=True)
        self.leakyrelu = nn.LeakyReLU(0.2)
        self.g_channels = self.in_channels if self.inter_channels == self.in_channels // 2 else self.inter_channels
        assert self.inter_channels > 0
        self.g = nn.Conv1d(self.in_channels, self.g_channels, kernel_size=1, stride=1, padding=0)
        self.theta = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self.phi = nn.Conv1d(self.in_channels, self.inter_channels, kernel_size=1, stride=1, padding=0)
        self.C_k = nn.Parameter(torch.zeros(self.adj.shape, dtype=torch.float))
        self.concat_project = nn.Sequential(nn.Conv2d(self.inter_channels * 2, 1, 1, 1, 0, bias=False))
        nn.init.kaiming_normal_(self.concat_project[0].weight)
        for conv in (self.g, self.theta, self.phi):
            nn.init.kaiming_normal_(conv.weight)
            nn.init.constant_(conv.bias, 0)

    def forward(self, x):
        raise NotImplementedError('GlobalGraph runs inside the fused HIP plan of SpatioTemporalModel')


class MultiGlobalGraph(nn.Module):
    def __init__(self, adj, in_channels, inter_channels, dropout=None):
        super(MultiGlobalGraph, self).__init__()
        self.num_non_local = in_channels // inter_channels
        self.attentions = nn.ModuleList([GlobalGraph(adj, in_channels, inter_channels) for _ in range(self.num_non_local)])
        self.cat_conv = nn.Conv2d(in_channels, in_channels, 1, bias=False)
        self.cat_bn = nn.BatchNorm2d(in_channels, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(dropout) if dropout is not None else None

    def forward(self, x):
        raise NotImplementedError('MultiGlobalGraph runs inside the fused HIP plan of SpatioTemporalModel')


class SingleGlobalGraph(nn.Module):
    """Present in the reference's namespace (global_attention.py:133-173) but unused by gast_net.py (:17 is commented out)."""

    def __init__(self, adj, in_channels, output_channels, dropout=None):
        super(SingleGlobalGraph, self).__init__()
        self.attentions = GlobalGraph(adj, in_channels, output_channels // 2)
        self.bn = nn.BatchNorm2d(in_channels, momentum=0.1)
        self.relu = nn.ReLU(inplace=True)
        self.dropout = nn.Dropout(dropout) if dropout is not None else None

    def forward(self, x):
        raise NotImplementedError('SingleGlobalGraph is not on the accelerated path (unused by the reference model)')
