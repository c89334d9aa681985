"""Channel-shared semantic graph convolution and its LocalGraph (reference model/sem_graph_conv.py).

The reference ships this module next to local_attention.py; gast_net.py does not import it (its LocalGraph is the channel-wise
one), but checkpoints and scripts written against the SemGCN-style layer can.  Same constructor signatures, parameter names,
shapes and initialisers (sem_graph_conv.py:15-33, 59-128).  The adjacency is ONE masked softmax shared by all channels
(e: (1, nnz)), i.e. the channel-wise kernels of the fused plan with `e` broadcast over the channels; bias defaults to True.
Fused forward-only plan under torch.no_grad() and a trainable path (forward kernel + the fused plan's backward kernels as
autograd.Functions) otherwise: gast_hip/modules.py, gast_hip/autograd_ops.py.
"""
from __future__ import absolute_import, division

import math

import torch
import torch.nn as nn

from model.local_attention import skeleton_patterns


class SemGraphConv(nn.Module):
    """
    Semantic graph convolution layer
    """

    def __init__(self, in_features, out_features, adj, bias=True):
        super(SemGraphConv, self).__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.W = nn.Parameter(torch.zeros(size=(2, in_features, out_features), dtype=torch.float))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        # plain attributes, not buffers (reference :23-24)
        self.adj = adj
        self.m = (self.adj > 0)
        self.e = nn.Parameter(torch.zeros(1, int(self.m.sum().item()), dtype=torch.float))
        nn.init.constant_(self.e.data, 1)
        if bias:
            self.bias = nn.Parameter(torch.zeros(out_features, dtype=torch.float))
            stdv = 1. / math.sqrt(self.W.size(2))
            self.bias.data.uniform_(-stdv, stdv)
        else:
            self.register_parameter('bias', None)

    def forward(self, input):
        """input: (B, T, J, C_in) -> (B, T, J, C_out):  (A o I) X W0 + (A o (1 - I)) X W1 + bias   (reference :35-52)"""
        from gast_hip.modules import graph_conv_forward
        return graph_conv_forward(self, input, shared=True)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class LocalGraph(nn.Module):
    def __init__(self, adj, input_dim, output_dim, dropout=None):
        super(LocalGraph, self).__init__()
        adj_sym, adj_con = skeleton_patterns(adj)          # (KeyError for unsupported skeletons, like the reference :88-90)
        self.gcn_sym = SemGraphConv(input_dim, output_dim, adj_sym)
        self.bn_1 = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.gcn_con = SemGraphConv(input_dim, output_dim, adj_con)
        self.bn_2 = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.relu = nn.ReLU()
        self.cat_conv = nn.Conv2d(2 * output_dim, output_dim, 1, bias=False)
        self.cat_bn = nn.BatchNorm2d(output_dim, momentum=0.1)
        self.dropout = nn.Dropout2d(dropout) if dropout is not None else None

    def forward(self, input):
        """input: (B, T, J, C) -> (B, T, J, C_out)   (reference :130-153)"""
        from gast_hip.modules import local_graph_forward
        return local_graph_forward(self, input, shared=True, dropout2d=True)
