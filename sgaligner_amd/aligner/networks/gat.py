"""GAT structure encoder -- drop-in for reference src/aligner/networks/gat.py:27-48 (`MultiGAT`) and
for the torch_geometric.nn.GATConv layers it stacks (PyG 2.2.0 parameter names, so released
checkpoints load with strict=True: lin_src.weight / lin_dst.weight (aliased), att_src, att_dst, bias)."""
import math

import numpy as np
import torch
import torch.nn as nn

from ... import ops


class GATConv(nn.Module):
    """Parameter holder with PyG-2.2.0 GATConv names / shapes / init (glorot weights, zero bias)."""

    def __init__(self, in_channels, out_channels, heads=1):
        super().__init__()
        self.in_channels, self.out_channels, self.heads = in_channels, out_channels, heads
        self.lin_src = nn.Linear(in_channels, heads * out_channels, bias=False)
        self.lin_dst = self.lin_src                     # shared when in_channels is an int (PyG 2.2.0)
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.zeros(heads * out_channels))
        for w in (self.lin_src.weight, self.att_src, self.att_dst):
            fan = w.size(-2) + w.size(-1)
            a = math.sqrt(6.0 / fan)
            nn.init.uniform_(w, -a, a)

    def params(self):
        return (self.lin_src.weight, self.att_src, self.att_dst, self.bias)


class MultiGAT(nn.Module):
    def __init__(self, n_units=[17, 128, 100], n_heads=[2, 2], dropout=0.0):
        super().__init__()
        self.num_layers = len(n_units) - 1
        self.dropout = dropout
        if dropout != 0.0:
            raise NotImplementedError('sgaligner_amd MultiGAT: dropout must be 0.0 (reference default, sg_aligner.py:39)')
        if self.num_layers != 2 or list(n_units[1:]) != [128, 128] or list(n_heads) != [2, 2]:
            raise NotImplementedError('sgaligner_amd MultiGAT: the HIP path implements n_units=[F,128,128], n_heads=[2,2] '
                                      '(hard-coded in the reference, sg_aligner.py:38,66)')
        layers = []
        for i in range(self.num_layers):                                    # gat.py:34-37
            in_c = n_units[i] * n_heads[i - 1] if i else n_units[i]
            layers.append(GATConv(in_c, n_units[i + 1], n_heads[i]))
        self.layer_stack = nn.ModuleList(layers)

    def forward_batched(self, x, graph_batch):
        """All graphs of a batch in one launch per layer (x [T,F], graph_batch: ops.GraphBatch)."""
        return ops.multi_gat(graph_batch, x, self.layer_stack[0].params(), self.layer_stack[1].params())

    def forward(self, x, edges):
        """Reference signature (gat.py:40): one graph, x [N,F], edges [2,E] (row 0 source, row 1 target)."""
        e = edges.t().to(torch.int64).contiguous()
        gb = ops.GraphBatch(np.asarray([x.shape[0]]), np.asarray([e.shape[0]]), e)
        return self.forward_batched(x, gb)
