"""A GATConv case small enough to evaluate BY HAND from the published layer definition -- independent of oracle/sga_oracle.py
and of the HIP kernels, so it pins both to the algorithm itself (PyG 2.2.0 is not installable here: SURVEY.md 8(c), row G).

Layer (Velickovic et al. 2018 as implemented by torch_geometric.nn.GATConv 2.2.0 with add_self_loops=True, shared lin_src/lin_dst,
concat=True, negative_slope=0.2; reference call sites src/aligner/networks/gat.py:36-37,44):
    h_j   = Theta x_j                                   per head
    e_ij  = LeakyReLU_0.2( <a_src, h_j> + <a_dst, h_i> ) for every edge j -> i of (edges without self loops) + one self loop per node
    alpha = softmax over the edges arriving at i (duplicate edges are separate terms)
    out_i = sum_j alpha_ij h_j + bias                   heads concatenated

Graph: 3 nodes; edge list (source, target) = (0,1), (0,1) [a duplicate], (2,1), (1,0), (2,2) [an explicit self loop: removed, then
every node gets exactly one].  So node 1 hears 0 twice, 2 once and itself; node 0 hears 1 and itself; node 2 hears only itself.

Only three of the 2 x 128 channels are non-zero, with numbers chosen so that every exponential is a small rational:

head 0, channel 0:  h = (ln 2, ln 3, -5 ln 2), a_src = 1, a_dst = 0  ->  e_ij = LeakyReLU(h_j) = (ln 2, ln 3, -ln 2), exp = (2, 3, 1/2)
    node 1: weights 2, 2, 1/2, 3 (sum 7.5)      out = (4 ln 2 + 0.5 (-5 ln 2) + 3 ln 3) / 7.5 = 0.2 ln 2 + 0.4 ln 3
    node 0: weights 3 (from 1), 2 (self)        out = (3 ln 3 + 2 ln 2) / 5
    node 2: self only                           out = -5 ln 2
head 0, channel 5:  h = (1, 10, 100), attention vectors 0 there  ->  same alphas as above
    node 1: (4 * 1 + 0.5 * 100 + 3 * 10) / 7.5 = 11.2 ;  node 0: (3 * 10 + 2 * 1) / 5 = 6.4 ;  node 2: 100
head 1, channel 0:  h = (ln 2, ln 3, -ln 3 - 5 ln 2), a_src = a_dst = 1  ->  e_ij = LeakyReLU(h_j + h_i)
    node 1 (h_i = ln 3): from 0: ln 6 -> 6 (twice); from 2: -5 ln 2 -> -ln 2 -> 1/2; self: 2 ln 3 -> 9;  sum 21.5
            out = (12 ln 2 + 0.5 (-ln 3 - 5 ln 2) + 9 ln 3) / 21.5 = (9.5 ln 2 + 8.5 ln 3) / 21.5
    node 0 (h_i = ln 2): from 1: ln 6 -> 6; self: 2 ln 2 -> 4;  out = (6 ln 3 + 4 ln 2) / 10
    node 2: self only                           out = -ln 3 - 5 ln 2
bias: +0.25 on (head 0, ch 0), -1 on (head 0, ch 5), +0.5 on (head 1, ch 0).
(The softmax's +1e-16 in the denominator changes nothing above 1e-16 relative.)"""
import math

import numpy as np

L2, L3 = math.log(2.0), math.log(3.0)
H, C = 2, 128
EDGES = np.array([[0, 1], [0, 1], [2, 1], [1, 0], [2, 2]], dtype=np.int64)      # (source, target)


def inputs():
    """h [3, 256] (= Theta x, head-major), att_src / att_dst / bias [256] as float64 numpy arrays."""
    h = np.zeros((3, H * C))
    h[:, 0] = (L2, L3, -5 * L2)
    h[:, 5] = (1.0, 10.0, 100.0)
    h[:, C + 0] = (L2, L3, -L3 - 5 * L2)
    a_s, a_d, b = np.zeros(H * C), np.zeros(H * C), np.zeros(H * C)
    a_s[0] = 1.0
    a_s[C + 0] = 1.0
    a_d[C + 0] = 1.0
    b[0], b[5], b[C + 0] = 0.25, -1.0, 0.5
    return h, a_s, a_d, b


def expected():
    out = np.zeros((3, H * C))
    out[:, 0] = ((3 * L3 + 2 * L2) / 5, 0.2 * L2 + 0.4 * L3, -5 * L2)
    out[:, 5] = (6.4, 11.2, 100.0)
    out[:, C + 0] = ((6 * L3 + 4 * L2) / 10, (9.5 * L2 + 8.5 * L3) / 21.5, -L3 - 5 * L2)
    _, _, _, b = inputs()
    return out + b[None, :]
