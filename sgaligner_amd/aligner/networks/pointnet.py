"""PointNet object encoder -- drop-in for reference src/aligner/networks/pointnet.py:87-175
(`PointNetfeat`), forward/backward on the fused HIP kernels (csrc/pointnet.hip)."""
import torch
import torch.nn as nn

from ... import ops
from .base import BaseNetwork


class PointNetfeat(BaseNetwork):
    """Same constructor, attributes and state_dict keys as the reference class (pointnet.py:88-118):
    conv{1,2,3} are Conv1d(k=1) holders for the [out,in,1] weights; bn{1,2,3} exist because the reference
    creates (and checkpoints) them although their OUTPUT IS DISCARDED (pointnet.py:141-142,154-155,
    158-159) -- they never enter y and receive no gradient.

    HIP path: global_feat=True, input_transform=False, feature_transform=False (the only configuration
    the reference instantiates, sg_aligner.py:58).  Anything else raises.
    """

    def __init__(self, global_feat=True, input_transform=True, feature_transform=False, point_size=3, out_size=1024,
                 batch_norm=True, init_weights=True, pointnet_str=None):
        super().__init__()
        if input_transform or feature_transform or not global_feat or point_size != 3:
            raise NotImplementedError('sgaligner_amd PointNetfeat: only global_feat=True, input_transform=False, '
                                      'feature_transform=False, point_size=3 is implemented (sg_aligner.py:58)')
        self.name = 'pnetenc'
        self.use_batch_norm = batch_norm
        self.point_size = point_size
        self.out_size = out_size
        self.global_feat = global_feat
        self.input_transform = input_transform
        self.feature_transform = feature_transform
        self.conv1 = nn.Conv1d(point_size, 64, 1)
        self.conv2 = nn.Conv1d(64, 128, 1)
        self.conv3 = nn.Conv1d(128, out_size, 1)
        if batch_norm:
            self.bn1 = nn.BatchNorm1d(64)
            self.bn2 = nn.BatchNorm1d(128)
            self.bn3 = nn.BatchNorm1d(out_size)
        if init_weights:                                     # pointnet.py:116-118
            self.init_weights('constant', 1, target_op='BatchNorm')
            self.init_weights('xavier_normal', 1)

    def forward(self, x, return_meta=False):
        """x [T,3,P] as in the reference (a permuted view of data_dict['tot_obj_pts'] [T,P,3])."""
        assert x.ndim > 2
        xt = x.permute(0, 2, 1)                               # -> [T,P,3]; free when x is the reference's view
        if not xt.is_contiguous():
            xt = xt.contiguous()
        y = ops.pointnet(xt.float() if xt.dtype != torch.float32 else xt, self.conv1.weight, self.conv1.bias,
                         self.conv2.weight, self.conv2.bias, self.conv3.weight, self.conv3.bias)
        if return_meta:
            return y, torch.zeros([1]), torch.zeros([1])       # pointnet.py:138,151 dummies
        return y
