"""CPU restatement of the reference's NaivePCT object encoder, eval mode.  TEST INFRASTRUCTURE ONLY.

Follows src/aligner/networks/pct.py: Embedding.forward :115-124, SA.forward :205-232, NaivePCT.forward :296-317, as a
function of a state_dict (reference key names).  BatchNorm in eval mode (running statistics), Dropout = identity.
Pinned against the reference module itself by oracle/make_golden.py (gen_pct) -> tests/golden/pct_*.npz."""
import math

import torch
import torch.nn.functional as F


def _bn(x, sd, prefix, eps=1e-5):
    shape = (1, -1, 1) if x.dim() == 3 else (1, -1)
    rm, rv = sd[prefix + '.running_mean'].reshape(shape), sd[prefix + '.running_var'].reshape(shape)
    w, b = sd[prefix + '.weight'].reshape(shape), sd[prefix + '.bias'].reshape(shape)
    return (x - rm) / torch.sqrt(rv + eps) * w + b


def _conv(x, w, b=None):
    y = torch.einsum('oc,bcn->bon', w[:, :, 0], x)
    return y if b is None else y + b.reshape(1, -1, 1)


def sa_forward(x, sd, p):
    """x [B, 128, N]"""
    da = sd[p + '.k_conv.weight'].shape[0]
    x_q = _conv(x, sd[p + '.q_conv.weight']).permute(0, 2, 1)
    x_k = _conv(x, sd[p + '.k_conv.weight'])
    x_v = _conv(x, sd[p + '.v_conv.weight'], sd[p + '.v_conv.bias'])
    energy = torch.bmm(x_q, x_k) / math.sqrt(da)
    attention = torch.softmax(energy, dim=-1)
    x_s = torch.bmm(x_v, attention)
    x_s = torch.relu(_bn(_conv(x_s, sd[p + '.trans_conv.weight'], sd[p + '.trans_conv.bias']), sd, p + '.after_norm'))
    return x + x_s


def naive_pct_forward(x, sd):
    """x [T, 3, N] -> [T, 256] (eval mode)."""
    x = torch.relu(_bn(_conv(x, sd['embedding.conv1.weight']), sd, 'embedding.bn1'))
    x = torch.relu(_bn(_conv(x, sd['embedding.conv2.weight']), sd, 'embedding.bn2'))
    x1 = sa_forward(x, sd, 'sa1')
    x2 = sa_forward(x1, sd, 'sa2')
    x3 = sa_forward(x2, sd, 'sa3')
    x4 = sa_forward(x3, sd, 'sa4')
    x = torch.cat([x1, x2, x3, x4], dim=1)
    x = F.leaky_relu(_bn(_conv(x, sd['linear.0.weight']), sd, 'linear.1'), 0.2)
    x = torch.max(x, dim=-1)[0]
    x = torch.relu(_bn(x @ sd['linear1.weight'].t(), sd, 'bn1'))
    x = torch.relu(_bn(x @ sd['linear2.weight'].t() + sd['linear2.bias'], sd, 'bn2'))
    return x


# ---- train mode (batch statistics; Dropout p = 0: the RNG stream of nn.Dropout is not part of the contract) --------
def _bn_train(x, sd, prefix, new_stats, eps=1e-5, momentum=0.1):
    dims = (0, 2) if x.dim() == 3 else (0,)
    shape = (1, -1, 1) if x.dim() == 3 else (1, -1)
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=dims)
    var = x.var(dim=dims, unbiased=False)
    new_stats[prefix + '.running_mean'] = (1 - momentum) * sd[prefix + '.running_mean'] + momentum * mean.detach()
    new_stats[prefix + '.running_var'] = (1 - momentum) * sd[prefix + '.running_var'] + momentum * var.detach() * n / max(n - 1, 1)
    return (x - mean.reshape(shape)) / torch.sqrt(var.reshape(shape) + eps) * sd[prefix + '.weight'].reshape(shape) + sd[prefix + '.bias'].reshape(shape)


def naive_pct_forward_train(x, sd):
    """Train-mode forward (pct.py:296-317 with nn.BatchNorm1d in training mode, Dropout p = 0).  `sd` maps reference
    keys to tensors (parameters may require grad).  Returns (y [T,256], new running statistics)."""
    ns = {}

    def sa(x, p):
        da = sd[p + '.k_conv.weight'].shape[0]
        x_q = _conv(x, sd[p + '.q_conv.weight']).permute(0, 2, 1)
        x_k = _conv(x, sd[p + '.k_conv.weight'])
        x_v = _conv(x, sd[p + '.v_conv.weight'], sd[p + '.v_conv.bias'])
        attention = torch.softmax(torch.bmm(x_q, x_k) / math.sqrt(da), dim=-1)
        x_s = torch.bmm(x_v, attention)
        x_s = torch.relu(_bn_train(_conv(x_s, sd[p + '.trans_conv.weight'], sd[p + '.trans_conv.bias']), sd, p + '.after_norm', ns))
        return x + x_s

    x = torch.relu(_bn_train(_conv(x, sd['embedding.conv1.weight']), sd, 'embedding.bn1', ns))
    x = torch.relu(_bn_train(_conv(x, sd['embedding.conv2.weight']), sd, 'embedding.bn2', ns))
    x1 = sa(x, 'sa1'); x2 = sa(x1, 'sa2'); x3 = sa(x2, 'sa3'); x4 = sa(x3, 'sa4')
    x = torch.cat([x1, x2, x3, x4], dim=1)
    x = F.leaky_relu(_bn_train(_conv(x, sd['linear.0.weight']), sd, 'linear.1', ns), 0.2)
    x = torch.max(x, dim=-1)[0]
    x = torch.relu(_bn_train(x @ sd['linear1.weight'].t(), sd, 'bn1', ns))
    x = torch.relu(_bn_train(x @ sd['linear2.weight'].t() + sd['linear2.bias'], sd, 'bn2', ns))
    return x, ns
