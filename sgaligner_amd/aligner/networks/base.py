"""Weight-initialisation helper with the semantics of reference src/aligner/networks/base.py:5-56
(`init_weights(init_type, gain, bias_value, target_op)`: initialise every sub-module's `weight` once,
optionally only modules whose class name contains `target_op`, biases to a constant)."""
import torch.nn as nn

_INITS = {
    'normal': lambda w, gain: nn.init.normal_(w, 0.0, gain),
    'xavier_normal': lambda w, gain: nn.init.xavier_normal_(w, gain=gain),
    'kaiming': lambda w, gain: nn.init.kaiming_normal_(w, a=0, mode='fan_in'),
    'orthogonal': lambda w, gain: nn.init.orthogonal_(w, gain=gain),
    'xavier_unifrom': lambda w, gain: nn.init.xavier_uniform_(w, gain=gain),   # (sic) reference spelling, base.py:33
    'constant': lambda w, gain: nn.init.constant_(w, gain),
}


class BaseNetwork(nn.Module):
    def init_weights(self, init_type='normal', gain=0.02, bias_value=0.0, target_op=None):
        if init_type not in _INITS:
            raise NotImplementedError(init_type)
        for m in self.modules():
            if m is self:
                continue
            if target_op is not None and type(m).__name__.find(target_op) == -1:
                continue
            if getattr(m, 'param_inited', False):
                continue
            if getattr(m, 'weight', None) is not None:
                _INITS[init_type](m.weight.data, gain)
            if getattr(m, 'bias', None) is not None:
                nn.init.constant_(m.bias.data, bias_value)
            m.param_inited = True

    def getParamList(self, x):
        return list(x.parameters())
