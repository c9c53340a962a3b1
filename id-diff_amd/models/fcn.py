"""``fcn`` score network on the fp32 matrix cores (reference: /root/reference/models/fcn.py:6-40).

mlp = Linear(D+1, H), Dropout, ELU, hidden_layers x [Linear(H, H), Dropout, ELU], Linear(H, D); the time is
appended to x as one more feature.  State-dict keys are ``mlp.{0,3,6,...}.{weight,bias}`` as in the
reference.  Each layer is one idiff_gemm_f32 launch with bias + ELU fused into the epilogue (Dropout is the
identity in eval mode); nn.Linear's [out, in] weight is already the K-contiguous Bt panel the kernel wants.
"""
import torch
import torch.nn as nn

from .. import _lib
from . import utils
from .base import HipScoreModel


@utils.register_model(name='fcn')
class FCN(HipScoreModel):
    def __init__(self, config):
        super().__init__()
        m = config.model
        self.state_size, self.hidden_nodes, self.hidden_layers = m.state_size, m.hidden_nodes, m.hidden_layers
        self.embedding_type = 'None'
        widths = [m.state_size + 1] + [m.hidden_nodes] * (m.hidden_layers + 1)
        layers = []
        for a, b in zip(widths[:-1], widths[1:]):
            layers += [nn.Linear(a, b), nn.Dropout(m.dropout), nn.ELU()]
        layers.append(nn.Linear(m.hidden_nodes, m.state_size))
        self.mlp = nn.Sequential(*layers)

    def _pack(self):
        linears = [l for l in self.mlp if isinstance(l, nn.Linear)]
        first = linears[0]
        kpad = (first.in_features + 3) // 4 * 4  # 101 -> 104: 16-byte rows for the vector loads
        w0 = torch.zeros(first.out_features, kpad, device=first.weight.device, dtype=torch.float32)
        w0[:, :first.in_features] = first.weight
        weights = [w0.contiguous()] + [l.weight.detach().float().contiguous() for l in linears[1:]]
        biases = [l.bias.detach().float().contiguous() for l in linears]
        return {"w": weights, "b": biases, "kpad": kpad}

    def forward(self, x, t, out_rowscale=None):
        x, t = self._check_inputs(x, t)
        if x.ndim != 2 or x.shape[1] != self.state_size:
            raise NotImplementedError("fcn on the manifold_dimension path takes [batch, state_size] inputs")
        pk = self.packed()
        rows = x.shape[0]
        kpad = pk["kpad"]
        tpad = torch.zeros(rows, kpad - self.state_size, device=x.device, dtype=torch.float32)
        tpad[:, 0] = t
        h = torch.empty(rows, kpad, device=x.device, dtype=torch.float32)
        _lib.concat_cols(x, self.state_size, tpad, kpad - self.state_size, h, rows)
        n_layers = len(pk["w"])
        for i, (w, b) in enumerate(zip(pk["w"], pk["b"])):
            last = i == n_layers - 1
            ep = _lib.make_epilogue(bias=b, act=None if last else "elu",
                                    rowscale=out_rowscale if last else None)
            h = _lib.gemm(h, w, epilogue=ep)
        return h
