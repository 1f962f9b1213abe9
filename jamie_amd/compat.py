"""Autograd-compatible model for the reference's own seam: `JAMIE(model_class=...)`.

The reference's `project_jamie` (jamie/jamie.py:472-479, 611, 734-741) builds `model_class(input_dim, output_dim,
preprocessing=..., preprocessing_inverse=..., dropout=...)`, calls `model(*data, corr=corr)` in train mode, forms its
four losses with torch ops on the returned tensors, calls `.backward()`, `clip_grad_norm_(model.parameters(), 1)` and
`optim.Adam(model.parameters()).step()`.  `edModelVarTorch` satisfies that protocol on the MI355X: the forward and
the backward of the whole network are the HIP kernels of `TrainEngine` wrapped in ONE `torch.autograd.Function`;
the model exposes ONE parameter (the flat fp32 buffer), so torch's clip and Adam act on exactly the same numbers as
the reference's 45 tensors (global norm and element-wise Adam do not depend on the tensor partition).

The fused path (`jamie_amd.JAMIE`) is faster (losses, clip and Adam are fused kernels too); this class exists so
that the reference's loop itself can run unchanged on the GPU:

    from jamie import JAMIE                      # the reference package
    from jamie_amd.compat import edModelVarTorch
    JAMIE(model_class=edModelVarTorch, device='cuda', use_f_tilde=False, pca_dim=None).fit_transform([X0, X1])

Randomness (dropout masks, eps) comes from the engine's Philox streams, not from torch's generator.
"""
import torch

from . import _native as nv
from .engine import TrainEngine
from .model import edModelVar


class _CoupledVAEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, flat, x0, x1, corr, owner):
        eng = owner._engine_for(x0.shape[0])
        eng.set_batch([x0.detach().float().contiguous(), x1.detach().float().contiguous()])
        B = x0.shape[0]
        identity = corr is None or (corr.shape[0] == corr.shape[1] and bool(torch.equal(
            corr, torch.eye(B, device=corr.device, dtype=corr.dtype))))
        c = None if identity else corr.detach().float().contiguous()
        ctx.lat = eng.forward_only(c)
        ctx.eng, ctx.corr = eng, c
        w = eng.ws
        outs = [w[0]['z'], w[1]['z'], w[0]['comb'], w[1]['comb'], w[0]['xhat'][0], w[1]['xhat'][0],
                w[0]['mu'], w[1]['mu'], w[1]['lv']]
        eng.state[1] += 1                      # a new Philox step per forward (the fused path does it in the norm kernel)
        return tuple(t.clone() for t in outs)

    @staticmethod
    def backward(ctx, dz0, dz1, dc0, dc1, dx0, dx1, dm0, dm1, dlv):
        eng = ctx.eng
        eng.state[1] -= 1                      # backward regenerates the dropout masks of ITS forward
        eng.backward_external(ctx.lat, [dz0, dz1], [dc0, dc1], [dx0, dx1], [dm0, dm1], dlv)
        eng.state[1] += 1
        return eng.grad.clone(), None, None, None, None


class edModelVarTorch(torch.nn.Module):
    """Reference-protocol model (reference jamie/model.py:116-282) whose train-mode forward/backward run on the
    HIP kernels.  Eval-mode calls, `impute`, `encoders[i]`, `fc_mus[i]`, `preprocessing*`, `num_modalities`,
    `state_dict()` (reference names) behave like `jamie_amd.edModelVar`."""

    def __init__(self, input_dim, output_dim, preprocessing=None, preprocessing_inverse=None, sigma=None,
                 dropout=None):
        super().__init__()
        self.inner = edModelVar(input_dim, output_dim, preprocessing=preprocessing,
                                preprocessing_inverse=preprocessing_inverse, dropout=dropout)
        self.flat = torch.nn.Parameter(self.inner.flat)           # shares storage with the kernels' buffer
        self.num_modalities = self.inner.num_modalities
        self.preprocessing = self.inner.preprocessing
        self.preprocessing_inverse = self.inner.preprocessing_inverse
        self.encoders = self.inner.encoders
        self.fc_mus = self.inner.fc_mus
        self._engines = {}

    def _engine_for(self, B):
        if B not in self._engines:
            eng = TrainEngine(self.inner, B, loss_weights=[0, 0, 0, 0])   # losses are the caller's
            eng.hyper[:4] = 0
            self._engines[B] = eng
        return self._engines[B]

    def to(self, device):
        if torch.device(device).type != 'cuda':
            raise nv.JamieHipError('edModelVarTorch lives on the GPU only (no CPU fallback)')
        return self

    def train(self, mode=True):
        self.inner.train(mode)
        return super().train(mode)

    def forward(self, *X, corr):
        if not self.training:
            return self.inner(*X, corr=corr)
        o = _CoupledVAEFn.apply(self.flat, X[0], X[1], corr, self)
        return [o[0], o[1]], [o[2], o[3]], [o[4], o[5]], [o[6], o[7]], o[8]

    def impute(self, X, compose):
        return self.inner.impute(X, compose)

    def state_dict(self, *a, **k):
        return self.inner.state_dict()

    def load_state_dict(self, sd, *a, **k):
        self.inner.load_state_dict(sd)
