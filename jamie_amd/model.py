"""Coupled-VAE model container for the MI355X path: the counterpart of the reference's `edModelVar`
(reference jamie/model.py:116-282) behind the same protocol the reference's `JAMIE` drives through its
`model_class=` seam (jamie/jamie.py:472-479, 611, 794, 806-837).

All parameters live in ONE flat fp32 device buffer (every tensor a 16-byte aligned view) so that the
optimiser is two streaming kernels and the data-parallel gradient exchange is one RCCL all-reduce.
Names exported by `state_dict()` are the reference's (`encoders.0.0.weight`, `fc_mus.1.bias`, ...).
"""
from collections import OrderedDict
import math

import numpy as np
import torch

from . import _native as nv

BN_EPS = 1e-5
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.01


def identity(x):
    """Identity preprocessing (picklable), reference utilities.py:48-50."""
    return x


def _align4(n):
    # 8 elements: 32-byte boundaries for the fp32 views and 16-byte boundaries for the bf16 copy of the same layout
    return (n + 7) // 8 * 8


class ParamLayout:
    """Offsets of every parameter tensor inside the flat buffer.

    Tensors (d = input_dim[i], L = output_dim), modality by modality inside a layer:
      enc0: W [2d,d] b [2d] bn0.g bn0.b [2d] | enc1: W [d,2d] b [d] bn1.g bn1.b [d] |
      head: W [2L,d] (rows 0..L-1 = fc_mus, L..2L-1 = fc_vars) b [2L] | sigma [M] |
      dec0: W [d,L] b [d] bn2.g bn2.b [d] | dec1: W [2d,d] b [2d] bn3.g bn3.b [2d] | dec2: W [d,2d] b [d]
    The two heads of a modality are adjacent so that mu and logvar come out of one GEMM.
    Memory: region `rep` (all the small tensors + head + dec0), then the W matrices of enc0 | enc1 | dec1 | dec2.
    """

    BIG_LAYERS = ('enc0', 'enc1', 'dec1', 'dec2')
    SHARD_ALIGN = 512            # elements: 8-element (16 B as bf16) pieces for every world size that divides 64

    def __init__(self, input_dim, output_dim, real_dim=None):
        """`input_dim`: the feature counts the kernels see; `real_dim` (default: the same) the model's own feature counts
        when `input_dim` is padded (edModelVar(pad_features=8)): every tensor then holds its real block in the leading
        rows / columns (`real[name]` = that block's shape) and zeros elsewhere."""
        self.input_dim = list(input_dim)
        self.real_dim = list(input_dim) if real_dim is None else list(real_dim)
        self.L = output_dim
        self.M = len(input_dim)
        self.entries = OrderedDict()
        self.real = OrderedDict()
        off = 0
        L = output_dim

        def add(name, shape, real_shape=None):
            nonlocal off
            n = int(np.prod(shape))
            self.entries[name] = (off, tuple(shape))
            self.real[name] = tuple(shape if real_shape is None else real_shape)
            off += _align4(n)

        # Memory order: one region `rep` with everything small (sigma, every bias and BatchNorm affine pair, and the two skinny
        # layers whole: heads and decoder layer 0 -- 1 % of the parameters, read in fp32 by the latent kernels), then the four
        # large weight-matrix regions layer by layer in FORWARD order.  The backward pass completes the large regions from
        # the END of the buffer towards the start, so a data-parallel exchange can start on contiguous tail regions while the
        # earlier layers are still in their backward GEMMs (`regions`); `rep` is complete last (with enc0) and adjacent to it.
        # A large region holds nothing but its weight matrices and is a multiple of SHARD_ALIGN elements long: under a
        # sharded optimiser (distributed.ShardedGradExchange) it is cut into world-size equal, 16-byte aligned pieces, while
        # `rep` stays replicated on every rank.
        layers = [('enc0', lambda d: [('enc0.W', (2 * d, d)), ('enc0.b', (2 * d,)), ('bn0.g', (2 * d,)), ('bn0.b', (2 * d,))]),
                  ('enc1', lambda d: [('enc1.W', (d, 2 * d)), ('enc1.b', (d,)), ('bn1.g', (d,)), ('bn1.b', (d,))]),
                  ('head', lambda d: [('head.W', (2 * L, d)), ('head.b', (2 * L,))]),
                  ('dec0', lambda d: [('dec0.W', (d, L)), ('dec0.b', (d,)), ('bn2.g', (d,)), ('bn2.b', (d,))]),
                  ('dec1', lambda d: [('dec1.W', (2 * d, d)), ('dec1.b', (2 * d,)), ('bn3.g', (2 * d,)), ('bn3.b', (2 * d,))]),
                  ('dec2', lambda d: [('dec2.W', (d, 2 * d)), ('dec2.b', (d,))])]

        def align_region():
            nonlocal off
            off = (off + self.SHARD_ALIGN - 1) // self.SHARD_ALIGN * self.SHARD_ALIGN

        def is_big(lname, nm):
            return lname in self.BIG_LAYERS and nm.endswith('.W')
        self.regions = {}
        start = off
        add('sigma', (self.M,))               # sigma's gradient is produced with the heads' (latent backward)
        for lname, spec in layers:
            for i, d in enumerate(input_dim):
                for (nm, shape), (_, rshape) in zip(spec(d), spec(self.real_dim[i])):
                    if not is_big(lname, nm):
                        add(f'm{i}.{nm}', shape, rshape)
        align_region()
        self.regions['rep'] = (start, off)
        for lname, spec in layers:
            if lname not in self.BIG_LAYERS:
                continue
            start = off
            for i, d in enumerate(input_dim):
                for (nm, shape), (_, rshape) in zip(spec(d), spec(self.real_dim[i])):
                    if is_big(lname, nm):
                        add(f'm{i}.{nm}', shape, rshape)
            align_region()
            self.regions[lname] = (start, off)
        self.total = off
        # BN running statistics (not optimised): separate flat buffer
        self.bn_entries = OrderedDict()
        boff = 0
        for i, d in enumerate(input_dim):
            r = self.real_dim[i]
            for k, n, nr in (('bn0', 2 * d, 2 * r), ('bn1', d, r), ('bn2', d, r), ('bn3', 2 * d, 2 * r)):
                self.bn_entries[f'm{i}.{k}.mean'] = (boff, (n,)); boff += _align4(n)
                self.bn_entries[f'm{i}.{k}.var'] = (boff, (n,)); boff += _align4(n)
                self.real[f'm{i}.{k}.mean'] = self.real[f'm{i}.{k}.var'] = (nr,)
        self.bn_total = boff
        self.padded = self.input_dim != self.real_dim

    def views(self, flat):
        return {k: flat[o:o + int(np.prod(s))].view(*s) for k, (o, s) in self.entries.items()}

    def bn_views(self, flat):
        return {k: flat[o:o + int(np.prod(s))].view(*s) for k, (o, s) in self.bn_entries.items()}

    def num_parameters(self):
        """Reference count: sum_i(8d^2 + 3dL + 19d + 2L) + M; padding is not counted."""
        return sum(int(np.prod(self.real[k])) for k in self.entries)

    def unpad(self, name, t):
        """The real block of (padded) tensor `t` of entry `name`."""
        r = self.real[name]
        if tuple(t.shape) == r:
            return t
        return t[tuple(slice(0, n) for n in r)]

    # ---- mapping to the reference's state_dict names (model.py:147-220) ----
    def reference_names(self):
        L = self.L
        out = OrderedDict()
        out['sigma'] = ('sigma', None)
        for i in range(self.M):
            p = f'm{i}.'
            out[f'encoders.{i}.0.weight'] = (p + 'enc0.W', None); out[f'encoders.{i}.0.bias'] = (p + 'enc0.b', None)
            out[f'encoders.{i}.1.weight'] = (p + 'bn0.g', None); out[f'encoders.{i}.1.bias'] = (p + 'bn0.b', None)
            out[f'encoders.{i}.4.weight'] = (p + 'enc1.W', None); out[f'encoders.{i}.4.bias'] = (p + 'enc1.b', None)
            out[f'encoders.{i}.5.weight'] = (p + 'bn1.g', None); out[f'encoders.{i}.5.bias'] = (p + 'bn1.b', None)
        for i in range(self.M):
            out[f'fc_mus.{i}.weight'] = (f'm{i}.head.W', slice(0, L)); out[f'fc_mus.{i}.bias'] = (f'm{i}.head.b', slice(0, L))
        for i in range(self.M):
            out[f'fc_vars.{i}.weight'] = (f'm{i}.head.W', slice(L, 2 * L)); out[f'fc_vars.{i}.bias'] = (f'm{i}.head.b', slice(L, 2 * L))
        for i in range(self.M):
            p = f'm{i}.'
            out[f'decoders.{i}.0.weight'] = (p + 'dec0.W', None); out[f'decoders.{i}.0.bias'] = (p + 'dec0.b', None)
            out[f'decoders.{i}.1.weight'] = (p + 'bn2.g', None); out[f'decoders.{i}.1.bias'] = (p + 'bn2.b', None)
            out[f'decoders.{i}.4.weight'] = (p + 'dec1.W', None); out[f'decoders.{i}.4.bias'] = (p + 'dec1.b', None)
            out[f'decoders.{i}.5.weight'] = (p + 'bn3.g', None); out[f'decoders.{i}.5.bias'] = (p + 'bn3.b', None)
            out[f'decoders.{i}.8.weight'] = (p + 'dec2.W', None); out[f'decoders.{i}.8.bias'] = (p + 'dec2.b', None)
        return out

    def reference_bn_names(self):
        out = OrderedDict()
        for i in range(self.M):
            for ref, mine in ((f'encoders.{i}.1', 'bn0'), (f'encoders.{i}.5', 'bn1'),
                              (f'decoders.{i}.1', 'bn2'), (f'decoders.{i}.5', 'bn3')):
                out[ref + '.running_mean'] = f'm{i}.{mine}.mean'
                out[ref + '.running_var'] = f'm{i}.{mine}.var'
        return out


def _linear_init(out_f, in_f):
    """torch.nn.Linear.reset_parameters (kaiming_uniform_(a=sqrt(5)) + bias bound 1/sqrt(fan_in)), drawn
    from the global CPU torch RNG exactly like the reference's `nn.Linear(in_f, out_f)`."""
    gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
    bound = math.sqrt(3.0) * (gain / math.sqrt(in_f))
    w = torch.empty(out_f, in_f).uniform_(-bound, bound)
    bb = 1 / math.sqrt(in_f)
    b = torch.empty(out_f).uniform_(-bb, bb)
    return w, b


class _EvalEncoder:
    """`model.encoders[i]` in eval mode (used as `fc_mus[i](encoders[i](x))`, reference jamie.py:836)."""

    def __init__(self, model, i):
        self.model, self.i = model, i

    def __call__(self, x):
        return self.model._encode_eval(self.i, x)


class _EvalMuHead:
    def __init__(self, model, i):
        self.model, self.i = model, i

    def __call__(self, h):
        return self.model._mu_eval(self.i, h)


class edModelVar:
    """MI355X counterpart of the reference's `edModelVar` (model.py:116-282); two modalities like the reference,
    or 3-4 fully paired ones (build-defined generalisation).

    Constructor arguments follow the reference (`input_dim`, `output_dim`, `preprocessing`,
    `preprocessing_inverse`, `sigma` (unused there too), `dropout`).  Parameters are initialised from the
    global torch CPU RNG in the reference's order, so `torch.manual_seed(s)` gives the reference's
    initial weights.  Training is driven by `jamie_amd.engine.TrainEngine`; this class owns the state and
    the eval-mode forward paths (`__call__`, `impute`, `encoders[i]`, `fc_mus[i]`).
    """

    def __init__(self, input_dim, output_dim, preprocessing=None, preprocessing_inverse=None, sigma=None,
                 dropout=None, device='cuda', pad_features=1):
        """`pad_features=8` (what the bf16 compute mode needs when a feature count is not a multiple of 8, e.g. BASELINE
        config 4's 500): the buffers are laid out for feature counts rounded up to that multiple (`pdims`); the padding
        weights are zero and provably stay zero (a padded input column is 0, so a padded hidden unit's pre-activation is
        constant 0, BatchNorm maps it to beta = 0, its outgoing weights are 0, hence every gradient in the padding is
        exactly 0 and Adam leaves 0 at 0), so the model computes exactly what the unpadded one does."""
        nv.require_gpu()
        if not 2 <= len(input_dim) <= 4:
            raise NotImplementedError('2..4 modalities (the reference itself supports exactly two, jamie.py:420)')
        self.input_dim = [int(d) for d in input_dim]
        self.output_dim = int(output_dim)
        self.num_modalities = len(input_dim)
        self.preprocessing = self.num_modalities * [identity] if preprocessing is None else preprocessing
        self.preprocessing_inverse = (self.num_modalities * [identity] if preprocessing_inverse is None
                                      else preprocessing_inverse)
        if dropout is None:                                   # model.py:144-145
            dropout = .6 if max(self.input_dim) > 64 else 0
        self.dropout = float(dropout)
        self.device = torch.device(device)
        pad = max(1, int(pad_features))
        self.pdims = [(d + pad - 1) // pad * pad for d in self.input_dim]
        self.layout = ParamLayout(self.pdims, self.output_dim, self.input_dim)
        self.training = True
        host = torch.zeros(self.layout.total)
        hv = self.layout.views(host)
        L = self.output_dim
        # ---- reference construction order (RNG order); with padding the real block is filled, the rest stays 0 ----
        def put(name, w, b):
            hv[name + '.W'][:w.shape[0], :w.shape[1]] = w
            hv[name + '.b'][:b.shape[0]] = b
        for i, d in enumerate(self.input_dim):
            put(f'm{i}.enc0', *_linear_init(2 * d, d))
            put(f'm{i}.enc1', *_linear_init(d, 2 * d))
        for i, d in enumerate(self.input_dim):
            w, b = _linear_init(L, d)
            hv[f'm{i}.head.W'][:L, :d], hv[f'm{i}.head.b'][:L] = w, b
        for i, d in enumerate(self.input_dim):
            w, b = _linear_init(L, d)
            hv[f'm{i}.head.W'][L:, :d], hv[f'm{i}.head.b'][L:] = w, b
        for i, d in enumerate(self.input_dim):
            put(f'm{i}.dec0', *_linear_init(d, L))
            put(f'm{i}.dec1', *_linear_init(2 * d, d))
            put(f'm{i}.dec2', *_linear_init(d, 2 * d))
        hv['sigma'][:] = torch.rand(self.num_modalities)
        for i in range(self.num_modalities):
            for k in ('bn0', 'bn1', 'bn2', 'bn3'):
                self.layout.unpad(f'm{i}.{k}.g', hv[f'm{i}.{k}.g']).fill_(1.0)
        self.flat = host.to(self.device)
        self.p = self.layout.views(self.flat)
        bn_host = torch.zeros(self.layout.bn_total)
        for k, v in self.layout.bn_views(bn_host).items():
            if k.endswith('.var'):
                v.fill_(1.0)
        self.bn_flat = bn_host.to(self.device)
        self.bn = self.layout.bn_views(self.bn_flat)
        self.num_batches_tracked = 0
        self.encoders = [_EvalEncoder(self, i) for i in range(self.num_modalities)]
        self.fc_mus = [_EvalMuHead(self, i) for i in range(self.num_modalities)]

    # ---- nn.Module-like protocol used by the reference's driver ----
    def to(self, device):
        if torch.device(device).type != 'cuda':
            raise nv.JamieHipError('jamie_amd.edModelVar lives on the GPU only (no CPU fallback)')
        return self

    def train(self, mode=True):
        self.training = mode
        return self

    def eval(self):
        return self.train(False)

    def parameters(self):
        return [self.flat]

    def num_parameters(self):
        return self.layout.num_parameters()

    def state_dict(self):
        out = OrderedDict()
        for ref, (mine, sl) in self.layout.reference_names().items():
            t = self.layout.unpad(mine, self.p[mine])
            out[ref] = (t if sl is None else t[sl]).detach().clone()
        for ref, mine in self.layout.reference_bn_names().items():
            out[ref] = self.layout.unpad(mine, self.bn[mine]).detach().clone()
        for i in range(self.num_modalities):
            for pre in (f'encoders.{i}.1', f'encoders.{i}.5', f'decoders.{i}.1', f'decoders.{i}.5'):
                out[pre + '.num_batches_tracked'] = torch.tensor(self.num_batches_tracked)
        return out

    def load_state_dict(self, sd):
        names = self.layout.reference_names()
        for ref, (mine, sl) in names.items():
            src = torch.as_tensor(sd[ref]).to(self.device, torch.float32)
            dst = self.layout.unpad(mine, self.p[mine])
            (dst if sl is None else dst[sl]).copy_(src)
        for ref, mine in self.layout.reference_bn_names().items():
            if ref in sd:
                self.layout.unpad(mine, self.bn[mine]).copy_(torch.as_tensor(sd[ref]).to(self.device, torch.float32))
        for k, v in sd.items():
            if k.endswith('num_batches_tracked'):
                self.num_batches_tracked = int(v)
                break

    # ---- eval-mode forward paths (BN uses running statistics, dropout is the identity) ----
    def _dev(self, x):
        x = torch.as_tensor(x)
        return x.to(self.device, torch.float32).contiguous()

    EVAL_GEMM_CFG = None
    EVAL_X3 = True

    @staticmethod
    def _eval_cfg(n, nout, k):
        """fp32 GEMM tile for the eval path: many rows make these large, square-ish products, where the 128x128x32 tile
        (16 waves of 32x32) runs at 120-130 TFLOP/s against ~110 for the 64x64 default of the skinny training shapes
        (tools/bench_gemm_sq.py, tools/bench_infer.py); the class attribute EVAL_GEMM_CFG overrides (diagnostics)."""
        if edModelVar.EVAL_GEMM_CFG is not None:
            return int(edModelVar.EVAL_GEMM_CFG)
        # 21: the same fp32 products on the bf16 matrix pipe (gemm_f32.hip: three-piece cuts, fp32-level error, 256 x 128 tiles): 4.5
        # against 3.2 M cells/s through `transform` / `impute` at config 2 (profiles/r05_bench_infer_bf16x3.log; 20, its 128 x 128
        # form: 4.15-4.4); EVAL_X3 = False: the fp32 pipe
        return (21 if edModelVar.EVAL_X3 else 17) if (n >= 2048 and nout >= 512 and k >= 512) else -1

    def _lin_bn_act(self, i, x, lin, bn):
        W, b = self.p[f'm{i}.{lin}.W'], self.p[f'm{i}.{lin}.b']
        n, k = x.shape[0], x.shape[1]          # k = W.shape[1], or the real feature count of a padded first layer
        out = torch.empty(n, W.shape[0], device=self.device)
        pr = nv.gemm_problem(x, W, out, n, W.shape[0], k, k, W.stride(0), W.shape[0], bias=b, epi=nv.EPI_BN_EVAL,
                             aux=(self.bn[f'm{i}.{bn}.mean'], self.bn[f'm{i}.{bn}.var'],
                                  self.p[f'm{i}.{bn}.g'], self.p[f'm{i}.{bn}.b']),
                             slope=LRELU_SLOPE, eps=BN_EPS)
        nv.gemm([pr], nv.NT, self._eval_cfg(n, W.shape[0], k))
        return out

    def _linear(self, x, W, b):
        n, k = x.shape[0], x.shape[1]
        out = torch.empty(n, W.shape[0], device=self.device)
        nv.gemm([nv.gemm_problem(x, W, out, n, W.shape[0], k, k, W.stride(0), W.shape[0], bias=b)], nv.NT,
                self._eval_cfg(n, W.shape[0], k))
        return out

    def _encode_eval(self, i, x):
        x = self._dev(x)
        return self._lin_bn_act(i, self._lin_bn_act(i, x, 'enc0', 'bn0'), 'enc1', 'bn1')

    def _mu_eval(self, i, h):
        L = self.output_dim
        return self._linear(self._dev(h), self.p[f'm{i}.head.W'][:L], self.p[f'm{i}.head.b'][:L])

    def _decode_eval(self, i, z):
        h = self._lin_bn_act(i, self._dev(z), 'dec0', 'bn2')
        h = self._lin_bn_act(i, h, 'dec1', 'bn3')
        out = self._linear(h, self.p[f'm{i}.dec2.W'], self.p[f'm{i}.dec2.b'])
        return out if not self.layout.padded else out[:, :self.input_dim[i]]

    def embed(self, x, i, chunk=65536):
        """fc_mus[i](encoders[i](x)) streamed over row chunks (no N x N `corr`)."""
        x = torch.as_tensor(x)
        outs = []
        for s in range(0, x.shape[0], chunk):
            outs.append(self._mu_eval(i, self._encode_eval(i, x[s:s + chunk])))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    def impute(self, X, compose, chunk=65536):
        """Reference model.py:277-282 in eval mode: decoders[to](fc_mu[from](encoders[from](X)))."""
        from_mod, to_mod = compose
        X = torch.as_tensor(X)
        outs = []
        for s in range(0, X.shape[0], chunk):
            outs.append(self._decode_eval(to_mod, self.embed(X[s:s + chunk], from_mod)))
        return torch.cat(outs, 0) if len(outs) > 1 else outs[0]

    def __call__(self, *X, corr=None):
        """Eval-mode `forward` (reference model.py:264-275): returns (zs, combined, X_hat, mus, logvar_last).
        In eval mode zs == mus.  `corr` = None means identity.  Train-mode forward/backward is
        `TrainEngine.step` (fused with losses and the optimiser)."""
        if self.training:
            raise nv.JamieHipError('train-mode forward is TrainEngine.step(); call .eval() first')
        L = self.output_dim
        M = self.num_modalities
        hs = [self._encode_eval(i, X[i]) for i in range(M)]
        mus = [self._mu_eval(i, hs[i]) for i in range(M)]
        logvar = self._linear(hs[M - 1], self.p[f'm{M - 1}.head.W'][L:], self.p[f'm{M - 1}.head.b'][L:])
        sig = self.p['sigma']
        if corr is None:
            comb = [sum(sig[j] * mus[j] for j in range(M)) / sig.sum()] * M
        elif M != 2:
            raise NotImplementedError('a correspondence block is only defined for two modalities')
        else:
            corr = self._dev(corr)
            n0, n1 = corr.shape
            c1 = torch.empty(n0, L, device=self.device)          # corr z1 and corr^T z0 on the fp32 MFMA GEMM (NN / TN)
            c0 = torch.empty(n1, L, device=self.device)
            nv.gemm([nv.gemm_problem(corr, mus[1], c1, n0, L, n1, n1, L, L)], nv.NN)
            nv.gemm([nv.gemm_problem(corr, mus[0], c0, n1, L, n0, n1, L, L)], nv.TN)
            comb = [(sig[0] * mus[0] + sig[1] * c1) / (sig[0] + sig[1] * corr.sum(1, keepdim=True)),
                    (sig[1] * mus[1] + sig[0] * c0) / (sig[1] + sig[0] * corr.sum(0).reshape(-1, 1))]
        X_hat = [self._decode_eval(i, comb[i]) for i in range(M)]
        return mus, comb, X_hat, mus, logvar
