"""CPU oracle for JAMIE's coupled-VAE hot path.  TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module;
`jamie_amd/` never does (the product path fails loudly when the HIP library is missing).

This is a plain-PyTorch (CPU, fp32 or fp64) restatement of the algorithm in the reference
(`/root/reference/jamie`, v4.4.5).  The reference's arithmetic is dispatched to PyTorch ATen CPU
kernels; the restatement calls the same functional ops (`F.linear`, `F.batch_norm`, `F.leaky_relu`,
`torch.cdist`, ...) but takes every random draw (dropout masks, reparameterisation noise, batch
indices) as an explicit input so that the HIP path can be fed the identical noise.

Parity pin: `tests/test_oracle_golden.py` checks this module against golden vectors produced by
importing the reference itself in the build container (`tools/make_goldens.py`, fixtures under
`tests/golden/`).  The reference ships no tests of its own, so those goldens are the pin.

Every function cites the reference file:line it follows.
"""
from collections import OrderedDict
import math

import numpy as np
import torch
import torch.nn.functional as F

BN_EPS = 1e-5          # torch.nn.BatchNorm1d default (model.py:152)
BN_MOMENTUM = 0.1
LRELU_SLOPE = 0.01     # torch.nn.LeakyReLU default (model.py:153)
STD_GUARD = 1e-7       # model.py:238
KL_WEIGHT = 32 * 1e-3  # jamie.py:632
ALIGN_WEIGHT = 32      # jamie.py:658
CLIP_NORM = 1.0        # jamie.py:739
ADAM_BETAS = (0.9, 0.999)
ADAM_EPS = 1e-8


# --------------------------------------------------------------------------------------------
# model state
# --------------------------------------------------------------------------------------------
def default_dropout(input_dim, dropout=None):
    """model.py:144-145."""
    if dropout is None:
        return .6 if max(input_dim) > 64 else 0
    return dropout


def _linear_init(out_f, in_f, dtype):
    """torch.nn.Linear.reset_parameters: kaiming_uniform_(a=sqrt(5)) on W, U(-1/sqrt(fan_in), ..) on b.
    Consumes the global torch RNG exactly like `nn.Linear(in_f, out_f)` (model.py:151 etc.)."""
    gain = math.sqrt(2.0 / (1 + math.sqrt(5) ** 2))
    std = gain / math.sqrt(in_f)
    bound = math.sqrt(3.0) * std
    w = torch.empty(out_f, in_f).uniform_(-bound, bound)
    bb = 1 / math.sqrt(in_f) if in_f > 0 else 0
    b = torch.empty(out_f).uniform_(-bb, bb)
    return w.to(dtype), b.to(dtype)


def init_state(input_dim, output_dim, dtype=torch.float32):
    """Create parameters and BN buffers in the reference's construction order (model.py:147-220),
    drawing from the global torch RNG in the same sequence, so `torch.manual_seed(s); init_state(...)`
    reproduces `torch.manual_seed(s); edModelVar(...)` bit for bit.  Names are the reference's
    `state_dict()` keys."""
    M = len(input_dim)
    L = output_dim
    P = OrderedDict()
    Bf = OrderedDict()

    def bn(prefix, n):
        P[prefix + '.weight'] = torch.ones(n, dtype=dtype)
        P[prefix + '.bias'] = torch.zeros(n, dtype=dtype)
        Bf[prefix + '.running_mean'] = torch.zeros(n, dtype=dtype)
        Bf[prefix + '.running_var'] = torch.ones(n, dtype=dtype)
        Bf[prefix + '.num_batches_tracked'] = torch.zeros((), dtype=torch.long)

    for i, d in enumerate(input_dim):                       # model.py:147-171
        P[f'encoders.{i}.0.weight'], P[f'encoders.{i}.0.bias'] = _linear_init(2 * d, d, dtype)
        bn(f'encoders.{i}.1', 2 * d)
        P[f'encoders.{i}.4.weight'], P[f'encoders.{i}.4.bias'] = _linear_init(d, 2 * d, dtype)
        bn(f'encoders.{i}.5', d)
    for i, d in enumerate(input_dim):                       # model.py:178-181
        P[f'fc_mus.{i}.weight'], P[f'fc_mus.{i}.bias'] = _linear_init(L, d, dtype)
    for i, d in enumerate(input_dim):                       # model.py:183-186
        P[f'fc_vars.{i}.weight'], P[f'fc_vars.{i}.bias'] = _linear_init(L, d, dtype)
    for i, d in enumerate(input_dim):                       # model.py:188-216
        P[f'decoders.{i}.0.weight'], P[f'decoders.{i}.0.bias'] = _linear_init(d, L, dtype)
        bn(f'decoders.{i}.1', d)
        P[f'decoders.{i}.4.weight'], P[f'decoders.{i}.4.bias'] = _linear_init(2 * d, d, dtype)
        bn(f'decoders.{i}.5', 2 * d)
        P[f'decoders.{i}.8.weight'], P[f'decoders.{i}.8.bias'] = _linear_init(d, 2 * d, dtype)
    P['sigma'] = torch.rand(M).to(dtype)                    # model.py:220
    return P, Bf


def param_count(input_dim, output_dim):
    """P = sum_i(8 d_i^2 + 3 d_i L + 19 d_i + 2 L) + M."""
    L = output_dim
    return sum(8 * d * d + 3 * d * L + 19 * d + 2 * L for d in input_dim) + len(input_dim)


# BN layers whose preceding Linear bias has a mathematically zero gradient (SURVEY.md §7 "dead
# parameters"): Adam turns rounding noise into O(1e-4) drifts there, so weight-parity checks skip them.
def is_dead_bias(name):
    return name.endswith('.bias') and any(
        name.startswith(p) and name.split('.')[2] in ('0', '4')
        for p in ('encoders.', 'decoders.'))


# --------------------------------------------------------------------------------------------
# noise
# --------------------------------------------------------------------------------------------
def draw_noise(input_dim, output_dim, B, p, dtype=torch.float32):
    """Draw one training step's random numbers from the GLOBAL torch RNG in the order the reference
    consumes them (SURVEY.md §7 "RNG parity"; model.py:222-223 -> 225-243 -> 261-262):
    encoder dropout masks per modality (B x 2d, B x d), eps per modality (B x L), decoder dropout masks
    per modality (B x d, B x 2d).  `Dropout(0)` draws nothing (ATen dropout returns early).
    Masks are 0/1 (`empty.bernoulli_(1-p)`), eps is `empty.normal_()`."""
    noise = {'enc_masks': [], 'dec_masks': [], 'eps': []}
    for d in input_dim:
        if p > 0:
            noise['enc_masks'].append([torch.empty(B, 2 * d).bernoulli_(1 - p).to(dtype),
                                       torch.empty(B, d).bernoulli_(1 - p).to(dtype)])
        else:
            noise['enc_masks'].append([None, None])
    for d in input_dim:
        noise['eps'].append(torch.empty(B, output_dim).normal_().to(dtype))
    for d in input_dim:
        if p > 0:
            noise['dec_masks'].append([torch.empty(B, d).bernoulli_(1 - p).to(dtype),
                                       torch.empty(B, 2 * d).bernoulli_(1 - p).to(dtype)])
        else:
            noise['dec_masks'].append([None, None])
    return noise



# --------------------------------------------------------------------------------------------
# bf16-operand emulation (NOT in the reference, which is fp32 throughout: this restates what the HIP path's bf16 compute
# mode does to the SAME algorithm, so that the benchmarked arithmetic can be compared tightly instead of against the
# fp32 oracle at bf16 tolerances).  A Linear layer has three products -- forward y = a W^T, input gradient dx = dy W,
# weight gradient dW = dy^T a (autograd of model.py:151 etc. via jamie.py:734) -- and the HIP step rounds the OPERANDS of a
# product to bf16 (round to nearest even, once, where the producer stores them) and accumulates in fp32; everything else
# (bias add, BatchNorm, LeakyReLU, dropout, latent block, losses, bias gradients) stays fp32.  `emulate` maps a layer key
# ('enc0', 'enc1', 'head', 'dec0', 'dec1', 'dec2') to (fwd, dx, dw) booleans: True = that product reads bf16 operands.
# `TrainEngine.operand_precision()` reports the map of the launch plan under test.
# --------------------------------------------------------------------------------------------
def bf16_round(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _LinearEmulated(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, W, b, fwd, dx, dw):
        ctx.save_for_backward(a, W)
        ctx.flags = (dx, dw)
        if fwd:
            return F.linear(bf16_round(a), bf16_round(W), b)
        return F.linear(a, W, b)

    @staticmethod
    def backward(ctx, dy):
        a, W = ctx.saved_tensors
        dx, dw = ctx.flags
        dyb = bf16_round(dy) if (dx or dw) else None
        da = (dyb @ bf16_round(W)) if dx else dy @ W
        dW = (dyb.t() @ bf16_round(a)) if dw else dy.t() @ a
        return da, dW, dy.sum(0), None, None, None


def _layer_key(name):
    """Reference parameter prefix -> layer key: 'encoders.i.0' enc0, 'encoders.i.4' enc1, 'fc_mus.i' / 'fc_vars.i' head,
    'decoders.i.0' dec0, 'decoders.i.4' dec1, 'decoders.i.8' dec2."""
    parts = name.split('.')
    if parts[0] in ('fc_mus', 'fc_vars'):
        return 'head'
    return {'encoders': 'enc', 'decoders': 'dec'}[parts[0]] + {'0': '0', '4': '1', '8': '2'}[parts[2]]


def linear(P, name, h, emulate=None):
    """`F.linear(h, P[name.weight], P[name.bias])`, or its bf16-operand emulation when `emulate` names the layer."""
    W, b = P[name + '.weight'], P[name + '.bias']
    flags = None if emulate is None else emulate.get(_layer_key(name))
    if flags is None or not any(flags):
        return F.linear(h, W, b)
    return _LinearEmulated.apply(h, W, b, bool(flags[0]), bool(flags[1]), bool(flags[2]))


# What the fused bf16 launch plan of the HIP step does at every BASELINE size (identity correspondence, F = 0): decoder layer 0's
# forward product and the heads' input gradient are exact fp32 inside the fused latent kernels, every other product is bf16.
EMULATE_HIP_BF16_FUSED = {'enc0': (True, True, True), 'enc1': (True, True, True), 'head': (True, False, True),
                          'dec0': (False, True, True), 'dec1': (True, True, True), 'dec2': (True, True, True)}

# --------------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------------
def _block(h, P, Bf, lin, bn, train, p, mask, emulate=None):
    """Linear -> BatchNorm1d -> LeakyReLU -> Dropout (model.py:151-154 and siblings)."""
    h = linear(P, lin, h, emulate)
    h = F.batch_norm(h, Bf[bn + '.running_mean'], Bf[bn + '.running_var'], P[bn + '.weight'],
                     P[bn + '.bias'], training=train, momentum=BN_MOMENTUM, eps=BN_EPS)
    if train:
        Bf[bn + '.num_batches_tracked'] += 1
    h = F.leaky_relu(h, LRELU_SLOPE)
    if train and p > 0:
        # ATen CPU dropout: noise.bernoulli_(1-p); noise.div_(1-p); input * noise
        h = h * (mask / (1 - p))
    return h


def encode(P, Bf, i, x, train=False, p=0., masks=(None, None), emulate=None):
    """model.py:222-223 (Sequential at :149-165)."""
    h = _block(x, P, Bf, f'encoders.{i}.0', f'encoders.{i}.1', train, p, masks[0], emulate)
    return _block(h, P, Bf, f'encoders.{i}.4', f'encoders.{i}.5', train, p, masks[1], emulate)


def decode(P, Bf, i, c, train=False, p=0., masks=(None, None), emulate=None):
    """model.py:261-262 (Sequential at :190-207)."""
    h = _block(c, P, Bf, f'decoders.{i}.0', f'decoders.{i}.1', train, p, masks[0], emulate)
    h = _block(h, P, Bf, f'decoders.{i}.4', f'decoders.{i}.5', train, p, masks[1], emulate)
    return linear(P, f'decoders.{i}.8', h, emulate)


def refactor(P, hs, train, eps, index=None, emulate=None):
    """model.py:225-243.  NB returns `logvar` of the LAST modality only (:243)."""
    if index is None:
        index = range(len(hs))
    zs, mus = [], []
    logvar = None
    for h, i in zip(hs, index):
        mu = linear(P, f'fc_mus.{i}', h, emulate)
        logvar = linear(P, f'fc_vars.{i}', h, emulate)
        std = torch.exp(logvar / 2)
        if not train:
            zs.append(mu)
        else:
            std = std + STD_GUARD
            zs.append(mu + eps[i] * std)       # Normal(mu, std).rsample()
        mus.append(mu)
    return zs, mus, logvar


def combine_identity(P, zs):
    """BUILD-DEFINED generalisation to M >= 2 fully paired modalities (no counterpart in the reference, which
    asserts two modalities: jamie.py:420): comb_i = sum_j sigma_j z_j / sum_j sigma_j for every i.  For M = 2 this
    is model.py:245-259 at corr = I."""
    sigma = P['sigma']
    c = sum(sigma[j] * zs[j] for j in range(len(zs))) / sigma[:len(zs)].sum()
    return [c for _ in zs]


def combine(P, zs, corr):
    """model.py:245-259 (two modalities; `(i + 1) % 2`).  `corr=None` = identity, any number of modalities."""
    if corr is None:
        return combine_identity(P, zs)
    assert len(zs) == 2, 'a correspondence block is only defined for two modalities (reference jamie.py:420)'
    sigma = P['sigma']
    out = []
    for i in range(2):
        j = (i + 1) % 2
        out.append((sigma[i] * zs[i] + sigma[j] * torch.mm(corr if i == 0 else corr.t(), zs[j]))
                   / (sigma[i] + sigma[j] * corr.sum(j).reshape(-1, 1)))
    return out


def forward(P, Bf, X, corr, train=False, p=0., noise=None, emulate=None):
    """edModelVar.forward, model.py:264-275."""
    M = len(X)
    enc_masks = noise['enc_masks'] if (train and noise is not None) else [(None, None)] * M
    dec_masks = noise['dec_masks'] if (train and noise is not None) else [(None, None)] * M
    eps = noise['eps'] if (train and noise is not None) else None
    hs = [encode(P, Bf, i, X[i], train, p, enc_masks[i], emulate) for i in range(M)]
    zs, mus, logvar = refactor(P, hs, train, eps, emulate=emulate)
    combined = combine(P, zs, corr)
    X_hat = [decode(P, Bf, i, combined[i], train, p, dec_masks[i], emulate) for i in range(M)]
    return zs, combined, X_hat, mus, logvar


def impute(P, Bf, x, from_mod, to_mod):
    """edModelVar.impute in eval mode, model.py:277-282."""
    h = encode(P, Bf, from_mod, x)
    z = refactor(P, [h], False, None, [from_mod])[0][0]
    return decode(P, Bf, to_mod, z)


def transform_one(P, Bf, x, i):
    """jamie.py:831-837: fc_mus[i](encoders[i](x)) in eval mode."""
    h = encode(P, Bf, i, x)
    return F.linear(h, P[f'fc_mus.{i}.weight'], P[f'fc_mus.{i}.bias'])


# --------------------------------------------------------------------------------------------
# losses (jamie.py:614-668)
# --------------------------------------------------------------------------------------------
def kl_anneal(epoch, min_epochs, epoch_DNN):
    """jamie.py:630-631 (numpy float64 arithmetic)."""
    c = (min_epochs / 2) if min_epochs > 0 else (epoch_DNN / 2)
    return 1 / (1 + np.exp(-5 * (epoch - c) / c))


def sim_diff(a, b, dist_method='euclidean'):
    """sim_diff_func, jamie.py:483-502.  Returns `diff` only (`sim` is never used by a loss)."""
    if dist_method == 'cosine':
        sim = torch.mm(a, b.t()) / (a.norm(dim=1).reshape(-1, 1) * b.norm(dim=1).reshape(1, -1))
        return 1 - sim
    return torch.cdist(a, b, p=2)


def losses(X, zs, combined, X_hat, mus, logvars, Fblk, anneal, dist_method='euclidean'):
    """The four losses in the reference's order: KL, Rec, CosSim, F (jamie.py:618-668).
    `logvars` is the last modality's [B, L] tensor, so `logvars[i]` is ROW i (the reference's quirk,
    SURVEY.md §7 "Bug-compatible KL")."""
    M = len(X)
    kl = sum(-.5 * torch.mean(1 + logvars[i] - mus[i].square() - logvars[i].exp(), axis=1).mean(axis=0)
             for i in range(M))
    l_kl = KL_WEIGHT * anneal * kl
    l_rec = sum((X_hat[i] - X[i]).square().mean(axis=1).mean(axis=0) for i in range(M))
    cos = sum(torch.diag(sim_diff(zs[i], combined[i], dist_method).square()).mean(axis=0) / zs[i].shape[1]
              for i in range(M))                       # jamie.py:649-657 (two terms there)
    l_cos = ALIGN_WEIGHT * cos
    if Fblk is None:                                   # F = 0 (use_f_tilde=False): mean(combined[0]^2)
        l_f = torch.square(combined[0]).mean(axis=1).mean(axis=0)
    else:
        l_f = torch.square(combined[0] - torch.mm(Fblk, combined[1])).mean(axis=1).mean(axis=0)
    return [l_kl, l_rec, l_cos, l_f]


LOSS_NAMES = ['KL', 'Rec', 'CosSim', 'F']


# --------------------------------------------------------------------------------------------
# batch blocks of P and F (jamie.py:585-604)
# --------------------------------------------------------------------------------------------
def row_normalise(blk):
    """jamie.py:587-589 / :593-595: divide rows by their sum, zero-sum rows by 1."""
    s = blk.sum(axis=1)
    s = torch.where(s == 0, torch.ones_like(s), s)
    return blk / s[:, None]


def p_block(Pmat, idx0, idx1, dtype=torch.float32):
    """jamie.py:586-589.  `Pmat=None` means P = I_N, for which P[idx0][:, idx1] is the index-equality
    matrix (no N x N array is formed)."""
    if Pmat is None:
        blk = (torch.as_tensor(idx0)[:, None] == torch.as_tensor(idx1)[None, :]).to(dtype)
    else:
        blk = Pmat[idx0][:, idx1]
    return row_normalise(blk)


def f_block(Fmat, idx0, idx1, B, dtype=torch.float32):
    """jamie.py:592-595.  `Fmat=None` means F = 0 (use_f_tilde=False, jamie.py:173)."""
    if Fmat is None:
        return torch.zeros(B, B, dtype=dtype)
    return row_normalise(Fmat[idx0][:, idx1])


# --------------------------------------------------------------------------------------------
# optimiser (jamie.py:481, 739-741)
# --------------------------------------------------------------------------------------------
def clip_grad_norm(grads, max_norm=CLIP_NORM):
    """torch.nn.utils.clip_grad_norm_(params, 1): global L2 norm, coef = max_norm / (norm + 1e-6),
    clamped to 1.  Returns the total norm."""
    norms = torch._foreach_norm(grads, 2.0)
    total = torch.linalg.vector_norm(torch.stack(norms), 2.0)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    torch._foreach_mul_(grads, coef)
    return total


class Adam:
    """torch.optim.Adam(lr, betas=(.9,.999), eps=1e-8, weight_decay=0, amsgrad=False), restated."""

    def __init__(self, params, lr):
        self.params = list(params)
        self.lr = lr
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = 0

    def step(self, grads):
        b1, b2 = ADAM_BETAS
        self.t += 1
        torch._foreach_lerp_(self.m, grads, 1 - b1)
        torch._foreach_mul_(self.v, b2)
        torch._foreach_addcmul_(self.v, grads, grads, 1 - b2)
        bc1 = 1 - b1 ** self.t
        bc2 = 1 - b2 ** self.t
        step_size = self.lr / bc1
        denom = torch._foreach_sqrt(self.v)
        torch._foreach_div_(denom, math.sqrt(bc2))
        torch._foreach_add_(denom, ADAM_EPS)
        torch._foreach_addcdiv_(self.params, self.m, denom, -step_size)


# --------------------------------------------------------------------------------------------
# one training step and the full loop
# --------------------------------------------------------------------------------------------
def train_step(P, Bf, opt, X, corr, Fblk, noise, p, anneal, loss_weights=None,
               dist_method='euclidean', do_step=True, return_grads=False, emulate=None, grad_bf16=False):
    """One iteration of the inner loop of project_jamie (jamie.py:611-741): forward, four losses,
    backward, clip, Adam.  `P` values must be leaf tensors with requires_grad=True.
    `emulate` (see `linear`): bf16-operand emulation of the HIP path's bf16 compute mode; `grad_bf16`: every gradient is
    rounded to bf16 once before the optimiser reads it while the clip norm is taken from the unrounded values (the HIP
    path's bf16 weight-gradient buffer)."""
    zs, comb, X_hat, mus, logvars = forward(P, Bf, X, corr, train=True, p=p, noise=noise, emulate=emulate)
    ls = losses(X, zs, comb, X_hat, mus, logvars, Fblk, anneal, dist_method)
    if loss_weights is not None:
        total = sum(lo * wt for lo, wt in zip(ls, loss_weights))
    else:
        total = sum(ls)
    names = list(P.keys())
    params = [P[k] for k in names]
    grads = list(torch.autograd.grad(total, params, allow_unused=True))
    grads = [g if g is not None else torch.zeros_like(q) for g, q in zip(grads, params)]
    out = {'losses': [float(l.detach()) for l in ls], 'total': float(total.detach()),
           'zs': [z.detach() for z in zs], 'combined': [c.detach() for c in comb],
           'mus': [m.detach() for m in mus], 'logvar': logvars.detach(),
           'X_hat': [x.detach() for x in X_hat]}
    if return_grads:
        out['grads'] = OrderedDict((k, g.clone()) for k, g in zip(names, grads))
    if do_step:
        with torch.no_grad():
            if grad_bf16:
                total_norm = torch.linalg.vector_norm(torch.stack(torch._foreach_norm(grads, 2.0)), 2.0)
                coef = torch.clamp(CLIP_NORM / (total_norm + 1e-6), max=1.0)
                grads = [bf16_round(g) * coef for g in grads]
                out['grad_norm'] = float(total_norm)
            else:
                out['grad_norm'] = float(clip_grad_norm(grads))
            opt.step(grads)
    return out


class Preclass:
    """utilities.py:654-678 (`preclass`): standardise with the statistics of the fitting sample
    (`axis=0` per feature; `axis=None` one global mean/std after PCA), NaN -> 0; and the inverse."""

    def __init__(self, sample, pca=None, axis=None):
        self.mean = np.asarray(sample.mean(axis))
        self.std = np.asarray(sample.std(axis))
        self.pca = pca

    def transform(self, X):
        out = X
        if self.pca is not None:
            out = self.pca.transform(out)
        out = out - self.mean
        with np.errstate(all='ignore'):
            out = out / self.std
        out[np.isnan(out)] = 0
        return out

    def inverse_transform(self, X):
        out = X * self.std + self.mean
        if self.pca is not None:
            out = self.pca.inverse_transform(out)
        return out


def sample_batch(rows, cols, batch_size, sampling_method):
    """jamie.py:552-579 for the 'diag' and 'zeros' samplers (global numpy RNG, like the reference)."""
    rep = min(cols) < batch_size                                    # jamie.py:553 (sic)
    if sampling_method == 'diag':
        s = np.random.choice(range(rows[0]), batch_size, replace=rep)
        return [s for _ in rows]
    if sampling_method == 'zeros':
        return [np.random.choice(range(r), batch_size, replace=rep) for r in rows]
    raise Exception(f'Sampling method {sampling_method} does not exist')


class OracleJAMIE:
    """Restatement of JAMIE.fit_transform -> project_jamie for `project_mode='jamie'`,
    `pca_dim=None`, with `P` either None (identity, 'diag' sampling) or a dense array and the
    correspondence `F` either absent (`use_f_tilde=False`) or given as `match_result`
    (jamie.py:113-222, 416-804).  Stages A/B (distances, Prime_Dual) are out of scope."""

    def __init__(self, output_dim=32, batch_size=512, epoch_DNN=10000, model_lr=1e-3, dropout=None,
                 PF_Ratio=None, loss_weights=None, dist_method='euclidean', min_epochs=2500,
                 min_increment=1e-8, max_steps_without_increment=500, use_early_stop=True,
                 manual_seed=666, match_result=None, dtype=torch.float32):
        self.output_dim = output_dim
        self.batch_size = batch_size
        self.epoch_DNN = epoch_DNN
        self.model_lr = model_lr
        self.dropout = dropout
        self.PF_Ratio = PF_Ratio
        self.loss_weights = loss_weights
        self.dist_method = dist_method
        self.min_epochs = min_epochs
        self.min_increment = min_increment
        self.max_steps_without_increment = max_steps_without_increment
        self.use_early_stop = use_early_stop
        self.manual_seed = manual_seed
        self.match_result = match_result
        self.dtype = dtype
        self.trace = None

    def fit_transform(self, dataset, P=None, record_trace=False):
        dt = self.dtype
        torch.manual_seed(self.manual_seed)                                   # jamie.py:142
        dataset = [np.asarray(d) for d in dataset]
        rows = [d.shape[0] for d in dataset]
        # --- project_jamie ---
        if P is None and rows[0] != rows[1]:                                  # jamie.py:423-428
            P = np.zeros((rows[0], rows[1]))
        Pm = None if P is None else torch.Tensor(np.asarray(P)).to(dt)
        Fm = None if self.match_result is None else torch.Tensor(np.asarray(self.match_result[0])).to(dt)
        self.pre = [Preclass(d, axis=0) for d in dataset]                     # jamie.py:462-465
        data = [torch.from_numpy(pc.transform(d)).float().to(dt) for pc, d in zip(self.pre, dataset)]
        cols = [d.shape[1] for d in data]
        self.P_, self.Bf = init_state(cols, self.output_dim, dt)               # jamie.py:472-479
        for v in self.P_.values():
            v.requires_grad_(True)
        # the reference's optimiser sees model.parameters() order (sigma first); order is irrelevant
        # to Adam's arithmetic, and clip's norm-of-norms differs only in rounding.
        names = ['sigma'] + [k for k in self.P_ if k != 'sigma']
        self.P_ = OrderedDict((k, self.P_[k]) for k in names)
        opt = Adam(self.P_.values(), self.model_lr)                           # jamie.py:481
        p = default_dropout(cols, self.dropout)
        B = self.batch_size
        len_dl = int(np.max(rows) / B)                                         # jamie.py:511-514
        if len_dl == 0:
            len_dl = 1
            B = int(np.max(rows))
        PF = 1 if self.PF_Ratio is None else self.PF_Ratio                    # jamie.py:517
        if Pm is None or (Pm.shape[0] == Pm.shape[1]
                          and torch.abs(Pm - torch.eye(rows[0], dtype=dt)).sum() == 0):
            method = 'diag'                                                    # jamie.py:518-519
        elif torch.abs(Pm).sum() != 0:
            raise NotImplementedError("'hybrid' sampling (jamie.py:523-530) is out of scope")
        else:
            method = 'zeros'
        best_running = np.inf
        streak = 0
        self.loss_history = {n: [] for n in LOSS_NAMES}
        if record_trace:
            self.trace = []
        for epoch in range(self.epoch_DNN):                                   # jamie.py:546
            best_batch = np.inf
            for _ in range(len_dl):
                idx = sample_batch(rows, cols, B, method)                      # jamie.py:552-583
                X = [data[i][idx[i]] for i in range(2)]
                corr = PF * p_block(Pm, idx[0], idx[1], dt) + (1 - PF) * f_block(Fm, idx[0], idx[1], B, dt)
                Fblk = f_block(Fm, idx[0], idx[1], B, dt)
                noise = draw_noise(cols, self.output_dim, B, p, dt)
                anneal = kl_anneal(epoch, self.min_epochs, self.epoch_DNN)
                st = train_step(self.P_, self.Bf, opt, X, corr, Fblk, noise, p, anneal,
                                self.loss_weights, self.dist_method)
                if record_trace:
                    self.trace.append({'idx': [np.asarray(i).copy() for i in idx], 'noise': noise,
                                       'anneal': float(anneal), 'losses': st['losses'],
                                       'corr': corr, 'F': Fblk})
                if st['total'] < best_batch:
                    best_batch = st['total']
            w = self.loss_weights if self.loss_weights is not None else [1, 1, 1, 1]
            for n, lo, wt in zip(LOSS_NAMES, st['losses'], w):               # jamie.py:752-761
                self.loss_history[n].append(lo * wt)
            if epoch > self.min_epochs:                                       # jamie.py:777-792
                if best_running - best_batch > self.min_increment:
                    best_running = best_batch
                    streak = 0
                else:
                    streak += 1
                if streak >= self.max_steps_without_increment and self.use_early_stop:
                    break
        with torch.no_grad():                                                 # jamie.py:794-799
            out = [transform_one(self.P_, self.Bf, data[i], i).numpy() for i in range(2)]
        return out

    def transform(self, dataset):
        """jamie.py:817-829: output [0] of an eval forward is `mus` (corr is irrelevant)."""
        with torch.no_grad():
            return [transform_one(self.P_, self.Bf,
                                  torch.tensor(self.pre[i].transform(np.asarray(dataset[i]))).float().to(self.dtype),
                                  i).numpy() for i in range(len(dataset))]

    def modal_predict(self, data, modality):
        """jamie.py:806-815 (returns float64 because the inverse scaling promotes)."""
        to = (modality + 1) % 2
        with torch.no_grad():
            x = torch.tensor(self.pre[modality].transform(np.asarray(data))).float().to(self.dtype)
            dec = impute(self.P_, self.Bf, x, modality, to)
        return np.array(self.pre[to].inverse_transform(dec.numpy()))


# --------------------------------------------------------------------------------------------
# correspondence stage: Prime_Dual and the distance matrices that feed it
# --------------------------------------------------------------------------------------------
def prime_dual(Kx, Ky, dx, dy, epoch_pd=2000, rho=10, epsilon=1e-3, delay=0, dtype=torch.float32, history=None):
    """jamie.py:314-414 (`integration_type == 'MultiOmics'`), all seven products per iteration as the reference
    writes them.  Kx [m,m], Ky [n,n] distance matrices (numpy); returns F [m,n] float32 numpy.
    `history` (list) receives the scaling factor `a` of every iteration."""
    Kx, Ky = np.asarray(Kx), np.asarray(Ky)
    if Kx.shape == (1, 1) and Ky.shape == (1, 1):                              # jamie.py:326-328
        return np.ones((1, 1), np.float32)
    N = int(np.maximum(Kx.shape[0], Ky.shape[0]))                              # jamie.py:330-334
    Kx = torch.from_numpy(Kx / N).float().to(dtype)
    Ky = torch.from_numpy(Ky / N).float().to(dtype)
    a = np.sqrt(dy / dx)                                                       # jamie.py:335
    m, n = Kx.shape[0], Ky.shape[0]
    Fm = torch.zeros(m, n, dtype=dtype)                                        # jamie.py:339-345
    Im, In, Inn = torch.ones(m, 1, dtype=dtype), torch.ones(n, 1, dtype=dtype), torch.ones(n, n, dtype=dtype)
    Lambda, Mu, S = torch.zeros(n, 1, dtype=dtype), torch.zeros(m, 1, dtype=dtype), torch.zeros(n, 1, dtype=dtype)
    pho1, pho2, delta = 0.9, 0.999, 10e-8                                      # jamie.py:347-349
    m1, m2 = torch.zeros(m, n, dtype=dtype), torch.zeros(m, n, dtype=dtype)
    i = 0
    while i < epoch_pd:
        FKy = Fm @ Ky                                                          # jamie.py:357-373
        grad = (4 * FKy @ (Fm.t() @ FKy) - 4 * a * (Kx @ FKy) + Mu @ In.t() + Im @ Lambda.t()
                + rho * (Fm @ Inn + Im @ (Im.t() @ Fm + (S - 2 * In).t())))
        i += 1
        m1 = pho1 * m1 + (1 - pho1) * grad                                     # jamie.py:375-381
        m2 = pho2 * m2 + (1 - pho2) * grad * grad
        step = (m1 / (1 - np.power(pho1, i))) / (torch.sqrt(m2 / (1 - np.power(pho2, i))) + delta)
        F_tmp = Fm - step
        F_tmp[F_tmp < 0] = 0
        Fm = (1 - epsilon) * Fm + epsilon * F_tmp                              # jamie.py:384
        grad_s = Lambda + rho * (Fm.t() @ Im - In + S)                         # jamie.py:387-390
        s_tmp = S - grad_s
        s_tmp[s_tmp < 0] = 0
        S = (1 - epsilon) * S + epsilon * s_tmp
        Mu = Mu + epsilon * (Fm @ In - Im)                                     # jamie.py:393-394
        Lambda = Lambda + epsilon * (Fm.t() @ Im - In + S)
        if i >= delay:                                                         # jamie.py:397-402
            a = torch.trace(Kx @ ((Fm @ Ky) @ Fm.t())) / torch.trace(Kx @ Kx)
        if history is not None:
            history.append(float(a))
    return Fm.float().numpy()


def distance_matrix(X, mode):
    """jamie.py:839-890 for the modes whose arithmetic is in the reference file or in scipy / sklearn
    ('geodesic' is unioncom's, absent from /root/reference)."""
    from scipy import stats
    from sklearn.metrics import pairwise_distances
    if mode == 'spearman':                                                     # jamie.py:857-869
        if X.shape[0] == 1:
            return np.array([0])
        d, _ = stats.spearmanr(X, axis=1)
        if np.isnan(d).any():
            raise Exception('Data is not well conditioned for spearman method (scipy.stats.spearmanr returned ``np.nan``)')
        if len(np.shape(d)) == 0:
            d = np.array([[1, d], [d, 1]])
        return (1 - np.array(d)) / 2
    if mode == 'pearson':                                                      # jamie.py:870-879 (dense input)
        if X.shape[0] == 1:
            return np.array([0])
        d = np.corrcoef(np.asarray(X))
        if len(np.shape(d)) == 0:
            d = np.array([[1, d], [d, 1]])
        return (1 - np.array(d)) / 2
    return pairwise_distances(X, metric=mode)                                  # jamie.py:880-882
