#!/usr/bin/env python
"""Generate golden fixtures under tests/golden/ by running the REFERENCE itself (imported from
/root/reference with the stub modules of tools/ref_stubs.py) on small seeded inputs.

Run in the build container only:  python tools/make_goldens.py
The fixtures are data (inputs + the reference's outputs); no reference source is copied.

Technique: a spy subclass of the reference's `edModelVar` is passed through the
reference's own `model_class=` seam (jamie/jamie.py:47,71).  It records the initial `state_dict()`,
and for every train-mode forward the torch RNG state, the batch and `corr`; parameter hooks record
the pre-clip gradients; `np.random.choice` is wrapped to record the sampler's index stream.  After the
run, the recorded RNG states are replayed with the same torch calls the reference's layers make
(`empty.bernoulli_(1-p)`, `empty.normal_()`) to materialise dropout masks and reparameterisation noise
explicitly, so tests can feed identical noise to the oracle and to the HIP path.
"""
import contextlib
import io
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_stubs  # noqa: E402

ref_stubs.install()
import matplotlib  # noqa: E402

matplotlib.use('Agg')
import jamie as ref  # noqa: E402  (the reference package)
from jamie.model import edModelVar as RefModel  # noqa: E402

OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
REC = {}


class Spy(RefModel):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        REC['init_state'] = {n: v.detach().clone() for n, v in self.state_dict().items()}
        REC['param_order'] = [n for n, _ in self.named_parameters()]
        for name, prm in self.named_parameters():
            prm.register_hook(lambda g, name=name: REC['grads'].setdefault(name, []).append(g.detach().clone()))

    def forward(self, *X, corr):
        if self.training:
            REC['rng'].append(torch.get_rng_state().clone())
            REC['X'].append([x.detach().clone() for x in X])
            REC['corr'].append(corr.detach().clone())
        out = super().forward(*X, corr=corr)
        if self.training:
            zs, comb, xh, mus, lv = out
            REC['fwd'].append({'zs': [t.detach().clone() for t in zs],
                               'combined': [t.detach().clone() for t in comb],
                               'mus': [t.detach().clone() for t in mus],
                               'logvar': lv.detach().clone()})
        return out


def replay_noise(rng_state, dims, L, B, p):
    """Same draws, same order, as Dropout/Normal.rsample inside edModelVar.forward (train mode)."""
    keep = torch.get_rng_state()
    torch.set_rng_state(rng_state)
    enc, dec, eps = [], [], []
    for d in dims:
        if p > 0:
            enc.append([torch.empty(B, 2 * d).bernoulli_(1 - p), torch.empty(B, d).bernoulli_(1 - p)])
    for d in dims:
        eps.append(torch.empty(B, L).normal_())
    for d in dims:
        if p > 0:
            dec.append([torch.empty(B, d).bernoulli_(1 - p), torch.empty(B, 2 * d).bernoulli_(1 - p)])
    torch.set_rng_state(keep)
    return enc, dec, eps


def synth(rng, N, dims, k=6):
    """SURVEY.md §8(d) generator at toy size: X_i = Z A_i + 0.1 E_i, fp32 values."""
    Z = rng.standard_normal((N, k))
    return [np.asarray(Z @ rng.standard_normal((k, d)) + 0.1 * rng.standard_normal((N, d)),
                       dtype=np.float32).astype(np.float64) for d in dims]


def run_case(name, rows, dims, L, B, epochs, np_seed, ctor=None, P=None, match_result=None,
             store_state=True, n_noise_steps=None, store_init=True):
    ctor = dict(ctor or {})
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
    if isinstance(rows, int):
        Z_rows = [rows, rows]
        data = synth(rng, rows, dims)
    else:
        Z_rows = list(rows)
        data = [synth(rng, r, [d])[0] for r, d in zip(rows, dims)]
    REC.clear()
    REC.update({'grads': {}, 'rng': [], 'X': [], 'corr': [], 'fwd': [], 'choice': []})
    kw = dict(output_dim=L, batch_size=B, epoch_DNN=epochs, pca_dim=None, model_class=Spy,
              use_f_tilde=match_result is not None, log_DNN=10 ** 9, manual_seed=666)
    if match_result is not None:
        kw['match_result'] = [np.asarray(match_result)]
    kw.update(ctor)
    orig_choice = np.random.choice

    def choice(*a, **k):
        r = orig_choice(*a, **k)
        REC['choice'].append(np.asarray(r).copy())
        return r
    np.random.seed(np_seed)
    np.random.choice = choice
    try:
        with contextlib.redirect_stdout(io.StringIO()):
            jm = ref.JAMIE(**kw)
            emb = jm.fit_transform(dataset=[d.copy() for d in data], P=None if P is None else P.copy())
    finally:
        np.random.choice = orig_choice
    with contextlib.redirect_stdout(io.StringIO()):
        tr = jm.transform([d.copy() for d in data])
        tr_one = [jm.transform_one(data[i].copy(), i) for i in range(2)]
        imp = [jm.modal_predict(data[i].copy(), i) for i in range(2)]
    p = jm.model.encoders[0][3].p
    Beff = int(jm.batch_size)
    steps = len(REC['rng'])
    nn = steps if n_noise_steps is None else min(n_noise_steps, steps)
    out = {}
    meta = {'name': name, 'rows': Z_rows, 'dims': list(dims), 'L': L, 'B': Beff, 'epochs': epochs,
            'np_seed': np_seed, 'p': float(p), 'steps': steps, 'noise_steps': nn,
            'ctor': {k: v for k, v in ctor.items()}, 'has_P': P is not None,
            'has_F': match_result is not None, 'sampling_method': jm.sampling_method,
            'param_order': REC['param_order'], 'loss_names': list(jm.loss_history.keys())}
    out['meta'] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    for i in range(2):
        out[f'data{i}'] = data[i].astype(np.float32)
        out[f'emb{i}'] = emb[i]
        out[f'transform{i}'] = tr[i]
        out[f'transform_one{i}'] = tr_one[i]
        out[f'impute_from{i}'] = imp[i]
    if P is not None:
        out['P'] = np.asarray(P, dtype=np.float32)
    if match_result is not None:
        out['F'] = np.asarray(match_result, dtype=np.float32)
    out['loss_history'] = np.array([jm.loss_history[k] for k in jm.loss_history], dtype=np.float64)
    # sampler stream: one entry per np.random.choice call
    out['choice'] = np.stack(REC['choice']).astype(np.int64)
    for s in range(nn):
        enc, dec, eps = replay_noise(REC['rng'][s], dims, L, Beff, p)
        for i in range(2):
            out[f's{s}.eps{i}'] = eps[i].numpy()
            if p > 0:
                for j in range(2):
                    out[f's{s}.encmask{i}{j}'] = np.packbits(enc[i][j].numpy().astype(np.uint8), axis=1)
                    out[f's{s}.decmask{i}{j}'] = np.packbits(dec[i][j].numpy().astype(np.uint8), axis=1)
    # first step internals
    out['s0.corr'] = REC['corr'][0].numpy()
    for i in range(2):
        out[f's0.X{i}'] = REC['X'][0][i].numpy()
        out[f's0.z{i}'] = REC['fwd'][0]['zs'][i].numpy()
        out[f's0.comb{i}'] = REC['fwd'][0]['combined'][i].numpy()
        out[f's0.mu{i}'] = REC['fwd'][0]['mus'][i].numpy()
    out['s0.logvar'] = REC['fwd'][0]['logvar'].numpy()
    if store_state:
        for n, v in REC['init_state'].items():
            if store_init:
                out['init.' + n] = v.numpy()
        # (mid-size fixture: the initial state is what torch.manual_seed(manual_seed) + the construction order give -- the
        #  oracle's init_state reproduces it bit for bit; stored as per-tensor float64 checksums instead of 3.5 MB of floats)
        out['init_names'] = np.array(list(REC['init_state'].keys()))
        out['init_checksum'] = np.array([[float(v.double().sum()), float(v.double().abs().sum())]
                                         for v in REC['init_state'].values()], dtype=np.float64)
        for n, v in jm.model.state_dict().items():
            out['final.' + n] = v.detach().numpy()
        for n, g in REC['grads'].items():
            out['grad0.' + n] = g[0].numpy()
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **out)
    print(f'{name}: steps={steps} p={p} method={jm.sampling_method} B={Beff} '
          f'losses(last)={[round(float(v[-1]), 6) for v in jm.loss_history.values()]} '
          f'-> {os.path.getsize(path) / 1024:.0f} KiB')


def main():
    os.makedirs(OUT, exist_ok=True)
    # G1: one step, dropout 0 (dims <= 64 -> default p = 0), identity corr (no replacement: min(d) >= B)
    run_case('g1_onestep_p0', 16, (24, 20), 4, 16, 1, 1)
    # G2: one step, DEFAULT dropout .6 (max(d) > 64), masks stored
    run_case('g2_onestep_p06', 32, (72, 66), 8, 32, 1, 2)
    # G3: 30 consecutive steps (5 epochs x 6 batches), explicit dropout .6, KL anneal varies
    run_case('g3_multistep', 96, (24, 20), 4, 16, 5, 3, ctor=dict(dropout=.6, min_epochs=4))
    # duplicates in the batch (min(d) < B -> replace=True) -> non-identity corr block
    run_case('g4_replace', 64, (24, 20), 4, 32, 2, 4)
    # dense F (match_result) mixed with P: general corr / F-loss path; loss weights
    rng = np.random.default_rng(7)
    Fm = np.abs(rng.standard_normal((48, 48))).astype(np.float32) * (rng.random((48, 48)) < .3)
    run_case('g5_F_pfratio', 48, (24, 20), 4, 24, 3, 5, ctor=dict(PF_Ratio=.5, loss_weights=[1, 2, 3, 4]),
             match_result=Fm)
    # unequal row counts -> P = 0 -> 'zeros' sampler, corr = 0
    run_case('g6_zeros', (40, 56), (24, 20), 4, 16, 2, 6)
    # cosine dist_method (jamie.py:484-494)
    run_case('g7_cosine', 32, (24, 20), 4, 16, 3, 8, ctor=dict(dist_method='cosine', dropout=.25))
    # KL-quirk witness: large lr so rows 0,1 of logvar_last move; more steps
    run_case('g8_klquirk', 20, (12, 10), 3, 20, 12, 9, ctor=dict(model_lr=5e-2, min_epochs=2))


def extra():
    """Fixtures added after the first eight (generated separately so that the first eight stay byte-identical)."""
    # G9: a run that STOPS EARLY (jamie.py:777-792): epoch_DNN = 60 but the streak of epochs without an improvement of
    # more than min_increment reaches max_steps_without_increment first; two batches per epoch, so `best_batch_loss`
    # (the minimum over the epoch's batches, jamie.py:729-731) differs from the last batch's loss that is recorded
    run_case('g9_earlystop', 48, (24, 20), 4, 24, 60, 10,
             ctor=dict(dropout=.3, min_epochs=6, min_increment=.25, max_steps_without_increment=3))


def mid():
    """G10 (round 3): a MID-SIZE run of the reference -- B = 128, (264, 200) features, latent 16, default dropout .6, three
    consecutive steps (one epoch of 384 cells) -- so that multi-tile / split-K launch paths of the HIP step are pinned to the
    reference's own numbers, not only to the oracle (every other fixture has d <= 72, B <= 32)."""
    run_case('g10_midsize', 384, (264, 200), 16, 128, 1, 11, store_init=False)


if __name__ == '__main__':
    if 'mid' in sys.argv[1:]:
        mid()
    elif 'extra' in sys.argv[1:]:
        extra()
    else:
        main()
