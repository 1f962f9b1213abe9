"""Pin the CPU oracle (oracle/jamie_oracle.py) against golden vectors produced by the reference itself
(tools/make_goldens.py).  CPU-only."""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import jamie_oracle as orc
from golden_util import CASES, Golden


def _ctor(meta):
    c = dict(meta['ctor'])
    kw = dict(output_dim=meta['L'], batch_size=meta['B'] if meta['B'] <= max(meta['rows']) else meta['B'],
              epoch_DNN=meta['epochs'])
    for k in ('dropout', 'PF_Ratio', 'loss_weights', 'dist_method', 'min_epochs', 'model_lr', 'min_increment',
              'max_steps_without_increment'):
        if k in c:
            kw[k] = c[k]
    return kw


@pytest.mark.parametrize('name', CASES)
def test_init_matches_reference(name):
    g = Golden(name)
    torch.manual_seed(666)
    P, Bf = orc.init_state(g.meta['dims'], g.meta['L'])
    ref = g.state('init')
    assert orc.param_count(g.meta['dims'], g.meta['L']) == sum(v.numel() for v in P.values())
    for k, v in P.items():
        assert torch.equal(v, ref[k]), k
    for k, v in Bf.items():
        assert torch.equal(v, ref[k]), k


@pytest.mark.parametrize('name', CASES)
def test_full_loop_replay(name):
    """Same seeds -> the restated loop (sampler, noise order, losses, clip, Adam, early-stop bookkeeping,
    final eval embedding, transform, modal_predict) reproduces the reference run."""
    g = Golden(name)
    m = g.meta
    kw = _ctor(m)
    if m['has_F']:
        kw['match_result'] = [g['F']]
    o = orc.OracleJAMIE(**kw)
    np.random.seed(m['np_seed'])
    if m['rows'][0] != m['rows'][1]:
        data = [g['data0'].astype(np.float64), g['data1'].astype(np.float64)]
    else:
        data = g.data()
    emb = o.fit_transform(data, P=g['P'] if m['has_P'] else None, record_trace=True)
    # sampler stream
    ch = g['choice']
    flat = [i for t in o.trace for i in (t['idx'][:1] if m['sampling_method'] == 'diag' else t['idx'])]
    assert len(flat) == len(ch)
    for a, b in zip(flat, ch):
        assert np.array_equal(a, b)
    # noise stream
    for s in range(m['noise_steps']):
        gn = g.noise(s)
        for i in range(2):
            assert torch.equal(o.trace[s]['noise']['eps'][i], gn['eps'][i])
            if m['p'] > 0:
                for j in range(2):
                    assert torch.equal(o.trace[s]['noise']['enc_masks'][i][j], gn['enc_masks'][i][j])
                    assert torch.equal(o.trace[s]['noise']['dec_masks'][i][j], gn['dec_masks'][i][j])
    assert torch.allclose(o.trace[0]['corr'], torch.from_numpy(g['s0.corr']), atol=0, rtol=0)
    lh = np.array([o.loss_history[k] for k in m['loss_names']])
    assert lh.shape == g['loss_history'].shape           # same number of epochs (g9: the reference stopped early)
    assert m['steps'] == len(o.trace)
    np.testing.assert_allclose(lh, g['loss_history'], rtol=2e-5, atol=1e-7)
    for i in range(2):
        np.testing.assert_allclose(emb[i], g[f'emb{i}'], rtol=1e-4, atol=1e-5)
    tr = o.transform(data)
    for i in range(2):
        np.testing.assert_allclose(tr[i], g[f'transform{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(tr[i], g[f'transform_one{i}'], rtol=1e-4, atol=1e-5)
        imp = o.modal_predict(data[i], i)
        assert imp.dtype == np.float64
        np.testing.assert_allclose(imp, g[f'impute_from{i}'], rtol=1e-4, atol=1e-5)
    # final weights (dead pre-BN biases excluded: Adam amplifies rounding noise there)
    fin = g.state('final')
    for k, v in o.P_.items():
        if orc.is_dead_bias(k):
            continue
        np.testing.assert_allclose(v.detach().numpy(), fin[k].numpy(), rtol=1e-4, atol=2e-6, err_msg=k)
    for k, v in o.Bf.items():
        np.testing.assert_allclose(v.numpy(), fin[k].numpy(), rtol=1e-5, atol=1e-6, err_msg=k)


@pytest.mark.parametrize('name', CASES)
def test_first_step_explicit_noise(name):
    """One step driven by the fixture's explicit noise/indices: internals and pre-clip gradients."""
    g = Golden(name)
    m = g.meta
    P = OrderedDict((k, v.clone().requires_grad_(True)) for k, v in g.state('init').items()
                    if not ('running' in k or 'num_batches' in k))
    Bf = OrderedDict((k, v.clone()) for k, v in g.state('init').items()
                     if ('running' in k or 'num_batches' in k))
    X = [torch.from_numpy(g[f's0.X{i}']) for i in range(2)]
    idx = g.step_indices(0)
    # the batch the reference fed to the model == standardised data rows at the recorded indices
    for i in range(2):
        d = g[f'data{i}'].astype(np.float64)
        pc = orc.Preclass(d, axis=0)
        np.testing.assert_array_equal(torch.from_numpy(pc.transform(d)).float().numpy()[idx[i]], X[i].numpy())
    corr = torch.from_numpy(g['s0.corr'])
    PF = m['ctor'].get('PF_Ratio') or 1
    Fm = torch.from_numpy(g['F']) if m['has_F'] else None
    Pm = torch.zeros(m['rows'][0], m['rows'][1]) if m['sampling_method'] == 'zeros' else None
    Fblk = orc.f_block(Fm, idx[0], idx[1], m['B'])
    corr2 = PF * orc.p_block(Pm, idx[0], idx[1]) + (1 - PF) * Fblk
    assert torch.equal(corr2, corr)
    c = m['ctor']
    anneal = orc.kl_anneal(0, c.get('min_epochs', 2500), m['epochs'])
    st = orc.train_step(P, Bf, None, X, corr, Fblk, g.noise(0), m['p'], anneal,
                        c.get('loss_weights'), c.get('dist_method', 'euclidean'),
                        do_step=False, return_grads=True)
    for i in range(2):
        np.testing.assert_allclose(st['zs'][i].numpy(), g[f's0.z{i}'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st['combined'][i].numpy(), g[f's0.comb{i}'], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(st['mus'][i].numpy(), g[f's0.mu{i}'], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(st['logvar'].numpy(), g['s0.logvar'], rtol=1e-5, atol=1e-6)
    for k, gr in st['grads'].items():
        ref = g['grad0.' + k] if ('grad0.' + k) in g else np.zeros_like(gr.numpy())
        if orc.is_dead_bias(k):
            assert np.abs(gr.numpy()).max() < 1e-5
            continue
        np.testing.assert_allclose(gr.numpy(), ref, rtol=2e-4, atol=2e-7, err_msg=k)


def test_kl_quirk_gradient_rows():
    """fc_vars of modality 0 gets no KL gradient; the last modality's only via batch rows 0 and 1
    (SURVEY.md §7).  Witness: with eps = 0 the reparameterisation carries no logvar gradient, so the
    fc_vars gradients come from the KL term alone."""
    torch.manual_seed(3)
    dims, L, B = (6, 5), 3, 8
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    X = [torch.randn(B, d) for d in dims]
    noise = {'enc_masks': [(None, None)] * 2, 'dec_masks': [(None, None)] * 2,
             'eps': [torch.zeros(B, L), torch.zeros(B, L)]}
    st = orc.train_step(P, Bf, None, X, torch.eye(B), torch.zeros(B, B), noise, 0., 1.0,
                        do_step=False, return_grads=True)
    assert st['grads']['fc_vars.0.weight'].abs().max() == 0
    assert st['grads']['fc_vars.1.weight'].abs().max() > 0
    # rows >= 2 do not contribute: perturbing them leaves the gradient unchanged
    X2 = [x.clone() for x in X]
    st2 = orc.train_step(P, Bf, None, X2, torch.eye(B), torch.zeros(B, B), noise, 0., 1.0,
                         do_step=False, return_grads=True)
    assert torch.equal(st['grads']['fc_vars.1.bias'], st2['grads']['fc_vars.1.bias'])


def test_identity_P_is_index_equality():
    rng = np.random.default_rng(0)
    N = 50
    idx0 = rng.integers(0, N, 20)
    idx1 = rng.integers(0, N, 20)
    dense = orc.p_block(torch.eye(N), idx0, idx1)
    assert torch.equal(dense, orc.p_block(None, idx0, idx1))


def test_m_modality_generalisation_reduces_to_reference_for_two():
    """The build-defined M-modality combine (identity correspondence) equals the reference's two-modality combine at
    corr = I, and the generalised loss code gives the same four losses for M = 2 ('M = 2 reduction')."""
    torch.manual_seed(1)
    dims, L, B = (12, 10), 3, 16
    P, Bf = orc.init_state(dims, L)
    X = [torch.randn(B, d) for d in dims]
    noise = orc.draw_noise(dims, L, B, 0.)
    a = orc.forward(P, dict(Bf), X, torch.eye(B), train=True, p=0., noise=noise)
    b = orc.forward(P, dict(Bf), X, None, train=True, p=0., noise=noise)
    for u, v in zip(a[1], b[1]):
        assert torch.allclose(u, v, rtol=1e-6, atol=1e-7)
    la = orc.losses(X, *a, torch.zeros(B, B), 0.3)
    lb = orc.losses(X, *b, None, 0.3)
    for u, v in zip(la, lb):
        assert torch.allclose(u, v, rtol=1e-6, atol=1e-8)


# ---- correspondence stage: the oracle's Prime_Dual against the reference's own outputs (tools/make_goldens_pd.py) ----
@pytest.mark.parametrize('name', ['pd1_delay0', 'pd2_delay', 'pd3_pipeline'])
def test_prime_dual_oracle_vs_reference_golden(name):
    import ast
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', name + '.npz'))
    m = ast.literal_eval(str(g['meta']))
    if 'Kx' in g:
        Kx, Ky = g['Kx'], g['Ky']
    else:                                              # stage A + B wiring: euclidean distances, then Prime_Dual
        Kx, Ky = orc.distance_matrix(g['X'], 'euclidean'), orc.distance_matrix(g['Y'], 'euclidean')
        np.testing.assert_allclose(Kx, g['dist0'], rtol=0, atol=1e-12)
        np.testing.assert_allclose(Ky, g['dist1'], rtol=0, atol=1e-12)
    F = orc.prime_dual(Kx, Ky, m['dx'], m['dy'], m['epoch_pd'], m['rho'], m['epsilon'], m['delay'])
    np.testing.assert_allclose(F, g['F'], rtol=1e-6, atol=1e-9)


# ---- bf16-operand emulation (the checker of the HIP path's bf16 compute mode; not a feature of the reference) ----
def test_emulation_switched_off_is_the_plain_oracle_and_linear_rounds_operands_only():
    torch.manual_seed(4)
    dims, L, B, p = (40, 24), 8, 32, 0.6
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    X = [torch.randn(B, d) for d in dims]
    noise = orc.draw_noise(dims, L, B, p)
    P2 = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in P.items())
    Bf2 = OrderedDict((k, v.clone()) for k, v in Bf.items())
    a = orc.train_step(P, Bf, None, X, torch.eye(B), None, noise, p, .5, do_step=False, return_grads=True)
    off = {k: (False, False, False) for k in orc.EMULATE_HIP_BF16_FUSED}
    b = orc.train_step(P2, Bf2, None, X, torch.eye(B), None, noise, p, .5, do_step=False, return_grads=True, emulate=off)
    assert a['losses'] == b['losses'] and all(torch.equal(a['grads'][k], b['grads'][k]) for k in a['grads'])
    # one layer by hand: y = bf(a) bf(W)^T + b;  da = bf(dy) bf(W);  dW = bf(dy)^T bf(a);  db = sum(dy) in fp32
    bf = orc.bf16_round
    h = torch.randn(B, 40, requires_grad=True)
    Pl = {'encoders.0.4.weight': torch.randn(24, 40, requires_grad=True), 'encoders.0.4.bias': torch.randn(24, requires_grad=True)}
    y = orc.linear(Pl, 'encoders.0.4', h, {'enc1': (True, True, True)})
    W, bb = Pl['encoders.0.4.weight'], Pl['encoders.0.4.bias']
    assert torch.equal(y, torch.nn.functional.linear(bf(h), bf(W), bb))
    dy = torch.randn(B, 24)
    gh, gW, gb = torch.autograd.grad(y, [h, W, bb], dy)
    assert torch.equal(gh, bf(dy) @ bf(W)) and torch.equal(gW, bf(dy).t() @ bf(h)) and torch.equal(gb, dy.sum(0))
    # a product left in fp32 is the plain one
    y2 = orc.linear(Pl, 'encoders.0.4', h, {'enc1': (True, False, True)})
    gh2, = torch.autograd.grad(y2, [h], dy)
    assert torch.equal(gh2, dy @ W)


def test_emulated_step_rounds_gradients_once_and_clips_by_the_unrounded_norm():
    torch.manual_seed(5)
    dims, L, B, p = (40, 24), 8, 32, 0.0
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    init = {k: v.detach().clone() for k, v in P.items()}
    X = [torch.randn(B, d) for d in dims]
    noise = orc.draw_noise(dims, L, B, p)
    st = orc.train_step(P, Bf, orc.Adam(P.values(), 1e-3), X, torch.eye(B), None, noise, p, .5, return_grads=True,
                        emulate=orc.EMULATE_HIP_BF16_FUSED, grad_bf16=True)
    gn = float(torch.sqrt(sum((g.double() ** 2).sum() for g in st['grads'].values())))
    assert abs(gn - st['grad_norm']) < 1e-5 * gn
    coef = min(1.0, 1.0 / (st['grad_norm'] + 1e-6))
    for k, g in st['grads'].items():
        gc = (orc.bf16_round(g) * coef).double()
        want = init[k].double() - 1e-3 * gc / (gc.abs() + 1e-8)
        assert float(((P[k].detach().double() - want).abs() - 1.2e-7 * want.abs()).max()) < 1e-8, k    # (fp32 storage)
