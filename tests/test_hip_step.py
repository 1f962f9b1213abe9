"""End-to-end parity of the HIP training step / inference with the reference:
  * against the golden fixtures (outputs of the reference itself, tools/make_goldens.py), driven with the
    fixtures' explicit noise and index streams;
  * against the CPU oracle on seeded inputs at larger sizes.
Run on the MI355X box:  pytest -m gpu"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

from oracle import jamie_oracle as orc
from golden_util import CASES, Golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def jam():
    import jamie_amd
    from jamie_amd import _native
    _native.require_gpu()
    return jamie_amd


def assert_mostly_close(got, want, rtol, atol, max_bad_frac=2e-3, rel_l2=2e-2, msg=''):
    """Two correct fp32 implementations of this network cannot agree element by element over several steps:
    LeakyReLU/BatchNorm make the loss piecewise smooth, and an activation within ~1e-7 of the kink takes a
    different branch in each (observed: 3 of 2M activations at step 2 of the (520,260) case), after which
    Adam amplifies the difference.  So the bulk must agree tightly (all but `max_bad_frac` of the elements
    within rtol/atol) and the whole tensor in relative L2."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    bad = np.abs(got - want) > atol + rtol * np.abs(want)
    frac = bad.mean() if bad.size else 0.0
    nrm = np.linalg.norm(want)
    rel = np.linalg.norm(got - want) / nrm if nrm > 0 else np.linalg.norm(got - want)
    assert frac <= max_bad_frac and rel <= rel_l2, f'{msg}: {frac:.2%} elements off, relL2 {rel:.2e}'


def _noise_to_dev(noise, p):
    out = {'eps': [e.cuda().contiguous() for e in noise['eps']], 'enc_masks': [], 'dec_masks': []}
    for k in ('enc_masks', 'dec_masks'):
        for pair in noise[k]:
            out[k].append([None if (m is None or p == 0) else m.to(torch.uint8).cuda().contiguous() for m in pair])
    return out


def _engine_from_golden(jam, g, lr=1e-3):
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    m = g.meta
    c = m['ctor']
    model = edModelVar(m['dims'], m['L'], dropout=c.get('dropout'))
    model.load_state_dict(g.state('init'))
    eng = TrainEngine(model, m['B'], lr=c.get('model_lr', lr), loss_weights=c.get('loss_weights'),
                      dist_method=c.get('dist_method', 'euclidean'))
    return model, eng


def _blocks(g, s, B):
    m = g.meta
    idx = g.step_indices(s)
    PF = m['ctor'].get('PF_Ratio') or 1
    Fm = torch.from_numpy(g['F']) if m['has_F'] else None
    Pm = torch.zeros(m['rows'][0], m['rows'][1]) if m['sampling_method'] == 'zeros' else None
    Fblk = orc.f_block(Fm, idx[0], idx[1], B)
    corr = PF * orc.p_block(Pm, idx[0], idx[1]) + (1 - PF) * Fblk
    return idx, corr, (Fblk if m['has_F'] else None)


@pytest.mark.parametrize('name', CASES)
def test_first_step_vs_reference_golden(jam, name):
    """Losses, pre-clip gradients and post-step weights of step 0 against the reference's own numbers."""
    g = Golden(name)
    m = g.meta
    model, eng = _engine_from_golden(jam, g)
    assert model.dropout == m['p']
    B = m['B']
    idx, corr, Fblk = _blocks(g, 0, B)
    for i in range(2):
        eng.ws[i]['x'].copy_(torch.from_numpy(g[f's0.X{i}']))
    c = m['ctor']
    eng.set_kl_anneal(orc.kl_anneal(0, c.get('min_epochs', 2500), m['epochs']))
    noise = _noise_to_dev(g.noise(0), m['p'])
    is_identity = torch.equal(corr, torch.eye(B))
    eng.forward_backward(None if is_identity else corr.cuda().contiguous(),
                         None if Fblk is None else Fblk.cuda().contiguous(), noise)
    torch.cuda.synchronize()
    for i in range(2):
        np.testing.assert_allclose(eng.ws[i]['z'].cpu().numpy(), g[f's0.z{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(eng.ws[i]['comb'].cpu().numpy(), g[f's0.comb{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(eng.ws[i]['mu'].cpu().numpy(), g[f's0.mu{i}'], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(eng.ws[1]['lv'].cpu().numpy(), g['s0.logvar'], rtol=1e-4, atol=1e-5)
    ls, total, _ = eng.read_losses()
    if m['steps'] == m['epochs']:          # (one batch per epoch: the recorded loss of epoch 0 is step 0's)
        np.testing.assert_allclose(ls, g['loss_history'][:, 0], rtol=1e-4, atol=1e-6)
    # gradients in the reference's names
    names = model.layout.reference_names()
    gv = eng.g
    for ref, (mine, sl) in names.items():
        if orc.is_dead_bias(ref):
            continue
        got = gv[mine] if sl is None else gv[mine][sl]
        want = g['grad0.' + ref] if ('grad0.' + ref) in g else np.zeros(tuple(got.shape), np.float32)
        scale = max(1e-6, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-3, atol=2e-5 * scale + 1e-6, err_msg=ref)
    if m['steps'] == 1:
        eng.optimizer_step()
        torch.cuda.synchronize()
        sd = model.state_dict()
        fin = g.state('final')
        for k, v in sd.items():
            if orc.is_dead_bias(k) or k.endswith('num_batches_tracked'):
                continue
            np.testing.assert_allclose(v.cpu().numpy(), fin[k].numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


@pytest.mark.parametrize('name', ['g3_multistep', 'g4_replace', 'g5_F_pfratio', 'g6_zeros', 'g7_cosine', 'g8_klquirk',
                                  'g10_midsize'])
def test_multistep_replay_vs_reference_golden(jam, name):
    """All steps of the fixture with its explicit index and noise streams: per-epoch loss history, final
    weights, final embeddings, transform and modal_predict against the reference's outputs."""
    g = Golden(name)
    m = g.meta
    c = m['ctor']
    model, eng = _engine_from_golden(jam, g)
    B = m['B']
    data = g.data() if m['rows'][0] == m['rows'][1] else [g['data0'].astype(np.float64), g['data1'].astype(np.float64)]
    pre = [orc.Preclass(d, axis=0) for d in data]
    dd = [torch.from_numpy(p.transform(d)).float().cuda().contiguous() for p, d in zip(pre, data)]
    steps_per_epoch = m['steps'] // m['epochs']
    hist = []
    s = 0
    for epoch in range(m['epochs']):
        eng.set_kl_anneal(orc.kl_anneal(epoch, c.get('min_epochs', 2500), m['epochs']))
        for _ in range(steps_per_epoch):
            idx, corr, Fblk = _blocks(g, s, B)
            eng.load_batch(dd, [torch.from_numpy(i.astype(np.int32)).cuda() for i in idx])
            is_identity = torch.equal(corr, torch.eye(B))
            eng.step(None if is_identity else corr.cuda().contiguous(),
                     None if Fblk is None else Fblk.cuda().contiguous(), _noise_to_dev(g.noise(s), m['p']))
            s += 1
        hist.append(eng.read_losses()[0])
    big_lr = c.get('model_lr', 1e-3) > 1e-2          # g8: lr 5e-2 amplifies rounding noise quickly
    np.testing.assert_allclose(np.array(hist).T, g['loss_history'], rtol=5e-2 if big_lr else 2e-3, atol=1e-5)
    sd = model.state_dict()
    fin = g.state('final')
    for k, v in sd.items():
        # dead pre-BN biases random-walk by +-lr per step on rounding noise; the BN running
        # MEAN that follows such a Linear contains that bias, so it is excluded too (running_var is not)
        if orc.is_dead_bias(k) or k.endswith('num_batches_tracked') or k.endswith('running_mean'):
            continue
        if k == 'sigma' and m['sampling_method'] == 'zeros':
            continue      # corr = 0 -> comb_i = z_i: sigma's gradient is rounding noise too (another dead parameter)
        assert_mostly_close(v.cpu().numpy(), fin[k].numpy(), rtol=2e-3, atol=5e-5,
                            max_bad_frac=0.5 if big_lr else 0.02, rel_l2=0.3 if big_lr else 2e-2, msg=k)
    # Eval-mode outputs depend on (dead bias - running mean of its history), i.e. on rounding noise that
    # Adam turned into +-lr moves, in the reference as much as here: compare in relative L2 only.
    model.eval()
    for i in range(2):
        emb = model.embed(dd[i], i).cpu().numpy()
        assert_mostly_close(emb, g[f'emb{i}'], rtol=0, atol=0, max_bad_frac=1.0,
                            rel_l2=0.1 if big_lr else 1.5e-2, msg=f'emb{i}')
        imp = pre[(i + 1) % 2].inverse_transform(model.impute(dd[i], [i, (i + 1) % 2]).cpu().numpy())
        assert_mostly_close(imp, g[f'impute_from{i}'], rtol=0, atol=0, max_bad_frac=1.0,
                            rel_l2=0.1 if big_lr else 1.5e-2, msg=f'impute{i}')


@pytest.mark.parametrize('name', CASES)
def test_inference_from_reference_weights(jam, name):
    """north_star parity claim: load the REFERENCE's trained weights; embeddings (`transform`) and imputed
    matrices (`modal_predict`) within rtol 1e-4 / atol 1e-5 of the reference's CPU outputs."""
    g = Golden(name)
    m = g.meta
    from jamie_amd.model import edModelVar
    model = edModelVar(m['dims'], m['L'])
    model.load_state_dict(g.state('final'))
    model.eval()
    data = [g['data0'].astype(np.float64), g['data1'].astype(np.float64)]
    pre = [orc.Preclass(d, axis=0) for d in data]
    for i in range(2):
        x = torch.from_numpy(pre[i].transform(data[i])).float()
        np.testing.assert_allclose(model.embed(x, i).cpu().numpy(), g[f'transform{i}'], rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(model.fc_mus[i](model.encoders[i](x)).cpu().numpy(), g[f'transform_one{i}'],
                                   rtol=1e-4, atol=1e-5)
        to = (i + 1) % 2
        imp = pre[to].inverse_transform(model.impute(x, [i, to]).cpu().numpy())
        np.testing.assert_allclose(imp, g[f'impute_from{i}'], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize('B,dims,L,p', [(256, (300, 180), 16, 0.6), (512, (520, 260), 32, 0.6), (96, (70, 50), 8, 0.0)])
def test_step_vs_oracle_seeded(jam, B, dims, L, p):
    """Three consecutive steps against the oracle at sizes where split-K, multi-tile GEMMs and the cached
    BN path are all exercised (identity corr, F = 0)."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    torch.manual_seed(123)
    model = edModelVar(dims, L, dropout=p)
    torch.manual_seed(123)
    P, Bf = orc.init_state(dims, L)
    sd = model.state_dict()
    for k, v in P.items():
        assert torch.equal(sd[k].cpu(), v), k
        v.requires_grad_(True)
    eng = TrainEngine(model, B)
    opt = orc.Adam(P.values(), 1e-3)
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((B, 16))
    X = [torch.from_numpy((Z @ rng.standard_normal((16, d)) + .1 * rng.standard_normal((B, d))).astype(np.float32))
         for d in dims]
    X = [(x - x.mean(0)) / x.std(0) for x in X]
    for step in range(3):
        torch.manual_seed(1000 + step)
        noise = orc.draw_noise(dims, L, B, p)
        anneal = 0.3 + 0.1 * step
        st = orc.train_step(P, Bf, opt, X, torch.eye(B), torch.zeros(B, B), noise, p, anneal, return_grads=True)
        for i in range(2):
            eng.ws[i]['x'].copy_(X[i])
        eng.set_kl_anneal(anneal)
        eng.forward_backward(None, None, _noise_to_dev(noise, p))
        ls, total, _ = eng.read_losses()
        np.testing.assert_allclose(ls, st['losses'], rtol=2e-4, atol=1e-6)
        names = model.layout.reference_names()
        for ref, (mine, sl) in names.items():
            if orc.is_dead_bias(ref):
                continue
            got = eng.g[mine] if sl is None else eng.g[mine][sl]
            want = st['grads'][ref].numpy()
            scale = max(1e-6, float(np.abs(want).max()))
            if step == 0:     # identical state: tight, element by element
                np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=2e-5 * scale + 1e-8,
                                           err_msg=f'step {step} {ref}')
            assert_mostly_close(got.cpu().numpy(), want, rtol=0, atol=0, max_bad_frac=1.0, rel_l2=2e-2,
                                msg=f'step {step} grad {ref}')
        eng.optimizer_step()
        sd = model.state_dict()
        for k, v in P.items():
            if orc.is_dead_bias(k):
                continue
            # Adam moves an element whose gradient is pure rounding noise by +-lr: allow a handful
            assert_mostly_close(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-3, atol=2e-5,
                                max_bad_frac=1e-4 if step == 0 else 5e-2, rel_l2=2e-3, msg=f'step {step} {k}')


def test_rng_mode_trains(jam):
    """Philox dropout/eps + device sampler: loss decreases and stays finite (no explicit noise)."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    torch.manual_seed(5)
    dims, L, B, N = (120, 80), 8, 128, 1024
    model = edModelVar(dims, L)
    eng = TrainEngine(model, B, lr=1e-3)
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((N, 8))
    data = [torch.from_numpy((Z @ rng.standard_normal((8, d)) + .1 * rng.standard_normal((N, d))).astype(np.float32))
            for d in dims]
    data = [((x - x.mean(0)) / x.std(0)).cuda().contiguous() for x in data]
    from jamie_amd import _native as nv
    idx = [torch.zeros(B, dtype=torch.int32, device='cuda') for _ in range(2)]
    first = last = None
    for step in range(150):
        nv.sample_indices(idx[0], N, 0, False, eng.state, 200)
        idx[1].copy_(idx[0])
        eng.load_batch(data, idx)
        eng.step()
        if step == 0:
            first = eng.read_losses()[1]
    last = eng.read_losses()[1]
    assert np.isfinite(last) and last < first
    assert int(eng.state[1].item()) == 150


def test_facade_fit_transform_matches_oracle_loop(jam):
    """JAMIE facade (numpy sampler = the reference's index stream) vs the oracle's restated loop with
    dropout 0: the only noise is eps, which differs (Philox vs torch), so compare with eps-free KL weight...
    instead check API behaviour: shapes, dtypes, loss history length, save/load round trip, determinism of
    transform/modal_predict and that impute returns float64."""
    import io
    import contextlib
    rng = np.random.default_rng(3)
    N, dims = 300, (40, 30)
    Z = rng.standard_normal((N, 5))
    data = [Z @ rng.standard_normal((5, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
    np.random.seed(42)
    with contextlib.redirect_stdout(io.StringIO()):
        jm = jam.JAMIE(output_dim=6, batch_size=64, epoch_DNN=30, min_epochs=10, pca_dim=None, use_f_tilde=False,
                       log_DNN=10 ** 9)
        emb = jm.fit_transform(dataset=data)
    assert [e.shape for e in emb] == [(N, 6), (N, 6)] and emb[0].dtype == np.float32
    assert set(jm.loss_history) == {'KL', 'Rec', 'CosSim', 'F'} and len(jm.loss_history['Rec']) == 30
    assert jm.loss_history['Rec'][-1] < jm.loss_history['Rec'][0]
    tr = jm.transform(data)
    np.testing.assert_allclose(tr[0], emb[0], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(jm.transform_one(data[1], 1), emb[1], rtol=1e-5, atol=1e-6)
    imp = jm.modal_predict(data[0], 0)
    assert imp.shape == (N, 30) and imp.dtype == np.float64
    np.testing.assert_array_equal(jm.impute(data[0], 0), imp)
    buf = io.BytesIO()
    jm.save_model(buf)
    buf.seek(0)
    jm2 = jam.JAMIE(output_dim=6, use_f_tilde=False)
    jm2.load_model(buf)
    np.testing.assert_array_equal(jm2.transform(data)[1], tr[1])
    # aligned cells should be closer than random pairs after training on a shared latent
    assert jm.test_closer(emb) < 0.35


def test_bench_two_ranks_share_one_gpu(tmp_path):
    """bench.py's multi-rank path end to end (row shards, parameter broadcast, gradient all-reduce, barrier,
    MAX over ranks, one JSON line) with two ranks on cuda:0 over gloo: everything except RCCL itself."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JAMIE_DIST_BACKEND='gloo', JAMIE_SHARE_GPU='1', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29577', os.path.join(root, 'bench.py'),
                        '--gpus', '2', '--steps', '6', '--warmup', '2', '--config', 'c1', '--dtype', 'f32'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['value'] > 0 and 'cpu_baseline' not in out
    assert np.isfinite(out['final_loss'])
    # the headline at N > 1 is north_star's arrangement (ONE all-reduce of the gradient, the full update on every rank); the
    # same job with the sharded optimiser is timed behind it
    assert out['config']['dp_optimizer'] == 'replicated' and out['sharded_optimizer']['dp_optimizer'] == 'sharded'
    assert out['sharded_optimizer']['value'] > 0
    # ... and the other way round on request
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29576', os.path.join(root, 'bench.py'),
                        '--gpus', '2', '--steps', '6', '--warmup', '2', '--config', 'c1', '--dtype', 'f32', '--dp-optimizer', 'auto'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['config']['dp_optimizer'] == 'sharded' and out['replicated_optimizer']['dp_optimizer'] == 'replicated'
    assert out['replicated_optimizer']['value'] > 0
    # bf16 compute at N > 1: the line also carries the same job timed with fp32 gradient messages
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29578', os.path.join(root, 'bench.py'),
                        '--gpus', '2', '--steps', '8', '--warmup', '2', '--config', 'c1'],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['dtype'] == 'bf16' and out['config']['grad_allreduce'] == 'bf16'
    assert out['grad_comm_f32']['grad_allreduce'] == 'f32' and out['grad_comm_f32']['value'] > 0
    # (config 1 in bf16 multiplies with transposed copies of its small weight matrices: the optimiser stays replicated there)
    assert out['grad_comm_f32']['dp_optimizer'] == out['config']['dp_optimizer'] == 'replicated'
    assert out['sharded_optimizer']['value'] is None and 'not available' in out['sharded_optimizer']['note']


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher (how the driver invokes it): the parent starts the two ranks itself
    before it touches the GPU, relays rank 0's ONE JSON line and reports n_gpus = 2; a request for more GPUs than the box
    has exits non-zero instead of printing a line for fewer; a failing rank makes the whole invocation fail."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT')}
    env = dict(clean, JAMIE_DIST_BACKEND='gloo', JAMIE_SHARE_GPU='1')
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
                        '--config', 'c1', '--dtype', 'f32'], capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['parallelism'] == 'dp2' and out['value'] > 0
    assert out['rccl']['world'] == 2 and out['rccl']['backend'] == 'gloo'
    assert out['config']['dp_optimizer'] == 'replicated' and np.isfinite(out['final_loss'])
    # the line explains its own exchange: measured all-reduce bandwidth at the step's message sizes (both message dtypes) and the
    # time the step's stream stood still waiting for messages (HIP events around finish()'s waits)
    msgs = out['exchange']['messages_per_step']
    assert msgs and sum(m['elements'] for m in msgs) >= out['config']['parameters']
    exp = out['exchange']['exposed_us_per_step']
    assert exp is not None and exp['n'] >= 1 and exp['median'] >= 0.0
    probe = out['rccl']['allreduce_probe']
    assert {(p['elements'], p['dtype']) for p in probe} == {(m['elements'], dt) for m in msgs for dt in ('bf16', 'f32')}
    assert all(p['median_us'] > 0 and p['algbw_GBps'] > 0 and p['busbw_GBps'] == pytest.approx(p['algbw_GBps'], rel=1e-6) for p in probe)
    assert out['steady_state']['ms_per_step'] > 0 and out['config']['settle_steps_before_warmup'] == 0
    # more GPUs than the box has (one): refused, no JSON line
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--config', 'c1'], capture_output=True, text=True, env=clean, timeout=300)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        assert 'GPU(s) visible' in r.stderr
    # a rank that fails takes the invocation down with it (the other rank is not left waiting forever)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1',
                        '--config', 'c1', '--batch', '-3'], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]


def test_bench_config5_path_two_ranks_at_reduced_cells(tmp_path):
    """BASELINE config 5's bench path -- `--config c5` (fp32 by default, (5000, 2000) features, latent 64, the noise term of the
    generator drawn on the device) -- at a reduced cell count, two ranks sharing cuda:0 over gloo: contiguous row shards, the
    fp32 gradient all-reduce of 233 M parameters in overlapped regions, one JSON line naming what ran."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, JAMIE_DIST_BACKEND='gloo', JAMIE_SHARE_GPU='1', MASTER_ADDR='127.0.0.1', JAMIE_BENCH_DEVICE_NOISE='1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29579', os.path.join(root, 'bench.py'),
                        '--gpus', '2', '--steps', '3', '--warmup', '1', '--config', 'c5', '--cells', '6001'],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 2 and out['dtype'] == 'f32' and out['config']['cells'] == 6001
    assert out['config']['features'] == [5000, 2000] and out['config']['parameters'] == 233477258
    assert 'drawn on the device' in out['config']['generator'] and out['config']['grad_allreduce'] == 'f32'
    assert np.isfinite(out['final_loss']) and out['value'] > 0


@pytest.mark.parametrize('mode', ['f32_buckets', 'bf16_overlapped_plan'])
def test_data_parallel_ranks_stay_identical(tmp_path, mode):
    """Two ranks (sharing cuda:0, gloo) training on different shards keep bit-identical parameters after
    several steps: the all-reduced gradient and the 1/world average give every rank the same update.
    `bf16_overlapped_plan`: bf16 compute, region-wise overlapped exchange with bf16 messages, recorded plan
    (the configuration bench.py runs at N > 1)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'dp.py'
    script.write_text(f'''
import sys, torch, numpy as np
sys.path.insert(0, {root!r})
from jamie_amd import distributed as jd, _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
rank, world, local = jd.init_from_env()
dev = torch.device('cuda', local)
torch.manual_seed(100 + rank)                     # different initial weights per rank ...
model = edModelVar((96, 64), 8, device=dev)
jd.broadcast_flat(model.flat)                     # ... until rank 0's are broadcast
mode = {mode!r}
g = torch.Generator(device=dev).manual_seed(50 + rank)
data = [torch.randn(512, d, generator=g, device=dev) for d in (96, 64)]
idx = [torch.zeros(64, dtype=torch.int32, device=dev) for _ in range(2)]
if mode == 'f32_buckets':
    eng = TrainEngine(model, 64, seed=1 + rank, world_size=world)
    ar = jd.GradAllReduce(n_buckets=3)
    for s in range(5):
        nv.sample_indices(idx[0], 512, 0, False, eng.state, 200)
        idx[1].copy_(idx[0])
        eng.load_batch(data, idx)
        eng.step(None, None, None, ar)
else:
    init = model.flat.clone()
    eng = TrainEngine(model, 64, seed=1 + rank, world_size=world, compute_dtype='bf16')
    ar = jd.OverlappedGradAllReduce(min_bytes=16384, comm_dtype=torch.bfloat16)
    plan = eng.make_plan(data, idx[0], 512, False, ar)
    for s in range(4):
        eng.run_plan(plan)
    assert ar.comm is not None and ar.comm.dtype == torch.bfloat16 and not ar.works
    # the step above leaves the reduced gradient in the bf16 message buffer (norm and Adam read it there); casting it
    # back into the fp32 gradient first, as a caller of finish() + optimizer_step() does, is the same update up to the
    # summation order of the norm (the clip coefficient's last bit)
    model2 = edModelVar((96, 64), 8, device=dev)
    model2.flat.copy_(init)
    eng2 = TrainEngine(model2, 64, seed=1 + rank, world_size=world, compute_dtype='bf16')
    ar2 = jd.OverlappedGradAllReduce(min_bytes=16384, comm_dtype=torch.bfloat16)
    for s in range(5):
        nv.sample_indices(idx[0], 512, 0, False, eng2.state, 200)
        eng2.load_batch(data, [idx[0], idx[0]])
        eng2.forward_backward(None, None, None, ar2)
        ar2.finish()
        eng2.optimizer_step()
    assert torch.allclose(model.flat, model2.flat, rtol=1e-5, atol=1e-7), (model.flat - model2.flat).abs().max()
flat = model.flat.clone()
others = [torch.zeros_like(flat) for _ in range(world)]
torch.distributed.all_gather(others, flat)
assert torch.equal(others[0], others[1]), (others[0] - others[1]).abs().max()
assert torch.isfinite(flat).all()
print('DP OK', rank)
torch.distributed.destroy_process_group()
''')
    env = dict(os.environ, JAMIE_DIST_BACKEND='gloo', JAMIE_SHARE_GPU='1', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29578', str(script)],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert r.stdout.count('DP OK') == 2


@pytest.mark.parametrize('B,dims,L,p', [(256, (304, 184), 16, 0.6), (512, (520, 264), 32, 0.0),
                                        (1024, (264, 136), 16, 0.6), (1024, (520, 264), 32, 0.0), (640, (264, 256), 8, 0.3)])
def test_bf16_compute_mode_tracks_fp32_oracle(jam, B, dims, L, p):
    """bf16 MFMA GEMMs (fp32 accumulate, fp32 master weights/optimiser/BN/losses): one step against the fp32
    oracle within bf16 rounding (operands carry 8 significant bits): losses 2 %, gradients 10 % in relative L2 (the first encoder layer sits behind six bf16 products: 5.7 % measured)."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    torch.manual_seed(7)
    model = edModelVar(dims, L, dropout=p)
    torch.manual_seed(7)
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    eng = TrainEngine(model, B, compute_dtype='bf16')
    rng = np.random.default_rng(1)
    Z = rng.standard_normal((B, 16))
    X = [torch.from_numpy((Z @ rng.standard_normal((16, d)) + .1 * rng.standard_normal((B, d))).astype(np.float32))
         for d in dims]
    X = [(x - x.mean(0)) / x.std(0) for x in X]
    torch.manual_seed(11)
    noise = orc.draw_noise(dims, L, B, p)
    st = orc.train_step(P, Bf, None, X, torch.eye(B), torch.zeros(B, B), noise, p, 0.4, do_step=False, return_grads=True)
    eng.set_batch([x.cuda() for x in X])
    eng.set_kl_anneal(0.4)
    eng.forward_backward(None, None, _noise_to_dev(noise, p))
    ls, total, _ = eng.read_losses()
    np.testing.assert_allclose(ls, st['losses'], rtol=2e-2, atol=1e-5)
    for ref, (mine, sl) in model.layout.reference_names().items():
        if orc.is_dead_bias(ref):
            continue
        gv = eng.grad_view(mine)
        got = (gv if sl is None else gv[sl]).cpu().numpy()
        assert_mostly_close(got, st['grads'][ref].numpy(), rtol=0, atol=0, max_bad_frac=1.0, rel_l2=1e-1, msg=ref)
    eng.optimizer_step()
    assert torch.equal(eng.wbf['m0.enc0.W'].float(), model.p['m0.enc0.W'].to(torch.bfloat16).float())
    # the big layers' dX products read W as stored (b_tr); so do the skinny head / latent layers when every layer is large
    # enough for the 128 x 128 k-row-major kernel (skinny_tr): then NO transposed weight copy exists
    assert ('m1.dec1' not in eng.wT) == (min(dims) >= 256)
    assert (len(eng.wT) == 0) == bool(eng.skinny_tr)
    eng.flush()            # the transposed copies are refreshed with the next batch launch, or on flush()
    for k, wt in eng.wT.items():
        assert torch.equal(wt.float(), model.p[k + '.W'].t().to(torch.bfloat16).float()), k


def test_bf16_mode_trains_like_fp32(jam):
    """Same data, same Philox noise streams: after 120 steps the bf16-compute run reaches the fp32 run's
    reconstruction loss within 5 %."""
    from jamie_amd import _native as nv
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    dims, L, B, N = (128, 80), 8, 128, 2048
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((N, 8))
    data = [torch.from_numpy((Z @ rng.standard_normal((8, d)) + .1 * rng.standard_normal((N, d))).astype(np.float32))
            for d in dims]
    data = [((x - x.mean(0)) / x.std(0)).cuda().contiguous() for x in data]
    final = {}
    for mode in ('f32', 'bf16'):
        torch.manual_seed(5)
        model = edModelVar(dims, L)
        eng = TrainEngine(model, B, compute_dtype=mode, seed=3)
        idx = [torch.zeros(B, dtype=torch.int32, device='cuda') for _ in range(2)]
        acc = []
        for step in range(120):
            nv.sample_indices(idx[0], N, 0, False, eng.state, 200)
            idx[1].copy_(idx[0])
            eng.load_batch(data, idx)
            eng.step()
            if step >= 110:
                acc.append(eng.read_losses()[0][1])
        final[mode] = float(np.mean(acc))
    assert np.isfinite(final['bf16']) and abs(final['bf16'] - final['f32']) < 0.05 * final['f32'], final


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_plan_replay_equals_eager_steps(jam, mode):
    """The recorded launch plan (one foreign call per launch) is bit-identical to issuing the step eagerly."""
    from jamie_amd import _native as nv
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    dims, L, B, N = (96, 64), 8, 64, 1024
    g = torch.Generator().manual_seed(0)
    data = [torch.randn(N, d, generator=g).cuda() for d in dims]
    flats = []
    for use_plan in (False, True, 'prefetch'):         # 'prefetch': next batch sampled + gathered under clip + Adam
        torch.manual_seed(9)
        model = edModelVar(dims, L)
        eng = TrainEngine(model, B, compute_dtype=mode, seed=21)
        idx = torch.zeros(B, dtype=torch.int32, device='cuda')
        if use_plan:
            plan = eng.make_plan(data, idx, N, prefetch=(use_plan == 'prefetch'))
            for _ in range(5):
                eng.run_plan(plan)
        else:
            for _ in range(6):
                nv.sample_indices(idx, N, 0, False, eng.state, 200)
                eng.load_batch(data, [idx, idx])
                eng.step()
        assert int(eng.state[1].item()) == 6 and model.num_batches_tracked == 6
        flats.append(model.flat.clone())
    assert torch.equal(flats[0], flats[1]) and torch.equal(flats[0], flats[2])


@pytest.mark.parametrize('mode,variant', [('f32', 'pipeline'), ('bf16', 'pipeline'), ('bf16', 'side')])
@pytest.mark.parametrize('use_plan', [False, True])
def test_pipelined_optimizer_is_bit_identical(jam, mode, use_plan, variant):
    """clip + Adam on the optimiser stream, group by group under the next step's forward pass
    (TrainEngine.enable_pipeline), produces exactly the parameters, moments and bf16 copies of the one-launch form."""
    from jamie_amd import _native as nv
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    dims, L, B, N = (264, 136), 8, 128, 2048
    g = torch.Generator().manual_seed(0)
    data = [torch.randn(N, d, generator=g).cuda() for d in dims]
    out = []
    for pipe in (False, True):
        torch.manual_seed(9)
        model = edModelVar(dims, L)
        eng = TrainEngine(model, B, compute_dtype=mode, seed=21)
        if pipe and variant == 'pipeline':
            eng.enable_pipeline()
        elif pipe:                      # only the transposed bf16 weight copies move to a side stream
            eng.enable_side_transposes()
        idx = torch.zeros(B, dtype=torch.int32, device='cuda')
        if use_plan:
            plan = eng.make_plan(data, idx, N)
            for _ in range(7):
                eng.run_plan(plan)
        else:
            for _ in range(8):
                nv.sample_indices(idx, N, 0, False, eng.state, 200)
                eng.load_batch(data, [idx, idx])
                eng.step()
        eng.flush()
        ls = eng.read_losses()
        out.append((model.flat.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(),
                    eng.wbf_flat.clone() if mode == 'bf16' else None,
                    {k: v.clone() for k, v in eng.wT.items()} if mode == 'bf16' else {}, ls))
    a, b = out
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    assert a[5] == b[5]
    if mode == 'bf16':
        assert torch.equal(a[3], b[3])
        for k in a[4]:
            assert torch.equal(a[4][k], b[4][k]), k


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_facade_device_sampler_plan_path(jam, mode):
    """sampler='device' + identity P takes the recorded-plan fast path; bf16 compute through the facade."""
    import io
    import contextlib
    rng = np.random.default_rng(4)
    N, dims = 600, (72, 40)
    Z = rng.standard_normal((N, 5))
    data = [Z @ rng.standard_normal((5, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
    with contextlib.redirect_stdout(io.StringIO()):
        jm = jam.JAMIE(output_dim=8, batch_size=64, epoch_DNN=25, min_epochs=10, pca_dim=None, use_f_tilde=False,
                       log_DNN=10 ** 9, sampler='device', compute_dtype=mode)
        emb = jm.fit_transform(dataset=data)
    assert jm.engine.compute_dtype == mode and int(jm.engine.state[1].item()) == 25 * (N // 64)
    assert len(jm.loss_history['Rec']) == 25 and jm.loss_history['Rec'][-1] < jm.loss_history['Rec'][0]
    assert np.isfinite(emb[0]).all() and jm.test_closer(emb) < 0.35


def test_autograd_model_class_seam_matches_oracle(jam):
    """The reference's own seam: a `model_class` whose train-mode forward returns autograd tensors, with the
    losses formed OUTSIDE by torch ops (here the oracle's restatement of jamie.py:618-668 on GPU tensors),
    `.backward()`, `clip_grad_norm_` and `torch.optim.Adam` on `model.parameters()` — dropout 0 so that the only
    noise is eps; compare one step's losses and gradients with the CPU oracle given the same eps."""
    from jamie_amd.compat import edModelVarTorch
    dims, L, B = (48, 40), 8, 64
    torch.manual_seed(2)
    model = edModelVarTorch(dims, L, dropout=0.0)
    torch.manual_seed(2)
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    X = [torch.randn(B, d, generator=g) for d in dims]
    idx = torch.randint(0, 40, (B,), generator=g)
    corr = orc.p_block(None, idx.numpy(), idx.numpy())
    Fblk = torch.zeros(B, B)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    zs, comb, xhat, mus, lv = model(*[x.cuda() for x in X], corr=corr.cuda())
    eps = [model._engines[B].ws[i]['eps'].cpu().clone() for i in range(2)]     # the Philox draw of this forward
    ls = orc.losses([x.cuda() for x in X], zs, comb, xhat, mus, lv, Fblk.cuda(), 0.7)
    sum(ls).backward()
    noise = {'enc_masks': [(None, None)] * 2, 'dec_masks': [(None, None)] * 2, 'eps': eps}
    st = orc.train_step(P, Bf, None, X, corr, Fblk, noise, 0., 0.7, do_step=False, return_grads=True)
    np.testing.assert_allclose([float(l.detach()) for l in ls], st['losses'], rtol=2e-4, atol=1e-6)
    views = model.inner.layout.views(model.flat.grad)
    for ref, (mine, sl) in model.inner.layout.reference_names().items():
        if orc.is_dead_bias(ref):
            continue
        got = (views[mine] if sl is None else views[mine][sl]).cpu().numpy()
        want = st['grads'][ref].numpy()
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-5 * max(1e-6, float(np.abs(want).max())) + 1e-7,
                                   err_msg=ref)
    before = model.flat.detach().clone()
    torch.nn.utils.clip_grad_norm_(model.parameters(), 1)
    opt.step()
    opt.zero_grad()
    assert not torch.equal(before, model.flat.detach())
    assert torch.equal(model.inner.flat, model.flat.detach())        # the kernels see the optimiser's update
    model.eval()
    out = model(*[x.cuda() for x in X], corr=corr.cuda())
    assert out[0][0].shape == (B, L) and torch.equal(out[0][0], out[3][0])   # eval: zs == mus


def _quiet(fn):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn()


def test_facade_dense_F_pf_ratio_loss_weights_and_replacement(jam):
    """Facade paths that need B x B blocks: dense F (`match_result`) mixed with P through PF_Ratio, loss weights,
    and duplicate indices (min(features) < batch_size -> replace=True, non-identity corr)."""
    rng = np.random.default_rng(11)
    N, dims = 96, (24, 20)
    Z = rng.standard_normal((N, 4))
    data = [Z @ rng.standard_normal((4, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
    Fm = np.abs(rng.standard_normal((N, N))) * (rng.random((N, N)) < .2)
    np.random.seed(1)
    jm = jam.JAMIE(output_dim=4, batch_size=32, epoch_DNN=15, min_epochs=6, pca_dim=None, use_f_tilde=True,
                   match_result=[Fm], PF_Ratio=.5, loss_weights=[1, 2, 3, 4], log_DNN=10 ** 9)
    emb = _quiet(lambda: jm.fit_transform(dataset=data))
    assert jm.sampling_method == 'diag' and emb[0].shape == (N, 4) and np.isfinite(emb[1]).all()
    assert len(jm.loss_history['F']) == 15 and all(np.isfinite(v).all() for v in jm.loss_history.values())
    assert jm.loss_history['F'][-1] > 0          # F != 0: the F loss is live
    # identity P given densely is recognised (no hybrid sampling), and an explicit dense P = I equals P = None
    np.random.seed(1)
    jm2 = jam.JAMIE(output_dim=4, batch_size=32, epoch_DNN=3, min_epochs=6, pca_dim=None, use_f_tilde=False,
                    log_DNN=10 ** 9)
    _quiet(lambda: jm2.fit_transform(dataset=data, P=np.eye(N)))
    assert jm2.sampling_method == 'diag'
    P = np.eye(N); P[0, 0] = 0; P[0, 1] = 1                 # not the identity any more -> partial correspondence
    jm3 = jam.JAMIE(output_dim=4, batch_size=32, epoch_DNN=1, pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9)
    _quiet(lambda: jm3.fit_transform(dataset=data, P=P))
    assert jm3.sampling_method == 'hybrid'
    # use_f_tilde=True without match_result runs stages A / B itself: tests/test_hip_correspondence.py


def test_facade_unequal_rows_zeros_sampler_and_small_n(jam):
    """Unequal row counts -> P = 0 -> 'zeros' sampler with corr = 0; N < batch_size -> batch_size = N
    (reference jamie.py:511-514); PCA preprocessing (`pca_dim`) with the global scaler."""
    rng = np.random.default_rng(12)
    data = [rng.standard_normal((40, 24)), rng.standard_normal((56, 20))]
    np.random.seed(2)
    jm = jam.JAMIE(output_dim=4, batch_size=16, epoch_DNN=4, pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9)
    emb = _quiet(lambda: jm.fit_transform(dataset=data))
    assert jm.sampling_method == 'zeros' and emb[0].shape == (40, 4) and emb[1].shape == (56, 4)
    assert jm.loss_history['CosSim'][-1] == pytest.approx(0.0, abs=1e-6)      # corr = 0 -> comb = z
    data2 = [rng.standard_normal((50, 30)), rng.standard_normal((50, 26))]
    np.random.seed(3)
    jm = jam.JAMIE(output_dim=4, batch_size=512, epoch_DNN=3, pca_dim=[8, None], use_f_tilde=False, log_DNN=10 ** 9)
    emb = _quiet(lambda: jm.fit_transform(dataset=data2))
    assert jm.batch_size == 50 and jm.model.input_dim == [8, 26] and emb[0].shape == (50, 4)
    imp = jm.modal_predict(data2[1], 1)            # back through the PCA inverse of modality 0
    assert imp.shape == (50, 30) and np.isfinite(imp).all()
    assert jm.transform(data2)[0].shape == (50, 4)


def test_facade_partial_correspondence_hybrid_sampler(jam):
    """Partially matched cells (the reference's headline use case): P knows the first half of the pairs.  The
    corrected 'hybrid' sampler draws ~80 % of each batch from the known pairs; alignment of the UNKNOWN half must
    still improve over an untrained model (shared latent structure)."""
    rng = np.random.default_rng(21)
    N, dims = 256, (40, 32)
    Z = rng.standard_normal((N, 4))
    data = [Z @ rng.standard_normal((4, d)) + .05 * rng.standard_normal((N, d)) for d in dims]
    P = np.zeros((N, N))
    P[np.arange(N // 2), np.arange(N // 2)] = 1
    np.random.seed(5)
    jm = jam.JAMIE(output_dim=4, batch_size=64, epoch_DNN=60, min_epochs=20, pca_dim=None, use_f_tilde=False,
                   log_DNN=10 ** 9)
    emb = _quiet(lambda: jm.fit_transform(dataset=data, P=P))
    assert jm.sampling_method == 'hybrid' and jm.num_corr == N // 2
    assert np.isfinite(emb[0]).all() and len(jm.loss_history['CosSim']) == 60
    known = _quiet(lambda: jm.test_closer([e[:N // 2] for e in emb]))
    unknown = _quiet(lambda: jm.test_closer([e[N // 2:] for e in emb]))
    assert known < 0.25 and unknown < 0.45


def test_full_size_config2_step_vs_oracle(jam):
    """BASELINE config 2 at its full per-step size (B = 512, 2000 + 1000 features, latent 32, 40.3 M parameters,
    default dropout 0.6): ONE fp32 step against the CPU oracle with identical explicit noise — losses, every
    gradient tensor (relative L2), the clip norm and the post-step weights; then size-independent properties of
    the same engine: the gradient-norm partials reproduce ||g||, the bias gradients of BN-fed Linears are ~0
    (column sums of a BatchNorm backward), and the BN running statistics follow the momentum rule."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    B, dims, L, p = 512, (2000, 1000), 32, 0.6
    torch.manual_seed(666)
    model = edModelVar(dims, L)
    assert model.dropout == p and model.num_parameters() == 40345130
    torch.manual_seed(666)
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    eng = TrainEngine(model, B)
    opt = orc.Adam(P.values(), 1e-3)
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((B, 16)).astype(np.float32)
    X = [torch.from_numpy(Z @ rng.standard_normal((16, d)).astype(np.float32)
                          + .1 * rng.standard_normal((B, d)).astype(np.float32)) for d in dims]
    X = [(x - x.mean(0)) / x.std(0) for x in X]
    torch.manual_seed(42)
    noise = orc.draw_noise(dims, L, B, p)
    st = orc.train_step(P, Bf, opt, X, torch.eye(B), torch.zeros(B, B), noise, p, 0.5, return_grads=True)
    eng.set_batch([x.cuda() for x in X])
    eng.set_kl_anneal(0.5)
    eng.forward_backward(None, None, _noise_to_dev(noise, p))
    ls, total, _ = eng.read_losses()
    np.testing.assert_allclose(ls, st['losses'], rtol=2e-4, atol=1e-6)
    for ref, (mine, sl) in model.layout.reference_names().items():
        got = (eng.g[mine] if sl is None else eng.g[mine][sl]).cpu().numpy()
        if orc.is_dead_bias(ref):
            assert np.abs(got).max() < 1e-4, ref                      # column sums of a BN backward
            continue
        assert_mostly_close(got, st['grads'][ref].numpy(), rtol=0, atol=0, max_bad_frac=1.0, rel_l2=2e-3, msg=ref)
    gnorm_ref = float(torch.sqrt(sum((g.double() ** 2).sum() for g in st['grads'].values())))
    eng.optimizer_step()
    assert abs(float(torch.sqrt(eng.norm_partials.double().sum())) - gnorm_ref) < 1e-4 * gnorm_ref
    assert abs(float(eng.grad.double().norm()) - gnorm_ref) < 1e-4 * gnorm_ref
    sd = model.state_dict()
    for k, v in P.items():
        if orc.is_dead_bias(k):
            continue
        assert_mostly_close(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-3, atol=2e-5, max_bad_frac=2e-4,
                            rel_l2=1e-4, msg=k)
    for k, v in Bf.items():
        if 'num_batches' in k:
            continue
        np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=1e-4, atol=1e-5, err_msg=k)


def test_fp32_step_on_the_bf16_pipe_stays_with_the_fp32_pipe(jam):
    """fp32 mode, config 2's layer sizes: the first step whose large products run as six bf16 MFMAs on three-piece cuts
    (engine.TUNING['f32_x3'], the default: gemm_f32.hip configuration 20) against the same step on the fp32 matrix pipe
    (configuration 17), weight-gradient matrix by weight-gradient matrix.  Typical distance 1e-6 (relative L2), i.e. fp32
    rounding: the median over the twelve matrices must stay below 5e-6 or three times the yardstick run's median.  A single matrix may be further off -- a pre-activation
    that lands on the other side of LeakyReLU's kink changes one element of dz by a factor of 100, ~1e-3 of its layer's dW; the
    fp32 pipe against ITSELF with other K slices (the yardstick run below) shows the same events (profiles/r05_x3_step_distance.log)
    -- so at most two matrices may exceed 1e-4, none 2e-2.  Losses agree to 1e-6.  Later steps are not compared: Adam's first
    update is lr * sign(g), which turns rounding noise in near-zero gradients into full-size parameter differences in any pair of
    fp32 implementations."""
    from jamie_amd import engine
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    B, dims, L = 512, (2000, 1000), 32
    g = torch.Generator().manual_seed(11)
    X = [torch.randn(B, d, generator=g).cuda() for d in dims]
    out = {}
    try:
        for name, knobs, cfg in (('x3', dict(f32_x3=True, f32_rows=None), engine.F32_CFG_X3),
                                 ('pipe', dict(f32_x3=False, f32_rows=None), engine.F32_CFG_ROWS),
                                 ('pipe_other_slices', dict(f32_x3=False, f32_rows='17:4,4;17:4,4'), engine.F32_CFG_ROWS)):
            engine.tune(**knobs)
            torch.manual_seed(666)
            model = edModelVar(dims, L)
            eng = TrainEngine(model, B, seed=3)
            assert eng.fcfg['enc0'] == cfg
            eng.set_batch(X)
            eng.step()
            torch.cuda.synchronize()
            assert torch.isfinite(eng.grad).all()
            out[name] = ({k: v.double().clone() for k, v in model.layout.views(eng.grad).items() if k.endswith('.W')}, eng.read_losses())
            del eng, model
    finally:
        engine.tune(f32_x3=True, f32_rows=None)

    def distances(a, b):
        return sorted(((a[k] - b[k]).norm() / b[k].norm()).item() for k in b)
    d_x3, d_yard = distances(out['x3'][0], out['pipe'][0]), distances(out['pipe_other_slices'][0], out['pipe'][0])
    print('per-matrix relative L2 distances of the first gradient, sorted\n  bf16 pipe vs fp32 pipe:', ' '.join(f'{v:.1e}' for v in d_x3),
          '\n  fp32 pipe vs itself with other K slices:', ' '.join(f'{v:.1e}' for v in d_yard))
    assert len(d_x3) == 12
    assert d_x3[len(d_x3) // 2] <= max(5e-6, 3 * d_yard[len(d_yard) // 2]), (d_x3, d_yard)
    assert sum(v > 1e-4 for v in d_x3) <= 2 and d_x3[-1] < 2e-2, d_x3
    for a, b in zip(out['x3'][1][0] + [out['x3'][1][1]], out['pipe'][1][0] + [out['pipe'][1][1]]):
        assert abs(a - b) <= 1e-6 * abs(b) + 1e-9, (out['x3'][1], out['pipe'][1])


def test_fp32_weight_gradients_grouped_at_the_end_equal_one_launch_per_layer(jam):
    """fp32, one GPU: the large layers' dW products wait for the end of the backward pass and go out as ONE launch
    (engine.TUNING['f32_dw_group'] = 4: 2560 tiles = 5.0 rounds of the chip's 512 slots instead of 4 x 1.25).  Same tiles,
    same k order, same per-tile sums of squares: gradients, clip norm and the updated parameters equal the
    one-launch-per-layer order (f32_dw_group = 1) bit for bit, at config 2's layer sizes."""
    from jamie_amd import engine
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    B, dims, L = 512, (2000, 1000), 32
    g = torch.Generator().manual_seed(5)
    X = [torch.randn(B, d, generator=g).cuda() for d in dims]
    out = []
    try:
        for group in (1, 4, 2):
            engine.tune(f32_dw_group=group)
            torch.manual_seed(666)
            model = edModelVar(dims, L)
            eng = TrainEngine(model, B, seed=3)
            assert eng._f32_dw_fused
            eng.set_batch(X)
            for _ in range(2):
                eng.step()
            out.append((eng.grad.clone(), eng.norm_partials.clone(), model.flat.clone()))
            del eng, model
    finally:
        engine.tune(f32_dw_group=4)
    for other in out[1:]:
        for a, b in zip(out[0], other):
            assert torch.equal(a, b)


@pytest.mark.parametrize('B,dims,L,p', [(64, (40, 32, 24), 8, 0.0), (256, (264, 200, 136), 16, 0.6),
                                        (96, (56, 44, 36), 10, 0.6), (64, (40, 32, 24, 20), 6, 0.0), (128, (72, 48, 40), 33, 0.6)])
def test_three_modalities_step_vs_generalised_oracle(jam, B, dims, L, p):
    """Three (or four) fully paired modalities (BASELINE config 4's shape class).  The reference has no 3-modality path
    (jamie.py:420), so parity is against the generalised oracle only (combine = sigma-weighted mean, KL rows 0..M-1 of
    the last modality's logvar): one step's losses, gradients and post-step weights.  Latent sizes that are NOT multiples
    of 4 (10, 6, 33: ragged decoder-weight rows in the fused latent launch) are part of the contract ('any 2 <= M <= 4')."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    torch.manual_seed(31)
    model = edModelVar(dims, L, dropout=p)
    torch.manual_seed(31)
    P, Bf = orc.init_state(dims, L)
    sd = model.state_dict()
    for k, v in P.items():
        assert torch.equal(sd[k].cpu(), v), k
        v.requires_grad_(True)
    assert model.num_parameters() == orc.param_count(dims, L)
    eng = TrainEngine(model, B)
    opt = orc.Adam(P.values(), 1e-3)
    g = torch.Generator().manual_seed(2)
    X = [torch.randn(B, d, generator=g) for d in dims]
    torch.manual_seed(77)
    noise = orc.draw_noise(dims, L, B, p)
    st = orc.train_step(P, Bf, opt, X, None, None, noise, p, 0.6, return_grads=True)
    eng.set_batch([x.cuda() for x in X])
    eng.set_kl_anneal(0.6)
    eng.forward_backward(None, None, _noise_to_dev(noise, p))
    ls, total, _ = eng.read_losses()
    np.testing.assert_allclose(ls, st['losses'], rtol=2e-4, atol=1e-6)
    for ref, (mine, sl) in model.layout.reference_names().items():
        if orc.is_dead_bias(ref):
            continue
        got = (eng.g[mine] if sl is None else eng.g[mine][sl]).cpu().numpy()
        want = st['grads'][ref].numpy()
        np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-5 * max(1e-6, float(np.abs(want).max())) + 1e-7,
                                   err_msg=ref)
    eng.optimizer_step()
    sd = model.state_dict()
    for k, v in P.items():
        if not orc.is_dead_bias(k):
            assert_mostly_close(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-3, atol=2e-5, max_bad_frac=1e-3,
                                rel_l2=1e-3, msg=k)
    model.eval()
    out = model(*[x.cuda() for x in X])
    assert len(out[0]) == len(dims) and out[2][2].shape == (B, dims[2])


def test_more_than_two_modalities_reject_latent_above_128(jam):
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    with pytest.raises(ValueError, match='output_dim <= 128'):
        TrainEngine(edModelVar((40, 32, 24), 132, dropout=0.), 32)


def test_three_modalities_facade_and_bf16(jam):
    rng = np.random.default_rng(8)
    N, dims = 512, (72, 48, 40)
    Z = rng.standard_normal((N, 5))
    data = [Z @ rng.standard_normal((5, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
    for mode in ('f32', 'bf16'):
        np.random.seed(3)
        jm = jam.JAMIE(output_dim=8, batch_size=32, epoch_DNN=12, min_epochs=6, pca_dim=None, use_f_tilde=False,
                       log_DNN=10 ** 9, sampler='device', compute_dtype=mode)
        emb = _quiet(lambda: jm.fit_transform(dataset=data))
        assert len(emb) == 3 and all(e.shape == (N, 8) and np.isfinite(e).all() for e in emb)
        assert jm.loss_history['Rec'][-1] < jm.loss_history['Rec'][0]
        assert jm.modal_predict(data[2], 2).shape == (N, dims[0])          # modality 2 -> (2 + 1) % 3 = 0


def test_batch_step_false_accumulates_like_reference(jam):
    """batch_step=False (reference jamie.py:734-749): two batches back-propagate into the same gradient buffers,
    then ONE clip + Adam step.  Engine (accumulate flag, incl. d sigma) against the oracle's summed gradients."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    B, dims, L, p = 128, (96, 72), 8, 0.6
    torch.manual_seed(5)
    model = edModelVar(dims, L, dropout=p)
    torch.manual_seed(5)
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    eng = TrainEngine(model, B)
    opt = orc.Adam(P.values(), 1e-3)
    rng = np.random.default_rng(1)
    total = None
    for b in range(2):
        X = [torch.from_numpy(rng.standard_normal((B, d)).astype(np.float32)) for d in dims]
        torch.manual_seed(77 + b)
        noise = orc.draw_noise(dims, L, B, p)
        st = orc.train_step(P, Bf, opt, X, torch.eye(B), torch.zeros(B, B), noise, p, 0.5, do_step=False, return_grads=True)
        total = st['grads'] if total is None else {k: total[k] + v for k, v in st['grads'].items()}
        for i in range(2):
            eng.ws[i]['x'].copy_(X[i])
        eng.set_kl_anneal(0.5)
        eng.accumulate = b > 0
        eng.forward_backward(None, None, _noise_to_dev(noise, p))
    eng.accumulate = False
    for ref, (mine, sl) in model.layout.reference_names().items():
        if orc.is_dead_bias(ref):
            continue
        got = eng.g[mine] if sl is None else eng.g[mine][sl]
        want = total[ref].numpy()
        scale = max(1e-6, float(np.abs(want).max()))
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=2e-3, atol=2e-5 * scale + 1e-8, err_msg=ref)
    glist = [total[k] for k in P]
    with torch.no_grad():
        orc.clip_grad_norm(glist)
        opt.step(glist)
    eng.optimizer_step()
    sd = model.state_dict()
    for k, v in P.items():
        if not orc.is_dead_bias(k):
            assert_mostly_close(sd[k].cpu().numpy(), v.detach().numpy(), rtol=1e-3, atol=2e-5, max_bad_frac=1e-4,
                                rel_l2=2e-3, msg=k)


def test_facade_batch_step_false_runs_one_step_per_epoch(jam):
    """Facade: batch_step=False trains (loss falls), steps once per epoch and early-stops on the epoch loss."""
    rng = np.random.default_rng(3)
    Z = rng.standard_normal((300, 6))
    data = [(Z @ rng.standard_normal((6, d)) + .1 * rng.standard_normal((300, d))).astype(np.float32) for d in (48, 40)]
    jm = jam.JAMIE(output_dim=8, epoch_DNN=12, batch_size=64, pca_dim=None, use_f_tilde=False, batch_step=False,
                   min_epochs=0, log_DNN=1000, record_loss=True)
    emb = _quiet(lambda: jm.fit_transform(dataset=data))
    assert emb[0].shape == (300, 8) and np.isfinite(emb[0]).all()
    hist = jm.loss_history['Rec']
    assert len(hist) == 12 and hist[-1] < hist[0]
    assert int(jm.engine.state[1]) == 12            # one optimiser step per epoch (4 batches each)


def test_facade_sparse_P_equals_dense_P(jam):
    """Partial correspondence given as a scipy.sparse matrix (CSR lookup on the device, no N x N array) trains exactly
    like the same P given densely: same sampler stream, same B x B blocks."""
    import scipy.sparse as sp
    rng = np.random.default_rng(8)
    N, dims = 320, (40, 32)
    Z = rng.standard_normal((N, 4))
    data = [Z @ rng.standard_normal((4, d)) + .05 * rng.standard_normal((N, d)) for d in dims]
    known = rng.choice(N, N // 3, replace=False)
    P = np.zeros((N, N), dtype=np.float32)
    P[known, known] = 1
    embs = []
    for Pm in (P, sp.coo_matrix(P)):
        np.random.seed(11)
        jm = jam.JAMIE(output_dim=4, batch_size=64, epoch_DNN=6, pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9,
                       dropout=0.0, sampler='numpy')       # (the default sampler would draw the sparse case's rows on the device)
        embs.append(_quiet(lambda: jm.fit_transform(dataset=data, P=Pm)))
        assert jm.sampling_method == 'hybrid' and jm.num_corr == N // 3
    for a, b in zip(*embs):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)
    # a sparse identity is recognised as fully paired data ('diag' sampling, no block at all)
    jm = jam.JAMIE(output_dim=4, batch_size=64, epoch_DNN=1, pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9)
    _quiet(lambda: jm.fit_transform(dataset=data, P=sp.identity(N, format='csr')))
    assert jm.sampling_method == 'diag'


def test_hybrid_assemble_kernel(jam):
    """Device 'hybrid' sampler: the share of paired slots follows true_ratio, paired slots carry a known pair, the others
    the candidate rows; deterministic for a given (seed, step)."""
    from jamie_amd import _native as nv
    B, K = 512, 300
    pairs = torch.stack([torch.arange(K, dtype=torch.int32) * 3, torch.arange(K, dtype=torch.int32) * 5 + 1], 1).cuda().contiguous()
    pidx = torch.randint(0, K, (B,), dtype=torch.int32).cuda()
    r0 = (torch.arange(B, dtype=torch.int32) + 100000).cuda()
    r1 = (torch.arange(B, dtype=torch.int32) + 200000).cuda()
    state = torch.tensor([5, 9, 0, 0], dtype=torch.int64, device='cuda')
    out = []
    for _ in range(2):
        i0, i1 = torch.zeros(B, dtype=torch.int32, device='cuda'), torch.zeros(B, dtype=torch.int32, device='cuda')
        nv.hybrid_assemble(pairs, pidx, r0, r1, K, 0.8, state, 203, i0, i1)
        out.append((i0.cpu(), i1.cpu()))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])
    i0, i1 = out[0]
    paired = i0 < 100000
    assert 0.72 < paired.float().mean().item() < 0.88
    k = pidx.cpu().long()
    assert torch.equal(i0[paired], pairs.cpu()[k[paired], 0]) and torch.equal(i1[paired], pairs.cpu()[k[paired], 1])
    assert torch.equal(i0[~paired], r0.cpu()[~paired]) and torch.equal(i1[~paired], r1.cpu()[~paired])
    state[1] = 10
    nv.hybrid_assemble(pairs, pidx, r0, r1, K, 0.8, state, 203, i0 := torch.zeros(B, dtype=torch.int32, device='cuda'),
                       torch.zeros(B, dtype=torch.int32, device='cuda'))
    assert not torch.equal(i0.cpu() < 100000, paired)            # another step, another choice of slots


@pytest.mark.parametrize('mode', ['f32', 'bf16'])
def test_facade_sparse_P_device_sampler_plan_path(jam, mode):
    """Partial correspondence (sparse P) with sampler='device': hybrid sampling, CSR block lookup and the step are one
    recorded plan; the known pairs pull the two modalities together (the alignment of the PAIRED cells improves)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(8)
    N, dims = 1536, (72, 40)
    Z = rng.standard_normal((N, 5))
    data = [Z @ rng.standard_normal((5, d)) + .05 * rng.standard_normal((N, d)) for d in dims]
    known = np.sort(rng.choice(N, N // 2, replace=False))
    P = sp.csr_matrix((np.ones(len(known), np.float32), (known, known)), shape=(N, N))
    jm = jam.JAMIE(output_dim=8, batch_size=128, epoch_DNN=40, min_epochs=10, pca_dim=None, use_f_tilde=False,
                   log_DNN=10 ** 9, sampler='device', compute_dtype=mode)
    emb = _quiet(lambda: jm.fit_transform(dataset=data, P=P))
    assert jm.sampling_method == 'hybrid' and jm.num_corr == N // 2
    assert int(jm.engine.state[1].item()) == 40 * (N // 128)
    assert all(np.isfinite(v).all() for v in jm.loss_history.values())
    assert jm.loss_history['CosSim'][-1] > 0                 # a general correspondence block: the alignment term is live
    assert jm.loss_history['Rec'][-1] < jm.loss_history['Rec'][0]
    assert jm.test_closer([e[known] for e in emb]) < 0.25    # FOSCTTM of the paired cells


# ---- device-side preprocessing and training checkpoints ----
@pytest.mark.parametrize('dtype', [np.float64, np.float32])
def test_device_standardisation_matches_numpy(jam, dtype):
    """jamie_col_stats / jamie_standardise against `preclass(axis=0)` (utilities.py:654-678): fp64 statistics in
    numpy's two-pass order; a constant feature (0/0 -> NaN -> 0) and a NaN cell behave as in the reference."""
    from jamie_amd import _native as nv
    from jamie_amd.utilities import preclass
    rng = np.random.default_rng(5)
    X = (rng.standard_normal((5000, 203)) * rng.uniform(0.1, 30, 203) + rng.uniform(-5, 5, 203)).astype(dtype)
    X[:, 7] = 3.25                                    # zero variance
    # the device computes in fp64 whatever the input type (a fp32 input is standardised more accurately than numpy's
    # fp32 arithmetic does it): the reference value is the fp64 one
    X64 = X.astype(np.float64)
    pc = preclass(X64, axis=0)
    want = pc.transform(X64.copy()).astype(np.float32)
    out, mean, sd = nv.standardise_columns(torch.from_numpy(X).cuda())
    np.testing.assert_allclose(mean.cpu().numpy(), X64.mean(0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(sd.cpu().numpy(), X64.std(0), rtol=1e-10, atol=1e-12)
    got = out.cpu().numpy()
    assert (got[:, 7] == 0).all()
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-6)
    Xn = X.copy()
    Xn[11, 3] = np.nan                                 # a NaN cell poisons its feature's statistics: all NaN -> 0
    out2, _, _ = nv.standardise_columns(torch.from_numpy(Xn).cuda())
    assert (out2[:, 3] == 0).all() and torch.isfinite(out2).all()


def test_facade_device_preprocessing_equals_host(jam):
    import io
    import contextlib
    rng = np.random.default_rng(8)
    N, dims = 700, (72, 40)
    Z = rng.standard_normal((N, 5))
    data = [3.0 * (Z @ rng.standard_normal((5, d))) + rng.standard_normal((N, d)) + 2.0 for d in dims]
    embs, pres = [], []
    for mode in ('host', 'device'):
        with contextlib.redirect_stdout(io.StringIO()):
            jm = jam.JAMIE(output_dim=8, batch_size=64, epoch_DNN=6, min_epochs=3, pca_dim=None, use_f_tilde=False,
                           log_DNN=10 ** 9, sampler='device', preprocess=mode)
            embs.append(jm.fit_transform(dataset=[d.copy() for d in data]))
        pres.append(jm)
    for a, b in zip(*embs):
        np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4)
    x_new = data[0][:50]
    np.testing.assert_allclose(pres[0].transform_one(x_new, 0), pres[1].transform_one(x_new, 0), rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(pres[0].modal_predict(x_new, 0), pres[1].modal_predict(x_new, 0), rtol=1e-3, atol=1e-3)


@pytest.mark.parametrize('sampler,mode', [('device', 'f32'), ('device', 'bf16'), ('numpy', 'f32')])
def test_checkpoint_resume_is_bit_identical(jam, tmp_path, sampler, mode):
    """8 epochs straight == 5 epochs + checkpoint + resume for 3 more: weights, BN statistics, loss history."""
    import io
    import contextlib
    rng = np.random.default_rng(4)
    N, dims = 640, (72, 40)
    Z = rng.standard_normal((N, 5))
    data = [Z @ rng.standard_normal((5, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
    kw = dict(output_dim=8, batch_size=64, min_epochs=3, pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9,
              sampler=sampler, compute_dtype=mode)
    ck = str(tmp_path / 'train.ckpt')
    with contextlib.redirect_stdout(io.StringIO()):
        np.random.seed(7)        # the reference's numpy sampler draws from numpy's global stream (never seeded by it)
        full = jam.JAMIE(epoch_DNN=8, **kw)
        e_full = full.fit_transform(dataset=data)
        np.random.seed(7)
        part = jam.JAMIE(epoch_DNN=5, checkpoint_path=ck, checkpoint_every=5, **kw)
        part.fit_transform(dataset=data)
        res = jam.JAMIE(epoch_DNN=8, **kw)
        e_res = res.fit_transform(dataset=data, resume_from=ck)
    assert torch.equal(full.model.flat, res.model.flat)
    sa, sb = full.model.state_dict(), res.model.state_dict()
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k
    assert full.loss_history == res.loss_history and len(res.loss_history['Rec']) == 8
    for a, b in zip(e_full, e_res):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        with contextlib.redirect_stdout(io.StringIO()):
            jam.JAMIE(epoch_DNN=8, **{**kw, 'output_dim': 4}).fit_transform(dataset=data, resume_from=ck)


@pytest.mark.parametrize('grad_bf16', [False, True])
def test_fused_gradient_norm_equals_the_norm_of_the_gradient(jam, grad_bf16):
    """bf16 single-GPU mode: the dW launches emit per-tile sums of squares and jamie_grad_sqnorm_ranges adds the rest;
    together they are ||g||^2 of the whole flat gradient (clip_grad_norm_, jamie.py:739), also when gradients
    accumulate (batch_step=False) and after a fall-back to the one-pass kernel.  With the weight gradients stored as bf16
    (`grad_bf16`, the default) the partials are still the squares of the fp32 accumulators: the norm agrees with the
    norm of the ROUNDED buffer to the rounding (sum of delta^2: ~1e-6 relative) and the bf16 buffer is the rounded fp32
    gradient bit for bit."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    dims, L, B = (520, 264), 32, 512
    torch.manual_seed(3)
    model = edModelVar(dims, L)
    eng = TrainEngine(model, B, compute_dtype='bf16', grad_bf16=grad_bf16)
    assert eng.fused_norm and eng.grad_bf16 == grad_bf16
    g = torch.Generator().manual_seed(1)
    for step in range(4 if not grad_bf16 else 2):
        eng.set_batch([torch.randn(B, d, generator=g).cuda() for d in dims])
        eng.accumulate = step == 2                      # step 2 adds its gradient to step 1's buffer
        fall_back = step == 3
        # (a backward pass that is to be followed by a gradient exchange emits no partial sums: one pass after the reduction)
        eng.forward_backward(allreduce=(lambda flat: None) if fall_back else None)
        want = float(eng.grad_flat().double().norm())
        eng.optimizer_step()
        n_live = eng.n_norm if fall_back else eng.n_dw_partials + eng.sq_ranges.blocks
        got = float(torch.sqrt(eng.norm_partials[:n_live].double().sum()))
        assert abs(got - want) < (5e-6 if grad_bf16 else 2e-6) * want, (step, got, want)
    assert int(eng.state[1].item()) == (2 if grad_bf16 else 4)
    if grad_bf16:
        # the same batch through an engine with the fp32 gradient buffer: grad16 == bf16(grad), element by element
        torch.manual_seed(3)
        model2 = edModelVar(dims, L)
        eng2 = TrainEngine(model2, B, compute_dtype='bf16', grad_bf16=False)
        torch.manual_seed(3)
        model3 = edModelVar(dims, L)
        eng3 = TrainEngine(model3, B, compute_dtype='bf16', grad_bf16=True)
        X = [torch.randn(B, d, generator=g).cuda() for d in dims]
        for e in (eng2, eng3):
            e.set_batch(X)
            e.forward_backward()
            e.optimizer_step()
        for key in eng3.dw_partial:
            a = eng3.g16[key + '.W']
            b = eng2.g[key + '.W'].to(torch.bfloat16)
            assert torch.equal(a, b), key
        for key, (o, shp) in model3.layout.entries.items():
            if key[:-2] not in eng3.dw_partial:            # the small ranges: bf16 copies of the fp32 values
                n = int(np.prod(shp))
                assert torch.equal(eng3.grad16[o:o + n], eng3.grad[o:o + n].to(torch.bfloat16)), key
        with pytest.raises(Exception):
            eng3.forward_backward()
            eng3.accumulate = True
            eng3.forward_backward()                         # accumulating onto bf16-written gradients is refused
