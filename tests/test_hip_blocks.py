"""B x B correspondence blocks from DENSE matrices and their mix (reference jamie/jamie.py:586-604) and the eval-mode forward
with a general correspondence block (jamie/model.py:245-275), against the oracle.
Run on the MI355X box:  pytest -m gpu"""
import numpy as np
import pytest
import torch

from oracle import jamie_oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def nv():
    from jamie_amd import _native
    _native.require_gpu()
    return _native


@pytest.mark.parametrize('N0,N1,B,off', [(300, 300, 64, (0, 0)), (517, 431, 200, (0, 0)), (900, 700, 128, (250, 100))])
def test_dense_block_and_mix_match_oracle(nv, N0, N1, B, off):
    """jamie_dense_block = rownorm(M[idx0][:, idx1]) (zero rows keep divisor 1), jamie_axpby = PF * P + (1 - PF) * F: the
    oracle's p_block / f_block and the reference's mix; `off` = the row / column offsets of a data-parallel shard."""
    rng = np.random.default_rng(N0 + B)
    P = (rng.random((N0, N1)) < 0.02).astype(np.float32) * rng.random((N0, N1)).astype(np.float32)
    F = np.abs(rng.standard_normal((N0, N1))).astype(np.float32) * (rng.random((N0, N1)) < 0.3)
    P[5] = 0                                                    # a zero row: divisor 1
    idx0 = rng.integers(0, N0 - off[0], B)
    idx1 = rng.integers(0, N1 - off[1], B)
    idx0[3] = 5 - off[0] if off[0] <= 5 else idx0[3]
    d = lambda a, dt=torch.float32: torch.as_tensor(a).to('cuda', dt).contiguous()      # noqa: E731
    i0, i1 = d(idx0, torch.int32), d(idx1, torch.int32)
    Pd, Fd = d(P), d(F)
    pb, fb, mix = (torch.empty(B, B, device='cuda') for _ in range(3))
    nv.dense_block(Pd, i0, i1, pb, off[0], off[1])
    nv.dense_block(Fd, i0, i1, fb, off[0], off[1])
    want_p = orc.p_block(torch.from_numpy(P), idx0 + off[0], idx1 + off[1])
    want_f = orc.f_block(torch.from_numpy(F), idx0 + off[0], idx1 + off[1], B)
    np.testing.assert_allclose(pb.cpu().numpy(), want_p.numpy(), rtol=2e-6, atol=0)
    np.testing.assert_allclose(fb.cpu().numpy(), want_f.numpy(), rtol=2e-6, atol=0)
    nv.axpby(mix, 0.3, pb, 0.7, fb)
    np.testing.assert_allclose(mix.cpu().numpy(), (0.3 * want_p + 0.7 * want_f).numpy(), rtol=3e-6, atol=1e-9)
    nv.axpby(mix, 0.3, pb)
    np.testing.assert_allclose(mix.cpu().numpy(), (0.3 * want_p).numpy(), rtol=3e-6, atol=0)
    raw = torch.empty(B, B, device='cuda')
    nv.dense_block(Pd, i0, i1, raw, off[0], off[1], normalise=False)
    assert torch.equal(raw.cpu(), torch.from_numpy(P)[idx0 + off[0]][:, idx1 + off[1]])
    both = torch.empty(B, B, device='cuda')
    nv.dense_block(Fd, i0, i1, both, off[0], off[1], w_blk=0.7, add=pb, w_add=0.3)
    np.testing.assert_allclose(both.cpu().numpy(), (0.3 * want_p + 0.7 * want_f).numpy(), rtol=3e-6, atol=1e-9)


def test_eval_forward_with_general_corr_matches_oracle(nv):
    """`model(*X, corr=C)` in eval mode (zs = mus, combined with a general block, decoded): reference model.py:264-275."""
    from jamie_amd.model import edModelVar
    dims, L, n = (96, 72), 8, 160
    torch.manual_seed(3)
    model = edModelVar(dims, L)
    torch.manual_seed(3)
    P, Bf = orc.init_state(dims, L)
    g = torch.Generator().manual_seed(1)
    for k, v in Bf.items():
        if k.endswith('running_mean'):
            v.copy_(0.1 * torch.randn(v.shape, generator=g))
        elif k.endswith('running_var'):
            v.copy_(0.5 + torch.rand(v.shape, generator=g))
    full = dict(P)
    full.update(Bf)
    model.load_state_dict(full)
    model.eval()
    X = [torch.randn(n, d, generator=g) for d in dims]
    corr = orc.row_normalise((torch.rand(n, n, generator=g) < 0.05).float() * torch.rand(n, n, generator=g))
    with torch.no_grad():
        zs, comb, xh, mus, lv = orc.forward(P, Bf, X, corr, train=False)
    got = model(*[x.cuda() for x in X], corr=corr.cuda())
    for i in range(2):
        np.testing.assert_allclose(got[0][i].cpu().numpy(), zs[i].numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(got[1][i].cpu().numpy(), comb[i].numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(got[2][i].cpu().numpy(), xh[i].numpy(), rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(got[4].cpu().numpy(), lv.numpy(), rtol=1e-4, atol=1e-5)
