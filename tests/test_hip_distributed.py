"""Data-parallel path with two ranks sharing cuda:0 over gloo (everything except RCCL itself; the real multi-GPU path is
backend "nccl" = RCCL, one GPU per rank): SURVEY.md §8(e).
  * N ranks fed IDENTICAL batches and seeds reproduce the 1-rank weights (fp32 messages: bit for bit; bf16 messages:
    within bf16 rounding of the gradient) -- agreement between the ranks alone would also hold with a wrong 1/world
    scale or a dropped region;
  * the facade with distributed=True: uneven shards, early stopping and BatchNorm statistics are rank-invariant.
Run on the MI355X box:  pytest -m gpu"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(script, port, nproc=2, timeout=600):
    env = dict(os.environ, JAMIE_DIST_BACKEND='gloo', JAMIE_SHARE_GPU='1', MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(nproc),
                        '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                       capture_output=True, text=True, env=env, timeout=timeout)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize('mode', ['f32', 'bf16_msgs', 'bf16_f32msgs', 'bf16_msgs_direct'])
def test_two_ranks_with_identical_batches_reproduce_one_rank(tmp_path, mode):
    script = tmp_path / 'same.py'
    script.write_text(f'''
import sys, torch, numpy as np
sys.path.insert(0, {ROOT!r})
from jamie_amd import distributed as jd, _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
rank, world, local = jd.init_from_env()
dev = torch.device('cuda', local)
mode = {mode!r}
# (_direct: layers large enough for the large-tile dW launches, whose epilogues then write bf16 straight into the message buffer)
dims, L, B, N = ((328, 264) if mode == 'bf16_msgs_direct' else (264, 136)), 8, 128, 1024
g = torch.Generator(device=dev).manual_seed(50)          # the SAME cells on every rank
data = [torch.randn(N, d, generator=g, device=dev) for d in dims]
compute = 'f32' if mode == 'f32' else 'bf16'
comm = torch.bfloat16 if mode in ('bf16_msgs', 'bf16_msgs_direct') else None
flats = {{}}
for label, w in (('one', 1), ('dp', world)):
    torch.manual_seed(3)
    model = edModelVar(dims, L, device=dev)
    eng = TrainEngine(model, B, seed=11, world_size=w, compute_dtype=compute)      # the SAME seed on every rank
    ar = jd.OverlappedGradAllReduce(min_bytes=1 << 16, comm_dtype=comm) if w > 1 else None
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    for s in range(4):
        nv.sample_indices(idx, N, 0, False, eng.state, 200)
        eng.load_batch(data, [idx, idx])
        eng.step(None, None, None, ar)
    flats[label] = model.flat.clone()
    assert int(eng.state[1].item()) == 4
    assert eng._direct_now == (w > 1 and mode == 'bf16_msgs_direct'), (label, eng._direct_now)
a, b = flats['one'], flats['dp']
d = (a - b).abs()
if mode == 'f32':
    # sum of two equal fp32 gradients times 1/2 is the gradient, and the norm comes from the same one-pass kernel
    assert torch.equal(a, b), float(d.max())
elif mode == 'bf16_f32msgs':
    # fp32 messages: the same gradient; only the norm's summation order differs (dW-epilogue partials on one GPU, the
    # one-pass kernel on the reduced gradient): the clip coefficient's last bits
    assert float(d.max()) < 1e-6, float(d.max())
else:
    # bf16 messages round the gradient to 8 bits: Adam's normalised update moves by ~2^-9 relative, except where a
    # gradient is rounding noise anyway (sign flips, bounded by 2 lr per step)
    frac = float((d > 4 * 1e-3 * 2.0 ** -8 * 4).float().mean())
    assert float(d.max()) <= 4 * 2e-3 + 1e-6 and frac < 5e-2, (float(d.max()), frac)
print('SAME OK', rank, float(d.max()))
torch.distributed.destroy_process_group()
''')
    out = _torchrun(script, 29581)
    assert out.count('SAME OK') == 2, out[-2000:]


@pytest.mark.parametrize('mode', ['f32', 'bf16_f32msgs', 'bf16_msgs_direct', 'f32_4ranks'])
def test_sharded_optimizer_matches_replicated_optimizer(tmp_path, mode):
    """distributed.ShardedGradExchange (reduce-scatter of the large regions, clip + Adam over this rank's packed pieces, all-gather
    of the updated weights under the next forward pass) against the all-reduce exchange with the full update on every rank:
    two ranks with DIFFERENT cells, 4 eager steps + 3 replayed plan steps.  The two arrangements sum the same gradients; only the
    order in which the squared norm is summed differs (pieces + rep against one pass), i.e. the clip coefficient's last bits:
    after one step (and gather_sharded_state()) parameters and both Adam moments agree to that; after seven, everywhere but on
    the parameters whose gradient is rounding noise.  The two ranks hold identical parameters throughout."""
    script = tmp_path / 'sharded.py'
    script.write_text(f'''
import sys, torch, numpy as np
sys.path.insert(0, {ROOT!r})
from jamie_amd import distributed as jd, _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
rank, world, local = jd.init_from_env()
dev = torch.device('cuda', local)
mode = {mode!r}.split('_4')[0]
dims, L, B, N = ((264, 136) if mode == 'f32' else (328, 264)), 8, 128, 1024      # (bf16: the large-tile launches, no transposed W)
g = torch.Generator(device=dev).manual_seed(50 + rank)          # different cells on every rank
data = [torch.randn(N, d, generator=g, device=dev) for d in dims]
compute = 'f32' if mode == 'f32' else 'bf16'
comm = torch.bfloat16 if mode == 'bf16_msgs_direct' else None
res, first = {{}}, {{}}
for label in ('replicated', 'sharded'):
    torch.manual_seed(3)
    model = edModelVar(dims, L, device=dev)
    eng = TrainEngine(model, B, seed=11, world_size=world, compute_dtype=compute)
    if label == 'sharded':
        ar = jd.ShardedGradExchange(comm_dtype=comm)
        eng.enable_sharded_optimizer(ar)
        assert eng._zs['S'] * world == sum(hi - lo for k, (lo, hi) in model.layout.regions.items() if k != 'rep')
    else:
        ar = jd.OverlappedGradAllReduce(min_bytes=1 << 16, comm_dtype=comm)
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    for s in range(4):
        nv.sample_indices(idx, N, 0, False, eng.state, 200)
        eng.load_batch(data, [idx, idx])
        eng.step(None, None, None, ar)
        if s == 0:                      # (a flush in the middle of training: the run goes on from the gathered state)
            eng.flush(collective=True)
            first[label] = (model.flat.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone())
    assert eng._direct_now == (mode == 'bf16_msgs_direct'), (label, eng._direct_now)
    plan = eng.make_plan(data, idx, N, False, ar)
    for s in range(2):
        eng.run_plan(plan)
    eng.flush(collective=True)
    torch.cuda.synchronize()
    assert int(eng.state[1].item()) == 7
    res[label] = (model.flat.clone(), eng.exp_avg.clone(), eng.exp_avg_sq.clone(), eng.read_losses()[1])
    both = [torch.zeros_like(model.flat) for _ in range(world)]
    torch.distributed.all_gather(both, model.flat)
    assert all(torch.equal(both[0], b) for b in both[1:]), (label, float((both[0] - both[-1]).abs().max()))
    if label == 'sharded' and compute == 'bf16':            # the bf16 weight copy every rank multiplies with = the master, rounded
        lo = model.layout.regions['enc0'][0]
        assert torch.equal(eng.wbf_flat[lo:], model.flat[lo:].to(torch.bfloat16))
# after ONE step the two arrangements differ by the clip coefficient's last bits alone
for k, (a, b) in enumerate(zip(first['replicated'], first['sharded'])):
    d = (a - b).abs()
    # (two ranks: a + b is the same sum in any order; with four the association of a sum depends on where a message is cut -- the
    #  replicated exchange sends the last layer's gradient in two messages, the sharded one reduce-scatters whole regions -- and
    #  Adam's first step turns a last-bit difference of a rounding-noise gradient into up to ~1e-6)
    tol = (2e-9 if world == 2 else 2e-6) if k == 0 else 2e-6 * float(a.abs().max())
    assert float(d.max()) <= tol, ('first step', k, float(d.max()), tol)
# ... and after seven they still agree except where a gradient is rounding noise anyway (the biases in front of a BatchNorm:
# Adam's normalised update of pure noise flips sign with the last bit of an activation; bounded by 2 lr per step)
a, b = res['replicated'][0], res['sharded'][0]
d = (a - b).abs()
if mode == 'f32':
    frac = float((d > 1e-6).float().mean())
    assert float(d.max()) <= 7 * 2e-3 and frac < 2e-3, (float(d.max()), frac)
    assert abs(res['replicated'][3] - res['sharded'][3]) <= 1e-4 * abs(res['replicated'][3])
else:
    # bf16 products: a last-bit difference in a weight flips bf16 roundings downstream, i.e. the two runs are two samples of the
    # same bf16 rounding noise (gradients differ by ~2^-9 relative, Adam's normalised update by that times lr per step)
    frac = float((d > 7 * 1e-3 * 2.0 ** -8 * 4).float().mean())
    assert float(d.max()) <= 7 * 2e-3 and frac < 5e-2, (float(d.max()), frac)
    assert abs(res['replicated'][3] - res['sharded'][3]) <= 2e-2 * abs(res['replicated'][3])
print('SHARDED OK', rank)
torch.distributed.destroy_process_group()
''')
    n = 4 if mode.endswith('4ranks') else 2
    out = _torchrun(script, 29583, nproc=n)
    assert out.count('SHARDED OK') == n, out[-2000:]


@pytest.mark.parametrize('variant', ['diag_numpy', 'diag_numpy_default', 'diag_device_bf16', 'hybrid_sparse'])
def test_facade_distributed_is_rank_invariant(tmp_path, variant):
    """JAMIE(distributed=True) with two ranks: 255 cells -> shards of 128 and 127 rows, i.e. 2 vs 1 batches of 64 per epoch
    if every rank used its own shard size (mismatched collective counts = deadlock); early stopping is decided on the
    rank-mean loss; BatchNorm running statistics are averaged before evaluation.  Both ranks must finish, after the same
    number of epochs, with identical parameters, statistics and embeddings."""
    script = tmp_path / 'facade.py'
    script.write_text(f'''
import sys, io, contextlib, torch, numpy as np
sys.path.insert(0, {ROOT!r})
import scipy.sparse as sp
import jamie_amd
from jamie_amd import distributed as jd
variant = {variant!r}
rank, world, local = jd.init_from_env()
rng = np.random.default_rng(0)
N, dims = 255, (48, 40)
Z = rng.standard_normal((N, 5))
data = [Z @ rng.standard_normal((5, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
np.random.seed(7 + rank)
kw = dict(output_dim=8, batch_size=64, epoch_DNN=60, min_epochs=8, min_increment=0.05, max_steps_without_increment=3,
          pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9, distributed=True)
if variant != 'diag_numpy_default':            # (the facade's default is the replicated optimiser: north_star's one all-reduce)
    kw.update(dp_optimizer='auto')
P = None
if variant == 'diag_device_bf16':
    kw.update(sampler='device', compute_dtype='bf16')
elif variant == 'hybrid_sparse':
    known = np.arange(0, N, 2)                                   # every other cell has a known partner (itself)
    P = sp.csr_matrix((np.ones(len(known), np.float32), (known, known)), shape=(N, N))
jm = jamie_amd.JAMIE(**kw)
with contextlib.redirect_stdout(io.StringIO()):
    emb = jm.fit_transform(dataset=data, P=P)
epochs = len(jm.loss_history['Rec'])
assert 9 <= epochs <= 60 and all(e.shape == (N, 8) and np.isfinite(e).all() for e in emb)
dev = jm.model.device
sig = torch.cat([jm.model.flat, jm.model.bn_flat, torch.from_numpy(np.concatenate(emb, 1).ravel()).to(dev),
                 torch.tensor([float(epochs)], device=dev)])
others = [torch.zeros_like(sig) for _ in range(world)]
torch.distributed.all_gather(others, sig)
assert torch.equal(others[0], others[1]), float((others[0] - others[1]).abs().max())
opt = 'sharded' if jm.engine._zs is not None else 'replicated'
# (dp_optimizer='auto': sharded, except in bf16 mode at these small sizes, where the products read transposed weight copies)
assert opt == ('replicated' if variant in ('diag_device_bf16', 'diag_numpy_default') else 'sharded'), opt
print('FACADE OK', rank, epochs, jm.sampling_method, opt)
torch.distributed.destroy_process_group()
''')
    out = _torchrun(script, 29582)
    assert out.count('FACADE OK') == 2, out[-2000:]
    eps = {ln.split()[3] for ln in out.splitlines() if ln.startswith('FACADE OK')}
    assert len(eps) == 1


def test_facade_distributed_sharded_checkpoint_resume_is_bit_identical(tmp_path):
    """Two ranks, sharded optimiser (`dp_optimizer='auto'`): 6 epochs with a checkpoint after the 4th ==
    4 epochs + checkpoint + resume for 2 more.  The checkpoint is written by rank 0 from the replicated buffers after every rank has gathered its pieces
    (parameters, both Adam moments); on resume the packed pieces are cut from the restored buffers."""
    script = tmp_path / 'resume.py'
    script.write_text(f'''
import sys, io, contextlib, torch, numpy as np
sys.path.insert(0, {ROOT!r})
import jamie_amd
from jamie_amd import distributed as jd
rank, world, local = jd.init_from_env()
rng = np.random.default_rng(4)
N, dims = 640, (72, 40)
Z = rng.standard_normal((N, 5))
data = [Z @ rng.standard_normal((5, d)) + .1 * rng.standard_normal((N, d)) for d in dims]
kw = dict(output_dim=8, batch_size=64, min_epochs=3, pca_dim=None, use_f_tilde=False, log_DNN=10 ** 9, sampler='device',
          distributed=True, use_early_stop=False, dp_optimizer='auto')
ck = {str(tmp_path / 'dp.ckpt')!r}
with contextlib.redirect_stdout(io.StringIO()):
    # (the straight run checkpoints at the same epoch: a checkpoint averages the per-rank BatchNorm statistics in place)
    full = jamie_amd.JAMIE(epoch_DNN=6, checkpoint_path=ck + '.full', checkpoint_every=4, **kw)
    e_full = full.fit_transform(dataset=data)
    part = jamie_amd.JAMIE(epoch_DNN=4, checkpoint_path=ck, checkpoint_every=4, **kw)
    part.fit_transform(dataset=data)
    torch.distributed.barrier()
    res = jamie_amd.JAMIE(epoch_DNN=6, **kw)
    e_res = res.fit_transform(dataset=data, resume_from=ck)
assert full.engine._zs is not None and res.engine._zs is not None
assert torch.equal(full.model.flat, res.model.flat), float((full.model.flat - res.model.flat).abs().max())
assert torch.equal(full.engine.exp_avg, res.engine.exp_avg) and torch.equal(full.engine.exp_avg_sq, res.engine.exp_avg_sq)
assert len(res.loss_history['Rec']) == 6
if rank == 0:                       # (the checkpoint carries rank 0's loss history; the losses are per-shard)
    assert full.loss_history == res.loss_history
assert torch.equal(full.model.bn_flat, res.model.bn_flat)
for a, b in zip(e_full, e_res):
    assert np.array_equal(a, b)
print('RESUME OK', rank)
torch.distributed.destroy_process_group()
''')
    out = _torchrun(script, 29584)
    assert out.count('RESUME OK') == 2, out[-2000:]


@pytest.mark.parametrize('mode', ['f32', 'bf16_msgs_direct'])
def test_sharded_exchange_on_rccl_with_one_rank(tmp_path, mode):
    """The sharded optimiser's exchange through RCCL ITSELF (backend "nccl"), which a one-GPU box only allows with one rank:
    reduce_scatter_tensor into the packed gradient piece, the all-reduce of `rep` and of the norm partials, all_gather_into_tensor
    of the updated weights into the (bf16 / fp32) weight buffer, asynchronous works waited for from recorded plans -- every
    collective degenerates to a copy on RCCL's stream, the calls, dtypes, views and stream hand-offs are the 8-GPU job's.  Against
    the plain one-GPU engine: the same gradients, the squared norm summed in another order."""
    script = tmp_path / 'one.py'
    script.write_text(f'''
import os, sys, torch, numpy as np
sys.path.insert(0, {ROOT!r})
from jamie_amd import distributed as jd, _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
os.environ.pop('JAMIE_DIST_BACKEND', None)
os.environ['WORLD_SIZE'] = '2'                      # (init_from_env skips a one-rank world)
torch.cuda.set_device(0)
torch.distributed.init_process_group('nccl', rank=0, world_size=1)
assert torch.distributed.get_backend() == 'nccl'
dev = torch.device('cuda', 0)
mode = {mode!r}
dims, L, B, N = ((264, 136) if mode == 'f32' else (328, 264)), 8, 128, 1024
g = torch.Generator(device=dev).manual_seed(50)
data = [torch.randn(N, d, generator=g, device=dev) for d in dims]
compute = 'f32' if mode == 'f32' else 'bf16'
res = {{}}
for label in ('plain', 'sharded'):
    torch.manual_seed(3)
    model = edModelVar(dims, L, device=dev)
    eng = TrainEngine(model, B, seed=11, compute_dtype=compute, grad_bf16=False)
    ar = None
    if label == 'sharded':
        ar = jd.ShardedGradExchange(comm_dtype=torch.bfloat16 if mode != 'f32' else None, single_rank_ok=True)
        assert ar.single and ar.world == 1
        eng.enable_sharded_optimizer(ar)
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    for s in range(2):
        nv.sample_indices(idx, N, 0, False, eng.state, 200)
        eng.load_batch(data, [idx, idx])
        eng.step(None, None, None, ar)
    if label == 'sharded':
        assert eng._direct_now == (mode != 'f32')
    plan = eng.make_plan(data, idx, N, False, ar)
    for s in range(2):
        eng.run_plan(plan)
    if label == 'sharded':              # a rank-LOCAL flush must not start collectives by itself (ADVICE r3): it raises while stale
        try:
            eng.flush()
            raise AssertionError('local flush() with stale sharded state did not raise')
        except nv.JamieHipError as err:
            assert 'EVERY rank' in str(err)
    eng.flush(collective=True)
    eng.flush()                         # (nothing stale any more: local and silent)
    torch.cuda.synchronize()
    assert int(eng.state[1].item()) == 5
    res[label] = (model.flat.clone(), eng.exp_avg.clone(), eng.read_losses()[1])
a, b = res['plain'][0], res['sharded'][0]
d = (a - b).abs()
if mode == 'f32':
    assert float(d.max()) <= 5 * 2e-3 and float((d > 1e-6).float().mean()) < 2e-3, (float(d.max()), float((d > 1e-6).float().mean()))
    assert abs(res['plain'][2] - res['sharded'][2]) <= 1e-4 * abs(res['plain'][2])
else:       # bf16 messages round the gradient once more than the plain engine's fp32 gradient buffer
    frac = float((d > 5 * 1e-3 * 2.0 ** -8 * 4).float().mean())
    assert float(d.max()) <= 5 * 2e-3 + 1e-6 and frac < 5e-2, (float(d.max()), frac)
    assert abs(res['plain'][2] - res['sharded'][2]) <= 2e-2 * abs(res['plain'][2])
print('RCCL ONE OK', float(d.max()))
torch.distributed.destroy_process_group()
''')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29586')
    env.pop('JAMIE_DIST_BACKEND', None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert 'RCCL ONE OK' in r.stdout


def test_native_collectives_one_rank_rccl(tmp_path):
    """jamie_allreduce / jamie_reduce_scatter / jamie_all_gather / jamie_comm_wait (csrc/comm.hip: RCCL behind the C ABI, the
    communicator created with an id broadcast through the torch.distributed group) in a one-rank RCCL group -- all a one-GPU box
    allows: every collective is then a copy, but the binding, the communicator, the stream / event hand-off, the dtypes and the
    replay from a recorded plan are the N-GPU job's.  Then the replicated exchange (north_star's ONE all-reduce of the gradient)
    end to end through it, against the same steps through torch.distributed: identical parameters."""
    script = tmp_path / 'native.py'
    script.write_text(f'''
import os, sys, torch
sys.path.insert(0, {ROOT!r})
from jamie_amd import distributed as jd, _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
os.environ.pop('JAMIE_DIST_BACKEND', None)
torch.cuda.set_device(0)
torch.distributed.init_process_group('nccl', rank=0, world_size=1)
dev = torch.device('cuda', 0)
assert jd.NativeComm.available() and nv.comm_version() >= 20000
nc = jd.NativeComm()
g = torch.Generator(device=dev).manual_seed(1)
for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(100003, generator=g, device=dev).to(dt)
    y = x.clone()
    nc.all_reduce(y).wait()
    out = torch.zeros(4096, device=dev, dtype=dt)
    nc.reduce_scatter(out, x[:4096]).wait()
    full = torch.zeros(4096, device=dev, dtype=dt)
    nc.all_gather(full, x[4096:8192]).wait()
    torch.cuda.synchronize()
    assert torch.equal(y, x) and torch.equal(out, x[:4096]) and torch.equal(full, x[4096:8192])
# recorded and replayed like kernel launches
z = torch.ones(1024, device=dev)
nv.begin_record()
w = nc.all_reduce(z)
w.wait()
plan = nv.end_record()
z.mul_(3.0)
nv.replay(plan)
torch.cuda.synchronize()
assert float(z.sum()) == 3072.0 and len(plan) == 2
try:
    nv.comm_allreduce(nc.h, torch.zeros(4, device=dev, dtype=torch.float16), 0)
    raise AssertionError('fp16 accepted')
except nv.JamieHipError:
    pass
# the replicated exchange through the C ABI == through torch.distributed
dims, L, B, N = (328, 264), 8, 128, 1024
data = [torch.randn(N, d, generator=g, device=dev) for d in dims]
res = {{}}
for native in (True, False):
    torch.manual_seed(3)
    model = edModelVar(dims, L, device=dev)
    eng = TrainEngine(model, B, seed=11, compute_dtype='bf16')
    ar = jd.OverlappedGradAllReduce(comm_dtype=torch.bfloat16, single_rank_ok=True, native=native, min_bytes=1 << 16)
    assert (ar.native is not None) == native and ar.single
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    plan = eng.make_plan(data, idx, N, False, ar)
    for s in range(3):
        eng.run_plan(plan)
    assert eng._direct_now
    torch.cuda.synchronize()
    res[native] = (model.flat.clone(), eng.read_losses()[1])
assert torch.equal(res[True][0], res[False][0]) and res[True][1] == res[False][1]
nc.close()
print('NATIVE COMM OK')
torch.distributed.destroy_process_group()
''')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29588')
    env.pop('JAMIE_DIST_BACKEND', None)
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    assert 'NATIVE COMM OK' in r.stdout


def test_two_ranks_over_rccl_when_two_gpus_are_present(tmp_path):
    """The real multi-GPU path -- backend "nccl" = RCCL, one GPU per rank -- end to end through `bench.py --gpus 2` (which starts its
    own ranks): runs wherever the box has two or more GPUs, skipped on the one-GPU boxes of the pool.  Both
    data-parallel arrangements and both message dtypes are timed in the one line; the ranks must agree on a finite loss."""
    import json
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip('one GPU visible: RCCL with more than one rank needs a GPU per rank')
    clean = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'JAMIE_DIST_BACKEND',
                                                               'JAMIE_SHARE_GPU')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '20', '--warmup', '5'],
                       capture_output=True, text=True, env=clean, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{')][0])
    assert out['n_gpus'] == 2 and out['rccl']['backend'] == 'nccl' and out['rccl']['world'] == 2
    assert out['config']['dp_optimizer'] == 'replicated' and out['value'] > 0 and out['final_loss'] == out['final_loss']
    assert out['grad_comm_f32']['value'] and out['sharded_optimizer']['value']
