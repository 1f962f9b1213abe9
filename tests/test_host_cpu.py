"""CPU-side tests: C-ABI library exports, host logic (layout, schedules, preprocessing, sharding), the
world_size-2 gradient exchange over gloo, and the no-fallback guarantees.  No GPU needed."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import jamie_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def lib():
    import jamie_amd
    jamie_amd.build_library()
    from jamie_amd import _native
    return _native


def test_library_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, 'include', 'jamie_hip.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    # the experiments block (-DJAMIE_EXPERIMENTS: kernels measured slower, not in the product library) is declared apart
    m = re.search(r'#ifdef JAMIE_EXPERIMENTS(.*?)#endif', hdr, flags=re.S)
    exp_decl = set(re.findall(r'\b(jamie_[a-z0-9_]+)\s*\(', m.group(1)))
    declared = set(re.findall(r'\b(jamie_[a-z0-9_]+)\s*\(', hdr[:m.start()] + hdr[m.end():]))
    assert len(declared) >= 15
    assert declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    assert exp_decl == set(lib.EXPERIMENT_EXPORTS) and not (exp_decl & declared)
    handle = lib.load()
    for name in declared:
        assert hasattr(handle, name), name
    if not os.environ.get('JAMIE_LIB'):
        # the default library exports ONLY what the header's product part declares (rejected experiments out)
        for name in exp_decl:
            assert not hasattr(handle, name), name
        out = subprocess.run(['nm', '-D', '--defined-only', lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = set(re.findall(r' T (jamie_[a-z0-9_]+)$', out, flags=re.M))
        assert exported == declared, exported ^ declared
    import jamie_amd
    exp_lib = os.path.join(os.path.dirname(jamie_amd.library_path()), 'libjamie_hip_exp.so')
    if os.path.exists(exp_lib):
        import ctypes
        h2 = ctypes.CDLL(exp_lib)
        for name in declared | exp_decl:
            assert hasattr(h2, name), name
    assert handle.jamie_version() >= 100
    assert handle.jamie_max_partials() >= 1024


def test_bench_roofline_kernel_names_exist_in_the_library(lib):
    """The kernel `bench.py` names in its fp32 `roofline` record (and `tools/collect_profiles.py` looks up in the rocprofv3 tables)
    is an instantiation the library really contains, and it is the one the engine's default tile configuration launches."""
    sys.path.insert(0, ROOT)
    import bench
    from jamie_amd import engine
    out = subprocess.run(['nm', '-C', lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert f'void {bench.F32_ROOFLINE_KERNEL}(GemmGroup)' in out, bench.F32_ROOFLINE_KERNEL
    assert 'clip_adam_kernel<' in out
    # <BM, BN, BK, WM, WN, A_KC, B_KC, FAST, TAG, MID, X3>: configuration 17 = 128x128x32 on 4 x 4 waves with the barrier in mid k-step
    assert engine.F32_CFG_ROWS == 17 and bench.F32_ROOFLINE_KERNEL.endswith('<128, 128, 32, 4, 4, true, true, 2, 1, true, 0>')
    # <..., MID, X3>: configuration 20 = 128x128x32 on 2 x 2 waves, the products as bf16 MFMAs on three-piece cuts
    assert f'void {bench.F32_X3_ROOFLINE_KERNEL}(GemmGroup)' in out, bench.F32_X3_ROOFLINE_KERNEL
    assert engine.F32_CFG_X3 == 21 and engine.TUNING['f32_x3'] is True


def test_package_reads_no_environment_switches():
    """jamie_amd/*.py read NO JAMIE_* variable besides the library path and the two test hooks of the data-parallel launcher
    (29 switches used to select measured-and-rejected variants; those are engine.TUNING keys now, set by
    engine.tune() / bench.py --tune), the C sources read the environment in the experiments build only, and nothing in the
    product imports the experiments module."""
    allow = {'JAMIE_HIP_LIB', 'JAMIE_LIB', 'JAMIE_DIST_BACKEND', 'JAMIE_SHARE_GPU', 'JAMIE_HIPCC_FLAGS'}
    pkg = os.path.join(ROOT, 'jamie_amd')
    for f in sorted(os.listdir(pkg)):
        if not f.endswith('.py'):
            continue
        txt = open(os.path.join(pkg, f)).read()
        code = re.sub('(\'\'\'|\"\"\").*?\\1', '', txt, flags=re.S)
        code = re.sub(r'#.*', '', code)
        for name in re.findall(r'JAMIE_[A-Z0-9_]+', code):
            assert name in allow or name in ('JAMIE_MAX_GEMM_GROUP', 'JAMIE_MAX_GEMM_GROUP_F32', 'JAMIE_EXPERIMENTS'), (f, name)
        if 'environ' in code:
            for name in re.findall(r"environ(?:\.get)?[\[(]\s*'([A-Z0-9_]+)'", code):
                assert name in allow | {'HIPCC', 'WORLD_SIZE', 'RANK', 'LOCAL_RANK'}, (f, name)
        if f != 'experiments.py':
            assert not re.search(r'^\s*(from\s+\.\s+import\s+.*\bexperiments\b|from\s+\.experiments|import\s+.*experiments)', code, flags=re.M), f
    for f in sorted(os.listdir(os.path.join(pkg, 'csrc'))):
        if f.endswith(('.hip', '.h')):
            txt = open(os.path.join(pkg, 'csrc', f)).read()
            outside = re.sub(r'#ifdef JAMIE_EXPERIMENTS.*?#endif', '', txt, flags=re.S)
            assert 'getenv' not in outside, f
    from jamie_amd import engine
    with pytest.raises(KeyError):
        engine.tune(no_such_knob=1)
    engine.tune(prefetch=1)


def test_ctypes_struct_sizes_match_header(lib, tmp_path):
    """sizeof() of every C struct equals the ctypes mirror (compiled with gcc from the public header)."""
    src = tmp_path / 'sz.c'
    src.write_text('#include <stdio.h>\n#include "jamie_hip.h"\nint main(){printf("%zu %zu %zu %zu\\n",'
                   'sizeof(jamie_gemm_problem),sizeof(jamie_bnact_fwd_problem),sizeof(jamie_bnact_bwd_problem),'
                   'sizeof(jamie_latent));printf("%zu %zu\\n",sizeof(jamie_pd_state),sizeof(jamie_latent_m));return 0;}\n')
    exe = tmp_path / 'sz'
    subprocess.run(['gcc', '-I', os.path.join(ROOT, 'include'), str(src), '-o', str(exe)], check=True)
    sizes = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    import ctypes
    assert sizes == [ctypes.sizeof(lib.GemmProblem), ctypes.sizeof(lib.BnFwdProblem),
                     ctypes.sizeof(lib.BnBwdProblem), ctypes.sizeof(lib.Latent), ctypes.sizeof(lib.PdState), ctypes.sizeof(lib.LatentM)]


def test_no_gpu_fails_loudly(lib):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from jamie_amd.model import edModelVar
    with pytest.raises(lib.JamieHipError):
        edModelVar([8, 6], 2)
    import jamie_amd
    with pytest.raises(lib.JamieHipError):
        jamie_amd.JAMIE(device='cpu')
    with pytest.raises(lib.JamieHipError):
        lib.ptr(torch.zeros(3))


def test_product_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, 'jamie_amd')):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', txt, flags=re.M), f


@pytest.mark.parametrize('dims,L', [((200, 100), 16), ((2000, 1000), 32), ((24, 20), 4), ((70, 66), 8)])
def test_param_layout(dims, L):
    from jamie_amd.model import ParamLayout
    lay = ParamLayout(dims, L)
    assert lay.num_parameters() == orc.param_count(dims, L)
    for name, (off, shape) in lay.entries.items():
        assert off % 4 == 0, name                      # 16-byte aligned views
    offs = sorted((o, int(np.prod(s))) for o, s in lay.entries.values())
    for (o0, n0), (o1, _) in zip(offs, offs[1:]):
        assert o0 + n0 <= o1                            # no overlap
    assert offs[-1][0] + offs[-1][1] <= lay.total
    torch.manual_seed(0)
    P, Bf = orc.init_state(dims, L)
    names = lay.reference_names()
    assert set(names) == set(P)
    flat = torch.zeros(lay.total)
    views = lay.views(flat)
    for ref, (mine, sl) in names.items():
        v = views[mine] if sl is None else views[mine][sl]
        assert tuple(v.shape) == tuple(P[ref].shape), ref
    assert set(lay.reference_bn_names()) == {k for k in Bf if 'num_batches' not in k}


def test_reference_parameter_count_formula():
    assert orc.param_count((200, 100), 16) == 420166
    assert orc.param_count((2000, 1000), 32) == 40345130
    assert orc.param_count((5000, 2000), 64) == 233477258


def test_kl_anneal_and_splitk():
    from jamie_amd.engine import choose_splitk, kl_anneal
    for e, me, ed in [(0, 2500, 10000), (1250, 2500, 10000), (7, 0, 30), (3, 4, 5)]:
        assert kl_anneal(e, me, ed) == pytest.approx(float(orc.kl_anneal(e, me, ed)), rel=1e-15)
    assert choose_splitk(512, 4000, 2000) == 1
    assert choose_splitk(512, 64, 2000) > 1
    assert choose_splitk(512, 64, 100) == 1
    for M, N, K in [(512, 2000, 4000), (512, 32, 1000), (64, 70, 50)]:
        s = choose_splitk(M, N, K)
        assert s >= 1 and (s == 1 or K // s >= (256 if N > 64 else 125)) and (N > 64 or s <= 8)


def test_preclass_matches_oracle():
    from jamie_amd.utilities import preclass
    rng = np.random.default_rng(0)
    X = rng.standard_normal((50, 7))
    X[:, 3] = 2.0                                        # zero-variance feature -> NaN -> 0
    a, b = preclass(X, axis=0), orc.Preclass(X, axis=0)
    Y = rng.standard_normal((9, 7))
    np.testing.assert_array_equal(a.transform(Y.copy()), b.transform(Y.copy()))
    assert np.all(a.transform(X.copy())[:, 3] == 0)
    np.testing.assert_array_equal(a.inverse_transform(Y), b.inverse_transform(Y))
    g = preclass(X, axis=None)
    assert g.mean.shape == () and np.isclose(g.transform(X.copy()).std(), 1)


def test_bf16_gemm_plans():
    """Deterministic tile / split-K rules of the bf16 GEMM launches (engine.plan_bf16_*): config 2 gets exactly one
    workgroup per CU on the forward launches; small or ragged problems fall back to the library default."""
    from jamie_amd.engine import plan_bf16_rows, plan_bf16_bwd, BF16_TILE, N_CU
    import math
    for shapes in ([(4000, 2000), (2000, 1000)], [(2000, 4000), (1000, 2000)]):
        cfg, sk = plan_bf16_rows(512, shapes)
        bm, bn = BF16_TILE[cfg]
        assert sum(math.ceil(512 / bm) * math.ceil(N / bn) * s for (N, K), s in zip(shapes, sk)) == N_CU
        assert all(K // s >= 256 for (N, K), s in zip(shapes, sk))
    assert plan_bf16_rows(512, [(4000, 2000), (2000, 1000)])[0] != plan_bf16_rows(512, [(2000, 4000), (1000, 2000)])[0]
    assert plan_bf16_rows(64, [(80, 40), (48, 24)])[0] == -1 and plan_bf16_bwd(64, [(80, 40)])[0] == -1
    cfg, sk = plan_bf16_bwd(512, [(2000, 4000), (1000, 2000)])
    assert cfg >= 0 and sk == [2, 1]
    cfg, sk = plan_bf16_rows(512, [(10000, 5000), (4000, 2000)])          # config-5 dimensions: enough tiles already
    assert sk == [1, 1]


def test_shard_bounds_partition():
    from jamie_amd.distributed import shard_bounds
    for n, w in [(100000, 8), (10, 3), (7, 8), (1000001, 8)]:
        b = [shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(x[1] == y[0] for x, y in zip(b, b[1:]))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_time_logger_labels():
    from jamie_amd.utilities import time_logger
    t = time_logger()
    t.log('Setup'); t.log('Step'); t.log('Step')
    assert list(t.history) == ['Setup', 'Step'] and len(t.history['Step']) == 2


_WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from jamie_amd import distributed as jd
rank, world, _ = jd.init_from_env('gloo')
assert world == 2
torch.manual_seed(rank)
flat = torch.randn(1003)
mine = flat.clone()
jd.broadcast_flat(flat)                       # parameters start from rank 0
ref0 = torch.manual_seed(0) and torch.randn(1003)
assert torch.equal(flat, ref0)
for nb in (1, 3):
    g = torch.full((1003,), float(rank + 1)) + torch.arange(1003) * 0.001
    jd.GradAllReduce(n_buckets=nb)(g)
    want = 3.0 + 2 * torch.arange(1003) * 0.001
    assert torch.allclose(g, want), nb
# the overlapped exchange: regions arrive from the end of the buffer (backward order), merged into buckets;
# fp32 messages are exact, bf16 messages (bf16 compute mode) agree to bf16 rounding and identically on both ranks
for comm, tol in ((None, 0.0), (torch.bfloat16, 2e-2)):
    ov = jd.OverlappedGradAllReduce(min_bytes=1024, comm_dtype=comm)
    g = torch.full((1003,), float(rank + 1)) + torch.arange(1003) * 0.001
    for lo, hi in ((900, 1003), (600, 900), (590, 600), (100, 590), (0, 100)):
        ov.region_done(g, lo, hi)
    ov.finish()
    want = 3.0 + 2 * torch.arange(1003) * 0.001
    assert torch.allclose(g, want, rtol=tol, atol=1e-6), comm
    both = [torch.zeros(1003) for _ in range(world)]
    dist.all_gather(both, g)
    assert torch.equal(both[0], both[1]), comm
    assert not ov.works and ov.pending is None
    # what the step put on the wire, in issue order (bench.py probes the all-reduce bandwidth at exactly these sizes): with
    # min_bytes = 1024 the merged backward-adjacent regions [600, 1003), [100, 600) and the tail [0, 100) as fp32, and
    # [100, 1003) + the tail as bf16 (half the bytes per element: the bucket fills later)
    msgs = ov.message_elements()
    assert msgs == ([403, 500, 100] if comm is None else [903, 100]), msgs
    ov.enable_exposure(True)                   # (no GPU here: nothing to bracket, nothing recorded)
    ov.region_done(g, 0, 1003); ov.finish()
    assert ov.exposed_us() is None and ov.message_elements() == [1003]
    dist.all_reduce(g)                         # (keep the ranks' buffers in step for what follows)
# producers that write the bf16 message buffer themselves (the engine's dW epilogues): precast regions are reduced where they
# stand, the fp32 buffer is neither read nor written, and the dry run (one process pretending to be N ranks) issues nothing
ov = jd.OverlappedGradAllReduce(min_bytes=1024, comm_dtype=torch.bfloat16)
g = torch.full((1003,), -7.0)                                  # (must stay untouched)
msg = ov.message_buffer(g)
assert msg.dtype == torch.bfloat16 and msg.numel() == 1003 and ov.message_buffer(g) is msg
msg.copy_((torch.full((1003,), float(rank + 1)) + torch.arange(1003) * 0.001).to(torch.bfloat16))
mine16 = msg.clone()
for lo, hi in ((900, 1003), (600, 900), (590, 600), (100, 590), (0, 100)):
    ov.region_done(g, lo, hi, precast=True)
ov.finish(copy_back=False)
both = [torch.zeros(1003, dtype=torch.bfloat16) for _ in range(world)]
dist.all_gather(both, mine16)
assert torch.equal(msg.float(), (both[0].float() + both[1].float()).to(torch.bfloat16).float())
assert torch.equal(g, torch.full((1003,), -7.0)) and not ov.works and ov.pending is None
# averaging happens inside the optimiser through grad_scale = 1/world: emulate the oracle update
from oracle import jamie_oracle as orc
p = [torch.ones(8)]
opt = orc.Adam(p, 1e-2)
gs = [(g[:8] / world).clone()]
orc.clip_grad_norm(gs); opt.step(gs)
out = [torch.zeros(8) for _ in range(world)]
dist.all_gather(out, p[0])
assert torch.equal(out[0], out[1])            # identical updates on every rank
lo, hi = jd.shard_bounds(11, rank, world)
assert (lo, hi) == ((0, 6) if rank == 0 else (6, 11))
# ---- sharded optimiser exchange: the large regions are reduce-scattered into this rank's packed pieces, the small region in
# front of them is all-reduced, updated pieces are all-gathered back (fp32 messages and bf16 messages written by the producer)
for comm in (None, torch.bfloat16):
    ex = jd.ShardedGradExchange(comm_dtype=comm)
    assert ex.world == 2 and ex.rank == rank
    n = 40 + 64 + 32
    g = (torch.arange(n, dtype=torch.float32) * 0.25 + rank + 1)
    mine = g.clone()
    spans = [(40, 104), (104, 136)]
    shard = torch.zeros(48, dtype=torch.float32 if comm is None else torch.bfloat16)
    ex.set_shards([(40, 104, shard[:32]), (104, 136, shard[32:])])
    if comm is not None:
        ex.message_buffer(g).copy_(g.to(torch.bfloat16))
    for lo_, hi_ in ((104, 136), (0, 40), (40, 104)):          # backward order; the small region is announced before the first one
        ex.region_done(g, lo_, hi_, precast=comm is not None)
    ex.finish(copy_back=False)
    both = [torch.zeros(n) for _ in range(world)]
    dist.all_gather(both, mine)
    total = both[0] + both[1] if comm is None else (both[0].to(torch.bfloat16).float() + both[1].to(torch.bfloat16).float()).to(torch.bfloat16).float()
    want = torch.cat([total[40 + 32 * rank:40 + 32 * (rank + 1)], total[104 + 16 * rank:104 + 16 * (rank + 1)]])
    assert torch.equal(shard.float(), want), (comm, shard, want)
    src = g if comm is None else ex.comm
    assert torch.equal(src[:40].float(), total[:40])                           # the small region: all-reduced in place
    part = torch.tensor([float(rank + 1), 10.0 * (rank + 1)])
    ex.sum_partials(part)
    assert part.tolist() == [3.0, 30.0]
    full = torch.full((64,), -1.0)
    piece = torch.arange(32, dtype=torch.float32) + 100 * rank
    ex.gather('enc0', full, piece)
    ex.wait_gather('enc0')
    assert torch.equal(full, torch.cat([torch.arange(32.0), torch.arange(32.0) + 100])) and not ex.gathers
    try:
        ex.set_shards([(0, 41, torch.zeros(20))])
        raise SystemExit('an odd region must be refused')
    except ValueError:
        pass
dist.destroy_process_group()
print('OK', rank)
'''


def test_bench_counts_gpus_from_sysfs_without_hip(tmp_path, monkeypatch):
    """bench.visible_gpus (the launcher parent of `bench.py --gpus N` must not initialise HIP before it starts the ranks,
    ADVICE r4): KFD topology nodes with SIMDs, cut down by plain index lists in *_VISIBLE_DEVICES; None when sysfs is silent."""
    sys.path.insert(0, ROOT)
    import bench
    for k in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        monkeypatch.delenv(k, raising=False)
    assert bench.visible_gpus(str(tmp_path / 'absent')) is None
    for i, simd in enumerate((0, 1024, 1024, 1024)):          # node 0: the CPU
        d = tmp_path / 'nodes' / str(i)
        d.mkdir(parents=True)
        (d / 'properties').write_text(f'cpu_cores_count {64 if simd == 0 else 0}\nsimd_count {simd}\nmem_banks_count 1\n')
    assert bench.visible_gpus(str(tmp_path / 'nodes')) == 3
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '0,2')
    assert bench.visible_gpus(str(tmp_path / 'nodes')) == 2
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', 'GPU-abcdef')   # UUID form: left to the ranks' own check
    assert bench.visible_gpus(str(tmp_path / 'nodes')) == 3
    src = open(os.path.join(ROOT, 'bench.py')).read()
    body = src[src.index('def spawn_ranks'):src.index('def main')]
    assert 'device_count' not in body and 'is_available' not in body      # the parent asks sysfs, never the runtime


def test_dry_run_exchange_issues_no_collective():
    """OverlappedGradAllReduce(dry_run_world=N) in one process without a process group: the region bookkeeping of an N-rank
    step, no collective (bench.py --dry-run-world)."""
    from jamie_amd import distributed as jd
    ov = jd.OverlappedGradAllReduce(min_bytes=64, comm_dtype=torch.bfloat16, dry_run_world=8)
    assert ov.world == 8 and ov.dry
    g = torch.arange(100, dtype=torch.float32)
    ov.region_done(g, 50, 100)
    ov.region_done(g, 0, 50)
    ov.finish(copy_back=False)
    assert torch.equal(ov.comm.float(), g.to(torch.bfloat16).float()) and not ov.works       # cast into the message buffer, nothing reduced
    ov.message_buffer(g).zero_()
    ov.region_done(g, 0, 100, precast=True)
    ov.finish(copy_back=False)
    assert float(ov.comm.abs().sum()) == 0.0 and torch.equal(g, torch.arange(100, dtype=torch.float32))


def test_dry_run_sharded_exchange_moves_this_ranks_pieces():
    """ShardedGradExchange(dry_run_world=N, dry_run_rank=r) in one process: no collective, but the data movement a rank sees
    besides the wire -- its piece of a reduce-scattered region lands in the packed shard, its updated piece in the full buffer."""
    from jamie_amd import distributed as jd
    ex = jd.ShardedGradExchange(dry_run_world=4, dry_run_rank=2)
    assert ex.world == 4 and ex.rank == 2 and ex.dry
    g = torch.arange(96, dtype=torch.float32)
    shard = torch.zeros(16)
    ex.set_shards([(32, 96, shard)])
    ex.region_done(g, 32, 96)
    ex.region_done(g, 0, 32)
    ex.finish(copy_back=False)
    assert torch.equal(shard, g[32 + 32:32 + 48]) and not ex.works and ex.pending is None
    full = torch.zeros(64)
    ex.gather('enc0', full, shard + 1)
    ex.wait_gather('enc0')
    assert torch.equal(full[32:48], shard + 1) and float(full[:32].abs().sum() + full[48:].abs().sum()) == 0.0


def test_gradient_exchange_world2_gloo(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29533')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count('OK') == 2


# ---- correspondence stage: host pieces ----
@pytest.mark.parametrize('mode', ['euclidean', 'cosine', 'spearman', 'pearson', 'cityblock'])
def test_distance_matrix_from_first_principles(mode):
    """Stage A's distance modes (reference jamie.py:839-890) against the DEFINITIONS written out with plain numpy loops /
    rank transforms -- not against the oracle's copy of the same scipy / sklearn calls (that would be a self-comparison)."""
    from jamie_amd.utilities import distance_matrix
    X = np.random.default_rng(3).standard_normal((30, 12))
    n = X.shape[0]
    want = np.zeros((n, n))

    def pearson(a, b):
        a, b = a - a.mean(), b - b.mean()
        return float(a @ b / np.sqrt((a @ a) * (b @ b)))

    def ranks(v):                           # no ties in continuous data
        r = np.empty(len(v))
        r[np.argsort(v)] = np.arange(1, len(v) + 1)
        return r
    for a in range(n):
        for b in range(n):
            x, y = X[a], X[b]
            if mode == 'euclidean':
                want[a, b] = np.sqrt(((x - y) ** 2).sum())
            elif mode == 'cityblock':
                want[a, b] = np.abs(x - y).sum()
            elif mode == 'cosine':
                want[a, b] = 1 - x @ y / np.sqrt((x @ x) * (y @ y))
            elif mode == 'pearson':
                want[a, b] = (1 - pearson(x, y)) / 2               # jamie.py:870-879
            else:
                want[a, b] = (1 - pearson(ranks(x), ranks(y))) / 2   # jamie.py:857-869
    got = distance_matrix(X, mode)
    assert got.shape == (n, n)
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-7)    # (sklearn's euclidean uses the Gram form: 1e-8 on the diagonal)


def test_geodesic_distances_properties():
    """unioncom's geodesic distances are restated (parity unpinned): check what the algorithm guarantees."""
    from jamie_amd.utilities import geodesic_distances
    rng = np.random.default_rng(0)
    X = np.concatenate([rng.standard_normal((30, 3)), rng.standard_normal((30, 3)) + 50.0])   # two far clusters
    D = geodesic_distances(X, kmax=7)
    assert D.shape == (60, 60) and np.isfinite(D).all()
    np.testing.assert_allclose(D, D.T)
    assert (np.diag(D) == 0).all() and (D[~np.eye(60, dtype=bool)] > 0).all()
    from sklearn.metrics import pairwise_distances
    E = pairwise_distances(X)
    assert (D[:30, :30] >= E[:30, :30] - 1e-9).all()  # a path is never shorter than the straight line
    within = max(D[:30, :30].max(), D[30:, 30:].max())
    np.testing.assert_allclose(D[:30, 30:], 2 * within)   # disconnected clusters: twice the largest finite distance


def test_pd_workspace_sizes(lib):
    assert lib.pd_workspace(100, 700) == (100 * 3, 4 * 700)


def test_bf16_message_sum_keeps_the_clip_norm():
    """Decision rule for the data-parallel gradient exchange (jamie_amd/distributed.py): bf16 messages are the default of
    the bf16 compute mode only if an 8-rank SUM carried out in bf16 (every hop of a ring reduce-scatter rounds the running
    sum to 8 significant bits) moves the clip norm ||sum g|| by less than 1e-3 relative; otherwise fp32 messages.
    Emulated on the CPU with gradients shaped like the step's (common signal + per-rank noise, five decades of scale)."""
    import torch
    g = torch.Generator().manual_seed(0)
    n, world = 1 << 20, 8
    scale = 10.0 ** (-5 * torch.rand(n, generator=g))
    common = torch.randn(n, generator=g) * scale
    grads = [common + 0.7 * torch.randn(n, generator=g) * scale for _ in range(world)]
    exact = sum(x.double() for x in grads)
    acc = grads[0].to(torch.bfloat16)
    for x in grads[1:]:
        acc = (acc.float() + x.to(torch.bfloat16).float()).to(torch.bfloat16)       # one ring hop
    rel_norm = abs(float(acc.double().norm()) - float(exact.norm())) / float(exact.norm())
    rel_l2 = float((acc.double() - exact).norm() / exact.norm())
    assert rel_norm < 1e-3, rel_norm            # measured 2e-5: round-to-nearest-even is unbiased, the norm averages it out
    assert rel_l2 < 1e-2, rel_l2                # per-element noise ~4e-3, an order below the bf16 GEMMs' own (6e-2)


def test_f32_split_k_plans_come_from_the_launch_model():
    """engine.plan_f32_rows picks the K slices of an fp32 forward / dX launch by simulating the grid on 256 CUs with two
    workgroups each (engine.launch_makespan): the plans measured best on the GPU (profiles/r03_c5_f32_plan_sweep*.log,
    r03_c2_f32_plan_check.log) must come out of it."""
    from jamie_amd.engine import F32_CFG_ROWS, F32_CFG_X3, launch_makespan, plan_f32_rows, tune
    B = 512
    # the default: configuration 20 (one workgroup per CU, ~1 us per k-step) -- its own model parameters, plans measured on the GPU
    # (profiles/r05_ab_f32_bf16x3_plans.log)
    assert plan_f32_rows(B, [(4000, 2000), (2000, 1000)]) == (F32_CFG_X3, [3, 2])          # 256 tiles of 256 x 128: one round
    assert plan_f32_rows(B, [(2000, 4000), (1000, 2000)]) == (20, [3, 2])                  # 256 tiles of 128 x 128: half the slabs of (6, 3)
    assert plan_f32_rows(B, [(10000, 5000), (4000, 2000)]) == (F32_CFG_X3, [4, 2])         # config 5's dimensions: measured best
    assert plan_f32_rows(B, [(5000, 10000), (2000, 4000)]) == (F32_CFG_X3, [5, 2])
    tune(f32_x3=False)
    try:
        _plans_of_the_fp32_pipe(B, F32_CFG_ROWS, launch_makespan, plan_f32_rows)
    finally:
        tune(f32_x3=True)


def _plans_of_the_fp32_pipe(B, F32_CFG_ROWS, launch_makespan, plan_f32_rows):
    assert launch_makespan([10.0] * 4, n_cu=2) == pytest.approx(20.0)             # two CUs, two workgroups each
    assert launch_makespan([10.0] * 3, n_cu=2) == pytest.approx(20.0)             # the lone one finishes earlier (10 / 0.87)
    assert launch_makespan([10.0] * 5, n_cu=2) == pytest.approx(20.0 + 10.0 / 0.87)
    assert launch_makespan([8.0, 2.0], n_cu=1) == pytest.approx(4.0 + 6.0 / 0.87)  # shared until the short one leaves
    assert plan_f32_rows(B, [(4000, 2000), (2000, 1000)]) == (F32_CFG_ROWS, [3, 2])          # config 2, d -> 2d
    cfg, sk = plan_f32_rows(B, [(2000, 4000), (1000, 2000)])                                 # config 2, 2d -> d: (3, 2) = (6, 3) measured
    assert cfg == F32_CFG_ROWS and sk in ([3, 2], [6, 3], [6, 4])
    assert plan_f32_rows(B, [(10000, 5000), (4000, 2000)]) == (F32_CFG_ROWS, [2, 1])         # config 5, d -> 2d
    assert plan_f32_rows(B, [(5000, 10000), (2000, 4000)]) == (F32_CFG_ROWS, [4, 2])         # config 5, 2d -> d
    assert plan_f32_rows(B, [(400, 200), (200, 100)]) == (-1, None)                          # small layers: the library default


def test_dp_model_prices_a_timeline_by_hand():
    """bench.dp_model (the exposure model printed by bench.py --dry-run-world): a replicated and a sharded timeline small enough to
    price by hand -- a collective starts 35 us after it is issued or 5 us after the one queued before it ends, takes
    bytes / algorithm bandwidth, and the main stream stalls where it waits."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(ROOT, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    n, bus = 8, bench.ASSUMED_BUS_GBS
    lat, gap = bench.ASSUMED_COLLECTIVE_LATENCY_US * 1e-3, bench.ASSUMED_COLLECTIVE_GAP_US * 1e-3
    bw_ar, bw_half = bus * n / (2.0 * (n - 1)), bus * n / (n - 1.0)             # GB/s = MB/ms
    # replicated: one 30 MB all-reduce announced at 0.30 ms, the step waits at 0.40 ms, a dry step takes 0.60 ms
    ev = [('message', 30e6, 0.30), ('finish', 0, 0.40), ('step_end', 0, 0.60)]
    m = bench.dp_model(ev, 0.60, 0.55, n, 2.0, 'fp32_messages')
    stall = 0.30 + lat + 30.0 / bw_ar - 0.40
    assert m['optimizer'] == 'replicated'
    assert m['as_run']['predicted_step_us'] == pytest.approx(1e3 * (0.60 + stall), rel=1e-9)
    assert m['as_run']['predicted_speedup'] == pytest.approx(n * 0.55 / (0.60 + stall), rel=1e-9)
    stall2 = 0.30 + lat + 60.0 / bw_ar - 0.40
    assert m['fp32_messages']['predicted_step_us'] == pytest.approx(1e3 * (0.60 + stall2), rel=1e-9)
    # the traced step may run longer than the timed one: positions scale with it
    m2 = bench.dp_model([(k, b, 2 * t) for k, b, t in ev], 0.60, 0.55, n, 2.0, 'fp32_messages')
    assert m2['as_run']['predicted_step_us'] == pytest.approx(m['as_run']['predicted_step_us'], rel=1e-9)
    # sharded: the reduce-scatter hides under the backward pass; the norm's all-reduce and the first all-gather do not
    ev = [('wait:enc0', 0, 0.02), ('reduce_scatter', 10e6, 0.20), ('finish', 0, 0.40), ('all_gather', 10e6, 0.45), ('step_end', 0, 0.50)]
    m = bench.dp_model(ev, 0.50, 0.55, n, 2.0, 'fp32_messages')
    assert m['optimizer'] == 'sharded' and m['as_run']['stalls']['gradient_wait_us'] == 0.0
    assert m['as_run']['stalls']['norm_all_reduce_us'] == pytest.approx(1e3 * lat)
    # steady state: step length T solves  T = 0.50 + lat + wait,  all-gather issued at 0.45 + lat into the step, needed at T + 0.02
    ag_end = 0.45 + lat + lat + 10.0 / bw_half          # (relative to the step's start; the wire is idle when it is issued)
    T = 0.50 + lat
    wait = max(0.0, ag_end - (T + 0.02))
    assert m['as_run']['stalls']['forward_waits_us'] == pytest.approx(1e3 * wait, abs=1e-6)
    assert m['as_run']['predicted_step_us'] == pytest.approx(1e3 * (T + wait), abs=1e-6)
    assert gap > 0


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus N` without a launcher starts its own ranks (tests/test_hip_step.py::test_bench_starts_its_own_ranks
    on the GPU box); with fewer than N devices visible -- here: none -- it exits non-zero and prints NO JSON line, rather than a
    line for fewer GPUs than were asked for (the driver invokes it exactly this way)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK')}
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2'], capture_output=True,
                       text=True, env=env, timeout=300)
    assert r.returncode != 0 and 'GPU(s) visible' in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    # a launcher's WORLD_SIZE that contradicts --gpus is refused as well (it used to be accepted when WORLD_SIZE was 1)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2'], capture_output=True,
                       text=True, env=dict(env, WORLD_SIZE='1', RANK='0'), timeout=300)
    assert r.returncode != 0 and not [ln for ln in r.stdout.splitlines() if ln.startswith('{')]

def test_three_piece_cut_of_an_fp32_value_is_exact():
    """The arithmetic fact gemm_f32.hip's configurations 20 / 21 rest on (restated in numpy): cutting an fp32 value by
    truncation -- hi = x with its low 16 bits cleared, mid = (x - hi) likewise, lo = (x - hi) - mid -- gives three bf16
    values (low 16 bits zero) whose sum is x EXACTLY; both subtractions are exact in fp32.  Every bf16 x bf16 product is exact
    in fp32 (8 x 8 significand bits), and the three products the kernels drop (mid lo, lo mid, lo lo) are at most 2^-21 and typically 2^-24 of |a||b| (an fp32 product's own rounding is 2^-24)."""
    rng = np.random.default_rng(0)
    x = (rng.standard_normal(200000) * np.exp(8 * rng.standard_normal(200000))).astype(np.float32)
    x = np.concatenate([x, np.float32([0.0, -0.0, 1.0, -1.0, 3.0e38, 1.2e-30, 1 + 2 ** -23, 255.99998])])

    def trunc(v):
        return (v.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
    hi = trunc(x)
    r1 = (x - hi).astype(np.float32)
    assert np.array_equal(r1.astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))        # exact subtraction
    mid = trunc(r1)
    r2 = (r1 - mid).astype(np.float32)
    assert np.array_equal(r2.astype(np.float64), r1.astype(np.float64) - mid.astype(np.float64))
    lo = trunc(r2)
    assert np.array_equal(lo, r2)                                                                          # nothing left behind
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    for piece in (hi, mid, lo):
        assert not (piece.view(np.uint32) & np.uint32(0xFFFF)).any()                                       # representable in bf16
    nz = x != 0
    assert (np.abs(mid[nz]) <= np.abs(x[nz]) * 2.0 ** -7).all() and (np.abs(lo[nz]) <= np.abs(x[nz]) * 2.0 ** -15).all()
    # the dropped terms of a product, relative to |a||b|
    a, b = x[:100000], x[100000:200000]
    am, al, bm, bl = mid[:100000].astype(np.float64), lo[:100000].astype(np.float64), mid[100000:200000].astype(np.float64), lo[100000:200000].astype(np.float64)
    dropped = np.abs(am * bl + al * bm + al * bl)
    scale = np.abs(a.astype(np.float64) * b.astype(np.float64))
    ok = scale > 0
    assert (dropped[ok] <= scale[ok] * 2.0 ** -21).all() and np.median(dropped[ok] / scale[ok]) < 2.0 ** -24
