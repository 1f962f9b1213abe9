"""GPU diagnostic (round 5): in-kernel timelines of the large-tile bf16 GEMM launches INSIDE a training step of config 2 -- the
operands in the cache state the step leaves them in, not a stand-alone loop over rotating buffers (tools/stamp_gemm_bf16.py).
Diagnostic build: tools/stamp_gemm_bf16.sh;  JAMIE_HIP_LIB=$PWD/tools/libjamie_stamp.so python tools/stamp_step_bf16.py
After warm replays of the recorded plan one more step is issued launch by launch; behind every jamie_gemm_bf16* call the stream
is synchronised and the stamps of that launch are read: per workgroup entry -> first DMA issue -> all prologue DMAs issued ->
tile 0 published -> k-loop done -> stores retired."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from jamie_amd import _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
nv.require_gpu()
lib = nv.load()
dev = torch.device('cuda', 0)
n_cells, dims, L = bench.CONFIGS['c2']
n_cells = int(os.environ.get('CELLS', 20000))
B = 512
data = bench.synth_shard(n_cells, 0, n_cells, dims, 0, 1, dev)
torch.manual_seed(666)
model = edModelVar(dims, L, device=dev)
eng = TrainEngine(model, B, lr=1e-3, seed=666, compute_dtype='bf16')
idx = torch.zeros(B, dtype=torch.int32, device=dev)
eng.set_kl_anneal(0.5)
plan = eng.make_plan(data, idx, n_cells, False, None)
for _ in range(int(os.environ.get('WARM', 60))):
    eng.run_plan(plan)
torch.cuda.synchronize()

fn = lib.jamie_debug_stamps
fn.argtypes = [C.c_void_p, C.c_int]
NB = 8192
us = lambda x: x / 100.0      # noqa: E731
q = lambda v: f'min {v.min():5.2f} med {np.median(v):5.2f} p90 {np.percentile(v, 90):5.2f} max {v.max():5.2f}'   # noqa: E731
count = [0]
orig = nv.gemm_bf16


def wrapped(problems, cfg=-1, ranges=None):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(nv.current_stream())
    orig(problems, cfg, ranges)
    e1.record(nv.current_stream())
    torch.cuda.synchronize()
    count[0] += 1
    buf = (C.c_ulonglong * (8 * NB))()
    assert fn(buf, NB) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(NB, 8).astype(np.int64)
    a = a[a[:, 3] > 0]
    if not len(a):
        print(f'== launch {count[0]} cfg {cfg}: {len(problems)} problems, no large-tile stamps, event {e0.elapsed_time(e1) * 1e3:.1f} us')
        return
    a = a[a[:, 0] > a[:, 0].max() - 30000]
    t0 = a[:, 0].min()
    shapes = [(p.M, p.N, p.K, p.splitk) for p in problems]
    print(f'== launch {count[0]} cfg {cfg} (M, N, K, slices) {shapes}: {len(a)} workgroups, event {e0.elapsed_time(e1) * 1e3:.1f} us, '
          f'last end {us(a[:, 3] - t0).max():.1f} us')
    print('   entry                      ', q(us(a[:, 0] - t0)))
    if (a[:, 6] > a[:, 0]).all() and (a[:, 1] >= a[:, 7]).all():
        print('   entry -> first DMA issue   ', q(us(a[:, 6] - a[:, 0])))
        print('   issue of the prologue tiles', q(us(a[:, 7] - a[:, 6])))
        print('   issued -> tile 0 published ', q(us(a[:, 1] - a[:, 7])))
    else:
        print('   entry -> tile 0 published  ', q(us(a[:, 1] - a[:, 0])))
    print('   k-loop                     ', q(us(a[:, 2] - a[:, 1])))
    print('   stores issued + retired    ', q(us(a[:, 3] - a[:, 2])))
    for pi in sorted(set(a[:, 4] // 1000)):
        m = a[:, 4] // 1000 == pi
        nk = a[m, 4] % 1000
        print(f'   problem {pi}: {m.sum()} wgs, nk {nk.min()}..{nk.max()}, start med {np.median(us(a[m, 0] - t0)):.2f} max {us(a[m, 0] - t0).max():.2f}, '
              f'first tile {np.median(us(a[m, 1] - a[m, 0])):.2f}, loop {np.median(us(a[m, 2] - a[m, 1])):.2f} = '
              f'{np.median(us(a[m, 2] - a[m, 1])) / max(1, np.median(nk)) * 1e3:.0f} ns/k-step, stores {np.median(us(a[m, 3] - a[m, 2])):.2f}, end max {us(a[m, 3] - t0).max():.2f}')


nv.gemm_bf16 = wrapped
plan2 = eng.make_plan(data, idx, n_cells, False, None)        # the recording step is a real step, issued launch by launch
torch.cuda.synchronize()
print('losses', eng.read_losses())
