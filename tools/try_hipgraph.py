"""GPU box: the recorded bf16 step at config 2 replayed (a) by the launch plan (one foreign call per launch) and (b) as a captured
HIP graph (torch.cuda.CUDAGraph around one plan replay): does graph replay shorten the gaps between dependent kernels?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar

nv.require_gpu()
dev = torch.device('cuda:0')
dims, L, B, N = (2000, 1000), 32, 512, 100000
torch.manual_seed(666)
model = edModelVar(list(dims), L, device=dev)
eng = TrainEngine(model, B, lr=1e-3, seed=666, compute_dtype='bf16')
data = [torch.randn(N, d, device=dev) for d in dims]
idx = torch.zeros(B, dtype=torch.int32, device=dev)
eng.set_kl_anneal(0.5)
plan = eng.make_plan(data, idx, N, False, None)


def timed(fn, n=300):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        eng.run_plan(plan)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.run_plan(plan)
torch.cuda.synchronize()
for rep in range(3):
    print(f'plan replay {timed(lambda: eng.run_plan(plan)):7.1f} us/step    graph replay {timed(g.replay):7.1f} us/step', flush=True)
print('losses', eng.read_losses())
