"""GPU box, one rank on RCCL (backend "nccl"): host time per step of the sharded optimiser's plan replay (launch replay + the
Python callables that announce regions, enqueue the collectives and wait for them) against the GPU time of the same steps,
config 2 in bf16.  Host time is what the loop takes to RETURN (the GPU queue absorbs it); if it approaches the GPU time the
step becomes host-bound.  Usage: python tools/bench_sharded_host_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29587')
torch.cuda.set_device(0)
torch.distributed.init_process_group('nccl', rank=0, world_size=1)
from jamie_amd import distributed as jd, _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
dev = torch.device('cuda', 0)
dims, L, B, N = (2000, 1000), 32, 512, 20000
data = [torch.randn(N, d, device=dev) for d in dims]
native = None if '--torch' not in sys.argv else False          # (--torch: the collectives through torch.distributed, round 3's path)
print('collectives through', 'torch.distributed' if native is False else 'the C ABI (jamie_allreduce / _reduce_scatter / _all_gather)')
for label in ('one GPU, no exchange', 'sharded exchange (one-rank RCCL group)', 'replicated exchange (one-rank RCCL group)', 'replicated exchange (dry run: no collective calls)'):
    torch.manual_seed(3)
    model = edModelVar(dims, L, device=dev)
    eng = TrainEngine(model, B, seed=11, compute_dtype='bf16', world_size=1)
    ar = None
    if label.startswith('sharded'):
        ar = jd.ShardedGradExchange(comm_dtype=torch.bfloat16, single_rank_ok=True, native=native)
        eng.enable_sharded_optimizer(ar)
    elif label.startswith('replicated exchange (one'):
        ar = jd.OverlappedGradAllReduce(comm_dtype=torch.bfloat16, single_rank_ok=True, native=native)
    elif label.startswith('replicated'):
        ar = jd.OverlappedGradAllReduce(comm_dtype=torch.bfloat16, dry_run_world=8)
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    plan = eng.make_plan(data, idx, N, False, ar)
    for _ in range(30):
        eng.run_plan(plan)
    torch.cuda.synchronize()
    n = 8            # (short enough for the launch queue to absorb the whole loop: ~30 launches per step)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run_plan(plan)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f'{label:52s} host {1e6 * (t1 - t0) / n:7.1f} us/step   host + GPU {1e6 * (t2 - t0) / n:7.1f} us/step', flush=True)
    eng.flush(collective=True)
    del plan, eng, model
torch.distributed.destroy_process_group()
