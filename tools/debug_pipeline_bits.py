"""GPU: bf16 pipelined optimiser against the one-launch form on the small test model -- where do the parameters differ?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
nv.require_gpu()
dims, L, B, N = (264, 136), 8, 128, 2048
g = torch.Generator().manual_seed(0)
data = [torch.randn(N, d, generator=g).cuda() for d in dims]
def run(pipe, steps):
    torch.manual_seed(9)
    model = edModelVar(dims, L)
    eng = TrainEngine(model, B, compute_dtype='bf16', seed=21)
    if pipe:
        eng.enable_pipeline()
    idx = torch.zeros(B, dtype=torch.int32, device='cuda')
    plan = eng.make_plan(data, idx, N)
    for _ in range(steps - 1):
        eng.run_plan(plan)
    eng.flush()
    torch.cuda.synchronize()
    return model, model.flat.clone(), eng.exp_avg.clone(), eng.grad.clone() if eng.grad is not None else None, eng.read_losses()
for steps in (5, 6, 8, 5, 6, 8):
    m, a, ma, ga, la = run(False, steps)
    _, b, mb, gb, lb = run(True, steps)
    d = (a - b).abs()
    print(f'steps {steps}: params differ in {int((d > 0).sum())} of {a.numel()} entries, max {d.max().item():.3e}; exp_avg differ {int(((ma - mb).abs() > 0).sum())}; losses {la[1]:.9g} vs {lb[1]:.9g}')
    if (d > 0).any():
        views = m.layout.views(d)
        print('   ', ', '.join(f'{k}: {int((v > 0).sum())}' for k, v in views.items() if (v > 0).any())[:600])
