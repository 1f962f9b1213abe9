"""GPU: where does the fp32 step on the bf16 pipe leave the fp32 pipe's result?  First-step gradient distances per region."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd import _native as nv, engine
from jamie_amd.engine import TrainEngine
from jamie_amd.model import edModelVar
nv.require_gpu()
B, dims, L = 512, (2000, 1000), 32
g = torch.Generator().manual_seed(11)
X = [torch.randn(B, d, generator=g).cuda() for d in dims]
def run(**knobs):
    base = dict(f32_x3=False, f32_rows=None, f32_rows_cfg=None, f32_dw_cfg=None, f32_dx_cfg=None, f32_dx_plan=True)
    base.update(knobs)
    engine.tune(**base)
    torch.manual_seed(666)
    model = edModelVar(dims, L)
    eng = TrainEngine(model, B, seed=3)
    eng.set_batch(X)
    eng.step()
    torch.cuda.synchronize()
    gr = {k: v.double().clone() for k, v in model.layout.views(eng.grad).items()}
    return gr, eng.read_losses()
ref, lref = run()
for name, kn in (('other slices (fp32 pipe)', dict(f32_rows='17:4,4;17:4,4')), ('rows on bf16 pipe', dict(f32_rows_cfg=20)),
                 ('rows on bf16 pipe, no dX plan', dict(f32_rows_cfg=20, f32_dx_plan=False)),
                 ('dX only on bf16 pipe', dict(f32_dx_cfg=20, f32_dx_plan=False))):
    gr, ls = run(**kn)
    tot = (sum(((gr[k] - ref[k]) ** 2).sum() for k in ref) / sum((ref[k] ** 2).sum() for k in ref)).sqrt().item()
    worst = sorted(((((gr[k] - ref[k]).norm() / (ref[k].norm() + 1e-30)).item(), k) for k in ref), reverse=True)[:6]
    print(f'{name}: whole gradient {tot:.3e}; losses {[f"{a:.9g}" for a in ls[0]]} vs {[f"{a:.9g}" for a in lref[0]]}')
    big = [k for k in ref if k.endswith('.W')]
    print('    W regions:', ', '.join(f'{k} {((gr[k] - ref[k]).norm() / ref[k].norm()).item():.1e}' for k in big), flush=True)
