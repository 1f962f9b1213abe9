"""GPU benchmark of the eval-mode path: streaming `transform` (embedding) and `impute` over
all cells, no N x N `corr`, BatchNorm folded into the GEMM epilogue.  Prints cells/s for both."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jamie_amd.model import edModelVar
N, dims, L = 100000, (2000, 1000), 32
if os.environ.get('EVAL_CFG'):           # fp32 GEMM tile configuration of the eval path (diagnostics: 12 against 17)
    edModelVar.EVAL_GEMM_CFG = int(os.environ['EVAL_CFG'])
torch.manual_seed(0)
model = edModelVar(dims, L).eval()
g = torch.Generator(device='cuda').manual_seed(1)
X = [torch.randn(N, d, generator=g, device='cuda') for d in dims]
for name, fn in (('transform (embed both modalities)', lambda: [model.embed(X[i], i, chunk=16384) for i in range(2)]),
                 ('impute modality 0 -> 1', lambda: model.impute(X[0], [0, 1], chunk=16384)),
                 ('impute modality 1 -> 0', lambda: model.impute(X[1], [1, 0], chunk=16384))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f'{name}: {dt*1e3:8.1f} ms for {N} cells -> {N/dt/1e6:6.2f} M cells/s')
