"""GPU check: partial-correspondence training through the facade at config-2 size: sparse P with
half of the cells paired (hybrid sampler, CSR block lookup, general [B,B] correspondence blocks), bf16."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp, torch
from jamie_amd import JAMIE
N, dims, epochs = 100000, (2000, 1000), int(os.environ.get('EPOCHS', '3'))
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, 16)).astype(np.float32)
data = [(Z @ rng.standard_normal((16, d)).astype(np.float32) + 0.1 * rng.standard_normal((N, d)).astype(np.float32)) for d in dims]
k = N // 2
P = sp.csr_matrix((np.ones(k, np.float32), (np.arange(k), np.arange(k))), shape=(N, N))
VARIANTS = [v.split(':') for v in os.environ.get('VARIANTS', 'bf16:numpy,bf16:device,f32:device').split(',')]
for dtype, sampler in VARIANTS:
    jm = JAMIE(output_dim=32, pca_dim=None, use_f_tilde=False, compute_dtype=dtype, epoch_DNN=epochs, min_epochs=2,
               log_DNN=10 ** 9, batch_size=512, debug=True, sampler=sampler)
    np.random.seed(0)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        emb = jm.fit_transform(dataset=[d.copy() for d in data], P=P)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * (N // 512)
    print(f'{dtype}, sampler={sampler}: sampling {jm.sampling_method}; fit_transform {dt:.2f} s for {steps} steps; losses', {k: round(v[-1], 4) for k, v in jm.loss_history.items()}, flush=True)
    for line in buf.getvalue().splitlines():
        if any(s in line for s in ('Setup', 'Step', 'Get subset', 'Output', 'Mapping')):
            print('   ', line.strip())
