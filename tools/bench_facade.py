"""GPU check: the JAMIE facade end to end at BASELINE config 2 (100k cells x (2000, 1000), latent 32, bf16, device sampler,
device preprocessing): wall time of fit_transform, steady-state cells/s of the training loop, transform / modal_predict."""
import contextlib, io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from jamie_amd import JAMIE
N, dims, epochs = 100000, (2000, 1000), int(os.environ.get('EPOCHS', '6'))
rng = np.random.default_rng(0)
Z = rng.standard_normal((N, 16)).astype(np.float32)
data = [(Z @ rng.standard_normal((16, d)).astype(np.float32) + 0.1 * rng.standard_normal((N, d)).astype(np.float32)) for d in dims]
for pre in ('device', 'host'):
    jm = JAMIE(output_dim=32, pca_dim=None, use_f_tilde=False, compute_dtype='bf16', sampler='device', preprocess=pre,
               epoch_DNN=epochs, min_epochs=2, log_DNN=10 ** 9, batch_size=512)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()) as buf:
        emb = jm.fit_transform(dataset=[d.copy() for d in data])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    steps = epochs * (N // 512)
    print(f'preprocess={pre}: fit_transform {dt:.2f} s for {steps} steps ({epochs} epochs); final Rec {jm.loss_history["Rec"][-1]:.4f}', flush=True)
    for line in buf.getvalue().splitlines():
        if any(k in line for k in ('Setup', 'Step', 'Output', 'Mapping', 'Distance')):
            print('   ', line.strip())
t0 = time.perf_counter(); e = jm.transform(data); torch.cuda.synchronize(); t1 = time.perf_counter()
imp = jm.modal_predict(data[0], 0); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f'transform (both modalities, from host numpy): {t1 - t0:.2f} s; modal_predict: {t2 - t1:.2f} s; FOSCTTM {jm.test_closer([x[:2000] for x in e]):.4f}')
