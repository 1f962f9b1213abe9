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


# ---- the training loop's steady-state step time THROUGH THE FACADE, by keyword set (two runs of different length: the
# difference divides out upload, preprocessing, model build and the final embedding) ----
def steady(label, **kw):
    t = {}
    for ep in (2, 6):
        jm = JAMIE(output_dim=32, pca_dim=None, use_f_tilde=False, epoch_DNN=ep, min_epochs=2, log_DNN=10 ** 9, batch_size=512, **kw)
        with contextlib.redirect_stdout(io.StringIO()):
            t0 = time.perf_counter()
            jm.fit_transform(dataset=[d.copy() for d in data])
            torch.cuda.synchronize()
            t[ep] = time.perf_counter() - t0
    ms = 1e3 * (t[6] - t[2]) / (4 * (N // 512))
    print(f'{label:62s} {ms:7.3f} ms/step  {512 / ms * 1e3:10.0f} cells/s   (fit_transform 6 epochs: {t[6]:.2f} s)', flush=True)


print('steady-state step through JAMIE.fit_transform (100k x (2000, 1000), latent 32, B = 512):')
steady('default keywords (fp32, sampler auto = device plan)')
steady("sampler='numpy' (fp32; the reference's np.random.choice stream)", sampler='numpy')
steady("compute_dtype='bf16' (sampler auto)", compute_dtype='bf16')
steady("compute_dtype='bf16', preprocess='device' (recommended)", compute_dtype='bf16', preprocess='device')
