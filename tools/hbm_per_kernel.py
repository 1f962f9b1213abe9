"""HBM bytes per kernel launch and per bf16 step from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_bench.sh
(gpurun_out/<tag>/pmc_fetch, pmc_write): FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes; launches per step as the engine
issues them at config 2 (bf16).  Usage: python tools/hbm_per_kernel.py <tag>"""
import collections, csv, glob, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]


def load(d, counter):
    f = max(glob.glob(os.path.join(root, 'gpurun_out', tag, d, '*', '*_counter_collection.csv')), key=os.path.getmtime)
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] == counter:
            tot[r['Kernel_Name']] += float(r['Counter_Value'])
            cnt[r['Kernel_Name']] += 1
    return tot, cnt


ft, fc = load('pmc_fetch', 'FETCH_SIZE')
wt, wc = load('pmc_write', 'WRITE_SIZE')
per_step = [('clip_adam', 1, 'clip + Adam (+ bf16 weight copy, + next batch gather)'),
            ('gemm_bf16_dma2_kernel<128, 128, 2, 4, 1, 2, 1>', 2, 'grouped dW + dX (dec2, dec1)'),
            ('gemm_bf16_dma2_kernel<128, 128, 2, 4, 0, 2, 1>', 2, 'enc1 dW + dX + skinny dW riders; d comb'),
            ('gemm_bf16_dma2_ride_kernel', 1, 'enc0 dW + range norm'),
            ('gemm_bf16_dma2_kernel<256, 128, 4, 4, 1, 3, 0>', 2, 'forward d -> 2d'),
            ('gemm_bf16_dma2_kernel<128, 128, 2, 4, 1, 3, 0>', 2, 'forward 2d -> d'),
            ('bn_act_bwd4', 4, 'BatchNorm backward'), ('bn_act_fwd4_kernel<4, 8>', 2, 'BatchNorm forward, 2d-wide layers (32-column strips)'),
            ('bn_act_fwd4_kernel<4, 4>', 2, 'BatchNorm forward, d-wide layers'), ('mse_cast_kernel', 1, 'MSE + d x_hat'),
            ('latent_m_bwd', 1, 'latent backward'), ('latent_m_fwd', 1, 'latent forward'),
            ('gemm_bf16_dma_kernel<64, 64, 2, 2, 0, 3, 0>', 1, 'heads forward')]
tot = 0.0
for key, n, what in per_step:
    ks = [k for k in fc if key in k]
    assert len(ks) == 1, (key, ks)
    k = ks[0]
    avg = (2 * ft[k] / fc[k] + wt[k] / wc[k]) * 1024
    tot += avg * n
    print(f'{avg / 1e6:8.1f} MB/launch x {n} = {avg * n / 1e6:8.1f} MB/step   {what}  [{key}]')
P = 40345130
print(f'total {tot / 1e9:.3f} GB/step; bf16 mode\'s own algorithmic bytes 34 P = {34 * P / 1e9:.3f} GB: ratio {tot / (34 * P):.3f}; '
      f'the fp32 byte model 44 P = {44 * P / 1e9:.3f} GB: ratio {tot / (44 * P):.3f}')
