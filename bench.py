#!/usr/bin/env python
"""Headline benchmark: training cells/sec of JAMIE's two-modality coupled VAE on MI355X
(BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c5dims] [--no-cpu-baseline]

N > 1 is launched by the driver as
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
One rank per GPU; cells are sharded by rows; every rank trains on B = 512 cells per step (weak scaling)
and the flat gradient is all-reduced once per step over RCCL.  A "step" is one pass of the hot path over
one batch: device sampler -> row gather -> forward -> losses -> backward -> (all-reduce) -> clip + Adam.
Inputs are synthetic (SURVEY.md §8(d) generator) and resident in HBM before the timed region.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (cells, dims, latent)
    'c1': (5000, (200, 100), 16),
    'c2': (100000, (2000, 1000), 32),
    'c4': (100000, (2000, 1000, 500), 64),       # 3 modalities (build-defined generalisation; no reference oracle)
    'c5dims': (100000, (5000, 2000), 64),
}
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def synth_shard(n_rows, dims, seed, device):
    """X_i = Z A_i + 0.1 E_i with a 16-dim latent, fp32, standardised per feature
    (what `preclass(axis=0)` does on the host in the reference, jamie.py:462-465)."""
    g = torch.Generator(device=device).manual_seed(1234)             # A_i shared by all ranks
    A = [torch.randn(16, d, generator=g, device=device) for d in dims]
    g2 = torch.Generator(device=device).manual_seed(seed)
    Z = torch.randn(n_rows, 16, generator=g2, device=device)
    out = []
    for a in A:
        x = Z @ a + 0.1 * torch.randn(n_rows, a.shape[1], generator=g2, device=device)
        x = (x - x.mean(0)) / x.std(0, unbiased=False)
        out.append(x.contiguous())
    return out


def flops_per_cell(dims, L):
    """F_cell = 6 P_mm - 4 sum d_i^2, P_mm = sum(8 d^2 + 3 d L)."""
    pmm = sum(8 * d * d + 3 * d * L for d in dims)
    return 6 * pmm - 4 * sum(d * d for d in dims)


def cpu_baseline(dims, L, B, budget_s=20.0):
    """The CPU oracle (plain-PyTorch restatement of the reference step, pinned to the reference by the
    golden fixtures) timed on this box's host cores: same dims, fp32, B = 512, default dropout."""
    from oracle import jamie_oracle as orc
    threads = int(os.environ.get("JAMIE_CPU_THREADS", "0")) or min(torch.get_num_threads(), 64)
    torch.set_num_threads(threads)
    torch.manual_seed(666)
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    opt = orc.Adam(P.values(), 1e-3)
    p = orc.default_dropout(dims)
    rng = np.random.default_rng(0)
    n = 8192
    Z = rng.standard_normal((n, 16)).astype(np.float32)
    data = [torch.from_numpy(Z @ rng.standard_normal((16, d)).astype(np.float32)
                             + 0.1 * rng.standard_normal((n, d)).astype(np.float32)) for d in dims]
    data = [(x - x.mean(0)) / x.std(0) for x in data]
    eye, zero = torch.eye(B), torch.zeros(B, B)

    def one():
        idx = np.random.choice(range(n), B, replace=False)
        X = [d[idx] for d in data]
        noise = orc.draw_noise(dims, L, B, p)
        orc.train_step(P, Bf, opt, X, eye, zero, noise, p, 0.5)
    one()                                     # warm-up (first step dropped)
    # the oracle scales poorly past a few dozen threads on many-core hosts (memory-bound elementwise ops and
    # dropout masks): probe a few thread counts briefly and time the sample with the fastest, stated in `cores`
    if not os.environ.get('JAMIE_CPU_THREADS'):
        best = (0.0, threads)
        for t in sorted({8, 16, 24, 32, 48, threads}):
            if t > (os.cpu_count() or t):
                continue
            torch.set_num_threads(t)
            one()
            t0 = time.perf_counter()
            one(); one()
            rate = 2 * B / (time.perf_counter() - t0)
            if rate > best[0]:
                best = (rate, t)
        threads = best[1]
        torch.set_num_threads(threads)
    t0 = time.perf_counter()
    steps = 0
    while steps < 60 and (time.perf_counter() - t0 < budget_s or steps < 3):
        one()
        steps += 1
    dt = time.perf_counter() - t0
    return {'value': B * steps / dt, 'unit': 'cells/s', 'cores': threads, 'kind': 'port',
            'sample': f'{steps} steps of B={B} at dims={tuple(dims)}, L={L}, N capped at {n}, fp32, '
                      f'{dt:.1f} s on {threads} torch threads (fastest of the probed counts; {os.cpu_count()} logical CPUs)'}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'],
                    help='GEMM operand type: bf16 (BASELINE config 2; fp32 accumulate/master) or f32 (parity config)')
    ap.add_argument('--grad-comm', default='auto', choices=['auto', 'f32', 'bf16'],
                    help='dtype of the gradient all-reduce messages (auto: the compute dtype)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pipeline', action='store_true',
                    help='clip + Adam on a second stream under the next forward pass (+2-3 %; default: main stream, so '
                         'that the roofline kernel is timed alone)')
    ap.add_argument('--transposed-weight-copies', action='store_true',
                    help='bf16 A/B: dX products on transposed bf16 weight copies (the earlier scheme) instead of reading W as stored')
    ap.add_argument('--skinny-tr', action='store_true', help='bf16 experiment: head / latent backward launches through the 128x128 k-row-major kernel')
    ap.add_argument('--prefetch', action='store_true', help='A/B: sample + gather the next batch on a side stream under clip + Adam (measured -1 %)')
    ap.add_argument('--side-transposes', action='store_true',
                    help='bf16: transposed weight copies on a side stream under the next forward pass')
    ap.add_argument('--opt-priority', type=int, default=0, help='HIP stream priority of the optimiser stream')
    ap.add_argument('--cpu-budget', type=float, default=15.0)
    args = ap.parse_args()

    from jamie_amd import distributed as jd
    rank, world, local = jd.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from jamie_amd import _native as nv
    nv.require_gpu()
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar

    n_cells, dims, L = CONFIGS[args.config]
    B = args.batch
    if args.dtype == 'bf16' and any(v % 8 for v in [L, B]):
        args.dtype = 'f32'          # bf16 operands need a latent size and a batch that are multiples of 8
    # feature counts that are not multiples of 8 (config 1: 100, config 4: 500): the bf16 engine pads them (model.py)
    pad = 8 if (args.dtype == 'bf16' and any(d % 8 for d in dims)) else 1
    lo, hi = jd.shard_bounds(n_cells, rank, world)
    data = synth_shard(hi - lo, dims, 1000 + rank, dev)
    torch.manual_seed(666)
    model = edModelVar(dims, L, device=dev, pad_features=pad)
    if world > 1:
        jd.broadcast_flat(model.flat)
    eng = TrainEngine(model, B, lr=1e-3, seed=666 + 7919 * rank, world_size=world, compute_dtype=args.dtype,
                      dx_from_weights=not args.transposed_weight_copies, skinny_tr=args.skinny_tr)
    data = eng.pad_cells(data)
    # gradient exchange in the compute precision: bf16 messages in bf16 mode (80 MB instead of 161 MB per step), fp32 otherwise
    comm = torch.bfloat16 if (args.dtype == 'bf16' and args.grad_comm == 'auto') or args.grad_comm == 'bf16' else None
    allreduce = jd.OverlappedGradAllReduce(comm_dtype=comm) if world > 1 else None
    idx = torch.zeros(B, dtype=torch.int32, device=dev)      # 'diag' sampling: same rows in both modalities
    # the reference's quirk `replace = min(features) < batch_size` (jamie.py:553) belongs to its two-modality loop; the
    # 3-modality generalisation always samples without replacement (duplicates would need a non-identity corr)
    rep = min(dims) < B and len(dims) == 2
    eng.set_kl_anneal(0.5)
    if args.pipeline:
        eng.enable_pipeline(args.opt_priority)
    if args.side_transposes:
        eng.enable_side_transposes()
    eng.enable_kernel_timing('enc_gemm', 'adam')
    # the step is a fixed launch sequence on static buffers: record it once, replay it (one foreign call per launch)
    plan = eng.make_plan(data, idx, hi - lo, rep, allreduce, prefetch=args.prefetch)

    def step():
        eng.run_plan(plan)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if os.environ.get('JAMIE_BENCH_NO_EVENTS') != '1':
        eng.enable_kernel_timing('enc_gemm', 'adam', every=8)   # drop the warm-up samples; sample every 8th step
    else:
        eng._timing = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    ls, total, _ = eng.read_losses()
    if not np.isfinite(total):
        raise SystemExit('non-finite loss in benchmark')
    cells_s = world * B * args.steps / dt

    if rank == 0:
        # roofline of the dominant kernel:
        #   f32 : the d <-> 2d Linear forward GEMM launch (both modalities grouped; 4 launches per step, each
        #         4*B*sum(d^2) FLOP) against the exact-fp32 MFMA peak;
        #   bf16: the step is HBM-bound on optimiser traffic; the dominant kernel is clip+Adam:
        #         28 bytes per parameter (read p, g, m, v; write p, m, v) against the HBM peak.
        if eng._timing is None:
            eng.enable_kernel_timing('enc_gemm', 'adam')
            for _ in range(20):
                step()
        timing_detail = {'enc_gemm': eng.kernel_timing_ms('enc_gemm', 'all'), 'adam': eng.kernel_timing_ms('adam', 'all')}
        # event pairs bracket one launch each; a host hiccup between the two records (GC, scheduler) shows up as a
        # multi-millisecond outlier in a handful of the samples, so the per-launch duration is the MEDIAN
        gemm_ms, adam_ms = timing_detail['enc_gemm']['median'], timing_detail['adam']['median']
        gemm_flop = 4.0 * B * sum(d * d for d in dims)
        traffic = None
        tf = os.path.join(ROOT, 'profiles', 'traffic.json')
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(f'{args.config}_{args.dtype}', {}).get('hbm_bytes_per_launch')
            except Exception:
                traffic = None
        if args.dtype == 'f32':
            achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
            roof = {'bound': 'mfma', 'kernel': 'gemm_f32_kernel<128, 128, 32, 4, 4, true, true, true, 1> (Linear d<->2d forward '
                                              'GEMM, both modalities in one launch; 4 launches/step)',
                    'achieved': achieved, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                    'frac': achieved / PEAK_F32_MFMA_TFLOPS, 'avg_launch_ms': gemm_ms, 'flop_per_launch': gemm_flop,
                    'traffic': traffic,
                    'whole_step_frac': cells_s / world * flops_per_cell(dims, L) / (PEAK_F32_MFMA_TFLOPS * 1e12)}
        else:
            n_par = model.layout.total
            adam_launches = len(eng.PIPE_GROUPS) if eng.pipeline else 1
            adam_bytes = 28.0 * n_par / adam_launches
            achieved = adam_bytes / (adam_ms * 1e-3) / 1e9
            step_bytes = 44.0 * n_par
            roof = {'bound': 'hbm', 'kernel': f'clip_adam_kernel (global-norm clip + Adam on the flat fp32 buffers; {adam_launches} launch(es)/step'
                                       + (', on the optimiser stream under the next forward pass)' if eng.pipeline else ')'),
                    'achieved': achieved, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': achieved / PEAK_HBM_GBS,
                    'avg_launch_ms': adam_ms, 'bytes_per_launch': adam_bytes, 'traffic': traffic,
                    'whole_step_frac': (step_bytes * cells_s / world / B) / (PEAK_HBM_GBS * 1e9),
                    'gemm_bf16_tflops': gemm_flop / (gemm_ms * 1e-3) / 1e12, 'gemm_avg_launch_ms': gemm_ms}
        out = {
            'metric': 'training cells/sec (two-modality coupled VAE)', 'value': cells_s, 'unit': 'cells/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.config}: {len(dims)}-modality synthetic {n_cells} cells x {tuple(dims)} features, '
                                   f'latent={L}, B={B}/GPU, dropout={model.dropout}, ' + ('bf16 MFMA GEMMs, fp32 accumulate/master/optimiser, ' if args.dtype == 'bf16' else 'fp32 MFMA, ') + 
                                   f'identity P (diag sampling), F=0',
                       'cells': n_cells, 'features': list(dims), 'latent': L, 'batch_per_gpu': B,
                       'parallelism': f'dp{world}', 'grad_allreduce': ('none' if world == 1 else ('bf16' if comm is not None else 'f32')),
                       'parameters': model.num_parameters(),
                       'flop_per_cell': flops_per_cell(dims, L)},
            'roofline': roof,
            'kernel_event_timing_ms': timing_detail,
            'final_loss': total,
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(dims, L, B, args.cpu_budget)
            out['gpu_over_cpu'] = cells_s / out['cpu_baseline']['value']
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
