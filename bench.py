#!/usr/bin/env python
"""Headline benchmark: training cells/sec of JAMIE's two-modality coupled VAE on MI355X
(BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c5dims] [--no-cpu-baseline]

N > 1 runs one rank per GPU.  Either a launcher starts the ranks,
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
      bench.py --gpus N --steps K --warmup W
or plain `python bench.py --gpus N` starts them itself: with WORLD_SIZE unset the parent process -- before it makes any
GPU call -- checks that N devices are visible (it exits non-zero otherwise: it never prints an N = 1 line for an N > 1
request), starts N children of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT set,
relays rank 0's JSON line and exits non-zero if any rank fails (`spawn_ranks`).
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
T_START = time.perf_counter()

CONFIGS = {
    # name: (cells, dims, latent)
    'c1': (5000, (200, 100), 16),
    'c2': (100000, (2000, 1000), 32),
    'c4': (100000, (2000, 1000, 500), 64),       # 3 modalities (build-defined generalisation; no reference oracle)
    'c5dims': (100000, (5000, 2000), 64),
    # BASELINE config 5 at its own size: 1 M cells (28 GB of fp32 over the job; one rank holds 28 / world GB), fp32 by default
    'c5': (1_000_000, (5000, 2000), 64),
}
DEFAULT_DTYPE = {'c5': 'f32'}
ASSUMED_BUS_GBS = 300.0          # all-reduce BUS bandwidth assumed by the exposure model (dp_model): stated, not measured
PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F32_ROOFLINE_KERNEL = 'gemm_f32_kernel<128, 128, 32, 4, 4, true, true, 2, 1, true, 0>'      # engine.F32_CFG_ROWS = 17 on the large layers
F32_X3_ROOFLINE_KERNEL = 'gemm_f32_kernel<256, 128, 32, 2, 2, true, true, 2, 1, false, 1>'  # engine.F32_CFG_X3 = 21 (the default: TUNING['f32_x3'])
X3_PRODUCTS = 6                  # bf16 MFMAs per fp32 product in configuration 20 (three-piece cuts: hh, hm, mh, hl, lh, mm)


def f32_roofline_terms(eng):
    """(peak in fp32-equivalent TFLOP/s, kernel name, note) of the large fp32 launches of `eng`: configuration 20 runs every fp32
    product as six bf16 MFMAs, so its ceiling is the dense bf16 peak / 6 (the fp32 pipe's own 157.3 TFLOP/s is not what binds it);
    configuration 17 runs on the fp32 matrix pipe."""
    from jamie_amd import engine as je
    if eng.fcfg.get('enc0', -1) in (je.F32_CFG_X3, 20):
        return (PEAK_BF16_MFMA_TFLOPS / X3_PRODUCTS, F32_X3_ROOFLINE_KERNEL,
                f'fp32 products on the bf16 matrix pipe: every element cut into three bf16 pieces, {X3_PRODUCTS} v_mfma_f32_32x32x16_bf16 per '
                f'fp32 product; achieved / peak are fp32-equivalent TFLOP/s, peak = {PEAK_BF16_MFMA_TFLOPS:.0f} / {X3_PRODUCTS} '
                f'(the fp32 pipe\'s own peak is {PEAK_F32_MFMA_TFLOPS})')
    return PEAK_F32_MFMA_TFLOPS, F32_ROOFLINE_KERNEL, 'fp32 matrix pipe (v_mfma_f32_32x32x2_f32)'
PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (never the 2:1-sparsity figure)
INGEST_TBPS = 17.0               # L2 -> LDS ingest of the whole chip: 256 CUs x 66 GB/s (MI355X_MICROARCH.md 'Indexed rows: gather into LDS';
                                 # DESIGN.md §4): what the M = 512 GEMM launches are bound by in bf16
STEADY_STEPS = 200               # the `steady_state` sub-record: more steps timed BEHIND the contract's K steps
OTHER_BUDGET_S = 150.0           # `other_configs` sub-records (C4 bf16, C5's dimensions fp32): skipped, and said so, beyond this
LEG_BUDGET_S = 240.0             # N > 1: a sub-record leg (another data-parallel arrangement, timed behind the headline) is abandoned after this long


def synth_shard(n_cells, lo, hi, dims, rank, world, device, device_noise=False):
    """SURVEY.md §8(d) generator: rng = np.random.default_rng(0); Z ~ N(0,1) [N, 16]; X_i = Z A_i + 0.1 E_i with
    A_i ~ N(0,1) [16, d_i], E_i ~ N(0,1); fp32; then standardised per feature (what `preclass(axis=0)` does in the
    reference, jamie.py:462-465).  Draw order: Z, every A_i, then every E_i, so that Z and the A_i are the same for every
    world size; with one rank the E_i continue the same generator, a rank of a larger world draws the E_i of its row shard
    from default_rng([0, 1 + rank]) (the shared stream cannot be advanced past another rank's rows)."""
    rng = np.random.default_rng(0)
    Z = rng.standard_normal((n_cells, 16), dtype=np.float32)[lo:hi]
    A = [rng.standard_normal((16, d), dtype=np.float32) for d in dims]
    erng = rng if world == 1 else np.random.default_rng([0, 1 + rank])
    Zd = torch.from_numpy(Z).to(device)
    # beyond 1e9 noise values per rank (config 5: 7e9 at one rank) the E_i are drawn ON THE DEVICE (torch Philox generator
    # seeded per rank, 100 000 rows at a time into the output) -- numpy would spend minutes and 28 GB of host memory on
    # them; Z and the A_i still come from default_rng(0), so the factor structure is the same for every world size
    on_device = device_noise or (hi - lo) * sum(dims) > 1_000_000_000 or os.environ.get('JAMIE_BENCH_DEVICE_NOISE') == '1'
    gen = torch.Generator(device=device).manual_seed(1000 + rank) if on_device else None
    out = []
    for a in A:
        ad = torch.from_numpy(a).to(device)
        if on_device:
            x = torch.empty(hi - lo, a.shape[1], device=device, dtype=torch.float32)
            for r0 in range(0, hi - lo, 100_000):
                blk = x[r0:r0 + 100_000]
                blk.normal_(generator=gen).mul_(0.1).addmm_(Zd[r0:r0 + 100_000], ad)
            mean = x.mean(0)
            var = torch.zeros_like(mean)
            for r0 in range(0, hi - lo, 100_000):
                var += (x[r0:r0 + 100_000] - mean).square().sum(0)
            std = (var / (hi - lo)).sqrt()
            for r0 in range(0, hi - lo, 100_000):
                x[r0:r0 + 100_000].sub_(mean).div_(std)
        else:
            E = torch.from_numpy(erng.standard_normal((hi - lo, a.shape[1]), dtype=np.float32)).to(device)
            x = Zd @ ad + 0.1 * E
            del E
            x = (x - x.mean(0)) / x.std(0, unbiased=False)
        out.append(x.contiguous())
    return out


ASSUMED_COLLECTIVE_LATENCY_US = 35.0   # issue -> first byte of a collective on an idle wire: a 256-byte all-reduce takes 35 us in a ONE-rank RCCL 2.26.6 group on the box (tools/rccl_probe.py)
ASSUMED_COLLECTIVE_GAP_US = 5.0        # between collectives queued back to back (the next one's launch hides under the current transfer)


def dp_model(trace_ms, step_ms_dry, step_ms_one, world, msg_scale_other, other_name):
    """Exposure model of the data-parallel exchange for `world` GPUs from ONE GPU's timeline (bench.py --dry-run-world).
    The dry run's HIP events give, relative to the step's first launch, when the backward pass announces each gradient message,
    where the step starts waiting for them (`finish`), when the updated weights' all-gathers are issued (sharded optimiser) and
    where the NEXT forward pass needs each gathered region (`wait:<layer>`).  All collectives go over ONE wire, in issue order:
    a collective starts ASSUMED_COLLECTIVE_LATENCY_US after it is issued, or ASSUMED_COLLECTIVE_GAP_US after the one before it
    ends, whichever is later, and then takes bytes / algbw with, for an ASSUMED bus bandwidth ASSUMED_BUS_GBS (a ring over
    xGMI is per-link bound: tools/rccl_probe.py measures the real figure): all-reduce algbw = busbw n / (2 (n - 1)),
    reduce-scatter and all-gather algbw = busbw n / (n - 1).  A few steady-state steps are simulated: the main stream stalls where
    it waits for a collective that has not finished, and the stall pushes everything behind it.  Collectives and kernels are
    assumed not to slow each other down (they will, somewhat: both use HBM).  Returns the block printed as `dp_model`."""
    n = world
    bw_ar = ASSUMED_BUS_GBS * n / (2.0 * (n - 1))
    bw_half = ASSUMED_BUS_GBS * n / (n - 1.0)
    lat, gap = ASSUMED_COLLECTIVE_LATENCY_US * 1e-3, ASSUMED_COLLECTIVE_GAP_US * 1e-3            # ms
    ev = sorted(trace_ms, key=lambda e: e[2])
    # the traced steps run a little longer than the timed ones (event records): positions are kept as fractions of the step
    t_end = [t for k, _, t in ev if k == 'step_end']
    if t_end and t_end[0] > 0:
        ev = [(k, b, t * step_ms_dry / t_end[0]) for k, b, t in ev if k != 'step_end']
    sharded = any(k == 'reduce_scatter' for k, _, _ in ev)
    out = {'world': n, 'optimizer': 'sharded' if sharded else 'replicated', 'assumed_bus_bandwidth_GBps': ASSUMED_BUS_GBS,
           'assumed_collective_latency_us': ASSUMED_COLLECTIVE_LATENCY_US, 'assumed_gap_between_queued_collectives_us': ASSUMED_COLLECTIVE_GAP_US,
           'all_reduce_algorithm_bandwidth_GBps': bw_ar, 'reduce_scatter_all_gather_algorithm_bandwidth_GBps': bw_half,
           'events': [{'kind': k, 'bytes': int(b), 'at_us': 1e3 * t} for k, b, t in ev],
           'traced_step_us': 1e3 * step_ms_dry,
           'step_us_dry_run': 1e3 * step_ms_dry, 'step_us_one_gpu': 1e3 * step_ms_one}

    def simulate(scale):
        wire, t0, gathered, dur, stalls = 0.0, 0.0, {}, step_ms_dry, {}
        for it in range(8):
            shift, order = 0.0, []
            stalls = {'forward_waits_us': 0.0, 'gradient_wait_us': 0.0, 'norm_all_reduce_us': 0.0}
            for k, b, t in ev:
                now = t0 + t + shift
                if k.startswith('wait:'):
                    end = gathered.get(k[5:])
                    if end is not None and end > now:
                        shift += end - now
                        stalls['forward_waits_us'] += 1e3 * (end - now)
                elif k in ('reduce_scatter', 'message'):
                    wire = max(wire + gap, now + lat) + (b * scale / 1e6) / (bw_half if k == 'reduce_scatter' else bw_ar)
                elif k == 'finish':
                    if wire > now:
                        shift += wire - now
                        stalls['gradient_wait_us'] += 1e3 * (wire - now)
                    if sharded:                 # the partial sums of squares: one small all-reduce the main stream waits for
                        shift += lat
                        stalls['norm_all_reduce_us'] += 1e3 * lat
                        wire = max(wire, t0 + t + shift)
                elif k == 'all_gather':
                    wire = max(wire + gap, now + lat) + (b / 1e6) / bw_half
                    order.append(wire)
            names = [k[5:] for k, _, _ in ev if k.startswith('wait:')]
            gathered = dict(zip(names, order))
            dur = step_ms_dry + shift
            t0 += dur
        return dur, stalls

    grad_bytes = sum(b for k, b, _ in ev if k in ('reduce_scatter', 'message'))
    for name, scale in (('as_run', 1.0), (other_name, msg_scale_other)):
        pred, stalls = simulate(scale)
        out[name] = {'gradient_message_bytes': int(grad_bytes * scale),
                     'weight_all_gather_bytes': int(sum(b for k, b, _ in ev if k == 'all_gather')),
                     'exposed_us': 1e3 * (pred - step_ms_dry), 'stalls': stalls,
                     'predicted_step_us': 1e3 * pred, 'predicted_scaling_efficiency': step_ms_one / pred,
                     'predicted_speedup': n * step_ms_one / pred}
    return out


def flops_per_cell(dims, L):
    """F_cell = 6 P_mm - 4 sum d_i^2, P_mm = sum(8 d^2 + 3 d L)."""
    pmm = sum(8 * d * d + 3 * d * L for d in dims)
    return 6 * pmm - 4 * sum(d * d for d in dims)


def host_cpu():
    """(model string, physical cores usable by this process, logical CPUs usable) from /proc/cpuinfo and the affinity mask."""
    model, cores = 'unknown', set()
    try:
        allowed = os.sched_getaffinity(0)
    except AttributeError:
        allowed = set(range(os.cpu_count() or 1))
    try:
        cpu = phys = core = None
        for ln in open('/proc/cpuinfo'):
            k, _, v = ln.partition(':')
            k, v = k.strip(), v.strip()
            if k == 'processor':
                cpu, phys, core = int(v), 0, None
            elif k == 'model name':
                model = v
            elif k == 'physical id':
                phys = int(v)
            elif k == 'core id':
                core = int(v)
            elif not k and cpu is not None:
                if cpu in allowed:
                    cores.add((phys, core if core is not None else cpu))
                cpu = None
    except OSError:
        pass
    return model, (len(cores) or len(allowed)), len(allowed)


def cpu_baseline(dims, L, B, budget_s=24.0):
    """The CPU oracle (plain-PyTorch restatement of the reference step, pinned to the reference by the golden fixtures)
    timed on this box's host cores: same dims, fp32, B = 512, default dropout, N capped at 8192, first
    step dropped; at n = all physical cores, at n = 8, and at the fastest of a short probe; the headline `value` / `cores`
    is the fastest of them
    (the oracle's elementwise ops and dropout masks stop scaling well before a many-core host is full)."""
    from oracle import jamie_oracle as orc
    model, n_phys, n_logical = host_cpu()
    torch.manual_seed(666)
    P, Bf = orc.init_state(dims, L)
    for v in P.values():
        v.requires_grad_(True)
    opt = orc.Adam(P.values(), 1e-3)
    p = orc.default_dropout(dims)
    rng = np.random.default_rng(0)
    n = 8192
    Z = rng.standard_normal((n, 16)).astype(np.float32)
    A = [rng.standard_normal((16, d)).astype(np.float32) for d in dims]
    data = [torch.from_numpy(Z @ a + 0.1 * rng.standard_normal((n, a.shape[1])).astype(np.float32)) for a in A]
    data = [(x - x.mean(0)) / x.std(0) for x in data]
    eye, zero = torch.eye(B), torch.zeros(B, B)
    np.random.seed(42)

    def one():
        idx = np.random.choice(range(n), B, replace=min(dims) < B)
        X = [d[idx] for d in data]
        noise = orc.draw_noise(dims, L, B, p)
        orc.train_step(P, Bf, opt, X, eye if len(dims) == 2 else None, zero if len(dims) == 2 else None, noise, p, 0.5)

    def timed(threads, max_steps, seconds):
        torch.set_num_threads(threads)
        one()                                   # warm-up at this thread count (first step dropped)
        t0 = time.perf_counter()
        steps = 0
        while steps < max_steps and (time.perf_counter() - t0 < seconds or steps < 3):
            one()
            steps += 1
        dt = time.perf_counter() - t0
        return {'value': B * steps / dt, 'unit': 'cells/s', 'cores': threads, 'steps': steps, 'seconds': round(dt, 2)}

    forced = int(os.environ.get('JAMIE_CPU_THREADS', '0'))
    n_all = forced or n_phys
    recs = {'all_physical_cores': timed(n_all, 40, budget_s * 0.4)}
    if not forced:
        recs['n8'] = timed(8, 20, budget_s * 0.3) if n_all != 8 else dict(recs['all_physical_cores'])
        best = None
        for t in sorted({16, 24, 32, 48} - {n_all, 8}):
            if t <= n_logical:
                r = timed(t, 3, 0.0)
                if best is None or r['value'] > best['value']:
                    best = r
        if best is not None and best['value'] > max(r['value'] for r in recs.values()):
            recs['fastest_probed'] = timed(best['cores'], 30, budget_s * 0.3)      # a proper sample at the fastest count
    # the headline record is the FASTEST thread count measured (the oracle's elementwise ops and dropout masks stop scaling
    # long before a 128-core host is full, so "all physical cores" alone would flatter the GPU/CPU ratio)
    top = max(recs.values(), key=lambda r: r['value'])
    out = dict(top)
    out.update({'kind': 'port', 'cpu_model': model, 'physical_cores': n_phys, 'logical_cpus': n_logical,
                'sample': f"{top['steps']} steps of B={B} at dims={tuple(dims)}, L={L}, N capped at {n}, fp32, "
                          f"{top['seconds']} s on {top['cores']} torch threads (the fastest of: all {n_all} physical cores "
                          f"usable by the process, 8, and a probe of 16/24/32/48; {model}, {n_logical} logical CPUs)"})
    out.update(recs)
    return out


def _static_traffic(key):
    """profiles/traffic.json[key] -> (record or None, note): HBM bytes per launch from separate rocprofv3 --pmc passes of this
    command (FETCH_SIZE x 2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section), committed with the profile they came from."""
    tf = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        rec = json.load(open(tf)).get(key)
    except Exception:       # noqa: BLE001
        rec = None
    if not rec or rec.get('hbm_bytes_per_launch') is None:
        return None, None
    return rec, (f"profiles/traffic.json[{key}] (static: FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 --pmc passes of this "
                 f"command, profile {rec.get('source')}; NOT measured in this run)")


def encoder_gemm_record(eng, ms, config, dtype):
    """Roofline of the kernel north_star names -- the encoder's first Linear (reference model.py:151: [B, d] x [2d, d]^T, both
    modalities in one grouped launch) -- from its HIP-event launch time `ms` (median over the timed steps): against the dense MFMA
    peak of the dtype, against HBM on its algorithmic bytes (operands once + ONE fp32 output; the split-K slabs the launch
    really writes are in `hbm_bytes`, from the counters) and against the resource that binds it at M = 512, the L2 -> LDS
    ingest of the CUs (INGEST_TBPS): every workgroup pulls its (BM + BN) x K operand panels in, so the launch moves
    sum(tiles x K x (BM + BN)) x element bytes through the CUs whatever the caches hold."""
    from jamie_amd import _native as nv
    from jamie_amd import engine as je
    B, dims = eng.B, eng.dims
    es = 2 if dtype == 'bf16' else 4
    flop = 4.0 * B * sum(d * d for d in dims)
    peak, f32_note = (PEAK_BF16_MFMA_TFLOPS, None) if dtype == 'bf16' else f32_roofline_terms(eng)[::2]
    alg = sum(B * d * es + 2 * d * d * es + B * 2 * d * 4 for d in dims)
    ingest, tiles, slices = 0.0, [], []
    for i, d in enumerate(dims):
        if dtype == 'bf16':
            cfg = eng.gcfg.get('enc0', -1)
            bm, bn = je.BF16_TILE.get(cfg, nv.gemm_bf16_tile(B, 2 * d))
        else:
            bm, bn = nv.gemm_tile(nv.NT, B, 2 * d, d, eng.fcfg.get('enc0', -1))
        ingest += -(-B // bm) * -(-2 * d // bn) * d * (bm + bn) * es
        tiles.append([int(bm), int(bn)])
        slices.append(int(eng.ws[i]['sk']['enc0']))
    rec, note = _static_traffic(f'{config}_{dtype}_encoder_gemm')
    hbm = rec['hbm_bytes_per_launch'] if rec else None
    prof_us = rec.get('rocprofv3_avg_launch_us') if rec else None
    t = ms * 1e-3
    out = {'kernel': 'forward d -> 2d Linear of the encoder (reference model.py:151), both modalities in one launch',
           'avg_launch_ms': ms, 'flop': flop, 'tflops': flop / t / 1e12, 'peak_tflops': peak, 'frac_mfma': flop / t / 1e12 / peak,
           'avg_launch_note': 'HIP events around the launch, median over the timed steps.  It is the first launch behind clip + Adam: an '
                              'event between the two makes the optimiser\'s write-back drain visible inside this bracket (~+8 us over the '
                              'kernel\'s own duration: rocprofv3_avg_launch_us, static, from the committed profile)',
           'rocprofv3_avg_launch_us': prof_us, 'frac_mfma_rocprofv3': (flop / (prof_us * 1e-6) / 1e12 / peak) if prof_us else None,
           'algorithmic_bytes': alg, 'frac_hbm_algorithmic': alg / t / 1e9 / PEAK_HBM_GBS,
           'hbm_bytes': hbm, 'hbm_bytes_source': note, 'frac_hbm': (hbm / t / 1e9 / PEAK_HBM_GBS) if hbm else None,
           'tile': tiles, 'k_slices': slices, 'ingest_model_bytes': ingest, 'ingest_peak_TBps': INGEST_TBPS,
           'frac_ingest': ingest / t / 1e12 / INGEST_TBPS,
           'binding': 'L2 -> LDS ingest of the CUs' if dtype == 'bf16' else 'MFMA: ' + f32_note}
    return out


def f32_record(model_dims, L, B, data, n_rows, rep, dev, steps=100, warmup=40, config='c2'):
    """The parity configuration (fp32 tensors; the large products on the bf16 matrix pipe as six MFMAs on three-piece cuts, or on the
    fp32 pipe with --tune f32_x3=False) on the same workload: cells/s, ms/step and the roofline of ITS dominant kernel, the forward
    d <-> 2d Linear GEMM launch (north_star: >= 60 % of the binding roofline on the encoder matmul).  Runs in a child process that
    starts cold: 40 untimed steps (the clock settles over the first ~15), 100 timed ones -- 0.15 s."""
    from jamie_amd.engine import TrainEngine
    from jamie_amd.model import edModelVar
    torch.manual_seed(666)
    model = edModelVar(model_dims, L, device=dev)
    eng = TrainEngine(model, B, lr=1e-3, seed=666, compute_dtype='f32')
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    eng.set_kl_anneal(0.5)
    plan = eng.make_plan(data, idx, n_rows, rep, None)
    for _ in range(warmup):
        eng.run_plan(plan)
    torch.cuda.synchronize()
    eng.enable_kernel_timing('enc_gemm', 'enc0_gemm', every=4)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_plan(plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm = eng.kernel_timing_ms('enc_gemm', 'all')
    tm0 = eng.kernel_timing_ms('enc0_gemm', 'all')
    gemm_flop = 4.0 * B * sum(d * d for d in model_dims)
    achieved = gemm_flop / (tm['median'] * 1e-3) / 1e12
    cells_s = B * steps / dt
    total = eng.read_losses()[1]
    peak, kname, pnote = f32_roofline_terms(eng)
    return {'value': cells_s, 'unit': 'cells/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup,
            'dtype': 'f32', 'final_loss': total,
            'roofline': {'bound': 'mfma', 'kernel': kname + ' (Linear d<->2d forward '
                                                     'GEMM, both modalities in one launch; 4 launches/step)',
                         'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'peak_note': pnote,
                         'frac': achieved / peak, 'avg_launch_ms': tm['median'], 'flop_per_launch': gemm_flop,
                         'whole_step_frac': cells_s * flops_per_cell(model_dims, L) / (peak * 1e12),
                         'encoder_gemm': encoder_gemm_record(eng, tm0['median'], config, 'f32')}}


def run_side_leg(kind, args, timeout_s=240):
    """One sub-record (`f32`, `c4_bf16`, `c5dims_f32`) in a CHILD process (bench.py --side-leg): a sub-record must never lose the
    headline, and a process that meets a GPU fault does not come back to print anything.  The child is started, never exec'ed into."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--side-leg', kind, '--config', args.config, '--batch', str(args.batch)]
    if args.tune:
        cmd += ['--tune', args.tune]
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    t0 = time.perf_counter()
    try:
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout_s, env=env)
    except subprocess.TimeoutExpired:
        return {'value': None, 'error': f'child process: no result within {timeout_s} s'}
    for ln in reversed(p.stdout.splitlines()):
        if ln.startswith('{"side_leg"'):
            rec = json.loads(ln)['record']
            rec['child_process_seconds'] = round(time.perf_counter() - t0, 1)
            return rec
    return {'value': None, 'error': f'child process exited with {p.returncode}: ' + (p.stderr.strip().splitlines() or ['no output'])[-1][:300]}


def other_config_record(name, dtype, dev, B=512, steps=20, warmup=5):
    """One more single-GPU configuration of BASELINE.json timed by the same process (driver-timed figures for
    C4 and for C5's dimensions): the workload at its own size (synthetic cells as in the headline, the noise term drawn on the
    device to keep the leg short), plan construction, `warmup` untimed and `steps` timed steps, the encoder-GEMM roofline."""
    from jamie_amd.engine import TrainEngine, kl_anneal
    from jamie_amd.model import edModelVar
    t_leg = time.perf_counter()
    n_cells, dims, L = CONFIGS[name]
    pad = 8 if (dtype == 'bf16' and any(d % 8 for d in dims)) else 1
    data = synth_shard(n_cells, 0, n_cells, dims, 0, 1, dev, device_noise=True)
    torch.manual_seed(666)
    model = edModelVar(dims, L, device=dev, pad_features=pad)
    eng = TrainEngine(model, B, lr=1e-3, seed=666, compute_dtype=dtype)
    data = eng.pad_cells(data)
    idx = torch.zeros(B, dtype=torch.int32, device=dev)
    eng.set_kl_anneal(kl_anneal(0, 2500, 10000))
    eng.enable_kernel_timing('enc0_gemm')
    rep = min(dims) < B and len(dims) == 2
    plan = eng.make_plan(data, idx, n_cells, rep, None)
    for _ in range(warmup):
        eng.run_plan(plan)
    torch.cuda.synchronize()
    eng.enable_kernel_timing('enc0_gemm', every=2)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.run_plan(plan)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm0 = eng.kernel_timing_ms('enc0_gemm', 'all')
    total = eng.read_losses()[1]
    if not np.isfinite(total):
        raise RuntimeError('non-finite loss')
    rec = {'value': B * steps / dt, 'unit': 'cells/s', 'ms_per_step': 1e3 * dt / steps, 'steps': steps, 'warmup': warmup, 'dtype': dtype,
           'workload': f'{name}: {len(dims)}-modality synthetic {n_cells} cells x {tuple(dims)} features, latent={L}, B={B}, one GPU '
                       f'(noise term of the generator drawn on the device)',
           'parameters': model.num_parameters(), 'flop_per_cell': flops_per_cell(dims, L), 'final_loss': total,
           'encoder_gemm': encoder_gemm_record(eng, tm0['median'], name, dtype) if tm0 else None}
    if dtype == 'f32':
        rec['whole_step_frac_mfma'] = rec['value'] * flops_per_cell(dims, L) / (f32_roofline_terms(eng)[0] * 1e12)
    rec['leg_seconds'] = round(time.perf_counter() - t_leg, 1)
    del plan, eng, model, data
    torch.cuda.empty_cache()
    return rec


def allreduce_probe(element_counts, dev, world, iters=10):
    """Measured all-reduce bandwidth of THIS job's communicator at the step's own message sizes (every rank calls this): per
    distinct message (element count) and per message dtype, 3 untimed + `iters` timed all-reduces, each bracketed by HIP events
    on the issuing stream; algbw = bytes / time, busbw = algbw x 2 (n - 1) / n (what one link direction carries in a ring).
    Read it against SURVEY.md 5's estimates for config 3's 161 MB (fp32) gradient: a single ring at ~90 GB/s algbw is 1.8 ms,
    direct reduce-scatter + all-gather over all 7 xGMI links ~0.26 ms."""
    dist = torch.distributed
    out = []
    for n in sorted(set(int(c) for c in element_counts if c > 0)):
        for dt, es, nm in ((torch.bfloat16, 2, 'bf16'), (torch.float32, 4, 'f32')):
            buf = torch.zeros(n, dtype=dt, device=dev)
            for _ in range(3):
                dist.all_reduce(buf)
            torch.cuda.synchronize()
            dist.barrier()
            ts = []
            for _ in range(iters):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dist.all_reduce(buf)
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1))
                buf.zero_()
            med = float(np.median(ts))
            alg = n * es / (med * 1e-3) / 1e9 if med > 0 else None
            out.append({'elements': n, 'dtype': nm, 'bytes': n * es, 'median_us': 1e3 * med, 'min_us': 1e3 * float(min(ts)),
                        'algbw_GBps': alg, 'busbw_GBps': alg * 2.0 * (world - 1) / world if alg else None, 'iters': iters})
            del buf
    return out


def rccl_record(world, log_dir):
    """What RCCL reported while it built the communicator (NCCL_DEBUG=INFO, subsystem INIT, rank 0's log file): evidence
    that the collective library saw `world` ranks, the channels / rings / trees it set up and the transports it chose."""
    import glob
    import re
    dist = torch.distributed
    rec = {'backend': dist.get_backend(), 'world': dist.get_world_size(), 'algo': None}
    try:
        rec['version'] = '.'.join(str(v) for v in torch.cuda.nccl.version())
    except Exception:       # noqa: BLE001
        rec['version'] = None
    try:
        text = ''
        for f in sorted(glob.glob(os.path.join(log_dir, 'rccl_*.log'))):
            text += open(f, errors='replace').read()
        if text:
            ranks = re.findall(r'nranks (\d+)', text, flags=re.I)
            chans = re.findall(r'(\d+) coll channels', text)
            rec['init_log'] = {
                'nranks_seen': sorted({int(r) for r in ranks}),
                'coll_channels': sorted({int(c) for c in chans}),
                'rings': len(re.findall(r'Channel \d+/\d+ *:', text)),
                'trees': len(re.findall(r'Trees \[', text)),
                'p2p_lines': len(re.findall(r'via P2P', text)),
                'algo_lines': [ln.split('NCCL INFO')[-1].strip()[:160] for ln in text.splitlines()
                               if re.search(r'Algo|algorithm|protocol', ln)][:6]}
            if rec['init_log']['algo_lines']:
                rec['algo'] = rec['init_log']['algo_lines'][0]
    except Exception as e:       # noqa: BLE001
        rec['init_log'] = {'error': str(e)}
    return rec


def visible_gpus(root='/sys/class/kfd/kfd/topology/nodes'):
    """GPUs visible to this process, counted WITHOUT a HIP call (the launcher parent must never initialise the runtime: it
    fork+execs the ranks): KFD's topology nodes with SIMDs, cut down by HIP_/ROCR_/CUDA_VISIBLE_DEVICES when those hold plain
    index lists.  None if sysfs has nothing to say: every rank checks for itself in main() anyway (and exits non-zero)."""
    import glob
    n = 0
    try:
        for f in glob.glob(os.path.join(root, '*', 'properties')):
            props = dict(ln.split(None, 1) for ln in open(f).read().splitlines() if ' ' in ln)
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    except (OSError, ValueError):
        return None
    if n == 0:
        return None
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is None:
            continue
        ids = [x for x in v.split(',') if x.strip() != '']
        if all(x.strip().lstrip('-').isdigit() for x in ids):
            n = min(n, len([x for x in ids if int(x) >= 0]))
        # (UUID-style lists: leave the count to the ranks' own check)
    return n


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher (WORLD_SIZE unset): be the launcher.  Called BEFORE anything
    touches the GPU (devices are counted from sysfs, `visible_gpus`: no HIP call in this parent); the children are fresh processes of this script, never an
    exec of a process that holds the device.  Rank 0's stdout is relayed (the ONE JSON line), every rank's stderr goes to
    ours.  Returns the exit code: 0 only if every rank exited 0."""
    import signal
    import socket
    import subprocess
    share = os.environ.get('JAMIE_SHARE_GPU') == '1'           # test hook: the ranks share cuda:0 (jamie_amd/distributed.py)
    have = visible_gpus()
    if have is not None and have < (1 if share else n):
        print(f'bench.py: --gpus {n} but {have} GPU(s) visible to this process; refusing to print a line for fewer GPUs '
              f'than requested', file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    base.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')         # dmabuf IPC: what RCCL needs on this pool's hosts
    base.setdefault('OMP_NUM_THREADS', str(max(1, (os.cpu_count() or n) // n)))
    procs = []
    for r in range(n):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, start_new_session=True))
    # rank 0's output is read while the ranks run (a full pipe must never block it); a rank that dies leaves the others in a
    # collective that cannot complete: they are ended (exact pids / their own process groups) after a grace period
    import threading
    out0 = []
    rd = threading.Thread(target=lambda: out0.append(procs[0].stdout.read()), daemon=True)
    rd.start()
    rc, failed_at = 0, None
    while any(p.poll() is None for p in procs):
        time.sleep(0.2)
        bad = [p for p in procs if p.poll() not in (None, 0)]
        if bad and failed_at is None:
            failed_at, rc = time.time(), bad[0].returncode
        if failed_at is not None and time.time() - failed_at > 15.0:
            for p in procs:
                if p.poll() is None:
                    try:
                        os.killpg(p.pid, signal.SIGKILL)
                    except OSError:
                        pass
    rd.join(10.0)
    for r, p in enumerate(procs):
        if p.returncode != 0:
            print(f'bench.py: rank {r} exited with code {p.returncode}', file=sys.stderr)
            rc = rc or p.returncode or 1
    text = (out0[0] if out0 else b'').decode(errors='replace')
    sys.stdout.write(text)
    sys.stdout.flush()
    if rc == 0 and not any(ln.startswith('{') for ln in text.splitlines()):
        print('bench.py: rank 0 printed no JSON line', file=sys.stderr)
        rc = 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='c2', choices=sorted(CONFIGS))
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--cells', type=int, default=0, help='override the configuration\'s cell count (tests: config c5\'s code path at a '
                    'reduced N; the JSON line names the count it ran)')
    ap.add_argument('--dtype', default=None, choices=['bf16', 'f32'],
                    help='GEMM operand type: bf16 (BASELINE config 2; fp32 accumulate/master; the default except for c5) or f32 '
                         '(the parity configuration; BASELINE config 5 is quoted in fp32)')
    ap.add_argument('--grad-comm', default='auto', choices=['auto', 'f32', 'bf16'],
                    help='dtype of the gradient all-reduce messages (auto: the compute dtype; tests/test_host_cpu.py::test_bf16_message_sum_keeps_the_clip_norm)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--grad-fp32', action='store_true', help='A/B (bf16 mode, one GPU): fp32 weight gradients (TrainEngine(grad_bf16=False)); default: the large dW launches '
                    'store their results as bf16, as the weight gradient of a torch.autocast(bfloat16) Linear is')
    ap.add_argument('--dry-run-world', type=int, default=0,
                    help='one GPU: run the per-rank step of an N-rank job with the collectives skipped (message casts, side stream, '
                         'one-pass norm of the "reduced" gradient): the compute-side cost of data parallelism; NOT a benchmark line')
    ap.add_argument('--dp-optimizer', default='replicated', choices=['auto', 'sharded', 'replicated'],
                    help='N > 1: replicated (the headline: north_star\'s ONE all-reduce of the gradient per step + the full update on '
                         'every rank); sharded = reduce-scatter the large gradient regions, clip + Adam over 1/N of the parameters per '
                         'rank, all-gather the updated weights under the next forward pass (distributed.ShardedGradExchange; timed '
                         'behind the headline as the `sharded_optimizer` sub-record); auto = sharded where the world size divides the regions')
    ap.add_argument('--no-f32-record', action='store_true', help='skip the short fp32 (parity configuration) leg')
    ap.add_argument('--pipeline', action='store_true',
                    help='clip + Adam on a second stream under the next forward pass (+2-3 %; default: main stream, so '
                         'that the roofline kernel is timed alone)')
    ap.add_argument('--transposed-weight-copies', action='store_true',
                    help='bf16 A/B: dX products on transposed bf16 weight copies (the earlier scheme) instead of reading W as stored')
    ap.add_argument('--no-skinny-tr', action='store_true', help='bf16 A/B: head / latent backward launches through the 64x64 kernel on transposed copies (round 1)')
    ap.add_argument('--prefetch', action='store_true', help='A/B: sample + gather the next batch on a side stream under clip + Adam (measured -1 %)')
    ap.add_argument('--side-transposes', action='store_true',
                    help='bf16: transposed weight copies on a side stream under the next forward pass')
    ap.add_argument('--opt-priority', type=int, default=0, help='HIP stream priority of the optimiser stream')
    ap.add_argument('--cpu-budget', type=float, default=24.0)
    ap.add_argument('--settle', type=int, default=0,
                    help='extra untimed plan replays in front of the W warm-up steps (default 0: --warmup is the only warm-up the '
                         'headline gets; the steady-state figure is the `steady_state` sub-record, timed behind the K steps)')
    ap.add_argument('--step-trace', action='store_true', help='diagnostic: a HIP event behind every warm-up and timed step; their spacings '
                    'ride in the line as `step_trace_us` (how the first steps after a cold start differ from the steady state)')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the `other_configs` sub-records (C4 bf16, C5 dims fp32)')
    ap.add_argument('--side-leg', default='', choices=['', 'f32', 'c4_bf16', 'c5dims_f32'],
                    help='(internal) run ONE sub-record and print it: the headline process starts its sub-records as child processes, so '
                         'that nothing that goes wrong in one of them can lose the headline line')
    ap.add_argument('--tune', default=os.environ.get('JAMIE_TUNE', ''),
                    help='A/B measurements (tools/ab.sh): "key=value+key=value" for jamie_amd.engine.tune() -- tile / split-K plans and the '
                         'older variant of every adopted change (engine.TUNING); also read from JAMIE_TUNE')
    args = ap.parse_args()
    if args.dtype is None:
        args.dtype = DEFAULT_DTYPE.get(args.config, 'bf16')
    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # no launcher: start the ranks from here, before this process makes any GPU call
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    # RCCL's own account of the communicator it builds (read back into the JSON line at N > 1): init-time log only
    log_dir = None
    if int(os.environ.get('WORLD_SIZE', '1')) > 1 and 'NCCL_DEBUG' not in os.environ:
        import tempfile
        log_dir = tempfile.mkdtemp(prefix='jamie_rccl_')
        os.environ.update(NCCL_DEBUG='INFO', NCCL_DEBUG_SUBSYS='INIT,GRAPH',
                          NCCL_DEBUG_FILE=os.path.join(log_dir, 'rccl_%h_%p.log'))
    from jamie_amd import distributed as jd
    w_env = int(os.environ.get('WORLD_SIZE', '1'))
    if w_env > 1 and os.environ.get('JAMIE_SHARE_GPU') != '1' and torch.cuda.device_count() < w_env:
        # (a rank process may ask the runtime; checked BEFORE the process group exists so that every rank ends at once)
        raise SystemExit(f'--gpus {w_env} but {torch.cuda.device_count()} GPU(s) visible')
    rank, world, local = jd.init_from_env()
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    from jamie_amd import _native as nv
    nv.require_gpu()
    from jamie_amd.engine import TrainEngine, kl_anneal
    from jamie_amd.model import edModelVar
    if args.tune:
        from jamie_amd import engine as _eng

        def _val(v):
            if v in ('True', 'False', 'None'):
                return {'True': True, 'False': False, 'None': None}[v]
            try:
                return int(v)
            except ValueError:
                try:
                    return float(v)
                except ValueError:
                    return v
        _eng.tune(**{k: _val(v) for k, v in (kv.split('=', 1) for kv in args.tune.split('+') if kv)})

    n_cells, dims, L = CONFIGS[args.config]
    if args.cells > 0:
        n_cells = args.cells
    B = args.batch
    if args.side_leg:
        if args.side_leg == 'f32':
            data_leg = synth_shard(n_cells, 0, n_cells, dims, 0, 1, dev)
            rec = f32_record(dims, L, B, data_leg, n_cells, min(dims) < B and len(dims) == 2, dev, config=args.config)
        else:
            cname, cdt = {'c4_bf16': ('c4', 'bf16'), 'c5dims_f32': ('c5dims', 'f32')}[args.side_leg]
            rec = other_config_record(cname, cdt, dev, B)
        print(json.dumps({'side_leg': args.side_leg, 'record': rec}), flush=True)
        return
    if args.dtype == 'bf16' and any(v % 8 for v in [L, B]):
        args.dtype = 'f32'          # bf16 operands need a latent size and a batch that are multiples of 8
    # feature counts that are not multiples of 8 (config 1: 100, config 4: 500): the bf16 engine pads them (model.py)
    pad = 8 if (args.dtype == 'bf16' and any(d % 8 for d in dims)) else 1
    lo, hi = jd.shard_bounds(n_cells, rank, world)
    data_real = synth_shard(n_cells, lo, hi, dims, rank, world, dev)
    comm = torch.bfloat16 if (args.dtype == 'bf16' and args.grad_comm == 'auto') or args.grad_comm == 'bf16' else None
    n_dp = max(world, args.dry_run_world)
    # the reference's quirk `replace = min(features) < batch_size` (jamie.py:553) belongs to its two-modality loop; the
    # 3-modality generalisation always samples without replacement (duplicates would need a non-identity corr)
    rep = min(dims) < B and len(dims) == 2
    # KL anneal per epoch as in the reference's loop (jamie.py:630-632, min_epochs 2500, epoch_DNN 10000): a device scalar,
    # rewritten (asynchronously) when the step count crosses an epoch boundary
    steps_per_epoch = max(1, int((hi - lo) / B))
    state = {'step': 0, 'epoch': -1}

    def make_exchange(eng_, comm_dtype, mode):
        """(exchange, 'sharded' | 'replicated' | None) for engine `eng_`."""
        if n_dp <= 1:
            return None, None
        if mode != 'replicated':
            try:
                ex = jd.ShardedGradExchange(comm_dtype=comm_dtype, dry_run_world=args.dry_run_world)
                eng_.enable_sharded_optimizer(ex)
                return ex, 'sharded'
            except ValueError as err:
                if mode == 'sharded':
                    raise SystemExit(f'--dp-optimizer sharded: {err}')
        return jd.OverlappedGradAllReduce(comm_dtype=comm_dtype, dry_run_world=args.dry_run_world), 'replicated'

    def build_job(mode):
        """Model, engine, exchange and the recorded plan of the timed job (the recording step is a real step)."""
        torch.manual_seed(666)
        model_ = edModelVar(dims, L, device=dev, pad_features=pad)
        if world > 1:
            jd.broadcast_flat(model_.flat)
        eng_ = TrainEngine(model_, B, lr=1e-3, seed=666 + 7919 * rank, world_size=n_dp, compute_dtype=args.dtype,
                           dx_from_weights=not args.transposed_weight_copies, skinny_tr=not args.no_skinny_tr,
                           grad_bf16=False if args.grad_fp32 else None)
        data_ = eng_.pad_cells(data_real)
        ar_, opt_ = make_exchange(eng_, comm, mode)
        idx_ = torch.zeros(B, dtype=torch.int32, device=dev)      # 'diag' sampling: same rows in both modalities
        eng_.set_kl_anneal(kl_anneal(0, 2500, 10000))
        if args.pipeline:
            eng_.enable_pipeline(args.opt_priority)
        if args.side_transposes:
            eng_.enable_side_transposes()
        eng_.enable_kernel_timing('enc_gemm', 'enc0_gemm', 'adam')
        # the step is a fixed launch sequence on static buffers: record it once, replay it (one foreign call per launch)
        plan_ = eng_.make_plan(data_, idx_, hi - lo, rep, ar_, prefetch=args.prefetch)
        # --settle N (default 0): N more untimed replays.  Round 4 ran 100 of them by default in front of the driver's W = 5; the
        # contract's warm-up is --warmup and nothing else, so the default is gone and the steady-state figure is
        # reported beside the headline instead (`steady_state`: STEADY_STEPS more steps timed behind the K steps).
        for _ in range(max(0, args.settle)):
            eng_.run_plan(plan_)
        torch.cuda.synchronize()
        return model_, eng_, data_, ar_, opt_, idx_, plan_

    mode0 = 'replicated' if (args.pipeline or args.side_transposes or args.transposed_weight_copies) else args.dp_optimizer
    dp_fallback = None
    try:
        model, eng, data, allreduce, dp_opt, idx, plan = build_job(mode0)
    except Exception as err:         # noqa: BLE001  (the same failure on every rank: an API the backend lacks, not a hang)
        if not (mode0 == 'auto' and n_dp > 1):
            raise
        dp_fallback = f'{type(err).__name__}: {err}'[:300]
        model, eng, data, allreduce, dp_opt, idx, plan = build_job('replicated')

    def set_anneal():
        ep = state['step'] // steps_per_epoch
        if ep != state['epoch']:
            state['epoch'] = ep
            eng.set_kl_anneal(kl_anneal(ep, 2500, 10000))
    state['epoch'] = 0
    state['step'] = 2 + max(0, args.settle)

    trace_ev = [] if args.step_trace else None

    def step():
        set_anneal()
        eng.run_plan(plan)
        state['step'] += 1
        if trace_ev is not None and len(trace_ev) < 400:
            e = torch.cuda.Event(enable_timing=True)
            e.record(nv.current_stream())
            trace_ev.append(e)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    if world > 1 and allreduce is not None and hasattr(allreduce, 'enable_exposure'):
        allreduce.enable_exposure(True, every=4)      # HIP events around finish()'s device-side waits (the exposed part of the exchange)
    events = os.environ.get('JAMIE_BENCH_NO_EVENTS') != '1'
    if events:
        # drop the warm-up samples.  In the timed region only the two launches the roofline block needs are bracketed, on every 8th
        # step (every 4th of a run under 16 steps): an event between two launches costs the step ~3 us (it ends the back-to-back
        # overlap of the launches around it) -- with the four forward GEMM launches bracketed as well a sampled step read 590
        # against 545 us (profiles/r05_step_trace_event_overhead.json).  The forward-GEMM average comes from the steady-state leg.
        # (fp32: the roofline kernel IS the forward GEMM launch, and 12 events are 0.3 % of eight 1.3 ms steps: all three labels)
        head_labels = ('enc0_gemm', 'adam') if args.dtype == 'bf16' else ('enc_gemm', 'enc0_gemm', 'adam')
        eng.enable_kernel_timing(*head_labels, every=8 if args.steps >= 16 else 4)
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
    exposure, msg_elems = None, []
    if world > 1 and allreduce is not None and hasattr(allreduce, 'exposed_us'):
        exposure = allreduce.exposed_us()
        msg_elems = allreduce.message_elements()
        allreduce.enable_exposure(False)
    # `steady_state`: STEADY_STEPS more steps timed the same way right BEHIND the contract's region (the driver's W = 5 / K = 20 is
    # 14 ms after a cold start; the GPU's clock and caches settle over the first ~100 steps: 2-3 % on one box).  A sub-record,
    # never `value`.
    steady = None
    timing_head = {k: eng.kernel_timing_ms(k, 'all') for k in head_labels} if events else None
    if args.dry_run_world <= 1:
        if events:
            eng.enable_kernel_timing('enc_gemm', 'enc0_gemm', 'adam', every=25)     # (this leg's own samples: 8 of 200 steps)
        barrier()
        ts = time.perf_counter()
        for _ in range(STEADY_STEPS):
            step()
        barrier()
        ds = time.perf_counter() - ts
        if world > 1:
            t = torch.tensor([ds], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ds = float(t.item())
        steady = {'value': world * B * STEADY_STEPS / ds, 'unit': 'cells/s', 'ms_per_step': 1e3 * ds / STEADY_STEPS, 'steps': STEADY_STEPS,
                  'note': f'{STEADY_STEPS} more steps timed behind the K timed steps of the headline (same barriers, max over ranks)'}
        if events:
            steady['kernel_event_timing_ms'] = {k: eng.kernel_timing_ms(k, 'all') for k in ('enc_gemm', 'enc0_gemm', 'adam')}
            if timing_head is not None and 'enc_gemm' not in timing_head:
                timing_head['enc_gemm'] = steady['kernel_event_timing_ms']['enc_gemm']      # (bf16: bracketed in the steady-state leg only)
    if not events:          # kernel timings from extra steps AFTER the timed region -- on every rank (a step is a collective)
        eng.enable_kernel_timing('enc_gemm', 'enc0_gemm', 'adam')
        for _ in range(20):
            step()
        barrier()
    if dp_opt == 'sharded':           # (outside the timed region: the replicated buffers current again, no all-gather left in flight)
        eng.flush(collective=True)
        barrier()
    ls, total, _ = eng.read_losses()
    if not np.isfinite(total):
        raise SystemExit('non-finite loss in benchmark')
    cells_s = world * B * args.steps / dt
    dp = None
    if args.dry_run_world > 1 and world == 1:
        # the exchange's timeline: events where the backward pass announces each message and where it starts waiting
        traces = []
        for _ in range(12):
            allreduce.enable_trace()
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record(nv.current_stream())
            step()
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(nv.current_stream())
            torch.cuda.synchronize()
            traces.append([(k, b, e0.elapsed_time(ev)) for k, b, ev in allreduce.trace] + [('step_end', 0, e0.elapsed_time(e1))])
        allreduce.enable_trace(False)
        med = [(traces[0][i][0], traces[0][i][1], float(np.median([t[i][2] for t in traces]))) for i in range(len(traces[0]))]
        # the one-GPU step on the same box, for the efficiency the model predicts
        torch.manual_seed(666)
        m1 = edModelVar(dims, L, device=dev, pad_features=pad)
        e1 = TrainEngine(m1, B, lr=1e-3, seed=666, compute_dtype=args.dtype, grad_bf16=False if args.grad_fp32 else None)
        e1.set_kl_anneal(0.5)
        idx1 = torch.zeros(B, dtype=torch.int32, device=dev)
        p1 = e1.make_plan(data, idx1, hi - lo, rep, None)
        for _ in range(30):
            e1.run_plan(p1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            e1.run_plan(p1)
        torch.cuda.synchronize()
        one_ms = 1e3 * (time.perf_counter() - t1) / args.steps
        dp = dp_model(med, 1e3 * dt / args.steps, one_ms, args.dry_run_world, 2.0 if comm is not None else 0.5,
                      'fp32_messages' if comm is not None else 'bf16_messages')
        del p1, e1, m1

    if rank == 0:
        # roofline of the dominant kernel:
        #   f32 : the d <-> 2d Linear forward GEMM launch (both modalities grouped; 4 launches per step, each
        #         4*B*sum(d^2) FLOP) against the exact-fp32 MFMA peak;
        #   bf16: the step is HBM-bound on optimiser traffic; the dominant kernel is clip+Adam:
        #         28 bytes per parameter (read p, g, m, v; write p, m, v) against the HBM peak.
        timing_detail = timing_head if timing_head is not None else {
            'enc_gemm': eng.kernel_timing_ms('enc_gemm', 'all'), 'enc0_gemm': eng.kernel_timing_ms('enc0_gemm', 'all'),
            'adam': eng.kernel_timing_ms('adam', 'all')}
        # event pairs bracket one launch each; a host hiccup between the two records (GC, scheduler) shows up as a
        # multi-millisecond outlier in a handful of the samples, so the per-launch duration is the MEDIAN
        if not timing_detail.get('enc_gemm'):          # (no steady-state leg, e.g. a dry run: the encoder launch stands in)
            timing_detail['enc_gemm'] = timing_detail['enc0_gemm']
        gemm_ms, adam_ms = timing_detail['enc_gemm']['median'], timing_detail['adam']['median']
        kdims = eng.dims                                   # what the kernels multiply (padded feature counts, if any)
        gemm_flop = 4.0 * B * sum(d * d for d in kdims)
        rec, traffic_source = _static_traffic(f'{args.config}_{args.dtype}')
        traffic = rec['hbm_bytes_per_launch'] if rec else None
        enc0 = encoder_gemm_record(eng, timing_detail['enc0_gemm']['median'], args.config, args.dtype) if timing_detail['enc0_gemm'] else None
        if args.dtype == 'f32':
            achieved = gemm_flop / (gemm_ms * 1e-3) / 1e12
            peak, kname, pnote = f32_roofline_terms(eng)
            roof = {'bound': 'mfma', 'kernel': kname + ' (Linear d<->2d forward '
                                              'GEMM, both modalities in one launch; 4 launches/step)',
                    'achieved': achieved, 'peak': peak, 'unit': 'TFLOP/s', 'peak_note': pnote,
                    'frac': achieved / peak, 'avg_launch_ms': gemm_ms, 'flop_per_launch': gemm_flop,
                    'traffic': traffic, 'traffic_source': traffic_source,
                    'whole_step_frac': cells_s / world * flops_per_cell(dims, L) / (peak * 1e12),
                    'encoder_gemm': enc0}
        else:
            n_par = model.layout.total
            adam_launches = len(eng.PIPE_GROUPS) if eng.pipeline else 1
            if dp_opt == 'sharded':          # (the timed launch is the one over this rank's packed shard; `rep` has its own small launch)
                n_par = eng._zs['S']
            adam_bytes = eng.adam_bytes_per_param() * n_par / adam_launches
            achieved = adam_bytes / (adam_ms * 1e-3) / 1e9
            # the step's algorithmic weight-side bytes in THIS dtype: forward + backward read the bf16 weight copy (2 + 2 B per
            # parameter), the dW epilogues write bf16 gradients (2), clip + Adam moves 26 (p, m, v in and out, g in) and writes
            # the bf16 copy (2): 34 B per parameter (fp32 gradients: 38).  SURVEY.md 8(d)'s 44 P is the fp32 byte model; the
            # fraction on it is kept beside this one, named.
            own_bpp = 8.0 + eng.adam_bytes_per_param() if eng.grad_bf16 else 10.0 + eng.adam_bytes_per_param()
            step_bytes = own_bpp * model.layout.total
            step_bytes_f32_model = 44.0 * model.layout.total
            roof = {'bound': 'hbm', 'kernel': f'clip_adam_kernel (global-norm clip + Adam on the flat fp32 buffers; {adam_launches} launch(es)/step'
                                       + (', on the optimiser stream under the next forward pass)' if eng.pipeline else
                                          (f'; sharded optimiser: this launch covers 1/{n_dp} of the large weight regions)' if dp_opt == 'sharded' else ')')),
                    'achieved': achieved, 'peak': PEAK_HBM_GBS, 'unit': 'GB/s', 'frac': achieved / PEAK_HBM_GBS,
                    'avg_launch_ms': adam_ms, 'bytes_per_launch': adam_bytes,
                    'bytes_per_parameter': eng.adam_bytes_per_param(), 'traffic': traffic, 'traffic_source': traffic_source,
                    'whole_step_frac': (step_bytes * cells_s / world / B) / (PEAK_HBM_GBS * 1e9),
                    'whole_step_bytes_per_parameter': own_bpp,
                    'whole_step_frac_on_fp32_byte_model_44P': (step_bytes_f32_model * cells_s / world / B) / (PEAK_HBM_GBS * 1e9),
                    'gemm_bf16_tflops': gemm_flop / (gemm_ms * 1e-3) / 1e12, 'gemm_avg_launch_ms': gemm_ms,
                    'encoder_gemm': enc0}
        out = {
            'metric': 'training cells/sec (two-modality coupled VAE)', 'value': cells_s, 'unit': 'cells/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * dt / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic',
            'config': {'workload': f'{args.config}: {len(dims)}-modality synthetic {n_cells} cells x {tuple(dims)} features, '
                                   f'latent={L}, B={B}/GPU, dropout={model.dropout}, '
                                   + (('bf16 MFMA GEMMs, fp32 accumulate/master/optimiser, ' + ('bf16 weight gradients (autocast semantics), ' if eng.grad_bf16 else 'fp32 weight gradients, ')) if args.dtype == 'bf16' else 'fp32 MFMA, ')
                                   + 'identity P (diag sampling), F=0, KL anneal per epoch',
                       'generator': 'SURVEY.md 8(d): numpy default_rng(0), 16-dim latent factor model + 0.1 noise, standardised per feature'
                                    + ('; the noise term drawn on the device (torch generator per rank)' if (hi - lo) * sum(dims) > 1_000_000_000 or os.environ.get('JAMIE_BENCH_DEVICE_NOISE') == '1' else ''),
                       'cells': n_cells, 'features': list(dims), 'latent': L, 'batch_per_gpu': B, 'settle_steps_before_warmup': max(0, args.settle),
                       'parallelism': f'dp{world}', 'grad_allreduce': ('none' if world == 1 else ('bf16' if comm is not None else 'f32')),
                       'dp_optimizer': dp_opt, **({'dp_optimizer_fallback': dp_fallback} if dp_fallback else {}),
                       'parameters': model.num_parameters(),
                       'flop_per_cell': flops_per_cell(dims, L)},
            'roofline': roof,
            'kernel_event_timing_ms': timing_detail,
            'final_loss': total,
        }
        if steady is not None:
            out['steady_state'] = steady
        if trace_ev:
            torch.cuda.synchronize()
            out['step_trace_us'] = [round(1e3 * a.elapsed_time(b), 1) for a, b in zip(trace_ev[:-1], trace_ev[1:])]
        if args.dry_run_world > 1:
            out['dry_run_world'] = args.dry_run_world
            out['dp_model'] = dp
            out['metric'] += f' [DRY RUN: per-rank compute path of a {args.dry_run_world}-rank step, collectives skipped; not a benchmark line]'
        if world > 1:
            out['rccl'] = rccl_record(world, log_dir) if log_dir else {'backend': torch.distributed.get_backend(),
                                                                        'world': torch.distributed.get_world_size(), 'algo': None}
            # measured, not modelled: the time the step's stream stood still in finish() waiting for messages on the wire
            out['exchange'] = {'exposed_us_per_step': exposure,
                               'messages_per_step': [{'elements': int(n), 'bytes': int(n) * (2 if comm is not None else 4)} for n in msg_elems],
                               'note': 'HIP events on the launch stream around the device-side waits of finish(), every 4th timed step '
                                       '(median / mean / max in us); messages in issue order'}
    if world > 1:
        # every rank: all-reduce bandwidth of this communicator at the step's message sizes (the headline is already measured)
        try:
            probe = allreduce_probe(msg_elems, dev, world)
        except Exception as err:         # noqa: BLE001
            probe = {'error': f'{type(err).__name__}: {err}'[:300]}
        if rank == 0:
            out['rccl']['allreduce_probe'] = probe
    if world > 1 and not args.no_f32_record:
        # N > 1: the same job in its other data-parallel arrangements, timed right behind the headline on every rank (a step is
        # a collective), so that one invocation reports them all:
        #   grad_comm_f32       (bf16 compute, bf16 messages in the headline) fp32 gradient messages: exact sums, twice the bytes
        #   sharded_optimizer   (replicated optimiser in the headline) reduce-scatter + 1/N of clip + Adam per rank + all-gather
        #   replicated_optimizer (sharded optimiser in the headline) all-reduce + the full clip + Adam on every rank
        legs = []
        if args.dtype == 'bf16' and comm is not None and args.grad_comm == 'auto':
            legs.append(('grad_comm_f32', None, args.dp_optimizer, 'the same job with fp32 gradient messages (exact sums)'))
        if dp_opt == 'sharded':
            legs.append(('replicated_optimizer', comm, 'replicated', 'the same job with an all-reduce and the full update on every rank'))
        elif dp_opt == 'replicated' and mode0 == 'replicated' and args.dp_optimizer == 'replicated':
            legs.append(('sharded_optimizer', comm, 'auto', 'the same job with the sharded optimiser (reduce-scatter, 1/N of clip + Adam '
                         'per rank, weight all-gather under the next forward pass)'))
        del plan

        def run_leg(key, comm2, mode2, note):
            torch.manual_seed(666)
            m2 = edModelVar(dims, L, device=dev, pad_features=pad)
            jd.broadcast_flat(m2.flat)
            e2 = TrainEngine(m2, B, lr=1e-3, seed=666 + 7919 * rank, world_size=world, compute_dtype=args.dtype)
            e2.set_kl_anneal(0.5)
            ar2, opt2 = make_exchange(e2, comm2, mode2)
            if key == 'sharded_optimizer' and opt2 != 'sharded':      # (not available for this job: layer sizes / world size)
                if rank == 0:
                    out[key] = {'value': None, 'note': 'the sharded optimiser is not available for this job (falls back to replicated)'}
                return
            idx2 = torch.zeros(B, dtype=torch.int32, device=dev)
            p2 = e2.make_plan(data, idx2, hi - lo, rep, ar2)
            n2 = max(10, args.steps // 4)
            for _ in range(min(args.warmup, 10)):
                e2.run_plan(p2)
            barrier()
            t2 = time.perf_counter()
            for _ in range(n2):
                e2.run_plan(p2)
            barrier()
            d2 = time.perf_counter() - t2
            tt = torch.tensor([d2], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            if rank == 0:
                out[key] = {'value': world * B * n2 / float(tt.item()), 'unit': 'cells/s', 'ms_per_step': 1e3 * float(tt.item()) / n2,
                            'steps': n2, 'grad_allreduce': 'f32' if comm2 is None else 'bf16', 'dp_optimizer': opt2, 'note': note}
            e2.flush(collective=True)

        # the headline is measured: a sub-record leg that hangs (a collective one rank never joins) or throws must not lose it
        import threading
        leg_state = {'key': None}

        def bail():
            if rank == 0:
                out[leg_state['key']] = {'value': None, 'error': f'leg exceeded {LEG_BUDGET_S:.0f} s and was abandoned'}
                print(json.dumps(out), flush=True)
            os._exit(0)

        for key, comm2, mode2, note in legs:
            leg_state['key'] = key
            watchdog = threading.Timer(LEG_BUDGET_S, bail)
            watchdog.daemon = True
            watchdog.start()
            try:
                run_leg(key, comm2, mode2, note)
            except Exception as err:         # noqa: BLE001
                if rank == 0:
                    out[key] = {'value': None, 'error': f'{type(err).__name__}: {err}'[:300]}
            finally:
                watchdog.cancel()
    if world == 1:
        # free the timed engine's buffers before the side legs
        del plan
        if args.dtype == 'bf16' and not args.no_f32_record and len(dims) == 2 and not any(d % 8 for d in dims):
            del eng, model
            torch.cuda.empty_cache()
            out['f32'] = run_side_leg('f32', args)
        if args.config == 'c2' and args.dtype == 'bf16' and not args.no_other_configs and B == 512 and args.dry_run_world <= 1:
            # driver-timed sub-records of the other single-GPU configurations, inside a stated budget: each leg is skipped (and says
            # so) when the process has already spent OTHER_BUDGET_S since it started
            try:
                del data_real, data
            except NameError:
                pass
            torch.cuda.empty_cache()
            out['other_configs'] = {'budget_s': OTHER_BUDGET_S}
            for key, cname, cdt in (('c4_bf16', 'c4', 'bf16'), ('c5dims_f32', 'c5dims', 'f32')):
                spent = time.perf_counter() - T_START
                if spent > OTHER_BUDGET_S:
                    out['other_configs'][key] = {'value': None, 'skipped': f'{spent:.0f} s spent before this leg: over the {OTHER_BUDGET_S:.0f} s budget'}
                    continue
                out['other_configs'][key] = run_side_leg(key, args)
        if not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(dims, L, B, args.cpu_budget)
            out['gpu_over_cpu'] = cells_s / out['cpu_baseline']['value']
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
