"""Training step of JAMIE's coupled VAE on one MI355X: the hot loop body of the reference's
`project_jamie` (jamie/jamie.py:546-741) as a fixed sequence of HIP kernel launches through the C ABI
(`include/jamie_hip.h`): gather -> encoder GEMM/BN/act x2 -> heads GEMM -> latent block -> decoder
GEMM/BN/act x2 -> output GEMM with fused MSE -> backward of all of it -> clip + Adam.

All buffers are allocated once per batch size; no step allocates, synchronises or reads back.
"""
import functools
import math

import numpy as np
import torch

from . import _native as nv
from .model import BN_EPS, BN_MOMENTUM, LRELU_SLOPE

KL_WEIGHT = 32 * 1e-3       # jamie.py:632
ALIGN_WEIGHT = 32           # jamie.py:658
LOSS_NAMES = ['KL', 'Rec', 'CosSim', 'F']

# Tuning knobs of the launch plans and the older variant of every adopted change, for A/B measurements on one box
# (tools/ab.sh "JAMIE_TUNE=key=value+key=value" -> bench.py --tune; profiles/r0*_ab_*.log).  NOT read from the environment: the package reads
# no JAMIE_* variable besides the library path and the two test hooks of distributed.py (tests/test_host_cpu.py asserts it).
# Set them with engine.tune(key=value, ...) BEFORE an engine is built.
TUNING = {
    'bf16_rows': None,            # "cfg:s0,s1[;cfg:s0,s1 for the N < K launches]": tile configuration + K slices of the bf16 forward launches
    'f32_rows': None,             # "[cfg:]s0,s1[;...]": the same for the fp32 forward / dX launches
    'bwd_k_per_slab': 1700.0,     # K per split-K slab of the bf16 dX products
    'sk_skinny': None,            # slab count of the skinny head / latent products
    'f32_dw_cfg': None, 'f32_dx_cfg': None,      # fp32 tile configurations of the dW / dX launches
    'f32_dw_small_cfg': None,     # ... of the skinny layers' dW launch (decoder layer 0, heads)
    'f32_rows_cfg': None,         # ... of the planned forward / dX launches (the planner's K slices stay)
    'prefetch': 1,                # BatchNorm prefetch riders: 0 off, 1 the next product's weights, 2 + saved activations
    'prefetch_f32': '0',          # ... in fp32 mode: '0' off (default: +38 us there), '1' on, 'bwd' backward only
    'stagger': True,              # flat optimiser buffers start 4 KB apart
    'adam_rotate': False,         # clip + Adam walks from the second layer's region (+5 us: off)
    'f32_dx_plan': True, 'f32_fused_norm': True,
    'f32_x3': True,               # fp32 mode: the large products on the bf16 matrix pipe, every fp32 element cut into three bf16 pieces
                                  # (gemm_f32.hip configuration 20: fp32-level error, 1313 -> 1059 us per step at config 2); False: the
                                  # fp32 matrix pipe (configuration 17)
    'fused_da2': True, 'fused_latent': True, 'direct_comm': True, 'cs_ride': True, 'late_dec0_dw': True, 'range_ride': True,
    'defer_final': True, 'fused_sampler': True, 'gather_ride': True,
    'dw_store_nt': True,          # weight gradients stored non-temporally (next read by the optimiser, a backward pass later)
    'split_last_dw': False,       # data parallel (replicated): the last layer's dW in two launches, its first part on the wire early.
                                  # OFF until an N > 1 A/B shows a gain: it adds a launch to the tail of the backward pass and turns
                                  # the merged rep + enc0 message into three all-reduces (~28 us of host enqueue each); bench.py
                                  # --tune split_last_dw=True is the A/B for the first multi-GPU box
    'fused_heads': False,         # bf16 mode, fused latent kernels, L <= 64: the heads' product mu | logvar = a2 W_h^T inside the latent forward
                                  # launch (every workgroup multiplies its 32 cells' rows by the whole 2L x d weight) instead of a GEMM launch
                                  # + 8 split-K slabs.  Built, bit-checked (tests/test_hip_configs.py) and measured SLOWER in round 5: +10.4 us
                                  # per step (profiles/r05_ab_fused_heads_rejected.log) -- a workgroup has to pull the whole weight (0.4 MB)
                                  # and its cells' rows (0.2 MB) through ONE CU's 55-66 GB/s: ~10 us, more than the split-K launch it replaces
    'mse_colpart': True,          # the decoder's output-bias gradient from jamie_mse_cast's per-tile column sums (no fp32 d x_hat in bf16 mode)
    'bn_panel': True,             # bf16 mode: the fp32 pre-activations / upstream gradients travel GEMM -> BatchNorm (-> BatchNorm backward)
                                  # in panels of 16 columns (jamie_hip.h: JAMIE_PANEL): a BatchNorm strip is whole contiguous blocks per slab
    'f32_pipe_solo': 'enc0,enc1',  # fp32, pipelined optimiser: the forward launches of these layers run beside clip + Adam on the optimiser
                                  # stream and take tile configuration 19 (= 17 at ONE workgroup per CU: half of each CU's wave slots
                                  # stay free for the streaming kernel); '' = off
    'f32_dw_group': 4,            # fp32, no gradient exchange: the large layers' dW products wait and go out `n` layers per launch
                                  # (4 layers = 2560 tiles of 128 x 128 = 5.0 rounds of 512 slots; one layer = 1.25 rounds); 1: off
}


def tune(**kw):
    """Set tuning knobs (see TUNING); unknown keys raise.  Plans cached by shape are dropped."""
    for k, v in kw.items():
        if k not in TUNING:
            raise KeyError(f'unknown tuning knob {k!r}')
        TUNING[k] = v
    _plan_f32_rows.cache_clear()


# hyper buffer slots (device float[16]; see include/jamie_hip.h)
H_KL, H_REC, H_ALIGN, H_F = 0, 1, 2, 3
H_LR, H_B1, H_B2, H_EPS, H_MAXNORM, H_GSCALE = 8, 9, 10, 11, 12, 13


def choose_splitk(M, N, K, bm=64, bn=64):
    """Split K so that a problem offers >= ~2 workgroups of 64x64 per CU (512 in all); each slice keeps at
    least 512 of K (160 for the skinny heads / latent products).  Slabs are summed by the consuming kernel.
    Measured (tools/bench_gemm.py): [512x2000x4000] 65 -> 89 TFLOP/s with 2 slices."""
    tiles = math.ceil(M / bm) * math.ceil(N / bn)
    if tiles >= 384:
        return 1
    s = max(1, math.ceil(512 / tiles))
    # (skinny heads / latent products: slices of >= 125, at most 8 -- what the fused latent kernels sum in ONE round trip; at
    #  config 2 six slabs instead of three took 3-5 us off the step and eight another 5, profiles/r02_ab_skinny_slabs.log:
    #  the launch is a latency chain of k-steps, not bandwidth)
    return int(max(1, min(s, K // 512) if N > 64 else min(s, 8, K // 125)))


# ---- bf16 GEMM launch plans (tile configuration + per-problem split-K), from tools/bench_gemm_bf16.py ----
# The LDS-DMA kernels pull ~17 TB/s from L2 chip-wide whatever the tile (DESIGN.md, bf16 GEMM), so a product's time
# is the A+B bytes pulled into the CUs: a 256x128 / 128x128 tile moves 3/8 / 1/2 of the bytes of 64x64, but at
# M = batch = 512 it only fills the 256 CUs when K is split.  Deterministic rules (no run-time tuning: split-K
# changes the summation order):
#   rows launch  ([B, N_i] = [B, K_i] x [N_i, K_i]^T, forward and dX):  256x128 tiles when every N_i >= K_i (d -> 2d),
#                 else 128x128; slices proportional to K_i such that the launch has ~one workgroup per CU
#                 (config 2: (3, 2) slices -> 256 workgroups; 37.5 -> 30.5 us and 34.2 -> 26.7 us per launch)
#   dW (+ dX) grouped launch: 128x128 tiles on 8 waves of 64x32 (cfg 29; +2.4 % end to end over 4 waves of 64x64, cfg 25),
#                 two LDS buffers (two workgroups per CU); dX slices ~ K_i / 1700
# (more waves per tile measured faster at equal tiles: 256x128 on 16 waves of 64x32 instead of 8 of 64x64: 26.8 -> 24.8 us per
#  forward launch; 128x128 on 8 waves of 64x32 instead of 4 of 64x64: 25.1 -> 22.4 us forward, +2.4 % end to end backward)
BF16_CFG_ROWS_WIDE, BF16_CFG_ROWS, BF16_CFG_DW = 31, 32, 29
BF16_TILE = {23: (256, 128), 24: (128, 128), 25: (128, 128), 29: (128, 128), 30: (128, 128), 31: (256, 128), 32: (128, 128)}
N_CU = 256
PANEL = 16                  # jamie_hip.h: JAMIE_PANEL (TrainEngine reads the library's own value: jamie_panel_width)
PANEL_KEYS = ('h1', 'h2', 'g1', 'g2', 'de2', 'de1', 'da2', 'da1')      # what a BatchNorm launch reads as fp32: GEMM slabs and their sums


def _big_enough(B, shapes):
    return B >= 128 and all(N >= 256 and K >= 256 for (N, K) in shapes)


def plan_bf16_rows(B, shapes):
    """shapes = [(N_i, K_i)] of one forward / dX launch -> (cfg, [splitk_i]); cfg -1 = the library default."""
    if not _big_enough(B, shapes):
        return -1, [choose_splitk(B, N, K) for (N, K) in shapes]
    if TUNING['bf16_rows']:
        parts = TUNING['bf16_rows'].split(';')
        part = parts[0] if (all(N >= K for (N, K) in shapes) or len(parts) == 1) else parts[1]
        cfg, sks = part.split(':')
        sks = [int(v) for v in sks.split(',')]
        return int(cfg), [sks[min(i, len(sks) - 1)] for i in range(len(shapes))]
    cfg = BF16_CFG_ROWS_WIDE if (B >= 256 and all(N >= K for (N, K) in shapes)) else BF16_CFG_ROWS
    bm, bn = BF16_TILE[cfg]
    tk = sum(math.ceil(B / bm) * math.ceil(N / bn) * K for (N, K) in shapes)
    c = N_CU / tk
    return cfg, [int(max(1, min(round(c * K), 8, K // 256))) for (N, K) in shapes]


def plan_bf16_bwd(B, shapes):
    """shapes = [(N_i, K_i)] of the dX problems grouped with their dW problems -> (cfg, [splitk_i of the dX])."""
    if not _big_enough(B, shapes):
        return -1, [choose_splitk(B, N, K) for (N, K) in shapes]
    # dX slices of ~1700 of K (config 2: (1, 1), (2, 1), (1, 1) for the three layers): re-swept at the end of round 2, when a
    # slab costs the BatchNorm launch that sums it more than it saves the GEMM (K / 1000: +5.7 us per step, unsplit K = 4000: +25,
    # profiles/r02_bwd_splitk_sweep.log)
    per = float(TUNING['bwd_k_per_slab'])
    return BF16_CFG_DW, [int(max(1, min(round(K / per), 4, K // 256))) for (N, K) in shapes]


F32_CFG_ROWS = 17           # 128x128x32 tile on 16 waves of 32x32, barrier in mid k-step (gemm_f32.hip; 12: the barrier at the end,
                            # +1.3 % per step; 8 waves of 64x32, cfg 4, is 5-7 % slower)
F32_CFG_DW = 17             # fp32 dW (TN, K = batch) of the large layers when the launches do not carry the fused clip norm (gradient
                            # exchange): the mid-barrier 128x128 tile (round 2's sweep had 64x64, configuration 1; with round 4's loop
                            # config 5's dimensions run 6.86 against 6.97 ms per step, config 2 the same: r04_f32_dp_dw_tile.log)
F32_CFG_DW_FUSED = 17       # ... 128x128x32 on 16 waves when the launch also writes its tiles' sums of squares (fused clip norm)
F32_CFG_SOLO = 19           # 17 with 56 KB of unused dynamic LDS: one workgroup per CU (forward launches beside the optimiser stream)
F32_CFG_X3 = 21             # 256x128x32 on four waves of 128x64, the products as six bf16 MFMAs on three-piece cuts (TUNING['f32_x3']): one
                            # workgroup per CU (144 KB of LDS), ~1.9 us per k-step of twice the work (fp32 pipe, 128x128: 2.05).  20 = the same
                            # on 128x128 tiles (1.0 us per k-step, three LDS stages): 1066 against 993 us per step at config 2
                            # (profiles/r05_ab_f32_bf16x3_256.log) -- per flop the two loops are within 5 % (both run at the matrix pipe's
                            # rate under the clock the chip holds), the larger tile halves the number of tile prologues and store tails
F32_X3_TILE_M = 256
F32_CFG_X3_128 = 20         # the 128 x 128 form (three LDS stages): the planner takes it where it needs fewer K slices


def f32_cfg(kind='rows'):
    """Tile configuration of the large fp32 launches: 'rows' (forward / dX), 'dw', 'dw_fused'."""
    if TUNING['f32_x3']:
        return F32_CFG_X3
    return {'rows': F32_CFG_ROWS, 'dw': F32_CFG_DW, 'dw_fused': F32_CFG_DW_FUSED}[kind]


def _f32_fused_cfg():
    """Tile configuration of the fp32 dW launches that also write their tiles' sums of squares (TUNING['f32_dw_cfg'] overrides)."""
    return int(TUNING['f32_dw_cfg']) if TUNING['f32_dw_cfg'] not in (None, '') else f32_cfg('dw_fused')


def launch_makespan(works, n_cu=N_CU, per_cu=2, solo=0.87):
    """Time (in units of work) a grid of workgroups takes on `n_cu` CUs that hold up to `per_cu` workgroups each.  The
    dispatcher hands out workgroups in grid order to the least occupied CU with a free slot (completions of one instant are
    retired first); the workgroups resident on a CU share its matrix pipe equally, and a workgroup alone on its CU gets `solo`
    of it (nothing fills its barrier bubbles).  `works` = the work of every workgroup in grid order.  The model behind
    plan_f32_rows: it ranks the split-K plans measured at config 5's dimensions in their measured order (14 plans,
    profiles/r03_c5_f32_plan_sweep*.log) and reproduces config 2's (3, 2) / (6, 3)."""
    import heapq
    works = list(works)
    clock, rem, ver = [0.0] * n_cu, [[] for _ in range(n_cu)], [0] * n_cu
    free = [set(range(n_cu))] + [set() for _ in range(per_cu)]     # free[k]: CUs with k resident workgroups
    heap, pos, end = [], 0, 0.0

    def rate(c):
        return solo if len(rem[c]) == 1 else 1.0 / len(rem[c])

    def advance(c, t):
        if rem[c]:
            done = (t - clock[c]) * rate(c)
            rem[c] = [r - done for r in rem[c]]
        clock[c] = t

    def schedule(c):
        ver[c] += 1
        if rem[c]:
            heapq.heappush(heap, (clock[c] + min(rem[c]) / rate(c), c, ver[c]))

    def dispatch(t):
        nonlocal pos
        while pos < len(works):
            k = next((k for k in range(per_cu) if free[k]), None)
            if k is None:
                return
            c = free[k].pop()
            advance(c, t)
            rem[c].append(works[pos])
            pos += 1
            free[k + 1].add(c)
            schedule(c)
    dispatch(0.0)
    while heap:
        t, c, v = heapq.heappop(heap)
        if v != ver[c]:
            continue
        advance(c, t)
        free[len(rem[c])].discard(c)
        rem[c] = [r for r in rem[c] if r > 1e-9]
        free[len(rem[c])].add(c)
        end = t
        schedule(c)
        if not (heap and heap[0][0] <= t + 1e-9):
            dispatch(t)
    return end


def plan_f32_rows(B, shapes):
    """(tile configuration, K slices per problem) of an fp32 forward / dX launch; see _plan_f32_rows."""
    return _plan_f32_rows(B, tuple(tuple(x) for x in shapes), TUNING['f32_rows'], TUNING['f32_rows_cfg'], bool(TUNING['f32_x3']))


@functools.lru_cache(maxsize=256)
def _plan_f32_rows(B, shapes, _knob, _tile=None, _x3=False):
    """fp32 forward / dX launch with shapes = [(N_i, K_i)]: the 128x128x32 tile (two workgroups per CU) with K slices of one
    common length, chosen by a model of the launch: workgroup work = k-steps + 3 (prologue, store tail), launch time =
    launch_makespan of the grid (every problem's slices in problem order) + what the extra slabs cost their consumer
    (written and read once more at ~3 TB/s; a k-step of a CU is ~2 us).  Config 2: d -> 2d (3, 2) slices, exactly two
    workgroups per CU (tools/bench_gemm_f32_tiles.py: 110 us against 119-122 us for the unsplit 64x64 tiles), 2d -> d (6, 3)
    (102 against 107 us); config 5's dimensions: (2, 1) and (4, 2), one and a half rounds of 79-k-step workgroups -- the rule of
    rounds 1-2 (the largest grid of at most two workgroups per CU) left a quarter of the CUs with one long workgroup there:
    8.61 -> 7.60 ms per step, forward launches 94 -> 120 TFLOP/s.  (-1, None): keep the default."""
    if B < 256 or any(N < 512 or K < 512 for (N, K) in shapes):
        return -1, None
    default_cfg = F32_CFG_X3 if _x3 else F32_CFG_ROWS
    if _knob:
        parts = _knob.split(';')
        part = parts[0] if (all(N >= K for (N, K) in shapes) or len(parts) == 1) else parts[1]
        cfg, _, part = part.rpartition(':')                 # ("cfg:s0,s1": another tile configuration)
        sks = [int(v) for v in part.split(',')]
        return (int(cfg) if cfg else default_cfg), [sks[min(i, len(sks) - 1)] for i in range(len(shapes))]
    slab_tbps = 3.0
    # (tile configuration, rows per tile, us per k-step, per-tile overhead in k-steps, workgroups per CU, a lone workgroup's rate)
    if _x3:      # bf16x3: 256 x 128 tiles (half the tile prologues / store tails) or 128 x 128 (fewer K slices fill the chip: fewer
        #          slabs for the consumer) -- whichever the model prices lower for this launch; one workgroup per CU either way
        #          (k-step times under the clock the chip holds in launches of this length: 1.0 us was measured on short ones;
        #          with it the model put config 5's launches on the small tile, measured 5.5 % slower there)
        kinds = [(F32_CFG_X3, F32_X3_TILE_M, 1.9, 4.0, 1, 1.0), (F32_CFG_X3_128, 128, 1.15, 6.0, 1, 1.0)]
    else:
        kinds = [(F32_CFG_ROWS, 128, 2.05, 3.0, 2, 0.87)]
    if _tile not in (None, ''):
        kinds = [k for k in kinds if k[0] == int(_tile)] or [(int(_tile),) + kinds[0][1:]]
    best = None
    for cfg, bm, kstep_us, overhead, per_cu, solo in kinds:
        tiles = [math.ceil(B / bm) * math.ceil(N / 128) for (N, K) in shapes]
        seen = set()
        for kc in range(256, max(K for (_, K) in shapes) + 1, 8):
            sk = tuple(min(8, max(1, math.ceil(K / kc))) for (_, K) in shapes)
            if sk in seen:
                continue
            seen.add(sk)
            works = [K / s / 32.0 + overhead for t, s, (_, K) in zip(tiles, sk, shapes) for _ in range(t * s)]
            slabs = sum((s - 1) * B * N * 8.0 for s, (N, _) in zip(sk, shapes)) / (slab_tbps * 1e6)        # us
            cost = launch_makespan(works, per_cu=per_cu, solo=solo) * kstep_us + slabs
            if best is None or cost < best[0] - 1e-9:
                best = (cost, cfg, list(sk))
    return (best[1], best[2]) if best else (-1, None)


def kl_anneal(epoch, min_epochs, epoch_DNN):
    """jamie.py:630-631."""
    c = (min_epochs / 2) if min_epochs > 0 else (epoch_DNN / 2)
    return float(1 / (1 + np.exp(-5 * (epoch - c) / c)))


class TrainEngine:
    def __init__(self, model, batch_size, lr=1e-3, loss_weights=None, dist_method='euclidean', seed=666,
                 world_size=1, compute_dtype='f32', dx_from_weights=True, skinny_tr=True, grad_bf16=None):
        """compute_dtype 'f32': exact-fp32 MFMA GEMMs (the parity configuration).  'bf16': bf16 MFMA GEMMs with
        fp32 accumulation, fp32 master weights / optimiser / BatchNorm / losses (BASELINE config 2); needs every
        feature count, the latent size and the batch size to be multiples of 8."""
        nv.require_gpu()
        if compute_dtype not in ('f32', 'bf16'):
            raise ValueError("compute_dtype must be 'f32' or 'bf16'")
        self.bf16 = compute_dtype == 'bf16'
        self.compute_dtype = compute_dtype
        # BN kernels emit the bf16 / transposed copies themselves when the strip is register-resident (B <= 1024)
        self.fuse_bf16 = self.bf16 and int(batch_size) <= 1024     # (8 rows per thread in the float4 BN kernels above 512)
        self.m = model
        self.dev = model.device
        self.B = B = int(batch_size)
        # the feature counts the kernels see (model.pdims: rounded up when the model was built with pad_features) and the
        # model's own (rdims: the reconstruction loss is a mean over the REAL features; the padding contributes exact zeros)
        self.dims = list(getattr(model, 'pdims', model.input_dim))
        self.rdims = list(model.input_dim)
        self.M = len(self.dims)
        self.L = L = model.output_dim
        self.p_drop = model.dropout
        if self.bf16 and any(v % 8 for v in list(self.dims) + [L, B]):
            raise ValueError('bf16 compute needs feature counts, latent size and batch size that are multiples of 8 '
                             '(build the model with pad_features=8 for other feature counts)')
        if self.M > 2 and L > 128:      # (the M-modality latent kernels keep a cell's latent row in registers / LDS)
            raise ValueError(f'more than two modalities need output_dim <= 128 (got {L})')
        self.cosine = dist_method == 'cosine'
        if dist_method not in ('euclidean', 'cosine'):
            raise ValueError("dist_method must be 'euclidean' or 'cosine' (jamie.py:483-502)")
        self.loss_weights = [1., 1., 1., 1.] if loss_weights is None else [float(w) for w in loss_weights]
        assert len(self.loss_weights) == 4, f'There are 4 losses and {len(self.loss_weights)} weights'
        f32 = dict(device=self.dev, dtype=torch.float32)
        n = model.layout.total
        # the streams clip + Adam walks in lock step (p, m, v, g, the bf16 weight copy) start 4 KB apart modulo the allocator's
        # 2 MB alignment: 211 instead of 215-220 us stand-alone (tools/bench_adam_offsets.py, profiles/r02_adam_buffer_placement.log)
        self._stagger = 4096 if TUNING['stagger'] else 0
        self.exp_avg = self._flat_alloc(n, torch.float32, 1)
        self.exp_avg_sq = self._flat_alloc(n, torch.float32, 2)
        self.grad = self._flat_alloc(n, torch.float32, 3)
        self.g = model.layout.views(self.grad)
        self.n_norm = nv.optim_blocks(n)
        self.norm_partials = torch.zeros(self.n_norm, **f32)
        # rng/state: [seed, step, 0, 0]
        self.state = torch.tensor([seed, 0, 0, 0], dtype=torch.int64, device=self.dev)
        if self.bf16 and TUNING['adam_rotate']:        # (A/B: measured +5 us per step, off)
            self.set_adam_start(True)
        hyper = torch.zeros(16)
        hyper[H_REC], hyper[H_ALIGN], hyper[H_F] = (self.loss_weights[1], self.loss_weights[2] * ALIGN_WEIGHT,
                                                     self.loss_weights[3])
        hyper[H_LR], hyper[H_B1], hyper[H_B2], hyper[H_EPS] = lr, 0.9, 0.999, 1e-8
        hyper[H_MAXNORM], hyper[H_GSCALE] = 1.0, 1.0 / world_size
        self._hyper_host = hyper
        self.hyper = hyper.to(self.dev)
        self.set_kl_anneal(1.0)
        self.losses = torch.zeros(8, **f32)
        self.reset_best()
        self.lat_partials = torch.zeros(20 * nv.load().jamie_max_partials(), **f32)
        self.lat_ticket = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.lat_colpart = torch.zeros(int(nv.load().jamie_latent_m_colpart_size(B, L)) if L <= 128 else 1, **f32)
        # ---- per-modality workspace ----
        self.ws = []
        # the heads / dcomb slab counts must agree between the modalities (one latent launch reads both)
        sk_head = min(choose_splitk(B, 2 * L, d) for d in self.dims)
        sk_dcomb = min(choose_splitk(B, L, d) for d in self.dims)
        if TUNING['sk_skinny']:
            sk_head = sk_dcomb = int(TUNING['sk_skinny'])
        # bf16: tile configuration of every large launch + per-modality slab counts (plan_bf16_*)
        self.gcfg, plan_sk = {}, {}
        if self.bf16:
            grouped = 2 * self.M <= nv.MAX_GEMM_GROUP
            for key, shp, bwd in (('enc0', [(2 * d, d) for d in self.dims], False),
                                  ('enc1', [(d, 2 * d) for d in self.dims], False),
                                  ('dec1', [(2 * d, d) for d in self.dims], False),
                                  ('dec2', [(d, 2 * d) for d in self.dims], False),
                                  ('d_e2', [(2 * d, d) for d in self.dims], True),
                                  ('d_e1', [(d, 2 * d) for d in self.dims], True),
                                  ('d_a1', [(2 * d, d) for d in self.dims], True)):
                self.gcfg[key], plan_sk[key] = (plan_bf16_bwd if (bwd and grouped) else plan_bf16_rows)(B, shp)
            self.gcfg['dw'] = BF16_CFG_DW if _big_enough(B, [(d, d) for d in self.dims]) else -1
            # the skinny head / latent backward launches go through the 128 x 128 k-row-major kernel as well: the same speed as
            # the 64 x 64 kernel on transposed copies (profiles/r02_ab_skinny_tr.log), and with no transposed weight copy left
            # the next batch's gather can ride in the optimiser launch (make_plan)
            self.skinny_tr = bool(skinny_tr) and self.gcfg['dw'] == BF16_CFG_DW and L % 8 == 0 and 2 * self.M <= nv.MAX_GEMM_GROUP
            if self.skinny_tr:
                self.gcfg['d_comb'] = self.gcfg['d_a2'] = BF16_CFG_DW
        # dW = dy^T a: the 128 x 128 large-tile kernel reads dy [B, out] and a [B, in] as the layers wrote them (a_tr + b_tr),
        # so only operands of the launches that do not take that kernel (skinny head / latent layers, small models) still
        # need a transposed [features, B] copy
        # fp32: tile + slices of the forward launches and of the dX launches with K = 2d (plan_f32_rows); the x_hat product is
        # split too and its MSE / gradient come from jamie_mse_cast (a fused epilogue needs an unsplit K: 126 us)
        self.fcfg = {}
        if not self.bf16:
            dx_nn = [] if not TUNING['f32_dx_plan'] else \
                [('d_e2', [(2 * d, d) for d in self.dims]), ('d_a1', [(2 * d, d) for d in self.dims])]
            for key, shp in [('enc0', [(2 * d, d) for d in self.dims]), ('enc1', [(d, 2 * d) for d in self.dims]),
                             ('dec1', [(2 * d, d) for d in self.dims]), ('dec2', [(d, 2 * d) for d in self.dims]),
                             ('d_e1', [(d, 2 * d) for d in self.dims])] + dx_nn:
                cfg, sks = plan_f32_rows(B, shp)
                if cfg >= 0:
                    self.fcfg[key] = cfg
                    plan_sk[key] = sks
        # ---- panel layout of the BatchNorm launches' fp32 inputs (bf16 mode, float4 BatchNorm kernels, large-tile GEMMs) ----
        # A BatchNorm workgroup owns a strip of 16 / 32 columns for all B rows: of a row-major [B, N] slab that is B segments of 64 /
        # 128 bytes, 4 N bytes apart, per slab.  In PANELS of 16 columns ([N / 16][B][16]) the same strip is one contiguous block of
        # B x 64 bytes per slab: the launches' load phases -- all they are bound by -- shorten by 1.5-2 us each
        # (profiles/r05_ab_bn_panel*.log).  The producers write that layout themselves (GEMM epilogue: c_panel; fused latent launches:
        # g1_panel / da2_panel), BatchNorm forward writes the summed pre-activation back in it and BatchNorm backward reads it; the
        # activations / gradients that leave a BatchNorm launch are bf16 row-major as before.  Which layers take it is decided per
        # step (_set_panels): a layer's buffers are in panels only when BOTH its forward and its backward producer can write them.
        self.PANEL = int(nv.load().jamie_panel_width())
        self.panel = bool(self.bf16 and self.fuse_bf16 and TUNING['bn_panel'] and all(d % 4 == 0 for d in self.dims)
                          and all(self.gcfg.get(k, -1) in BF16_TILE for k in ('enc0', 'enc1', 'dec1', 'd_e2', 'd_e1', 'd_a1')))
        self._pan = {'bn0': False, 'bn1': False, 'bn2': False, 'bn3': False}
        self.need_T = set()
        if self.bf16:
            for dy_key, a_key, lin in (('dxhat', 'e2', 'dec2'), ('de2', 'e1', 'dec1'), ('de1', 'comb', 'dec0'),
                                       ('dml', 'a2', 'head'), ('da2', 'a1', 'enc1'), ('da1', 'x', 'enc0')):
                if not self._dw_tr(lin):
                    self.need_T.update((dy_key, a_key))
        for i, d in enumerate(self.dims):
            w = {}
            sk = {'enc0': choose_splitk(B, 2 * d, d), 'enc1': choose_splitk(B, d, 2 * d),
                  'head': sk_head, 'dec0': 1, 'dec1': choose_splitk(B, 2 * d, d),
                  'd_e2': choose_splitk(B, 2 * d, d),     # dxhat  D3   : [B,d]  x [d,2d]
                  'd_e1': choose_splitk(B, d, 2 * d),     # dg2p   D2   : [B,2d] x [2d,d]
                  'd_comb': sk_dcomb,                     # dg1p   D1   : [B,d]  x [d,L]
                  'd_a2': 1,                              # dml    Wh   : [B,2L] x [2L,d]
                  'd_a1': choose_splitk(B, 2 * d, d)}     # dh2p   W2   : [B,d]  x [d,2d]
            for key, v in plan_sk.items():
                sk[key] = v[i]
            w['sk'] = sk
            w['x'] = torch.empty(B, d, **f32)
            w['h1'] = self._slabs(sk['enc0'], 2 * d); w['a1'] = torch.empty(B, 2 * d, **f32)
            w['h2'] = self._slabs(sk['enc1'], d); w['a2'] = torch.empty(B, d, **f32)
            w['ml'] = torch.empty(sk['head'], B, 2 * L, **f32)
            for k in ('mu', 'lv', 'z', 'eps', 'comb', 'cz', 'H', 'ch'):
                w[k] = torch.empty(B, L, **f32)
            w['g1'] = self._slabs(1, d); w['e1'] = torch.empty(B, d, **f32)
            w['g2'] = self._slabs(sk['dec1'], 2 * d); w['e2'] = torch.empty(B, 2 * d, **f32)
            w['dxhat'] = torch.empty(B, d, **f32)
            if self.gcfg.get('dec2', -1) >= 0 or self.fcfg.get('dec2', -1) >= 0:   # split-K x_hat slabs, MSE in jamie_mse_cast
                w['xh'] = torch.empty(sk['dec2'], B, d, **f32)
                # ... which also leaves the column sums of d x_hat per 64-row tile: the decoder's output-bias gradient is then a sum
                # of B / 64 rows (and in bf16 mode nothing reads an fp32 d x_hat any more: it is not written)
                w['dxhat_cp'] = torch.zeros((B + 63) // 64, d, **f32)
            w['de2'] = self._slabs(sk['d_e2'], 2 * d)
            w['de1'] = self._slabs(sk['d_e1'], d)
            w['dcomb'] = torch.empty(sk['d_comb'] + 1, B, L, **f32)   # +1 slab: external d(combined) (autograd seam)
            w['xhat'] = None                                           # allocated on first forward_only()
            w['dml'] = torch.empty(B, 2 * L, **f32)
            w['da2'] = self._slabs(1, d)
            w['da1'] = self._slabs(sk['d_a1'], 2 * d)
            for k, nn in (('bn0', 2 * d), ('bn1', d), ('bn2', d), ('bn3', 2 * d)):
                w[k + '.mean'] = torch.empty(nn, **f32); w[k + '.invstd'] = torch.empty(nn, **f32)
            w['idx'] = torch.zeros(B, dtype=torch.int32, device=self.dev)
            if self.M > 2 and i > 0:
                w['comb'] = self.ws[0]['comb']          # M > 2 (identity corr): one combined embedding for all
            if self.bf16:
                bf = dict(device=self.dev, dtype=torch.bfloat16)
                for k, nf in (('x', d), ('a1', 2 * d), ('a2', d), ('comb', L), ('e1', d), ('e2', 2 * d), ('dxhat', d),
                              ('de2', 2 * d), ('de1', d), ('dml', 2 * L), ('da2', d), ('da1', 2 * d)):
                    w[k + '_bf'] = torch.empty(B, nf, **bf)
                    w[k + '_T'] = torch.empty(nf, B, **bf)
            self.ws.append(w)
        # dec2 (MSE epilogue) uses the 64x128 tile config unless N <= 64
        self.wT = {}
        if self.bf16:
            # bf16 copies of the weights: same flat layout, plus K-contiguous transposes for the dX products
            self.wbf_flat = self._flat_alloc(model.layout.total, torch.bfloat16, 4)
            self.wbf = model.layout.views(self.wbf_flat)
            # dX = dy W: the large-tile kernel reads W [out, in] as stored (b_tr: [k][n] LDS image, transposed fragment
            # reads), so only the layers whose backward launch does not take that kernel keep a transposed copy
            self.wT = {}
            for i, d in enumerate(self.dims):
                for lin, key in (('enc1', 'd_a1'), ('head', 'd_a2'), ('dec0', 'd_comb'), ('dec1', 'd_e1'), ('dec2', 'd_e2')):
                    if dx_from_weights and self.gcfg.get(key, -1) in BF16_TILE:
                        continue
                    nout, nin = model.p[f'm{i}.{lin}.W'].shape
                    self.wT[f'm{i}.{lin}'] = torch.empty(nin, nout, device=self.dev, dtype=torch.bfloat16)
            self.refresh_weights_bf16()
        # ---- gradient norm without a second pass over the weight gradients (single GPU, bf16 large-tile dW launches):
        # every dW tile's epilogue writes its sum of squares into `norm_partials`; a small kernel adds the ranges the
        # GEMMs do not produce (biases, BatchNorm affine parameters, sigma, the skinny head / latent matrices)
        self.fused_norm, self._fuse_now, self._norm_ready = False, False, False
        self._lat_deferred, self._ranges_done = None, False
        big = {'dec2': 'd_e2', 'dec1': 'd_e1', 'enc1': 'd_a1', 'enc0': 'dw'}
        # (the skinny head / latent layers' dW take the large tile too: their sums of squares come from the epilogue as well -- at
        #  L = 64 those two matrices alone are more range chunks than the range-norm launch carries)
        skinny_lins = ('head', 'dec0') if (self.bf16 and getattr(self, 'skinny_tr', False)) else ()
        # (fp32 mode: the TN dW launches of the large layers take the 128 x 128 tile then -- the same speed as the 64 x 64 one
        #  inside the step, profiles/r02_f32_dw_tile_sweep.log, and a quarter of the partial sums)
        f32_fused = (not self.bf16 and world_size == 1 and B >= 256 and TUNING['f32_fused_norm']
                     and all(min(model.p[f'm{i}.{lin}.W'].shape) >= 512 for lin in big for i in range(self.M)))
        self._f32_dw_fused = f32_fused
        if f32_fused or (self.bf16 and world_size == 1 and all(self.gcfg.get(k, -1) in BF16_TILE for k in big.values())):
            bm_d, bn_d = nv.gemm_tile(nv.TN, 1 << 20, 1 << 20, B, _f32_fused_cfg()) if f32_fused else BF16_TILE[self.gcfg['dw']]
            self.dw_partial, off, covered = {}, 0, []
            for lin in tuple(big) + skinny_lins:
                for i in range(self.M):
                    o, shp = model.layout.entries[f'm{i}.{lin}.W']
                    t = math.ceil(shp[0] / bm_d) * math.ceil(shp[1] / bn_d)
                    self.dw_partial[f'm{i}.{lin}'] = (off, t)
                    off += t
                    covered.append((o, o + shp[0] * shp[1]))
            covered.sort()
            rest, pos = [], 0
            for lo, hi in covered:
                if lo > pos:
                    rest.append((pos, lo - pos))
                pos = hi
            if n > pos:
                rest.append((pos, n - pos))
            self.sq_ranges = nv.SqRanges(rest)
            # the same without d sigma and the head-bias gradients: a step with the fused latent kernels defers their
            # finalisation to the range-norm launch's extra workgroup, which also adds their squares (optimizer_step)
            cut = [model.layout.entries['sigma']] + [model.layout.entries[f'm{i}.head.b'] for i in range(self.M)]
            cut = sorted((o, o + (int(np.prod(shp)) + 3) // 4 * 4) for o, shp in cut)     # (whole 4-aligned slots: the padding is zero)
            rest2 = []
            for lo, ln in rest:
                hi = lo + ln
                for c0, c1 in cut:
                    if c0 >= hi or c1 <= lo:
                        continue
                    if c0 > lo:
                        rest2.append((lo, c0 - lo))
                    lo = max(lo, c1)
                if hi > lo:
                    rest2.append((lo, hi - lo))
            self.sq_ranges_nofin = nv.SqRanges(rest2)
            if off + self.sq_ranges.blocks + 2 <= nv.load().jamie_max_norm_partials() and self.sq_ranges.blocks <= 128:
                self.fused_norm = True
                self.n_dw_partials = off
                self.norm_partials = torch.zeros(max(off + self.sq_ranges.blocks, self.n_norm) + 2 + sum((d + 63) // 64 for d in self.dims), **f32)
        # ---- weight gradients in bf16 (bf16 compute mode, one GPU, fused norm): the large dW launches round their fp32
        # accumulators once on the way out into `grad16` (same flat layout as `grad`), the small ranges (biases, BatchNorm
        # affine parameters, sigma, the skinny matrices) are copied there by the range-norm kernel, and clip + Adam reads
        # 2 instead of 4 bytes of gradient per parameter (26 instead of 28 bytes per parameter in all; the dW launches
        # store 80 instead of 161 MB per step).  The norm partials are sums of squares of the fp32 values.  Not with
        # accumulating gradients (batch_step=False): set_grad_bf16(False) before the first backward pass of such a run.
        self._f32_dw_fused = self._f32_dw_fused and self.fused_norm
        # Default ON in bf16 compute mode: this is what torch.autocast(bfloat16) does to the same step -- the weight gradient of an
        # autocast Linear is the OUTPUT of a bf16 matmul (rounded to bf16) before it is accumulated into the fp32 .grad -- and
        # it takes 23 us off the step (profiles/r02_ab_grad_bf16.log: clip + Adam reads 2 bytes less per parameter, the dW
        # launches store half).  grad_bf16=False keeps fp32 weight gradients.
        self.grad_bf16 = (True if grad_bf16 is None else bool(grad_bf16)) and self.fused_norm and self.bf16
        self._g16_now = self._g16_last = self._g16_pending = False
        self._direct_now, self._direct = False, None       # (data parallel, bf16 messages: see _direct_setup)
        if self.grad_bf16:
            self.grad16 = self._flat_alloc(n, torch.bfloat16, 3)
            self.g16 = model.layout.views(self.grad16)
        bm_t, bn_t = (nv.gemm_bf16_tile(B, max(self.dims)) if self.bf16 else
                      nv.gemm_tile(nv.NT, B, max(self.dims), 2 * max(self.dims)))   # tile of the grouped launch
        self.rec_tiles = [math.ceil(B / bm_t) * math.ceil(d / bn_t) for d in self.dims]
        self.rec_partials = torch.zeros(sum(self.rec_tiles), **f32)
        self.rsum = torch.empty(B, **f32); self.qsum = torch.empty(B, **f32)
        self.fc1 = torch.empty(B, L, **f32); self.fte = torch.empty(B, L, **f32)
        self.corr = torch.empty(B, B, **f32)
        self.accumulate = False          # True: gradients add to the buffer (batch_step=False, jamie.py:736-749)
        self._dsig_tmp = torch.zeros(self.M, **f32)
        self._timing = None
        self._timing_every, self._timing_step = 1, 0
        self.pipeline = False            # enable_pipeline(): optimiser on its own stream, overlapped with the next forward
        self._zs = None                  # enable_sharded_optimizer(): packed shard state of a data-parallel run
        self._dw_wait = None             # fp32 backward pass: the large layers' dW products waiting for _flush_dw
        self._dw_small = []              # ... and the skinny layers' that go with them
        self._dxhat_cs = 'dxhat'         # which buffer the decoder's output-bias gradient is summed from (_forward)
        self.side_transposes, self._wT_pending, self._wT_stale = False, False, False

    # ---- pipelined optimiser: clip + Adam of step t on a second HIP stream, under the forward pass of step t+1 ----
    # Adam is HBM-bound (28 B/parameter) and the forward GEMMs are bound by the L2 -> LDS fabric and the matrix pipe, so
    # the two share the chip well.  The flat buffers hold the small tensors first and then the large weight matrices in
    # forward order, so the update is issued as four contiguous launches (rep + enc0 | enc1 | dec1 | dec2, about a quarter
    # of the parameters each); the
    # forward pass of the next step waits, layer by layer, for the event of the group it is about to read.  Every
    # element sees exactly the update of the one-launch form (bit-identical, tests/test_hip_step.py).
    PIPE_GROUPS = (('rep', 'enc0'), ('enc1',), ('dec1',), ('dec2',))
    PIPE_LINS = (('enc0', 'head', 'dec0'), ('enc1',), ('dec1',), ('dec2',))       # (the layers whose weights a group holds)
    PIPE_WAIT = {'enc0': 0, 'enc1': 1, 'dec1': 2, 'dec2': 3}

    def enable_side_transposes(self):
        """bf16 mode: make the transposed weight copies on a side stream, overlapped with the next forward pass."""
        if not self.bf16:
            return
        self.side_transposes = True
        self.side_stream = torch.cuda.Stream(device=self.dev)
        self._ev_adam, self._ev_wT = torch.cuda.Event(), torch.cuda.Event()

    def _wait_wT(self):
        if not self.side_transposes:
            return

        def fn():
            if self._wT_pending:
                nv.current_stream().wait_event(self._ev_wT)
        self._both(fn)

    def set_adam_start(self, on=True):
        """clip + Adam walks the flat buffers from the start of the SECOND layer and wraps around (state[2], float4 units): the
        first layer's parameters are then updated last and its bf16 weights are the freshest lines in the caches when the next
        step's first product starts (the one forward launch no BatchNorm launch can prefetch for).  Every element sees the
        same update; only the order of the streams changes."""
        lo = self.m.layout.regions['enc1'][0] if on else 0
        self.state[2] = (lo // 4) if lo % 4 == 0 else 0

    def enable_pipeline(self, priority=0):
        """Run clip + Adam (and the bf16 weight transposes) on a side stream; `flush()` before anything other than
        the next training step reads the parameters."""
        self.pipeline = True
        self.opt_stream = torch.cuda.Stream(device=self.dev, priority=priority)
        self._ev_grads = torch.cuda.Event()
        self._ev_params = [torch.cuda.Event() for _ in self.PIPE_GROUPS]
        self._opt_pending = False

    def _both(self, fn):
        """Run `fn` now and, while a plan is being recorded, make it a plan entry as well."""
        nv.record_callable(fn)
        fn()

    def _wait_params(self, lin):
        if self._zs is not None and lin in self._zs['names']:       # sharded optimiser: this layer's all-gathered weights
            ex = self._zs['ex']
            self._both(lambda: ex.wait_gather(lin))
            return
        g = self.PIPE_WAIT.get(lin) if self.pipeline else None
        if g is None:
            return
        ev = self._ev_params[g]

        def fn():
            if self._opt_pending:
                nv.current_stream().wait_event(ev)
        self._both(fn)

    def flush(self, collective=False):
        """Make torch's current stream wait for the optimiser stream (parameters, Adam moments, bf16 copies).  Local by
        default: with a sharded optimiser whose packed pieces are newer than the replicated buffers it RAISES unless
        `collective=True` -- gathering them takes all-gathers that every rank must join (gather_sharded_state), so a
        rank-local flush (a callback on rank 0, an exception handler) must never start them by itself."""
        if self._zs is not None and self._zs.get('stale'):
            if not collective:
                raise nv.JamieHipError('flush(): the sharded optimiser holds state newer than the replicated buffers; call '
                                       'flush(collective=True) (or gather_sharded_state()) on EVERY rank first')
            self.gather_sharded_state()
        if self.pipeline and self._opt_pending:
            for ev in self._ev_params:
                torch.cuda.current_stream().wait_event(ev)
        if self.side_transposes and self._wT_pending:
            torch.cuda.current_stream().wait_event(self._ev_wT)
        if self.bf16 and self._wT_stale:                 # the skinny weights' transposed copies ride on the next batch launch
            self.refresh_weights_bf16(transposes_only=True)
            self._wT_stale = False

    # ---- data parallel with a sharded optimiser (distributed.ShardedGradExchange) ----
    def enable_sharded_optimizer(self, ex):
        """Cut each large weight region into `ex.world` pieces; this rank keeps fp32 master weights, Adam moments, the reduced
        gradient and (bf16 mode) the bf16 weight copy of piece `ex.rank` of every region in PACKED buffers, so that clip + Adam
        is one launch over 1 / world of the parameters.  The small region `rep` stays replicated."""
        lay, n, r = self.m.layout, int(ex.world), int(ex.rank)
        if n < 2 and not getattr(ex, 'single', False):
            raise ValueError('a sharded optimiser needs more than one rank')
        if self.pipeline or self.accumulate or self.side_transposes or any(k.split('.')[1] in lay.BIG_LAYERS for k in self.wT):
            raise ValueError('sharded optimiser: not with the pipelined optimiser, accumulating gradients or transposed copies '
                             'of the large weight matrices (bf16 mode with layers under 256 features)')
        spans, off = [], 0
        for name in lay.BIG_LAYERS:
            lo, hi = lay.regions[name]
            if (hi - lo) % (8 * n):
                raise ValueError(f'region {name} ({hi - lo} elements) does not split into {n} 16-byte aligned pieces')
            s = (hi - lo) // n
            spans.append((name, lo, hi, s, off))
            off += s
        bf_msgs = ex.comm_dtype is not None
        zs = {'ex': ex, 'n': n, 'r': r, 'spans': spans, 'names': tuple(lay.BIG_LAYERS), 'S': off, 'bf_msgs': bf_msgs,
              'p': self._flat_alloc(off, torch.float32, 0), 'm': self._flat_alloc(off, torch.float32, 1),
              'v': self._flat_alloc(off, torch.float32, 2),
              'g': self._flat_alloc(off, torch.bfloat16 if bf_msgs else torch.float32, 3),
              'w16': self._flat_alloc(off, torch.bfloat16, 4) if self.bf16 else None}
        for name, lo, hi, s, o in spans:
            a = lo + r * s
            zs['p'][o:o + s].copy_(self.m.flat[a:a + s])
            zs['m'][o:o + s].copy_(self.exp_avg[a:a + s])
            zs['v'][o:o + s].copy_(self.exp_avg_sq[a:a + s])
            if self.bf16:
                zs['w16'][o:o + s].copy_(self.wbf_flat[a:a + s])
        rlo, rhi = lay.regions['rep']
        zs['n_a'] = nv.optim_blocks(off)
        zs['partials'] = torch.zeros(zs['n_a'] + nv.optim_blocks(rhi - rlo), device=self.dev, dtype=torch.float32)
        if zs['partials'].numel() > nv.load().jamie_max_norm_partials():
            raise ValueError('sharded optimiser: too many norm partials')
        ex.set_shards([(lo, hi, zs['g'][o:o + s]) for _, lo, hi, s, o in spans])
        self._zs = zs

    def _sharded_step(self, sample=None, casts=None):
        """Norm of the reduced gradient from the pieces + `rep`, clip + Adam over the packed shard and over `rep`, then the
        all-gathers of the updated weights (waited for layer by layer in the next forward pass: _wait_params)."""
        zs = self._zs
        ex, na, part = zs['ex'], zs['n_a'], zs['partials']
        rlo, rhi = self.m.layout.regions['rep']
        g_rep = (ex.comm if zs['bf_msgs'] else self.grad)[rlo:rhi]
        nv.grad_sqnorm(zs['g'], part[:na], self.state)                 # (+ the step counter)
        nv.grad_sqnorm(g_rep, part[na:], None)
        self._both(lambda: ex.sum_partials(part[:na]))
        self._norm_ready = self._g16_pending = self._ranges_done = False
        self._lat_deferred = None
        self._launch('adam', lambda: nv.clip_adam(zs['p'], zs['g'], zs['m'], zs['v'], part, self.hyper, self.state,
                                                   zs['w16'], sample, casts))
        nv.clip_adam(self.m.flat[rlo:rhi], g_rep, self.exp_avg[rlo:rhi], self.exp_avg_sq[rlo:rhi], part, self.hyper, self.state,
                     self.wbf_flat[rlo:rhi] if self.bf16 else None)
        full = self.wbf_flat if self.bf16 else self.m.flat
        piece = zs['w16'] if self.bf16 else zs['p']

        def gathers():
            for name, lo, hi, s, o in zs['spans']:                    # forward order
                ex.gather(name, full[lo:hi], piece[o:o + s])
            zs['stale'] = True               # (set here: a replayed plan runs this callable, not the Python around it)
        self._both(gathers)
        if self.bf16:
            self._wT_stale = True

    def gather_sharded_state(self):
        """Make the replicated flat buffers current again: wait for the weight all-gathers and all-gather what only the owners
        hold (the fp32 master weights of the large regions in bf16 mode, the Adam moments)."""
        zs = self._zs
        if zs is None or not zs.get('stale'):
            return
        ex = zs['ex']
        ex.wait_all_gathers()
        for name, lo, hi, s, o in zs['spans']:
            pairs = [(self.exp_avg, zs['m']), (self.exp_avg_sq, zs['v'])] + ([(self.m.flat, zs['p'])] if self.bf16 else [])
            for full, piece in pairs:
                ex.gather(name, full[lo:hi], piece[o:o + s])
                ex.wait_gather(name)
        zs['stale'] = False

    # ---- host-side knobs (all written into device scalars so the launch sequence is capturable) ----
    def set_kl_anneal(self, anneal):
        self._hyper_host[H_KL] = self.loss_weights[0] * KL_WEIGHT * anneal
        self.hyper[H_KL:H_KL + 1].copy_(self._hyper_host[H_KL:H_KL + 1], non_blocking=True)

    def reset_best(self):
        self.losses[5] = float('inf')

    # ---- per-kernel timing with HIP events on the launch stream (bench.py's roofline leg) ----
    def enable_kernel_timing(self, *labels, every=1):
        """Bracket the launches tagged `labels` with HIP events on every `every`-th step (creating and recording
        ~10 events per step costs the host 3-10 % on a loaded box, so benchmarks sample)."""
        self._timing = {label: [] for label in labels}
        self._timing_every = max(1, int(every))
        self._timing_step = 0

    def _ev(self, label, which):
        """Record a HIP event on the launch stream (works both eagerly and as a step of a replayed plan)."""
        def rec():
            if self._timing is None or label not in self._timing or self._timing_step % self._timing_every:
                return
            e = torch.cuda.Event(enable_timing=True)
            e.record(nv.current_stream())
            lst = self._timing[label]
            if which == 0:
                lst.append([e, None])
            elif lst and lst[-1][1] is None:
                lst[-1][1] = e
        if not nv.record_callable(rec):
            rec()

    def _launch(self, label, fn):
        labels = label if isinstance(label, tuple) else (label,)
        for lb in labels:
            self._ev(lb, 0)
        fn()
        for lb in labels:
            self._ev(lb, 1)

    def kernel_timing_ms(self, label, stat='mean'):
        if self._timing is None or not self._timing.get(label):
            return None
        torch.cuda.synchronize()
        t = np.array([a.elapsed_time(b) for a, b in self._timing[label] if b is not None])
        if stat == 'all':
            return {'mean': float(t.mean()), 'median': float(np.median(t)), 'p90': float(np.percentile(t, 90)),
                    'max': float(t.max()), 'n': int(t.size)}
        return float(t.mean())

    def _flat_alloc(self, n, dtype, slot):
        """A zeroed flat buffer of `n` elements that starts `slot * self._stagger` bytes into its allocation."""
        es = 4 if dtype == torch.float32 else 2
        off = slot * self._stagger // es
        return torch.zeros(n + off, device=self.dev, dtype=dtype)[off:]

    def _slabs(self, S, N):
        """[S, B, N] fp32 slab buffer of a BatchNorm input.  With the panel layout available the slabs are ceil(N / 16) * 16 * B
        floats apart (room for N rounded up to whole panels); the tensor is the row-major [S, B, N] view of that storage, which is
        what the buffer holds whenever its layer does NOT take the panel layout in a step (rows(): either layout as [S, B, N])."""
        if not self.panel:
            return torch.empty(S, self.B, N, device=self.dev, dtype=torch.float32)
        npad = (N + self.PANEL - 1) // self.PANEL * self.PANEL
        flat = torch.zeros(S * self.B * npad, device=self.dev, dtype=torch.float32)
        return torch.as_strided(flat, (S, self.B, N), (self.B * npad, N, 1))

    BN_OF = {'h1': 'bn0', 'da1': 'bn0', 'h2': 'bn1', 'da2': 'bn1', 'g1': 'bn2', 'de1': 'bn2', 'g2': 'bn3', 'de2': 'bn3'}

    def _paneled(self, key, cfg=None):
        """Is workspace `key` in the panel layout in the current step?  `cfg`: the tile configuration of the GEMM launch about to
        write it -- only the large-tile kernel can (an inconsistency here would hand BatchNorm a layout it does not expect)."""
        on = bool(self.panel and self._pan.get(self.BN_OF.get(key), False))
        if on and cfg is not None and cfg not in BF16_TILE:
            raise nv.JamieHipError(f'internal: {key} is to be written in panels by tile configuration {cfg}')
        return on

    def rows(self, i, key):
        """Workspace `key` of modality `i` as a row-major [S, B, N] tensor, whichever layout the last step kept it in (tests /
        diagnostics: a copy when the buffer is in panels)."""
        t = self.ws[i][key]
        if not self._paneled(key):
            return t
        S, B, N = t.shape
        PANEL = self.PANEL
        npad = (N + PANEL - 1) // PANEL * PANEL
        flat = torch.as_strided(t, (S, npad // PANEL, B, PANEL), (B * npad, B * PANEL, PANEL, 1))
        return flat.permute(0, 2, 1, 3).reshape(S, B, npad)[:, :, :N]

    def _set_panels(self, fused_latent):
        """Which BatchNorm layers keep their fp32 inputs in panels this step: bn0 / bn3 always (both producers are large-tile GEMM
        launches); bn2 when decoder layer 0 comes out of the fused latent launch; bn1 when the heads' input gradient comes out of
        the fused latent backward launch or a large-tile GEMM."""
        on = self.panel
        fused_tail = fused_latent and self._fuse_da2()
        self._pan = {'bn0': on, 'bn3': on, 'bn2': on and fused_latent,
                     'bn1': on and (fused_tail or self.gcfg.get('d_a2', -1) in BF16_TILE)}

    # ---- bf16 compute mode: bf16 / bf16-transposed copies of GEMM operands ----
    def refresh_weights_bf16(self, transposes_only=False, lins=('enc0', 'enc1', 'head', 'dec0', 'dec1', 'dec2')):
        """bf16 copy of every weight matrix (written by the Adam kernel itself during training) and the
        K-contiguous transposed copies the dX products read."""
        probs = []
        for i, d in enumerate(self.dims):
            for lin in lins:
                W = self.m.p[f'm{i}.{lin}.W']
                wt = self.wT.get(f'm{i}.{lin}')
                if transposes_only and wt is None:
                    continue
                if transposes_only:      # after Adam: transpose the bf16 copy it wrote (half the bytes of the fp32 master)
                    probs.append(nv.cast_problem(self.wbf[f'm{i}.{lin}.W'], None, wt))
                else:
                    probs.append(nv.cast_problem(W, self.wbf[f'm{i}.{lin}.W'], wt))
        if probs:
            nv.cast_transpose(probs)

    def _wT_problems(self):
        """cast_transpose problems of the transposed bf16 weight copies that exist (the skinny layers' only)."""
        return [nv.cast_problem(self.wbf[k + '.W'], None, wt) for k, wt in self.wT.items()]

    def _cast(self, key):
        """fp32 activation / gradient `key` ([B, n] or slab 0 of [S, B, n]) -> bf16 [B, n] and bf16 [n, B]."""
        if not self.bf16 or (self.fuse_bf16 and key in ('a1', 'a2', 'e1', 'e2', 'de2', 'de1', 'da2', 'da1')) \
                or (self.M == 2 and key in ('comb', 'dml')):
            return
        probs = []
        for w in self.ws:
            src = w[key]
            src2 = src[0] if src.dim() == 3 else src
            probs.append(nv.cast_problem(src2, w[key + '_bf'], w[key + '_T'] if key in self.need_T else None))
        nv.cast_transpose(probs)

    # ---- pieces ----
    def _mask(self, noise, kind, i, j):
        if noise is None or self.p_drop == 0:
            return None
        m = noise[kind][i][j]
        width = self.dims[i] * (2 if (kind == 'enc_masks') == (j == 0) else 1)
        if m is not None and m.shape[1] != width:        # explicit masks (parity tests) of a padded model
            m = torch.nn.functional.pad(m, (0, width - m.shape[1])).contiguous()
            noise[kind][i][j] = m
        return m

    def _prefetch(self, *items):
        """Ranges for the BatchNorm launches' prefetch rider (jamie_bn_act_fwd_pf / _bwd_pf): what the NEXT launches stream from
        HBM-cold memory, read into the Infinity Cache by 64 extra workgroups of a launch that has bandwidth to spare.
        Items: 'W:<layer>' = that layer's weights, both modalities (one contiguous range of the bf16 copy); '<key>' = a workspace
        tensor of every modality (saved activations the dW products read, the pre-activations the next BatchNorm backward reads).
        Measured (one box, interleaved, profiles/r03_ab_prefetch.log): bf16 622.2 -> 615.9 us per step with the weights alone (the
        forward launches 27.6 -> 24.6 us each; 64 rider workgroups: 16 / 32 stretch the BatchNorm launch, 695 / 641 us; 128 / 256
        and more loads in flight: no better); fp32 1512 -> 1551 us (those products are bound by the matrix pipe, not by their first
        touch of the weights, and the riders delay the BatchNorm launch): bf16 mode only.  The saved activations on top of the
        weights (JAMIE_PREFETCH=2): 631.9 against 628.3 us with the weights alone (632.6 without): the extra ranges stretch the
        BatchNorm launches by what the next launches gain -- the default (1) prefetches the weights only.  JAMIE_PREFETCH=0: off."""
        mode = str(TUNING['prefetch'])
        f32_mode = str(TUNING['prefetch_f32'])
        if mode == '0' or self.pipeline or (not self.bf16 and f32_mode == '0'):
            return None
        out = []
        for it in items:
            if it.startswith('W:') and self._zs is not None:
                continue          # (sharded optimiser: those weights are ARRIVING by all-gather; a rider would read what RCCL writes)
            if it.startswith('W:'):
                lo, hi = self.m.layout.regions[it[2:]]
                lo = (lo + 7) // 8 * 8                   # (16-byte aligned in the bf16 copy)
                out.append((self.wbf_flat if self.bf16 else self.m.flat)[lo:hi])
            elif mode != '1':
                for w in self.ws:
                    t = w.get(it)
                    if t is not None and t.numel() * t.element_size() >= (1 << 20):
                        out.append(t[0] if t.dim() == 3 else t)
        return out[:8] or None

    def _bn_fwd(self, layer, h_key, out_key, stream_base, noise, kind, j, prefetch=()):
        probs = []
        for i, d in enumerate(self.dims):
            w, P, bn = self.ws[i], self.m.p, self.m.bn
            h = w[h_key]
            pr = nv.BnFwdProblem()
            pr.h, pr.nslab, pr.slab_stride = nv.ptr(h), h.shape[0], h.stride(0)
            pr.panel = int(self._paneled(h_key))
            pr.gamma, pr.beta = nv.ptr(P[f'm{i}.{layer}.g']), nv.ptr(P[f'm{i}.{layer}.b'])
            pr.running_mean, pr.running_var = nv.ptr(bn[f'm{i}.{layer}.mean']), nv.ptr(bn[f'm{i}.{layer}.var'])
            pr.save_mean, pr.save_invstd = nv.ptr(w[layer + '.mean']), nv.ptr(w[layer + '.invstd'])
            pr.out, pr.mask = nv.ptr(w[out_key]), nv.ptr(self._mask(noise, kind, i, j))
            pr.B, pr.N, pr.rng_stream = self.B, h.shape[2], stream_base + 8 * i
            if self.fuse_bf16:      # bf16 + transposed bf16 copies straight from the strip; no fp32 activation
                pr.out, pr.out_bf16 = None, nv.ptr(w[out_key + '_bf'])
                pr.outT_bf16 = nv.ptr(w[out_key + '_T']) if out_key in self.need_T else None
            probs.append(pr)
        if not self.bf16 and str(TUNING['prefetch_f32']) == 'bwd':
            prefetch = ()
        nv.bn_act_fwd(probs, self.p_drop, self.state, BN_MOMENTUM, BN_EPS, LRELU_SLOPE, self._prefetch(*prefetch))

    def _bn_bwd(self, layer, da_key, h_key, lin, stream_base, noise, kind, j, colsums=None, prefetch=()):
        probs = []
        for i, d in enumerate(self.dims):
            w, P = self.ws[i], self.m.p
            da, h = w[da_key], w[h_key]
            pr = nv.BnBwdProblem()
            pr.da, pr.nslab, pr.slab_stride = nv.ptr(da), da.shape[0], da.stride(0)
            pr.panel = int(self._paneled(da_key))             # (da and h of one layer share the layout: BN_OF)
            pr.h, pr.gamma, pr.beta = nv.ptr(h), nv.ptr(P[f'm{i}.{layer}.g']), nv.ptr(P[f'm{i}.{layer}.b'])
            pr.save_mean, pr.save_invstd = nv.ptr(w[layer + '.mean']), nv.ptr(w[layer + '.invstd'])
            pr.dgamma, pr.dbeta = nv.ptr(self.g[f'm{i}.{layer}.g']), nv.ptr(self.g[f'm{i}.{layer}.b'])
            pr.dbias_lin = nv.ptr(self.g[f'm{i}.{lin}.b'])
            pr.mask = nv.ptr(self._mask(noise, kind, i, j))
            pr.B, pr.N, pr.rng_stream, pr.accumulate = self.B, h.shape[2], stream_base + 8 * i, int(self.accumulate)
            if self.fuse_bf16:
                pr.dh_bf16, pr.skip_f32 = nv.ptr(w[da_key + '_bf']), 1
                pr.dhT_bf16 = nv.ptr(w[da_key + '_T']) if da_key in self.need_T else None
            probs.append(pr)
        nv.bn_act_bwd(probs, self.p_drop, self.state, LRELU_SLOPE, colsums, self._prefetch(*prefetch))

    def _fwd_block(self, a_key, lin, h_key, sk_key, layer, out_key, stream_base, noise, kind, j):
        """out = Dropout(LeakyReLU(BatchNorm(a W^T + b)))  (model.py:151-154 and siblings): the product (split-K slabs), then the
        BatchNorm launch that sums them.  (The ONE-launch form with an in-launch split-K hand-off was built and measured slower in
        round 3: +38 / +120 us per step, profiles/r03_ab_fused_bn_rejected.log; it lives in the experiments build.)"""
        self._fwd_gemm(a_key, lin, h_key, sk_key)
        self._bn_fwd(layer, h_key, out_key, stream_base, noise, kind, j,
                     prefetch={'enc0': ('W:enc1',), 'dec1': ('W:dec2', 'x')}.get(lin, ()))      # (x: the MSE launch reads it)
        self._cast(out_key)

    def _fwd_gemm(self, a_key, lin, out_key, sk_key, with_bias=True):
        """out[B, out_f] (slabs) = a[B, in_f] W^T (+ b)."""
        probs = []
        for i, d in enumerate(self.dims):
            w, P = self.ws[i], self.m.p
            a, W, out = w[a_key], P[f'm{i}.{lin}.W'], w[out_key]
            nout, nin = W.shape
            if self.bf16:
                a, W = w[a_key + '_bf'], self.wbf[f'm{i}.{lin}.W']
            probs.append(nv.gemm_problem(a, W, out, self.B, nout, nin, nin, nin, nout,
                                         bias=P[f'm{i}.{lin}.b'] if with_bias else None,
                                         splitk=w['sk'][sk_key], slab_stride=out.stride(0),
                                         c_panel=self._paneled(out_key, self.gcfg.get(sk_key, -1))))
        cfg = self.gcfg.get(sk_key, -1)
        fcfg = self.fcfg.get(sk_key, -1)
        if (not self.bf16 and self.pipeline and fcfg == F32_CFG_ROWS and lin in str(TUNING['f32_pipe_solo']).split(',')):      # (not 20: one per CU as it is)
            fcfg = F32_CFG_SOLO           # (same tile, same K slices, same sums: one workgroup per CU while clip + Adam streams beside it)
        self._wait_params(lin)
        # 'enc_gemm': every large forward launch; 'enc0_gemm': the encoder's first Linear alone (model.py:151, d -> 2d, both
        # modalities: the matmul north_star's roofline target names; bench.py's roofline.encoder_gemm)
        label = 'enc_gemm' if lin in ('enc0', 'enc1', 'dec1') or (lin == 'dec2' and fcfg >= 0) else lin
        self._launch((label, 'enc0_gemm') if lin == 'enc0' else label,
                     (lambda: nv.gemm_bf16(probs, cfg)) if self.bf16 else (lambda: nv.gemm(probs, nv.NT, fcfg)))

    def _dx_gemm(self, dy_key, lin, out_key, sk_key):
        """dx[B, in_f] (slabs) = dy[B, out_f] W."""
        probs = []
        for i, d in enumerate(self.dims):
            w, P = self.ws[i], self.m.p
            dy, W, out = w[dy_key], P[f'm{i}.{lin}.W'], w[out_key]
            nout, nin = W.shape
            if self.bf16 and f'm{i}.{lin}' not in self.wT:      # dx = dy W on W [out, in] as stored (b_tr)
                probs.append(nv.gemm_problem(w[dy_key + '_bf'], self.wbf[f'm{i}.{lin}.W'], out, self.B, nin, nout,
                                             nout, nin, nin, splitk=w['sk'][sk_key], slab_stride=out.stride(0), b_tr=True,
                                             c_panel=self._paneled(out_key, self.gcfg.get(sk_key, -1))))
            elif self.bf16:   # dx = dy W  ==  dy (W^T)^T with the K-contiguous transposed copy (skinny layers)
                probs.append(nv.gemm_problem(w[dy_key + '_bf'], self.wT[f'm{i}.{lin}'], out, self.B, nin, nout,
                                             nout, nout, nin, splitk=w['sk'][sk_key], slab_stride=out.stride(0),
                                             c_panel=self._paneled(out_key, self.gcfg.get(sk_key, -1))))
            else:
                probs.append(nv.gemm_problem(dy, W, out, self.B, nin, nout, nout, nin, nin,
                                             splitk=w['sk'][sk_key], slab_stride=out.stride(0)))
        if self.bf16:
            nv.gemm_bf16(probs, self.gcfg.get(sk_key, -1))
        else:
            cfg = self.fcfg.get(sk_key, -1)
            if cfg < 0 and TUNING['f32_dx_cfg'] is not None and sk_key in ('d_e2', 'd_a1'):
                cfg = int(TUNING['f32_dx_cfg'])
            nv.gemm(probs, nv.NN, cfg)

    def _dw_problems(self, dy_key, a_key, lin, only=None):
        probs = []
        for i, d in enumerate(self.dims):
            if only is not None and i not in only:
                continue
            w = self.ws[i]
            dy, a, dW = w[dy_key], w[a_key], self.g[f'm{i}.{lin}.W']
            nout, nin = dW.shape
            if self.bf16:
                probs.append(self._dw_problem(i, dy_key, a_key, lin))
            else:
                probs.append(nv.gemm_problem(dy, a, dW, nout, nin, self.B, nout, nin, nin, accumulate=self.accumulate,
                                             store_nt=bool(TUNING['dw_store_nt']), partial=self._dw_partial(i, lin)))
        return probs

    def _dw_gemm(self, dy_key, a_key, lin, extra=None, ranges=None, only=None):
        """dW[out_f, in_f] = dy[B, out_f]^T a[B, in_f] into the flat gradient buffer.  `extra` = [(dy_key, a_key, lin)] of
        skinny layers whose dW rides in the same launch; `ranges` (bf16 large-tile launch only): the range-norm work rides too."""
        probs = self._dw_problems(dy_key, a_key, lin, only)
        for ex in (extra or []):
            # (fp32: a skinny layer does not ride in a bf16x3 launch -- its product would then depend on which launch carried it;
            #  it keeps the fp32 pipe and a launch of its own, as in the grouped order of _flush_dw)
            if len(probs) + self.M <= (nv.MAX_GEMM_GROUP if self.bf16 else nv.MAX_GEMM_GROUP_F32) \
                    and (not self.bf16 or self._dw_cfg(lin) == self._dw_cfg(ex[2])) \
                    and (self.bf16 or self._f32_dw_cfg(lin) != F32_CFG_X3 or self._f32_dw_cfg(ex[2]) == F32_CFG_X3):
                probs += self._dw_problems(*ex)
            else:
                self._dw_gemm(*ex)
        if self.bf16:
            nv.gemm_bf16(probs, self._dw_cfg(lin), ranges)
        else:
            nv.gemm(probs, nv.TN, self._f32_dw_cfg(lin))

    def _flush_dw(self, last):
        """fp32: the large layers' dW products that waited (+ `last`, the first layer's), TUNING['f32_dw_group'] layers per launch,
        the latest gradients first.  One layer is 640 tiles of 128 x 128 at config 2 = 1.25 rounds of the chip's 512 slots (half
        the chip idles through the second round); four layers are 5.0 rounds."""
        todo = ([last] if last else []) + self._dw_wait[::-1]
        small, self._dw_wait, self._dw_small = self._dw_small, None, []
        per = max(1, min(int(TUNING['f32_dw_group']), nv.MAX_GEMM_GROUP_F32 // self.M))
        while todo:
            chunk, todo = todo[:per], todo[per:]
            probs = []
            for dy_key, a_key, lin in chunk:
                probs += self._dw_problems(dy_key, a_key, lin)
            nv.gemm(probs, nv.TN, _f32_fused_cfg())
        # the skinny layers' dW (decoder layer 0, heads: 0.3 GFLOP but 16 dependent k-steps, 19 us) keep a launch of their own: as 48
        # more 128 x 128 tiles of the grouped launch, first or last in its grid, they cost 9-12 us more (the launch is exactly 10 tiles
        # per CU without them; profiles/r04_ab_f32_skinny_dw_in_grouped_launch_rejected.log)
        if small:
            self._dw_gemm(*small[0], extra=small[1:])

    def _f32_dw_cfg(self, lin):
        """fp32 dW launch (TN, K = batch): tile configuration (-1: the library's 64 x 64 default)."""
        env = TUNING['f32_dw_cfg']
        big = self.B >= 256 and all(min(self.m.p[f'm{i}.{lin}.W'].shape) >= 512 for i in range(self.M))
        if big and self._f32_dw_fused:
            return _f32_fused_cfg()           # (the partial sums are laid out for this tile)
        small = TUNING['f32_dw_small_cfg']
        return (int(env) if env not in (None, '') else f32_cfg('dw')) if big else (int(small) if small not in (None, '') else -1)

    def _dw_cfg(self, lin):
        """Tile configuration of the dW launch of layer `lin` (-1: the library default for small / skinny problems)."""
        big = all(min(self.m.p[f'm{i}.{lin}.W'].shape) >= 256 for i in range(self.M))
        if getattr(self, 'skinny_tr', False) and lin in ('head', 'dec0'):
            big = True
        return self.gcfg.get('dw', -1) if big else -1

    def _dw_tr(self, lin):
        return self.bf16 and self._dw_cfg(lin) in (24, 25, 29, 30, 32)

    def _dw_problem(self, i, dy_key, a_key, lin):
        # store_nt: the weight gradient is next read by the optimiser, a whole backward pass later; stored through the caches
        # its 161 MB per step stayed behind as dirty lines whose write-back ran into the following kernels (bench.py, bf16:
        # 670 -> 706 k cells/s, clip + Adam 228 -> 208 us; no effect on the dword stores of the fp32 kernel)
        w = self.ws[i]
        dW = self.g[f'm{i}.{lin}.W']
        nout, nin = dW.shape
        if self._dw_tr(lin):      # dy [B, out], a [B, in] row-major as produced: no transposed copies
            if self._direct_now:      # data parallel, bf16 messages: straight into the exchange buffer (no fp32 copy, no cast pass)
                return nv.gemm_problem(w[dy_key + '_bf'], w[a_key + '_bf'], self._direct['views'][f'm{i}.{lin}.W'], nout, nin,
                                       self.B, nout, nin, nin, a_tr=True, b_tr=True, store_nt=bool(TUNING['dw_store_nt']), c_bf16=True)
            g16 = self._g16_now and f'm{i}.{lin}' in self.dw_partial
            return nv.gemm_problem(w[dy_key + '_bf'], w[a_key + '_bf'], self.g16[f'm{i}.{lin}.W'] if g16 else dW, nout, nin,
                                   self.B, nout, nin, nin, accumulate=self.accumulate, partial=self._dw_partial(i, lin),
                                   a_tr=True, b_tr=True, store_nt=bool(TUNING['dw_store_nt']), c_bf16=g16)
        # (dy^T) (a^T)^T on the [features, B] copies, K (= batch) contiguous
        return nv.gemm_problem(w[dy_key + '_T'], w[a_key + '_T'], dW, nout, nin, self.B, self.B, self.B, nin,
                               accumulate=self.accumulate, partial=self._dw_partial(i, lin), store_nt=True)

    def _dw_partial(self, i, lin):
        """Slice of `norm_partials` the dW launch of m{i}.{lin} fills (None: the separate norm kernel reads the gradient)."""
        if not self._fuse_now or f'm{i}.{lin}' not in self.dw_partial:
            return None
        off, t = self.dw_partial[f'm{i}.{lin}']
        return self.norm_partials[off:off + t]

    def _bwd_gemms(self, dy_key, lin, a_key, out_key, sk_key, extra=None, ranges=None):
        """dW (into the gradient buffer) and dX (slabs) of one Linear layer.  In bf16 mode both are the same
        K-contiguous NT product, so the four problems (2 modalities x {dW, dX}) go out as ONE grouped launch.
        `extra` = [(dy_key, a_key, lin)]: the dW problems of skinny layers ride in the same launch."""
        if not self.bf16 and self._dw_wait is not None and self._f32_dw_cfg(lin) == _f32_fused_cfg():
            # fp32: the weight gradient is not on the critical chain -- it waits for the end of the pass (_flush_dw)
            self._dw_wait.append((dy_key, a_key, lin))
            self._dw_small += list(extra or [])
            self._dx_gemm(dy_key, lin, out_key, sk_key)
            return
        if not self.bf16 or 2 * self.M > nv.MAX_GEMM_GROUP:
            self._dw_gemm(dy_key, a_key, lin, extra, ranges)
            self._dx_gemm(dy_key, lin, out_key, sk_key)
            return
        riding, left, n_prob = [], [], 2 * self.M
        for ex in (extra or []):
            if n_prob + self.M <= nv.MAX_GEMM_GROUP and self._dw_cfg(ex[2]) == self.gcfg.get(sk_key, -1) and self._dw_tr(ex[2]):
                riding.append(ex)
                n_prob += self.M
            else:
                left.append(ex)
        if left:                      # (more than two modalities: the skinny layers' dW share a launch of their own)
            self._dw_gemm(*left[0], extra=left[1:])
        # the dX tiles run 2-3x as long as the dW tiles (K = features / slices vs K = batch): they go first in the
        # grid so that the short dW tiles fill in behind them (in-kernel stamps: the launch ends 4-5 us earlier).
        # (fp32's order -- dX launches alone, every dW tile in one last launch -- costs bf16 +33 us per step: here the dW tiles ARE
        #  the filler of the dX launches, profiles/r04_ab_bf16_dw_all_in_last_launch_rejected.log)
        probs = []
        for i, d in enumerate(self.dims):
            w = self.ws[i]
            nout, nin = self.g[f'm{i}.{lin}.W'].shape
            if f'm{i}.{lin}' not in self.wT:                    # W [out, in] as stored (b_tr): no transposed copy
                probs.append(nv.gemm_problem(w[dy_key + '_bf'], self.wbf[f'm{i}.{lin}.W'], w[out_key], self.B, nin, nout,
                                             nout, nin, nin, splitk=w['sk'][sk_key], slab_stride=w[out_key].stride(0), b_tr=True,
                                             c_panel=self._paneled(out_key, self.gcfg.get(sk_key, -1))))
            else:
                probs.append(nv.gemm_problem(w[dy_key + '_bf'], self.wT[f'm{i}.{lin}'], w[out_key], self.B, nin, nout,
                                             nout, nout, nin, splitk=w['sk'][sk_key], slab_stride=w[out_key].stride(0),
                                             c_panel=self._paneled(out_key, self.gcfg.get(sk_key, -1))))
        for i, d in enumerate(self.dims):
            probs.append(self._dw_problem(i, dy_key, a_key, lin))
        for ex in riding:
            probs += self._dw_problems(*ex)
        nv.gemm_bf16(probs, self.gcfg.get(sk_key, -1), ranges)

    def _latent_desc_m(self, corr, Fblk, noise):
        """Identity correspondence, F = 0, euclidean alignment (every BASELINE config; any 2 <= M <= 4): the fused latent
        kernels (jamie_latent_m_*): forward = heads' slabs -> mu / logvar / z / comb / loss partials AND decoder layer 0
        in one launch; backward = d(mu | logvar) + head-bias gradients + losses in two."""
        if corr is not None or Fblk is not None or self.cosine:
            raise NotImplementedError('more than two modalities: identity correspondence, F = 0, euclidean only')
        B, L = self.B, self.L
        d = nv.LatentM()
        d.B, d.L, d.M = B, L, self.M
        for i in range(self.M):
            w = self.ws[i]
            d.ml[i] = nv.ptr(w['ml']); d.head_bias[i] = nv.ptr(self.m.p[f'm{i}.head.b'])
            d.eps_in[i] = nv.ptr(noise['eps'][i]) if noise is not None else None
            for k in ('mu', 'lv', 'z', 'eps', 'dml'):
                getattr(d, k)[i] = nv.ptr(w[k])
            d.dcomb[i] = nv.ptr(w['dcomb'])
            d.comb_alias[i] = nv.ptr(w['comb'])
            d.g1[i] = nv.ptr(w['g1'])
            d.dec0_W[i], d.dec0_b[i] = nv.ptr(self.m.p[f'm{i}.dec0.W']), nv.ptr(self.m.p[f'm{i}.dec0.b'])
            d.d[i] = self.dims[i]
            d.dbias_head[i] = nv.ptr(self.g[f'm{i}.head.b'])
            if self._fuse_da2():
                d.head_W[i], d.da2[i] = nv.ptr(self.m.p[f'm{i}.head.W']), nv.ptr(w['da2'])
                d.da2_panel = int(self._paneled('da2'))
            if self.bf16:
                if getattr(self, '_heads_in_latent', False):
                    d.heads_a_bf16[i], d.heads_W_bf16[i] = nv.ptr(w['a2_bf']), nv.ptr(self.wbf[f'm{i}.head.W'])
                d.dml_bf16[i] = nv.ptr(w['dml_bf'])
                d.dmlT_bf16[i] = nv.ptr(w['dml_T']) if 'dml' in self.need_T else None
                d.comb_bf16[i] = nv.ptr(w['comb_bf'])
                d.combT_bf16[i] = nv.ptr(w['comb_T']) if 'comb' in self.need_T else None
        d.g1_panel = int(self._paneled('g1'))
        d.comb = nv.ptr(self.ws[0]['comb'])
        d.colpart, d.accumulate, d.ticket = nv.ptr(self.lat_colpart), int(self.accumulate), nv.ptr(self.lat_ticket)
        d.ml_nslab, d.ml_slab_stride = self.ws[0]['ml'].shape[0], B * 2 * L
        d.sigma, d.hyper, d.partials = nv.ptr(self.m.p['sigma']), nv.ptr(self.hyper), nv.ptr(self.lat_partials)
        d.dcomb_nslab, d.dcomb_slab_stride = self.ws[0]['sk']['d_comb'], B * L
        d.dsigma = nv.ptr(self._dsig_tmp if self.accumulate else self.g['sigma'])
        d.rec_partials, d.n_rec_partials = nv.ptr(self.rec_partials), self.rec_partials.numel()
        d.losses, d.rng_stream = nv.ptr(self.losses), 100
        return d

    def _fuse_da2(self):
        """The heads' input gradient d a2 = d(mu | logvar) W_head as extra workgroups of the fused latent backward launch
        (exact fp32, K = 2L) instead of a GEMM launch; the heads' dW then rides in the next layer's launch."""
        return (all(d % 4 == 0 for d in self.dims) and all(w['sk']['d_a2'] == 1 for w in self.ws)
                and (not self.bf16 or self.skinny_tr) and TUNING['fused_da2'])

    def _fused_latent(self, corr, Fblk):
        if self.M == 2 and not TUNING['fused_latent']:      # (A/B: the general kernels)
            return False
        return corr is None and Fblk is None and not self.cosine and self.L <= 128

    def _latent_desc(self, corr, Fblk, noise, fused=False):
        if self.M != 2 or fused:
            return self._latent_desc_m(corr, Fblk, noise)
        B, L = self.B, self.L
        d = nv.Latent()
        d.B, d.L = B, L
        for i in range(2):
            w = self.ws[i]
            d.ml[i] = nv.ptr(w['ml']); d.head_bias[i] = nv.ptr(self.m.p[f'm{i}.head.b'])
            d.eps_in[i] = nv.ptr(noise['eps'][i]) if noise is not None else None
            for k in ('mu', 'lv', 'z', 'eps', 'comb', 'cz', 'H', 'ch', 'dml'):
                getattr(d, k)[i] = nv.ptr(w[k])
            d.dcomb[i] = nv.ptr(w['dcomb'])
        d.ml_nslab, d.ml_slab_stride = self.ws[0]['ml'].shape[0], B * 2 * L
        d.sigma, d.corr, d.Fblk, d.hyper = nv.ptr(self.m.p['sigma']), nv.ptr(corr), nv.ptr(Fblk), nv.ptr(self.hyper)
        d.rsum, d.qsum, d.fc1, d.fte = nv.ptr(self.rsum), nv.ptr(self.qsum), nv.ptr(self.fc1), nv.ptr(self.fte)
        d.partials = nv.ptr(self.lat_partials)
        d.dcomb_nslab, d.dcomb_slab_stride = self.ws[0]['sk']['d_comb'], B * L
        d.dsigma = nv.ptr(self._dsig_tmp if self.accumulate else self.g['sigma'])
        d.rec_partials, d.n_rec_partials = nv.ptr(self.rec_partials), self.rec_partials.numel()
        d.losses = nv.ptr(self.losses)
        d.cosine, d.rng_stream = int(self.cosine), 100
        if self.bf16:        # the latent kernels write the bf16 / transposed copies of comb and d(mu|logvar) themselves
            for i in range(2):
                w = self.ws[i]
                d.comb_bf16[i], d.combT_bf16[i] = nv.ptr(w['comb_bf']), nv.ptr(w['comb_T'])
                d.dml_bf16[i], d.dmlT_bf16[i] = nv.ptr(w['dml_bf']), nv.ptr(w['dml_T'])
        return d

    # ---- the step ----
    def _batch_problems(self, data, idx):
        """cast_transpose problems of the batch: row gather + fp32 / bf16 (/ transposed) copies (fp32 mode: the gather only)."""
        if not self.bf16:
            return [nv.cast_problem(data[i], None, None, rows=idx[i], dst32=self.ws[i]['x']) for i in range(self.M)]
        return [nv.cast_problem(data[i], self.ws[i]['x_bf'], self.ws[i]['x_T'] if 'x' in self.need_T else None,
                                rows=idx[i], dst32=self.ws[i]['x']) for i in range(self.M)]

    def load_batch(self, data, idx, with_wT=True):
        """x_i = data_i[idx_i]  (jamie.py:583).  `idx` = list of int32 device tensors.  `with_wT=False`: the batch is being
        loaded while the optimiser may still be writing the weights (prefetch): leave the weight transposes to _backward."""
        if self.bf16:      # gather + bf16 copy (+ transposed copy) of the batch, and the skinny weights' transposes: one launch
            # (pipelined optimiser: the optimiser stream makes those transposes behind each group's update; made here, beside it,
            #  they read weights that clip + Adam may be half-way through and race its own copies: the next backward pass then
            #  ran on stale skinny weights every few steps -- found by test_pipelined_optimizer_is_bit_identical run on its own)
            with_wT = with_wT and not self.pipeline
            probs = self._batch_problems(data, idx)
            nv.cast_transpose(probs + (self._wT_problems() if with_wT else []))
            if with_wT:
                self._wT_stale = False
            return
        for i in range(self.M):
            nv.gather_rows(data[i], idx[i], self.ws[i]['x'])

    def _region(self, ar, name):
        """Tell an overlapping all-reduce that the gradients of parameter region `name` have been launched."""
        if ar is not None and hasattr(ar, 'region_done'):
            if name not in self.m.layout.regions:      # (heads / decoder layer 0: their gradients travel with `rep`)
                return
            lo, hi = self.m.layout.regions[name]
            if self._direct_now:
                fn = lambda: ar.region_done(self.grad, lo, hi, precast=True)   # noqa: E731
            else:
                fn = lambda: ar.region_done(self.grad, lo, hi)   # noqa: E731
            nv.record_callable(fn)
            fn()

    def pad_cells(self, data):
        """[N, d_i] cell matrices -> [N, pdims_i] (zero columns appended) when the model is padded; else unchanged."""
        return [x if x.shape[1] == d else torch.nn.functional.pad(x, (0, d - x.shape[1])).contiguous()
                for x, d in zip(data, self.dims)]

    def set_grad_bf16(self, on):
        """Switch the bf16 weight-gradient buffer (default on in bf16 compute mode on one GPU) off / on; off for runs that
        accumulate gradients over batches (batch_step=False)."""
        self.grad_bf16 = bool(on) and self.fused_norm and hasattr(self, 'grad16')

    def grad_view(self, name):
        """The gradient of parameter tensor `name` (fp32, without padding) as the last backward pass left it: the large
        weight matrices live in the bf16 buffer when that pass wrote them there."""
        if self._g16_last and name.endswith('.W') and (self._direct_now or name[:-2] in self.dw_partial):
            return self.m.layout.unpad(name, self.g16[name].float())
        return self.m.layout.unpad(name, self.g[name])

    def grad_flat(self):
        """The whole flat gradient of the last backward pass as one fp32 tensor (tests / diagnostics)."""
        out = self.grad.clone()
        if self._g16_last:
            keys = [k[:-2] for k in self.m.layout.entries if k.endswith('.W')] if self._direct_now else self.dw_partial
            for key in keys:
                o, shp = self.m.layout.entries[key + '.W']
                n = shp[0] * shp[1]
                out[o:o + n] = self.grad16[o:o + n].float()
        return out

    def set_batch(self, X):
        """Use the given [B, d_i] fp32 matrices as the batch (tests; the training loop uses `load_batch`)."""
        for i in range(self.M):
            x = self.ws[i]['x']
            if X[i].shape[1] != x.shape[1]:
                x.zero_()
            x[:, :X[i].shape[1]].copy_(X[i])
        self._cast('x')
        if self.bf16 and self._wT_stale:
            self.refresh_weights_bf16(transposes_only=True)
            self._wT_stale = False

    def forward_backward(self, corr=None, Fblk=None, noise=None, allreduce=None):
        """Forward, losses and backward for the batch already in the workspace.  `corr` None = identity,
        `Fblk` None = 0; `noise` (explicit masks / eps, for parity tests) None = Philox streams."""
        lat = self._forward(corr, Fblk, noise, True)
        self._backward(lat, noise, allreduce)

    def forward_only(self, corr=None, noise=None):
        """Train-mode forward without losses (autograd seam): fills z, comb, mu, lv, xhat; returns the latent
        descriptor to hand to `backward_external`."""
        return self._forward(corr, None, noise, False)

    def backward_external(self, lat, dz, dcomb, dxhat, dmu, dlv_last, noise=None):
        """Backward from caller-supplied gradients of the forward outputs (lists per modality; None = zero)."""
        if self.M != 2:
            raise NotImplementedError('the autograd seam follows the reference: two modalities')
        B, L = self.B, self.L
        keep = []
        for i in range(2):
            w = self.ws[i]
            w['dxhat'].copy_(dxhat[i]) if dxhat[i] is not None else w['dxhat'].zero_()
            ext = w['dcomb'][w['sk']['d_comb']]
            ext.copy_(dcomb[i]) if dcomb[i] is not None else ext.zero_()
            for name, src in (('dz_ext', dz[i]), ('dmu_ext', dmu[i])):
                t = None if src is None else src.contiguous()
                keep.append(t)
                getattr(lat, name)[i] = nv.ptr(t)
        t = None if dlv_last is None else dlv_last.contiguous()
        keep.append(t)
        lat.dlv_ext = nv.ptr(t)
        lat.dcomb_nslab = self.ws[0]['sk']['d_comb'] + 1
        self._cast('dxhat')
        self._backward(lat, noise, None)
        return keep

    def _forward(self, corr, Fblk, noise, fused_losses):
        B, L = self.B, self.L
        self._dxhat_cs = 'dxhat'
        self._set_panels(fused_losses and self._fused_latent(corr, Fblk))
        # ---------------- forward ----------------
        self._fwd_block('x', 'enc0', 'h1', 'enc0', 'bn0', 'a1', 10, noise, 'enc_masks', 0)
        self._fwd_block('a1', 'enc1', 'h2', 'enc1', 'bn1', 'a2', 11, noise, 'enc_masks', 1)
        fused = fused_losses and self._fused_latent(corr, Fblk)
        self._heads_in_latent = bool(fused and self.bf16 and self.L <= 64 and TUNING['fused_heads'])
        if not self._heads_in_latent:
            self._fwd_gemm('a2', 'head', 'ml', 'head', with_bias=False)
        lat = self._latent_desc(corr, Fblk, noise, fused)
        nv.latent_fwd(lat, self.state)
        if not fused:            # (the fused kernel has written g1 = comb W^T + b and the bf16 copies of comb itself)
            self._cast('comb')
            self._fwd_gemm('comb', 'dec0', 'g1', 'dec0')
        self._bn_fwd('bn2', 'g1', 'e1', 12, noise, 'dec_masks', 0, prefetch=('W:dec1',))
        self._cast('e1')
        self._fwd_block('e1', 'dec1', 'g2', 'dec1', 'bn3', 'e2', 13, noise, 'dec_masks', 1)
        if not fused_losses:                                              # plain x_hat (autograd seam)
            for w, d in zip(self.ws, self.dims):
                if w['xhat'] is None:
                    w['xhat'] = torch.empty(1, B, d, device=self.dev, dtype=torch.float32)
            self._fwd_gemm('e2', 'dec2', 'xhat', 'dec0')
            return lat
        if (self.bf16 and self.gcfg.get('dec2', -1) >= 0) or self.fcfg.get('dec2', -1) >= 0:   # split-K x_hat GEMM, then MSE (+ casts)
            self._fwd_gemm('e2', 'dec2', 'xh', 'dec2')
            probs, off = [], 0
            for i, d in enumerate(self.dims):
                w = self.ws[i]
                rd = self.rdims[i]
                cp = bool(TUNING['mse_colpart'])
                probs.append(nv.mse_problem(w['xh'], w['x'], None if (self.bf16 and cp) else w['dxhat'], w.get('dxhat_bf'),
                                            w['dxhat_T'] if 'dxhat' in self.need_T else None,
                                            partial=self.rec_partials[off:off + self.rec_tiles[i]],
                                            scale=self.loss_weights[1] * 2.0 / (B * rd), pscale=1.0 / (B * rd),
                                            colpart=w['dxhat_cp'] if cp else None))
                off += self.rec_tiles[i]
            nv.mse_cast(probs)
            self._dxhat_cs = 'dxhat_cp' if TUNING['mse_colpart'] else 'dxhat'
            return lat
        probs, off = [], 0
        for i, d in enumerate(self.dims):                                 # x_hat GEMM + fused MSE
            w, P = self.ws[i], self.m.p
            W = P[f'm{i}.dec2.W']
            probs.append(nv.gemm_problem(w['e2_bf'] if self.bf16 else w['e2'], self.wbf[f'm{i}.dec2.W'] if self.bf16 else W,
                                         w['dxhat'], B, d, 2 * d, 2 * d, 2 * d, d,
                                         bias=P[f'm{i}.dec2.b'], epi=nv.EPI_MSE, aux=(w['x'], None, None, None),
                                         aux_ld=d, partial=self.rec_partials[off:off + self.rec_tiles[i]],
                                         scale=self.loss_weights[1] * 2.0 / (B * self.rdims[i]),
                                         pscale=1.0 / (B * self.rdims[i])))
            off += self.rec_tiles[i]
        self._wait_params('dec2')
        self._launch('enc_gemm', (lambda: nv.gemm_bf16(probs)) if self.bf16 else (lambda: nv.gemm(probs, nv.NT)))
        self._cast('dxhat')
        return lat

    def _direct_setup(self, allreduce):
        """Data parallel with bf16 messages: every producer writes its gradients into the exchange's bf16 buffer itself -- the
        dW epilogues store bf16 there (c_bf16; no fp32 copy of the weight gradients), the few KB of bias / BatchNorm / sigma
        gradients are copied by a rider of the last GEMM launch (range_norm.h with the message buffer as the
        bf16 copy; no step-counter increment) -- so the fp32 -> bf16 cast pass over the 161 MB gradient (a second stream beside
        the backward GEMMs, two events per region) does not exist.  Returns None where the path does not apply."""
        ok = (allreduce is not None and hasattr(allreduce, 'message_buffer') and getattr(allreduce, 'comm_dtype', None) == torch.bfloat16
              and (getattr(allreduce, 'world', 1) > 1 or getattr(allreduce, 'single', False)) and self.bf16 and self.grad.is_cuda and not self.accumulate
              and self.gcfg.get('dw', -1) == BF16_CFG_DW and self.skinny_tr and 3 * self.M <= nv.MAX_GEMM_GROUP
              and all(self.gcfg.get(k, -1) == BF16_CFG_DW for k in ('d_e2', 'd_e1', 'd_a1'))
              and self._fused_latent(None, None) and self._fuse_da2() and TUNING['direct_comm'])
        if not ok:
            return None
        comm = allreduce.message_buffer(self.grad)
        if self._direct is None or self._direct['comm'] is not comm:
            lay = self.m.layout
            # the small tensors all live in region `rep`, which is announced last (with enc0): ONE rider, on the enc0 dW launch,
            # copies what no GEMM epilogue writes there (everything but the two skinny weight matrices per modality)
            lo, hi = lay.regions['rep']
            big = sorted((o, o + int(np.prod(shp))) for k, (o, shp) in lay.entries.items() if k.endswith('.W') and lo <= o < hi)
            rest, pos = [], lo
            for a, b in big:
                if a > pos:
                    rest.append((pos, a - pos))
                pos = b
            if hi > pos:
                rest.append((pos, hi - pos))
            # the latent block's finalisation (d sigma, the head-bias gradients: fp32 and bf16 copies) is deferred to that same
            # rider: those slots are cut out of its ranges
            cut = sorted((o, o + (int(np.prod(shp)) + 3) // 4 * 4) for o, shp in
                         [lay.entries['sigma']] + [lay.entries[f'm{i}.head.b'] for i in range(self.M)])
            small = []
            for lo_, ln_ in rest:
                hi_ = lo_ + ln_
                for c0, c1 in cut:
                    if c0 >= hi_ or c1 <= lo_:
                        continue
                    if c0 > lo_:
                        small.append((lo_, c0 - lo_))
                    lo_ = max(lo_, c1)
                if hi_ > lo_:
                    small.append((lo_, hi_ - lo_))
            rg = nv.SqRanges(small)
            rides = {'enc0': (self.grad, comm, rg, torch.zeros(rg.blocks + 1, device=self.dev, dtype=torch.float32), None, None)}
            self._direct = {'comm': comm, 'views': lay.views(comm), 'rides': rides}
        return self._direct

    def _backward(self, lat, noise, allreduce, sample=None):
        B, L = self.B, self.L
        acc = self.accumulate
        direct = self._direct_setup(allreduce) if isinstance(lat, nv.LatentM) and lat.da2[0] else None
        self._direct_now = direct is not None
        dr = dict(direct['rides']) if direct else {}
        if direct:                     # the finaliser rides with the small pieces (enc0 launch, the last of the pass)
            lat.defer_final = 1
            dr['enc0'] = dr['enc0'][:5] + (lat,)
        self._wait_wT()
        if self.bf16 and self._wT_stale:          # an optimiser step without a new batch since (tests): refresh here
            self.refresh_weights_bf16(transposes_only=True)
            self._wT_stale = False
        self._fuse_now = self.fused_norm and allreduce is None      # a reduced gradient needs its norm taken afterwards
        self._dw_wait = [] if (not self.bf16 and allreduce is None and int(TUNING['f32_dw_group']) > 1
                               and self.M <= nv.MAX_GEMM_GROUP_F32) else None
        self._dw_small = []
        self._g16_now = self.grad_bf16 and self._fuse_now and not self.accumulate
        if self.accumulate and self._g16_pending:
            raise nv.JamieHipError('gradients accumulate onto a backward pass that wrote bf16 weight gradients: call '
                                   'set_grad_bf16(False) before the first backward pass of an accumulating run')
        self._g16_last = self._g16_pending = self._g16_now
        if self._direct_now:           # (grad_view / grad_flat read the weight gradients from the message buffer)
            self.grad16, self.g16, self._g16_last = direct['comm'], direct['views'], True
        # the decoder's output-bias gradient (column sums of d x_hat) rides in the first BatchNorm-backward launch as extra
        # workgroups (47 short ones beside 375 long ones) instead of being a launch of its own at the head of the backward
        # pass (with a gradient exchange too: the biases live in region `rep`, which is announced last)
        ride = bool(TUNING['cs_ride'])
        # (column sums of d x_hat: of the per-tile sums jamie_mse_cast left, where that launch made d x_hat; of d x_hat itself otherwise)
        cs_items = [(self.ws[i][self._dxhat_cs], self.g[f'm{i}.dec2.b']) for i in range(len(self.dims))]
        if not ride:
            nv.colsum_group(cs_items, acc)
        self._bwd_gemms('dxhat', 'dec2', 'e2', 'de2', 'd_e2', ranges=dr.get('dec2'))
        self._region(allreduce, 'dec2')
        self._bn_bwd('bn3', 'de2', 'g2', 'dec1', 13, noise, 'dec_masks', 1,
                     colsums=nv.colsum_problems(cs_items, acc) if ride else None,
                     prefetch=('W:dec1', 'e1_bf', 'g1'))   # de2[0] <- dg2p   (next: dX / dW of dec1, then BatchNorm backward on g1)
        self._cast('de2')
        self._bwd_gemms('de2', 'dec1', 'e1', 'de1', 'd_e1', ranges=dr.get('dec1'))
        self._region(allreduce, 'dec1')
        self._bn_bwd('bn2', 'de1', 'g1', 'dec0', 12, noise, 'dec_masks', 0, prefetch=('h2', 'a2_bf'))   # de1[0] <- dg1p
        self._cast('de1')
        # the heads' input gradient comes out of the latent backward launch (fused kernels); the dW products of the two skinny
        # layers (decoder layer 0, heads: K = batch, the longest tiles of their launches) then ride in the next big layer's
        # launch, and what is left here is the short d comb product alone
        late_dw = []
        fused_tail = isinstance(lat, nv.LatentM) and bool(lat.da2[0])
        if fused_tail and (self._direct_now or TUNING['late_dec0_dw']):
            self._dx_gemm('de1', 'dec0', 'dcomb', 'd_comb')
            late_dw.append(('de1', 'comb', 'dec0'))
        else:
            self._bwd_gemms('de1', 'dec0', 'comb', 'dcomb', 'd_comb')
            self._region(allreduce, 'dec0')
        if sample is not None:                 # + the NEXT batch's sampler as an extra workgroup (make_plan)
            nv.latent_bwd(lat, sample, self.state)
        else:
            nv.latent_bwd(lat)                                                  # dml, dsigma, losses
        if acc:                                                                 # batch_step=False: d(sigma) accumulates
            self.g['sigma'].add_(self._dsig_tmp)
        if not (isinstance(lat, nv.LatentM) and lat.colpart):                   # (fused kernels: bf16 dml + head-bias gradients done)
            self._cast('dml')
            nv.colsum_group([(self.ws[i]['dml'], self.g[f'm{i}.head.b']) for i in range(len(self.dims))], acc)
        if isinstance(lat, nv.LatentM) and lat.da2[0]:      # d a2 came out of the latent launch: only the heads' dW is left,
            late_dw.append(('dml', 'a2', 'head'))           # and it rides in the next layer's launch
        else:
            self._bwd_gemms('dml', 'head', 'a2', 'da2', 'd_a2')
            self._region(allreduce, 'head')
        # (fp32, deferred dW: the skinny layers' dW launch -- 0.3 GFLOP, 16 dependent k-steps, 20 us -- on a second stream beside this
        #  BatchNorm backward launch costs 30 us more than it saves: profiles/r04_ab_f32_skinny_dw_side_stream_rejected.log)
        self._bn_bwd('bn1', 'da2', 'h2', 'enc1', 11, noise, 'enc_masks', 1, prefetch=('W:enc1', 'a1_bf', 'h1', 'x_bf'))   # da2[0] <- dh2p
        self._cast('da2')
        self._bwd_gemms('da2', 'enc1', 'a1', 'da1', 'd_a1', extra=late_dw, ranges=dr.get('enc1'))
        for ex in late_dw:
            self._region(allreduce, ex[2])
        self._region(allreduce, 'enc1')
        self._bn_bwd('bn0', 'da1', 'h1', 'enc0', 10, noise, 'enc_masks', 0)   # da1[0] <- dh1p
        self._cast('da1')
        # (the range-norm launch on a second stream beside this last dW product: +11 us per step for the two cross-stream
        #  events, profiles/r02_ab_range_norm_side_stream_rejected.log)
        # ... but as EXTRA workgroups of that last dW launch the range norm (and the latent finalisation) costs nothing: every
        # other gradient exists by now
        ride = (self._fuse_now and self.bf16 and self._dw_cfg('enc0') == BF16_CFG_DW and TUNING['range_ride'])
        # Data parallel with the replicated optimiser: this is the LAST gradient of the pass, and whatever of it is still on the wire
        # when the pass ends is exposed in full.  The first M - 1 modalities' weight gradients (config 2: 16 of the layer's 20 MB
        # as bf16) go out in a launch of their own and their message is issued at once; the last modality's launch carries the
        # riders, and only its part (+ `rep`) is announced at the end.  (Reasoned, not measured: no multi-GPU box -- so it is an
        # opt-in knob, TUNING['split_last_dw'], not the default.)
        split = (allreduce is not None and hasattr(allreduce, 'region_done') and self._zs is None and self.M >= 2
                 and TUNING['split_last_dw'] and not self.accumulate
                 and (getattr(allreduce, 'world', 1) > 1 or getattr(allreduce, 'single', False)))
        if split:
            lo, hi = self.m.layout.regions['enc0']
            cut = self.m.layout.entries[f'm{self.M - 1}.enc0.W'][0]
            self._dw_gemm('da1', 'x', 'enc0', only=range(self.M - 1))
            if self._direct_now:
                fn = lambda: allreduce.region_done(self.grad, lo, cut, precast=True, force=True)   # noqa: E731
            else:
                fn = lambda: allreduce.region_done(self.grad, lo, cut, force=True)   # noqa: E731
            nv.record_callable(fn)
            fn()
            self._dw_gemm('da1', 'x', 'enc0', only=[self.M - 1], ranges=dr.get('enc0'))
            self._ranges_done = False
            rlo, rhi = self.m.layout.regions['rep']
            for a, b in ((cut, hi), (rlo, rhi)):
                if self._direct_now:
                    fn = lambda a=a, b=b: allreduce.region_done(self.grad, a, b, precast=True)   # noqa: E731
                else:
                    fn = lambda a=a, b=b: allreduce.region_done(self.grad, a, b)   # noqa: E731
                nv.record_callable(fn)
                fn()
            self._norm_ready = self._fuse_now
            self.m.num_batches_tracked += 1
            return
        if self._dw_wait is not None:
            big_last = self._f32_dw_cfg('enc0') == _f32_fused_cfg()
            self._flush_dw(('da1', 'x', 'enc0') if big_last else None)
            if not big_last:
                self._dw_gemm('da1', 'x', 'enc0')
        else:
            self._dw_gemm('da1', 'x', 'enc0', ranges=self._range_args()[1] if ride else dr.get('enc0'))
        self._ranges_done = ride
        self._region(allreduce, 'rep')        # (adjacent to enc0: an all-reduce exchange merges the two into one message)
        self._region(allreduce, 'enc0')
        self._norm_ready = self._fuse_now
        self.m.num_batches_tracked += 1

    def _range_args(self):
        """(live slice of `norm_partials` the optimiser sums, arguments of the range-norm work): the sums of squares of the
        gradient ranges no dW launch covers, + the deferred latent finalisation in an extra workgroup."""
        g16 = self.grad16 if self._g16_now else None
        if self._lat_deferred is not None:
            n_live = self.n_dw_partials + self.sq_ranges_nofin.blocks + 1
            norm = self.norm_partials[:n_live]
            return norm, (self.grad, g16, self.sq_ranges_nofin, norm[self.n_dw_partials:], self.state, self._lat_deferred)
        n_live = self.n_dw_partials + self.sq_ranges.blocks
        norm = self.norm_partials[:n_live]
        return norm, (self.grad, g16, self.sq_ranges, norm[self.n_dw_partials:], self.state, None)

    def _range_norm(self, launch=True):
        """The range-norm launch (unless the last dW launch of the backward pass carried it: `launch` False); returns the live
        slice of `norm_partials`."""
        norm, (g, g16, ranges, part, state, fin) = self._range_args()
        if launch:
            nv.grad_sqnorm_ranges(g, ranges, part, state, g16, fin)
        return norm

    def optimizer_step(self, g16=None, after_norm=None, sample=None, casts=None):
        """clip_grad_norm_(params, 1) + Adam.step (+ zero_grad: gradients are overwritten next step).
        `g16`: the reduced gradient as a flat bf16 tensor with the layout of `self.grad` (data-parallel exchange with
        bf16 messages); default: `self.grad`."""
        grad = self.grad if g16 is None else g16
        if g16 is not None:
            norm = self.norm_partials[:self.n_norm]
            nv.grad_sqnorm(grad, norm, self.state)
        elif self._norm_ready:         # the dW launches of this backward pass wrote their tiles' sums of squares
            norm = self._range_norm(launch=not self._ranges_done)
            if self._g16_now:
                grad = self.grad16
        else:                        # (also: reduced gradient, external backward) one pass over the whole buffer
            norm = self.norm_partials[:self.n_norm]
            nv.grad_sqnorm(self.grad, norm, self.state)
        self._norm_ready = self._g16_pending = self._ranges_done = False
        self._lat_deferred = None
        if after_norm is not None:       # e.g. the next batch's sampler + gather on a side stream, under clip + Adam
            after_norm()
        if not self.pipeline:
            # `sample` / `casts`: extra workgroups of the launch draw the NEXT step's batch indices, or (indices drawn earlier, in
            # the latent backward launch) gather and cast the next batch itself (make_plan)
            self._launch('adam', lambda: nv.clip_adam(self.m.flat, grad, self.exp_avg, self.exp_avg_sq,
                                                       norm, self.hyper, self.state,
                                                       self.wbf_flat if self.bf16 else None, sample, casts))
            if self.bf16 and self.side_transposes:
                # the K-contiguous W^T copies are only read by the NEXT backward pass: they are made on a side stream
                # under the next forward pass (36 us of HBM-bound copying off the critical path)
                side = self.side_stream
                self._both(lambda: (self._ev_adam.record(nv.current_stream()), side.wait_event(self._ev_adam)))
                nv.set_stream(side)
                try:
                    self.refresh_weights_bf16(transposes_only=True)
                    self._both(lambda: self._ev_wT.record(side))
                finally:
                    nv.set_stream(None)
                self._wT_pending = True
            elif self.bf16:
                # the transposed copies of the skinny head / latent weights (0.6 MB; the big layers need none) are only read
                # by the next backward pass: they ride on the next batch's gather / cast launch (load_batch, set_batch)
                # instead of a launch of their own; _backward refreshes them itself if no batch was loaded in between
                self._wT_stale = True
            return
        # pipelined: the optimiser stream takes over once the gradient norm is known; it hands the parameter groups
        # back one by one (events) while the main stream already runs the next step's sampler, gather and forward
        opt = self.opt_stream
        self._both(lambda: (self._ev_grads.record(nv.current_stream()), opt.wait_event(self._ev_grads)))
        nv.set_stream(opt)
        try:
            regions = self.m.layout.regions
            for g, names in enumerate(self.PIPE_GROUPS):
                lo, hi = regions[names[0]][0], regions[names[-1]][1]
                self._launch('adam', lambda: nv.clip_adam(self.m.flat[lo:hi], grad[lo:hi], self.exp_avg[lo:hi],
                                                           self.exp_avg_sq[lo:hi], norm, self.hyper,
                                                           self.state, self.wbf_flat[lo:hi] if self.bf16 else None))
                if self.bf16:
                    self.refresh_weights_bf16(transposes_only=True, lins=self.PIPE_LINS[g])
                ev = self._ev_params[g]
                self._both(lambda ev=ev: ev.record(opt))
        finally:
            nv.set_stream(None)
        self._opt_pending = True

    def step(self, corr=None, Fblk=None, noise=None, allreduce=None, after_norm=None, sample=None, next_casts=None):
        """One training step.  `allreduce`: None (single GPU), a callable on the flat gradient, or an
        `OverlappedGradAllReduce` that is fed parameter regions as the backward pass completes them.
        `sample` (nv.sample_args): the NEXT batch's sampler rides in the clip + Adam launch; with `next_casts` (the next
        batch's gather problems) it rides in the latent backward launch instead (sample.step_add = 1) and the gather
        rides in clip + Adam."""
        # the fused latent backward kernel leaves its finalisation (losses, d sigma, head-bias gradients) to the range-norm
        # launch that follows in this very step when that launch exists (bf16 mode, one GPU, fused gradient norm)
        defer = (self.fused_norm and self.sq_ranges_nofin.blocks <= 128 and allreduce is None and not self.accumulate
                 and self._fused_latent(corr, Fblk)
                 and TUNING['defer_final'])
        lat = self._forward(corr, Fblk, noise, True)
        if defer and isinstance(lat, nv.LatentM):
            lat.defer_final = 1
            self._lat_deferred = lat
        self._backward(lat, noise, allreduce, sample if next_casts else None)
        g16 = None
        if self._zs is not None:
            if allreduce is not self._zs['ex']:
                raise nv.JamieHipError('sharded optimiser: step() must be given the exchange it was enabled with')
            fn = lambda: allreduce.finish(copy_back=False)     # noqa: E731
            nv.record_callable(fn)
            fn()
            if after_norm is not None:
                raise nv.JamieHipError('sharded optimiser: no side-stream batch prefetch')
            self._sharded_step(None if next_casts else sample, next_casts)
            return
        if allreduce is not None:
            # bf16 messages: the reduced gradient stays in the exchange's bf16 buffer; norm and Adam read it there
            in_place = (getattr(allreduce, 'comm_dtype', None) == torch.bfloat16 and self.grad.is_cuda
                        and (getattr(allreduce, 'world', 1) > 1 or getattr(allreduce, 'single', False)))
            if in_place:
                fn = lambda: allreduce.finish(copy_back=False)     # noqa: E731
            else:
                fn = allreduce.finish if hasattr(allreduce, 'finish') else (lambda: allreduce(self.grad))
            nv.record_callable(fn)
            fn()
            if in_place:
                g16 = allreduce.comm
        self.optimizer_step(g16, after_norm, None if next_casts else sample, next_casts)

    # ---- recorded launch plan: one foreign call per launch, no descriptor rebuilding (host cost ~3 us/launch) ----
    def make_plan(self, data, idx, n_rows, replace=False, allreduce=None, prefetch=False):
        """Record one full step (device sampler -> gather -> step) on static buffers and return the plan.
        The recording step is a real step.  'diag' sampling: both modalities use the same index tensor.
        `prefetch`: the sampler and the gather of the NEXT batch run on a side stream right after the gradient-norm
        kernel (which advances the step counter the sampler draws from), i.e. under clip + Adam, instead of in front of
        the next forward pass: the same index stream and the same bits.  Measured: -1 % (clip + Adam loses more to the
        interference than the 16 us are worth, like every other overlap tried on this step), so it is off by default."""
        def next_batch(sample=True):
            if sample:
                nv.sample_indices(idx, n_rows, 0, replace, self.state, 200)
            self.load_batch(data, [idx] * self.M, with_wT=not prefetch)
            if replace:
                nv.corr_from_indices(idx, idx, self.corr)
        corr = self.corr if replace else None
        if not prefetch:
            # the sampler of the NEXT batch rides in this step's clip + Adam launch (one extra workgroup: the same index
            # stream, no launch of its own); the first batch is drawn here, before the recording
            fused = not self.pipeline and (replace or idx.numel() <= 2048) and TUNING['fused_sampler']
            # ... and where the latent backward launch can carry the sampler (fused M-modality kernels, identity
            # correspondence, no weight transposes waiting for Adam's output) the next batch's GATHER rides in clip + Adam
            # too: the batch buffers are free once the last dW product has run, and the step is then fwd + bwd + norm + Adam
            early = (fused and not self.wT and not replace and self._fused_latent(corr, None)
                     and all(x.shape[1] % 4 == 0 and x.dtype == torch.float32 for x in data)
                     and TUNING['gather_ride'])
            if fused:
                nv.sample_indices(idx, n_rows, 0, replace, self.state, 200)
            if early:
                next_batch(sample=False)
                self._plan_keep = (nv.sample_args(idx, n_rows, 0, replace, 200, step_add=1),
                                   self._batch_problems(data, [idx] * self.M))
            elif fused:
                self._plan_keep = (nv.sample_args(idx, n_rows, 0, replace, 200), None)
            nv.begin_record()
            try:
                if not early:
                    next_batch(sample=not fused)
                self.step(corr, None, None, allreduce, sample=self._plan_keep[0] if fused else None,
                          next_casts=self._plan_keep[1] if early else None)
            finally:
                plan = nv.end_record()
            return plan
        side = torch.cuda.Stream(device=self.dev)
        ev_norm, ev_batch = torch.cuda.Event(), torch.cuda.Event()
        pending = [False]
        next_batch()                                    # the recording step's own batch, on the main stream
        self._wT_stale = self.bf16                      # the recorded backward pass refreshes the weight transposes itself

        def wait_batch():
            if pending[0]:
                nv.current_stream().wait_event(ev_batch)

        def after_norm():
            self._both(lambda: (ev_norm.record(nv.current_stream()), side.wait_event(ev_norm)))
            nv.set_stream(side)
            try:
                next_batch()
                self._both(lambda: ev_batch.record(side))
            finally:
                nv.set_stream(None)
            pending[0] = True
        nv.begin_record()
        try:
            self._both(wait_batch)
            self.step(corr, None, None, allreduce, after_norm)
        finally:
            plan = nv.end_record()
        self._plan_keep = (side, ev_norm, ev_batch)
        return plan

    def run_plan(self, plan):
        nv.replay(plan)
        self.m.num_batches_tracked += 1
        self._timing_step += 1

    def operand_precision(self, corr=None, Fblk=None, allreduce=None):
        """What bf16 compute mode rounds, for parity tests that restate the step with the same roundings (the oracle's
        `emulate` argument): ({layer: (fwd, dx, dw)}, grad_bf16).  True = that product of the Linear layer -- forward
        y = a W^T, input gradient dx = dy W, weight gradient dW = dy^T a -- reads bf16 operands (rounded to nearest even once,
        where the producing kernel stores them; fp32 accumulation); `grad_bf16`: the optimiser reads every gradient rounded to
        bf16 while the clip norm is taken from the fp32 values.  With identity correspondence the fused latent launches compute
        decoder layer 0's forward product and the heads' input gradient in exact fp32.  fp32 mode: (None, False)."""
        if not self.bf16:
            return None, False
        fused = self.M != 2 or self._fused_latent(corr, Fblk)
        da2 = fused and self._fuse_da2()
        prec = {k: (True, True, True) for k in ('enc0', 'enc1', 'dec1', 'dec2')}
        prec['head'] = (True, not da2, True)
        prec['dec0'] = (not fused, True, True)
        return prec, bool(self.grad_bf16 and self.fused_norm and allreduce is None and not self.accumulate)

    def adam_bytes_per_param(self):
        """Algorithmic HBM bytes per parameter of one clip + Adam launch: read p, g, m, v; write p, m, v (fp32; the
        gradient is 2 bytes when the dW launches wrote it as bf16)."""
        return 26.0 if self.grad_bf16 else 28.0

    def read_losses(self):
        """Device sync: [KL, Rec, CosSim, F] (weighted), total, running min of total."""
        v = self.losses.tolist()
        return v[:4], v[4], v[5]
