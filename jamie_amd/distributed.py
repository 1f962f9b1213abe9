"""Data-parallel plumbing: one process per GPU, cells sharded by rows, ONE all-reduce of the flat
gradient buffer per step (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; gloo on CPU
for tests).  The reference has no distributed code; this is new.

BatchNorm statistics stay per-rank (no SyncBN): north_star allows exactly one collective per step.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if launched under torchrun."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world <= 1:
        return 0, 1, 0
    rank = int(os.environ['RANK'])
    local = int(os.environ.get('LOCAL_RANK', rank))
    # test hooks: JAMIE_DIST_BACKEND=gloo and JAMIE_SHARE_GPU=1 let two ranks share cuda:0 on a 1-GPU box
    # (the real multi-GPU path is RCCL = backend "nccl", one GPU per rank)
    backend = os.environ.get('JAMIE_DIST_BACKEND', backend)
    if os.environ.get('JAMIE_SHARE_GPU') == '1':
        local = 0
    if not dist.is_initialized():
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend)
    return rank, world, local


def shard_bounds(n_rows, rank, world):
    """Contiguous row shard [lo, hi) of rank `rank`: sizes differ by at most one."""
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class NativeComm:
    """The process group's collectives issued through the C ABI (jamie_allreduce / jamie_reduce_scatter / jamie_all_gather in
    csrc/comm.hip: thin RCCL calls on a communicator of their own, created with an id that rank 0 broadcasts through the
    existing torch.distributed group): ONE foreign call per collective (~5 us of host time, against ~28 us through
    torch.distributed) that a recorded launch plan replays like a kernel launch.  Every collective runs on the communicator's
    own stream behind an event of the issuing stream; `Work.wait()` makes the CURRENT stream wait for it (device side only)."""

    class Work:
        def __init__(self, comm, slot):
            self.comm, self.slot = comm, slot

        def wait(self):
            from . import _native as nv
            nv.comm_wait(self.comm.h, self.slot)

    def __init__(self, group=None, device=None):
        from . import _native as nv
        self.nv = nv
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        dev = device if device is not None else torch.device('cuda', torch.cuda.current_device())
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            uid.copy_(torch.frombuffer(bytearray(nv.comm_unique_id()), dtype=torch.uint8))
        if self.world > 1:
            dist.broadcast(uid, src=0, group=group)
        self.h = nv.comm_create(bytes(uid.cpu().numpy().tobytes()), self.rank, self.world)
        self._slot = 0

    @staticmethod
    def available(group=None):
        """RCCL is the process group's backend (so every rank has a GPU of its own) and the library binds librccl."""
        try:
            if not (dist.is_initialized() and dist.get_backend(group) == 'nccl' and torch.cuda.is_available()):
                return False
            from . import _native as nv
            return nv.comm_version() > 0
        except Exception:       # noqa: BLE001
            return False

    def _next(self):
        self._slot = (self._slot + 1) % 32
        return self._slot

    def all_reduce(self, buf):
        s = self._next()
        self.nv.comm_allreduce(self.h, buf, s)
        return self.Work(self, s)

    def reduce_scatter(self, dst, src):
        s = self._next()
        self.nv.comm_reduce_scatter(self.h, src, dst, s)
        return self.Work(self, s)

    def all_gather(self, out, piece):
        s = self._next()
        self.nv.comm_all_gather(self.h, piece, out, s)
        return self.Work(self, s)

    def close(self):
        if self.h:
            self.nv.comm_destroy(self.h)
            self.h = None


class GradAllReduce:
    """All-reduce (SUM) of the flat gradient buffer in `n_buckets` contiguous chunks issued
    asynchronously; the 1/world average is applied inside the clip+Adam kernel (hyper[grad_scale]).
    With xGMI's point-to-point links a few large messages beat many small ones; the default is one."""

    def __init__(self, group=None, n_buckets=1):
        self.group = group
        self.n_buckets = max(1, int(n_buckets))
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1

    def __call__(self, flat):
        if self.world == 1:
            return
        n = flat.numel()
        if self.n_buckets == 1:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            return
        step = (n + self.n_buckets - 1) // self.n_buckets
        step = (step + 3) // 4 * 4
        works = [dist.all_reduce(flat[s:min(n, s + step)], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for s in range(0, n, step)]
        for w in works:
            w.wait()


class OverlappedGradAllReduce:
    """The same single logical all-reduce, issued region by region while the backward pass is still running:
    `region_done(flat[a:b])` is called right after the kernels that produce that contiguous region were
    launched (RCCL orders the collective after them on the device and runs it on its own stream);
    `finish()` waits for all of them before the gradient norm.  Adjacent regions are merged until a
    bucket reaches `min_bytes`: xGMI is point-to-point, a few large messages beat many small ones.  The default (16 MB) lets
    every large layer's region go out on its own (20 MB as bf16, 40 MB as fp32 at config 2): what stays exposed after the
    backward pass is then the LAST layer's message alone, not a merged pair (reasoned, not measured: no multi-GPU box).

    `comm_dtype=torch.bfloat16` (the default of the bf16 compute mode) exchanges the gradient as bf16: a region is cast
    into a persistent bf16 buffer, all-reduced there and cast back into the fp32 gradient in `finish()`.  At config 2
    the fp32 exchange is 161 MB per step, about as long as the whole 0.93 ms bf16 step on 8 GPUs and mostly exposed
    (the last region is only ready when the backward pass ends); bf16 halves it, at the precision the bf16 GEMMs
    produced the gradient with.  Every rank receives the same reduced values, so the replicas stay identical."""

    def __init__(self, group=None, min_bytes=16 << 20, comm_dtype=None, dry_run_world=0, native=None, single_rank_ok=False):
        """`dry_run_world` = N > 1 (one process, no process group): everything the N-rank step does on the device EXCEPT the
        collectives themselves -- region bookkeeping, message casts, side stream and events -- so that the per-rank compute
        path of the data-parallel step can be timed on a one-GPU box (bench.py --dry-run-world).
        `native=True`: issue the collectives through the C ABI (NativeComm: jamie_allreduce & co.) instead of
        torch.distributed.  OFF by default: measured on a one-rank RCCL group (tools/bench_sharded_host_time.py,
        profiles/r04_host_time_native_vs_torch.log) the host pays the same ~28 us per collective either way -- it is RCCL's own
        enqueue, not torch.distributed's wrapper -- and the three extra event calls per collective make the native path slower
        (replicated exchange 276 against 214 us of host time per step, sharded 574 against 575).  `single_rank_ok`: run the exchange in a ONE-rank process group too (every collective is then a copy on
        the backend's stream): the one way to put this code path through RCCL itself on a one-GPU box."""
        self.group = group
        self.single = bool(single_rank_ok) and dist.is_initialized() and dist.get_world_size(group) == 1
        self.native = None
        if native:
            self.native = NativeComm(group)
        self.min_bytes = min_bytes
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dry = int(dry_run_world) > 1 and self.world == 1
        if self.dry:
            self.world = int(dry_run_world)
        self.comm_dtype = None if comm_dtype in (None, torch.float32) else comm_dtype
        self.comm = None             # persistent low-precision exchange buffer (same offsets as the flat gradient)
        self.works = []
        self.stream = None           # side stream of the cast + collective (low-precision messages on the GPU)
        self.pending = None          # (flat, lo, hi, precast) not yet issued
        self._msgs, self._last_msgs = [], []      # element counts of the messages of the current / the last finished step
        self._casts = {}             # (flat ptr, message-buffer ptr, lo, hi) -> prepared fp32 -> bf16 cast launch of that region
        self.trace = None            # enable_trace(): [(kind, bytes, HIP event)] of one step (bench.py's exposure model)
        self.exposure = None         # enable_exposure(): [(event, event)] around finish()'s device-side waits
        self._exp_every, self._exp_count = 1, 0

    def enable_trace(self, on=True):
        """Record a HIP event on the launch stream wherever a message is issued and where `finish()` starts waiting: the
        timeline bench.py's `dp_model` block prices the exchange against (dry runs and real ones alike)."""
        self.trace = [] if on else None

    def enable_exposure(self, on=True, every=1):
        """Bracket the device-side waits of `finish()` with HIP events on the launch stream (every `every`-th step): the time the
        step's stream stands still waiting for gradient messages that are still on the wire -- the EXPOSED part of the exchange,
        measured, where bench.py's `dp_model` only models it.  `exposed_us()` reads them back."""
        self.exposure = [] if on else None
        self._exp_every, self._exp_count = max(1, int(every)), 0

    def exposed_us(self):
        """{'median', 'mean', 'max', 'n'} of the bracketed waits in microseconds (device sync), or None."""
        if not self.exposure:
            return None
        torch.cuda.synchronize()
        t = sorted(1e3 * a.elapsed_time(b) for a, b in self.exposure)
        return {'median': t[len(t) // 2], 'mean': sum(t) / len(t), 'max': t[-1], 'n': len(t)}

    def message_elements(self):
        """Element counts of the messages the last step issued, in issue order (bench.py's all-reduce probe)."""
        return list(self._last_msgs)

    def message_buffer(self, flat):
        """The persistent low-precision exchange buffer (same offsets as the flat gradient): a producer that writes its
        gradients into it directly (the bf16 dW epilogues, TrainEngine) announces its regions with `precast=True`."""
        if self.comm is None or self.comm.numel() != flat.numel() or self.comm.device != flat.device:
            self.comm = torch.zeros(flat.numel(), dtype=self.comm_dtype, device=flat.device)
            self._casts.clear()          # (the prepared cast launches hold raw pointers into the old buffer)
        return self.comm

    def region_done(self, flat, lo, hi, precast=False, force=False):
        """`precast`: the region's gradients are already in `message_buffer()` (every region of a step alike): no cast pass,
        no side stream -- the collective is issued where it stands (RCCL orders it behind the launches before it).
        `force`: issue the message now, whatever its size (the first part of the LAST layer's gradient: what is still on the
        wire when the backward pass ends is then the small second part only)."""
        if self.world == 1 and not self.single:
            return
        precast = bool(precast)
        pend = self.pending
        if pend is not None and (pend[0] is not flat or pend[3] != precast):      # another buffer / another path: no merging
            self._issue(*pend)
            pend = None
        if pend is not None and pend[2] == lo:       # forward-adjacent
            lo = pend[1]
        elif pend is not None and pend[1] == hi:     # backward-adjacent (the usual case)
            hi = pend[2]
        elif pend is not None:
            self._issue(*pend)
        self.pending = (flat, lo, hi, precast)
        if force or (hi - lo) * (flat.element_size() if self.comm_dtype is None else 2) >= self.min_bytes:
            self._issue(*self.pending)
            self.pending = None

    def _collective(self, buf, lo, hi):
        """The collective of one message (`buf` = the message bytes of gradient range [lo, hi)); None in a dry run."""
        if self.dry:
            return None
        if self.native is not None and buf.is_cuda:
            return self.native.all_reduce(buf)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def _issue(self, flat, lo, hi, precast=False):
        self._msgs.append(int(hi - lo))
        if self.trace is not None and flat.is_cuda:
            from . import _native as nv
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(nv.current_stream())
            self.trace.append((self._kind(lo, hi), (hi - lo) * (flat.element_size() if self.comm_dtype is None else 2), ev))
        if self.comm_dtype is None or precast:
            buf = (flat if self.comm_dtype is None else self.message_buffer(flat))[lo:hi]
            self.works.append((self._collective(buf, lo, hi), None, None, None))
            return
        buf = self.message_buffer(flat)[lo:hi]
        if flat.is_cuda:
            # the cast into the message buffer (a pass over the region: 45 us per step at config 2) and the collective
            # go to a side stream behind an event, so the backward kernels that follow on the main stream do not queue
            # behind the cast
            if self.stream is None:
                self.stream = torch.cuda.Stream(device=flat.device)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            key = (flat.data_ptr(), buf.data_ptr(), lo, hi)
            cast = self._casts.get(key)
            if cast is None:
                from . import _native as nv
                cast = self._casts[key] = nv.FlatCast(flat[lo:hi], buf)
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                cast.run(self.stream)                     # fp32 region -> bf16 message buffer (HIP launch, no ATen op)
                work = self._collective(buf, lo, hi)
        else:
            buf.copy_(flat[lo:hi])
            work = self._collective(buf, lo, hi)
        self.works.append((work, flat, lo, hi))

    def _kind(self, lo, hi):
        return 'message'

    def _copies_back(self, lo, hi):
        return True

    def finish(self, copy_back=True):
        """Wait for every message.  With low-precision messages the reduced gradient is cast back into the fp32 buffer,
        unless `copy_back=False`: then it stays in `self.comm` (same offsets) and the caller's norm / Adam kernels read
        it from there (jamie_grad_sqnorm_bf16, jamie_clip_adam_g16), which saves the cast-back pass."""
        if self.pending is not None:
            self._issue(*self.pending)
            self.pending = None
        if self.trace is not None and torch.cuda.is_available() and self.works:
            from . import _native as nv
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(nv.current_stream())
            self.trace.append(('finish', 0, ev))
        e0 = None
        if self.exposure is not None and self.works and torch.cuda.is_available():
            self._exp_count += 1
            if self._exp_count % self._exp_every == 0:
                from . import _native as nv
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record(nv.current_stream())
        for w, flat, lo, hi in self.works:
            if w is not None:
                w.wait()
            elif self.stream is not None:       # (dry run: what work.wait() does for a device collective)
                torch.cuda.current_stream().wait_stream(self.stream)
            if flat is not None and copy_back and self._copies_back(lo, hi):
                flat[lo:hi].copy_(self.comm[lo:hi])
        if e0 is not None:
            from . import _native as nv
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record(nv.current_stream())
            self.exposure.append((e0, e1))
        self.works = []
        self._last_msgs, self._msgs = self._msgs, []

    def __call__(self, flat):          # non-overlapped use
        self.region_done(flat, 0, flat.numel())
        self.finish()


class ShardedGradExchange(OverlappedGradAllReduce):
    """Data parallel with a SHARDED optimiser (the ZeRO stage-1 arrangement): clip + Adam is the one kernel of the step that is
    bound by HBM bytes per PARAMETER (28 B each, 29 % of the one-GPU step at config 2), and with replicated parameters every
    rank repeats it in full.  Here each of the four large weight regions (model.ParamLayout: 99 % of the parameters) is cut into
    `world` equal pieces and rank r owns piece r of every region -- its fp32 master weights and Adam moments live in packed
    shard buffers (TrainEngine.enable_sharded_optimizer) -- so that

      * the gradient of a large region is REDUCE-SCATTERED as the backward pass completes it (half the wire time of the
        all-reduce it replaces; rank r receives the sum of piece r into its packed gradient shard),
      * the small region `rep` (biases, BatchNorm affine pairs, sigma, the skinny head / latent layers) is all-reduced and
        updated by every rank, as before,
      * the squared norm of the reduced gradient = sum over ranks of (own pieces) + (rep): one all-reduce of the partial sums,
      * clip + Adam runs over the packed shard: 1 / world of the bytes,
      * the updated weights (bf16 copy in bf16 compute mode, fp32 otherwise) are ALL-GATHERED region by region in FORWARD
        order while the next step's sampler, gather and first layers already run; a layer's product waits for its own region.

    The wire carries the same bytes per step as the all-reduce (reduce-scatter + all-gather = all-reduce), but the second half
    moves from the end of the backward pass, where nothing can overlap it, to the next forward pass.  Every rank applies the
    same update to `rep` and receives the same gathered bytes, so the replicas stay bit-identical.  The fp32 master copy of
    the large regions is only current on its owner in bf16 mode: TrainEngine.gather_sharded_state() before anything other
    than the next training step reads `model.flat` (evaluation, checkpoints)."""

    def __init__(self, group=None, comm_dtype=None, dry_run_world=0, dry_run_rank=0, single_rank_ok=False, native=None):
        """`single_rank_ok`: run the whole exchange in a one-rank process group as well (every collective is then a copy on the
        backend's own stream): the one way to put this code path through RCCL itself on a one-GPU box
        (tests/test_hip_distributed.py::test_sharded_exchange_on_rccl_with_one_rank)."""
        super().__init__(group, min_bytes=1 << 62, comm_dtype=comm_dtype, dry_run_world=dry_run_world, native=native,
                         single_rank_ok=single_rank_ok)
        self.rank = int(dry_run_rank) if self.dry or not dist.is_initialized() else dist.get_rank(group)
        self.spans = {}              # (lo, hi) of a sharded region -> this rank's packed gradient piece (length (hi - lo) / world)
        self.gathers = {}            # layer name -> pending all-gather of its updated weights

    def set_shards(self, spans):
        """`spans` = [(lo, hi, dst)]: gradient range [lo, hi) is reduce-scattered, this rank's piece lands in `dst`."""
        for lo, hi, dst in spans:
            if (hi - lo) % self.world or dst.numel() != (hi - lo) // self.world:
                raise ValueError(f'sharded region [{lo}, {hi}) does not split into {self.world} pieces of {dst.numel()}')
        self.spans = {(int(lo), int(hi)): dst for lo, hi, dst in spans}

    def region_done(self, flat, lo, hi, precast=False, force=False):
        if self.world == 1 and not self.single:
            return
        if (lo, hi) in self.spans:               # a sharded region is one message, never merged with its neighbours
            if self.pending is not None:
                self._issue(*self.pending)
                self.pending = None
            self._issue(flat, lo, hi, bool(precast))
            return
        if self.single:                          # (the base class drops everything in a one-rank world)
            if self.pending is not None:
                self._issue(*self.pending)
            self.pending = (flat, lo, hi, bool(precast))
            return
        super().region_done(flat, lo, hi, precast, force)

    def _kind(self, lo, hi):
        return 'reduce_scatter' if (lo, hi) in self.spans else 'message'

    def _copies_back(self, lo, hi):
        return (lo, hi) not in self.spans

    def _collective(self, buf, lo, hi):
        dst = self.spans.get((lo, hi))
        if dst is None:
            return super()._collective(buf, lo, hi)
        if self.dry:      # (what the device does besides the wire: this rank's piece arrives in the packed shard)
            s = dst.numel()
            dst.copy_(buf[self.rank * s:(self.rank + 1) * s])
            return None
        if self.native is not None and buf.is_cuda:
            return self.native.reduce_scatter(dst, buf)
        return dist.reduce_scatter_tensor(dst, buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def sum_partials(self, t):
        """In-place sum over the ranks of the partial sums of squares of the pieces each rank owns."""
        if not self.dry and (self.world > 1 or self.single):
            if self.native is not None and t.is_cuda:
                self.native.all_reduce(t).wait()
                return
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def gather(self, name, out, piece):
        """Start the all-gather of region `name`: `piece` (this rank's updated piece) -> `out` (the whole region)."""
        if self.trace is not None and out.is_cuda:
            from . import _native as nv
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(nv.current_stream())
            self.trace.append(('all_gather', out.numel() * out.element_size(), ev))
        if self.dry:
            s = piece.numel()
            out[self.rank * s:(self.rank + 1) * s].copy_(piece)
            return
        if self.native is not None and out.is_cuda:
            self.gathers[name] = self.native.all_gather(out, piece)
            return
        self.gathers[name] = dist.all_gather_into_tensor(out, piece, group=self.group, async_op=True)

    def wait_gather(self, name):
        if self.trace is not None and torch.cuda.is_available():
            from . import _native as nv
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(nv.current_stream())
            self.trace.append(('wait:' + name, 0, ev))
        w = self.gathers.pop(name, None)
        if w is not None:
            w.wait()

    def wait_all_gathers(self):
        for name in list(self.gathers):
            self.wait_gather(name)


def broadcast_flat(flat, src=0, group=None):
    """Make every rank start from rank `src`'s parameters."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat, src=src, group=group)


def average_(t, group=None):
    """In-place mean over the ranks (BatchNorm running statistics before evaluation / checkpoints)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
        t.div_(dist.get_world_size(group))
    return t


def mean_scalar(value, device, group=None):
    """Mean of a host scalar over the ranks (rank-invariant early-stop decision)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return float(t.item()) / dist.get_world_size(group)
