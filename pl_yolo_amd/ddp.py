"""Data-parallel training over the GPUs of one node: one process per GPU, identical weights, rank-local BatchNorm
statistics and rank-local num_fg normalisation (the reference has no SyncBN and no cross-rank num_fg reduction,
yolox_loss.py:148-154), and exactly one exchange per iteration: the MEAN of the flat fp32 gradient buffer, with RCCL
over xGMI (torch.distributed backend "nccl" is RCCL on ROCm; "gloo" is used by the CPU tests).

The exchange is bucketed and overlapped with the backward plan:
  * the flat gradient buffer holds the live parameters in `model.parameters()` order (dead Bottleneck.bn pairs behind
    them, never exchanged); it is cut from the END -- the head, whose gradients are complete first -- into buckets of
    ~25 MB (PLYOLO_BUCKET_MB); optionally (PLYOLO_BUCKET_TAIL_MB > 0, default 0 = off) one small bucket for the first backbone
    layers, the last to finish.  xGMI is point-to-point: few large collectives beat many small ones;
  * a bucket is READY when the backward of every layer owning one of its parameters has been recorded.  At that point
    the backward plan hands the weight-gradient work queued so far to its lane, unpacks the finished weight-gradient
    slabs into the flat buffer there, and records a HOST HOOK on a communication lane that waits for both lanes
    (plyolo_plan_hook).  When the eager replay reaches the hook -- the host runs far ahead of the GPU -- the callback
    enqueues the bucket's all-reduce on that lane's stream; it executes while the data-gradient chain of the layers
    upstream is still running, and the plan's final join waits for all of it.
Single bucket / no overlap remains available as GradAllReduce.all_reduce_ (and is what the bucketed path must equal bit
for bit: tests/test_ddp_cpu.py)."""
import os
import warnings

import torch
import torch.distributed as dist

from . import graph as G
from ._lib import call, PlyoloError

FORCE_COLLECTIVE = False   # self-test: issue the collectives even in a one-rank group (bench.py PLYOLO_BENCH_FORCE_DDP)
DEFER = os.environ.get("PLYOLO_DDP_DEFER", "1") == "1"   # 1: collectives are started from the hooks and awaited once, after the plan; 0: the hook's lane waits for each
TIME_EXPOSED = False        # bench.py: bracket wait_all() with events -> the part of the exchange the step could not hide
COMM_LANE = int(os.environ.get("PLYOLO_COMM_LANE", "1"))   # plan lane of the exchange (0 main, 1 weight gradients, 2.. head levels); 1 = on the weight-gradient lane itself


def bucket_bytes():
    return int(float(os.environ.get("PLYOLO_BUCKET_MB", "25")) * 1e6)


def plan_buckets(offs, sizes, ready, n_live, target_bytes, tail_bytes=0):
    """Cut the live range [0, n_live) of the flat gradient buffer into buckets, last parameters first.

    offs / sizes: element offset and (aligned) element count of every live parameter in flat order; ready[i]: forward
    index of the layer that owns parameter i (its gradient is final once the backward has passed that layer), None for
    a parameter no layer uses.  Returns [(start, end, ready)] in emission order (end of the buffer first), `ready`
    non-increasing -- bucket k never leaves before bucket k-1.

    tail_bytes > 0: the FRONT of the buffer -- the first layers of the backbone, whose gradients are final only when the
    backward ends -- gets a bucket of its own of about that size.  Those layers (large maps, a few thousand parameters) take a
    large share of the backward's time and hold almost none of the gradient bytes: the exposed collective at the end of the
    step shrinks from "whatever was left over" (11 MB for YOLOX-s at 25 MB buckets) to that tail, and the bucket before it
    leaves while the large-map layers are still running."""
    cuts = []                                   # parameter indices where a bucket starts
    acc = 0
    front = 0
    if tail_bytes > 0:
        t = 0
        for i in range(len(offs)):
            t += sizes[i] * 4
            if t >= tail_bytes:
                front = i + 1                   # parameters [0, front) form the tail bucket
                break
        if front >= len(offs):
            front = 0                           # the whole buffer is smaller than the tail: no split
    for i in range(len(offs) - 1, -1, -1):
        acc += sizes[i] * 4
        if acc >= target_bytes or i == 0 or i == front:
            cuts.append(i)
            acc = 0
    buckets, end = [], n_live
    for i in cuts:
        hi = len(offs) if end == n_live else next(k for k in range(len(offs)) if offs[k] == end)
        rs = [ready[k] for k in range(i, hi) if ready[k] is not None]
        buckets.append([offs[i], end, min(rs) if rs else None])
        end = offs[i]
    inf = max([b[2] for b in buckets if b[2] is not None] + [0])
    prev = inf
    for b in buckets:   # unused-only buckets are ready from the start; enforce emission order
        b[2] = prev if b[2] is None else min(b[2], prev)
        prev = b[2]
    return [tuple(b) for b in buckets]


def tail_bytes():
    return int(float(os.environ.get("PLYOLO_BUCKET_TAIL_MB", "0")) * 1e6)


class GradAllReduce:
    """Averages (ranges of) a flat gradient buffer across ranks."""

    def __init__(self, group=None):
        self.group = group
        self.world = dist.get_world_size(group)

    def __deepcopy__(self, memo):
        """ModelEMA deep-copies the model (ema.py:41) and with it this object: the copy shares the process group (a ProcessGroup can
        neither be copied nor pickled)."""
        return GradAllReduce(self.group)

    def active(self):
        return self.world > 1 or FORCE_COLLECTIVE

    def all_reduce_(self, flat, start=0, end=None, defer=False):
        """In-place mean of flat[start:end] over the ranks, ordered after the work already queued on the current stream.
        defer=False: the current stream also waits for the result (returns flat).  defer=True: the collective is only
        STARTED (it runs on the process group's own stream); returns finish(), to be called once with the stream that
        consumes the gradients current -- nothing else on the launching stream waits for the exchange."""
        if not self.active():
            return (lambda: None) if defer else flat
        view = flat[start:end if end is not None else flat.numel()]
        if dist.get_backend(self.group) == "nccl":   # RCCL averages inside the collective: no second pass over the buffer
            work = dist.all_reduce(view, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
            if defer:
                return work.wait                      # a stream-side wait, the host does not block
            work.wait()
            return flat
        # gloo (CPU tests, two processes on one GPU) has no AVG
        if defer:
            work = dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

            def finish():
                work.wait()
                view.mul_(1.0 / self.world)
            return finish
        dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)
        view.mul_(1.0 / self.world)
        return flat

    def all_reduce_buckets_(self, flat, buckets):
        for (a, b, _) in buckets:
            self.all_reduce_(flat, a, b)
        return flat


def _op_params(op):
    """Every parameter whose gradient the backward of launch-plan op `op` produces."""
    ps = []
    for attr in ("pc", "pc_cls", "pc_ro"):
        pc = getattr(op, attr, None)
        if pc is not None:
            for (w, b, _) in pc.sources:
                ps += [w, b]
    for attr in ("bn", "bn_a", "bn_b"):
        bn = getattr(op, attr, None)
        if bn is not None:
            ps += [bn.weight, bn.bias]
    if isinstance(op, G.DwConvUnitOp):
        ps.append(op.w)
    if isinstance(op, G.ImplicitHeadOp):
        ps += [op.conv.weight, op.conv.bias, op.ia, op.im]
    return [p for p in ps if p is not None]


class BucketSchedule:
    """Records the bucketed exchange into one backward plan (see the module docstring)."""

    def __init__(self, runner, session, g):
        self.runner, self.s, self.g, self.ddp = runner, session, g, runner.ddp
        flat = runner.flat
        # graph.record_ops() reports i once EVERY op with a forward index >= i has recorded its backward (whatever lane
        # it sits on, whatever order the lanes were issued in), so a parameter is final at its owner's own index
        def owner_index(op):
            return op.index
        ready_of = {}
        for op in g.ops:
            for p in _op_params(op):
                i = owner_index(op)
                ready_of[id(p)] = min(ready_of.get(id(p), i), i)
        live = [(p, o) for p, o in zip(flat["params"], flat["offs"]) if o < flat["n_live"]]
        offs = [o for _, o in live]
        sizes = [(offs[i + 1] if i + 1 < len(offs) else flat["n_live"]) - offs[i] for i in range(len(offs))]
        ready = [ready_of.get(id(p)) for p, _ in live]
        self.buckets = plan_buckets(offs, sizes, ready, flat["n_live"], bucket_bytes(), tail_bytes())
        self.next_bucket = 0
        # weight-gradient slabs are unpacked (and the ImplicitHead extras run) as soon as their layer is done
        self.entry_op = []
        op_of_pc = {}
        for op in g.ops:
            for attr in ("pc", "pc_cls", "pc_ro"):
                pc = getattr(op, attr, None)
                if pc is not None:
                    op_of_pc[id(pc)] = owner_index(op)
        ei = 0
        for c in g.convs:
            for _ in c.sources:
                self.entry_op.append(op_of_pc.get(id(c), 0))
                ei += 1
        assert ei == len(g.pack_entries)
        self.entries_left = list(range(len(g.pack_entries)))
        self.post_left = list(g.post_unpack)
        self.errors = []
        self.pending = []
        self.exposed = []       # (start, end) event pairs of wait_all(), when TIME_EXPOSED

    # ---- recording
    def after(self, done_idx):
        """Called by graph.record_ops once every op with forward index >= done_idx has recorded its backward."""
        ready = []
        while self.next_bucket < len(self.buckets) and self.buckets[self.next_bucket][2] >= done_idx:
            ready.append(self.next_bucket)
            self.next_bucket += 1
        if ready:
            self._emit(done_idx, ready)

    def finish(self):
        self.after(0)
        if self.entries_left or self.post_left:      # layers whose parameters sit in no bucket (cannot happen; be safe)
            self._emit(0, [])
        assert self.next_bucket == len(self.buckets)

    def _emit(self, done_idx, ready):
        g, plan = self.g, self.g.plan
        wl = G.WGRAD_LANE if g.use_lanes else 0
        g.flush_param_grads()
        plan.lane(wl)
        g.flush_reduce()                 # queued slab folds first: the unpack below reads slab 0
        todo = [i for i in self.entries_left if self.entry_op[i] >= done_idx]
        if todo:
            self.entries_left = [i for i in self.entries_left if self.entry_op[i] < done_idx]
            t, nblk = g.pack_subtable(todo)
            call("plyolo_unpack_wgrads_flat", t.data_ptr(), len(todo), nblk, 0, None)
        for op in [o for o in self.post_left if o.index >= done_idx]:
            op.post_unpack()
            self.post_left.remove(op)
        dbg = int(os.environ.get("PLYOLO_DDP_DBG", "0"))   # diagnostics: 1 no comm lane at all, 2 comm lane waits only for the wgrad lane
        if ready and dbg != 1:
            evs = [plan.record(wl)]
            if wl != 0 and dbg != 2:
                evs.append(plan.record(0))       # BatchNorm / bias gradients are written on the compute lanes
                for l in g.side_lanes:
                    if l != wl:
                        evs.append(plan.record(l))
            plan.lane(COMM_LANE)
            for ev in evs:
                plan.wait(COMM_LANE, ev)
            for k in ready:
                plan.hook(COMM_LANE, k)
        plan.lane(0)

    # ---- replay
    def install(self, plan):
        self._streams = {}
        plan.set_hook(self._exchange)
        self.plan = plan

    def _exchange(self, k, stream_ptr):
        """Host hook of bucket k (called by the replay when it reaches the hook, the GPU far behind): START the collective,
        ordered after the lane's work so far.  Nothing in the plan waits for it -- the buckets are disjoint ranges of the flat
        buffer and no later launch touches a bucket that has left -- the consumer's stream does, in wait_all()."""
        a, b, _ = self.buckets[k]
        flat = self.runner.flat["g"]
        if flat.is_cuda:
            st = self._streams.get(stream_ptr)
            if st is None:
                st = self._streams[stream_ptr] = torch.cuda.ExternalStream(stream_ptr)
            with torch.cuda.stream(st):
                self.pending.append(self.ddp.all_reduce_(flat, a, b, defer=DEFER))
        else:
            self.pending.append(self.ddp.all_reduce_(flat, a, b, defer=DEFER))

    def wait_all(self):
        """After the backward plan has been issued: the caller's current stream waits for every started collective."""
        pend, self.pending = self.pending, []
        timed = TIME_EXPOSED and torch.cuda.is_available() and self.runner.flat["g"].is_cuda
        if timed:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()          # reached when the backward plan has joined the caller's stream
        for fin in pend:
            if callable(fin):
                fin()
        if timed:
            ev1.record()          # reached when the last collective has finished: ev0 -> ev1 = exposed communication
            self.exposed.append((ev0, ev1))

    def check(self):
        e = getattr(self.plan, "hook_error", None)
        if e is not None:
            self.plan.hook_error = None
            raise PlyoloError("gradient exchange failed inside the backward plan: %r" % (e,))


def _check_hw_queues():
    """The plans run three launch lanes (+ RCCL's own stream) on the runtime's hardware queues.  Measured on MI355X / ROCm 7.2
    (INTEGRATION.md): GPU_MAX_HW_QUEUES = 4 (the runtime's default) is the optimum, 3: +3.5 %, 2: +22 %, and FIVE OR
    MORE HALVE the throughput (9.26 -> 16.9 ms per step).  A launch script that exports a larger value -- common advice for
    multi-stream RCCL jobs -- would cost every rank a factor of two: say so once, loudly, where the data-parallel run starts."""
    v = os.environ.get("GPU_MAX_HW_QUEUES")
    if v is None:
        return
    try:
        n = int(v)
    except ValueError:
        return
    if n != 4:
        warnings.warn("GPU_MAX_HW_QUEUES=%d: pl_yolo_amd's launch plans are tuned for the runtime's default of 4 hardware queues "
                      "(measured: 3 queues +3.5 %% step time, 2 +22 %%, 5 or more about 2x slower); unset it or set it to 4" % n,
                      RuntimeWarning, stacklevel=3)


def attach(model, group=None):
    """Make `model` (a pl_yolo_amd OneStageD) average its gradients across ranks inside every backward, and start from
    rank 0's weights."""
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised")
    _check_hw_queues()
    r = model.runner()
    dev = next(model.parameters()).device
    if not r._adopted_ok(dev):
        r.adopt(dev)            # parameters / buffers become views of three flat buffers ...
    for key in ("w", "fbuf", "ibuf"):
        dist.broadcast(r.flat[key], src=0, group=group)   # ... so the start state is three collectives, not one per tensor
    model.__dict__['_ddp'] = GradAllReduce(group)         # kept on the model: a runner rebuilt later (compute_dtype change) re-attaches it
    r.ddp = model.__dict__['_ddp']
    r.sessions = {}                                       # backward plans recorded before attach() have no exchange
    return model
