"""Runs a detector (backbone -> neck -> head -> loss) through recorded HIP launch plans.

One `DetectorRunner` per OneStageD.  It
  * adopts the module's parameters/buffers into flat fp32 device buffers (the
    nn.Parameters become views, so state_dict / optimizers / EMA keep working, while
    gradients land in one contiguous buffer = one RCCL all-reduce bucket);
  * traces the module tree once per (batch, image size, label rows, mode) into a
    `Graph`, allocates every activation / gradient / workspace buffer up front, and
    records the forward and backward launch plans;
  * replays the plans (optionally as captured hipGraphs) for each step.
"""
import ctypes as C
import os

import torch

from . import _lib, graph as G
from ._lib import BF16, F32, call, PlyoloError


def _align(n, a=64):
    return (n + a - 1) // a * a


class _Session:
    """Everything that belongs to one traced shape/mode."""
    pass


class DetectorRunner:
    def __init__(self, model, dtype="bf16"):
        self.model_ref = model
        self.dtype = BF16 if dtype in ("bf16", BF16) else F32
        self.sessions = {}
        self.flat = None
        # PLYOLO_GRAPH: "1" every plan as a hipGraph, "0" every plan eagerly, "auto" (default) single-lane
        # plans as hipGraphs (smallest launch gaps) and multi-lane plans eagerly on their streams
        mode = os.environ.get("PLYOLO_GRAPH", "auto")
        self.use_graph = {"1": True, "0": False}.get(mode, "auto")
        self._side = None  # hipGraph capture/replay needs a non-default stream
        self.ddp = None  # set by pl_yolo_amd.ddp.attach()

    def __deepcopy__(self, memo):  # ModelEMA deep-copies the module (ema.py:41)
        return None

    # ------------------------------------------------------------ flat storage
    def _adopted_ok(self, device):
        if self.flat is None or self.flat["device"] != device:
            return False
        f = self.flat
        for p, off in ((f["params"][0], 0), (f["params"][-1], f["offs"][-1])):
            if p.data_ptr() != f["w"].data_ptr() + off * 4:
                return False
        return True

    def adopt(self, device):
        """Move parameters / float buffers / counters into flat device buffers and
        re-point the module's tensors at views of them."""
        model = self.model_ref
        params = list(model.parameters())
        if not params:
            raise PlyoloError("model has no parameters")
        # parameters no launch plan ever touches (Bottleneck.bn) go behind the live ones: [0, n_live) is what a
        # data-parallel step exchanges
        dead = {id(p) for m in model.modules() if hasattr(m, "dead_parameters") for p in m.dead_parameters()}
        params = [p for p in params if id(p) not in dead] + [p for p in params if id(p) in dead]
        offs, n, n_live = [], 0, 0
        for p in params:
            offs.append(n)
            n += _align(p.numel())
            if id(p) not in dead:
                n_live = n
        w = torch.zeros(n, dtype=torch.float32, device=device)
        gbuf = torch.zeros(n, dtype=torch.float32, device=device)
        with torch.no_grad():
            for p, o in zip(params, offs):
                if p.dtype != torch.float32:
                    raise PlyoloError("parameters must be fp32 master weights")
                w[o:o + p.numel()].copy_(p.data.reshape(-1))
                p.data = w[o:o + p.numel()].view(p.shape)
        fb = [b for b in model.buffers() if b.dtype.is_floating_point]
        ib = [b for b in model.buffers() if not b.dtype.is_floating_point]
        boffs, nb = [], 0
        for b in fb:
            boffs.append(nb)
            nb += _align(b.numel())
        bw = torch.zeros(max(nb, 8), dtype=torch.float32, device=device)
        iw = torch.zeros(max(len(ib), 1), dtype=torch.int64, device=device)
        with torch.no_grad():
            for b, o in zip(fb, boffs):
                bw[o:o + b.numel()].copy_(b.data.reshape(-1))
                b.data = bw[o:o + b.numel()].view(b.shape)
            for i, b in enumerate(ib):
                iw[i:i + 1].copy_(b.data.reshape(-1).to(torch.int64))
                b.data = iw[i:i + 1].view(b.shape)
        self.flat = dict(device=device, params=params, offs=offs, w=w, g=gbuf, n=n, n_live=n_live, fbuf=bw, ibuf=iw,
                         off_of={id(p): o for p, o in zip(params, offs)})
        self.sessions = {}

    def grad_ptr_of(self, p):
        if p is None or not p.requires_grad:
            return None
        return self.flat["g"].data_ptr() + self.flat["off_of"][id(p)] * 4

    def grad_view(self, p):
        o = self.flat["off_of"][id(p)]
        return self.flat["g"][o:o + p.numel()].view(p.shape)

    # ------------------------------------------------------------------ tracing
    def _session(self, B, H, W, M, mode, device):
        if not self._adopted_ok(device):
            self.adopt(device)
        # use_l1 is baked into the recorded loss launches: flipping it on a live model (YOLOX's last epochs) traces a new session
        key = (B, H, W, M, mode, self.dtype, self.model_ref.training, bool(getattr(self.model_ref.loss, "use_l1", False)))
        s = self.sessions.pop(key, None)
        if s is None:
            before = torch.cuda.memory_allocated(device)
            s = self._build(B, H, W, M, mode, device)
            s.bytes = max(torch.cuda.memory_allocated(device) - before, 0)
            self._evict(device, s.bytes)
        self.sessions[key] = s          # dicts keep insertion order: the most recently used session is last
        return s

    def _evict(self, device, incoming):
        """Multi-scale training (the reference resizes its batches every few iterations) traces one session -- activations,
        gradients, weight-gradient slabs, plans -- per input shape.  They stay resident while they fit: when the sessions would
        hold more than PLYOLO_SESSION_BUDGET (fraction of the device memory, default 0.5) the least recently used ones are
        dropped (an autograd node that still refers to one keeps it alive until its backward has run)."""
        frac = float(os.environ.get("PLYOLO_SESSION_BUDGET", "0.5"))
        budget = frac * torch.cuda.get_device_properties(device).total_memory
        held = incoming + sum(getattr(v, "bytes", 0) for v in self.sessions.values())
        while held > budget and self.sessions:
            k = next(iter(self.sessions))
            held -= getattr(self.sessions.pop(k), "bytes", 0)

    def _build(self, B, H, W, M, mode, device):
        model = self.model_ref
        if H % 32 or W % 32:
            raise PlyoloError("input size must be a multiple of 32 (got %dx%d)" % (H, W))
        training = model.training
        g = G.Graph(self.dtype, training, device)
        g.grad_ptr_of = self.grad_ptr_of
        s = _Session()
        s.g, s.mode, s.B, s.H, s.W, s.M = g, mode, B, H, W, M
        s.generation = 0          # bumped by every forward of this session (see _check_generation)
        s.stem_kind = getattr(model.backbone, "stem_kind", "focus")
        if s.stem_kind == "focus":   # CSPDarkNet: space-to-depth gather (12 real channels)
            image = g.new_act(B, H // 2, W // 2, 16 if self.dtype == BF16 else 12, "focus")
        else:                         # EELAN: plain RGB image, padded to a 16-byte channel vector
            image = g.new_act(B, H, W, 8 if self.dtype == BF16 else 4, "rgb")
        s.image = image
        feats = model.backbone.emit(g, image)
        if model.neck is not None:
            feats = model.neck.emit(g, feats)
        if not isinstance(feats, (list, tuple)):
            feats = [feats]
        nc = model.head.num_classes
        strides = list(model.loss.strides) if model.loss is not None else [W // f.W for f in feats]
        if len(strides) != len(feats):
            raise PlyoloError("loss.stride has %d entries for %d feature levels" % (len(strides), len(feats)))
        if model.head.n_anchors == 1:
            head = G.HeadBuffers(g, B, nc, [(f.H, f.W) for f in feats], strides, max(M, 1))
        else:
            anchors = getattr(model.loss, "anchors_list", None)
            if anchors is None:
                raise PlyoloError("anchor-based head needs a loss plugin with anchors (yolov7)")
            head = G.V7HeadBuffers(g, B, nc, model.head.n_anchors, [(f.H, f.W) for f in feats], strides, anchors)
        s.head = head
        model.head.emit(g, feats, head)
        if mode == "train":
            if model.head.n_anchors == 1:
                head.alloc_loss()
            else:
                head.alloc_loss(M)
            model.loss.emit(g, head, True)
        elif mode == "eval":
            head.alloc_eval()
            model.loss.emit(g, head, False)
        elif mode == "maps_grad":
            head.alloc_grad_only()
        g.allocate()
        g.build_pack_table(self.grad_ptr_of)
        # ---- record the forward plan
        s.fwd = G.Plan()
        s.fwd.is_fwd = True
        with s.fwd:
            g.plan = s.fwd
            g.pack_weights()
            g.zero_fwd_stats()
            # PLYOLO_FWD_LANES=0 records the forward on one lane, which then replays as ONE hipGraph: measured 2 %
            # slower than the eager replay with the head levels side by side (a dependent launch costs ~1.8 us in a
            # graph against ~5 us eagerly for EMPTY kernels, tools/micro/graph_floor.hip, but behind real kernels the
            # command processor has the next dispatch ready either way)
            G.record_ops(g, s.fwd, g.ops, "fwd", lanes=os.environ.get("PLYOLO_FWD_LANES", "1") != "0")
        s.bwd = None
        s.used_params = []
        if mode in ("train", "maps_grad"):
            s.bwd = G.Plan()
            with s.bwd:
                g.plan = s.bwd
                if g.dtype != BF16:  # the fp32 parity wgrad accumulates with atomics; the MFMA path overwrites its slabs
                    call("plyolo_memset_async", g.dwp_arena.data_ptr(), 0, g.dwp_arena.numel() * 4, None)
                g.zero_bwd_stats()
                s.sched = None
                if self.ddp is not None and self.ddp.active():
                    # data parallel: the gradient buckets leave through host hooks of the backward plan as soon as the
                    # layers that fill them are done (pl_yolo_amd/ddp.py: BucketSchedule)
                    from . import ddp as D
                    s.sched = D.BucketSchedule(self, s, g)
                    g.plan_bn_red()
                    G.record_ops(g, s.bwd, list(reversed(g.ops)), "bwd", after=s.sched.after)
                    g.check_bn_red()
                    s.sched.finish()
                    g.join_lanes()
                else:
                    g.plan_bn_red()
                    G.record_ops(g, s.bwd, list(reversed(g.ops)), "bwd")
                    g.check_bn_red()
                    g.join_lanes()
                    g.unpack_wgrads()
                    for op in g.post_unpack:
                        op.post_unpack()
            if s.sched is not None:
                s.sched.install(s.bwd)
            seen = set()
            for op in g.ops:
                bns = [op.bn] if isinstance(op, (G.ConvUnitOp, G.BnOnlyOp, G.DwConvUnitOp, G.LnWidthOp)) else ([op.bn_a, op.bn_b] if isinstance(op, G.ConvPairOp) else [])
                if isinstance(op, G.DwConvUnitOp) and id(op.w) not in seen:    # depthwise weights: no pack entry, gradient written directly
                    seen.add(id(op.w))
                    s.used_params.append(op.w)
                for bn in bns:
                    if bn is None:
                        continue
                    for p in (bn.weight, bn.bias):
                        if p is not None and id(p) not in seen:
                            seen.add(id(p))
                            s.used_params.append(p)
            for (_, w, b) in g.pack_entries:
                for p in (w, b):
                    if p is not None and id(p) not in seen:
                        seen.add(id(p))
                        s.used_params.append(p)
            for op in g.post_unpack:
                for p in (op.conv.bias, op.ia, op.im):
                    if id(p) not in seen:
                        seen.add(id(p))
                        s.used_params.append(p)
            s.used_params = [p for p in s.used_params if p.requires_grad]
            s.grad_views = [self.grad_view(p) for p in s.used_params]
        return s

    # ---------------------------------------------------------------- execution
    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _run_plan(self, plan):
        """Replay a launch plan ordered after the work already queued on torch's current
        stream.  Eager replays go straight onto that stream; hipGraph replays use a
        private stream (capture is not allowed on the legacy default stream) fenced with
        events on both sides."""
        use_graph = self.use_graph if self.use_graph != "auto" else (plan.lanes() <= 1 and not plan.hooks())
        if self.use_graph == "auto" and os.environ.get("PLYOLO_GRAPH_FWD") == "1" and getattr(plan, "is_fwd", False):
            use_graph = True    # experiment: forward plans as hipGraphs, backward plans eagerly
        if use_graph and plan.hooks():
            raise PlyoloError("a data-parallel backward plan issues its gradient buckets from host hooks, which a hipGraph replay "
                              "cannot run: use the eager replay (PLYOLO_GRAPH=0 / auto)")
        if not use_graph:
            plan.run(self._stream(), False)
            return
        cur = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        self._side.wait_stream(cur)
        plan.run(self._side.cuda_stream, True)
        cur.wait_stream(self._side)

    def _check_input(self, x):
        if not x.is_cuda:
            raise PlyoloError("pl_yolo_amd runs on an MI355X device tensor (got a %s tensor); there is no CPU path" % x.device.type)
        if x.dim() != 4 or x.shape[1] != 3:
            raise PlyoloError("expected an image batch [B,3,H,W], got %s" % (tuple(x.shape),))
        if x.dtype != torch.float32:
            x = x.float()
        return x.contiguous()

    def _focus(self, s, x):
        """Stage the caller's NCHW fp32 image batch (the only per-step pointer) into the plan's
        NHWC input matrix: Focus gather for CSPDarkNet, plain layout change for EELAN."""
        g = s.g
        if s.stem_kind == "focus":
            call("plyolo_focus_s2d", g.dtype, x.data_ptr(), s.B, s.H, s.W, g.aptr(s.image), s.image.C, self._stream())
        else:
            if not getattr(s, "_img_zeroed", False):
                s.image.storage.tensor.zero_()  # pad channels stay zero forever
                s._img_zeroed = True
            call("plyolo_nchw_f32_to_nhwc", g.dtype, s.B, s.H, s.W, 3, x.data_ptr(), g.aptr(s.image), s.image.ld, self._stream())

    def forward_train(self, x, labels):
        x = self._check_input(x)
        if labels.dim() != 3 or labels.shape[2] != 5 or labels.shape[0] != x.shape[0]:
            raise PlyoloError("labels must be [B, M, 5] rows (cls, cx, cy, w, h); got %s" % (tuple(labels.shape),))
        B, _, H, W = x.shape
        s = self._session(B, H, W, labels.shape[1], "train", x.device)
        s.head.labels.view(labels.shape).copy_(labels)
        self._focus(s, x)
        self._run_plan(s.fwd)
        s.generation += 1
        return s

    def _run_backward(self, s):
        """Replay the backward plan.  A failing launch or gradient-exchange hook surfaces as the REAL error (the exception the hook
        stored, not the replay's generic 'launch N (hook) failed'), the collectives already started are still awaited so the next
        step does not inherit them, and in a process group the failure takes the job down instead of leaving the peers in a collective."""
        try:
            self._run_plan(s.bwd)
        except BaseException as e:
            if s.sched is not None:
                try:
                    s.sched.wait_all()
                except BaseException:
                    pass
                s.sched.pending = []
                hook_err = getattr(s.sched.plan, "hook_error", None)
                s.sched.plan.hook_error = None
                if hook_err is not None:
                    raise PlyoloError("gradient exchange failed inside the backward plan: %r" % (hook_err,)) from hook_err
            raise e
        if s.sched is not None:
            s.sched.wait_all()
            s.sched.check()

    def backward_train(self, s, gout):
        s.head.gout[:gout.numel()].copy_(gout.reshape(-1))
        self._run_backward(s)

    def forward_eval(self, x):
        x = self._check_input(x)
        B, _, H, W = x.shape
        s = self._session(B, H, W, 1, "eval", x.device)
        self._focus(s, x)
        self._run_plan(s.fwd)
        hd = s.head
        return hd.eval_out.view(hd.eval_shape).clone()

    def forward_maps(self, x, want_grad=False):
        x = self._check_input(x)
        B, _, H, W = x.shape
        s = self._session(B, H, W, 1, "maps_grad" if want_grad else "maps", x.device)
        self._focus(s, x)
        self._run_plan(s.fwd)
        s.generation += 1
        hd = s.head
        outs = []
        for (h, w), r0 in zip(hd.sizes, hd.lvl_row):
            blk = hd.raw[r0 * hd.nch:(r0 + B * h * w) * hd.nch].view(B, h, w, hd.nch)
            outs.append(blk.permute(0, 3, 1, 2).contiguous())  # API edge: NCHW like the reference
        return (s, outs) if want_grad else outs

    def backward_maps(self, s, grads):
        s.head.set_map_grads(grads)
        self._run_backward(s)


def _check_generation(ctx, s):
    """A session's activation / z / statistics / head buffers are shared by every forward of that shape, so a
    backward is only valid for the LAST forward of its session: gradient accumulation over micro-batches (two
    forwards, then two backwards) or a `model(x)` call between forward and backward would silently back-propagate
    through the wrong activations.  Refused loudly instead."""
    if ctx.generation != s.generation:
        raise PlyoloError("backward of a stale forward: another forward with the same (batch, size, label rows, mode) ran after "
                          "the one this loss came from, and the HIP launch plan keeps ONE set of activation buffers per shape; "
                          "run forward -> backward pairs back to back (gradient accumulation over such pairs is supported)")


def _accumulating(runner, s):
    """The (parameter, flat view) pairs whose `.grad` still IS the flat view of the previous backward: the caller did not drop those
    gradients, so autograd would ADD the new gradient to whatever they hold now -- the previous micro-batch (gradient accumulation,
    Lightning's accumulate_grad_batches), zeros (zero_grad(set_to_none=False)), a clipped or masked gradient -- while the backward plan
    overwrites the flat buffer.  Decided per parameter and by identity only: the views share ONE version counter with their flat
    buffer, so an in-place change of one gradient is indistinguishable from a change of all of them; adding the current contents back
    is right in every case (it costs two passes over 4 bytes per parameter on such steps; dropping the gradients --
    zero_grad(set_to_none=True), torch's default -- costs nothing).  Returns (kept pairs, all kept?)."""
    if not s.used_params or not runner.flat:
        return [], False
    kept = []
    for p, gv in zip(s.used_params, s.grad_views):
        g = p.grad
        if g is not None and g.data_ptr() == gv.data_ptr():
            kept.append((p, gv))
    return kept, len(kept) == len(s.used_params)


def _publish_grads(runner, s):
    """The backward plan has written every parameter gradient into the runner's flat
    buffer; expose them as `.grad` WITHOUT a copy (returning them through autograd would
    make AccumulateGrad clone all ~240 tensors every step).  Semantics match autograd
    for the reference's loop (optimizer.zero_grad() -> .grad is None, or zeroed in place, before
    backward); a pre-existing foreign .grad tensor is accumulated into, like autograd would;
    a second backward without zero_grad accumulates too (_Accumulate)."""
    for p, gv in zip(s.used_params, s.grad_views):
        g = p.grad
        if g is None or g.data_ptr() == gv.data_ptr():
            p.grad = gv
        else:
            g.add_(gv)


class _Accumulate:
    """Around a backward replay: when the caller is accumulating (see _accumulating) the gradient of the previous micro-batches
    is set aside before the plan overwrites the flat buffer and added back behind it -- two extra passes over the 4 bytes per
    parameter (36 MB for YOLOX-s), only on accumulating steps; `.grad` keeps pointing at the flat buffer.  When only SOME parameters
    kept their gradient (zero_grad on one parameter group, a masked or dropped gradient) the part set aside is added back for exactly
    those.  In a process group the plan's buckets average the NEW gradient only; the part set aside is already the cross-rank mean."""

    def __init__(self, runner, s):
        self.runner, self.prev = runner, None
        self.kept, self.all = _accumulating(runner, s)

    def __enter__(self):
        if self.kept:
            self.prev = self.runner.flat["g"].clone()
        return self

    def __exit__(self, et, ev, tb):
        if self.kept and et is None:
            flat = self.runner.flat["g"]
            if self.all:
                flat.add_(self.prev)
            else:   # only some gradients were kept: add the part set aside back parameter by parameter (rare; a few hundred small adds)
                base, esz = flat.data_ptr(), flat.element_size()
                for _, gv in self.kept:
                    off = (gv.data_ptr() - base) // esz
                    gv.add_(self.prev[off:off + gv.numel()].view_as(gv))
        self.prev = None
        return False


class _TrainStep(torch.autograd.Function):
    """Whole-detector autograd node: forward replays the forward plan, backward the
    backward plan; parameter gradients are views of the runner's flat gradient buffer."""

    @staticmethod
    def forward(ctx, runner, x, labels, *params):
        s = runner.forward_train(x, labels)
        ctx.runner, ctx.session, ctx.generation = runner, s, s.generation
        return s.head.losses.clone()

    @staticmethod
    def backward(ctx, gout):
        runner, s = ctx.runner, ctx.session
        _check_generation(ctx, s)
        with _Accumulate(runner, s):
            runner.backward_train(s, gout.contiguous().float())
        _publish_grads(runner, s)
        return (None, None, None) + (None,) * len(s.used_params)


class _MapsStep(torch.autograd.Function):
    """labels=None in training mode: raw head maps out, caller's loss gradient back in."""

    @staticmethod
    def forward(ctx, runner, x, *params):
        s, outs = runner.forward_maps(x, want_grad=True)
        ctx.runner, ctx.session, ctx.generation = runner, s, s.generation
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        runner, s = ctx.runner, ctx.session
        _check_generation(ctx, s)
        with _Accumulate(runner, s):
            runner.backward_maps(s, [g.contiguous().float() for g in grads])
        _publish_grads(runner, s)
        return (None, None) + (None,) * len(s.used_params)


def maps_step(runner, x):
    x = runner._check_input(x)
    B, _, H, W = x.shape
    s = runner._session(B, H, W, 1, "maps_grad", x.device)
    return list(_MapsStep.apply(runner, x, *s.used_params))


def train_step(runner, x, labels):
    x = runner._check_input(x)
    B, _, H, W = x.shape
    s = runner._session(B, H, W, labels.shape[1], "train", x.device)
    out = _TrainStep.apply(runner, x, labels, *s.used_params)
    return out
