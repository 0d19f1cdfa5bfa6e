"""Tensor contract of the SUB-modules: `model.backbone(x)`, `model.neck(feats)`, `model.head(feats)`, a `CSPLayer(x)`, a `BaseConv(x)`.

The reference's plugins are plain nn.Modules (backbone `Tensor -> list[3]`, models/backbones/darknet_csp.py:61-75; neck
`list -> list`, models/necks/pafpn_csp.py:60-86; head `list -> list of raw NCHW maps`, models/heads/decoupled_head.py:77-95; the
blocks of models/layers/network_blocks.py).  Here a module only DESCRIBES its launches (`emit`); the detector traces the whole
network once (runner.py).  This file gives every describer the reference's tensor contract as well, through the same machinery:
one traced session per (input shapes, dtype, mode), NHWC staging at the API edge, the recorded forward plan, and -- in training
mode with gradients enabled -- one autograd node whose backward replays the recorded backward plan.  It is the boundary for
callers that mix these modules with foreign ones; the detector's hot path never comes through here.

No CPU path: a CPU tensor or a missing library raises PlyoloError.  The loss plugins have their own stand-alone entry
(`loss(list, labels) -> dict | Tensor`, pl_yolo_amd/losses.py).  Eval mode has no backward: an input that requires a gradient raises
(the reference's modules would propagate it), parameters that do are warned about once."""
import os
import warnings

import torch

from . import graph as G
from ._lib import BF16, F32, call, PlyoloError


def _align(n, a=64):
    return (n + a - 1) // a * a


class _Session:
    pass


def _is_head(m):
    return hasattr(m, "n_anchors") and hasattr(m, "num_classes")


def _dtype_of(module):
    name = getattr(module, "compute_dtype", None) or os.environ.get("PLYOLO_DTYPE", "bf16")
    if name not in ("bf16", "fp32"):
        raise PlyoloError("compute_dtype must be 'bf16' or 'fp32'")
    return name


class ModuleRunner:
    """Traced sessions of ONE sub-module (kept on the module: `module.__dict__['_mrunner']`)."""

    def __init__(self, module):
        self.m = module
        self.sessions = {}
        self.gflat = None

    def __deepcopy__(self, memo):
        return None

    # ---------------------------------------------------------------- parameter gradients
    def _flat(self, device):
        params = [p for p in self.m.parameters()]
        # the recorded launches hold the ADDRESSES of the master weights and of the BatchNorm buffers: if the tensors moved since a
        # session was traced (the detector adopted the parameters into its flat buffers, .to(), load_state_dict(assign=True)), every
        # session of this module is stale and is dropped
        key = tuple((id(p), p.data_ptr()) for p in params) + tuple((id(b), b.data_ptr()) for b in self.m.buffers())
        if self.gflat is None or self.gflat["key"] != key or self.gflat["g"].device != device:
            offs, n = {}, 0
            for p in params:
                if p.dtype != torch.float32:
                    raise PlyoloError("parameters must be fp32 master weights")
                offs[id(p)] = n
                n += _align(p.numel())
            self.gflat = dict(key=key, offs=offs, g=torch.zeros(max(n, 8), dtype=torch.float32, device=device))
            self.sessions = {}
        return self.gflat

    def grad_ptr_of(self, p):
        if p is None or not p.requires_grad:
            return None
        return self.gflat["g"].data_ptr() + self.gflat["offs"][id(p)] * 4

    def grad_view(self, p):
        o = self.gflat["offs"][id(p)]
        return self.gflat["g"][o:o + p.numel()].view(p.shape)

    # ---------------------------------------------------------------- tracing
    def _build(self, shapes, image, in_is_list, dtype, training, want_grad, device):
        m = self.m
        g = G.Graph(BF16 if dtype == "bf16" else F32, training, device)
        g.grad_ptr_of = self.grad_ptr_of
        s = _Session()
        s.g, s.image, s.stem_kind = g, image, None
        vec = g.vec
        ins = []
        if image:
            B, _, H, W = shapes[0]
            s.stem_kind = getattr(m, "stem_kind", "focus")
            if s.stem_kind == "focus":
                if H % 2 or W % 2:
                    raise PlyoloError("Focus needs an even image size")
                a = g.new_act(B, H // 2, W // 2, 16 if g.dtype == BF16 else 12, "focus")
            else:
                a = g.new_act(B, H, W, 8 if g.dtype == BF16 else 4, "rgb")
            ins.append(a)
        else:
            for (B, C_, H, W) in shapes:
                if C_ % vec:
                    raise PlyoloError("sub-module input needs a channel count that is a multiple of %d in %s (got %d)" % (vec, dtype, C_))
                ins.append(g.new_act(B, H, W, C_, "in"))
        s.ins = ins
        s.head = None
        if _is_head(m):
            nc = m.num_classes
            strides = [8 * (2 ** i) for i in range(len(ins))]      # only the decode / loss read them; the raw maps do not
            B = ins[0].N
            if m.n_anchors == 1:
                head = G.HeadBuffers(g, B, nc, [(f.H, f.W) for f in ins], strides, 1)
            else:
                anchors = [[10.0, 13.0] * m.n_anchors for _ in ins]
                head = G.V7HeadBuffers(g, B, nc, m.n_anchors, [(f.H, f.W) for f in ins], strides, anchors)
            m.emit(g, ins, head)
            if want_grad:
                head.alloc_grad_only()
            s.head = head
            outs = []
        else:
            res = m.emit(g, ins if in_is_list else ins[0])
            outs = list(res) if isinstance(res, (list, tuple)) else [res]
            s.out_is_list = isinstance(res, (list, tuple))
            for o in outs:
                o.needs_tensor = True
        s.outs = outs
        g.allocate()
        g.build_pack_table(self.grad_ptr_of)
        s.fwd = G.Plan()
        s.fwd.is_fwd = True
        with s.fwd:
            g.plan = s.fwd
            g.pack_weights()
            g.zero_fwd_stats()
            G.record_ops(g, s.fwd, g.ops, "fwd", lanes=False)
        s.bwd = None
        s.used_params = []
        if want_grad:
            # the caller's output gradients arrive in the gradient storages of the output views: mark them written
            for o in outs:
                st = o.storage
                g.grad_storage(st)
                for i in range(o.c_off, o.c_off + o.C):
                    st.ginit[i] = True
            s.bwd = G.Plan()
            with s.bwd:
                g.plan = s.bwd
                if g.dtype != BF16:
                    call("plyolo_memset_async", g.dwp_arena.data_ptr(), 0, g.dwp_arena.numel() * 4, None)
                g.zero_bwd_stats()
                g.plan_bn_red()
                G.record_ops(g, s.bwd, list(reversed(g.ops)), "bwd")
                g.check_bn_red()
                g.join_lanes()
                g.unpack_wgrads()
                for op in g.post_unpack:
                    op.post_unpack()
            seen = set()
            for op in g.ops:
                bns = [op.bn] if isinstance(op, (G.ConvUnitOp, G.BnOnlyOp, G.DwConvUnitOp, G.LnWidthOp)) else ([op.bn_a, op.bn_b] if isinstance(op, G.ConvPairOp) else [])
                if isinstance(op, G.DwConvUnitOp) and id(op.w) not in seen:
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
        return s

    # ---------------------------------------------------------------- execution
    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def _stage_inputs(self, s, xs):
        g = s.g
        if s.image:
            x = xs[0].float().contiguous()
            B, _, H, W = x.shape
            a = s.ins[0]
            if s.stem_kind == "focus":
                call("plyolo_focus_s2d", g.dtype, x.data_ptr(), B, H, W, g.aptr(a), a.C, self._stream())
            else:
                if not getattr(s, "_img_zeroed", False):
                    a.storage.tensor.zero_()
                    s._img_zeroed = True
                call("plyolo_nchw_f32_to_nhwc", g.dtype, B, H, W, 3, x.data_ptr(), g.aptr(a), a.ld, self._stream())
            return
        for a, x in zip(s.ins, xs):
            st = a.storage
            st.tensor.view(st.rows, st.ld)[:, a.c_off:a.c_off + a.C].copy_(x.permute(0, 2, 3, 1).reshape(-1, a.C))

    @staticmethod
    def _act_to_nchw(a, t):
        st = a.storage
        return t.view(st.rows, st.ld)[:, a.c_off:a.c_off + a.C].float().view(a.N, a.H, a.W, a.C).permute(0, 3, 1, 2).contiguous()

    def _outputs(self, s):
        if s.head is not None:
            hd, B = s.head, s.head.B
            outs = []
            for (h, w), r0 in zip(hd.sizes, hd.lvl_row):
                blk = hd.raw[r0 * hd.nch:(r0 + B * h * w) * hd.nch].view(B, h, w, hd.nch)
                outs.append(blk.permute(0, 3, 1, 2).contiguous())
            return outs
        return [self._act_to_nchw(o, o.storage.tensor) for o in s.outs]

    def session(self, xs, image, in_is_list, dtype, training, want_grad):
        device = xs[0].device
        self._flat(device)
        # the recorded backward holds the gradient ADDRESS of every parameter that required a gradient at trace time (`grad_ptr_of`,
        # `used_params`): freezing / unfreezing a parameter afterwards is another session
        mask = tuple(p.requires_grad for p in self.m.parameters()) if want_grad else ()
        key = (tuple(tuple(x.shape) for x in xs), image, in_is_list, dtype, training, want_grad, mask)
        s = self.sessions.get(key)
        if s is None:
            s = self._build([tuple(x.shape) for x in xs], image, in_is_list, dtype, training, want_grad, device)
            s.generation = 0
            self.sessions[key] = s
        return s

    def forward(self, s, xs):
        self._stage_inputs(s, xs)
        s.fwd.run(self._stream(), False)
        s.generation += 1
        return self._outputs(s)

    def backward(self, s, grads):
        g = s.g
        if s.head is not None:
            s.head.set_map_grads(grads)
        else:
            for o, gr in zip(s.outs, grads):
                st = o.storage
                gs = g.grad_storage(st)
                gs.view(st.rows, st.ld)[:, o.c_off:o.c_off + o.C].copy_(gr.permute(0, 2, 3, 1).reshape(-1, o.C))
        s.bwd.run(self._stream(), False)
        dxs = []
        for a in s.ins:
            st = a.storage
            if s.image or st.grad is None or not all(st.ginit[a.c_off:a.c_off + a.C]):
                dxs.append(None)
            else:
                dxs.append(self._act_to_nchw(a, st.grad))
        return dxs


class _ModuleStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, runner, s, n_in, *tensors):
        xs = tensors[:n_in]
        outs = runner.forward(s, xs)
        ctx.runner, ctx.s, ctx.generation = runner, s, s.generation
        ctx.par_ids = [id(t) for t in tensors[n_in:]]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        runner, s = ctx.runner, ctx.s
        if ctx.generation != s.generation:
            raise PlyoloError("backward of a stale sub-module forward: another forward of the same shapes ran in between (one set of activation buffers per shape)")
        dxs = runner.backward(s, [g.contiguous().float() for g in gouts])
        # parameter gradients: autograd's rule, p.grad += new (the plan wrote them into the runner's flat buffer)
        pg = {id(p): runner.grad_view(p).clone() for p in s.used_params}
        return (None, None, None) + tuple(dxs) + tuple(pg.get(i) for i in ctx.par_ids)


def run(module, *args):
    """`module(*args)` with the reference's tensor contract."""
    if not args:
        raise PlyoloError("%s() needs its input tensor(s)" % type(module).__name__)
    x = args[0]
    in_is_list = isinstance(x, (list, tuple))
    xs = list(x) if in_is_list else [x]
    if len(args) > 1:
        raise PlyoloError("%s takes one input (a tensor, or the list of feature maps)" % type(module).__name__)
    for t in xs:
        if not torch.is_tensor(t) or t.dim() != 4:
            raise PlyoloError("expected NCHW tensors, got %r" % (type(t),))
        if not t.is_cuda:
            raise PlyoloError("pl_yolo_amd runs on an MI355X device tensor (got a %s tensor); there is no CPU path" % t.device.type)
    p0 = next(module.parameters(), None)
    if p0 is not None and not p0.is_cuda:
        raise PlyoloError("move the module to the MI355X first (.to('cuda')); there is no CPU path")
    r = module.__dict__.get("_mrunner")
    if r is None:
        r = ModuleRunner(module)
        module.__dict__["_mrunner"] = r
    image = (not in_is_list) and xs[0].shape[1] == 3 and (hasattr(module, "stem_kind") or type(module).__name__ == "Focus")
    dtype = _dtype_of(module)
    params = [p for p in module.parameters() if p.requires_grad]
    want_grad = module.training and torch.is_grad_enabled() and (any(t.requires_grad for t in xs) or len(params) > 0)
    if not module.training and torch.is_grad_enabled():
        # eval mode runs the inference plan (BatchNorm folded into the convolution epilogues, no `z` kept): it has no backward.  The
        # reference's nn.Modules do propagate gradients in eval mode (running statistics as constants) -- a caller who needs that
        # (a frozen `neck.eval()` between a trainable backbone and head) must not get detached outputs silently:
        if any(t.requires_grad for t in xs):
            raise PlyoloError("%s is in eval mode and its input requires a gradient: the eval-mode launch plan has no backward, the "
                              "gradient would be cut here; call it in train() mode, or under torch.no_grad() / with detached inputs if "
                              "no gradient is wanted" % type(module).__name__)
        if params and not module.__dict__.get("_warned_eval_grad"):
            module.__dict__["_warned_eval_grad"] = True
            warnings.warn("%s called in eval mode with gradients enabled: its outputs carry no grad_fn (the parameters get no gradient "
                          "from this call); wrap the call in torch.no_grad() to silence this" % type(module).__name__, stacklevel=3)
    s = r.session(xs, image, in_is_list, dtype, module.training, want_grad)
    if want_grad:
        outs = list(_ModuleStep.apply(r, s, len(xs), *xs, *params))
    else:
        outs = r.forward(s, xs)
    if s.head is not None or getattr(s, "out_is_list", False):
        return outs
    return outs[0]
