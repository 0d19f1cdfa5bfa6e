"""Launch-plan builder: lowers a traced detector graph to recorded HIP launches.

The Python module tree (pl_yolo_amd.layers / backbones / necks / heads / losses,
mirroring the reference's plugin classes) owns the parameters and *describes* the
network by calling the emit helpers of a `Graph`.  The graph is shape-specialised,
then lowered once into two launch plans inside libplyolo_hip.so (forward and
backward: `plyolo_plan_*`), which are replayed -- eagerly or as a captured hipGraph --
for every step.  No tracing compiler, no per-op Python dispatch on the hot path.

Memory model (HBM, everything resident for the whole run):
  * activations: NHWC matrices `[N*H*W, ld]` (`Storage`); channel concatenation is a
    strided view (`Act.c_off`) into a wider Storage, so `torch.cat` never copies;
  * every conv unit keeps its raw conv output `z` (for the BN/SiLU backward), its
    per-channel coefficients and the packed bf16 weights;
  * gradients mirror the activation storages; "first writer writes, later writers
    accumulate" is resolved at lowering time (no zero-fill passes).
"""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import BF16, F32, ACT, STAT_SLOTS, BnStats, Split, BnBwdSplit, BnBwdFuse, BnRed, BN_RED_SEGS, ConvDesc, PackEntry, YoloxDesc, yolov7_desc, call, ptr

BN_EPS_DEFAULT = 1e-3


def _align(n, a=64):
    return (n + a - 1) // a * a


class Storage:
    """A dense [rows, ld] activation matrix (rows = N*H*W pixels)."""

    def __init__(self, N, H, W, ld, name=""):
        self.N, self.H, self.W, self.ld, self.name = N, H, W, ld, name
        self.rows = N * H * W
        self.tensor = None
        self.grad = None
        self.ginit = [False] * ld  # per channel: has the gradient been written yet (lowering-time state)
        self.views = 0
        self.acts = []


class Act:
    """A channel slice [c_off, c_off+C) of a Storage."""

    def __init__(self, storage, C_, c_off=0):
        self.storage, self.C, self.c_off = storage, C_, c_off
        storage.views += 1
        storage.acts.append(self)
        self.fixed = False  # True once something depends on the placement
        # lazy activation (training plans): `producer` = the conv unit whose BatchNorm + activation would write this
        # tensor; consumers that can apply them while they stage their input register in `lazy_users`, every other
        # reader sets `needs_tensor`.  Resolved by Graph.resolve_lazy() after the trace: `lazy` = LazySrc or None.
        self.producer = None
        self.lazy_users = []
        self.needs_tensor = False
        self.lazy = None

    N = property(lambda s: s.storage.N)
    H = property(lambda s: s.storage.H)
    W = property(lambda s: s.storage.W)
    ld = property(lambda s: s.storage.ld)
    M = property(lambda s: s.storage.rows)

    def rebind(self, storage, c_off):
        self.needs_tensor = True   # placed inside a concat matrix: the consumer reads the matrix, not this unit's z
        self.storage.views -= 1
        self.storage.acts.remove(self)
        self.storage, self.c_off = storage, c_off
        storage.views += 1
        storage.acts.append(self)


class LazySrc:
    """Where a lazy activation really lives: channels [c_off, c_off + C) of the producer's raw conv output `z`
    (pitch z.ld) + the producer's coefficient table (scale | shift | mean | invstd, row pitch coef_ld) + activation."""

    def __init__(self, op, z, c_off, act):
        self.op, self.z, self.c_off, self.act = op, z, c_off, act


class Graph:
    def __init__(self, dtype, training, device):
        self.dtype = dtype
        self.training = training
        self.device = device
        self.tdtype = torch.bfloat16 if dtype == BF16 else torch.float32
        self.esize = 2 if dtype == BF16 else 4
        self.vec = 8 if dtype == BF16 else 4
        self.storages = []
        self.ops = []
        self.convs = []      # PackedConv records (weight arena bookkeeping)
        self.keep = []       # tensors that must outlive the plans
        self.scratch_elems = 0  # shared dz scratch (activation dtype)
        self.scratch_f32 = 0    # shared fp32 scratch (maxpool backward)
        self.image_act = None
        self.post_unpack = []   # ops with work that must follow unpack_wgrads
        self.plan = None        # the Plan being recorded (ops use it for lanes / events)
        self.cur_lane, self.cur_lane_fwd = 0, None
        self.side_lanes = []    # compute lanes besides lane 0 used by the plan being recorded (record_ops)
        self.pending = {}       # lane -> queued weight-gradient closures (defer_param_grads)
        # weight gradients on their own lane (PLYOLO_LANES=0 keeps every launch on lane 0)
        self.use_lanes = os.environ.get("PLYOLO_LANES", "1") != "0"
        self.reduce_slabs = os.environ.get("PLYOLO_REDUCE_SLABS", "1") == "1"
        self.reduce_batch = int(os.environ.get("PLYOLO_REDUCE_BATCH", "4"))   # layers per batched slab-fold launch (1 = one launch per layer; round 3, four alternations: 2 -> 9.055, 3 -> 9.03, 4 -> 9.02 ms)
        self.reduce_queue = []
        # merge same-input conv pairs (ConvPairOp) in training plans; inference plans fuse BatchNorm + activation
        # into each convolution's epilogue instead (ConvUnitOp.fwd), which needs one output matrix per conv
        self.pair_convs = os.environ.get("PLYOLO_PAIR", "1") == "1" and training
        self.fuse_eval = os.environ.get("PLYOLO_FUSE_EVAL", "1") == "1"
        # lazy activations (training, PLYOLO_LAZY=1): a BaseConv whose output only feeds convolutions does not write it --
        # the consumers apply BatchNorm + activation to the raw conv output while staging it (plyolo_conv_desc.x_coef).
        # Bit-identical to the materialised path and OFF by default: measured on YOLOX-s B=32 it removes 36 of the 63
        # bn_act_fwd passes (-0.72 ms) but costs +0.75 ms in the forward convolutions and +1.8 ms in the weight-gradient
        # kernels (10.5 -> 12.2 ms/step): SiLU is 2 transcendental + ~6 plain VALU instructions per element, the whole
        # chip sustains ~3.8 T SiLU/s -- the same order as the HBM stream itself -- so inside a loader the work does not
        # disappear, it lengthens every workgroup's load -> stage -> MFMA chain
        # PLYOLO_FUSE_BNBWD: pointwise units form dz inside their data gradient's loader (plyolo_conv2d_dgrad_bn: one launch and one
        # pass over dout / z less per unit, bit-identical).  Round 3: 26 launches fewer per YOLOX-s step and no measurable change of the
        # step time (profiles/r03_ab_bn_fusion.txt), 2 % slower on YOLOX-x at 1280x1280 without its size limit -- off.  Round 4: with the
        # loader's SiLU instance (the run-time activation switch compiled to a branch per element) and the units of the large maps gone
        # to plyolo_conv2d_bwd_pw it takes the remaining small-map pointwise units: 15 launches and ~0.4 GB less, 8.66 vs 8.69 / 8.63
        # vs 8.66 ms -- on by default
        self.fuse_bnbwd = os.environ.get("PLYOLO_FUSE_BNBWD", "1") == "1"
        # PLYOLO_FUSE_BNBWD3 (default OFF, round 5: bit-identical and slower -- the fused launch reads dout AND z over the 1.4x halo and still writes dz for the weight gradient: 4.8 E in a bandwidth-bound kernel against 5.4 E split over a stream-rate pass and an MFMA-bound one, profiles/r05_ab_dz_on_load.txt): the same for the 3x3 stride-1 units -- the halo loader of their data gradient takes
        # (dout, z) pairs and stages dz (csrc/conv_mfma_bnb.hip); the bn_act_bwd_dz launch of the unit leaves the data-gradient chain
        self.fuse_bnbwd3 = os.environ.get("PLYOLO_FUSE_BNBWD3", "0") == "1" and dtype == BF16 and training
        # units without a data gradient (the first convolution): dz formed in the weight gradient's loader (plyolo_conv2d_wgrad_bn)
        self.fuse_wgbn = os.environ.get("PLYOLO_FUSE_WGBN", "1") == "1" and training and dtype == BF16
        self.fwd_res_in_dz = os.environ.get("PLYOLO_RES_IN_DZ", "1") == "1"    # A/B switch: 0 = a copy_add launch per shortcut
        # PLYOLO_FUSE_PWBWD (default on, round 4): the whole backward of a large-map pointwise unit behind its BatchNorm reduction --
        # dz, data gradient and weight gradient -- is ONE persistent launch (plyolo_conv2d_bwd_pw): dout, z and x are read once, dz
        # never reaches HBM, the bn_act_bwd_dz pass and the 1x1 weight-gradient launch of those units disappear
        self.fuse_pwbwd = os.environ.get("PLYOLO_FUSE_PWBWD", "1") == "1" and dtype == BF16 and training
        self.diag_skip_r = os.environ.get("PLYOLO_DIAG_SKIP_R", "0") == "1"   # diagnostics only (results are wrong): no bn_act_bwd_reduce launches
        # PLYOLO_FUSE_BNRED (default on, round 4): bn_act_bwd_reduce of a unit rides the store loop of the data gradient that writes the
        # unit's output gradient LAST (plan_bn_red): dx is not read back, one launch less per unit on the data-gradient chain
        self.fuse_bnred = os.environ.get("PLYOLO_FUSE_BNRED", "1") == "1" and dtype == BF16 and training
        self.red_plan = {}      # (id(producer op), id(input Act)) -> [(c0, c1, unit op, channel offset inside the unit)]
        self.gm_log = []        # (op, Act) of every gradient write of the backward recording (plan_bn_red is verified against it)
        self.cur_op = None
        self.lazy_acts = os.environ.get("PLYOLO_LAZY", "0") in ("1", "2") and training
        # PLYOLO_LAZY=2: selective -- only where EVERY reader is a pointwise (1x1 stride-1) convolution: those kernels (and their
        # 1x1 weight gradients) are HBM-bound with an idle VALU, and there is no halo to re-pay the activation on
        self.lazy_pw_only = os.environ.get("PLYOLO_LAZY", "0") == "2"
        if self.lazy_acts and dtype == BF16 and not (_lib.lib().plyolo_build_flags() & 1):
            raise _lib.PlyoloError("PLYOLO_LAZY needs the lazy-input kernel instances, which the shipped libplyolo_hip.so does not carry "
                                   "(measured slower): rebuild with `make -C pl_yolo_amd/csrc OPTIN=1`")

    # ------------------------------------------------------------------ batched slab folds
    def queue_reduce(self, pc):
        """Slab folds are tiny launches (69 per YOLOX-s step, ~12 us each, mostly launch floor): they are queued and run as ONE
        two-stage launch per `reduce_batch` layers on the weight-gradient lane (and before anything reads slab 0)."""
        self.reduce_queue.append((pc, self.plan.cur if self.plan is not None else 0))   # + the lane its slabs were written on
        if len(self.reduce_queue) >= self.reduce_batch:
            self.flush_reduce()

    def flush_reduce(self):
        """One batched fold launch for the queued layers.  With lanes it always sits on the weight-gradient lane, behind an
        event of every OTHER lane that wrote one of the queued slabs (ImplicitHead levels on a side lane, ops that issue their
        weight gradient inline): whichever lane happens to be current when the queue fills, the fold is ordered after every
        writer; the caller's lane is restored."""
        qq, self.reduce_queue = self.reduce_queue, []
        if not qq:
            return
        q = [pc for pc, _ in qq]
        plan, prev = self.plan, None
        if self.use_lanes and plan is not None:
            prev = plan.cur
            for l in sorted({l for _, l in qq} - {WGRAD_LANE}):
                plan.wait(WGRAD_LANE, plan.record(l))
            plan.lane(WGRAD_LANE)
        from ._lib import ReduceJob
        arr = (ReduceJob * len(q))()
        max_cols = max_groups = 1
        total = 0.0
        for i, pc in enumerate(q):
            per, groups = C.c_int(), C.c_int()
            call("plyolo_reduce_slabs_plan", pc.nslab, pc.dwp_elems, C.byref(per), C.byref(groups))
            arr[i].dwp, arr[i].nslab, arr[i].per, arr[i].groups, arr[i].elems = pc.dwp, pc.nslab, per.value, groups.value, pc.dwp_elems
            max_cols = max(max_cols, (pc.dwp_elems // 4 + 255) // 256)
            max_groups = max(max_groups, groups.value)
            total += 4.0 * pc.dwp_elems * pc.nslab
        t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.keep.append(t)
        call("plyolo_reduce_slabs_multi", t.data_ptr(), len(q), max_cols, max_groups, total, None)
        if prev is not None:
            plan.lane(prev)

    # ------------------------------------------------------------------ lazy activations
    def resolve_lazy(self):
        """After the trace: decide per conv unit whether its activated output is materialised (bn_act_fwd, as the
        reference does) or stays lazy.  Lazy needs: every reader is a convolution that registered as lazy-capable, no
        residual is added by the unit, and the output was not placed into a concat matrix."""
        n = 0
        for op in self.ops:
            outs = op.lazy_outputs() if hasattr(op, "lazy_outputs") else []
            ok = self.lazy_acts and bool(outs) and all(a.lazy_users and not a.needs_tensor for a, _ in outs)
            if ok and self.lazy_pw_only:
                ok = all(isinstance(u, HeadPredOp) or (u.k == 1 and u.stride == 1) for a, _ in outs for u in a.lazy_users)
            for a, c_off in outs:
                if ok:
                    a.lazy = LazySrc(op, op.z, c_off, op.act)
                    if a.storage in self.storages:
                        self.storages.remove(a.storage)   # never allocated forward; its gradient matrix still is
                    n += 1
            if hasattr(op, "lazy_out"):
                op.lazy_out = ok
        self.n_lazy = n

    def src(self, a):
        """(device pointer, pitch) a convolution reads for activation view `a`: the tensor itself, or the producer's z."""
        if a.lazy is not None:
            lz = a.lazy
            return lz.z.tensor.data_ptr() + lz.c_off * self.esize, lz.z.ld
        return self.aptr(a), a.ld

    def set_lazy(self, desc, a):
        """Fill the lazy-input fields of a conv descriptor for input view `a` (no-op for a materialised input)."""
        if a.lazy is None:
            desc.x_coef, desc.x_coef_ld, desc.x_act = None, 0, 0
            return desc
        lz = a.lazy
        desc.x_coef = lz.op.coef.data_ptr() + lz.c_off * 4
        desc.x_coef_ld, desc.x_act = lz.op.Cout, lz.act
        return desc

    # ------------------------------------------------------------------ lanes
    def add_op(self, op):
        op.index = len(self.ops)     # position in forward order
        op.lane = self.cur_lane
        op.lane_fwd = self.cur_lane_fwd if self.cur_lane_fwd is not None else self.cur_lane   # forward plans may place an op elsewhere
        self.ops.append(op)

    def on_lane(self, lane):
        """`with g.on_lane(l):` -- ops emitted inside run on launch lane `l` (its own HIP stream under the eager
        multi-stream replay).  Lanes are a pure placement hint: record_ops() derives every cross-lane ordering from the
        tensors the ops read and write, so any assignment is correct; a good one puts independent chains side by side
        (the PAFPN bottom-up path and the small head levels beside the 80x80 head level).  PLYOLO_LANES=0 keeps
        everything on lane 0."""
        return _OnLane(self, lane if self.use_lanes else 0, None)

    def on_lanes(self, lane_fwd, lane_bwd):
        """Like on_lane, with a lane of its own for the FORWARD plan: the weight-gradient lane (1) idles there and can carry a
        branch; in the backward plan the same ops sit on `lane_bwd`."""
        return _OnLane(self, lane_bwd if self.use_lanes else 0, lane_fwd if self.use_lanes else 0)

    def fork(self):
        """Compatibility spelling: `with g.fork() as r: with r.branch(lane): ...` == `with g.on_lane(lane): ...`."""
        return _Fork(self)

    def defer_param_grads(self, lane, fn):
        """Queue weight-gradient work (a closure issuing launches) for the weight-gradient lane.  The hand-off
        costs an event on the main lane (a ~7 us bubble between two kernels), so it is paid once per
        WGRAD_BATCH conv units instead of once per unit: with private dz buffers the queued wgrads may start
        any time after their dz kernel, a few layers of lag cost nothing."""
        if os.environ.get("PLYOLO_DIAG_SKIP_WG", "0") == "1":     # diagnostics only (results are wrong): no weight-gradient launches at all
            return
        q = self.pending.setdefault(lane, [])
        q.append(fn)
        if len(q) >= WGRAD_BATCH:
            self.flush_param_grads(lane)

    def flush_param_grads(self, lane=None):
        for l in ([lane] if lane is not None else list(self.pending)):
            q = self.pending.get(l) or []
            if not q:
                continue
            plan = self.plan
            ev = plan.record(l)
            plan.lane(WGRAD_LANE)
            plan.wait(WGRAD_LANE, ev)
            for fn in q:
                fn()
            plan.lane(l)
            self.pending[l] = []

    # ------------------------------------------------------------------ BatchNorm-backward reduction inside data gradients
    def plan_bn_red(self):
        """Before the backward plan is recorded: which data gradient writes the output gradient of every BatchNorm unit LAST?  When
        that writer is a convolution data gradient with a RED kernel instance and its dx covers the unit's output view (one of up to
        three channel segments of a concatenated input), the unit's bn_act_bwd_reduce launch is dropped and the writer folds the two
        per-channel sums while it stores dx (plyolo_conv2d_dgrad_red / plyolo_conv2d_bwd_pw_red).  A unit is only taken when ALL its
        output views are covered (both halves of a merged pair).  Gradient accumulations into one buffer run in list order
        (record_ops), so "last in reversed(ops)" is the last writer whatever lanes the ops sit on; record_ops logs every actual
        gradient write and check_bn_red() compares."""
        self.red_plan, self.gm_log = {}, []
        for op in self.ops:
            if hasattr(op, "red_done"):
                op.red_done = False
        if not self.fuse_bnred:
            return
        lib = _lib.lib()
        views = []     # (unit, Act view, channel offset inside the unit)
        for op in self.ops:
            if op.__class__ in (ConvUnitOp, ConvPairOp) and op.act > ACT["lrelu"]:
                continue       # hswish / gelu units keep their reduce launch (the folded form carries the three cheap activations only)
            if isinstance(op, ConvUnitOp) and op.bn is not None and op.conv_b is None and hasattr(op, "coef"):
                views.append((op, op.out, 0))
            elif isinstance(op, ConvPairOp) and hasattr(op, "coef"):
                views.append((op, op.out_a, 0))
                views.append((op, op.out_b, op.Ca))
        # which ops will run their backward at all?  An op whose output receives no gradient returns early (`if not g.grad_ready(self.out)`)
        # and writes nothing: listing it as a writer would plan a fold into a launch that never happens (check_bn_red would then refuse
        # the whole graph).  Same readiness rule as the bwd methods, simulated over the views: all channels of an output written.
        sim, live = {}, set()

        # (a gradient the caller supplies -- the outputs of a sub-module traced on its own, module_runner.py -- is marked written before)
        def _ready(a):
            f = sim.setdefault(id(a.storage), list(a.storage.ginit))
            return all(f[a.c_off:a.c_off + a.C])

        def _mark(a):
            f = sim.setdefault(id(a.storage), list(a.storage.ginit))
            for i in range(a.c_off, a.c_off + a.C):
                f[i] = True
        for op in reversed(self.ops):
            try:
                ins, outs = op_io(op)
            except TypeError:
                ins, outs = [], []
            aouts = [o for o in outs if isinstance(o, Act)]
            if aouts and not any(_ready(o) for o in aouts):
                continue
            live.add(id(op))
            for a in ins:
                if isinstance(a, Act) and not (a is getattr(op, "x", None) and getattr(op, "need_dgrad", True) is False):
                    _mark(a)
        # gradient writes in backward order: (op, input Act, is a convolution data gradient with a RED instance)
        writes = []
        for op in reversed(self.ops):
            if id(op) not in live:
                continue
            if isinstance(op, ConvUnitOp):
                if op.res is not None:
                    writes.append((op, op.res, False))
                if op.need_dgrad:
                    ok = op.bn is not None and hasattr(op, "desc_d") and (bool(op.pw_slabs) or lib.plyolo_conv2d_dgrad_red_fits(C.byref(op.desc_d)) == 1)
                    bnb = not op.pw_slabs and op.bn is not None and hasattr(op, "desc_d") and self.dgrad_bn_fits(op.desc_d, op.act)
                    writes.append((op, op.x, ok or bnb))
            elif isinstance(op, ConvPairOp):
                ok = hasattr(op, "desc_d") and (bool(op.pw_slabs) or lib.plyolo_conv2d_dgrad_red_fits(C.byref(op.desc_d)) == 1)
                bnb = not op.pw_slabs and hasattr(op, "desc_d") and self.dgrad_bn_fits(op.desc_d, op.act) and (op.k != 3 or op.Ca % 32 == 0)
                writes.append((op, op.x, ok or bnb))
            elif isinstance(op, HeadPredOp):
                writes.append((op, op.reg_feat, lib.plyolo_conv2d_dgrad_red_fits(C.byref(op.dgrad_descs()[0])) == 1))
                writes.append((op, op.cls_feat, lib.plyolo_conv2d_dgrad_red_fits(C.byref(op.dgrad_descs()[1])) == 1))
            else:
                try:
                    ins, _ = op_io(op)
                except TypeError:
                    ins = []
                for a in ins:
                    if isinstance(a, Act):
                        writes.append((op, a, False))
        # ... only where it pays: a big reduce launch streams at ~5 TB/s, the same bytes in a conv epilogue cost more than they save
        # (YOLOX-s stem, 210 MB of z: reduce 85 us alone, +106 us in the stride-2 data gradient that would take it; same box, means
        # of three alternations: no limit 8.81 ms, 120 MB 8.63, 64 MB 8.64, 30 MB 8.76)
        max_mb = float(os.environ.get("PLYOLO_BNRED_MAX_MB", "64"))
        plan, covered = {}, {}
        for (u, v, choff) in views:
            rv = _res(v)
            last = None
            for (wop, a, ok) in writes:
                if _overlap(_res(a), rv):
                    last = (wop, a, ok)
            if last is None:
                continue
            wop, a, ok = last
            # (the persistent pointwise kernel requests the unit's z with its tile and waits for it behind its MFMAs: twice the limit)
            if v.M * v.C * self.esize > max_mb * 1.0e6 * (2.0 if getattr(wop, "pw_slabs", 0) else 1.0):
                continue
            ra = _res(a)
            if not ok or ra[0] != rv[0] or ra[1] > rv[1] or ra[2] < rv[2]:
                continue
            plan.setdefault((id(wop), id(a)), []).append((rv[1] - ra[1], rv[2] - ra[1], u, choff))
            covered.setdefault(id(u), []).append((u, v))
        # a producer takes at most BN_RED_SEGS segments; a unit needs all its views taken
        for key in [k for k, segs in plan.items() if len(segs) > BN_RED_SEGS]:
            for (_, _, u, _) in plan.pop(key):
                covered[id(u)] = [(uu, vv) for (uu, vv) in covered.get(id(u), []) if False]
        nviews = {}
        for (u, v, choff) in views:
            nviews[id(u)] = nviews.get(id(u), 0) + 1
        full = {uid for uid, lst in covered.items() if len(lst) == nviews.get(uid, 0) and lst}
        for key, segs in plan.items():
            keep = [sg for sg in segs if id(sg[2]) in full]
            if keep:
                self.red_plan[key] = keep
        for (u, v, choff) in views:
            if id(u) in full and any(any(sg[2] is u for sg in segs) for segs in self.red_plan.values()):
                u.red_done = True

    def red_for(self, op, a):
        """plyolo_bn_red of producer `op` for its input view `a` (None: nothing planned).  The returned struct must stay alive."""
        segs = self.red_plan.get((id(op), id(a)))
        if not segs:
            return None
        r = BnRed()
        r.n = len(segs)
        for i, (c0, c1, u, choff) in enumerate(segs):
            sg = r.seg[i]
            ct = u.Cout
            sg.c0, sg.c1 = c0, c1
            sg.z, sg.z_ld = u.z.tensor.data_ptr() + choff * self.esize, ct
            sg.coef, sg.coef_ld = u.coef.data_ptr() + choff * 4, ct
            sg.bslots, sg.slot_ld = self.bstat_arena.data_ptr() + (u.slot_off + choff) * 8, ct
            sg.act = u.act
        self.keep.append(r)
        return r

    def check_bn_red(self):
        """After the backward recording: every planned producer really was the last gradient writer of its segments."""
        for (opid, aid), segs in self.red_plan.items():
            for (c0, c1, u, choff) in segs:
                views = [u.out] if isinstance(u, ConvUnitOp) else [u.out_a, u.out_b]
                v = views[0] if choff == 0 else views[1]
                rv = _res(v)
                last = None
                for (wop, a) in self.gm_log:
                    if _overlap(_res(a), rv):
                        last = (wop, a)
                if last is None or id(last[0]) != opid or id(last[1]) != aid:
                    raise _lib.PlyoloError("plan_bn_red: the planned last writer of a unit's output gradient is not the recorded one "
                                           "(%s planned, %s recorded)" % (opid, type(last[0]).__name__ if last else None))

    def dgrad_bn_fits(self, desc_d, act):
        """True when the unit's dz is formed inside its data gradient's loader (plyolo_conv2d_dgrad_bn): pointwise units
        (PLYOLO_FUSE_BNBWD) and 3x3 stride-1 units (PLYOLO_FUSE_BNBWD3)."""
        if not (self.fuse_bnbwd3 if desc_d.ksize == 3 else self.fuse_bnbwd):
            return False
        return _lib.lib().plyolo_conv2d_dgrad_bn_fits(C.byref(desc_d), act) == 1

    def pw_bwd_slabs(self, desc, act, ok=True):
        """Private weight-gradient slabs of plyolo_conv2d_bwd_pw for this unit, or 0 when the unit keeps the separate
        dz / data-gradient / weight-gradient launches (not covered, switched off, lazy inputs in play)."""
        if not (ok and self.fuse_pwbwd and not self.lazy_acts):
            return 0
        lib = _lib.lib()
        if lib.plyolo_conv2d_bwd_pw_fits(C.byref(desc), act) != 1:
            return 0
        return max(lib.plyolo_conv2d_bwd_pw_slabs(C.byref(desc)), 0)

    def dz_buffer(self, op, elems):
        """dz scratch of one conv unit's backward.  With lanes every unit owns its buffer (the weight-gradient
        lane may still be reading layer i's dz when the main lane is many layers upstream; private buffers
        need no reuse events -- measured +2 % over a 4-deep rotation, and 288 GB of HBM make it free);
        without lanes one shared buffer per lane is reused by every unit.  Returns (device pointer, key)."""
        if self.use_lanes:
            if not hasattr(op, "dz_buf"):
                op.dz_buf = torch.empty(max(elems, 8), dtype=self.tdtype, device=self.device)
            return op.dz_buf.data_ptr(), None
        return self.scratch[op.lane][0].data_ptr(), None

    # ------------------------------------------------------------------ tracing
    def new_act(self, N, H, W, C_, name=""):
        st = Storage(N, H, W, C_, name)
        self.storages.append(st)
        return Act(st, C_)

    def concat(self, acts):
        """torch.cat(dim=1) as a zero-copy strided placement whenever the inputs are
        fresh producer outputs; otherwise a copy op is inserted."""
        a0 = acts[0]
        total = sum(a.C for a in acts)
        st = Storage(a0.N, a0.H, a0.W, total, "cat")
        st.is_cat, st.nested = True, False
        self.storages.append(st)
        off = 0
        for a in acts:
            assert (a.N, a.H, a.W) == (a0.N, a0.H, a0.W), "concat: spatial mismatch"
            if (not a.fixed) and a.storage.views == 1 and a.c_off == 0 and a.storage.ld == a.C:
                old = a.storage
                a.rebind(st, off)
                a.fixed = True
                self.storages.remove(old)
            elif a.fixed and a.c_off == 0 and a.storage.ld == a.C and getattr(a.storage, "is_cat", False) and not a.storage.nested:
                # nested concat: relocate the whole inner concat matrix (all its views) into this one
                old = a.storage
                for v in list(old.acts):
                    v.rebind(st, off + v.c_off)
                self.storages.remove(old)
            else:
                dst = Act(st, a.C, off)
                self.add_op(CopyOp(self, a, dst))
            off += a.C
        out = Act(st, total, 0)
        out.fixed = True
        return out

    # --------------------------------------------------------------- allocation
    def allocate(self):
        dev = self.device
        self.resolve_lazy()
        for st in self.storages:
            st.tensor = torch.empty(st.rows * st.ld, dtype=self.tdtype, device=dev)
        # shared dz scratch (only used without lanes; with lanes every conv unit owns its dz buffer, see dz_buffer)
        main_lanes = sorted({op.lane for op in self.ops})
        self.scratch = {l: [torch.empty(max(self.scratch_elems if not self.use_lanes else 8, 8), dtype=self.tdtype, device=dev)]
                        for l in main_lanes}
        self.scratch32 = torch.zeros(max(self.scratch_f32, 8), dtype=torch.float32, device=dev)
        cmax = max([c.Cout_total for c in self.convs] + [8])
        # fp64 stat slots of every BatchNorm (forward: sum z, sum z^2; backward: sum du, sum du*zhat);
        # one memset per plan zeroes a whole arena (zero_fwd_stats / zero_bwd_stats)
        off = 0
        for op in self.ops:
            if isinstance(op, (ConvUnitOp, ConvPairOp, BnOnlyOp, DwConvUnitOp)) and op.bn is not None:
                op.slot_off = off
                off += STAT_SLOTS * 2 * op.Cout
        self.stat_arena = torch.zeros(max(off, 8), dtype=torch.float64, device=dev)
        self.bstat_arena = torch.zeros(max(off, 8), dtype=torch.float64, device=dev)
        # weight arenas
        wp_n = sum(_align(c.wp_elems) for c in self.convs)
        dwp_n = sum(_align(c.dwp_elems * c.nslab) for c in self.convs)
        wpd_n = sum(_align(c.wpd_elems) for c in self.convs)
        self.wp_arena = torch.zeros(max(wp_n, 8), dtype=self.tdtype, device=dev)
        self.wpd_arena = torch.zeros(max(wpd_n, 8), dtype=self.tdtype, device=dev)
        self.dwp_arena = torch.zeros(max(dwp_n, 8), dtype=torch.float32, device=dev)
        nb = sum(_align(c.Cout_total) for c in self.convs)
        self.bias_arena = torch.zeros(max(nb, 8), dtype=torch.float32, device=dev)
        self.dbias_arena = torch.zeros(max(nb, 8), dtype=torch.float32, device=dev)
        o1 = o2 = o3 = o4 = 0
        entries = []
        self.max_pack_elems = 8
        for c in self.convs:
            c.wp = self.wp_arena.data_ptr() + o1 * self.esize
            c.dwp = self.dwp_arena.data_ptr() + o4 * 4
            c.wpd = self.wpd_arena.data_ptr() + o2 * self.esize if c.need_dgrad else None
            c.bp = self.bias_arena.data_ptr() + o3 * 4
            c.dbp = self.dbias_arena.data_ptr() + o3 * 4
            o1 += _align(c.wp_elems)
            o4 += _align(c.dwp_elems * c.nslab)
            o2 += _align(c.wpd_elems)
            o3 += _align(c.Cout_total)
            for (w, b, co_off) in c.sources:
                e = PackEntry()
                e.w = w.data_ptr()
                e.wp, e.wpd, e.dwp = c.wp, c.wpd, c.dwp
                e.dw = None
                e.b = b.data_ptr() if b is not None else None
                e.bp, e.dbp, e.db = c.bp, c.dbp, None
                e.Cout, e.Cin, e.Cin_p, e.ksize = w.shape[0], w.shape[1], c.Cin_p, c.ksize
                e.Cout_total, e.Cout_p8, e.co_off = c.Cout_total, c.Cout_p8, co_off
                e.nslab = 1 if self.reduce_slabs else c.nslab
                entries.append((e, w, b))
                self.max_pack_elems = max(self.max_pack_elems, c.ksize * c.ksize * w.shape[0] * c.Cin_p)
        self.pack_entries = entries

    def flush_reduce_on_lane(self):
        """Run the queued slab folds now, on the lane the weight gradients run on (before anything reads slab 0)."""
        if not self.reduce_queue:
            return
        if self.use_lanes:
            self.flush_param_grads()
            self.plan.lane(WGRAD_LANE)
            self.flush_reduce()
            self.plan.lane(0)
        else:
            self.flush_reduce()

    def join_lanes(self):
        """Everything recorded on the weight-gradient lane so far happens before what lane 0 records next."""
        self.flush_reduce_on_lane()
        if self.use_lanes:
            self.flush_param_grads()
            self.plan.wait(0, self.plan.record(WGRAD_LANE))

    def zero_fwd_stats(self):
        if self.training:
            call("plyolo_memset_async", self.stat_arena.data_ptr(), 0, self.stat_arena.numel() * 8, None)

    def zero_bwd_stats(self):
        call("plyolo_memset_async", self.bstat_arena.data_ptr(), 0, self.bstat_arena.numel() * 8, None)

    def pack_subtable(self, indices):
        """Device table of a subset of the pack entries (build_pack_table must have filled the gradient pointers), planned for a
        balanced launch of its own.  Returns (device table, total workgroups)."""
        arr = (PackEntry * max(len(indices), 1))()
        for j, i in enumerate(indices):
            arr[j] = self.pack_entries[i][0]
        total = _lib.lib().plyolo_pack_plan(C.cast(arr, C.c_void_p), max(len(indices), 1))
        t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        self.keep.append(t)
        return t, total

    def build_pack_table(self, grad_ptr_of):
        """Device copy of the PackEntry table; `grad_ptr_of(param)` -> device address of
        that parameter's gradient (inside the flat gradient buffer) or None."""
        n = len(self.pack_entries)
        arr = (PackEntry * max(n, 1))()
        for i, (e, w, b) in enumerate(self.pack_entries):
            e.dw = grad_ptr_of(w)
            e.db = grad_ptr_of(b) if b is not None else None
            arr[i] = e
        # workgroups of the one-launch weight packing / gradient unpacking in proportion to every entry's size
        self.pack_blocks = _lib.lib().plyolo_pack_plan(C.cast(arr, C.c_void_p), max(n, 1)) if n else 0
        raw = bytes(arr)
        host = torch.frombuffer(bytearray(raw), dtype=torch.uint8)
        self.pack_table = host.to(self.device)
        self.n_pack = n

    def pack_weights(self):
        call("plyolo_pack_weights_flat", self.pack_table.data_ptr(), self.n_pack, self.dtype, self.pack_blocks, None)

    def unpack_wgrads(self):
        call("plyolo_unpack_wgrads_flat", self.pack_table.data_ptr(), self.n_pack, self.pack_blocks, 0, None)

    # ----------------------------------------------------------------- gradients
    def grad_storage(self, st):
        if st.grad is None:
            st.grad = torch.empty(st.rows * st.ld, dtype=self.tdtype, device=self.device)
        return st.grad

    def aptr(self, a):
        return a.storage.tensor.data_ptr() + a.c_off * self.esize

    def gptr(self, a):
        return self.grad_storage(a.storage).data_ptr() + a.c_off * self.esize

    def grad_ready(self, a):
        s = a.storage.ginit[a.c_off:a.c_off + a.C]
        return all(s)

    def grad_mode(self, a):
        """Prepare writing a gradient contribution into view `a`: returns the accumulate
        flag (0 = first writer).  A partially initialised view gets its missing channels
        zero-filled first so that a single accumulate launch is correct."""
        st = a.storage
        if self.cur_op is not None:
            self.gm_log.append((self.cur_op, a))
        flags = st.ginit[a.c_off:a.c_off + a.C]
        if not any(flags):
            acc = 0
        elif all(flags):
            acc = 1
        else:
            c = 0
            while c < a.C:
                if not flags[c]:
                    e = c
                    while e < a.C and not flags[e]:
                        e += 1
                    assert c % self.vec == 0 and (e - c) % self.vec == 0
                    call("plyolo_copy_add", self.dtype, a.M, e - c, None, 0,
                         self.grad_storage(st).data_ptr() + (a.c_off + c) * self.esize, st.ld, 0, None)
                    c = e
                else:
                    c += 1
            acc = 1
        for i in range(a.c_off, a.c_off + a.C):
            st.ginit[i] = True
        return acc


WGRAD_LANE = 1      # weight gradients (+ their slab reductions); branch lanes are 2, 3, ...
WGRAD_BATCH = int(os.environ.get("PLYOLO_WGRAD_BATCH", "1"))   # conv units per hand-off event to the weight-gradient lane (measured: 1 == 4 > 8)


class _OnLane:
    def __init__(self, g, lane, lane_fwd):
        self.g, self.lane, self.lane_fwd = g, lane, lane_fwd

    def __enter__(self):
        self.prev = (self.g.cur_lane, self.g.cur_lane_fwd)
        self.g.cur_lane, self.g.cur_lane_fwd = self.lane, self.lane_fwd
        return self

    def __exit__(self, *exc):
        self.g.cur_lane, self.g.cur_lane_fwd = self.prev
        return False


class _Fork:
    def __init__(self, g):
        self.g = g

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def branch(self, lane):
        return self.g.on_lane(lane)


def _res(a):
    """Resource key (buffer, lo, hi) of what an op touches: a channel range of an activation Storage, or a token tuple."""
    if isinstance(a, Act):
        return (id(a.storage), a.c_off, a.c_off + a.C)
    return a


def _overlap(r1, r2):
    return r1[0] == r2[0] and r1[1] < r2[2] and r2[1] < r1[2]


def op_io(op):
    """(inputs, outputs) of a launch-plan op as activation views / resource tokens.  The forward reads the inputs and
    writes the outputs; the backward reads the GRADIENTS of the outputs and writes / accumulates the gradients of the
    inputs -- the same resources with the roles swapped, which is all record_ops() needs."""
    if isinstance(op, CopyOp):
        return [op.src], [op.dst]
    if isinstance(op, ConvPairOp):
        return [op.x], [op.out_a, op.out_b]
    if isinstance(op, (ConvUnitOp, DwConvUnitOp, BnOnlyOp)):
        return [op.x] + ([op.res] if op.res is not None else []), [op.out]
    if isinstance(op, (BicubicUpsampleOp, ActOp, LnWidthOp, UpsampleOp, MaxPool2x2Op)):
        return [op.x], [op.out]
    if isinstance(op, SppPoolsOp):
        return [op.x, ("scratch32", 0, 1)], list(op.outs)
    if isinstance(op, ImplicitHeadOp):
        return [op.x], [(("head", id(op.head)), op.level, op.level + 1)]
    if isinstance(op, HeadPredOp):
        return [op.cls_feat, op.reg_feat], [(("head", id(op.head)), op.level, op.level + 1)]
    if isinstance(op, (YoloxLossOp, YoloV7LossOp, YoloxEvalDecodeOp, YoloV7EvalDecodeOp)):
        return [(("head", id(op.head)), 0, 1 << 20)], []
    raise TypeError("record_ops: op_io() does not know %s" % type(op).__name__)


def record_ops(g, plan, ops, method, lanes=True, after=None):
    """Record `op.<method>()` for every op of `ops` (forward order, or reversed for the backward plan), each on its
    lane, with exactly the cross-lane events the data flow needs.

    `ops` is a valid serial order.  Two ops conflict when one writes a resource the other touches (op_io; in a backward
    plan a gradient that several consumers accumulate into is written by all of them, so they stay ordered as listed,
    which is also the order the first-writer / accumulate flags are resolved in).  Conflicts between ops of ONE lane are
    ordered by the lane; for a conflict across lanes the later op waits for an event recorded right behind the earlier
    one.  The issue order is a list schedule: the lanes take turns (lane 0 first) and a lane whose next op still waits
    for an unrecorded producer is skipped -- every lane gets launches while the host is still issuing the others, and
    the recorded order stays a valid serialisation (single-stream replays, hipGraph capture, the profiler).
    `after(i)` (data-parallel bucket schedule) is called whenever every op with forward index >= i has been recorded."""
    n = len(ops)
    use = lanes and g.use_lanes
    lane = [((o.lane_fwd if method == "fwd" else o.lane) if use else 0) for o in ops]
    g.side_lanes = sorted(set(lane) - {0})
    done = [False] * len(g.ops)
    bound = [len(g.ops)]

    def mark(op):
        if after is None:
            return
        done[op.index] = True
        b = bound[0]
        while b > 0 and done[b - 1]:
            b -= 1
        if b != bound[0]:
            bound[0] = b
            after(b)

    if not g.side_lanes:
        for op in ops:
            plan.lane(0)
            g.cur_op = op
            getattr(op, method)()
            g.cur_op = None
            mark(op)
        plan.lane(0)
        g.flush_reduce_on_lane()
        return
    bwd = method == "bwd"
    rd, wr = [], []
    for o in ops:
        ins, outs = op_io(o)
        ins, outs = [_res(a) for a in ins], [_res(a) for a in outs]
        rd.append(outs if bwd else ins)
        wr.append(ins if bwd else outs)

    def conflict(j, i):
        for w in wr[j]:
            for t in rd[i] + wr[i]:
                if _overlap(w, t):
                    return True
        for w in wr[i]:
            for t in rd[j]:
                if _overlap(w, t):
                    return True
        return False

    # deps[i]: for every other lane the LATEST earlier op there that conflicts with op i (older ones are implied by the
    # lane's order), unless this lane already waited for that lane at or beyond it
    deps, need_ev = [[] for _ in range(n)], [False] * n
    synced = {}
    for i in range(n):
        li = lane[i]
        latest = {}
        for j in range(i - 1, -1, -1):
            lj = lane[j]
            if lj == li or lj in latest:
                continue
            if conflict(j, i):
                latest[lj] = j
        for lj, j in latest.items():
            if synced.get((li, lj), -1) >= j:
                continue
            synced[(li, lj)] = j
            deps[i].append(j)
            need_ev[j] = True
    queues = {}
    for i in range(n):
        queues.setdefault(lane[i], []).append(i)
    order = sorted(queues)
    plan.lane(0)
    ev0 = plan.record(0)         # whatever the plan recorded before its ops (weight packing, stat-slot fills)
    started = {0}
    issued, ev = [False] * n, [None] * n
    left = n
    while left:
        progressed = False
        for l in order:
            q = queues[l]
            if not q or not all(issued[j] for j in deps[q[0]]):
                continue
            i = q.pop(0)
            plan.lane(l)
            if l not in started:
                started.add(l)
                plan.wait(l, ev0)
            for j in deps[i]:
                plan.wait(l, ev[j])
            g.cur_op = ops[i]
            getattr(ops[i], method)()
            g.cur_op = None
            plan.lane(l)
            if need_ev[i]:
                ev[i] = plan.record(l)
            issued[i] = True
            left -= 1
            progressed = True
            mark(ops[i])
        assert progressed, "record_ops: the op list is not a valid serial order"
    for l in order:                 # what follows the ops on lane 0 (slab folds, unpack, optimizer) sees every lane's results
        if l != 0:
            g.flush_param_grads(l)
            plan.wait(0, plan.record(l))
    plan.lane(0)
    g.flush_reduce_on_lane()   # slab folds still queued (backward plans): nothing may read slab 0 before them


class PackedConv:
    """Bookkeeping for one packed convolution (possibly fed by several torch convs)."""

    def __init__(self, g, sources, ksize, Cin_p, need_dgrad=True):
        self.g = g
        self.sources = sources  # list of (weight Parameter, bias Parameter|None, co_off)
        self.ksize, self.Cin_p = ksize, Cin_p
        self.Cout_total = sum(w.shape[0] for (w, _, _) in sources)
        self.Cout_p8 = (self.Cout_total + 7) // 8 * 8
        taps = ksize * ksize
        self.dwp_elems = taps * self.Cout_total * Cin_p            # wgrad slab (plain [tap][Cout][Cin_p])
        self.wp_elems, self.wpd_elems = _lib.pack_elems(g.dtype, self.Cout_total, Cin_p, ksize)
        self.need_dgrad = need_dgrad
        self.nslab = 1
        self.wp = self.wpd = self.dwp = self.bp = self.dbp = None
        g.convs.append(self)

    def reduce_slabs(self):
        """Sum the private wgrad slabs into slab 0 right behind the wgrad launch, on the weight-gradient lane
        (PLYOLO_REDUCE_SLABS=0: leave all slabs to the final unpack launch on the main lane instead).
        Measured on YOLOX-s B=32: 2356 vs 2327 img/s -- the 80 small reductions are hidden beside the main
        lane, the single slab-summing unpack (0.6 ms) is not."""
        if self.g.reduce_slabs and self.nslab > 1:
            if self.g.reduce_batch > 1:
                self.g.queue_reduce(self)
            else:
                call("plyolo_reduce_slabs", self.dwp, self.nslab, self.dwp_elems, None)

    def set_slabs(self, desc):
        """Number of private wgrad slabs the backward launch of `desc` writes."""
        n = _lib.lib().plyolo_conv2d_wgrad_slabs(C.byref(desc))
        if n <= 0:
            _lib.check(-1, "plyolo_conv2d_wgrad_slabs")
        self.nslab = n


def conv_desc(g, N, H, W, Cin, Cout, k, stride, x_ld, y_ld, y_f32=0):
    d = ConvDesc()
    d.dtype, d.N, d.H, d.W, d.Cin, d.Cout = g.dtype, N, H, W, Cin, Cout
    d.ksize, d.stride, d.x_ld, d.y_ld, d.y_f32 = k, stride, x_ld, y_ld, y_f32
    return d


# ------------------------------------------------------------------------- ops
def _copy_desc(d):
    """A by-value copy of a ctypes descriptor."""
    c = type(d)()
    C.memmove(C.byref(c), C.byref(d), C.sizeof(d))
    return c


class CopyOp:
    def __init__(self, g, src, dst):
        self.g, self.src, self.dst = g, src, dst
        src.needs_tensor = True

    def fwd(self):
        g = self.g
        call("plyolo_copy_add", g.dtype, self.src.M, self.src.C, g.aptr(self.src), self.src.ld, g.aptr(self.dst), self.dst.ld, 0, None)

    def bwd(self):
        g = self.g
        if not g.grad_ready(self.dst):
            return
        acc = g.grad_mode(self.src)
        call("plyolo_copy_add", g.dtype, self.src.M, self.src.C, g.gptr(self.dst), self.dst.ld, g.gptr(self.src), self.src.ld, acc, None)


class ConvUnitOp:
    """BaseConv: conv (no bias) -> BatchNorm (batch stats in training) -> activation
    [+ residual].  reference models/layers/network_blocks.py:7-40, :86-90."""

    def __init__(self, g, x, conv_w, bn, act, stride, residual=None, need_dgrad=True, cin_pad=None, conv_b=None):
        self.g, self.x, self.bn, self.act, self.stride, self.res = g, x, bn, ACT[act], stride, residual
        self.conv_b = conv_b   # deploy form (BatchNorm folded into the convolution: BaseConv.fuse / RepConv.fuse_repvgg_block)
        Cout, Cin, k, _ = conv_w.shape
        self.k = k
        self.Cin_p = cin_pad or Cin
        assert x.C == self.Cin_p, "conv input channels %d != %d" % (x.C, self.Cin_p)
        pad = (k - 1) // 2
        self.OH = (x.H + 2 * pad - k) // stride + 1
        self.OW = (x.W + 2 * pad - k) // stride + 1
        self.pc = PackedConv(g, [(conv_w, conv_b, 0)], k, self.Cin_p, need_dgrad)
        self.need_dgrad = need_dgrad
        self.out = g.new_act(x.N, self.OH, self.OW, Cout, "a")
        self.out.producer = self
        self.lazy_out = False          # set by Graph.resolve_lazy(): the activated output is never written
        self.red_done = False          # set by Graph.plan_bn_red(): the data gradient(s) behind `out` fold this unit's bn_act_bwd_reduce
        x.lazy_users.append(self)      # this unit can apply its producer's BatchNorm + activation while staging x
        if residual is not None:
            residual.needs_tensor = True
        self.z = Storage(x.N, self.OH, self.OW, Cout, "z")
        g.storages.append(self.z)
        self.Cout = Cout
        self.desc = conv_desc(g, x.N, x.H, x.W, self.Cin_p, Cout, k, stride, self.Cin_p, Cout)
        self.pc.set_slabs(self.desc)
        # whole backward in one launch (pointwise units of the large maps): its slab count replaces the weight-gradient kernel's
        self.pw_slabs = g.pw_bwd_slabs(self.desc, self.act, bn is not None and need_dgrad and conv_b is None)
        if self.pw_slabs:
            self.pc.nslab = self.pw_slabs
        g.scratch_elems = max(g.scratch_elems, self.z.rows * Cout)
        g.add_op(self)

    def lazy_outputs(self):
        """[(activation view, channel offset inside z)] that may stay lazy: a residual is added by bn_act_fwd, so a unit
        with a shortcut always writes its output."""
        return [(self.out, 0)] if (self.bn is not None and self.res is None and self.g.training and self.act <= ACT["lrelu"]) else []

    def _alloc_small(self):
        g = self.g
        if hasattr(self, "coef"):
            return
        # desc: forward + weight gradient (x as stored, or the producer's z with its coefficients); desc_d: data gradient
        # (its x_ld is the pitch of the GRADIENT matrix of x, which a lazy input still owns)
        self.xptr, self.desc.x_ld = g.src(self.x)
        g.set_lazy(self.desc, self.x)
        self.desc_d = _copy_desc(self.desc)
        self.desc_d.x_ld = self.x.ld
        self.desc_d.x_coef, self.desc_d.x_coef_ld, self.desc_d.x_act = None, 0, 0
        self.coef = torch.empty(4 * self.Cout, dtype=torch.float32, device=g.device)

    def fwd(self):
        g, bn = self.g, self.bn
        self._alloc_small()
        zt = self.z.tensor
        train_stats = g.training and bn is not None
        if (not g.training) and (bn is not None or self.conv_b is not None) and g.dtype == BF16 and g.fuse_eval and self.act <= ACT["lrelu"]:
            # inference: BatchNorm is a fixed affine -> applied with the activation in the conv epilogue; the
            # activated tensor is written straight into its (possibly concat-slice) destination.  The deploy form
            # (BatchNorm already folded into weights + bias) is the same epilogue with scale 1, shift = bias
            if bn is not None:
                call("plyolo_bn_eval_coef", self.Cout, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                     float(bn.eps), self.coef.data_ptr(), None)
            else:
                call("plyolo_bias_coef", self.Cout, self.pc.bp, self.coef.data_ptr(), None)
            if not hasattr(self, "desc_eval"):
                self.desc_eval = conv_desc(g, self.desc.N, self.desc.H, self.desc.W, self.Cin_p, self.Cout, self.k, self.stride,
                                           self.x.ld, self.out.ld)
            call("plyolo_conv2d_fwd_bn_act", C.byref(self.desc_eval), g.aptr(self.x), self.pc.wp, self.coef.data_ptr(), self.act,
                 g.aptr(self.res) if self.res is not None else None, self.res.ld if self.res is not None else 0, g.aptr(self.out), None)
            return
        slots = g.stat_arena.data_ptr() + self.slot_off * 8 if train_stats else None
        call("plyolo_conv2d_fwd", C.byref(self.desc), self.xptr, self.pc.wp, self.pc.bp if self.conv_b is not None else None,
             zt.data_ptr(), slots, None)
        coef, st = None, None
        if bn is not None:
            coef = self.coef.data_ptr()
            if g.training:   # scale/shift derived from the stat slots inside bn_act_fwd (no finalize launch)
                st = BnStats()
                st.slots, st.count = slots, float(self.out.M)
                st.gamma, st.beta = ptr(bn.weight), ptr(bn.bias)
                st.eps, st.momentum = float(bn.eps), float(bn.momentum)
                st.running_mean, st.running_var = ptr(bn.running_mean), ptr(bn.running_var)
                st.num_batches_tracked = ptr(bn.num_batches_tracked)
                if self.lazy_out:
                    # every reader applies scale/shift + activation itself while staging z: only the per-channel
                    # coefficients (and the running statistics) are produced here -- one tiny launch instead of a
                    # read-z / write-a pass over the whole tensor
                    call("plyolo_bn_finalize", C.byref(st), self.Cout, coef, None)
                    return
            else:
                call("plyolo_bn_eval_coef", self.Cout, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean),
                     ptr(bn.running_var), float(bn.eps), coef, None)
        call("plyolo_bn_act_fwd", g.dtype, self.out.M, self.Cout, zt.data_ptr(), self.Cout, coef, self.act,
             g.aptr(self.res) if self.res is not None else None, self.res.ld if self.res is not None else 0,
             g.aptr(self.out), self.out.ld, C.byref(st) if st is not None else None, None, None)

    def bwd(self):
        g, bn = self.g, self.bn
        if not g.grad_ready(self.out):
            return  # nothing downstream contributes a gradient
        if bn is None and self.conv_b is not None:
            raise NotImplementedError("training the deploy form (conv + bias, BatchNorm folded) is not supported by the HIP plan")
        M, Cout = self.out.M, self.Cout
        dout, zt = g.gptr(self.out), self.z.tensor.data_ptr()
        plan, lanes, me = g.plan, g.use_lanes, self.lane
        pw_one = bool(self.pw_slabs)          # dz + data gradient + weight gradient in one launch (plyolo_conv2d_bwd_pw)
        # no data gradient at all (the first convolution): dz is only read by the weight gradient -- formed in ITS loader, never stored
        wg_bn = (bn is not None and not pw_one and g.fuse_wgbn and not self.need_dgrad and self.res is None
                 and _lib.lib().plyolo_conv2d_wgrad_bn_fits(C.byref(self.desc), self.act) == 1)
        dz, key = (None, None) if (pw_one or wg_bn) else g.dz_buffer(self, M * Cout)
        # the shortcut's share of the gradient (network_blocks.py:89-90): forwarded by the bn_act_bwd_dz pass that reads dout anyway;
        # units without that pass (no BatchNorm, fused pointwise path) copy it with a launch of its own
        fused = bn is not None and not pw_one and self.need_dgrad and g.dgrad_bn_fits(self.desc_d, self.act)
        res_in_dz = self.res is not None and bn is not None and g.fwd_res_in_dz and not pw_one and not fused
        res_in_dgrad = False      # a fused 3x3 data gradient copies the shortcut's share while it reads dout (first writer only)
        if self.res is not None:
            acc_res = g.grad_mode(self.res)
            res_in_dgrad = fused and self.k == 3 and acc_res == 0 and g.fwd_res_in_dz
            if not (res_in_dz or res_in_dgrad):
                call("plyolo_copy_add", g.dtype, M, Cout, dout, self.out.ld, g.gptr(self.res), self.res.ld, acc_res, None)
        if bn is None:
            # BaseConv(norm=None): out = act(z)  (ecmnet.py:158 Bottleneck.conv1) -> dz = dout * act'(z)
            call("plyolo_act_bwd", g.dtype, M, Cout, dout, self.out.ld, zt, Cout, self.act, dz, Cout, 0, None)
        else:
            bslots = g.bstat_arena.data_ptr() + self.slot_off * 8
            if not (g.diag_skip_r or self.red_done):
                call("plyolo_bn_act_bwd_reduce", g.dtype, M, Cout, dout, self.out.ld, zt, Cout, self.coef.data_ptr(), self.act, bslots, None, None)
            if pw_one:
                f = BnBwdFuse()
                f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout, self.out.ld, zt, Cout, self.coef.data_ptr(), bslots
                f.gamma, f.dgamma, f.dbeta = ptr(bn.weight), g.grad_ptr_of(bn.weight), g.grad_ptr_of(bn.bias)
                f.act = self.act
                self.keep_f = f
                acc = g.grad_mode(self.x)
                red = g.red_for(self, self.x)
                call("plyolo_conv2d_bwd_pw_red", C.byref(self.desc_d), C.byref(f), self.xptr, self.pc.wpd, g.gptr(self.x), acc, self.pc.dwp,
                     C.byref(red) if red is not None else None, None)
                if lanes:     # only the slab fold is left for the weight-gradient lane
                    g.defer_param_grads(me, self.pc.reduce_slabs)
                else:
                    self.pc.reduce_slabs()
                return
            # pointwise and 3x3 stride-1 units: dz is formed inside the data gradient's loader (one launch and one pass over dout / z less)
            if wg_bn:
                f = BnBwdFuse()
                f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout, self.out.ld, zt, Cout, self.coef.data_ptr(), bslots
                f.gamma, f.dgamma, f.dbeta = ptr(bn.weight), g.grad_ptr_of(bn.weight), g.grad_ptr_of(bn.bias)
                f.act = self.act
                self.keep_f = f

                def wgrad_bn():
                    call("plyolo_conv2d_wgrad_bn", C.byref(self.desc), C.byref(f), self.xptr, self.pc.dwp, None)
                    self.pc.reduce_slabs()
                if lanes:
                    g.defer_param_grads(me, wgrad_bn)
                else:
                    wgrad_bn()
                return
            if fused:
                f = BnBwdFuse()
                f.dout, f.dout_ld, f.z, f.z_ld, f.coef, f.bslots = dout, self.out.ld, zt, Cout, self.coef.data_ptr(), bslots
                f.gamma, f.dgamma, f.dbeta = ptr(bn.weight), g.grad_ptr_of(bn.weight), g.grad_ptr_of(bn.bias)
                f.act, f.dz, f.dz_ld = self.act, dz, Cout
                if res_in_dgrad:
                    f.fwd_to, f.fwd_ld = g.gptr(self.res), self.res.ld
                self.keep_f = f
            else:
                sp = None
                if res_in_dz:
                    sp = Split()
                    sp.fwd_to, sp.fwd_ld, sp.fwd_acc = g.gptr(self.res), self.res.ld, acc_res
                call("plyolo_bn_act_bwd_dz", g.dtype, M, Cout, dout, self.out.ld, zt, Cout, self.coef.data_ptr(), bslots, ptr(bn.weight),
                     g.grad_ptr_of(bn.weight), g.grad_ptr_of(bn.bias), 0, self.act, dz, Cout, C.byref(sp) if sp is not None else None, None, None)
        def dgrad():
            if self.need_dgrad:
                acc = g.grad_mode(self.x)
                red = g.red_for(self, self.x)
                if fused:
                    call("plyolo_conv2d_dgrad_bn_red", C.byref(self.desc_d), C.byref(f), self.pc.wpd, g.gptr(self.x), acc,
                         C.byref(red) if red is not None else None, None)
                else:
                    call("plyolo_conv2d_dgrad_red", C.byref(self.desc_d), dz, self.pc.wpd, g.gptr(self.x), acc,
                         C.byref(red) if red is not None else None, None)

        def wgrad():
            call("plyolo_conv2d_wgrad", C.byref(self.desc), self.xptr, dz, self.pc.dwp, None)
            self.pc.reduce_slabs()

        if lanes:
            # the weight gradient only feeds the optimizer: it runs on its own lane, concurrently with the
            # data-gradient chain of the layers upstream
            dgrad()
            g.defer_param_grads(me, wgrad)
        else:
            if fused:   # the fused data gradient produces dz: it has to run first
                dgrad()
                wgrad()
            else:
                wgrad()
                dgrad()


class ConvPairOp:
    """Two BaseConv units with the SAME input, kernel size, stride and activation (CSPLayer conv1 || conv2,
    network_blocks.py:108-110,123-124; the ELAN conv1 || conv2 pairs) run as ONE convolution with the
    output channels concatenated: one conv + one bn_act launch forward, one reduce / dz / dgrad / wgrad
    backward (the input gradient is written once instead of overwrite + accumulate), x is read once.
    The two BatchNorm modules keep their own parameters (second parameter set of the BN kernels) and the
    two activated outputs go to their own matrices (channel-split store)."""

    def __init__(self, g, x, conv_a, bn_a, conv_b, bn_b, act, stride=1):
        self.g, self.x, self.bn_a, self.bn_b, self.act, self.stride = g, x, bn_a, bn_b, ACT[act], stride
        Ca, Cin, k, _ = conv_a.shape
        Cb = conv_b.shape[0]
        assert tuple(conv_b.shape[1:]) == (Cin, k, k) and x.C == Cin and Ca % g.vec == 0 and Cb % g.vec == 0
        self.k, self.Cin_p, self.Ca, self.Cb, self.Cout = k, Cin, Ca, Cb, Ca + Cb
        pad = (k - 1) // 2
        self.OH = (x.H + 2 * pad - k) // stride + 1
        self.OW = (x.W + 2 * pad - k) // stride + 1
        self.pc = PackedConv(g, [(conv_a, None, 0), (conv_b, None, Ca)], k, Cin, True)
        self.out_a = g.new_act(x.N, self.OH, self.OW, Ca, "a")
        self.out_b = g.new_act(x.N, self.OH, self.OW, Cb, "a")
        self.out_a.producer = self.out_b.producer = self
        self.lazy_out = False
        self.red_done = False
        x.lazy_users.append(self)
        self.out = self.out_a
        self.z = Storage(x.N, self.OH, self.OW, self.Cout, "z")
        g.storages.append(self.z)
        self.desc = conv_desc(g, x.N, x.H, x.W, Cin, self.Cout, k, stride, Cin, self.Cout)
        self.pc.set_slabs(self.desc)
        self.pw_slabs = g.pw_bwd_slabs(self.desc, self.act)
        if self.pw_slabs:
            self.pc.nslab = self.pw_slabs
        g.scratch_elems = max(g.scratch_elems, self.z.rows * self.Cout)
        self.need_dgrad, self.bn, self.res = True, bn_a, None
        g.add_op(self)

    def lazy_outputs(self):
        """Both halves or none (one bn_act launch writes both)."""
        return [(self.out_a, 0), (self.out_b, self.Ca)] if (self.g.training and self.act <= ACT["lrelu"]) else []

    def _split(self, base_ptr_fn, act_b):
        sp = Split()
        sp.split, sp.p2, sp.ld2 = self.Ca, base_ptr_fn(act_b), act_b.ld
        return sp

    def fwd(self):
        g, a, b = self.g, self.bn_a, self.bn_b
        if not hasattr(self, "coef"):
            self.xptr, self.desc.x_ld = g.src(self.x)
            g.set_lazy(self.desc, self.x)
            self.desc_d = _copy_desc(self.desc)
            self.desc_d.x_ld = self.x.ld
            self.desc_d.x_coef, self.desc_d.x_coef_ld, self.desc_d.x_act = None, 0, 0
            self.coef = torch.empty(4 * self.Cout, dtype=torch.float32, device=g.device)
        zt = self.z.tensor
        slots = g.stat_arena.data_ptr() + self.slot_off * 8 if g.training else None
        call("plyolo_conv2d_fwd", C.byref(self.desc), self.xptr, self.pc.wp, None, zt.data_ptr(), slots, None)
        st = None
        if g.training:
            st = BnStats()
            st.slots, st.count = slots, float(self.out_a.M)
            st.gamma, st.beta, st.gamma2, st.beta2 = ptr(a.weight), ptr(a.bias), ptr(b.weight), ptr(b.bias)
            st.eps, st.momentum, st.split = float(a.eps), float(a.momentum), self.Ca
            st.running_mean, st.running_var, st.num_batches_tracked = ptr(a.running_mean), ptr(a.running_var), ptr(a.num_batches_tracked)
            st.running_mean2, st.running_var2, st.num_batches_tracked2 = ptr(b.running_mean), ptr(b.running_var), ptr(b.num_batches_tracked)
            if self.lazy_out:
                call("plyolo_bn_finalize", C.byref(st), self.Cout, self.coef.data_ptr(), None)
                return
        else:
            for bn, off, n in ((a, 0, self.Ca), (b, self.Ca, self.Cb)):
                call("plyolo_bn_eval_coef_at", n, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                     float(bn.eps), self.coef.data_ptr(), self.Cout, off, None)
        sp = self._split(g.aptr, self.out_b)
        call("plyolo_bn_act_fwd", g.dtype, self.out_a.M, self.Cout, zt.data_ptr(), self.Cout, self.coef.data_ptr(), self.act,
             None, 0, g.aptr(self.out_a), self.out_a.ld, C.byref(st) if st is not None else None, C.byref(sp), None)

    def bwd(self):
        g, a, b = self.g, self.bn_a, self.bn_b
        ra, rb = g.grad_ready(self.out_a), g.grad_ready(self.out_b)
        if not (ra or rb):
            return
        if not (ra and rb):
            raise NotImplementedError("ConvPairOp: both outputs must receive a gradient")
        M, Cout = self.out_a.M, self.Cout
        zt = self.z.tensor.data_ptr()
        dsp = self._split(g.gptr, self.out_b)
        bslots = g.bstat_arena.data_ptr() + self.slot_off * 8
        if not (g.diag_skip_r or self.red_done):
            call("plyolo_bn_act_bwd_reduce", g.dtype, M, Cout, g.gptr(self.out_a), self.out_a.ld, zt, Cout, self.coef.data_ptr(), self.act,
                 bslots, C.byref(dsp), None)
        plan, lanes, me = g.plan, g.use_lanes, self.lane
        if self.pw_slabs:    # pointwise pair (CSP conv1 || conv2) of a large map: dz + data gradient + weight gradient in one launch
            f = BnBwdFuse()
            f.dout, f.dout_ld, f.dout2, f.dout2_ld, f.dout_split = g.gptr(self.out_a), self.out_a.ld, dsp.p2, dsp.ld2, self.Ca
            f.z, f.z_ld, f.coef, f.bslots = zt, Cout, self.coef.data_ptr(), bslots
            f.gamma, f.dgamma, f.dbeta = ptr(a.weight), g.grad_ptr_of(a.weight), g.grad_ptr_of(a.bias)
            f.par_split, f.gamma2, f.dgamma2, f.dbeta2 = self.Ca, ptr(b.weight), g.grad_ptr_of(b.weight), g.grad_ptr_of(b.bias)
            f.act = self.act
            self.keep_f = f
            acc = g.grad_mode(self.x)
            red = g.red_for(self, self.x)
            call("plyolo_conv2d_bwd_pw_red", C.byref(self.desc_d), C.byref(f), self.xptr, self.pc.wpd, g.gptr(self.x), acc, self.pc.dwp,
                 C.byref(red) if red is not None else None, None)
            if lanes:
                g.defer_param_grads(me, self.pc.reduce_slabs)
            else:
                self.pc.reduce_slabs()
            return
        dz, key = g.dz_buffer(self, M * Cout)
        fused = g.dgrad_bn_fits(self.desc_d, self.act) and (self.k != 3 or self.Ca % 32 == 0)   # (3x3 loader: whole 32-channel chunks per matrix)
        if fused:    # pair (CSP conv1 || conv2, the first 3x3 round of a head level): dz is formed inside the data gradient's loader
            f = BnBwdFuse()
            f.dout, f.dout_ld, f.dout2, f.dout2_ld, f.dout_split = g.gptr(self.out_a), self.out_a.ld, dsp.p2, dsp.ld2, self.Ca
            f.z, f.z_ld, f.coef, f.bslots = zt, Cout, self.coef.data_ptr(), bslots
            f.gamma, f.dgamma, f.dbeta = ptr(a.weight), g.grad_ptr_of(a.weight), g.grad_ptr_of(a.bias)
            f.par_split, f.gamma2, f.dgamma2, f.dbeta2 = self.Ca, ptr(b.weight), g.grad_ptr_of(b.weight), g.grad_ptr_of(b.bias)
            f.act, f.dz, f.dz_ld = self.act, dz, Cout
            self.keep_f = f
        else:
            p2 = BnBwdSplit()
            p2.split, p2.gamma2, p2.dgamma2, p2.dbeta2 = self.Ca, ptr(b.weight), g.grad_ptr_of(b.weight), g.grad_ptr_of(b.bias)
            call("plyolo_bn_act_bwd_dz", g.dtype, M, Cout, g.gptr(self.out_a), self.out_a.ld, zt, Cout, self.coef.data_ptr(), bslots,
                 ptr(a.weight), g.grad_ptr_of(a.weight), g.grad_ptr_of(a.bias), 0, self.act, dz, Cout, C.byref(dsp), C.byref(p2), None)
        acc = g.grad_mode(self.x)

        def wgrad():
            call("plyolo_conv2d_wgrad", C.byref(self.desc), self.xptr, dz, self.pc.dwp, None)
            self.pc.reduce_slabs()

        def dgrad():
            red = g.red_for(self, self.x)
            if fused:
                call("plyolo_conv2d_dgrad_bn_red", C.byref(self.desc_d), C.byref(f), self.pc.wpd, g.gptr(self.x), acc,
                     C.byref(red) if red is not None else None, None)
            else:
                call("plyolo_conv2d_dgrad_red", C.byref(self.desc_d), dz, self.pc.wpd, g.gptr(self.x), acc,
                     C.byref(red) if red is not None else None, None)

        if lanes:
            dgrad()
            g.defer_param_grads(me, wgrad)
        else:
            if fused:     # the fused data gradient produces dz: it has to run first
                dgrad()
                wgrad()
            else:
                wgrad()
                dgrad()


class DwConvUnitOp:
    """BaseConv with a depthwise 3x3 convolution (groups == channels, stride 1): conv -> BatchNorm -> activation [+ residual]
    (reference models/backbones/ecmnet.py:157,160 / models/necks/pafpn_al.py:162,165: Bottleneck.conv0 / conv3).  An HBM stream
    (csrc/dwconv.hip), not an MFMA contraction; the fp32 master weight [C,1,3,3] is read directly, its gradient written directly."""

    def __init__(self, g, x, conv_w, bn, act, residual=None):
        self.g, self.x, self.w, self.bn, self.act, self.res = g, x, conv_w, bn, ACT[act], residual
        Cc = conv_w.shape[0]
        assert tuple(conv_w.shape) == (Cc, 1, 3, 3) and x.C == Cc and Cc % g.vec == 0, "depthwise 3x3: [C,1,3,3] weights, C %% %d == 0" % g.vec
        if bn is None:
            raise NotImplementedError("depthwise conv unit without BatchNorm")
        self.Cout = Cc
        x.needs_tensor = True
        if residual is not None:
            residual.needs_tensor = True
        self.out = g.new_act(x.N, x.H, x.W, Cc, "a")
        self.z = Storage(x.N, x.H, x.W, Cc, "z")
        g.storages.append(self.z)
        g.scratch_elems = max(g.scratch_elems, self.z.rows * Cc)
        g.add_op(self)

    def fwd(self):
        g, bn, x = self.g, self.bn, self.x
        if not hasattr(self, "coef"):
            self.coef = torch.empty(4 * self.Cout, dtype=torch.float32, device=g.device)
        zt = self.z.tensor.data_ptr()
        slots = g.stat_arena.data_ptr() + self.slot_off * 8 if g.training else None
        call("plyolo_dwconv3x3_fwd", g.dtype, x.N, x.H, x.W, self.Cout, g.aptr(x), x.ld, ptr(self.w), zt, self.Cout, slots, None)
        st = None
        if g.training:
            st = BnStats()
            st.slots, st.count = slots, float(self.out.M)
            st.gamma, st.beta = ptr(bn.weight), ptr(bn.bias)
            st.eps, st.momentum = float(bn.eps), float(bn.momentum)
            st.running_mean, st.running_var = ptr(bn.running_mean), ptr(bn.running_var)
            st.num_batches_tracked = ptr(bn.num_batches_tracked)
        else:
            call("plyolo_bn_eval_coef", self.Cout, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                 float(bn.eps), self.coef.data_ptr(), None)
        call("plyolo_bn_act_fwd", g.dtype, self.out.M, self.Cout, zt, self.Cout, self.coef.data_ptr(), self.act,
             g.aptr(self.res) if self.res is not None else None, self.res.ld if self.res is not None else 0,
             g.aptr(self.out), self.out.ld, C.byref(st) if st is not None else None, None, None)

    def bwd(self):
        g, bn, x = self.g, self.bn, self.x
        if not g.grad_ready(self.out):
            return
        M, Cc = self.out.M, self.Cout
        dout, zt = g.gptr(self.out), self.z.tensor.data_ptr()
        if self.res is not None:
            acc = g.grad_mode(self.res)
            call("plyolo_copy_add", g.dtype, M, Cc, dout, self.out.ld, g.gptr(self.res), self.res.ld, acc, None)
        bslots = g.bstat_arena.data_ptr() + self.slot_off * 8
        call("plyolo_bn_act_bwd_reduce", g.dtype, M, Cc, dout, self.out.ld, zt, Cc, self.coef.data_ptr(), self.act, bslots, None, None)
        dz, _ = g.dz_buffer(self, M * Cc)
        call("plyolo_bn_act_bwd_dz", g.dtype, M, Cc, dout, self.out.ld, zt, Cc, self.coef.data_ptr(), bslots, ptr(bn.weight),
             g.grad_ptr_of(bn.weight), g.grad_ptr_of(bn.bias), 0, self.act, dz, Cc, None, None, None)
        acc = g.grad_mode(x)
        call("plyolo_dwconv3x3_dgrad", g.dtype, x.N, x.H, x.W, Cc, dz, Cc, ptr(self.w), g.gptr(x), x.ld, acc, None)
        if not hasattr(self, "partial"):
            nb = _lib.lib().plyolo_dwconv3x3_wgrad_blocks(g.dtype, x.N, x.H, x.W, Cc)
            self.partial = torch.empty(max(nb, 1) * Cc * 9, dtype=torch.float32, device=g.device)

        def wgrad():
            call("plyolo_dwconv3x3_wgrad", g.dtype, x.N, x.H, x.W, Cc, g.aptr(x), x.ld, dz, Cc, self.partial.data_ptr(),
                 g.grad_ptr_of(self.w), 0, None)

        if g.use_lanes:
            g.defer_param_grads(self.lane, wgrad)
        else:
            wgrad()


class BicubicUpsampleOp:
    """nn.Upsample(scale_factor=2, mode="bicubic") (models/necks/pafpn_al.py:25)."""

    def __init__(self, g, x):
        self.g, self.x = g, x
        x.needs_tensor = True
        self.out = g.new_act(x.N, 2 * x.H, 2 * x.W, x.C, "up3")
        g.add_op(self)

    def fwd(self):
        g, x = self.g, self.x
        call("plyolo_bicubic2x_fwd", g.dtype, x.N, x.H, x.W, x.C, g.aptr(x), x.ld, g.aptr(self.out), self.out.ld, None)

    def bwd(self):
        g, x = self.g, self.x
        if not g.grad_ready(self.out):
            return
        acc = g.grad_mode(x)
        call("plyolo_bicubic2x_bwd", g.dtype, x.N, x.H, x.W, x.C, g.gptr(self.out), self.out.ld, g.gptr(x), x.ld, acc, None)


class ActOp:
    """A bare activation out = act(x) (RepConv applies SiLU to a SUM of BatchNorm outputs,
    yolov7_neck.py:211).  Forward = bn_act_fwd without coefficients; backward din (+)= dout*act'(x)."""

    def __init__(self, g, x, act):
        self.g, self.x, self.act = g, x, ACT[act]
        x.needs_tensor = True
        self.out = g.new_act(x.N, x.H, x.W, x.C, "act")
        g.add_op(self)

    def fwd(self):
        g, x = self.g, self.x
        call("plyolo_bn_act_fwd", g.dtype, x.M, x.C, g.aptr(x), x.ld, None, self.act, None, 0, g.aptr(self.out), self.out.ld, None, None, None)

    def bwd(self):
        g, x = self.g, self.x
        if not g.grad_ready(self.out):
            return
        acc = g.grad_mode(x)
        call("plyolo_act_bwd", g.dtype, x.M, x.C, g.gptr(self.out), self.out.ld, g.aptr(x), x.ld, self.act, g.gptr(x), x.ld, acc, None)


class BnOnlyOp:
    """BatchNorm applied directly to a tensor (no convolution, no activation), optionally + residual:
    the identity branch of RepConv (yolov7_neck.py:191,204-209)."""

    def __init__(self, g, x, bn, residual=None):
        self.g, self.x, self.bn, self.res = g, x, bn, residual
        x.needs_tensor = True
        if residual is not None:
            residual.needs_tensor = True
        self.Cout = x.C
        self.out = g.new_act(x.N, x.H, x.W, x.C, "bn")
        g.scratch_elems = max(g.scratch_elems, x.M * x.C)
        g.add_op(self)

    def fwd(self):
        g, x, bn = self.g, self.x, self.bn
        if not hasattr(self, "coef"):
            self.coef = torch.empty(4 * x.C, dtype=torch.float32, device=g.device)
        st = None
        if g.training:
            slots = g.stat_arena.data_ptr() + self.slot_off * 8
            call("plyolo_channel_stats", g.dtype, x.M, x.C, g.aptr(x), x.ld, slots, None)
            st = BnStats()
            st.slots, st.count = slots, float(x.M)
            st.gamma, st.beta = ptr(bn.weight), ptr(bn.bias)
            st.eps, st.momentum = float(bn.eps), float(bn.momentum)
            st.running_mean, st.running_var = ptr(bn.running_mean), ptr(bn.running_var)
            st.num_batches_tracked = ptr(bn.num_batches_tracked)
        else:
            call("plyolo_bn_eval_coef", x.C, ptr(bn.weight), ptr(bn.bias), ptr(bn.running_mean), ptr(bn.running_var),
                 float(bn.eps), self.coef.data_ptr(), None)
        call("plyolo_bn_act_fwd", g.dtype, x.M, x.C, g.aptr(x), x.ld, self.coef.data_ptr(), 0,
             g.aptr(self.res) if self.res is not None else None, self.res.ld if self.res is not None else 0,
             g.aptr(self.out), self.out.ld, C.byref(st) if st is not None else None, None, None)

    def bwd(self):
        g, x, bn = self.g, self.x, self.bn
        if not g.grad_ready(self.out):
            return
        M, Cc = x.M, x.C
        dout = g.gptr(self.out)
        if self.res is not None:
            acc = g.grad_mode(self.res)
            call("plyolo_copy_add", g.dtype, M, Cc, dout, self.out.ld, g.gptr(self.res), self.res.ld, acc, None)
        bslots = g.bstat_arena.data_ptr() + self.slot_off * 8
        call("plyolo_bn_act_bwd_reduce", g.dtype, M, Cc, dout, self.out.ld, g.aptr(x), x.ld, self.coef.data_ptr(), 0, bslots, None, None)
        plan, me = g.plan, self.lane
        dz, key = g.dz_buffer(self, M * Cc)
        call("plyolo_bn_act_bwd_dz", g.dtype, M, Cc, dout, self.out.ld, g.aptr(x), x.ld, self.coef.data_ptr(), bslots, ptr(bn.weight),
             g.grad_ptr_of(bn.weight), g.grad_ptr_of(bn.bias), 0, 0, dz, Cc, None, None, None)
        acc = g.grad_mode(x)    # here dz IS the gradient w.r.t. the normalised tensor itself
        call("plyolo_copy_add", g.dtype, M, Cc, dz, Cc, g.gptr(x), x.ld, acc, None)


class LnWidthOp:
    """norm = "ln" of BaseConv: nn.LayerNorm(out_channels) over the WIDTH of the conv output, then the activation
    (reference models/layers/normalization.py:9-10, network_blocks.py:36-37).  torch applies LayerNorm((C,)) to the last axis of
    an NCHW tensor, so it only runs when W == C; the same condition is checked here with torch's error."""

    def __init__(self, g, x, ln, act):
        if x.W != ln.normalized_shape[0] or len(ln.normalized_shape) != 1:
            raise RuntimeError("Given normalized_shape=%s, expected input with shape [*, %d], but got input of size [%d, %d, %d, %d]"
                               % (list(ln.normalized_shape), ln.normalized_shape[0], x.N, x.C, x.H, x.W))
        self.g, self.x, self.ln, self.act = g, x, ln, ACT[act]
        self.bn = ln            # the attribute the runner / the bucket schedule read the affine parameters from
        x.needs_tensor = True
        self.out = g.new_act(x.N, x.H, x.W, x.C, "ln")
        g.add_op(self)

    def fwd(self):
        g, x, ln = self.g, self.x, self.ln
        if not hasattr(self, "stats"):
            self.stats = torch.empty(x.N * x.H * x.C * 2, dtype=torch.float32, device=g.device)
        call("plyolo_lnw_act_fwd", g.dtype, x.N, x.H, x.W, x.C, g.aptr(x), x.ld, ptr(ln.weight), ptr(ln.bias), float(ln.eps), self.act,
             g.aptr(self.out), self.out.ld, self.stats.data_ptr(), None)

    def bwd(self):
        g, x, ln = self.g, self.x, self.ln
        if not g.grad_ready(self.out):
            return
        acc = g.grad_mode(x)
        call("plyolo_lnw_act_bwd", g.dtype, x.N, x.H, x.W, x.C, g.gptr(self.out), self.out.ld, g.aptr(x), x.ld, self.stats.data_ptr(),
             ptr(ln.weight), ptr(ln.bias), self.act, g.gptr(x), x.ld, acc, g.grad_ptr_of(ln.weight), g.grad_ptr_of(ln.bias), 0, None)


class UpsampleOp:
    def __init__(self, g, x):
        self.g, self.x = g, x
        x.needs_tensor = True
        self.out = g.new_act(x.N, 2 * x.H, 2 * x.W, x.C, "up")
        g.add_op(self)

    def fwd(self):
        g, x = self.g, self.x
        call("plyolo_upsample2x_fwd", g.dtype, x.N, x.H, x.W, x.C, g.aptr(x), x.ld, g.aptr(self.out), self.out.ld, None)

    def bwd(self):
        g, x = self.g, self.x
        if not g.grad_ready(self.out):
            return
        acc = g.grad_mode(x)
        call("plyolo_upsample2x_bwd", g.dtype, x.N, x.H, x.W, x.C, g.gptr(self.out), self.out.ld, g.gptr(x), x.ld, acc, None)


class SppPoolsOp:
    """MaxPool2d(k, stride 1, pad k//2) for k = 5, 9, 13 (network_blocks.py:141-153).
    Forward runs the exact cascade 5 -> 5 -> 5; backward uses the independent-pool
    rule (first maximum in row-major window order takes the gradient, like ATen)."""

    def __init__(self, g, x, ks=(5, 9, 13)):
        self.g, self.x, self.ks = g, x, tuple(ks)
        x.needs_tensor = True
        self.outs = [g.new_act(x.N, x.H, x.W, x.C, "pool%d" % k) for k in ks]
        g.scratch_f32 = max(g.scratch_f32, x.M * x.C)
        g.add_op(self)

    def fwd(self):
        g, x = self.g, self.x
        nk = len(self.ks)
        ks = (C.c_int * nk)(*self.ks)
        if nk <= 3 and _lib.lib().plyolo_spp_pools_fwd_fits(g.dtype, x.H, x.W, x.C, nk, ks) == 1:
            outs = (C.c_void_p * nk)(*[g.aptr(o) for o in self.outs])
            olds = (C.c_int * nk)(*[o.ld for o in self.outs])
            call("plyolo_spp_pools_fwd", g.dtype, x.N, x.H, x.W, x.C, nk, ks, g.aptr(x), x.ld, outs, olds, None)
            return
        src, prev_k = x, 1
        for k, o in zip(self.ks, self.outs):
            # pool_k(x) == pool_{k-prev+1}(pool_prev(x)) for stride-1 max pools
            call("plyolo_maxpool_s1_fwd", g.dtype, x.N, x.H, x.W, x.C, k - prev_k + 1, g.aptr(src), src.ld, g.aptr(o), o.ld, None)
            src, prev_k = o, k

    def bwd(self):
        g, x = self.g, self.x
        ready = [g.grad_ready(o) for o in self.outs]
        if not any(ready):
            return
        if _lib.lib().plyolo_spp_pools_bwd_fits(g.dtype, x.H, x.W):
            nk = len(self.ks)
            ks = (C.c_int * nk)(*self.ks)
            douts = (C.c_void_p * nk)(*[g.gptr(o) if r else None for o, r in zip(self.outs, ready)])
            dlds = (C.c_int * nk)(*[o.ld for o in self.outs])
            call("plyolo_spp_pools_bwd", g.dtype, x.N, x.H, x.W, x.C, nk, ks, g.aptr(x), x.ld, douts, dlds,
                 g.gptr(x), x.ld, g.grad_mode(x), None)
            return
        n = x.M * x.C
        call("plyolo_memset_async", g.scratch32.data_ptr(), 0, n * 4, None)
        for k, o, r in zip(self.ks, self.outs, ready):
            if r:
                call("plyolo_maxpool_s1_bwd", g.dtype, x.N, x.H, x.W, x.C, k, g.aptr(x), x.ld, g.gptr(o), o.ld,
                     g.scratch32.data_ptr(), None)
        acc = g.grad_mode(x)
        call("plyolo_f32_to_act", g.dtype, x.M, x.C, g.scratch32.data_ptr(), g.gptr(x), x.ld, acc, None)


class MaxPool2x2Op:
    """MaxPool2d(2, 2) of the YOLOv7 Transition blocks."""

    def __init__(self, g, x):
        self.g, self.x = g, x
        x.needs_tensor = True
        self.out = g.new_act(x.N, x.H // 2, x.W // 2, x.C, "mp2")
        g.add_op(self)

    def fwd(self):
        g, x = self.g, self.x
        call("plyolo_maxpool2x2_fwd", g.dtype, x.N, x.H, x.W, x.C, g.aptr(x), x.ld, g.aptr(self.out), self.out.ld, None)

    def bwd(self):
        g, x = self.g, self.x
        if not g.grad_ready(self.out):
            return
        acc = g.grad_mode(x)
        call("plyolo_maxpool2x2_bwd", g.dtype, x.N, x.H, x.W, x.C, g.aptr(x), x.ld, g.gptr(self.out), self.out.ld,
             g.gptr(x), x.ld, acc, None)


class ImplicitHeadOp:
    """One level of the YOLOv7 ImplicitHead: y = im * (conv1x1(x + ia) + b)
    (models/heads/implicit_head.py:26-36).  Lowered as  u = W x + (b + W ia)  [MFMA conv with an
    effective bias, fp32 output]  and  y = im * u;  backward: du = im*dy feeds the ordinary
    dgrad / wgrad / bias-grad kernels, plus a tiny launch for d(ia), d(im) and the ia term of dW."""

    def __init__(self, g, head, level, x, conv, ia, im):
        self.g, self.head, self.level, self.x = g, head, level, x
        x.needs_tensor = True
        self.conv, self.ia, self.im = conv, ia, im
        self.Cout, self.Cin = conv.weight.shape[0], conv.weight.shape[1]
        self.pc = PackedConv(g, [(conv.weight, None, 0)], 1, self.Cin)  # bias handled by implicit_bias
        N, H, W = x.N, x.H, x.W
        self.desc = conv_desc(g, N, H, W, self.Cin, self.Cout, 1, 1, self.Cin, head.nch, 1)
        self.du_ld = (self.Cout + 7) // 8 * 8
        if g.dtype == BF16:
            self.pc.set_slabs(conv_desc(g, N, H, W, self.Cin, self.Cout, 1, 1, self.Cin, self.du_ld, 0))
        g.add_op(self)
        g.post_unpack.append(self)

    def _rows(self):
        return self.x.M

    def fwd(self):
        g, hd = self.g, self.head
        r0 = hd.lvl_row[self.level]
        self.desc.x_ld = self.x.ld
        call("plyolo_implicit_bias", ptr(self.conv.weight), ptr(self.ia), ptr(self.conv.bias), self.pc.bp, self.Cout, self.Cin, None)
        u = hd.u.data_ptr() + r0 * hd.nch * 4
        call("plyolo_conv2d_fwd", C.byref(self.desc), g.aptr(self.x), self.pc.wp, self.pc.bp, u, None, None)
        call("plyolo_scale_channels", u, ptr(self.im), hd.raw.data_ptr() + r0 * hd.nch * 4, self._rows(), self.Cout, None)

    def bwd(self):
        g, hd = self.g, self.head
        r0 = hd.lvl_row[self.level]
        rows = self._rows()
        dev = g.device
        if not hasattr(self, "du"):
            self.du = torch.zeros(rows * self.du_ld, dtype=g.tdtype, device=dev)
            self.nblk = _lib.lib().plyolo_implicit_bwd_blocks(rows)
            self.partial = torch.empty(self.nblk * self.Cout, dtype=torch.float32, device=dev)
        dy = hd.draw.data_ptr() + r0 * hd.nch * 4
        u = hd.u.data_ptr() + r0 * hd.nch * 4
        call("plyolo_implicit_bwd", g.dtype, dy, u, ptr(self.im), self.du.data_ptr(), self.du_ld, self.partial.data_ptr(), rows, self.Cout, None)
        d = conv_desc(g, self.desc.N, self.desc.H, self.desc.W, self.Cin, self.Cout, 1, 1, self.x.ld, self.du_ld, 0)
        self.keep = d
        acc = g.grad_mode(self.x)
        call("plyolo_conv2d_dgrad", C.byref(d), self.du.data_ptr(), self.pc.wpd, g.gptr(self.x), acc, None)

        # bias sums, weight gradient and its slab fold only feed the optimizer: weight-gradient lane, like every other unit
        # (du is private to this level and written once per plan, so the deferred readers need no reuse event)
        def param_grads():
            call("plyolo_bias_grad", g.dtype, self.du.data_ptr(), rows, self.Cout, self.du_ld, self.pc.dbp, None)
            call("plyolo_conv2d_wgrad", C.byref(d), g.aptr(self.x), self.du.data_ptr(), self.pc.dwp, None)
            self.pc.reduce_slabs()

        if g.use_lanes:
            g.defer_param_grads(self.lane, param_grads)
        else:
            param_grads()

    def post_unpack(self):
        """After unpack_wgrads wrote dW: d(im), d(ia), d(bias) and the ia term of dW."""
        g = self.g
        call("plyolo_implicit_param_grads", self.partial.data_ptr(), self.nblk, ptr(self.conv.weight), ptr(self.ia), self.pc.dbp,
             g.grad_ptr_of(self.im), g.grad_ptr_of(self.ia), g.grad_ptr_of(self.conv.weight), g.grad_ptr_of(self.conv.bias),
             self.Cout, self.Cin, None)


class V7HeadBuffers:
    """Level-major raw YOLOv7 head output: level l = dense NHWC [B, h, w, na*(5+C)]."""

    def __init__(self, g, B, num_classes, na, sizes, strides, anchors):
        self.g, self.B, self.nc, self.na = g, B, num_classes, na
        self.ch = 5 + num_classes
        self.nch = na * self.ch
        self.sizes, self.strides, self.anchors = list(sizes), list(strides), anchors
        self.lvl_row, self.lvl_off = [], []
        r = a = 0
        for (h, w) in sizes:
            self.lvl_row.append(r)
            self.lvl_off.append(a)
            r += B * h * w
            a += na * h * w
        self.rows, self.A = r, a
        dev = g.device
        self.raw = torch.empty(self.rows * self.nch, dtype=torch.float32, device=dev)
        self.u = torch.empty(self.rows * self.nch, dtype=torch.float32, device=dev)
        self.draw = None

    def alloc_grad_only(self):
        self.draw = torch.zeros(self.rows * self.nch, dtype=torch.float32, device=self.g.device)

    def alloc_loss(self, max_labels):
        """Buffers of the training branch (yolov7_loss.py:80-153): labels in, 4 loss scalars out."""
        dev = self.g.device
        self.M = max(int(max_labels), 1)
        self.desc = yolov7_desc(self.B, self.M, self.nc, self.sizes, self.strides, self.anchors)
        self.labels = torch.zeros(self.B * self.M * 5, dtype=torch.float32, device=dev)
        self.losses = torch.zeros(4, dtype=torch.float32, device=dev)
        self.gout = torch.zeros(4, dtype=torch.float32, device=dev)
        self.ws_bytes = _lib.lib().plyolo_yolov7_workspace(C.byref(self.desc))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        self.draw = torch.empty(self.rows * self.nch, dtype=torch.float32, device=dev)

    def set_map_grads(self, grads):
        for (h, w), r0, gm in zip(self.sizes, self.lvl_row, grads):
            n = self.B * h * w
            self.draw.view(self.rows, self.nch)[r0:r0 + n].copy_(gm.permute(0, 2, 3, 1).reshape(n, self.nch))

    def alloc_eval(self):
        self.eval_out = torch.empty(self.B * self.A * self.ch, dtype=torch.float32, device=self.g.device)
        self.eval_shape = (self.B, self.A, self.ch)
        self.anchor_t = torch.tensor(self.anchors, dtype=torch.float32).reshape(-1).to(self.g.device)


class YoloV7LossOp:
    """YOLOv7Loss train branch (yolov7_loss.py:80-368) -> csrc/yolov7_loss.hip."""

    def __init__(self, g, head):
        self.g, self.head = g, head
        g.add_op(self)

    def fwd(self):
        hd = self.head
        call("plyolo_yolov7_loss_fwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.losses.data_ptr(),
             hd.ws.data_ptr(), hd.ws_bytes, None)

    def bwd(self):
        hd = self.head
        call("plyolo_yolov7_loss_bwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.gout.data_ptr(),
             hd.draw.data_ptr(), hd.ws.data_ptr(), hd.ws_bytes, None)


class YoloV7EvalDecodeOp:
    def __init__(self, g, head):
        self.g, self.head = g, head
        g.add_op(self)

    def fwd(self):
        hd = self.head
        for l, ((h, w), s) in enumerate(zip(hd.sizes, hd.strides)):
            call("plyolo_yolov7_eval_decode", hd.raw.data_ptr() + hd.lvl_row[l] * hd.nch * 4, hd.B, h, w, hd.na, hd.nc, int(s),
                 hd.anchor_t.data_ptr() + l * hd.na * 2 * 4, hd.eval_out.data_ptr(), hd.A, hd.lvl_off[l], None)

    def bwd(self):
        pass


class HeadPredOp:
    """The three bare 1x1 prediction convs of one DecoupledHead level
    (decoupled_head.py:43-62,86-93), packed as two convs (cls: C_cls outputs from the
    cls branch; reg+obj: 5 outputs from the reg branch) that write straight into the
    level's [B, h, w, 5+C] block of the level-major raw prediction tensor."""

    def __init__(self, g, head, level, cls_feat, reg_feat, cls_conv, reg_conv, obj_conv):
        self.g, self.head, self.level = g, head, level
        self.cls_feat, self.reg_feat = cls_feat, reg_feat
        cls_feat.lazy_users.append(self)
        reg_feat.lazy_users.append(self)
        self.nc = cls_conv.weight.shape[0]
        Cin = cls_conv.weight.shape[1]
        self.pc_cls = PackedConv(g, [(cls_conv.weight, cls_conv.bias, 0)], 1, Cin)
        self.pc_ro = PackedConv(g, [(reg_conv.weight, reg_conv.bias, 0), (obj_conv.weight, obj_conv.bias, 4)], 1, Cin)
        N, H, W = cls_feat.N, cls_feat.H, cls_feat.W
        self.d_cls = conv_desc(g, N, H, W, Cin, self.nc, 1, 1, Cin, head.nch, 1)
        self.d_ro = conv_desc(g, N, H, W, Cin, 5, 1, 1, Cin, head.nch, 1)
        if g.dtype == BF16:
            self.pc_ro.set_slabs(conv_desc(g, N, H, W, Cin, 5, 1, 1, Cin, 16, 0))
            self.pc_cls.set_slabs(conv_desc(g, N, H, W, Cin, self.nc, 1, 1, Cin, head.cls_ld, 0))
        if not hasattr(head, "pred_ops"):
            head.pred_ops = []
        head.pred_ops.append(self)
        g.add_op(self)

    def bias_jobs(self):
        """(dy pointer, rows, channels, pitch, packed bias gradient) of the reg+obj and of the cls prediction conv of this level."""
        g, hd = self.g, self.head
        row0, M = hd.lvl_row[self.level], self.cls_feat.M
        if g.dtype == BF16:
            return [(hd.d_regobj.data_ptr() + row0 * 16 * 2, M, 5, 16, self.pc_ro.dbp),
                    (hd.d_cls.data_ptr() + row0 * hd.cls_ld * 2, M, self.nc, hd.cls_ld, self.pc_cls.dbp)]
        base = hd.draw.data_ptr() + row0 * hd.nch * 4
        return [(base, M, 5, hd.nch, self.pc_ro.dbp), (base + 5 * 4, M, self.nc, hd.nch, self.pc_cls.dbp)]

    def dgrad_descs(self):
        """(reg+obj, cls) descriptors of the two data gradients (x_ld = pitch of the features' gradient matrices)."""
        g, hd = self.g, self.head
        if g.dtype == BF16:
            return (conv_desc(g, self.d_ro.N, self.d_ro.H, self.d_ro.W, self.d_ro.Cin, 5, 1, 1, self.reg_feat.ld, 16, 0),
                    conv_desc(g, self.d_cls.N, self.d_cls.H, self.d_cls.W, self.d_cls.Cin, self.nc, 1, 1, self.cls_feat.ld, hd.cls_ld, 0))
        d_ro, d_cl = _copy_desc(self.d_ro), _copy_desc(self.d_cls)
        d_ro.x_ld, d_cl.x_ld = self.reg_feat.ld, self.cls_feat.ld
        d_ro.x_coef = d_cl.x_coef = None
        return d_ro, d_cl

    def fwd(self):
        g, hd = self.g, self.head
        base = hd.raw.data_ptr() + hd.lvl_row[self.level] * hd.nch * 4
        self.x_cls, self.d_cls.x_ld = g.src(self.cls_feat)
        self.x_ro, self.d_ro.x_ld = g.src(self.reg_feat)
        g.set_lazy(self.d_cls, self.cls_feat)
        g.set_lazy(self.d_ro, self.reg_feat)
        call("plyolo_conv2d_fwd", C.byref(self.d_ro), self.x_ro, self.pc_ro.wp, self.pc_ro.bp, base, None, None)
        call("plyolo_conv2d_fwd", C.byref(self.d_cls), self.x_cls, self.pc_cls.wp, self.pc_cls.bp, base + 5 * 4, None, None)

    def bwd(self):
        g, hd = self.g, self.head
        row0 = hd.lvl_row[self.level]
        M = self.cls_feat.M
        d_ro, d_cl = self.dgrad_descs()
        if g.dtype == BF16:
            dro = hd.d_regobj.data_ptr() + row0 * 16 * 2
            dcl = hd.d_cls.data_ptr() + row0 * hd.cls_ld * 2
            ld_ro, ld_cl = 16, hd.cls_ld
        else:
            base = hd.draw.data_ptr() + row0 * hd.nch * 4
            dro, dcl = base, base + 5 * 4
            ld_ro = ld_cl = hd.nch
        # d_ro / d_cl: data gradients (x_ld = pitch of the features' gradient matrices); w_ro / w_cl: weight gradients
        # (x = the features as the forward read them: stored, or the producer's z + coefficients)
        w_ro, w_cl = _copy_desc(d_ro), _copy_desc(d_cl)
        w_ro.x_ld, w_cl.x_ld = self.d_ro.x_ld, self.d_cls.x_ld
        g.set_lazy(w_ro, self.reg_feat)
        g.set_lazy(w_cl, self.cls_feat)
        self.keep = (d_ro, d_cl, w_ro, w_cl)
        # the data gradients continue the main chain; the parameter gradients (bias sums, weight gradients)
        # only feed the optimizer and go to the weight-gradient lane (their inputs -- the loss gradients and the
        # forward features -- are not written again in this plan)
        lanes, me = g.use_lanes, self.lane
        acc = g.grad_mode(self.reg_feat)
        red = g.red_for(self, self.reg_feat)
        call("plyolo_conv2d_dgrad_red", C.byref(d_ro), dro, self.pc_ro.wpd, g.gptr(self.reg_feat), acc, C.byref(red) if red is not None else None, None)
        acc = g.grad_mode(self.cls_feat)
        red = g.red_for(self, self.cls_feat)
        call("plyolo_conv2d_dgrad_red", C.byref(d_cl), dcl, self.pc_cls.wpd, g.gptr(self.cls_feat), acc, C.byref(red) if red is not None else None, None)

        def param_grads():
            if not getattr(hd, "bias_fused", False):     # else: one launch for every level, behind the loss backward (YoloxLossOp.bwd)
                call("plyolo_bias_grad", g.dtype, dro, M, 5, ld_ro, self.pc_ro.dbp, None)
                call("plyolo_bias_grad", g.dtype, dcl, M, self.nc, ld_cl, self.pc_cls.dbp, None)
            call("plyolo_conv2d_wgrad", C.byref(w_ro), self.x_ro, dro, self.pc_ro.dwp, None)
            self.pc_ro.reduce_slabs()
            call("plyolo_conv2d_wgrad", C.byref(w_cl), self.x_cls, dcl, self.pc_cls.dwp, None)
            self.pc_cls.reduce_slabs()

        if lanes:
            g.defer_param_grads(me, param_grads)
        else:
            param_grads()


class HeadBuffers:
    """Level-major raw prediction tensor + the loss-side buffers shared by all levels."""

    def __init__(self, g, B, num_classes, sizes, strides, max_labels):
        self.g = g
        self.B, self.nc, self.nch = B, num_classes, 5 + num_classes
        self.sizes, self.strides = list(sizes), list(strides)
        self.lvl_off, self.lvl_row = [], []
        a = r = 0
        for (h, w) in sizes:
            self.lvl_off.append(a)
            self.lvl_row.append(r)
            a += h * w
            r += B * h * w
        self.A, self.rows = a, r
        self.M = max_labels
        self.cls_ld = (num_classes + 7) // 8 * 8
        dev = g.device
        self.raw = torch.empty(self.rows * self.nch, dtype=torch.float32, device=dev)
        self.desc = YoloxDesc()
        d = self.desc
        d.B, d.A, d.C, d.M, d.nlevels = B, self.A, num_classes, max_labels, len(sizes)
        for i, ((h, w), s) in enumerate(zip(sizes, strides)):
            d.lvl_h[i], d.lvl_w[i], d.lvl_stride[i] = h, w, int(s)
            d.lvl_off[i], d.lvl_row[i] = self.lvl_off[i], self.lvl_row[i]

    def alloc_loss(self):
        g, dev = self.g, self.g.device
        BA = self.B * self.A
        self.labels = torch.zeros(self.B * self.M * 5, dtype=torch.float32, device=dev)
        self.fg = torch.empty(BA, dtype=torch.uint8, device=dev)
        self.mgt = torch.empty(BA, dtype=torch.int32, device=dev)
        self.miou = torch.empty(BA, dtype=torch.float32, device=dev)
        self.losses = torch.zeros(8, dtype=torch.float32, device=dev)
        self.gout = torch.zeros(8, dtype=torch.float32, device=dev)
        self.ws_bytes = _lib.lib().plyolo_yolox_workspace(C.byref(self.desc))
        self.ws = torch.empty(self.ws_bytes, dtype=torch.uint8, device=dev)
        if g.dtype == BF16:
            self.d_regobj = torch.zeros(self.rows * 16, dtype=torch.bfloat16, device=dev)
            self.d_cls = torch.zeros(self.rows * self.cls_ld, dtype=torch.bfloat16, device=dev)
            self.draw = None
        else:
            self.draw = torch.empty(self.rows * self.nch, dtype=torch.float32, device=dev)
            self.d_regobj = self.d_cls = None

    def alloc_grad_only(self):
        """Head-map gradient buffers for the labels=None training path (the caller's own
        loss back-propagates into the raw maps)."""
        g, dev = self.g, self.g.device
        if g.dtype == BF16:
            self.d_regobj = torch.zeros(self.rows * 16, dtype=torch.bfloat16, device=dev)
            self.d_cls = torch.zeros(self.rows * self.cls_ld, dtype=torch.bfloat16, device=dev)
            self.draw = None
        else:
            self.draw = torch.empty(self.rows * self.nch, dtype=torch.float32, device=dev)
            self.d_regobj = self.d_cls = None

    def set_map_grads(self, grads):
        """API edge (not the hot path): NCHW fp32 gradients of the raw head maps -> the
        level-major gradient buffers the head backward kernels read."""
        B = self.B
        for (h, w), r0, gm in zip(self.sizes, self.lvl_row, grads):
            n = B * h * w
            flat = gm.permute(0, 2, 3, 1).reshape(n, self.nch)
            if self.draw is not None:
                self.draw.view(self.rows, self.nch)[r0:r0 + n].copy_(flat)
            else:
                self.d_regobj.view(self.rows, 16)[r0:r0 + n, :5].copy_(flat[:, :5])
                self.d_cls.view(self.rows, self.cls_ld)[r0:r0 + n, :self.nc].copy_(flat[:, 5:])

    def alloc_eval(self):
        self.eval_out = torch.empty(self.B * self.A * self.nch, dtype=torch.float32, device=self.g.device)
        self.eval_shape = (self.B, self.A, self.nch)


class YoloxLossOp:
    def __init__(self, g, head):
        self.g, self.head = g, head
        g.add_op(self)

    def fwd(self):
        hd = self.head
        call("plyolo_yolox_loss_fwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.fg.data_ptr(),
             hd.mgt.data_ptr(), hd.miou.data_ptr(), hd.losses.data_ptr(), hd.ws.data_ptr(), hd.ws_bytes, None)

    def bwd(self):
        hd, g = self.head, self.g
        call("plyolo_yolox_loss_bwd", C.byref(hd.desc), hd.raw.data_ptr(), hd.labels.data_ptr(), hd.fg.data_ptr(),
             hd.mgt.data_ptr(), hd.miou.data_ptr(), hd.losses.data_ptr(), hd.gout.data_ptr(), ptr(hd.draw),
             ptr(hd.d_regobj), ptr(hd.d_cls), hd.cls_ld, None)
        # the bias gradients of every level's prediction convs only need the loss gradient just written: ONE launch pair for all
        # of them (12 launches as two per conv), on the weight-gradient lane
        ops = getattr(hd, "pred_ops", [])
        jobs = [j for op in ops for j in op.bias_jobs()]
        V = g.vec
        hd.bias_fused = bool(jobs) and len(jobs) <= 8 and all(ld % V == 0 and (c + V - 1) // V * V <= ld and p % 16 == 0 for (p, m, c, ld, d) in jobs)
        if hd.bias_fused:
            from ._lib import BiasJob
            arr = (BiasJob * len(jobs))()
            for i, (p, m, c, ld, d) in enumerate(jobs):
                arr[i].dy, arr[i].M, arr[i].C, arr[i].ld, arr[i].db = p, m, c, ld, d

            def bias_grads():
                call("plyolo_bias_grad_multi", g.dtype, C.cast(arr, C.c_void_p), len(jobs), None)

            if g.use_lanes:
                g.defer_param_grads(self.lane, bias_grads)
            else:
                bias_grads()


class YoloxEvalDecodeOp:
    def __init__(self, g, head):
        self.g, self.head = g, head
        g.add_op(self)

    def fwd(self):
        hd = self.head
        call("plyolo_yolox_eval_decode", C.byref(hd.desc), hd.raw.data_ptr(), hd.eval_out.data_ptr(), None)

    def bwd(self):
        pass


class Plan:
    """Owns one plyolo_plan handle."""

    def __init__(self):
        self.h = _lib.lib().plyolo_plan_create()
        self.graph_ready = False
        self.cur = 0          # lane of the launches being recorded (mirrors plyolo_plan_lane)

    def __enter__(self):
        call("plyolo_plan_begin", self.h)
        self.cur = 0
        return self

    def __exit__(self, *exc):
        _lib.lib().plyolo_plan_end(self.h)
        return False

    def size(self):
        return _lib.lib().plyolo_plan_size(self.h)

    def lanes(self):
        return _lib.lib().plyolo_plan_lanes(self.h)

    def hooks(self):
        return _lib.lib().plyolo_plan_hooks(self.h)

    def hook(self, lane, hook_id):
        call("plyolo_plan_hook", self.h, lane, hook_id)

    def set_hook(self, fn):
        """fn(id, stream_ptr) -> None, called on the host thread by eager replays at every recorded hook."""
        def tramp(hook_id, stream, user):
            try:
                fn(int(hook_id), int(stream or 0))
                return 0
            except BaseException as e:      # never let an exception cross the C frame
                self.hook_error = e
                return 1
        self.hook_error = None
        self._hook_c = C.CFUNCTYPE(C.c_int, C.c_int, C.c_void_p, C.c_void_p)(tramp)   # keep the thunk alive
        call("plyolo_plan_set_hook", self.h, C.cast(self._hook_c, C.c_void_p), None)

    # lanes (concurrent launch sequences inside a hipGraph replay), see include/plyolo.h
    def lane(self, l):
        call("plyolo_plan_lane", self.h, l)
        self.cur = l

    def record(self, lane):
        ev = _lib.lib().plyolo_plan_record(self.h, lane)
        if ev < 0:
            _lib.check(ev, "plyolo_plan_record")
        return ev

    def wait(self, lane, ev):
        call("plyolo_plan_wait", self.h, lane, ev)

    def run(self, stream, use_graph=False):
        if use_graph:
            if not self.graph_ready:
                call("plyolo_plan_graph_instantiate", self.h, stream)
                self.graph_ready = True
            call("plyolo_plan_graph_launch", self.h, stream)
        else:
            call("plyolo_plan_run", self.h, stream)

    def profile(self, stream):
        """Replay with a hipEvent pair around every launch.  Returns a list of
        (label, ms, algorithmic_flops, algorithmic_bytes), one entry per launch."""
        n = self.size()
        ms = (C.c_float * max(n, 1))()
        call("plyolo_plan_profile", self.h, stream, C.cast(ms, C.c_void_p), n)
        out = []
        buf = C.create_string_buffer(96)
        fl, by = C.c_double(), C.c_double()
        for i in range(n):
            call("plyolo_plan_op_info", self.h, i, buf, 96, C.byref(fl), C.byref(by))
            if buf.value not in (b"record", b"wait"):   # ordering markers, not launches
                out.append((buf.value.decode(), float(ms[i]), fl.value, by.value, _lib.lib().plyolo_plan_op_lane(self.h, i)))
        return out

    def lane_times(self, stream):
        """One eager multi-lane replay; returns [ms to the end of lane 0, lane 1, ..., ms to the join]."""
        n = self.lanes() + 1
        ms = (C.c_float * n)()
        call("plyolo_plan_lane_times", self.h, stream, C.cast(ms, C.c_void_p), n)
        return [float(v) for v in ms]

    def __del__(self):
        try:
            if self.h:
                _lib.lib().plyolo_plan_destroy(self.h)
                self.h = None
        except Exception:
            pass
