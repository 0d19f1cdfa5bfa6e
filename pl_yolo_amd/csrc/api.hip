// C-ABI glue: probes, error text, launch plans (record / replay / hipGraph) and the
// dtype dispatch of the convolution entry points.
#include <dlfcn.h>
#include <stdarg.h>

#include <mutex>
#include <unordered_map>

#include "common.h"

namespace plyolo {

static thread_local std::string g_err;
static thread_local Plan* g_rec = nullptr;

void set_error(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
}
Plan* recording_plan() { return g_rec; }

hipError_t ensure_dynamic_lds(const void* kernel, size_t bytes) {
  if (bytes <= 64 * 1024) return hipSuccess;
  static std::mutex mu;
  static std::unordered_map<const void*, size_t> granted;
  std::lock_guard<std::mutex> lock(mu);
  size_t& have = granted[kernel];
  if (bytes <= have) return hipSuccess;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  if (e == hipSuccess) have = bytes;
  return e;
}

static thread_local std::string g_ann_label;
static thread_local double g_ann_flops = 0.0, g_ann_bytes = 0.0;
void annotate(const char* label, double flops, double bytes) {
  g_ann_label = label ? label : "";
  g_ann_flops = flops;
  g_ann_bytes = bytes;
}
void take_annotation(PlanOp* op) {
  if (op) {
    op->label = g_ann_label.empty() ? "other" : g_ann_label;
    op->flops = g_ann_flops;
    op->bytes = g_ann_bytes;
  }
  g_ann_label.clear();
  g_ann_flops = g_ann_bytes = 0.0;
}

int conv_mfma_fwd(const plyolo_conv_desc*, const void*, const void*, const float*, void*, double*, const float*, int, const void*, int, void*);
void conv_mfma_pack_elems(int Cout_total, int Cin_p, int ksize, size_t* wp, size_t* wpd);
int conv_mfma_dgrad(const plyolo_conv_desc*, const void*, const void*, void*, int, const plyolo_bn_red*, void*, int*);
int conv_mfma_dgrad_red_fits(const plyolo_conv_desc*);
int conv_mfma_dgrad_bn_fits(const plyolo_conv_desc*, int);
int conv_mfma_dgrad_bn(const plyolo_conv_desc*, const plyolo_bn_bwd_fuse*, const void*, void*, int, const plyolo_bn_red*, void*);
int conv_mfma_wgrad(const plyolo_conv_desc*, const void*, const void*, float*, void*);
int conv_mfma_wgrad_slabs(const plyolo_conv_desc*);
int conv_mfma_wgrad_bn_fits(const plyolo_conv_desc*, int);
int conv_mfma_wgrad_bn(const plyolo_conv_desc*, const plyolo_bn_bwd_fuse*, const void*, float*, void*);
bool conv_pw_enabled();
int conv_pw_fwd(const plyolo_conv_desc*, const void*, const void*, const float*, void*, double*, const float*, int, const void*, int, void*);
int conv_pw_dgrad(const plyolo_conv_desc*, const void*, const void*, void*, int, const plyolo_bn_red*, void*, int*);
int conv_pw_dgrad_bn_fits(const plyolo_conv_desc*, int);
int conv_pw_dgrad_bn(const plyolo_conv_desc*, const plyolo_bn_bwd_fuse*, const void*, void*, int, const plyolo_bn_red*, void*);
int conv_pw_bwd_fits(const plyolo_conv_desc*, int);
int conv_pw_bwd_slabs(const plyolo_conv_desc*);
int conv_pw_bwd(const plyolo_conv_desc*, const plyolo_bn_bwd_fuse*, const void*, const void*, void*, int, float*, const plyolo_bn_red*, void*);
int conv_ref_fwd(const plyolo_conv_desc*, const void*, const void*, const float*, void*, double*, void*);
int conv_ref_dgrad(const plyolo_conv_desc*, const void*, const void*, void*, int, void*);
int conv_ref_wgrad(const plyolo_conv_desc*, const void*, const void*, float*, void*);

// 1x1 stride-1 layers run on the dedicated pointwise kernel (conv_pw.hip); PLYOLO_PW=0 keeps them on the 3x3 kernel's one-tap path
static bool lazy_3x3_ok(const plyolo_conv_desc* d) { return d->x_coef_ld % 4 == 0 && ((size_t)d->x_coef & 15) == 0; }   // 16-byte coefficient loads
static bool is_pointwise(const plyolo_conv_desc* d) { return d->ksize == 1 && d->stride == 1 && conv_pw_enabled(); }

static int check_conv(const plyolo_conv_desc* d, const char* who, bool fwd) {
  PLY_CHECK_ARG(d != nullptr, "%s: null descriptor", who);
  PLY_CHECK_ARG(d->dtype == PLYOLO_BF16 || d->dtype == PLYOLO_F32, "%s: bad dtype %d", who, d->dtype);
  PLY_CHECK_ARG(d->ksize == 1 || d->ksize == 3, "%s: ksize must be 1 or 3 (got %d)", who, d->ksize);
  PLY_CHECK_ARG(d->stride == 1 || d->stride == 2, "%s: stride must be 1 or 2 (got %d)", who, d->stride);
  PLY_CHECK_ARG(d->N > 0 && d->H > 0 && d->W > 0 && d->Cin > 0 && d->Cout > 0, "%s: non-positive dimension", who);
  PLY_CHECK_ARG(d->x_ld >= d->Cin, "%s: x_ld %d < Cin %d", who, d->x_ld, d->Cin);
  if (d->dtype == PLYOLO_BF16) {
    PLY_CHECK_ARG(d->Cin % 8 == 0 && d->x_ld % 8 == 0, "%s: bf16 path needs Cin and x_ld multiples of 8 (Cin %d, x_ld %d)", who, d->Cin, d->x_ld);
    if (fwd && !d->y_f32) PLY_CHECK_ARG(d->Cout % 8 == 0 && d->y_ld % 8 == 0, "%s: bf16 output needs Cout and y_ld multiples of 8 (Cout %d, y_ld %d)", who, d->Cout, d->y_ld);
    if (!fwd) PLY_CHECK_ARG(d->y_ld % 8 == 0 && d->y_ld >= ((d->Cout + 7) & ~7), "%s: bf16 dy rows must hold Cout rounded up to 8 channels (Cout %d, y_ld %d)", who, d->Cout, d->y_ld);
    // the MFMA kernels keep 32-bit offsets: elements within one image, bytes within one weight pack
    const double img = (double)d->H * d->W * (d->x_ld > d->y_ld ? d->x_ld : d->y_ld);
    const double pack = 2.0 * d->ksize * d->ksize * (double)((d->Cout + 31) / 32 * 32) * ((d->Cin + 31) / 32 * 32);
    PLY_CHECK_ARG(img < 2147483000.0 && pack < 2147483000.0, "%s: image plane or weight pack beyond the 32-bit offsets of the bf16 kernels", who);
    PLY_CHECK_ARG((double)d->N * d->H * d->W < 2147483000.0, "%s: more than 2^31 pixels", who);
  }
  if (d->x_coef != nullptr) {
#ifndef PLYOLO_OPTIN
    PLY_CHECK_ARG(d->dtype != PLYOLO_BF16, "%s: this libplyolo_hip.so was built without the lazy-input instances of the bf16 kernels (plyolo_conv_desc.x_coef: measured slower, opt-in) -- rebuild with `make -C pl_yolo_amd/csrc OPTIN=1`", who);
#endif
    PLY_CHECK_ARG(d->x_coef_ld >= d->Cin && d->x_act >= 0 && d->x_act <= PLYOLO_ACT_LRELU, "%s: lazy input needs x_coef_ld >= Cin and a valid x_act", who);
    PLY_CHECK_ARG(d->dtype != PLYOLO_BF16 || is_pointwise(d) || lazy_3x3_ok(d), "%s: lazy input is not available for this bf16 convolution shape", who);
  }
  return 0;
}

__global__ void fill_bytes_kernel(unsigned char* p, unsigned value, size_t bytes) {
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= bytes) return;
  if (i + 16 <= bytes && (((size_t)p + i) & 15) == 0) {
    const unsigned w = value * 0x01010101u;
    *(u32x4*)(p + i) = u32x4{w, w, w, w};
  } else {
    for (size_t k = i; k < bytes && k < i + 16; ++k) p[k] = (unsigned char)value;
  }
}
hipError_t fill_async(void* p, int value, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
  const size_t nthr = (bytes + 15) / 16;
  hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, (unsigned char*)p, (unsigned)(value & 0xff), bytes);
  return hipGetLastError();
}

}  // namespace plyolo

using namespace plyolo;

extern "C" {

int plyolo_version(void) { return PLYOLO_ABI_VERSION; }
int plyolo_build_flags(void) {
  int f = 0;
#ifdef PLYOLO_OPTIN
  f |= PLYOLO_BUILD_OPTIN;
#endif
#ifdef PLYOLO_DIAG_ABLATE
  f |= PLYOLO_BUILD_DIAG;
#endif
  return f;
}
const char* plyolo_arch(void) { return "gfx950"; }
const char* plyolo_last_error(void) { return g_err.c_str(); }

plyolo_plan* plyolo_plan_create(void) { return (plyolo_plan*)new Plan(); }
void plyolo_plan_destroy(plyolo_plan* p) {
  Plan* q = (Plan*)p;
  if (!q) return;
  if (q->exec) (void)hipGraphExecDestroy(q->exec);
  if (q->graph) (void)hipGraphDestroy(q->graph);
  for (auto e : q->events) (void)hipEventDestroy(e);
  for (auto st : q->side) (void)hipStreamDestroy(st);
  for (auto e : q->io_events) (void)hipEventDestroy(e);
  if (g_rec == q) g_rec = nullptr;
  delete q;
}
int plyolo_plan_begin(plyolo_plan* p) {
  PLY_CHECK_ARG(p != nullptr, "plan_begin: null plan");
  PLY_CHECK_ARG(g_rec == nullptr, "plan_begin: another plan is already recording on this thread");
  g_rec = (Plan*)p;
  return 0;
}
int plyolo_plan_end(plyolo_plan* p) {
  PLY_CHECK_ARG(g_rec == (Plan*)p, "plan_end: this plan is not recording");
  g_rec = nullptr;
  return 0;
}
// ---- lanes: independent launch sequences that a hipGraph replay may run concurrently ----------
int plyolo_plan_lane(plyolo_plan* p, int lane) {
  PLY_CHECK_ARG(g_rec == (Plan*)p && p != nullptr, "plan_lane: this plan is not recording");
  PLY_CHECK_ARG(lane >= 0 && lane < 16, "plan_lane: lane must be 0..15");
  Plan* q = (Plan*)p;
  q->cur_lane = lane;
  if (lane + 1 > q->nlanes) q->nlanes = lane + 1;
  return 0;
}
int plyolo_plan_record(plyolo_plan* p, int lane) {
  PLY_CHECK_ARG(g_rec == (Plan*)p && p != nullptr, "plan_record: this plan is not recording");
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(lane >= 0 && lane < q->nlanes, "plan_record: unknown lane %d", lane);
  PlanOp op;
  op.kind = 1; op.lane = lane; op.ev = q->nevents++;
  op.label = "record";
  q->ops.emplace_back(std::move(op));
  return q->nevents - 1;
}
int plyolo_plan_wait(plyolo_plan* p, int lane, int ev) {
  PLY_CHECK_ARG(g_rec == (Plan*)p && p != nullptr, "plan_wait: this plan is not recording");
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(lane >= 0 && lane < 16 && ev >= 0 && ev < q->nevents, "plan_wait: bad lane/event");
  if (lane + 1 > q->nlanes) q->nlanes = lane + 1;
  PlanOp op;
  op.kind = 2; op.lane = lane; op.ev = ev;
  op.label = "wait";
  q->ops.emplace_back(std::move(op));
  return 0;
}

// Host hook: at replay, when the issue loop reaches this point of `lane`, the registered callback runs on the host with
// that lane's stream -- work it enqueues there (an RCCL collective of the framework the host uses) is ordered after
// everything recorded on the lane before the hook, and the plan's join waits for it.
int plyolo_plan_hook(plyolo_plan* p, int lane, int id) {
  PLY_CHECK_ARG(g_rec == (Plan*)p && p != nullptr, "plan_hook: this plan is not recording");
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(lane >= 0 && lane < 16, "plan_hook: bad lane %d", lane);
  if (lane + 1 > q->nlanes) q->nlanes = lane + 1;
  PlanOp op;
  op.kind = 3; op.lane = lane; op.ev = id;
  op.label = "hook";
  q->ops.emplace_back(std::move(op));
  q->nhooks++;
  return 0;
}
int plyolo_plan_set_hook(plyolo_plan* p, int (*fn)(int, void*, void*), void* user) {
  PLY_CHECK_ARG(p != nullptr, "plan_set_hook: null plan");
  ((Plan*)p)->hook = fn;
  ((Plan*)p)->hook_user = user;
  return 0;
}
int plyolo_plan_hooks(const plyolo_plan* p) { return p ? ((const Plan*)p)->nhooks : 0; }

int plyolo_plan_lanes(const plyolo_plan* p) { return p ? ((const Plan*)p)->nlanes : 0; }
int plyolo_plan_size(const plyolo_plan* p) { return p ? (int)((const Plan*)p)->ops.size() : 0; }
// Issue every recorded launch: lane l on its own stream (forked from / joined into `s`), events as
// recorded.  Used both under stream capture (hipGraph) and for eager multi-stream replay.
// diagnostics (plyolo_plan_stamp_times): a timing event behind every `every`-th launch of `lane`
struct StampCfg { int lane = -1, every = 0; std::vector<hipEvent_t>* evs = nullptr; std::vector<int>* at = nullptr; };
static hipError_t issue_lanes(Plan* q, hipStream_t s, bool lanes, size_t* failed, hipEvent_t* tev = nullptr, StampCfg* stamp = nullptr) {
  if (!lanes || q->nlanes <= 1) {  // single stream, recorded order (a valid serialisation of the lanes)
    hipError_t le = hipSuccess;
    for (size_t i = 0; i < q->ops.size() && le == hipSuccess; ++i) {
      if (q->ops[i].kind == 3 && lanes && q->hook) {   // hooks run in eager replays only (never under stream capture)
        if (q->hook(q->ops[i].ev, (void*)s, q->hook_user) != 0) le = hipErrorUnknown;
      }
      if (q->ops[i].kind != 0) continue;
      le = q->ops[i].fn(s);
      if (le != hipSuccess && failed) *failed = i;
    }
    return le;
  }
  // Lane 0 runs on a stream of the plan's own, created in one go with the side streams (PLYOLO_OWN_MAIN=0: on the caller's stream).
  // The runtime hands hardware queues to streams round-robin in creation order, so streams created back to back sit
  // on distinct queues whatever the application created before -- the caller's stream only forks into and joins from the plan.
  static const int own_main = getenv("PLYOLO_OWN_MAIN") ? atoi(getenv("PLYOLO_OWN_MAIN")) : 1;
  const int want = q->nlanes - 1 + (own_main ? 1 : 0);
  while ((int)q->side.size() < want) {
    // (stream priorities were tried for the weight-gradient lane -- hipStreamCreateWithPriority, low or high: ANY non-default
    // priority on one lane doubled the step, 10.5 -> 22 ms on ROCm 7.2; every lane stays at the default priority)
    hipStream_t st;
    hipError_t e = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    if (e != hipSuccess) return e;
    q->side.push_back(st);
  }
  hipStream_t caller = s;
  hipEvent_t ev_io[2] = {nullptr, nullptr};
  if (own_main) {
    while (q->io_events.size() < 2) {
      hipEvent_t ev;
      hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
      if (e != hipSuccess) return e;
      q->io_events.push_back(ev);
    }
    ev_io[0] = q->io_events[0]; ev_io[1] = q->io_events[1];
    s = q->side[(size_t)q->nlanes - 1];           // the plan's main stream (last of the set)
    hipError_t e = hipEventRecord(ev_io[0], caller);
    if (e == hipSuccess) e = hipStreamWaitEvent(s, ev_io[0], 0);
    if (e != hipSuccess) return e;
  }
  const size_t need_ev = (size_t)q->nevents + 2 * (size_t)q->nlanes;
  while (q->events.size() < need_ev) {
    hipEvent_t ev;
    hipError_t e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
    q->events.push_back(ev);
  }
  auto lane_stream = [&](int l) { return l == 0 ? s : q->side[(size_t)l - 1]; };
  hipEvent_t* fork_ev = q->events.data() + q->nevents;  // [nlanes] fork, [nlanes] join
  hipError_t le = hipSuccess;
  if (tev) le = hipEventRecord(tev[0], s);
  if (q->nlanes > 1 && le == hipSuccess) {
    le = hipEventRecord(fork_ev[0], s);
    for (int l = 1; l < q->nlanes && le == hipSuccess; ++l) le = hipStreamWaitEvent(lane_stream(l), fork_ev[0], 0);
  }
  int stamp_n = 0;
  for (size_t i = 0; i < q->ops.size() && le == hipSuccess; ++i) {
    const PlanOp& op = q->ops[i];
    if (op.kind == 0) {
      le = op.fn(lane_stream(op.lane));
      if (stamp && op.lane == stamp->lane && le == hipSuccess && ++stamp_n % stamp->every == 0) {
        hipEvent_t e;
        le = hipEventCreate(&e);
        if (le == hipSuccess) le = hipEventRecord(e, lane_stream(op.lane));
        if (le == hipSuccess) { stamp->evs->push_back(e); stamp->at->push_back((int)i); }
      }
    }
    else if (op.kind == 1) le = hipEventRecord(q->events[(size_t)op.ev], lane_stream(op.lane));
    else if (op.kind == 2) le = hipStreamWaitEvent(lane_stream(op.lane), q->events[(size_t)op.ev], 0);
    else if (q->hook && q->hook(op.ev, (void*)lane_stream(op.lane), q->hook_user) != 0) le = hipErrorUnknown;
    if (le != hipSuccess && failed) *failed = i;
  }
  if (tev)
    for (int l = 0; l < q->nlanes && le == hipSuccess; ++l) le = hipEventRecord(tev[1 + l], lane_stream(l));
  // The joins run on the error path too: work already enqueued on the plan's private streams must stay ordered before whatever
  // the caller does next on ITS stream (release or reuse the buffers while Python raises) -- the first error is what is returned.
  hipError_t je = hipSuccess;
  for (int l = 1; l < q->nlanes; ++l) {
    hipError_t e = hipEventRecord(fork_ev[q->nlanes + l], lane_stream(l));
    if (e == hipSuccess) e = hipStreamWaitEvent(s, fork_ev[q->nlanes + l], 0);
    if (e != hipSuccess && je == hipSuccess) je = e;
  }
  if (own_main) {
    hipError_t e = hipEventRecord(ev_io[1], s);
    if (e == hipSuccess) e = hipStreamWaitEvent(caller, ev_io[1], 0);
    if (e != hipSuccess && je == hipSuccess) je = e;
  }
  if (le != hipSuccess && je != hipSuccess) {   // the joins failed as well: drain the plan's streams before handing the error back
    for (auto st : q->side) (void)hipStreamSynchronize(st);
  }
  return le != hipSuccess ? le : je;
}

int plyolo_plan_run(plyolo_plan* p, void* stream) {
  PLY_CHECK_ARG(p != nullptr, "plan_run: null plan");
  PLY_CHECK_ARG(g_rec == nullptr, "plan_run: cannot replay while recording");
  Plan* q = (Plan*)p;
  size_t failed = 0;
  const hipError_t e = issue_lanes(q, (hipStream_t)stream, true, &failed);
  if (e != hipSuccess) {
    set_error("plan_run: launch %zu (%s) failed: %s", failed, failed < q->ops.size() ? q->ops[failed].label.c_str() : "?", hipGetErrorString(e));
    return -2;
  }
  return 0;
}
int plyolo_plan_graph_instantiate(plyolo_plan* p, void* stream) {
  PLY_CHECK_ARG(p != nullptr, "plan_graph_instantiate: null plan");
  Plan* q = (Plan*)p;
  // a captured replay cannot run host hooks: a data-parallel plan graphed through the C ABI would silently lose its gradient exchange
  PLY_CHECK_ARG(q->nhooks == 0, "plan_graph_instantiate: the plan has %d host hook(s) (plyolo_plan_hook), which a hipGraph replay cannot run; replay it with plyolo_plan_run", q->nhooks);
  hipStream_t s = (hipStream_t)stream;
  if (q->exec) { (void)hipGraphExecDestroy(q->exec); q->exec = nullptr; }
  if (q->graph) { (void)hipGraphDestroy(q->graph); q->graph = nullptr; }
  hipError_t e = hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  if (e != hipSuccess) { set_error("hipStreamBeginCapture: %s", hipGetErrorString(e)); return -2; }
  // captured single-lane: re-launching a multi-branch graph with event edges gave wrong results on
  // ROCm 7.2 (second launch), and the eager multi-stream replay is the faster path anyway
  const hipError_t le = issue_lanes(q, s, false, nullptr);
  e = hipStreamEndCapture(s, &q->graph);
  if (le != hipSuccess || e != hipSuccess) {
    set_error("graph capture failed: %s / %s", hipGetErrorString(le), hipGetErrorString(e));
    return -2;
  }
  e = hipGraphInstantiate(&q->exec, q->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) { set_error("hipGraphInstantiate: %s", hipGetErrorString(e)); return -2; }
  return 0;
}
// Replay with a hipEvent pair around every recorded launch; ms_out[i] = device time of launch i.
int plyolo_plan_profile(plyolo_plan* p, void* stream, float* ms_out, int n) {
  PLY_CHECK_ARG(p != nullptr && ms_out != nullptr, "plan_profile: null argument");
  PLY_CHECK_ARG(g_rec == nullptr, "plan_profile: cannot replay while recording");
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(n >= (int)q->ops.size(), "plan_profile: ms_out too small (%d < %zu)", n, q->ops.size());
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev(q->ops.size() + 1);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) { set_error("plan_profile: hipEventCreate failed"); return -2; }
  hipError_t err = hipEventRecord(ev[0], s);
  for (size_t i = 0; i < q->ops.size() && err == hipSuccess; ++i) {
    if (q->ops[i].kind == 0) err = q->ops[i].fn(s);
    if (err == hipSuccess) err = hipEventRecord(ev[i + 1], s);
  }
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  if (err == hipSuccess)
    for (size_t i = 0; i < q->ops.size(); ++i) (void)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (err != hipSuccess) { set_error("plan_profile: %s", hipGetErrorString(err)); return -2; }
  return 0;
}
// Diagnostic multi-lane replay: ms_out[l] = time from the start of the replay to the end of lane l's last launch,
// ms_out[nlanes] = to the join on the caller's stream.  Synchronises the stream.
int plyolo_plan_lane_times(plyolo_plan* p, void* stream, float* ms_out, int n) {
  PLY_CHECK_ARG(p != nullptr && ms_out != nullptr, "plan_lane_times: null argument");
  PLY_CHECK_ARG(g_rec == nullptr, "plan_lane_times: cannot replay while recording");
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(q->nlanes > 1 && n >= q->nlanes + 1, "plan_lane_times: needs a multi-lane plan and %d outputs", q->nlanes + 1);
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> ev((size_t)q->nlanes + 2);
  for (auto& e : ev)
    if (hipEventCreate(&e) != hipSuccess) { set_error("plan_lane_times: hipEventCreate failed"); return -2; }
  size_t failed = 0;
  hipError_t err = issue_lanes(q, s, true, &failed, ev.data());
  if (err == hipSuccess) err = hipEventRecord(ev[(size_t)q->nlanes + 1], s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  if (err == hipSuccess)
    for (int l = 0; l <= q->nlanes; ++l) (void)hipEventElapsedTime(&ms_out[l], ev[0], ev[(size_t)l + 1]);
  for (auto& e : ev) (void)hipEventDestroy(e);
  if (err != hipSuccess) { set_error("plan_lane_times: %s", hipGetErrorString(err)); return -2; }
  return 0;
}
// Diagnostic multi-lane replay with a timing event behind every `every`-th launch of `lane`: ms_out[k] = time from the start of
// the replay to stamp k, op_out[k] = index of the recorded op the stamp follows.  Returns the number of stamps (<= cap) or a
// negative code.  Synchronises the stream.
int plyolo_plan_stamp_times(plyolo_plan* p, void* stream, int lane, int every, float* ms_out, int* op_out, int cap) {
  PLY_CHECK_ARG(p != nullptr && ms_out != nullptr && op_out != nullptr && every > 0 && cap > 0, "plan_stamp_times: bad arguments");
  PLY_CHECK_ARG(g_rec == nullptr, "plan_stamp_times: cannot replay while recording");
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(q->nlanes > 1 && lane >= 0 && lane < q->nlanes, "plan_stamp_times: needs a multi-lane plan and one of its lanes");
  hipStream_t s = (hipStream_t)stream;
  std::vector<hipEvent_t> tev((size_t)q->nlanes + 2), evs;
  std::vector<int> at;
  for (auto& e : tev)
    if (hipEventCreate(&e) != hipSuccess) { set_error("plan_stamp_times: hipEventCreate failed"); return -2; }
  StampCfg cfg;
  cfg.lane = lane; cfg.every = every; cfg.evs = &evs; cfg.at = &at;
  size_t failed = 0;
  hipError_t err = issue_lanes(q, s, true, &failed, tev.data(), &cfg);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  int n = 0;
  if (err == hipSuccess)
    for (size_t k = 0; k < evs.size() && n < cap; ++k, ++n) {
      (void)hipEventElapsedTime(&ms_out[n], tev[0], evs[k]);
      op_out[n] = at[k];
    }
  for (auto& e : tev) (void)hipEventDestroy(e);
  for (auto& e : evs) (void)hipEventDestroy(e);
  if (err != hipSuccess) { set_error("plan_stamp_times: %s", hipGetErrorString(err)); return -2; }
  return n;
}
int plyolo_plan_op_info(const plyolo_plan* p, int i, char* label, int label_cap, double* flops, double* bytes) {
  const Plan* q = (const Plan*)p;
  PLY_CHECK_ARG(q && i >= 0 && i < (int)q->ops.size(), "plan_op_info: bad index");
  if (label && label_cap > 0) snprintf(label, (size_t)label_cap, "%s", q->ops[i].label.c_str());
  if (flops) *flops = q->ops[i].flops;
  if (bytes) *bytes = q->ops[i].bytes;
  return 0;
}

int plyolo_plan_op_lane(const plyolo_plan* p, int i) {
  const Plan* q = (const Plan*)p;
  PLY_CHECK_ARG(q && i >= 0 && i < (int)q->ops.size(), "plan_op_lane: bad index");
  return q->ops[i].lane;
}

// ---- RCCL: the gradient exchange of one bucket through the C ABI.  The communicator belongs to the host
// (ncclCommInitRank over the xGMI ring); the library resolves ncclAllReduce at run time from the RCCL the process already
// uses (or from the path given to plyolo_rccl_set_library), so libplyolo_hip.so carries no link-time RCCL dependency.
static void* g_rccl_handle = nullptr;
typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
static nccl_allreduce_fn g_nccl_allreduce = nullptr;
int plyolo_rccl_set_library(const char* path) {
  void* h = dlopen(path && path[0] ? path : "librccl.so", RTLD_NOW | RTLD_GLOBAL);
  PLY_CHECK_ARG(h != nullptr, "rccl_set_library: cannot load %s: %s", path ? path : "librccl.so", dlerror());
  void* f = dlsym(h, "ncclAllReduce");
  PLY_CHECK_ARG(f != nullptr, "rccl_set_library: %s has no ncclAllReduce", path ? path : "librccl.so");
  g_rccl_handle = h;
  g_nccl_allreduce = (nccl_allreduce_fn)f;
  return 0;
}
// In-place all-reduce of `count` fp32 gradients over `comm` (an ncclComm_t) on `stream`; average != 0: ncclAvg (the mean the
// data-parallel step needs), else ncclSum.  Enqueues only; ordering is the stream's.
int plyolo_rccl_allreduce_bucket(void* comm, float* grads, size_t count, int average, void* stream) {
  PLY_CHECK_ARG(comm != nullptr && grads != nullptr && count > 0, "rccl_allreduce_bucket: null communicator / buffer or empty bucket");
  if (!g_nccl_allreduce) {
    void* f = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    if (f) g_nccl_allreduce = (nccl_allreduce_fn)f;
    else if (plyolo_rccl_set_library(nullptr) != 0) return -1;
  }
  const int ncclFloat32 = 7, ncclSum = 0, ncclAvg = 4;   // rccl.h: ncclDataType_t / ncclRedOp_t
  const int rc = g_nccl_allreduce(grads, grads, count, ncclFloat32, average ? ncclAvg : ncclSum, comm, (hipStream_t)stream);
  if (rc != 0) { set_error("ncclAllReduce failed (ncclResult_t %d)", rc); return -2; }
  return 0;
}

int plyolo_plan_graph_launch(plyolo_plan* p, void* stream) {
  Plan* q = (Plan*)p;
  PLY_CHECK_ARG(q && q->exec, "plan_graph_launch: plan has no instantiated graph");
  hipError_t e = hipGraphLaunch(q->exec, (hipStream_t)stream);
  if (e != hipSuccess) { set_error("hipGraphLaunch: %s", hipGetErrorString(e)); return -2; }
  return 0;
}

int plyolo_conv2d_fwd(const plyolo_conv_desc* d, const void* x, const void* wp, const float* bias, void* y, double* stats, void* stream) {
  if (check_conv(d, "conv2d_fwd", true)) return -1;
  if (d->dtype != PLYOLO_BF16) return conv_ref_fwd(d, x, wp, bias, y, stats, stream);
  if (is_pointwise(d)) return conv_pw_fwd(d, x, wp, bias, y, stats, nullptr, 0, nullptr, 0, stream);
  return conv_mfma_fwd(d, x, wp, bias, y, stats, nullptr, 0, nullptr, 0, stream);
}
int plyolo_conv2d_fwd_bn_act(const plyolo_conv_desc* d, const void* x, const void* wp, const float* coef, int act,
                             const void* res, int r_ld, void* y, void* stream) {
  if (check_conv(d, "conv2d_fwd_bn_act", true)) return -1;
  PLY_CHECK_ARG(d->dtype == PLYOLO_BF16 && !d->y_f32 && coef != nullptr, "conv2d_fwd_bn_act: bf16 activations and a coefficient vector required");
  PLY_CHECK_ARG(act >= PLYOLO_ACT_NONE && act <= PLYOLO_ACT_LRELU, "conv2d_fwd_bn_act: the fused epilogue covers none / silu / relu / lrelu (got %d); use plyolo_bn_act_fwd", act);
  PLY_CHECK_ARG(!res || r_ld % 8 == 0, "conv2d_fwd_bn_act: residual pitch must be a multiple of 8");
  if (is_pointwise(d)) return conv_pw_fwd(d, x, wp, nullptr, y, nullptr, coef, act, res, r_ld, stream);
  return conv_mfma_fwd(d, x, wp, nullptr, y, nullptr, coef, act, res, r_ld, stream);
}

int plyolo_pack_elems(int dtype, int Cout_total, int Cin_p, int ksize, size_t* wp_elems, size_t* wpd_elems) {
  PLY_CHECK_ARG(wp_elems && wpd_elems && Cout_total > 0 && Cin_p > 0 && (ksize == 1 || ksize == 3), "pack_elems: bad arguments");
  if (dtype == PLYOLO_BF16) {
    conv_mfma_pack_elems(Cout_total, Cin_p, ksize, wp_elems, wpd_elems);
  } else {
    const size_t taps = (size_t)ksize * ksize;
    *wp_elems = taps * Cout_total * Cin_p;
    *wpd_elems = taps * Cin_p * ((Cout_total + 7) / 8 * 8);
  }
  return 0;
}

int plyolo_conv2d_dgrad(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate, void* stream) {
  if (check_conv(d, "conv2d_dgrad", false)) return -1;
  if (d->dtype != PLYOLO_BF16) return conv_ref_dgrad(d, dy, wpd, dx, accumulate, stream);
  if (is_pointwise(d)) return conv_pw_dgrad(d, dy, wpd, dx, accumulate, nullptr, stream, nullptr);
  return conv_mfma_dgrad(d, dy, wpd, dx, accumulate, nullptr, stream, nullptr);
}
static int check_red(const plyolo_bn_red* r, const plyolo_conv_desc* d, const char* who) {
  PLY_CHECK_ARG(r->n >= 0 && r->n <= PLYOLO_BN_RED_SEGS, "%s: bad segment count", who);
  for (int k = 0; k < r->n; ++k) {
    const plyolo_bn_red_seg& g = r->seg[k];
    PLY_CHECK_ARG(g.c0 >= 0 && g.c1 > g.c0 && g.c1 <= d->Cin && g.c0 % 8 == 0 && g.c1 % 8 == 0, "%s: segment %d: channels [%d, %d) must be multiples of 8 inside [0, Cin)", who, k, g.c0, g.c1);
    PLY_CHECK_ARG(g.z && g.coef && g.bslots && g.z_ld % 8 == 0 && g.z_ld >= g.c1 - g.c0 && g.coef_ld >= g.c1 - g.c0 && g.slot_ld >= g.c1 - g.c0, "%s: segment %d: incomplete", who, k);
    PLY_CHECK_ARG(g.act >= PLYOLO_ACT_NONE && g.act <= PLYOLO_ACT_LRELU, "%s: segment %d: the folded reduction carries none / silu / relu / lrelu (hswish and gelu units keep plyolo_bn_act_bwd_reduce)", who, k);
    for (int j = 0; j < k; ++j) PLY_CHECK_ARG(g.c0 >= r->seg[j].c1 || g.c1 <= r->seg[j].c0, "%s: segments %d and %d overlap", who, j, k);
  }
  return 0;
}
/* 1 when plyolo_conv2d_dgrad_red has an instance for this data gradient (bf16; the tile configuration the launch itself picks) */
int plyolo_conv2d_dgrad_red_fits(const plyolo_conv_desc* d) {
  if (check_conv(d, "conv2d_dgrad_red_fits", false)) return -1;
  if (d->dtype != PLYOLO_BF16) return 0;
  int fits = 0;
  if (is_pointwise(d)) conv_pw_dgrad(d, nullptr, nullptr, nullptr, 0, nullptr, nullptr, &fits);
  else fits = conv_mfma_dgrad_red_fits(d);
  return fits;
}
int plyolo_conv2d_dgrad_red(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate, const plyolo_bn_red* red, void* stream) {
  if (red == nullptr || red->n == 0) return plyolo_conv2d_dgrad(d, dy, wpd, dx, accumulate, stream);
  if (check_conv(d, "conv2d_dgrad_red", false)) return -1;
  PLY_CHECK_ARG(d->dtype == PLYOLO_BF16, "conv2d_dgrad_red: bf16 only");
  if (check_red(red, d, "conv2d_dgrad_red")) return -1;
  if (is_pointwise(d)) return conv_pw_dgrad(d, dy, wpd, dx, accumulate, red, stream, nullptr);
  return conv_mfma_dgrad(d, dy, wpd, dx, accumulate, red, stream, nullptr);
}
int plyolo_conv2d_dgrad_bn_fits(const plyolo_conv_desc* d, int act) {
  if (check_conv(d, "conv2d_dgrad_bn_fits", false)) return -1;
  if (d->ksize == 3) return conv_mfma_dgrad_bn_fits(d, act);
  return conv_pw_dgrad_bn_fits(d, act);
}
int plyolo_conv2d_dgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* wpd, void* dx, int accumulate, void* stream) {
  return plyolo_conv2d_dgrad_bn_red(d, f, wpd, dx, accumulate, nullptr, stream);
}
int plyolo_conv2d_dgrad_bn_red(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* wpd, void* dx, int accumulate,
                               const plyolo_bn_red* red, void* stream) {
  if (check_conv(d, "conv2d_dgrad_bn", false)) return -1;
  if (red && red->n > 0 && check_red(red, d, "conv2d_dgrad_bn_red")) return -1;
  PLY_CHECK_ARG(f && f->dout && f->z && f->coef && f->bslots && (f->dz || d->ksize == 3), "conv2d_dgrad_bn: incomplete plyolo_bn_bwd_fuse");
  if (d->ksize == 3) {
    PLY_CHECK_ARG(conv_mfma_dgrad_bn_fits(d, f->act) == 1, "conv2d_dgrad_bn: this 3x3 unit is not covered (ask plyolo_conv2d_dgrad_bn_fits; use plyolo_bn_act_bwd_dz + plyolo_conv2d_dgrad)");
    PLY_CHECK_ARG(f->dout_ld % 8 == 0 && f->z_ld % 8 == 0 && f->z_ld >= d->Cout && (!f->dz || (f->dz_ld % 8 == 0 && f->dz_ld >= d->Cout)) && (!f->fwd_to || (f->fwd_ld % 8 == 0 && f->fwd_ld >= d->Cout && !f->dout2)),
                  "conv2d_dgrad_bn: pitches must be multiples of 8 and hold Cout channels (a forwarded gradient needs a single output-gradient matrix)");
    PLY_CHECK_ARG(!f->dout2 || (f->dout_split > 0 && f->dout_split < d->Cout && f->dout_split % 32 == 0 && f->dout2_ld % 8 == 0), "conv2d_dgrad_bn: bad output-gradient split (3x3 units: a multiple of 32)");
    PLY_CHECK_ARG(f->par_split == 0 || (f->par_split > 0 && f->par_split < d->Cout), "conv2d_dgrad_bn: bad parameter split");
    PLY_CHECK_ARG(wpd && dx, "conv2d_dgrad_bn: null weights / dx");
    return conv_mfma_dgrad_bn(d, f, wpd, dx, accumulate, red, stream);
  }
  PLY_CHECK_ARG(conv_pw_dgrad_bn_fits(d, f->act) == 1, "conv2d_dgrad_bn: this unit is not covered (ask plyolo_conv2d_dgrad_bn_fits; use plyolo_bn_act_bwd_dz + plyolo_conv2d_dgrad)");
  PLY_CHECK_ARG(f->dout_ld % 8 == 0 && f->z_ld % 8 == 0 && f->dz_ld % 8 == 0 && f->dz_ld >= d->Cout && f->z_ld >= d->Cout, "conv2d_dgrad_bn: pitches must be multiples of 8 and hold Cout channels");
  PLY_CHECK_ARG(!f->dout2 || (f->dout_split > 0 && f->dout_split < d->Cout && f->dout_split % 8 == 0 && f->dout2_ld % 8 == 0), "conv2d_dgrad_bn: bad output-gradient split");
  PLY_CHECK_ARG(f->par_split == 0 || (f->par_split > 0 && f->par_split < d->Cout), "conv2d_dgrad_bn: bad parameter split");
  return conv_pw_dgrad_bn(d, f, wpd, dx, accumulate, red, stream);
}
int plyolo_conv2d_wgrad_bn_fits(const plyolo_conv_desc* d, int act) {
  if (check_conv(d, "conv2d_wgrad_bn_fits", false)) return -1;
  return conv_mfma_wgrad_bn_fits(d, act);
}
int plyolo_conv2d_wgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, float* dwp, void* stream) {
  if (check_conv(d, "conv2d_wgrad_bn", false)) return -1;
  PLY_CHECK_ARG(f && f->dout && f->z && f->coef && f->bslots && x && dwp, "conv2d_wgrad_bn: incomplete arguments");
  PLY_CHECK_ARG(conv_mfma_wgrad_bn_fits(d, f->act) == 1, "conv2d_wgrad_bn: this unit is not covered (ask plyolo_conv2d_wgrad_bn_fits; use plyolo_bn_act_bwd_dz + plyolo_conv2d_wgrad)");
  PLY_CHECK_ARG(f->dout_ld % 8 == 0 && f->z_ld % 8 == 0 && f->z_ld >= d->Cout && f->dout_ld >= d->Cout && !f->dout2 && f->par_split == 0,
                "conv2d_wgrad_bn: pitches must be multiples of 8 and hold Cout channels; one output-gradient matrix, one parameter set");
  return conv_mfma_wgrad_bn(d, f, x, dwp, stream);
}
int plyolo_conv2d_bwd_pw_fits(const plyolo_conv_desc* d, int act) {
  if (check_conv(d, "conv2d_bwd_pw_fits", false)) return -1;
  return conv_pw_bwd_fits(d, act);
}
int plyolo_conv2d_bwd_pw_slabs(const plyolo_conv_desc* d) {
  if (check_conv(d, "conv2d_bwd_pw_slabs", false)) return -1;
  return conv_pw_bwd_slabs(d);
}
int plyolo_conv2d_bwd_pw(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, const void* wpd, void* dx, int accumulate,
                         float* dwp, void* stream) {
  return plyolo_conv2d_bwd_pw_red(d, f, x, wpd, dx, accumulate, dwp, nullptr, stream);
}
int plyolo_conv2d_bwd_pw_red(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, const void* wpd, void* dx, int accumulate,
                             float* dwp, const plyolo_bn_red* red, void* stream) {
  if (check_conv(d, "conv2d_bwd_pw", false)) return -1;
  if (red && red->n > 0 && check_red(red, d, "conv2d_bwd_pw_red")) return -1;
  PLY_CHECK_ARG(f && f->dout && f->z && f->coef && f->bslots && x && wpd && dx && dwp, "conv2d_bwd_pw: incomplete arguments");
  PLY_CHECK_ARG(conv_pw_bwd_fits(d, f->act) == 1, "conv2d_bwd_pw: this unit is not covered (ask plyolo_conv2d_bwd_pw_fits; use plyolo_bn_act_bwd_dz + plyolo_conv2d_dgrad + plyolo_conv2d_wgrad)");
  PLY_CHECK_ARG(f->dout_ld % 8 == 0 && f->z_ld % 8 == 0 && f->z_ld >= d->Cout, "conv2d_bwd_pw: pitches must be multiples of 8 and hold Cout channels");
  PLY_CHECK_ARG(!f->dout2 || (f->dout_split > 0 && f->dout_split < d->Cout && f->dout_split % 8 == 0 && f->dout2_ld % 8 == 0), "conv2d_bwd_pw: bad output-gradient split");
  PLY_CHECK_ARG(f->par_split == 0 || (f->par_split > 0 && f->par_split < d->Cout), "conv2d_bwd_pw: bad parameter split");
  return conv_pw_bwd(d, f, x, wpd, dx, accumulate, dwp, red, stream);
}
int plyolo_conv2d_wgrad_slabs(const plyolo_conv_desc* d) {
  if (check_conv(d, "conv2d_wgrad_slabs", false)) return -1;
  return d->dtype == PLYOLO_BF16 ? conv_mfma_wgrad_slabs(d) : 1;
}
int plyolo_conv2d_wgrad(const plyolo_conv_desc* d, const void* x, const void* dy, float* dwp, void* stream) {
  if (check_conv(d, "conv2d_wgrad", false)) return -1;
  return d->dtype == PLYOLO_BF16 ? conv_mfma_wgrad(d, x, dy, dwp, stream) : conv_ref_wgrad(d, x, dy, dwp, stream);
}

}  // extern "C"
