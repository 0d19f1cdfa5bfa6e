/*
 * plyolo.h -- C ABI of libplyolo_hip.so: the MI355X (gfx950) kernels behind the
 * pl_YOLO detector hot path  OneStageD.forward(x, labels)  + eval decode / NMS.
 *
 * Boundary rules:
 *   - plain pointers + sizes only, no torch types; every buffer is allocated by
 *     the caller (device memory), the library owns no tensor state;
 *   - every op enqueues on the caller's hipStream_t (passed as void*), never
 *     synchronises, never allocates, never throws across the boundary;
 *   - return value 0 = ok, negative = error; text via plyolo_last_error()
 *     (thread local);
 *   - while a plan is being recorded on the calling thread (plyolo_plan_begin)
 *     op calls are RECORDED into the plan instead of launched; the plan replays
 *     them (plyolo_plan_run) or replays them from an instantiated hipGraph
 *     (plyolo_plan_graph_launch).  Descriptors are copied at record time.
 *
 * Layouts: activations are NHWC ("pixel rows" of C channels, row pitch `ld`
 * elements so that channel-concatenation is a strided write, not a copy);
 * dtype PLYOLO_BF16 = bf16 storage + bf16 MFMA + fp32 accumulation,
 * PLYOLO_F32 = fp32 storage and arithmetic (parity mode).  Convolution weights
 * are consumed in packed form ([tap][Cout][Cin] for fwd, [tap][Cin][Cout] for
 * dgrad), produced from the torch OIHW fp32 master copy by plyolo_pack_weights.
 *
 * Each entry point cites the reference code it replaces (file:line into
 * Iywie/pl_YOLO).  The reference has no native FFI: these are the bindings its
 * Python plugin layer (PL_Modules/build_detection.py) reaches through
 * pl_yolo_amd/_lib.py (ctypes) -- see INTEGRATION.md.
 */
#ifndef PLYOLO_H
#define PLYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PLYOLO_BF16 0
#define PLYOLO_F32 1

#define PLYOLO_ACT_NONE 0
#define PLYOLO_ACT_SILU 1
#define PLYOLO_ACT_RELU 2
#define PLYOLO_ACT_LRELU 3 /* slope 0.1, models/layers/activation.py:13 */
#define PLYOLO_ACT_HSWISH 4 /* x * relu6(x + 3) / 6, models/layers/activation.py:22-26 */
#define PLYOLO_ACT_GELU 5 /* nn.GELU(): 0.5 x (1 + erf(x / sqrt 2)), models/layers/activation.py:16-17 */

/* ---------------------------------------------------------------- probes */
/* ABI version: bumped whenever a struct of this header changes size or layout, or an entry point changes its arguments (the structs
 * carry no size field).  A host compiled against this header must find plyolo_version() == PLYOLO_ABI_VERSION in the library it loads
 * and refuse to run otherwise -- e.g. plyolo_bn_bwd_fuse grew `fwd_to` / `fwd_ld` in version 5: an older host calling
 * plyolo_conv2d_dgrad_bn on a 3x3 unit would make the library read past its struct.  Hosts zero-initialise every struct they pass.
 *   1-4  rounds 1-4 (not tracked per change)    5  plyolo_bn_bwd_fuse.fwd_to / fwd_ld    6  this header */
#define PLYOLO_ABI_VERSION 6
int plyolo_version(void);            /* == PLYOLO_ABI_VERSION of the header the library was built from */
const char* plyolo_arch(void);       /* "gfx950" */
/* What this build of the library carries beyond the default kernels: PLYOLO_BUILD_OPTIN = the measured-slower opt-in paths
 * (lazy-input instances behind plyolo_conv_desc.x_coef, the weights-stationary 3x3 kernel, the tap-row split of the weight
 * gradient; `make OPTIN=1`), PLYOLO_BUILD_DIAG = the compile-time ablation instances (`make DIAG=1`). */
#define PLYOLO_BUILD_OPTIN 1
#define PLYOLO_BUILD_DIAG 2
int plyolo_build_flags(void);
const char* plyolo_last_error(void); /* thread-local message of the last failure */

/* ------------------------------------------------------------------ plans */
typedef struct plyolo_plan plyolo_plan;
plyolo_plan* plyolo_plan_create(void);
void plyolo_plan_destroy(plyolo_plan*);
int plyolo_plan_begin(plyolo_plan*);           /* start recording on this thread */
int plyolo_plan_end(plyolo_plan*);             /* stop recording */
/* Lanes: launches recorded after plyolo_plan_lane(p, l) belong to lane l (default 0).  Launches of
 * one lane stay ordered; different lanes run concurrently under plyolo_plan_run (each lane is issued on
 * its own stream, forked from / joined into the caller's stream).  hipGraph replays are captured on
 * one stream in recorded order.  Order
 * across lanes is expressed with events: ev = plyolo_plan_record(p, lane) marks "everything recorded
 * so far on `lane`"; plyolo_plan_wait(p, lane2, ev) makes the later launches of lane2 wait for it.
 * plyolo_plan_run replays eagerly on real streams with the same fork / events / join;
 * plyolo_plan_profile times the launches one by one on a single stream in recorded order, which must
 * therefore be a valid serial order (record before wait). */
int plyolo_plan_lane(plyolo_plan* p, int lane);
int plyolo_plan_record(plyolo_plan* p, int lane);          /* returns the event id (>= 0) or < 0 */
int plyolo_plan_wait(plyolo_plan* p, int lane, int ev);
int plyolo_plan_lanes(const plyolo_plan* p);                /* number of lanes the plan uses (>= 1) */
/* Host hooks: plyolo_plan_hook(p, lane, id) marks a point of `lane`; when an EAGER replay (plyolo_plan_run) reaches it,
 * fn(id, stream of that lane, user) runs on the host thread.  Whatever the callback enqueues on that stream is ordered
 * after the lane's earlier launches and before the plan's join -- the data-parallel runner issues each gradient
 * bucket's RCCL all-reduce this way, while the rest of the backward plan is still executing.  hipGraph
 * replays and plyolo_plan_profile skip hooks.  fn returns 0, anything else aborts the replay. */
int plyolo_plan_hook(plyolo_plan* p, int lane, int id);
int plyolo_plan_set_hook(plyolo_plan* p, int (*fn)(int id, void* stream, void* user), void* user);
int plyolo_plan_hooks(const plyolo_plan* p);                /* number of recorded hooks */
int plyolo_plan_size(const plyolo_plan*);      /* number of recorded launches */
int plyolo_plan_run(plyolo_plan*, void* stream); /* replay eagerly */
int plyolo_plan_graph_instantiate(plyolo_plan*, void* stream); /* capture into a hipGraphExec */
int plyolo_plan_graph_launch(plyolo_plan*, void* stream);
/* Measurement aid: replay with a hipEvent pair around every launch (synchronises the
 * stream); ms_out[i] = device time of launch i.  plan_op_info returns the launch's
 * kernel label and its ALGORITHMIC flops / HBM bytes (what roofline.achieved is priced on). */
int plyolo_plan_profile(plyolo_plan*, void* stream, float* ms_out, int n);
int plyolo_plan_op_info(const plyolo_plan*, int i, char* label, int label_cap, double* flops, double* bytes);
/* diagnostics: one eager multi-lane replay with a timing event behind every `every`-th launch of `lane`; ms_out[k] = ms from the
 * start of the replay to stamp k, op_out[k] = index of the recorded op it follows; returns the stamp count (<= cap).  Synchronises. */
int plyolo_plan_stamp_times(plyolo_plan*, void* stream, int lane, int every, float* ms_out, int* op_out, int cap);
/* launch lane of recorded op i (plyolo_plan_lane at record time), or a negative code */
int plyolo_plan_op_lane(const plyolo_plan*, int i);
/* Measurement aid for multi-lane plans: one eager multi-stream replay; ms_out[l] = time from the start of the
 * replay to the end of lane l's last launch (l < lanes), ms_out[lanes] = to the join.  Synchronises the stream. */
int plyolo_plan_lane_times(plyolo_plan*, void* stream, float* ms_out, int n);

/* ------------------------------------------------------------ RCCL gradient exchange
 * Data-parallel training exchanges exactly one thing: the mean of the flat fp32 gradient buffer, bucket by bucket
 * (the reference leaves this to torch DDP: PL_Modules/pl_detection.py + Lightning's ddp strategy).  `comm` is the host's
 * ncclComm_t; ncclAllReduce is resolved at run time from the RCCL already in the process, or from the library named with
 * plyolo_rccl_set_library (NULL = "librccl.so").  In place, enqueue-only, on the caller's stream. */
int plyolo_rccl_set_library(const char* path);
int plyolo_rccl_allreduce_bucket(void* comm, float* grads, size_t count, int average, void* stream);

/* ------------------------------------------------------------ convolution
 * Replaces nn.Conv2d inside BaseConv (models/layers/network_blocks.py:18-26)
 * and the bare prediction convs of DecoupledHead (models/heads/decoupled_head.py:43-62)
 * -- forward, and what autograd's convolution_backward does for them. */
typedef struct plyolo_conv_desc {
  int dtype;          /* PLYOLO_BF16 | PLYOLO_F32: storage of x / y / packed w */
  int N, H, W;        /* input batch and spatial size */
  int Cin, Cout;
  int ksize;          /* 1 or 3 (square, pad (k-1)/2) */
  int stride;         /* 1 or 2 */
  int x_ld, y_ld;     /* pixel pitch (elements) of x and y */
  int y_f32;          /* 1: y (fwd) / dy (bwd) is fp32 regardless of dtype (head preds) */
  /* Lazy input (training plans): x is the RAW output z of the producing BaseConv's convolution and its
   * BatchNorm + activation (network_blocks.py:30-37) are applied while x is staged -- forward and weight
   * gradient read act(x[c] * x_coef[c] + x_coef[x_coef_ld + c]) wherever they would read x[c] (zero padding
   * stays zero); the activated tensor is never written.  x_coef = NULL: x is used as stored.  x_coef points
   * at channel 0 of x inside a planar (scale | shift | ...) table with row pitch x_coef_ld
   * (plyolo_bn_finalize).  Ignored by plyolo_conv2d_dgrad. */
  const float* x_coef;
  int x_coef_ld;
  int x_act;          /* PLYOLO_ACT_* applied after the affine */
} plyolo_conv_desc;

/* y[n,oh,ow,co] = sum x[n,oh*s+kh-p,ow*s+kw-p,ci] * w[co,ci,kh,kw] (+ bias[co]).
 * wp: packed fwd weights.  bias: fp32[Cout] or NULL.
 * stats: NULL, or the fp64 stat slots [PLYOLO_STAT_SLOTS][2][Cout] (see the BatchNorm section;
 * zeroed by the caller) that receive the sums of y and y^2 (train-mode BatchNorm statistics,
 * fused epilogue). */
int plyolo_conv2d_fwd(const plyolo_conv_desc* d, const void* x, const void* wp, const float* bias,
                      void* y, double* stats, void* stream);
/* Inference-mode BaseConv in ONE launch (bf16 path): y = act(conv(x) * coef[0:Cout] + coef[Cout:2*Cout]) with
 * coef from plyolo_bn_eval_coef -- BatchNorm and the activation are applied to the accumulators in the
 * epilogue, the raw convolution output is never written. */
int plyolo_conv2d_fwd_bn_act(const plyolo_conv_desc* d, const void* x, const void* wp, const float* coef, int act,
                             const void* res, int r_ld, void* y, void* stream);   /* res: + residual after act, or NULL */
/* dx = conv_transpose(dy, w).  wpd: packed dgrad weights.  accumulate!=0: dx += */
int plyolo_conv2d_dgrad(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx,
                        int accumulate, void* stream);
/* Weight gradient, written as plyolo_conv2d_wgrad_slabs(d) partial slabs
 * dwp[slab][tap][Cout][Cin] (fp32): slab s holds the contribution of one spatial split.
 * plyolo_unpack_wgrads sums the slabs in a fixed order (deterministic; no fp32 atomics).
 * bf16 path: every slab element is overwritten (no zero-fill needed).  fp32 parity path:
 * one slab, accumulated with atomics -> the caller zeroes it first. */
int plyolo_conv2d_wgrad_slabs(const plyolo_conv_desc* d);
int plyolo_conv2d_wgrad(const plyolo_conv_desc* d, const void* x, const void* dy, float* dwp, void* stream);
/* dbias[co] = sum over pixels of dy[m][co] (head prediction convs, decoupled_head.py:43-62). */
int plyolo_bias_grad(int dtype, const void* dy, int M, int C, int ld, float* dbias, void* stream);
/* The same sums for up to PLYOLO_BIAS_JOBS_MAX matrices in one launch pair (the prediction convs of all head levels).  `jobs` is a
 * HOST array (copied into the launch); every matrix must satisfy the vector-path conditions: rows 16-byte aligned, ld a multiple of
 * the 16-byte vector length and >= C rounded up to it.  nblk is filled by the library. */
#define PLYOLO_BIAS_JOBS_MAX 8
typedef struct plyolo_bias_job {
  const void* dy;
  int M, C, ld;
  float* db;
  int nblk;
} plyolo_bias_job;
int plyolo_bias_grad_multi(int dtype, const plyolo_bias_job* jobs, int njobs, void* stream);

/* Weight (re)packing, one launch for a whole table of convolutions. The table
 * lives in DEVICE memory: n entries of plyolo_pack_entry. */
typedef struct plyolo_pack_entry {
  const float* w;   /* OIHW fp32 master weights (torch layout), [Cout][Cin][k][k] */
  void* wp;         /* fwd pack, plyolo_pack_elems() elements: fp32 [tap][Cout_total][Cin_p];
                       bf16 = MFMA B-fragment order [tap][ceil(Cout_total/32)][ceil(Cin_p/16)][64][8] */
  void* wpd;        /* dgrad pack or NULL: fp32 [tap][Cin_p][Cout_p8];
                       bf16 [tap][ceil(Cin_p/32)][ceil(Cout_total/16)][64][8] */
  float* dwp;       /* [nslab][tap][Cout_total][Cin_p] fp32 wgrad slabs (unpack source) */
  float* dw;        /* OIHW fp32 gradient (unpack destination), or NULL */
  const float* b;   /* fp32 bias [Cout] or NULL */
  float* bp;        /* packed bias [Cout_total] */
  float* dbp;       /* packed bias gradient [Cout_total] */
  float* db;        /* bias gradient [Cout] or NULL */
  int Cout, Cin, Cin_p, ksize;
  int Cout_total;   /* rows of the packed matrix (several torch convs may share one
                       packed conv: reg_preds(4)+obj_preds(1), decoupled_head.py:55-62) */
  int Cout_p8;      /* Cout_total rounded up to 8 (dgrad contraction length) */
  int co_off;       /* first packed row of this entry */
  int nslab;        /* number of wgrad slabs to sum in unpack (>= 1) */
  int blk0, nblk;   /* flat launches (plyolo_pack_plan): this entry owns workgroups [blk0, blk0 + nblk) -- its share of the
                       work, so that the 512-channel layers do not run on 64 workgroups while the rest of the chip idles */
} plyolo_pack_entry;
int plyolo_pack_weights(const plyolo_pack_entry* table_dev, int n, int dtype, int max_elems, void* stream);
/* Element counts of the two packs (the buffers must be zero-initialised ONCE: pad positions are
 * never written). */
int plyolo_pack_elems(int dtype, int Cout_total, int Cin_p, int ksize, size_t* wp_elems, size_t* wpd_elems);
/* dwp slab 0 += slabs 1..nslab-1 in a fixed order (elems = floats per slab, multiple of 4): the
 * host may run this right after a wgrad launch and then unpack with nslab = 1. */
int plyolo_reduce_slabs(float* dwp, int nslab, size_t elems, void* stream);
/* The same fold for several weight tensors in two launches (all slab groups, then all group heads): per / groups from
 * plyolo_reduce_slabs_plan, so the summation order -- and the result -- equals plyolo_reduce_slabs' bit for bit. */
typedef struct plyolo_reduce_job {
  float* dwp;
  int nslab, per, groups;
  size_t elems;
} plyolo_reduce_job;
int plyolo_reduce_slabs_plan(int nslab, size_t elems, int* per, int* groups);
int plyolo_reduce_slabs_multi(const plyolo_reduce_job* jobs_dev, int njobs, int max_cols, int max_groups, double total_bytes,
                              void* stream);

/* dw (OIHW) (+)= permute(sum of the nslab slabs of dwp) for the whole table. */
int plyolo_unpack_wgrads(const plyolo_pack_entry* table_dev, int n, int max_elems, int accumulate, void* stream);
/* Load-balanced forms of the two launches above.  plyolo_pack_plan fills blk0 / nblk of a HOST copy of the table in
 * proportion to each entry's element count and returns the total number of workgroups; the table is then uploaded and
 * plyolo_pack_weights_flat / plyolo_unpack_wgrads_flat launch exactly that many (results identical to the unbalanced forms). */
int plyolo_pack_plan(plyolo_pack_entry* table_host, int n);
int plyolo_pack_weights_flat(const plyolo_pack_entry* table_dev, int n, int dtype, int total_blocks, void* stream);
int plyolo_unpack_wgrads_flat(const plyolo_pack_entry* table_dev, int n, int total_blocks, int accumulate, void* stream);

/* ------------------------------------------------- BatchNorm + activation
 * Replaces nn.BatchNorm2d(eps=1e-3, momentum=0.03) + SiLU of BaseConv
 * (models/layers/normalization.py:8, network_blocks.py:30-37) fwd and bwd.
 *
 * Per-channel batch statistics travel in STAT SLOTS: fp64 [PLYOLO_STAT_SLOTS][2][C]
 * accumulators (sum, sum of squares) that the producing kernel adds to with agent-scope fp64
 * atomics (one add per workgroup and channel; fp64 makes the order of the adds irrelevant to
 * the fp32 result).  The CALLER zeroes the slots before the producing launch.  The consuming
 * kernel sums the slots itself, so there is no separate finalize launch on the hot path. */
#define PLYOLO_STAT_SLOTS 8
typedef struct plyolo_bn_stats {
  const double* slots;        /* [PLYOLO_STAT_SLOTS][2][C] filled by plyolo_conv2d_fwd */
  double count;               /* N*OH*OW */
  const float* gamma;         /* may be NULL (1) */
  const float* beta;          /* may be NULL (0) */
  float eps, momentum;
  float* running_mean;        /* updated in place when non-NULL (unbiased var, like torch) */
  float* running_var;
  int64_t* num_batches_tracked;
  /* Two BatchNorm modules normalising one merged convolution (CSP conv1 || conv2, same input):
   * channels [split, C) take their affine parameters / running statistics from the second set,
   * indexed from 0.  split == 0: unused. */
  int split;
  const float* gamma2;
  const float* beta2;
  float* running_mean2;
  float* running_var2;
  int64_t* num_batches_tracked2;
} plyolo_bn_stats;
/* Second destination (or source) of a channel-split activation matrix: columns [split, C) live at
 * p2 (pitch ld2), indexed from 0.  NULL or split == 0: one matrix. */
typedef struct plyolo_split {
  int split;
  void* p2;
  int ld2;
  /* plyolo_bn_act_bwd_dz only: the incoming gradient dout also flows to a second consumer -- the shortcut of a Bottleneck
   * (network_blocks.py:89-90: y = conv2(conv1(x)) + x) -- and is copied (fwd_acc == 0) or added (fwd_acc != 0) into fwd_to
   * [M][C] (pitch fwd_ld) by the same pass that reads it; NULL: nothing.  Not combined with a channel split. */
  void* fwd_to;
  int fwd_ld;
  int fwd_acc;
} plyolo_split;
/* Standalone reduction of the slots -> coef[0:C]=scale, [C:2C]=shift, [2C:3C]=mean, [3C:4C]=invstd
 * (+ running statistics).  Not needed when plyolo_bn_act_fwd is given `st`. */
int plyolo_bn_finalize(const plyolo_bn_stats* st, int C, float* coef, void* stream);
/* eval mode: coef from running statistics */
int plyolo_bn_eval_coef(int C, const float* gamma, const float* beta, const float* running_mean,
                        const float* running_var, float eps, float* coef, void* stream);
/* ------------------------------------------------------------ e-yolox family: depthwise 3x3 conv, bicubic x2 upsample
 * Depthwise conv = nn.Conv2d(C, C, 3, 1, 1, groups=C, bias=False) inside BaseConv (models/backbones/ecmnet.py:157,160;
 * models/necks/pafpn_al.py:162,165) forward / data gradient / weight gradient; NHWC activations with pixel pitch *_ld,
 * fp32 master weights [C][1][3][3] read directly (rounded to bf16 in bf16 mode like the MFMA weight packs).
 * stats (fwd): NULL or the fp64 stat slots [PLYOLO_STAT_SLOTS][2][C] receiving sum / sum of squares of y. */
int plyolo_dwconv3x3_fwd(int dtype, int N, int H, int W, int C, const void* x, int x_ld, const float* w, void* y, int y_ld,
                         double* stats, void* stream);
int plyolo_dwconv3x3_dgrad(int dtype, int N, int H, int W, int C, const void* dy, int dy_ld, const float* w, void* dx, int dx_ld,
                           int accumulate, void* stream);
/* dw[c][tap] (+)= sum_pixels dz * x: `partial` holds plyolo_dwconv3x3_wgrad_blocks() fp32 slabs of [C][9] (per-workgroup
 * partials, folded in a fixed order by a second launch: deterministic, no atomics). */
int plyolo_dwconv3x3_wgrad_blocks(int dtype, int N, int H, int W, int C);
int plyolo_dwconv3x3_wgrad(int dtype, int N, int H, int W, int C, const void* x, int x_ld, const void* dz, int dz_ld, float* partial,
                           float* dw, int accumulate, void* stream);
/* nn.Upsample(scale_factor=2, mode="bicubic") (models/necks/pafpn_al.py:25), align_corners=False, A = -0.75; in [N,H,W,C]
 * -> out [N,2H,2W,C]; backward = exact transpose as a gather (din (+)= ...). */
int plyolo_bicubic2x_fwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, void* out, int o_ld, void* stream);
int plyolo_bicubic2x_bwd(int dtype, int N, int H, int W, int C, const void* dout, int d_ld, void* din, int i_ld, int accumulate,
                         void* stream);

/* ------------------------------------------------------------ GPU input pipeline (per-image transforms of a batch)
 * Replaces augment_hsv + _mirror + preproc of models/data/augmentation/data_augments.py:88-133: for every image descriptor the
 * uint8 HWC BGR source is (optionally) colour-jittered in HSV, mirrored, resized by the letterbox ratio r (bilinear, 8-bit
 * fixed point like cv2.resize), padded with 114 to out_h x out_w and written as fp32 CHW (values 0..255, BGR order) --
 * out [B, 3, out_h, out_w], exactly what OneStageD.forward takes.  The OpenCV algorithms are restated (csrc/augment.hip). */
typedef struct plyolo_aug_image {
  const unsigned char* src;   /* device pointer, [h][w][3] uint8 */
  int h, w;
  int dh, dw;                 /* resized extent int(h * r), int(w * r) with r = min(out_h / h, out_w / w), computed by the
                               * host in float64 exactly like preproc (data_augments.py:93-97) */
  int flip;                   /* mirror horizontally (data_augments.py:129-133) */
  int hsv;                    /* apply the HSV jitter with the three gains below (data_augments.py:113-127) */
  double hgain, sgain, vgain; /* r = uniform(-1, 1, 3) * [0.015, 0.7, 0.4] + 1  (float64, as numpy draws them) */
} plyolo_aug_image;
int plyolo_preproc_batch(const plyolo_aug_image* imgs_dev, int B, int out_h, int out_w, float* out, void* stream);

/* Mosaic / random affine / mixup, pixel side (reference models/data/mosaic_detection.py: mosaic :61-118, cv2.warpAffine inside
 * random_perspective :328-344, mixup :169-247).  uint8 HWC BGR device images; the small descriptors are HOST memory and are
 * copied into the launch.  The random decisions and the label arithmetic are the host's (pl_yolo_amd/data.py). */
typedef struct plyolo_mosaic_tile {
  const unsigned char* src;   /* source image [h, w, 3] */
  int h, w;
  int dh, dw;                 /* its size after cv2.resize(..., INTER_LINEAR): int(h * scale), int(w * scale) */
  int lx1, ly1, lx2, ly2;     /* rectangle of the 2H x 2W canvas it fills (get_mosaic_coordinate :256-274) */
  int sx1, sy1;               /* top-left of the visible part inside the resized image */
} plyolo_mosaic_tile;
/* canvas [canvas_h, canvas_w, 3] = 114, except the four rectangles */
int plyolo_mosaic4(const plyolo_mosaic_tile* tiles_host, int canvas_h, int canvas_w, unsigned char* canvas, void* stream);
/* cv2.warpAffine(src, M, dsize=(dw, dh), borderValue=border) with inv6_host = the INVERSE of M (cv::invertAffineTransform, 6 doubles) */
int plyolo_warp_affine_u8(const unsigned char* src, int sh, int sw, const double* inv6_host, unsigned char* dst, int dh, int dw, int border,
                          void* stream);
/* cv2.warpPerspective(src, M, dsize=(dw, dh), borderValue=border) with inv9_host = the INVERSE of the 3x3 M (cv::invert, 9 doubles):
 * what random_perspective calls when `perspective` is non-zero (mosaic_detection.py:319-323) */
int plyolo_warp_perspective_u8(const unsigned char* src, int sh, int sw, const double* inv9_host, unsigned char* dst, int dh, int dw, int border,
                               void* stream);
/* dst [out_h, out_w, 3]: cv2.resize(src, (dw, dh)) in the top-left corner, `pad` elsewhere */
int plyolo_resize_pad_u8(const unsigned char* src, int h, int w, int dh, int dw, unsigned char* dst, int out_h, int out_w, int pad, void* stream);
/* out [th, tw, 3] = uint8(0.5 * origin + 0.5 * crop): `other` [bh, bw, 3] (mirrored if flip), zero-padded to >= th x tw, cut at (y_off, x_off) */
int plyolo_mixup_blend_u8(const unsigned char* origin, int th, int tw, const unsigned char* other, int bh, int bw, int flip, int x_off, int y_off,
                          unsigned char* out, void* stream);

/* Rounding cut-out (reference models/data/augmentation/cutout_round.py:6-55; MosaicDetection applies it to an image before the
 * mosaic, mosaic_detection.py:90-91, 160-161).  The host draws the holes (numpy.random, in the reference's order), tests them
 * against the label boxes (bbox_ioa, models/utils/bbox.py:76-94) and forms the fill colour from the strip sums below. */
/* sums_dev[r][c] (3 per rectangle) = sum of channel c over rectangle r = rects_dev[r] = {y0, y1, x0, x1} (half-open, inside the image) */
int plyolo_rect_sums_u8(const unsigned char* img, int H, int W, const int* rects_dev, int n, unsigned long long* sums_dev, void* stream);
#define PLYOLO_MAX_HOLES 8
typedef struct plyolo_rect { int x1, y1, x2, y2; } plyolo_rect;   /* half-open, inside the image */
/* img[y1:y2, x1:x2] = uint8(mixup * fill + (1 - mixup) * img[y1:y2, x1:x2]) for every hole IN ORDER (float64, two rounded products
 * and their rounded sum, truncated); holes_host / fill3_host are HOST memory, copied into the launch */
int plyolo_cutout_holes_u8(unsigned char* img, int H, int W, const plyolo_rect* holes_host, int n, const double* fill3_host, double mixup,
                           void* stream);

/* ------------------------------------------------------------ deploy-time folding (inference export)
 * Replaces RepConv._fuse_bn_tensor / get_equivalent_kernel_bias / fuse_conv_bn / fuse_repvgg_block
 * (models/necks/yolov7_neck.py:213-348) and prepares BaseConv.fuseforward (network_blocks.py:39-40): fp32 weights in
 * the torch layout [Cout][Cin][k][k], fp32 results, one launch each. */
typedef struct plyolo_bn_params {
  const float* gamma;         /* [C] or NULL (1) */
  const float* beta;          /* [C] or NULL (0) */
  const float* running_mean;  /* [C] */
  const float* running_var;   /* [C] */
  float eps;
} plyolo_bn_params;
/* w_out[co][k] = w[co][k] * gamma[co] / sqrt(var[co] + eps);  b_out[co] = beta[co] - mean[co] * gamma[co] / sqrt(var[co] + eps)
 * (+ conv_bias[co] * gamma / std when the convolution has a bias).  K = Cin * ksize * ksize. */
int plyolo_fold_conv_bn(const float* w, const float* conv_bias, const plyolo_bn_params* bn, int Cout, int K, float* w_out,
                        float* b_out, void* stream);
/* One 3x3 kernel + bias equivalent to  bn3(conv3x3(x)) + bn1(conv1x1(x)) [+ bn_id(x)]  in inference mode:
 * w3 [Cout][Cin][3][3], w1 [Cout][Cin]; bn_id NULL when the block has no identity branch (needs Cout == Cin otherwise). */
int plyolo_repconv_fuse(const float* w3, const plyolo_bn_params* bn3, const float* w1, const plyolo_bn_params* bn1,
                        const plyolo_bn_params* bn_id, int Cout, int Cin, float* w_out, float* b_out, void* stream);
/* coef = (1 | bias | 0 | 1) [4][C]: the fused-epilogue coefficients of a BatchNorm-free conv unit with bias
 * (plyolo_conv2d_fwd_bn_act with scale 1, shift = bias) -- the deploy form of BaseConv / RepConv. */
int plyolo_bias_coef(int C, const float* bias, float* coef, void* stream);

/* same, for one module of a merged convolution: coef rows are C_total wide, this module's C channels
 * start at column c_off */
int plyolo_bn_eval_coef_at(int C, const float* gamma, const float* beta, const float* running_mean,
                           const float* running_var, float eps, float* coef, int C_total, int c_off, void* stream);
/* out = act(z*scale+shift) (+ res).  z [M][C] pitch z_ld; out pitch o_ld; res pitch r_ld or NULL.
 * st == NULL: coef is an INPUT.  st != NULL (train mode): every workgroup derives scale/shift from
 * the stat slots, and coef (the 4C values above, needed by the backward kernels) plus the running
 * statistics are WRITTEN by this launch. */
int plyolo_bn_act_fwd(int dtype, int M, int C, const void* z, int z_ld, float* coef, int act,
                      const void* res, int r_ld, void* out, int o_ld, const plyolo_bn_stats* st,
                      const plyolo_split* out_split, void* stream);
/* slots += per-channel (sum, sum of squares) of an activation matrix x [M][C] (pitch x_ld): statistics of
 * a BatchNorm applied directly to a tensor (RepConv's identity branch, yolov7_neck.py:191). */
int plyolo_channel_stats(int dtype, int M, int C, const void* x, int x_ld, double* slots, void* stream);
/* din (+)= dout * act'(z): backward of a bare activation layer (forward = plyolo_bn_act_fwd with coef NULL). */
int plyolo_act_bwd(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld, int act,
                   void* din, int di_ld, int accumulate, void* stream);
/* bslots (fp64 [PLYOLO_STAT_SLOTS][2][C], zeroed by the caller) += sum du, sum du*zhat
 * with du = dout * act'(u) */
int plyolo_bn_act_bwd_reduce(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld,
                             const float* coef, int act, double* bslots, const plyolo_split* dout_split,
                             void* stream);
/* dz = A*du + B*z + Cc with A,B,Cc derived from bslots inside the launch; also writes
 * dgamma / dbeta (fp32 [C], may be NULL; accumulate != 0: +=). */
/* par2 (optional): {split, gamma2, dgamma2, dbeta2} for channels [split, C) of a merged convolution. */
typedef struct plyolo_bn_bwd_split {
  int split;
  const float* gamma2;
  float* dgamma2;
  float* dbeta2;
} plyolo_bn_bwd_split;
int plyolo_bn_act_bwd_dz(int dtype, int M, int C, const void* dout, int d_ld, const void* z, int z_ld,
                         const float* coef, const double* bslots, const float* gamma, float* dgamma,
                         float* dbeta, int accumulate, int act, void* dz, int dz_ld,
                         const plyolo_split* dout_split, const plyolo_bn_bwd_split* par2, void* stream);

/* Data gradient of a POINTWISE (1x1, stride 1) or -- round 5 -- 3x3 stride-1 BaseConv unit with the unit's own BatchNorm + activation backward fused in:
 * what autograd does for act(bn(conv(x))) between the gradient of the activated output and dx (network_blocks.py:30-37).
 * The rows of dz = plyolo_bn_act_bwd_dz(dout, z) are formed while they are staged for the matrix cores -- dout and z are read
 * once, dz is written once (for plyolo_conv2d_wgrad) and never read back by this path, dx (+)= dz . W.  Bit-identical to the
 * two separate launches.  bslots must hold the sums of plyolo_bn_act_bwd_reduce; dgamma / dbeta are written (not accumulated).
 * plyolo_conv2d_dgrad_bn_fits: 1 if the unit is covered (bf16, 1x1 stride 1, act none/silu/relu/lrelu, Cout % 8 == 0, Cin spans
 * at most two output blocks of the kernel; or 3x3 stride 1, SiLU, Cout % 8 == 0, Cout <= 512, tiles with a loader instance --
 * csrc/conv_mfma_bnb.hip: the halo tile is requested as (dout, z) pairs and staged as dz, padding stays zero), else 0. */
typedef struct plyolo_bn_bwd_fuse {
  const void* dout; int dout_ld;        /* gradient of the activated output [M][Cout] */
  const void* dout2; int dout2_ld;      /* merged pairs: channels [dout_split, Cout) live here (NULL: one matrix) */
  int dout_split;
  const void* z; int z_ld;              /* raw conv output of the forward */
  const float* coef;                    /* (scale | shift | mean | invstd) [4][Cout] */
  const double* bslots;                 /* [PLYOLO_STAT_SLOTS][2][Cout] */
  const float* gamma; float* dgamma; float* dbeta;
  int par_split;                        /* merged pairs: parameters of channels >= par_split come from the second set (0: none) */
  const float* gamma2; float* dgamma2; float* dbeta2;
  int act;
  void* dz; int dz_ld;                  /* out: [M][Cout] (3x3 units: may be NULL when nothing reads dz) */
  void* fwd_to; int fwd_ld;             /* 3x3 units only, optional: dout is also COPIED here (the Bottleneck shortcut's share of the
                                         * gradient, network_blocks.py:89-90, when this unit is its first writer); NULL: not forwarded */
} plyolo_bn_bwd_fuse;
int plyolo_conv2d_dgrad_bn_fits(const plyolo_conv_desc* d, int act);
int plyolo_conv2d_dgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* wpd, void* dx,
                           int accumulate, void* stream);
/* ... + the BatchNorm-backward reduction of the unit(s) that produced x (struct plyolo_bn_red, below) */
struct plyolo_bn_red;
int plyolo_conv2d_dgrad_bn_red(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* wpd, void* dx,
                               int accumulate, const struct plyolo_bn_red* red, void* stream);

/* The weight gradient of a BaseConv unit WITHOUT a data gradient (the first convolution of a network: nothing upstream takes a
 * gradient) behind plyolo_bn_act_bwd_reduce, with the unit's plyolo_bn_act_bwd_dz in its loader (csrc/conv_wgrad_mfma.hip, BNB
 * instances): such a unit's dz is read by the weight gradient only, so f->dout and f->z are read once and dz never reaches HBM (f->dz
 * is ignored and may be NULL).  Same private slabs as plyolo_bn_act_bwd_dz + plyolo_conv2d_wgrad, bit for bit
 * (plyolo_conv2d_wgrad_slabs(d) of them at dwp); dgamma / dbeta are written as plyolo_bn_act_bwd_dz writes them.
 * plyolo_conv2d_wgrad_bn_fits: 1 if covered (bf16, 3x3 stride 1, SiLU, at most 64 output and 32 input channels), else 0. */
int plyolo_conv2d_wgrad_bn_fits(const plyolo_conv_desc* d, int act);
int plyolo_conv2d_wgrad_bn(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, float* dwp, void* stream);

/* The WHOLE backward of a pointwise BaseConv unit behind plyolo_bn_act_bwd_reduce, in one persistent launch (csrc/conv_pw_bwd.hip):
 * dz = plyolo_bn_act_bwd_dz(dout, z) is formed tile by tile in LDS and feeds both dx (+)= dz . W (== plyolo_conv2d_dgrad, bit for
 * bit) and dW = dz^T . x (== plyolo_conv2d_wgrad up to the order of its fp32 sums); dout, z and x are read once, dz never reaches
 * HBM (f->dz is ignored and may be NULL).  What autograd computes for act(bn(conv1x1(x))), network_blocks.py:18-40, from the
 * gradient of the activated output: dx, the weight gradient, dgamma, dbeta.  x and dx share d->x_ld.  The weight gradient leaves
 * as plyolo_conv2d_bwd_pw_slabs(d) private fp32 slabs [slab][Cout][Cin] at dwp (fold them with plyolo_reduce_slabs).
 * plyolo_conv2d_bwd_pw_fits: 1 if the unit is covered (bf16, 1x1 stride 1, act none/silu/relu/lrelu, Cout and Cin in {32, 64, 128}
 * and equal or 2:1, output gradient >= PLYOLO_PWBWD_MIN_MB), else 0. */
/* BatchNorm-backward reduction of the UPSTREAM unit(s), folded into the data gradient that writes their output gradient LAST
 * (plyolo_conv2d_dgrad_red / plyolo_conv2d_bwd_pw_red): while a data-gradient kernel stores the final dx rows -- the gradient of
 * the activated output a_U of the unit(s) U that produced x -- it also reads U's raw conv output z_U and adds
 *     sum du,  sum du * zhat      (du = dx * act'(z_U * scale + shift), zhat = (z_U - mean) * invstd)
 * to U's fp64 backward stat slots: exactly what plyolo_bn_act_bwd_reduce(dx, z_U) would add in a launch of its own, which then
 * reads dx and z_U again.  Up to PLYOLO_BN_RED_SEGS channel segments [c0, c1) of dx (a concatenated input: one segment per
 * producing unit; a segment without a BatchNorm unit is simply left out); every pointer is pre-offset to the segment's first
 * channel; z [M][..] pitch z_ld; coef rows (scale | shift | mean | invstd) coef_ld apart; bslots [PLYOLO_STAT_SLOTS][2][slot_ld]
 * of the unit (slot_ld = its channel count).  c0 / c1 multiples of 8.  act: none / silu / relu / lrelu (a hswish or gelu unit keeps
 * its plyolo_bn_act_bwd_reduce launch).  The caller zeroes bslots, and must be the LAST writer of those dx channels. */
#define PLYOLO_BN_RED_SEGS 3
typedef struct plyolo_bn_red_seg {
  int c0, c1;
  const void* z; int z_ld;
  const float* coef; int coef_ld;
  double* bslots; int slot_ld;
  int act;
} plyolo_bn_red_seg;
typedef struct plyolo_bn_red {
  int n;
  plyolo_bn_red_seg seg[PLYOLO_BN_RED_SEGS];
} plyolo_bn_red;
/* plyolo_conv2d_dgrad + the reduction above (bf16 only; red == NULL or red->n == 0: plain plyolo_conv2d_dgrad) */
int plyolo_conv2d_dgrad_red(const plyolo_conv_desc* d, const void* dy, const void* wpd, void* dx, int accumulate,
                            const plyolo_bn_red* red, void* stream);
/* 1 when plyolo_conv2d_dgrad_red has a kernel instance for this data gradient (bf16; decided with the launch's own tile selection) */
int plyolo_conv2d_dgrad_red_fits(const plyolo_conv_desc* d);

int plyolo_conv2d_bwd_pw_fits(const plyolo_conv_desc* d, int act);
int plyolo_conv2d_bwd_pw_slabs(const plyolo_conv_desc* d);
int plyolo_conv2d_bwd_pw(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, const void* wpd, void* dx,
                         int accumulate, float* dwp, void* stream);
/* ... + the BatchNorm-backward reduction of the unit(s) that produced x (plyolo_bn_red above) */
int plyolo_conv2d_bwd_pw_red(const plyolo_conv_desc* d, const plyolo_bn_bwd_fuse* f, const void* x, const void* wpd, void* dx,
                             int accumulate, float* dwp, const plyolo_bn_red* red, void* stream);

/* norm = "ln" of BaseConv (reference models/layers/normalization.py:9-10): nn.LayerNorm(out_channels) on an NCHW tensor
 * normalises the last axis, the image WIDTH, with affine parameters gamma[W], beta[W] (torch requires W == out_channels), eps
 * inside the square root, biased variance; the activation of BaseConv follows.  x / out / dout / dx are [N*H*W, ld] NHWC
 * matrices of C channels; stats [N*H*C][2] fp32 (mean, 1/std of every line) is written by the forward and read by the backward.
 * The backward writes (or accumulates into) dx and the fp32 parameter gradients; its sums run in a fixed order. */
int plyolo_lnw_act_fwd(int dtype, int N, int H, int W, int C, const void* x, int x_ld, const float* gamma, const float* beta, float eps, int act,
                       void* out, int o_ld, float* stats, void* stream);
int plyolo_lnw_act_bwd(int dtype, int N, int H, int W, int C, const void* dout, int d_ld, const void* x, int x_ld, const float* stats,
                       const float* gamma, const float* beta, int act, void* dx, int dx_ld, int accumulate_dx, float* dgamma, float* dbeta,
                       int accumulate_params, void* stream);

/* -------------------------------------------------- data movement kernels */
/* Focus space-to-depth (network_blocks.py:50-65): NCHW fp32 image ->
 * NHWC [N,H/2,W/2,Cp] with channel blocks TL,BL,TR,BR (3 each) then zero pad. */
int plyolo_focus_s2d(int dtype, const float* img, int N, int H, int W, void* out, int Cp, void* stream);
/* out[m][0:C] (pitch o_ld) = (accumulate ? out : 0) + in[m][0:C] (pitch i_ld) */
int plyolo_copy_add(int dtype, int M, int C, const void* in, int i_ld, void* out, int o_ld, int accumulate,
                    void* stream);
/* nearest x2 upsample (models/necks/pafpn_csp.py:22) fwd, and its bwd (2x2 sum) */
int plyolo_upsample2x_fwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, void* out, int o_ld,
                          void* stream);
int plyolo_upsample2x_bwd(int dtype, int N, int H, int W, int C, const void* dout, int d_ld, void* din, int i_ld,
                          int accumulate, void* stream);
/* MaxPool2d(k, stride 1, pad k/2) (network_blocks.py:144): fwd, and bwd that
 * routes dout to the FIRST maximum in row-major window order (ATen rule),
 * accumulating into din (fp32 [N,H,W,C] scratch, caller zeroes). */
int plyolo_maxpool_s1_fwd(int dtype, int N, int H, int W, int C, int k, const void* in, int i_ld, void* out,
                          int o_ld, void* stream);
int plyolo_maxpool_s1_bwd(int dtype, int N, int H, int W, int C, int k, const void* in, int i_ld,
                          const void* dout, int d_ld, float* din_f32, void* stream);
/* MaxPool2d(2, stride 2) of the YOLOv7 Transition blocks (models/backbones/eelan.py:126-141,
 * models/necks/yolov7_neck.py:149-164); bwd routes to the first maximum (ATen rule). */
/* Backward of all stride-1 pools of one SPP block (network_blocks.py:141-153 under autograd) in one
 * launch: din (+)= sum_j scatter_kj(douts[j]) with ATen's first-maximum rule.  douts[j] may be NULL
 * (that pool's output received no gradient).  Requires plyolo_spp_pools_bwd_fits(dtype,H,W). */
int plyolo_spp_pools_bwd_fits(int dtype, int H, int W);
/* Forward of the same block in ONE launch: outs[j] = MaxPool2d(ks[j], 1, ks[j]/2)(in) for an increasing cascade of odd windows
 * (5, 9, 13), computed as pool_5 o pool_5 o pool_5 on an (image, 16-byte channel vector) plane in LDS.  _fits: 1 if the plane fits. */
int plyolo_spp_pools_fwd_fits(int dtype, int H, int W, int C, int nk, const int* ks);
int plyolo_spp_pools_fwd(int dtype, int N, int H, int W, int C, int nk, const int* ks, const void* in, int i_ld, void* const* outs,
                         const int* o_lds, void* stream);
int plyolo_spp_pools_bwd(int dtype, int N, int H, int W, int C, int nk, const int* ks, const void* in, int i_ld,
                         const void* const* douts, const int* d_lds, void* din, int di_ld, int accumulate, void* stream);
int plyolo_maxpool2x2_fwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, void* out, int o_ld,
                          void* stream);
int plyolo_maxpool2x2_bwd(int dtype, int N, int H, int W, int C, const void* in, int i_ld, const void* dout,
                          int d_ld, void* din, int di_ld, int accumulate, void* stream);
/* ImplicitHead pieces (models/heads/implicit_head.py:5-62): y = m * (W(x + a) + b).
 * implicit_bias: out[co] = b[co] + sum_ci W[co][ci]*a[ci];  scale_channels: y = m*u;
 * implicit_bwd: du = m*dy (activation dtype, pitch du_ld) + per-block partials of dm = sum dy*u;
 * implicit_param_grads: dm, da = W^T sdu, dW += sdu (x) a, db = sdu  (sdu = column sums of du). */
int plyolo_implicit_bias(const float* W, const float* a, const float* b, float* out, int Cout, int Cin, void* stream);
int plyolo_scale_channels(const float* u, const float* m, float* y, size_t rows, int C, void* stream);
int plyolo_implicit_bwd_blocks(size_t rows);
int plyolo_implicit_bwd(int dtype, const float* dy, const float* u, const float* m, void* du, int du_ld,
                        float* partial, size_t rows, int C, void* stream);
int plyolo_implicit_param_grads(const float* partial, int nblk, const float* W, const float* a, const float* sdu,
                                float* dm, float* da, float* dW, float* db, int Cout, int Cin, void* stream);
/* out (pitch o_ld) (+)= convert(in fp32 [M][C]) */
int plyolo_f32_to_act(int dtype, int M, int C, const float* in, void* out, int o_ld, int accumulate, void* stream);
int plyolo_memset_async(void* p, int value, size_t bytes, void* stream);
/* NHWC activation -> NCHW fp32 (API edge: returning feature maps to torch callers) and back */
int plyolo_nhwc_to_nchw_f32(int dtype, int N, int H, int W, int C, const void* in, int i_ld, float* out,
                            void* stream);
int plyolo_nchw_f32_to_nhwc(int dtype, int N, int H, int W, int C, const float* in, void* out, int o_ld,
                            void* stream);

/* ------------------------------------------------------------- YOLOX loss
 * Replaces YOLOXLoss (models/losses/yolox/yolox_loss.py:20-228),
 * get_in_boxes_info (:231-315), dynamic_k_matching (:318-370),
 * bboxes_iou / IOUloss (models/layers/losses/iou_loss.py:7-50,391-414). */
typedef struct plyolo_yolox_desc {
  int B, A, C;          /* batch, total anchors, classes */
  int M;                /* label rows per image */
  int nlevels;          /* <= 8 */
  int lvl_h[8], lvl_w[8], lvl_stride[8];
  int lvl_off[8];       /* first anchor of the level in the per-image anchor order */
  int lvl_row[8];       /* first row of the level's dense [B, h*w, 5+C] block (level-major raw) */
  int use_l1;           /* YOLOXLoss(use_l1=True), yolox_loss.py:128-135,157-158: + L1 of the raw box outputs against get_l1_type */
} plyolo_yolox_desc;
size_t plyolo_yolox_workspace(const plyolo_yolox_desc* d);
/* raw: fp32 head output (tx,ty,tw,th,obj,cls..), LEVEL-major: level l is the dense NHWC
 * head map [B, h_l, w_l, 5+C] at row lvl_row[l] (so the head convs write it directly
 * and the reference's permute at yolox_loss.py:210-213 disappears).
 * labels [B,M,5] fp32 rows (cls,cx,cy,w,h), zero padded.
 * Outputs (batch-major, anchor order of the reference): fg u8[B,A]; matched_gt i32[B,A]
 * (-1 bg); matched_iou f32[B,A];
 * losses f32[8] = {loss, loss_iou, loss_obj, loss_cls, num_fg, num_gt, proportion, loss_l1 (0 without use_l1)}. */
int plyolo_yolox_loss_fwd(const plyolo_yolox_desc* d, const float* raw, const float* labels, uint8_t* fg,
                          int32_t* matched_gt, float* matched_iou, float* losses, void* workspace,
                          size_t ws_bytes, void* stream);
/* d(sum_i gout[i]*losses[i])/d(raw), gout fp32[8] on the device (entries 0..3 and 7 are read) or NULL (= d loss).
 * Level-major rows like raw.  Exactly one of the two forms is written:
 *   draw_f32 [rows,5+C]                                  (parity mode), or
 *   d_regobj bf16 [rows,16] (5 used) + d_cls bf16 [rows,cls_ld]   (MFMA mode). */
int plyolo_yolox_loss_bwd(const plyolo_yolox_desc* d, const float* raw, const float* labels, const uint8_t* fg,
                          const int32_t* matched_gt, const float* matched_iou, const float* losses,
                          const float* gout, float* draw_f32, void* d_regobj, void* d_cls, int cls_ld,
                          void* stream);
/* eval branch (yolox_loss.py:25-36): out [B,A,5+C] batch-major = (x1,y1,x2,y2,sig(obj),sig(cls)) */
int plyolo_yolox_eval_decode(const plyolo_yolox_desc* d, const float* raw, float* out, void* stream);

/* YOLOv7 eval branch (models/losses/yolov7/yolov7_loss.py:50-78), one call per level:
 * raw_level fp32 [B,h,w,na*(5+C)] -> rows [lvl_off, lvl_off+na*h*w) of out [B,A_total,5+C]. */
int plyolo_yolov7_eval_decode(const float* raw_level, int B, int h, int w, int na, int nc, int stride,
                              const float* anchors_dev, float* out, int A_total, int lvl_off, void* stream);

/* YOLOv7 training loss (models/losses/yolov7/yolov7_loss.py:80-153 with build_targets :155-306,
 * find_3_positive :308-368, bbox_iou(CIoU) :376-410) on the level-major raw head output
 * (level l = dense fp32 NHWC [B, h_l, w_l, na*(5+C)] starting at row lvl_row[l]).
 * labels [B,M,5] rows (cls,cx,cy,w,h) in pixels, zero padded.  losses[4] = loss, box, obj, cls
 * (already weighted: 0.05 / 1 / 0.5*C/80).  loss_bwd: gout = DEVICE pointer to the 4 upstream gradients of
 * losses[0..3]; draw (same layout as raw) = sum_i gout[i] * d losses[i] / d raw.
 * Ties among equal costs/IoUs go to the lowest candidate index (the reference's torch.topk leaves
 * them unspecified); candidates are enumerated in the reference's order. */
typedef struct plyolo_yolov7_desc {
  int B, M, C, na, nlevels;              /* na = 3, nlevels = 3 */
  int lvl_h[3], lvl_w[3], lvl_stride[3], lvl_row[3];
  float anchors[3][3][2];                /* pixels, [level][anchor][w,h] */
  int cand_cap;                          /* per-image candidate capacity, >= 45*M */
} plyolo_yolov7_desc;
size_t plyolo_yolov7_workspace(const plyolo_yolov7_desc* d);
int plyolo_yolov7_loss_fwd(const plyolo_yolov7_desc* d, const float* raw, const float* labels, float* losses,
                           void* workspace, size_t ws_bytes, void* stream);
int plyolo_yolov7_loss_bwd(const plyolo_yolov7_desc* d, const float* raw, const float* labels, const float* gout, float* draw,
                           void* workspace, size_t ws_bytes, void* stream);
/* diagnostics/tests: counts[B] and entries[B][cand_cap][6] = (level, anchor, gj, gi, gt row, last) */
int plyolo_yolov7_matched(const plyolo_yolov7_desc* d, const void* workspace, int32_t* counts_dev, int32_t* entries_dev,
                          void* stream);

/* ----------------------------------------------------------- postprocess
 * Replaces postprocess() (models/evaluators/postprocess.py:7-48) including
 * torchvision.ops.batched_nms / nms (un-vendored third party; algorithm per
 * torchvision/ops/boxes.py: coordinate trick when 4*n <= 20000 else per class). */
typedef struct plyolo_nms_desc {
  int B, A, C;
  float conf_thre, nms_thre;
  int class_agnostic;
  int max_nms;   /* 10000 */
  int max_det;   /* 300 */
  int numel_threshold; /* 20000 (torchvision GPU rule) */
} plyolo_nms_desc;
size_t plyolo_postprocess_workspace(const plyolo_nms_desc* d);
/* pred [B,A,5+C] fp32 (eval decode output) -> det f32 [B,max_det,6], count i32[B],
 * ncand i32[B] (boxes that entered NMS) */
int plyolo_postprocess(const plyolo_nms_desc* d, const float* pred, float* det, int32_t* count, int32_t* ncand,
                       void* workspace, size_t ws_bytes, void* stream);
/* Evaluation formatting (models/evaluators/postprocess.py:95-138 + models/utils/bbox.py:58-63), device side: for every
 * image descriptor the n detection rows (x1,y1,x2,y2,score,cls, row pitch ld floats) are divided by `scale` IN PLACE (the
 * reference's side effect) and written as packed (x1, y1, x2, y2, w, h, score, cls) rows out[(row0 + r) * 8] -- one launch and one
 * device->host copy per validation batch instead of one blocking copy per box.  imgs_dev: device array of B descriptors. */
typedef struct plyolo_fmt_image {
  float* det;
  int n, ld, row0;
  float scale;
} plyolo_fmt_image;
int plyolo_format_detections(const plyolo_fmt_image* imgs_dev, int B, int max_rows, float* out, void* stream);
/* NMS only: boxes [B,n,6] (x1,y1,x2,y2,score,cls) already filtered, n per image in nbox[B] */
int plyolo_batched_nms(const plyolo_nms_desc* d, const float* boxes, int n_max, const int32_t* nbox, float* det,
                       int32_t* count, void* workspace, size_t ws_bytes, void* stream);

/* -------------------------------------------------- optimizer step (a25)
 * SGD(momentum, no wd, no nesterov) + EMA over flat fp32 buffers
 * (PL_Modules/pl_detection.py:58-64,107-111; models/utils/ema.py:48-60). */
int plyolo_sgd_momentum(float* p, const float* g, float* mom, size_t n, const float* lr_dev, float lr,
                        float momentum, int first_step, void* stream);
int plyolo_ema_update(float* ema, const float* model, size_t n, float decay, void* stream);

#ifdef __cplusplus
}
#endif
#endif
