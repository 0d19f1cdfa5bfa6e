// YOLOX loss side on gfx950: decode, SimOTA label assignment, IoU/obj/cls losses
// (forward and analytic backward) and the eval decode.  fp32 throughout.
//
// Restates /root/reference semantics (file:line into Iywie/pl_YOLO):
//   decode                      models/losses/yolox/yolox_loss.py:175-228
//   get_in_boxes_info           models/losses/yolox/yolox_loss.py:231-315
//   pair-wise cost              models/losses/yolox/yolox_loss.py:84-108
//   dynamic_k_matching          models/losses/yolox/yolox_loss.py:318-370
//   targets + losses            models/losses/yolox/yolox_loss.py:120-173
//   bboxes_iou / IOUloss(giou)  models/layers/losses/iou_loss.py:391-414 / :7-50
//
// Structure (no [G, N_c, C] tensors are ever materialised, no host syncs):
//   k_prep    one workgroup per 128 anchors: decode, candidate mask; the candidates' class-independent part of the BCE
//             cost  S_a = sum_c -log(1-p_ac)  and their pair costs / IoUs against every GT, dealt out over all threads
//   k_topk    one workgroup per (image, GT): streams the candidates once, keeps the
//             10 largest IoUs and the 10 cheapest (cost, anchor) pairs per thread,
//             merges them with wavefront reductions, derives dynamic k and votes
//   k_loss    one thread per anchor: vote resolution (0 votes -> background, 1 vote -> that GT, >1 votes -> argmin over
//             ALL GTs of the pair cost, lowest GT wins ties), then the per-anchor loss terms and block partials;
//   k_final   fixed-order sum
//   k_bwd     d(loss)/d(raw) for every (anchor, channel), coalesced
// Tie rule: equal costs are ordered by lowest anchor index (the reference's
// torch.sort is unstable there).
#include "common.h"

namespace {

struct LossWs {
  float* dec;      // [B,A,4]
  uint8_t* cand;   // [B,A]
  int* cnt;        // [B,A]
  int* lastg;      // [B,A]
  int* G;          // [B]
  float* partial;  // [nblk,NPART]
  float* costm;    // [B,M,A] pair cost of (GT g, anchor a), written for candidate anchors only
  float* ioum;     // [B,M,A] pair IoU
};

constexpr int MAXM = 256;  // label rows per image supported by the LDS staging
constexpr int NPART = 5;   // block partials of k_loss: iou, obj, cls, num_fg, l1

DEVINL float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// Hardware-rate versions for the SimOTA cost only (v_exp / v_log / v_rcp / v_sqrt: ~1-2 ulp).  The cost
// feeds discrete choices, never the loss value; its libm-precise form made k_prep VALU-bound (80 classes x
// exp, div, sqrt, log1p per candidate anchor).
DEVINL float sig_fast(float x) { return __frcp_rn(1.0f + __expf(-x)); }

DEVINL int anchor_level(const plyolo_yolox_desc& d, int a) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < d.nlevels && a >= d.lvl_off[i]) l = i;
  return l;
}
// The head output is LEVEL-major: level l is a dense NHWC block [B, h_l*w_l, 5+C] that starts
// at row lvl_row[l]; row of (image b, anchor a):
DEVINL size_t raw_row(const plyolo_yolox_desc& d, int b, int a) {
  const int l = anchor_level(d, a);
  return (size_t)d.lvl_row[l] + (size_t)b * (d.lvl_h[l] * d.lvl_w[l]) + (a - d.lvl_off[l]);
}

DEVINL void anchor_geom(const plyolo_yolox_desc& d, int a, float* xs, float* ys, float* st) {
  const int l = anchor_level(d, a);
  const int f = a - d.lvl_off[l];
  // grid quirk kept verbatim (yolox_loss.py:198-200): (f % h, f / h); equals (f % w, f / w) for square maps
  *xs = (float)(f % d.lvl_h[l]);
  *ys = (float)(f / d.lvl_h[l]);
  *st = (float)d.lvl_stride[l];
}

DEVINL float pair_iou(const float* gt, const float* pb) {
  // bboxes_iou(xyxy=False): `en = (tl < br)` gate, no eps
  const float tlx = fmaxf(gt[0] - gt[2] / 2, pb[0] - pb[2] / 2), tly = fmaxf(gt[1] - gt[3] / 2, pb[1] - pb[3] / 2);
  const float brx = fminf(gt[0] + gt[2] / 2, pb[0] + pb[2] / 2), bry = fminf(gt[1] + gt[3] / 2, pb[1] + pb[3] / 2);
  const float area_a = gt[2] * gt[3], area_b = pb[2] * pb[3];
  const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
  const float area_i = (brx - tlx) * (bry - tly) * en;
  return area_i / (area_a + area_b - area_i);
}

DEVINL void in_masks(const float* gt, float xc, float yc, float st, bool* in_box, bool* in_ctr) {
  const float l = gt[0] - 0.5f * gt[2], r = gt[0] + 0.5f * gt[2];
  const float t = gt[1] - 0.5f * gt[3], b = gt[1] + 0.5f * gt[3];
  *in_box = fminf(fminf(xc - l, yc - t), fminf(r - xc, b - yc)) > 0.0f;
  const float rad = 2.5f * st;
  const float cl = gt[0] - rad, cr = gt[0] + rad, ct = gt[1] - rad, cb = gt[1] + rad;
  *in_ctr = fminf(fminf(xc - cl, yc - ct), fminf(cr - xc, cb - yc)) > 0.0f;
}

// cost[g,a] and iou[g,a] -- the ONE definition used by both k_topk and k_resolve
DEVINL float pair_cost(const float* delta_a, const float* dec_a, float S_a, const float* lab_g, float xc, float yc, float st,
                       float* iou_out) {
  const float iou = pair_iou(lab_g + 1, dec_a);
  *iou_out = iou;
  const float iou_cost = -__logf(iou + 1e-8f);
  // F.binary_cross_entropy(sqrt(sig(cls)*sig(obj)), onehot).sum(C): S_a holds the t=0 terms of all classes,
  // delta_a[c] = (t=1 term) - (t=0 term) of class c (both clamped at 100 like torch)
  const float cls_cost = S_a + delta_a[(int)lab_g[0]];
  bool ib, ic;
  in_masks(lab_g + 1, xc, yc, st, &ib, &ic);
  return (cls_cost + 3.0f * iou_cost) + 100000.0f * ((ib && ic) ? 0.0f : 1.0f);
}

// One workgroup = up to 128 consecutive anchors of ONE level of one image.  Their raw rows are contiguous
// in the level-major head output, so they are staged into LDS with coalesced dword loads (a thread walking
// its own 340-byte row straight from global costs a cache line per lane and load: x16 L2 traffic); each
// thread then owns one anchor: decode, candidate test, the class-independent cost term S, and -- for
// candidates -- the pair cost / IoU against every GT of the image, written to the [G, A] matrices that
// k_topk and k_resolve read (one definition of the cost, computed once).
constexpr int PREP_A = 128;   // anchors per workgroup
constexpr int PREP_T = 256;   // threads: TWO per anchor (round 3) -- the halves split the classes of the BCE term and the GTs of the
                              // pair-cost loop, so a candidate anchor's serial chain is half as long and a CU holds twice the waves
                              // for the same LDS (the 43 KB row tile limits a CU to three workgroups): 95 -> ~60 us at B=32
__global__ __launch_bounds__(PREP_T) void k_prep(const plyolo_yolox_desc d, const float* raw, const float* labels, LossWs ws) {
  extern __shared__ __align__(16) float prep_smem[];
  const int nch = 5 + d.C;
  float* rows = prep_smem;                 // [PREP_A][nch]
  float* lab = prep_smem + PREP_A * nch;   // [M][5]
  float* spart = lab + d.M * 5;            // [PREP_A][4] partial class sums of the candidates (four class quarters)
  float* sgeo = spart + 4 * PREP_A;        // [PREP_A][8] candidates: decoded box, anchor centre
  int* clist = (int*)(sgeo + 8 * PREP_A);  // [PREP_A] candidate anchors of this tile
  int* smask = clist + PREP_A;             // [PREP_T] per-thread "inside some GT box / centre region" of its half of the GTs
  __shared__ int sG, s_nc;
  const int b = blockIdx.y, tid = threadIdx.x;
  // level and chunk of this workgroup
  int l = 0, chunk = blockIdx.x;
  for (; l < d.nlevels; ++l) {
    const int nc = (d.lvl_h[l] * d.lvl_w[l] + PREP_A - 1) / PREP_A;
    if (chunk < nc) break;
    chunk -= nc;
  }
  if (l >= d.nlevels) return;
  const int hw = d.lvl_h[l] * d.lvl_w[l];
  const int f0 = chunk * PREP_A, n = min(PREP_A, hw - f0);
  const float* src = raw + ((size_t)d.lvl_row[l] + (size_t)b * hw + f0) * nch;
  // the image's label rows (M <= MAXM = 256: at most 5 values per thread) are requested first and ride the same round trip as the tile
  float lv[(MAXM * 5 + PREP_T - 1) / PREP_T];
#pragma unroll
  for (int k = 0; k < (MAXM * 5 + PREP_T - 1) / PREP_T; ++k) {
    const int i = tid + k * PREP_T;
    lv[k] = i < d.M * 5 ? labels[(size_t)b * d.M * 5 + i] : 0.f;
  }
  {
    // batched copy: BATCH loads in flight per thread before the first LDS store (a load -> store loop
    // serialises the round trips), 16-byte vectors when the block of rows is 16-byte aligned
    constexpr int BATCH = 12;   // a full 128 x 85 tile is 10.6 vectors per thread: ONE round trip (4 -> three of them per workgroup)
    const int total = n * nch;
    if ((((size_t)src) & 15) == 0) {
      const int nv = total >> 2;
      const f32x4* s4 = (const f32x4*)src;
      f32x4* r4 = (f32x4*)rows;
      for (int i0 = 0; i0 < nv; i0 += BATCH * PREP_T) {
        f32x4 v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int i = i0 + tid + k * PREP_T;
          v[k] = i < nv ? s4[i] : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int i = i0 + tid + k * PREP_T;
          if (i < nv) r4[i] = v[k];
        }
      }
      for (int i = (nv << 2) + tid; i < total; i += PREP_T) rows[i] = src[i];
    } else {
      for (int i0 = 0; i0 < total; i0 += BATCH * PREP_T) {
        float v[BATCH];
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int i = i0 + tid + k * PREP_T;
          v[k] = i < total ? src[i] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < BATCH; ++k) {
          const int i = i0 + tid + k * PREP_T;
          if (i < total) rows[i] = v[k];
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < (MAXM * 5 + PREP_T - 1) / PREP_T; ++k) {
    const int i = tid + k * PREP_T;
    if (i < d.M * 5) lab[i] = lv[k];
  }
  if (tid == 0) sG = 0;
  __syncthreads();
  {
    int local = 0;
    for (int g = tid; g < d.M; g += PREP_T) {
      const float s = lab[g * 5] + lab[g * 5 + 1] + lab[g * 5 + 2] + lab[g * 5 + 3] + lab[g * 5 + 4];
      local += s > 0.f ? 1 : 0;
    }
    if (local) atomicAdd(&sG, local);
  }
  __syncthreads();
  const int G = sG;
  if (blockIdx.x == 0 && tid == 0) ws.G[b] = G;
  const int la = tid & (PREP_A - 1), half = tid / PREP_A;   // anchor of this thread inside the tile, which half of the work
  const bool live = la < n;
  const int fa = f0 + (live ? la : 0), a = d.lvl_off[l] + fa;
  const size_t ba = (size_t)b * d.A + a;
  const float* r = rows + (live ? la : 0) * nch;
  // one level per workgroup: the stride is uniform; grid quirk kept verbatim (yolox_loss.py:198-200), see anchor_geom
  const float st = (float)d.lvl_stride[l];
  const float xs = (float)(fa % d.lvl_h[l]), ys = (float)(fa / d.lvl_h[l]);
  const float xc = xs * st + 0.5f * st, yc = ys * st + 0.5f * st;
  // candidate test: the two threads of an anchor split the GTs
  bool any = false;
  for (int g = half; g < G; g += 2) {
    bool ib, ic;
    in_masks(lab + g * 5 + 1, xc, yc, st, &ib, &ic);
    any |= ib | ic;
  }
  smask[tid] = any ? 1 : 0;
  if (tid == 0) s_nc = 0;
  __syncthreads();
  const bool cand = live && (smask[la] | smask[PREP_A + la]) != 0;
  if (live && half == 0) {
    const float d0 = (r[0] + xs) * st, d1 = (r[1] + ys) * st, d2 = expf(r[2]) * st, d3 = expf(r[3]) * st;
    *(f32x4*)(ws.dec + ba * 4) = f32x4{d0, d1, d2, d3};
    ws.cand[ba] = cand ? 1 : 0;
    ws.cnt[ba] = 0;      // votes of k_topk (every anchor belongs to exactly one workgroup here: no separate fill launch)
    if (cand) {
      float* ge = sgeo + la * 8;
      ge[0] = d0; ge[1] = d1; ge[2] = d2; ge[3] = d3; ge[4] = xc; ge[5] = yc;
      clist[atomicAdd(&s_nc, 1)] = la;     // the ORDER of the list only decides which thread works on which candidate
    }
  }
  __syncthreads();
  const int nc = s_nc;
  // Candidates are a third of the anchors and come in clusters: with one (half-)thread per anchor most lanes of a wave sat idle
  // through 40 classes x 5 transcendentals and 15 pair costs.  The work is dealt out over the whole workgroup instead:
  //   items (candidate, quarter of the classes): the class-independent BCE term  S_a = sum_c -log(1-p_ac), in four partial sums;
  //   items (GT, candidate): pair cost and IoU.
  const int csz = (d.C + 3) / 4;
  for (int item = tid; item < nc * 4; item += PREP_T) {
    float* r2 = rows + clist[item >> 2] * nch;
    const int c_lo = (item & 3) * csz, c_hi = min(d.C, c_lo + csz);
    // p = sqrt(sig(cls) * sig(obj)) with sig(x) = 1 / (1 + e^-x):  p = rsqrt((1 + e^-cls) * (1 + e^-obj)) and
    // -log p = (log(1 + e^-cls) + log(1 + e^-obj)) / 2 -- four hardware-rate transcendentals per class (exp, log, rsq, log) and
    // no division / square-root refinement sequences; the obj factors are per anchor.  Like the previous form this is the
    // COST only (discrete choices, ~1e-6 relative), never the loss value.
    const float eo = 1.0f + __expf(-r2[4]);
    const float lo = __logf(eo);
    float Sp = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
      const float ec = 1.0f + __expf(-r2[5 + c]);
      const float p = __builtin_amdgcn_rsqf(ec * eo);
      const float t0 = -fmaxf(__logf(1.0f - p), -100.0f);
      const float t1 = fminf(0.5f * (__logf(ec) + lo), 100.0f);
      Sp += t0;
      r2[5 + c] = t1 - t0;   // the raw class logit of this LDS row is not needed again (every item owns its classes)
    }
    spart[item] = Sp;
  }
  __syncthreads();
  if (nc > 0) {
    // a thread keeps ONE candidate (box, centre, S, row in registers) and walks every nsplit-th GT
    const int nsplit = PREP_T / nc, ci = tid % nc, gs = tid / nc;
    if (gs < nsplit) {
      const int la2 = clist[ci];
      const float* ge = sgeo + la2 * 8;
      const float dec2[4] = {ge[0], ge[1], ge[2], ge[3]};
      const float xc2 = ge[4], yc2 = ge[5];
      const float S = ((spart[ci * 4] + spart[ci * 4 + 1]) + spart[ci * 4 + 2]) + spart[ci * 4 + 3];   // fixed order
      const float* delta = rows + la2 * nch + 5;
      float* crow = ws.costm + (size_t)b * d.M * d.A + (d.lvl_off[l] + f0 + la2);
      float* irow = ws.ioum + (size_t)b * d.M * d.A + (d.lvl_off[l] + f0 + la2);
      for (int g = gs; g < G; g += nsplit) {
        float iou;
        const float cost = pair_cost(delta, dec2, S, lab + g * 5, xc2, yc2, st, &iou);
        crow[(size_t)g * d.A] = cost;
        irow[(size_t)g * d.A] = iou;
      }
    }
  }
}

DEVINL unsigned orderable(float c) {
  const unsigned u = __float_as_uint(c);
  return (u & 0x80000000u) ? ~u : (u ^ 0x80000000u);  // monotone float -> uint
}
DEVINL float unorderable(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u); }
// ascending sort key: (cost, anchor) -- equal costs resolve to the lowest anchor index
DEVINL unsigned long long cost_key(float c, int idx) { return ((unsigned long long)orderable(c) << 32) | (unsigned)idx; }
// descending sort key for IoU values; the low word only makes keys unique
DEVINL unsigned long long iou_key(float v, int idx) { return ((unsigned long long)orderable(v) << 32) | (unsigned)(~idx); }

DEVINL unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(v, o);
    v = other < v ? other : v;
  }
  return v;
}
DEVINL unsigned long long wave_max_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(v, o);
    v = other > v ? other : v;
  }
  return v;
}

__global__ __launch_bounds__(256) void k_topk(const plyolo_yolox_desc d, const float* raw, const float* labels, LossWs ws) {
  const int b = blockIdx.y, g = blockIdx.x;
  if (g >= ws.G[b]) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  __shared__ unsigned long long red_k[2][4], red_i[2][4];
  __shared__ int red_n[4], red_b[4];
  __shared__ int sel[10];
  __shared__ int s_k, s_nc, s_nb;

  // Per-thread sorted lists: the 10 largest IoUs and the 10 cheapest (cost, anchor) pairs of this thread's candidates.
  // Exact pruning (round 3; the insertion sort over EVERY candidate made this kernel VALU-bound, 74 us):
  //   * an IoU of exactly 0 never changes the sum of the top-10 IoUs -> it is not inserted;
  //   * a candidate outside (in_box AND in_center) of this GT costs >= 1e5, one inside < 1e5 (80 clamped BCE terms + 3 * 18.5 at
  //     most): while k <= the number of inside candidates the k cheapest are all inside -> only those are inserted in the first
  //     scan, and the (rare: a GT with fewer than k in-centre anchors) other case rescans with every candidate.
  unsigned long long top_ik[10], top_key[10];
  const float* crow = ws.costm + ((size_t)b * d.M + g) * d.A;
  const float* irow = ws.ioum + ((size_t)b * d.M + g) * d.A;
  const uint8_t* cnd = ws.cand + (size_t)b * d.A;
  int nc = 0, nboth = 0;
#pragma unroll
  for (int i = 0; i < 10; ++i) { top_ik[i] = 0ull; top_key[i] = ~0ull; }
  // The scan is a chain of dependent round trips when written "flag, then (if set) the two values": 33 anchors per thread x two
  // latencies = the whole 50 us of this kernel.  The flag and both values of TKB anchors are requested together instead (the
  // [G, A] rows hold stale bytes for non-candidates -- allocated, never used: the flag decides).
  constexpr int TKB = 4;
  for (int a0 = tid; a0 < d.A; a0 += 256 * TKB) {
    uint8_t cf[TKB];
    float iv[TKB], cv[TKB];
#pragma unroll
    for (int j = 0; j < TKB; ++j) {
      const int a = a0 + j * 256, ac = a < d.A ? a : d.A - 1;
      cf[j] = a < d.A ? cnd[ac] : (uint8_t)0;
      iv[j] = irow[ac];
      cv[j] = crow[ac];
    }
#pragma unroll
    for (int j = 0; j < TKB; ++j) {
      if (!cf[j]) continue;
      const int a = a0 + j * 256;
      ++nc;
      const float iou = iv[j], cost = cv[j];   // computed once by k_prep
      if (iou > 0.f) {
        unsigned long long x = iou_key(iou, a);
#pragma unroll
        for (int i = 0; i < 10; ++i)
          if (x > top_ik[i]) { const unsigned long long t = top_ik[i]; top_ik[i] = x; x = t; }
      }
      if (cost < 100000.0f) {
        ++nboth;
        unsigned long long k = cost_key(cost, a);
#pragma unroll
        for (int i = 0; i < 10; ++i)
          if (k < top_key[i]) { const unsigned long long t = top_key[i]; top_key[i] = k; k = t; }
      }
    }
  }
  {
    int v = nc, w = nboth;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { v += __shfl_xor(v, o); w += __shfl_xor(w, o); }
    if (lane == 0) { red_n[wave] = v; red_b[wave] = w; }
    __syncthreads();
    if (tid == 0) { s_nc = red_n[0] + red_n[1] + red_n[2] + red_n[3]; s_nb = red_b[0] + red_b[1] + red_b[2] + red_b[3]; }
    __syncthreads();
  }
  const int NC = s_nc, NB = s_nb;
  const int n_k = NC < 10 ? NC : 10;
  // dynamic k: the n_k largest IoUs, summed in descending order (sequential fp32 sum, like the reference's sorted slice)
  float iou_sum = 0.f;
  for (int round = 0; round < n_k; ++round) {
    const int pb = round & 1;
    const unsigned long long hi = top_ik[0];
    unsigned long long mi = wave_max_u64(hi);
    if (lane == 0) red_i[pb][wave] = mi;
    __syncthreads();
    mi = red_i[pb][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) mi = red_i[pb][w] > mi ? red_i[pb][w] : mi;
    if (mi == 0ull) break;      // only zeros left (block-uniform): they add nothing
    iou_sum += unorderable((unsigned)(mi >> 32));
    if (hi == mi) {
#pragma unroll
      for (int i = 0; i < 9; ++i) top_ik[i] = top_ik[i + 1];
      top_ik[9] = 0ull;
    }
  }
  int k = (int)iou_sum;  // .int() truncation (yolox_loss.py:340)
  if (k < 1) k = 1;
  if (k >= NC - 1) {
    // yolox_loss.py:343-344: k >= N_c - 1  ->  every candidate is taken
    for (int a = tid; a < d.A; a += 256) {
      if (cnd[a]) {
        const size_t ba = (size_t)b * d.A + a;
        atomicAdd(&ws.cnt[ba], 1);
        ws.lastg[ba] = g;
      }
    }
    return;
  }
  if (k > NB) {   // fewer in-centre candidates than k: the k cheapest reach into the >= 1e5 costs -> full scan (block-uniform branch)
#pragma unroll
    for (int i = 0; i < 10; ++i) top_key[i] = ~0ull;
    for (int a = tid; a < d.A; a += 256) {
      if (!cnd[a]) continue;
      unsigned long long kk = cost_key(crow[a], a);
#pragma unroll
      for (int i = 0; i < 10; ++i)
        if (kk < top_key[i]) { const unsigned long long t = top_key[i]; top_key[i] = kk; kk = t; }
    }
  }
  __syncthreads();   // red_* reuse
  // the k cheapest (cost, anchor) pairs: each round the (unique) owner of the block-wide best key pops it
  for (int round = 0; round < k; ++round) {
    const int pb = round & 1;
    const unsigned long long hk = top_key[0];
    unsigned long long mk = wave_min_u64(hk);
    if (lane == 0) red_k[pb][wave] = mk;
    __syncthreads();
    mk = red_k[pb][0];
#pragma unroll
    for (int w = 1; w < 4; ++w) mk = red_k[pb][w] < mk ? red_k[pb][w] : mk;
    if (tid == 0) sel[round] = (int)(unsigned)(mk & 0xffffffffull);
    if (hk == mk && mk != ~0ull) {
#pragma unroll
      for (int i = 0; i < 9; ++i) top_key[i] = top_key[i + 1];
      top_key[9] = ~0ull;
    }
  }
  __syncthreads();
  if (tid < k) {
    const int a = sel[tid];
    const size_t ba = (size_t)b * d.A + a;
    atomicAdd(&ws.cnt[ba], 1);
    ws.lastg[ba] = g;
  }
}

// IOUloss(loss_type="giou") of iou_loss.py:13-43 (note the (area_c - area_i)/area_c penalty) and its gradient
DEVINL float giou_loss(const float* p, const float* t, float* grad /* d loss / d(cx,cy,w,h) or null */) {
  const float plx = p[0] - p[2] / 2, ply = p[1] - p[3] / 2, phx = p[0] + p[2] / 2, phy = p[1] + p[3] / 2;
  const float glx = t[0] - t[2] / 2, gly = t[1] - t[3] / 2, ghx = t[0] + t[2] / 2, ghy = t[1] + t[3] / 2;
  const float tlx = fmaxf(plx, glx), tly = fmaxf(ply, gly), brx = fminf(phx, ghx), bry = fminf(phy, ghy);
  const float area_p = p[2] * p[3], area_g = t[2] * t[3];
  const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
  const float iw = brx - tlx, ih = bry - tly;
  const float I = iw * ih * en;
  const float U = area_p + area_g - I + 1e-16f;
  const float iou = I / U;
  const float ctlx = fminf(plx, glx), ctly = fminf(ply, gly), cbrx = fmaxf(phx, ghx), cbry = fmaxf(phy, ghy);
  const float cw = cbrx - ctlx, ch = cbry - ctly;
  const float Ac = cw * ch;
  const float ac = fmaxf(Ac, 1e-16f);
  const float giou = iou - (Ac - I) / ac;
  const float loss = 1.0f - fminf(fmaxf(giou, -1.0f), 1.0f);
  if (grad) {
    const float dL = (giou >= -1.0f && giou <= 1.0f) ? -1.0f : 0.0f;
    const float dI = dL * ((U + I) / (U * U) + 1.0f / ac);
    const float dAp = dL * (-I / (U * U));
    const float dAc = dL * (-(1.0f / ac) + ((Ac >= 1e-16f) ? (Ac - I) / (ac * ac) : 0.0f));
    // maximum/minimum backward: the larger (smaller) argument takes the gradient, ties split 1/2
    auto wmax = [](float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); };
    auto wmin = [](float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); };
    const float g_phx = dI * ih * en * wmin(phx, ghx) + dAc * ch * wmax(phx, ghx);
    const float g_plx = -dI * ih * en * wmax(plx, glx) - dAc * ch * wmin(plx, glx);
    const float g_phy = dI * iw * en * wmin(phy, ghy) + dAc * cw * wmax(phy, ghy);
    const float g_ply = -dI * iw * en * wmax(ply, gly) - dAc * cw * wmin(ply, gly);
    grad[0] = g_phx + g_plx;
    grad[1] = g_phy + g_ply;
    grad[2] = 0.5f * (g_phx - g_plx) + dAp * p[3];
    grad[3] = 0.5f * (g_phy - g_ply) + dAp * p[2];
  }
  return loss;
}

DEVINL float bce_logits(float x, float t) {
  // (1-t)*x - log_sigmoid(x),  log_sigmoid(x) = min(x,0) - log1p(exp(-|x|))
  return (1.0f - t) * x - (fminf(x, 0.0f) - log1pf(expf(-fabsf(x))));
}

// get_l1_type (yolox_loss.py:373-378): the regression target in the RAW output space of the matched anchor
DEVINL void l1_target(const float* gt_box, float xs, float ys, float st, float* t) {
  t[0] = gt_box[0] / st - xs;
  t[1] = gt_box[1] / st - ys;
  t[2] = logf(gt_box[2] / st + 1e-8f);
  t[3] = logf(gt_box[3] / st + 1e-8f);
}

// One thread per anchor.  First the vote resolution (0 votes -> background, 1 vote -> that GT, > 1 votes -> argmin over ALL GTs of
// the pair cost, lowest GT wins ties), then the anchor's loss terms and the block partials.  (Round 3: resolution and loss were
// two launches; the class term walked the foreground lanes of a wave one after the other through THREE dependent round trips each
// -- matched GT -> label row -> logits; now every foreground lane fetches its own metadata up front and the rows of two
// foreground anchors are in flight per step.)
__global__ __launch_bounds__(256) void k_loss(const plyolo_yolox_desc d, const float* raw, const float* labels, LossWs ws,
                                              uint8_t* fg, int32_t* mgt, float* miou) {
  const size_t total = (size_t)d.B * d.A;
  const int nch = 5 + d.C;
  float s_iou = 0.f, s_obj = 0.f, s_cls = 0.f, s_fg = 0.f, s_l1 = 0.f;
  const size_t ba = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  bool f = false;
  int cg = 0;          // class of the matched GT (foreground lanes)
  float io = 0.f;      // matched IoU
  unsigned rrow = 0u;  // level-major row of this anchor (B*A*(5+C) < 2^32 is checked by the backward; rows fit 32 bits)
  if (ba < total) {
    const int b = (int)(ba / d.A), a = (int)(ba - (size_t)b * d.A);
    rrow = (unsigned)raw_row(d, b, a);
    const float* r = raw + (size_t)rrow * nch;
    const float obj_logit = r[4];
    const int c = ws.cnt[ba];
    int g = -1;
    if (c != 0) {
      g = ws.lastg[ba];
      if (c > 1) {   // several GTs voted for this anchor: it goes to the cheapest one (first minimum)
        const int G = ws.G[b];
        float best = INFINITY;
        g = 0;
        for (int gg = 0; gg < G; ++gg) {
          const float cost = ws.costm[((size_t)b * d.M + gg) * d.A + a];
          if (cost < best) { best = cost; g = gg; }
        }
      }
      io = ws.ioum[((size_t)b * d.M + g) * d.A + a];
      f = true;
    }
    fg[ba] = f ? 1 : 0;
    mgt[ba] = g;
    miou[ba] = io;
    s_obj = bce_logits(obj_logit, f ? 1.0f : 0.0f);
    if (f) {
      const float* lg = labels + ((size_t)b * d.M + g) * 5;
      cg = (int)lg[0];
      s_fg = 1.f;
      s_iou = giou_loss(ws.dec + ba * 4, lg + 1, nullptr);
      if (d.use_l1) {   // nn.L1Loss(reduction="none") of the raw box outputs against get_l1_type (yolox_loss.py:157-158)
        float xs, ys, st, t[4];
        anchor_geom(d, a, &xs, &ys, &st);
        l1_target(lg + 1, xs, ys, st, t);
        s_l1 = fabsf(r[0] - t[0]) + fabsf(r[1] - t[1]) + fabsf(r[2] - t[2]) + fabsf(r[3] - t[3]);
      }
    }
  }
  {
    // class term of the (rare) foreground anchors: the whole wave walks the 80 logits of each foreground
    // lane together (coalesced row read, 2 steps) instead of one lane looping 80 times while 63 idle
    const int lane = threadIdx.x & 63;
    unsigned long long m = __ballot(f);
    while (m) {
      const int s0 = __ffsll((long long)m) - 1;
      m &= m - 1;
      const bool two = m != 0ull;
      const int s1 = two ? __ffsll((long long)m) - 1 : s0;
      if (two) m &= m - 1;
      const float* fr0 = raw + (size_t)__shfl(rrow, s0) * nch;
      const float* fr1 = raw + (size_t)__shfl(rrow, s1) * nch;
      const int cg0 = __shfl(cg, s0), cg1 = __shfl(cg, s1);
      const float io0 = __shfl(io, s0), io1 = __shfl(io, s1);
      float part = 0.f;
      for (int c = lane; c < d.C; c += 64) {
        const float x0 = fr0[5 + c], x1 = fr1[5 + c];      // both rows requested before either is used
        part += bce_logits(x0, c == cg0 ? io0 : 0.0f);
        if (two) part += bce_logits(x1, c == cg1 ? io1 : 0.0f);
      }
      s_cls += part;   // every lane carries a share; the block reduction below sums them all
    }
  }
  __shared__ float red[4][NPART];
  float v[NPART] = {s_iou, s_obj, s_cls, s_fg, s_l1};
#pragma unroll
  for (int i = 0; i < NPART; ++i) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[i] += __shfl_xor(v[i], o);
  }
  if ((threadIdx.x & 63) == 0)
    for (int i = 0; i < NPART; ++i) red[threadIdx.x >> 6][i] = v[i];
  __syncthreads();
  if (threadIdx.x < NPART) ws.partial[(size_t)blockIdx.x * NPART + threadIdx.x] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

__global__ void k_final(int nblk, int B, LossWs ws, float* losses) {
  __shared__ double red[256][NPART];
  double s[NPART] = {0, 0, 0, 0, 0};
  for (int i = threadIdx.x; i < nblk; i += 256)
    for (int j = 0; j < NPART; ++j) s[j] += ws.partial[(size_t)i * NPART + j];
  for (int j = 0; j < NPART; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o)
      for (int j = 0; j < NPART; ++j) red[threadIdx.x][j] += red[threadIdx.x + o][j];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    int ngt = 0;
    for (int b = 0; b < B; ++b) ngt += ws.G[b];
    const double nfg = red[0][3];
    const double N = nfg > 1.0 ? nfg : 1.0;
    const float li = (float)(red[0][0] / N), lo = (float)(red[0][1] / N), lc = (float)(red[0][2] / N);
    const float l1 = (float)(red[0][4] / N);   // 0 without use_l1 (yolox_loss.py:157-160)
    losses[0] = 5.0f * li + lo + lc + l1;
    losses[1] = li;
    losses[2] = lo;
    losses[3] = lc;
    losses[4] = (float)nfg;
    losses[5] = (float)ngt;
    losses[6] = (float)(N / (ngt > 1 ? (double)ngt : 1.0));  // proportion (yolox_loss.py:171)
    losses[7] = l1;
  }
}

// d(sum_i gout[i]*losses[i]) / d(raw): losses[0] = 5*iou + obj + cls (+ l1), so the terms carry
// w_iou = 5*g0+g1, w_obj = g0+g2, w_cls = g0+g3, w_l1 = g0+g7.  One thread per (level-major row, channel).
template <bool BF16OUT>
__global__ void k_bwd(const plyolo_yolox_desc d, const float* raw, const float* labels, const uint8_t* fg, const int32_t* mgt,
                      const float* miou, const float* losses, const float* gout, float* draw, bf16_t* d_regobj, bf16_t* d_cls,
                      int cls_ld) {
  const int nch = 5 + d.C;
  const size_t total = (size_t)d.B * d.A * nch;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  // 32-bit index arithmetic (the host checks total < 2^32): a 64-bit division costs ~100 instructions per element
  const unsigned row = (unsigned)idx / (unsigned)nch;
  const int c = (int)((unsigned)idx - row * (unsigned)nch);
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < d.nlevels && row >= (unsigned)d.lvl_row[i]) l = i;
  const int hw = d.lvl_h[l] * d.lvl_w[l];
  const int rr = (int)(row - (unsigned)d.lvl_row[l]);
  const int b = (int)((unsigned)rr / (unsigned)hw);
  const int a = d.lvl_off[l] + (rr - b * hw);
  const size_t ba = (size_t)b * d.A + a;
  const float g0 = gout ? gout[0] : 1.0f;
  const float w_iou = gout ? 5.0f * g0 + gout[1] : 5.0f, w_obj = gout ? g0 + gout[2] : 1.0f, w_cls = gout ? g0 + gout[3] : 1.0f;
  const float nfg = losses[4];
  const float invN = 1.0f / (nfg > 1.0f ? nfg : 1.0f);
  const float* r = raw + (size_t)row * nch;
  const bool f = fg[ba] != 0;
  float g = 0.f;
  if (c == 4) {
    g = (sig_fast(r[4]) - (f ? 1.0f : 0.0f)) * invN * w_obj;
  } else if (f) {
    const float* lg = labels + ((size_t)b * d.M + mgt[ba]) * 5;
    if (c >= 5) {
      const float t = (c - 5 == (int)lg[0]) ? miou[ba] : 0.0f;
      g = (sig_fast(r[c]) - t) * invN * w_cls;
    } else {
      float xs, ys, st;
      anchor_geom(d, a, &xs, &ys, &st);
      float p[4] = {(r[0] + xs) * st, (r[1] + ys) * st, expf(r[2]) * st, expf(r[3]) * st};
      float gr[4];
      giou_loss(p, lg + 1, gr);
      const float chain = (c < 2) ? st : p[c];  // d cx/d tx = s ; d w/d tw = w
      g = w_iou * gr[c] * chain * invN;
      if (d.use_l1) {   // d|r - t| = sign(r - t), 0 at equality (torch's l1_loss backward)
        float t[4];
        l1_target(lg + 1, xs, ys, st, t);
        const float df = r[c] - t[c];
        g += (gout ? g0 + gout[7] : 1.0f) * ((df > 0.f) ? 1.f : (df < 0.f ? -1.f : 0.f)) * invN;
      }
    }
  }
  if (BF16OUT) {
    if (c < 5) d_regobj[(size_t)row * 16 + c] = f2bf(g);
    else d_cls[(size_t)row * cls_ld + (c - 5)] = f2bf(g);
  } else {
    draw[idx] = g;
  }
}

// bf16 gradient matrices, one 16-byte vector per thread: vector 0 of a row = (d tx, d ty, d tw, d th, d obj, 0, 0, 0)
// of the [rows,16] reg+obj matrix, vectors 1.. = 8 class channels of the [rows,cls_ld] matrix.  Same arithmetic as
// k_bwd<true> (which stores one bf16 per thread); the row bookkeeping is done once per 8 channels.
__global__ __launch_bounds__(256) void k_bwd_vec(const plyolo_yolox_desc d, const float* __restrict__ raw, const float* labels,
                                                 const uint8_t* fg, const int32_t* mgt, const float* miou, const float* losses,
                                                 const float* gout, bf16_t* d_regobj, bf16_t* d_cls, int cls_ld) {
  const int nch = 5 + d.C;
  const unsigned vpr = 1u + (unsigned)cls_ld / 8u;
  const unsigned total = (unsigned)d.B * (unsigned)d.A * vpr;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const unsigned row = idx / vpr;
  const int v = (int)(idx - row * vpr);
  int l = 0;
#pragma unroll
  for (int i = 1; i < 8; ++i)
    if (i < d.nlevels && row >= (unsigned)d.lvl_row[i]) l = i;
  const int hw = d.lvl_h[l] * d.lvl_w[l];
  const int rr = (int)(row - (unsigned)d.lvl_row[l]);
  const int b = (int)((unsigned)rr / (unsigned)hw);
  const int a = d.lvl_off[l] + (rr - b * hw);
  const size_t ba = (size_t)b * d.A + a;
  const float g0 = gout ? gout[0] : 1.0f;
  const float w_iou = gout ? 5.0f * g0 + gout[1] : 5.0f, w_obj = gout ? g0 + gout[2] : 1.0f, w_cls = gout ? g0 + gout[3] : 1.0f;
  const float nfg = losses[4];
  const float invN = 1.0f / (nfg > 1.0f ? nfg : 1.0f);
  const float* r = raw + (size_t)row * nch;
  const bool f = fg[ba] != 0;
  float g[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = 0.f;
  if (v == 0) {
    g[4] = (sig_fast(r[4]) - (f ? 1.0f : 0.0f)) * invN * w_obj;
    if (f) {
      const float* lg = labels + ((size_t)b * d.M + mgt[ba]) * 5;
      float xs, ys, st;
      anchor_geom(d, a, &xs, &ys, &st);
      float p[4] = {(r[0] + xs) * st, (r[1] + ys) * st, expf(r[2]) * st, expf(r[3]) * st};
      float gr[4];
      giou_loss(p, lg + 1, gr);
#pragma unroll
      for (int c = 0; c < 4; ++c) g[c] = w_iou * gr[c] * ((c < 2) ? st : p[c]) * invN;
      if (d.use_l1) {
        float t[4];
        l1_target(lg + 1, xs, ys, st, t);
        const float w_l1 = (gout ? g0 + gout[7] : 1.0f) * invN;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float df = r[c] - t[c];
          g[c] += w_l1 * ((df > 0.f) ? 1.f : (df < 0.f ? -1.f : 0.f));
        }
      }
    }
  } else if (f) {
    const float* lg = labels + ((size_t)b * d.M + mgt[ba]) * 5;
    const int cls = (int)lg[0], c0 = (v - 1) * 8;
    const float t_iou = miou[ba];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (c0 + j < d.C) g[j] = (sig_fast(r[5 + c0 + j]) - ((c0 + j == cls) ? t_iou : 0.0f)) * invN * w_cls;
  }
  u32x4 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) o[j] = pack2bf(g[2 * j], g[2 * j + 1]);
  if (v == 0) *(u32x4*)(d_regobj + (size_t)row * 16) = o;
  else *(u32x4*)(d_cls + (size_t)row * cls_ld + (v - 1) * 8) = o;
}

// eval branch: out is BATCH-major [B,A,5+C] (the reference's return layout)
__global__ void k_eval_decode(const plyolo_yolox_desc d, const float* raw, float* out) {
  const int nch = 5 + d.C;
  const size_t total = (size_t)d.B * d.A * nch;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t ba = idx / nch;
  const int c = (int)(idx - ba * nch);
  const int b = (int)(ba / d.A), a = (int)(ba - (size_t)b * d.A);
  const float* r = raw + raw_row(d, b, a) * nch;
  float v;
  if (c >= 4) {
    v = sigmoidf_(r[c]);
  } else {
    float xs, ys, st;
    anchor_geom(d, a, &xs, &ys, &st);
    const int ax = c & 1;  // 0: x, 1: y
    const float ctr = (r[ax] + (ax ? ys : xs)) * st;
    const float ext = expf(r[2 + ax]) * st;
    v = (c < 2) ? ctr - ext / 2 : ctr + ext / 2;
  }
  out[idx] = v;
}

// YOLOv7 eval branch (reference models/losses/yolov7/yolov7_loss.py:50-78) for ONE level:
// raw [B,h,w,na*(5+C)] (channel = a*(5+C)+c) -> out[b][lvl_off + (a*h+gy)*w+gx][5+C] =
// (x1,y1,x2,y2, sig(obj), sig(cls..)),  xy = (sig*2-0.5+grid)*stride, wh = (sig*2)^2*anchor
__global__ void k_v7_eval_decode(const float* raw, int B, int h, int w, int na, int nc, float stride, const float* anchors, float* out,
                                 int A_total, int lvl_off) {
  const int ch = 5 + nc;
  const size_t total = (size_t)B * h * w * na * ch;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % ch);
  size_t t = idx / ch;
  const int a = (int)(t % na);
  t /= na;
  const int gx = (int)(t % w);
  t /= w;
  const int gy = (int)(t % h);
  const int b = (int)(t / h);
  const float* r = raw + (((size_t)(b * h + gy) * w + gx) * na + a) * ch;
  float v;
  if (c >= 4) {
    v = sigmoidf_(r[c]);
  } else {
    const int ax = c & 1;
    const float ctr = (sigmoidf_(r[ax]) * 2.0f - 0.5f + (float)(ax ? gy : gx)) * stride;
    const float e = sigmoidf_(r[2 + ax]) * 2.0f;
    const float ext = e * e * anchors[a * 2 + ax];
    v = (c < 2) ? ctr - ext / 2 : ctr + ext / 2;
  }
  out[((size_t)b * A_total + lvl_off + ((size_t)a * h + gy) * w + gx) * ch + c] = v;
}

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

LossWs carve(const plyolo_yolox_desc* d, void* workspace, size_t* used) {
  const size_t BA = (size_t)d->B * d->A;
  unsigned char* p = (unsigned char*)workspace;
  size_t off = 0;
  LossWs ws;
  ws.dec = (float*)(p + off); off += align256(BA * 16);
  ws.cnt = (int*)(p + off); off += align256(BA * 4);
  ws.lastg = (int*)(p + off); off += align256(BA * 4);
  ws.cand = (uint8_t*)(p + off); off += align256(BA);
  ws.G = (int*)(p + off); off += align256((size_t)d->B * 4);
  const size_t nblk = (BA + 255) / 256;
  ws.partial = (float*)(p + off); off += align256(nblk * NPART * 4);
  ws.costm = (float*)(p + off); off += align256(BA * d->M * 4);
  ws.ioum = (float*)(p + off); off += align256(BA * d->M * 4);
  *used = off;
  return ws;
}

}  // namespace

using plyolo::submit;

extern "C" {

size_t plyolo_yolox_workspace(const plyolo_yolox_desc* d) {
  size_t used;
  carve(d, nullptr, &used);
  return used;
}

int plyolo_yolox_loss_fwd(const plyolo_yolox_desc* dp, const float* raw, const float* labels, uint8_t* fg, int32_t* matched_gt,
                          float* matched_iou, float* losses, void* workspace, size_t ws_bytes, void* stream) {
  const plyolo_yolox_desc d = *dp;
  PLY_CHECK_ARG(d.M <= MAXM, "yolox_loss: at most %d label rows per image (got %d)", MAXM, d.M);
  PLY_CHECK_ARG(d.nlevels >= 1 && d.nlevels <= 8, "yolox_loss: 1..8 levels");
  size_t need;
  const LossWs ws = carve(&d, workspace, &need);
  PLY_CHECK_ARG(ws_bytes >= need, "yolox_loss: workspace too small (%zu < %zu)", ws_bytes, need);
  const size_t BA = (size_t)d.B * d.A;
  const int nblk = (int)((BA + 255) / 256);
  plyolo::annotate("yolox_loss_fwd", 0.0, (double)BA * (5 + d.C) * 4.0 * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    int nchunk = 0;
    for (int l = 0; l < d.nlevels; ++l) nchunk += cdiv(d.lvl_h[l] * d.lvl_w[l], PREP_A);
    const size_t prep_lds = ((size_t)PREP_A * (5 + d.C) + (size_t)d.M * 5 + (4 + 8 + 1) * PREP_A + PREP_T) * 4;
    if (hipError_t ea = plyolo::ensure_dynamic_lds((const void*)k_prep, prep_lds); ea != hipSuccess) return ea;
    hipLaunchKernelGGL(k_prep, dim3(nchunk, d.B), dim3(PREP_T), prep_lds, s, d, raw, labels, ws);
    hipLaunchKernelGGL(k_topk, dim3(d.M, d.B), dim3(256), 0, s, d, raw, labels, ws);
    hipLaunchKernelGGL(k_loss, dim3(nblk), dim3(256), 0, s, d, raw, labels, ws, fg, matched_gt, matched_iou);
    hipLaunchKernelGGL(k_final, dim3(1), dim3(256), 0, s, nblk, d.B, ws, losses);
    return hipGetLastError();
  });
}

int plyolo_yolox_loss_bwd(const plyolo_yolox_desc* dp, const float* raw, const float* labels, const uint8_t* fg,
                          const int32_t* matched_gt, const float* matched_iou, const float* losses, const float* gout,
                          float* draw_f32, void* d_regobj, void* d_cls, int cls_ld, void* stream) {
  const plyolo_yolox_desc d = *dp;
  PLY_CHECK_ARG((draw_f32 != nullptr) != (d_regobj != nullptr && d_cls != nullptr), "yolox_loss_bwd: give draw_f32 OR (d_regobj, d_cls)");
  const size_t total = (size_t)d.B * d.A * (5 + d.C);
  PLY_CHECK_ARG(total < (1ull << 32), "yolox_loss_bwd: B*A*(5+C) must be below 2^32");
  const unsigned grid = (unsigned)cdivz(total, 256);
  plyolo::annotate("yolox_loss_bwd", 0.0, (double)total * 6.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    if (draw_f32)
      hipLaunchKernelGGL(k_bwd<false>, dim3(grid), dim3(256), 0, s, d, raw, labels, fg, matched_gt, matched_iou, losses, gout, draw_f32,
                         (bf16_t*)nullptr, (bf16_t*)nullptr, 0);
    else if (cls_ld % 8 == 0 && cls_ld >= d.C && ((uintptr_t)d_cls & 15) == 0 && ((uintptr_t)d_regobj & 15) == 0) {
      const size_t nvec = (size_t)d.B * d.A * (1 + cls_ld / 8);
      hipLaunchKernelGGL(k_bwd_vec, dim3((unsigned)cdivz(nvec, 256)), dim3(256), 0, s, d, raw, labels, fg, matched_gt, matched_iou, losses,
                         gout, (bf16_t*)d_regobj, (bf16_t*)d_cls, cls_ld);
    } else
      hipLaunchKernelGGL(k_bwd<true>, dim3(grid), dim3(256), 0, s, d, raw, labels, fg, matched_gt, matched_iou, losses, gout,
                         (float*)nullptr, (bf16_t*)d_regobj, (bf16_t*)d_cls, cls_ld);
    return hipGetLastError();
  });
}

int plyolo_yolox_eval_decode(const plyolo_yolox_desc* dp, const float* raw, float* out, void* stream) {
  const plyolo_yolox_desc d = *dp;
  const size_t total = (size_t)d.B * d.A * (5 + d.C);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_eval_decode, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, d, raw, out);
    return hipGetLastError();
  });
}

int plyolo_yolov7_eval_decode(const float* raw_level, int B, int h, int w, int na, int nc, int stride, const float* anchors_dev,
                              float* out, int A_total, int lvl_off, void* stream) {
  const size_t total = (size_t)B * h * w * na * (5 + nc);
  plyolo::annotate("yolov7_eval_decode", 0.0, 8.0 * total);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_v7_eval_decode, dim3((unsigned)cdivz(total, 256)), dim3(256), 0, s, raw_level, B, h, w, na, nc, (float)stride,
                       anchors_dev, out, A_total, lvl_off);
    return hipGetLastError();
  });
}

}  // extern "C"
