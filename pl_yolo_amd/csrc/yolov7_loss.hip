// YOLOv7 training loss for gfx950.
//
// Replaces the train branch of YOLOv7Loss.__call__ (reference
// models/losses/yolov7/yolov7_loss.py:80-153), build_targets (:155-306), find_3_positive
// (:308-368) and bbox_iou(CIoU) (:376-410), plus what autograd derives from them.
//
// Data: raw head output fp32, level-major; level l is a dense NHWC block
// [B, h_l, w_l, na*(5+C)] starting at row lvl_row[l] (channel = a*(5+C) + c).
//
//   k_v7_cand    one workgroup per image: the (level, offset class, anchor, GT) slots of
//                find_3_positive are visited IN THE REFERENCE'S ORDER and compacted with a
//                ballot prefix, so candidate n here is candidate n there (the only part that needs the order)
//   k_v7_cdec    one thread per candidate, whole chip: decode + the class-independent part of the cost
//   k_v7_rows    one wave per (image, GT), whole chip: IoU/cost rows in LDS, dynamic k from the ten
//                largest IoUs, votes for the k cheapest candidates (ties -> lowest index)
//   k_v7_match   one workgroup per image: conflict resolution, ordered list of matched (cell, GT)
//                entries, objectness targets
//   (round 3: the candidate and the per-GT stages ran inside the per-image workgroups -- 32 workgroups on 256 CUs, 80 classes of
//    libm arithmetic per candidate in a serial loop, every selection round re-reading its row from global memory: 395 + 275 us)
//   k_v7_obj / k_v7_pos / k_v7_final    the three loss terms (deterministic block partials)
//   k_v7_bwd_obj / k_v7_bwd_pos         d loss / d raw
// These are latency-bound integer/compare kernels; no MFMA.
#include <limits.h>

#include "common.h"

namespace {

constexpr int V7_MAXM = 256;  // label rows per image
constexpr int V7_NL = 3, V7_NA = 3;
constexpr float V7_BALANCE[3] = {0.4f, 1.0f, 4.0f};  // yolov7_loss.py:26

struct V7Ws {
  int* ngt;       // [B]
  int* ncand;     // [B]
  int* cand;      // [B][cap][5]  level, a, gj, gi, t
  float* cbox;    // [B][cap][4]  decoded x1,y1,x2,y2 (pixels)
  float* cS;      // [B][cap]     sum_c BCE(logit(y_c), 0)
  long long* coff; // [B][cap]    element offset of the candidate's (cell, anchor) prediction inside raw
  float* rows;    // [B][4][2][cap]  per-wave IoU / cost rows
  int* selcnt;    // [B][cap]
  int* selgt;     // [B][cap]
  int* nmatch;    // [B]
  int* match;     // [B][cap][6]  level, a, gj, gi, t, last
  float* partial; // [nblk_obj] obj partials, then [B][2] box / cls partials
  int* nlvl;      // [NL]  (zeroed per call, together with tobj)
  float* tobj;    // [rows][na]
};

DEVINL float sig(float x) { return 1.0f / (1.0f + expf(-x)); }
DEVINL float frac1(float x) { return x - floorf(x); }  // torch's x % 1. for the values met here
// F.binary_cross_entropy_with_logits(x, t)
DEVINL float bcewl(float x, float t) { return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x))); }

DEVINL const float* cell_ptr(const plyolo_yolov7_desc& d, const float* raw, int b, int l, int a, int gj, int gi) {
  const int nch = d.na * (5 + d.C);
  return raw + ((size_t)d.lvl_row[l] + ((size_t)b * d.lvl_h[l] + gj) * d.lvl_w[l] + gi) * nch + a * (5 + d.C);
}

// logit of y = sqrt(sig(cls) * sig(obj))   (yolov7_loss.py:236-244)
DEVINL float pair_logit(float cls_logit, float obj_logit) {
#pragma clang fp contract(off)
  const float y = sqrtf(sig(cls_logit) * sig(obj_logit));
  return logf(y / (1.0f - y));
}

// iou_loss.py:395-399,412-414 (xyxy) and the cost of pairing GT g with a candidate (:249-252)
DEVINL void pair_terms(const float* gt, const float* bx, float S, float xcls, float* iou_o, float* cost_o) {
#pragma clang fp contract(off)
  const float tlx = fmaxf(gt[0], bx[0]), tly = fmaxf(gt[1], bx[1]);
  const float brx = fminf(gt[2], bx[2]), bry = fminf(gt[3], bx[3]);
  const float en = (tlx < brx && tly < bry) ? 1.f : 0.f;
  const float inter = (brx - tlx) * (bry - tly) * en;
  const float area_a = (gt[2] - gt[0]) * (gt[3] - gt[1]);
  const float area_b = (bx[2] - bx[0]) * (bx[3] - bx[1]);
  const float iou = inter / (area_a + area_b - inter);
  *iou_o = iou;
  *cost_o = (S - xcls) + 3.0f * (-logf(iou + 1e-8f));
}

DEVINL void gt_xyxy(const float* L, float* o) {
  o[0] = L[1] - L[3] / 2; o[1] = L[2] - L[4] / 2; o[2] = L[1] + L[3] / 2; o[3] = L[2] + L[4] / 2;
}

// ordered block compaction helper: returns this thread's slot (or -1) and advances base
DEVINL int ordered_slot(bool ok, int* s_cnt, int& base) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long bal = __ballot(ok);
  const int wpre = __popcll(bal & ((1ull << lane) - 1ull));
  __syncthreads();  // s_cnt free
  if (lane == 0) s_cnt[wave] = __popcll(bal);
  __syncthreads();
  int woff = 0, tot = 0;
  for (int k = 0; k < 4; ++k) { if (k < wave) woff += s_cnt[k]; tot += s_cnt[k]; }
  const int pos = ok ? base + woff + wpre : -1;
  base += tot;
  return pos;
}

__global__ __launch_bounds__(256) void k_v7_cand(const plyolo_yolov7_desc d, const float* raw, const float* labels, V7Ws ws) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ float s_lab[V7_MAXM][5];
  __shared__ int s_cnt[4];
  __shared__ int s_ngt;
  if (tid == 0) s_ngt = 0;
  __syncthreads();
  for (int m = tid; m < d.M; m += 256) {
    const float* L = labels + ((size_t)b * d.M + m) * 5;
    float sum = 0.f;
    for (int i = 0; i < 5; ++i) { s_lab[m][i] = L[i]; sum += L[i]; }
    if (sum > 0.f) atomicAdd(&s_ngt, 1);  // :83 -- the first n_gt rows are the valid ones (:87-88)
  }
  __syncthreads();
  const int nt = s_ngt, ch = 5 + d.C, cap = d.cand_cap;
  const int per_oc = d.na * nt, per_level = 5 * per_oc, S = d.nlevels * per_level;
  int base = 0;
  for (int s0 = 0; s0 < S; s0 += 256) {
    const int s = s0 + tid;
    bool ok = false;
    int l = 0, a = 0, t = 0, gi = 0, gj = 0;
    float st = 1.f, aw = 1.f, ah = 1.f;
    if (s < S) {
      l = s / per_level;
      int r = s - l * per_level;
      const int oc = r / per_oc;
      r -= oc * per_oc;
      a = r / nt;
      t = r - a * nt;
      st = (float)d.lvl_stride[l];
      aw = d.anchors[l][a][0] / st;
      ah = d.anchors[l][a][1] / st;  // :334
      const float gw = s_lab[t][3] / st, gh = s_lab[t][4] / st;
      const float rw = gw / aw, rh = gh / ah;
      if (fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh)) < 4.0f) {  // :341-342
        const float gx = s_lab[t][1] / st, gy = s_lab[t][2] / st;
        const int W = d.lvl_w[l], H = d.lvl_h[l];
        float offx = 0.f, offy = 0.f;
        ok = true;
        if (oc == 1) { ok = frac1(gx) < 0.5f && gx > 1.0f; offx = 0.5f; }                                     // j (:349)
        else if (oc == 2) { ok = frac1(gy) < 0.5f && gy > 1.0f; offy = 0.5f; }                                // k
        else if (oc == 3) { const float q = (float)W - gx; ok = frac1(q) < 0.5f && q > 1.0f; offx = -0.5f; }   // l (:350)
        else if (oc == 4) { const float q = (float)H - gy; ok = frac1(q) < 0.5f && q > 1.0f; offy = -0.5f; }   // m
        gi = min(max((int)(gx - offx), 0), W - 1);  // :362,367
        gj = min(max((int)(gy - offy), 0), H - 1);
      }
    }
    const int pos = ordered_slot(ok, s_cnt, base);
    if (pos >= 0 && pos < cap) {
      const size_t o = (size_t)b * cap + pos;
      int* c = ws.cand + o * 5;
      c[0] = l; c[1] = a; c[2] = gj; c[3] = gi; c[4] = t;
    }
  }
  if (tid == 0) {
    ws.ngt[b] = nt;
    ws.ncand[b] = base < cap ? base : cap;
  }
}

// one thread per candidate: decoded box (:203-204), S = sum_c BCE(logit(y_c), 0) in class order (:236-246), the offset of its
// prediction row, and a cleared vote counter
__global__ __launch_bounds__(64) void k_v7_cdec(const plyolo_yolov7_desc d, const float* raw, V7Ws ws) {
  const int b = blockIdx.y, n = blockIdx.x * 64 + threadIdx.x, cap = d.cand_cap;
  if (n >= ws.ncand[b]) return;
  const size_t o = (size_t)b * cap + n;
  const int* c = ws.cand + o * 5;
  const int l = c[0], a = c[1], gj = c[2], gi = c[3];
  const float st = (float)d.lvl_stride[l];
  const float aw = d.anchors[l][a][0] / st, ah = d.anchors[l][a][1] / st;  // :334
  const float* p = cell_ptr(d, raw, b, l, a, gj, gi);
  ws.coff[o] = (long long)(p - raw);
  ws.selcnt[o] = 0;
  {
#pragma clang fp contract(off)
    const float cx = (sig(p[0]) * 2.0f - 0.5f + (float)gi) * st;  // :203
    const float cy = (sig(p[1]) * 2.0f - 0.5f + (float)gj) * st;
    const float ew = sig(p[2]) * 2.0f, eh = sig(p[3]) * 2.0f;
    const float bw = ew * ew * aw * st, bh = eh * eh * ah * st;   // :204
    float* bx = ws.cbox + o * 4;
    bx[0] = cx - bw / 2; bx[1] = cy - bh / 2; bx[2] = cx + bw / 2; bx[3] = cy + bh / 2;
  }
  float Ssum = 0.f;
  const float po = p[4];
  for (int c0 = 0; c0 < d.C; c0 += 8) {       // eight logits requested together, summed in class order
    float x[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) x[q] = p[5 + min(c0 + q, d.C - 1)];
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (c0 + q < d.C) Ssum += bcewl(pair_logit(x[q], po), 0.f);
  }
  ws.cS[o] = Ssum;
}

// wave-wide (value, index) selection: larger (or smaller) value wins, ties -> lower index
template <bool MAXI>
DEVINL void wave_pick(float& v, int& i) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    const float ov = __shfl_xor(v, off);
    const int oi = __shfl_xor(i, off);
    const bool better = MAXI ? (ov > v) : (ov < v);
    if (better || (ov == v && oi < i)) { v = ov; i = oi; }
  }
}

// bbox_iou(box.T, tbox, x1y1x2y2=False, CIoU=True) (:376-410); box = (px,py,pw,ph), tb = target.
// grad (optional) = d ciou / d (px,py,pw,ph) with alpha held constant (:404-405).
DEVINL float ciou_terms(const float* bx, const float* tb, float* grad) {
#pragma clang fp contract(off)
  const float eps = 1e-7f;
  const float b1x1 = bx[0] - bx[2] / 2, b1x2 = bx[0] + bx[2] / 2, b1y1 = bx[1] - bx[3] / 2, b1y2 = bx[1] + bx[3] / 2;
  const float b2x1 = tb[0] - tb[2] / 2, b2x2 = tb[0] + tb[2] / 2, b2y1 = tb[1] - tb[3] / 2, b2y2 = tb[1] + tb[3] / 2;
  const float iw = fminf(b1x2, b2x2) - fmaxf(b1x1, b2x1), ih = fminf(b1y2, b2y2) - fmaxf(b1y1, b2y1);
  const float iwc = fmaxf(iw, 0.f), ihc = fmaxf(ih, 0.f);
  const float inter = iwc * ihc;
  const float w1 = b1x2 - b1x1, h1 = b1y2 - b1y1 + eps, w2 = b2x2 - b2x1, h2 = b2y2 - b2y1 + eps;
  const float uni = w1 * h1 + w2 * h2 - inter + eps;
  const float iou = inter / uni;
  const float cw = fmaxf(b1x2, b2x2) - fminf(b1x1, b2x1), chh = fmaxf(b1y2, b2y2) - fminf(b1y1, b2y1);
  const float c2 = cw * cw + chh * chh + eps;
  const float sx = b2x1 + b2x2 - b1x1 - b1x2, sy = b2y1 + b2y2 - b1y1 - b1y2;
  const float rho2 = (sx * sx + sy * sy) / 4;
  const float kk = 4.0f / (3.14159265358979323846f * 3.14159265358979323846f);
  const float A = atanf(w2 / h2) - atanf(w1 / h1);
  const float v = kk * A * A;
  const float alpha = v / (v - iou + (1.0f + eps));
  const float ci = iou - (rho2 / c2 + v * alpha);
  if (grad) {
    // partials w.r.t. the four edges of box 1: index 0 = x1, 1 = x2, 2 = y1, 3 = y2
    float diw[4] = {(iw > 0.f && b1x1 > b2x1) ? -1.f : 0.f, (iw > 0.f && b1x2 < b2x2) ? 1.f : 0.f, 0.f, 0.f};
    float dih[4] = {0.f, 0.f, (ih > 0.f && b1y1 > b2y1) ? -1.f : 0.f, (ih > 0.f && b1y2 < b2y2) ? 1.f : 0.f};
    const float dw1[4] = {-1.f, 1.f, 0.f, 0.f}, dh1[4] = {0.f, 0.f, -1.f, 1.f};
    const float dcw[4] = {b1x1 < b2x1 ? -1.f : 0.f, b1x2 > b2x2 ? 1.f : 0.f, 0.f, 0.f};
    const float dch[4] = {0.f, 0.f, b1y1 < b2y1 ? -1.f : 0.f, b1y2 > b2y2 ? 1.f : 0.f};
    const float dsx[4] = {-1.f, -1.f, 0.f, 0.f}, dsy[4] = {0.f, 0.f, -1.f, -1.f};
    const float den = h1 * h1 + w1 * w1;
    float ge[4];
    for (int e = 0; e < 4; ++e) {
      const float dinter = diw[e] * ihc + iwc * dih[e];
      const float duni = dw1[e] * h1 + w1 * dh1[e] - dinter;
      const float diou = (dinter * uni - inter * duni) / (uni * uni);
      const float dc2 = 2 * cw * dcw[e] + 2 * chh * dch[e];
      const float drho = (2 * sx * dsx[e] + 2 * sy * dsy[e]) / 4;
      const float dA = -(dw1[e] * h1 - w1 * dh1[e]) / den;  // d atan(w1/h1) = (dw1*h1 - w1*dh1)/(h1^2+w1^2)
      const float dv = 2 * kk * A * dA;
      ge[e] = diou - (drho * c2 - rho2 * dc2) / (c2 * c2) - alpha * dv;
    }
    grad[0] = ge[0] + ge[1];
    grad[1] = ge[2] + ge[3];
    grad[2] = (ge[1] - ge[0]) / 2;
    grad[3] = (ge[3] - ge[2]) / 2;
  }
  return ci;
}

// decoded prediction (grid units, relative to its cell) and target box of a matched entry (:109-117)
DEVINL void entry_boxes(const plyolo_yolov7_desc& d, const float* p, const float* L, int l, int a, int gj, int gi, float* bx, float* tb,
                        float* sg) {
#pragma clang fp contract(off)
  const float st = (float)d.lvl_stride[l];
  const float aw = d.anchors[l][a][0] / st, ah = d.anchors[l][a][1] / st;
  for (int i = 0; i < 4; ++i) sg[i] = sig(p[i]);
  bx[0] = sg[0] * 2.0f - 0.5f;
  bx[1] = sg[1] * 2.0f - 0.5f;
  const float ew = sg[2] * 2.0f, eh = sg[3] * 2.0f;
  bx[2] = ew * ew * aw;
  bx[3] = eh * eh * ah;
  tb[0] = L[1] / st - (float)gi;
  tb[1] = L[2] / st - (float)gj;
  tb[2] = L[3] / st;
  tb[3] = L[4] / st;
}

// one wave per (image, GT): the GT's IoU / cost rows over the image's candidates live in LDS (the selection rounds re-read them
// 10 + k times), four candidates' operands are requested together, votes go to the candidates' counters
__global__ __launch_bounds__(64) void k_v7_rows(const plyolo_yolov7_desc d, const float* raw, const float* labels, V7Ws ws) {
  const int b = blockIdx.y, g = blockIdx.x, lane = threadIdx.x;
  const int cap = d.cand_cap, N = ws.ncand[b], G = ws.ngt[b];
  if (g >= G || N == 0) return;
  extern __shared__ __align__(16) float v7_rows_smem[];
  float* iou_row = v7_rows_smem;          // [cap]
  float* cost_row = v7_rows_smem + cap;   // [cap]
  const float* lab = labels + (size_t)b * d.M * 5;
  const float* cbox = ws.cbox + (size_t)b * cap * 4;
  const float* cS = ws.cS + (size_t)b * cap;
  const long long* coff = ws.coff + (size_t)b * cap;
  int* selcnt = ws.selcnt + (size_t)b * cap;
  int* selgt = ws.selgt + (size_t)b * cap;
  float gt[4];
  gt_xyxy(lab + g * 5, gt);
  const int gcls = (int)lab[g * 5];
  constexpr int UB = 4;
  for (int n0 = lane; n0 < N; n0 += 64 * UB) {
    const float* pp[UB];
    float S[UB], bx[UB][4], xc[UB], xo[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int nc = min(n0 + u * 64, N - 1);
      pp[u] = raw + coff[nc];
      S[u] = cS[nc];
#pragma unroll
      for (int q = 0; q < 4; ++q) bx[u][q] = cbox[nc * 4 + q];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) { xc[u] = pp[u][5 + gcls]; xo[u] = pp[u][4]; }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int n = n0 + u * 64;
      if (n < N) {
        float iou, cost;
        pair_terms(gt, bx[u], S[u], pair_logit(xc[u], xo[u]), &iou, &cost);
        iou_row[n] = iou;
        cost_row[n] = cost;
      }
    }
  }
  __syncthreads();
  // dynamic k = clamp(int(sum of the 10 largest IoUs), 1)   (:225-226)
  const int nk = N < 10 ? N : 10;
  float sum = 0.f;
  for (int r = 0; r < nk; ++r) {
    float bv = -2.f;
    int bi = INT_MAX;
#pragma unroll 4
    for (int n = lane; n < N; n += 64) {
      const float v = iou_row[n];
      if (v > bv) { bv = v; bi = n; }
    }
    wave_pick<true>(bv, bi);
    sum += bv;
    if (bi != INT_MAX && (bi & 63) == lane) iou_row[bi] = -1.f;
    __syncthreads();
  }
  int k = (int)sum;
  if (k < 1) k = 1;
  if (k > N) k = N;
  for (int r = 0; r < k; ++r) {  // the k cheapest candidates (:255-259)
    float bv = INFINITY;
    int bi = INT_MAX;
#pragma unroll 4
    for (int n = lane; n < N; n += 64) {
      const float v = cost_row[n];
      if (v < bv) { bv = v; bi = n; }
    }
    wave_pick<false>(bv, bi);
    if (bi == INT_MAX) break;  // nothing finite left
    if ((bi & 63) == lane) {
      cost_row[bi] = INFINITY;
      atomicAdd(&selcnt[bi], 1);
      selgt[bi] = g;
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_v7_match(const plyolo_yolov7_desc d, const float* raw, const float* labels, V7Ws ws) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int cap = d.cand_cap, N = ws.ncand[b], G = ws.ngt[b];
  __shared__ int s_cnt[4];
  __shared__ int s_lvl[V7_NL];
  const float* lab = labels + (size_t)b * d.M * 5;
  const int* cand = ws.cand + (size_t)b * cap * 5;
  const float* cbox = ws.cbox + (size_t)b * cap * 4;
  const float* cS = ws.cS + (size_t)b * cap;
  int* selcnt = ws.selcnt + (size_t)b * cap;
  int* selgt = ws.selgt + (size_t)b * cap;
  if (N == 0 || G == 0) {
    if (tid == 0) ws.nmatch[b] = 0;
    return;
  }
  if (tid < V7_NL) s_lvl[tid] = 0;
  __syncthreads();
  // conflict resolution (:262-268) + ordered list of matched entries
  int base = 0;
  int* match = ws.match + (size_t)b * cap * 6;
  for (int n0 = 0; n0 < N; n0 += 256) {
    const int n = n0 + tid;
    bool ok = false;
    int gsel = 0;
    if (n < N) {
      const int cnt = selcnt[n];
      ok = cnt > 0;
      gsel = selgt[n];
      if (cnt > 1) {
        const int* c = cand + n * 5;
        const float* p = cell_ptr(d, raw, b, c[0], c[1], c[2], c[3]);
        float best = INFINITY;
        for (int g = 0; g < G; ++g) {
          float gt[4], iou, cost;
          gt_xyxy(lab + g * 5, gt);
          pair_terms(gt, cbox + n * 4, cS[n], pair_logit(p[5 + (int)lab[g * 5]], p[4]), &iou, &cost);
          if (cost < best) { best = cost; gsel = g; }
        }
      }
    }
    const int pos = ordered_slot(ok, s_cnt, base);
    if (pos >= 0) {
      const int* c = cand + n * 5;
      int* m = match + pos * 6;
      m[0] = c[0]; m[1] = c[1]; m[2] = c[2]; m[3] = c[3]; m[4] = gsel; m[5] = 1;
      atomicAdd(&s_lvl[c[0]], 1);
    }
  }
  __syncthreads();
  const int Mn = base;
  // objectness target: clamp(ciou, 0) of the LAST entry that names a cell (:126)
  for (int e = tid; e < Mn; e += 256) {
    int* m = match + e * 6;
    bool last = true;
    for (int e2 = e + 1; e2 < Mn && last; ++e2) {
      const int* m2 = match + e2 * 6;
      if (m2[0] == m[0] && m2[1] == m[1] && m2[2] == m[2] && m2[3] == m[3]) last = false;
    }
    m[5] = last ? 1 : 0;
    if (last) {
      float bx[4], tb[4], sg[4];
      const float* p = cell_ptr(d, raw, b, m[0], m[1], m[2], m[3]);
      entry_boxes(d, p, lab + m[4] * 5, m[0], m[1], m[2], m[3], bx, tb, sg);
      const float ci = ciou_terms(bx, tb, nullptr);
      const size_t row = (size_t)d.lvl_row[m[0]] + ((size_t)b * d.lvl_h[m[0]] + m[2]) * d.lvl_w[m[0]] + m[3];
      ws.tobj[row * d.na + m[1]] = fmaxf(ci, 0.f);
    }
  }
  if (tid == 0) ws.nmatch[b] = Mn;
  if (tid < V7_NL && s_lvl[tid] > 0) atomicAdd(&ws.nlvl[tid], s_lvl[tid]);
}

DEVINL float block_sum(float v, float* s_red) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return s_red[0] + s_red[1] + s_red[2] + s_red[3];
}

DEVINL int level_of_row(const plyolo_yolov7_desc& d, size_t row) {
  int l = 0;
  for (int k = 1; k < d.nlevels; ++k)
    if (row >= (size_t)d.lvl_row[k]) l = k;
  return l;
}

// obj term: sum_l balance_l * mean BCEwl(obj, tobj)  (:140); GRAD: d/d obj logits
template <bool GRAD>
__global__ __launch_bounds__(256) void k_v7_obj(const plyolo_yolov7_desc d, const float* raw, V7Ws ws, size_t total,
                                                const float* gout4, float* draw) {
  __shared__ float s_red[4];
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (idx < total) {
    const size_t row = idx / d.na;
    const int a = (int)(idx - row * d.na), l = level_of_row(d, row);
    const size_t o = row * (size_t)(d.na * (5 + d.C)) + a * (5 + d.C) + 4;
    const float scale = V7_BALANCE[l] / ((float)d.B * d.na * d.lvl_h[l] * d.lvl_w[l]);
    const float x = raw[o], t = ws.tobj[idx];
    if (GRAD) draw[o] = (gout4[0] + gout4[2]) * scale * (sig(x) - t);
    else v = bcewl(x, t) * scale;
  }
  if (!GRAD) {
    v = block_sum(v, s_red);
    if (threadIdx.x == 0) ws.partial[blockIdx.x] = v;
  }
}

// backward, dense part in ONE pass over draw: zero everywhere except the objectness channel, which gets d/d obj of the obj term
// (the zero fill of the 274 MB gradient tensor and the strided objectness writes were two passes).  A wave owns 64 consecutive
// (row, anchor) pairs = 64 * (5 + C) contiguous floats: every lane computes ONE objectness gradient, then the wave streams the
// chunk out in 16-byte vectors, picking the gradients up from the owning lanes.
__global__ __launch_bounds__(256) void k_v7_bwd_dense(const plyolo_yolov7_desc d, const float* raw, V7Ws ws, unsigned npairs,
                                                      const float* gout4, float* draw) {
  const unsigned ch = 5u + (unsigned)d.C;
  const float inv_ch = 1.0f / (float)ch;
  const float gw = gout4[0] + gout4[2];
  const unsigned lane = threadIdx.x & 63u, wv = blockIdx.x * 4u + (threadIdx.x >> 6), nwv = gridDim.x * 4u;
  for (unsigned chunk = wv; chunk * 64u < npairs; chunk += nwv) {
    const unsigned pair = chunk * 64u + lane;
    float gobj = 0.f;
    if (pair < npairs) {
      const unsigned row = pair / (unsigned)d.na;
      const int l = level_of_row(d, row);
      const float scale = V7_BALANCE[l] / ((float)d.B * d.na * d.lvl_h[l] * d.lvl_w[l]);
      gobj = gw * scale * (sig(raw[(size_t)pair * ch + 4]) - ws.tobj[pair]);
    }
    const unsigned npc = min(64u, npairs - chunk * 64u), nfl = npc * ch;
    float* base = draw + (size_t)chunk * 64u * ch;            // 64 * ch floats: a multiple of 16 bytes
    for (unsigned e0 = 0u; e0 < nfl; e0 += 256u) {      // wave-uniform trip count: every lane takes part in the shuffles
      const unsigned e = e0 + lane * 4u;
      unsigned q = (unsigned)(((float)e + 0.5f) * inv_ch), r = e - q * ch;   // pair (inside the chunk) and channel of element e
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float gq = __shfl(gobj, (int)(q & 63u));
        o[i] = r == 4u ? gq : 0.f;
        if (++r == ch) { r = 0u; ++q; }
      }
      if (e + 4u <= nfl) *(f32x4*)(base + e) = f32x4{o[0], o[1], o[2], o[3]};
      else
        for (unsigned i = 0; e + i < nfl; ++i) base[e + i] = o[i];     // (no iteration when e >= nfl)
    }
  }
}

// box (:121-122) and class (:131-134) terms over the matched entries of one image
// V7_NP workgroups per image over (entry, class) items: a thread per entry walked its 80 class terms (libm) one after the other
// while most of the workgroup idled; item c == 0 of an entry also carries its box term.  Fixed item -> thread mapping, fixed-order
// partials: deterministic.
constexpr int V7_NP = 8;
__global__ __launch_bounds__(256) void k_v7_pos(const plyolo_yolov7_desc d, const float* raw, const float* labels, V7Ws ws, int nblk_obj) {
  __shared__ float s_red[4];
  const int b = blockIdx.y, cap = d.cand_cap, Mn = ws.nmatch[b];
  const int* match = ws.match + (size_t)b * cap * 6;
  const float* lab = labels + (size_t)b * d.M * 5;
  float box = 0.f, cls = 0.f;
  const int items = Mn * d.C;
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += V7_NP * 256) {
    const int e = it / d.C, c = it - e * d.C;
    const int* m = match + e * 6;
    const int l = m[0], a = m[1], gj = m[2], gi = m[3], t = m[4];
    const float* p = cell_ptr(d, raw, b, l, a, gj, gi);
    const float x = p[5 + c];
    const float nl = (float)ws.nlvl[l];
    const int tc = (int)lab[t * 5];
    cls += bcewl(x, c == tc ? 1.f : 0.f) / (nl * d.C);
    if (c == 0) {
      float bx[4], tb[4], sg[4];
      entry_boxes(d, p, lab + t * 5, l, a, gj, gi, bx, tb, sg);
      box += (1.0f - ciou_terms(bx, tb, nullptr)) / nl;
    }
  }
  box = block_sum(box, s_red);
  cls = block_sum(cls, s_red);
  if (threadIdx.x == 0) {
    ws.partial[nblk_obj + (b * V7_NP + blockIdx.x) * 2 + 0] = box;
    ws.partial[nblk_obj + (b * V7_NP + blockIdx.x) * 2 + 1] = cls;
  }
}

__global__ __launch_bounds__(256) void k_v7_final(const plyolo_yolov7_desc d, V7Ws ws, int nblk_obj, float* out) {
  __shared__ float s_red[4];
  float o = 0.f, bx = 0.f, cl = 0.f;
  for (int i = threadIdx.x; i < nblk_obj; i += 256) o += ws.partial[i];
  for (int i = threadIdx.x; i < d.B * V7_NP; i += 256) { bx += ws.partial[nblk_obj + i * 2]; cl += ws.partial[nblk_obj + i * 2 + 1]; }
  o = block_sum(o, s_red);
  bx = block_sum(bx, s_red);
  cl = block_sum(cl, s_red);
  if (threadIdx.x == 0) {
    const float box = bx * 0.05f, obj = o * 1.0f, cls = cl * (0.5f * d.C / 80.0f);  // :27-29,146-148
    out[0] = box + obj + cls;
    out[1] = box;
    out[2] = obj;
    out[3] = cls;
  }
}

// d loss / d raw for the box and class channels of matched cells (a cell matched twice receives both)
__global__ __launch_bounds__(256) void k_v7_bwd_pos(const plyolo_yolov7_desc d, const float* raw, const float* labels, V7Ws ws,
                                                    const float* gout4, float* draw) {
  const int b = blockIdx.y, cap = d.cand_cap, Mn = ws.nmatch[b], ch = 5 + d.C;
  const int* match = ws.match + (size_t)b * cap * 6;
  const float* lab = labels + (size_t)b * d.M * 5;
  const int items = Mn * (ch - 1);
  // gridDim.x workgroups per image (one workgroup walked ~50 items per thread, each behind its own chain of loads)
  for (int it = blockIdx.x * 256 + threadIdx.x; it < items; it += gridDim.x * 256) {
    const int e = it / (ch - 1);
    int c = it - e * (ch - 1);
    const int* m = match + e * 6;
    const float* p = cell_ptr(d, raw, b, m[0], m[1], m[2], m[3]);
    float* g = draw + (p - raw);
    const float nl = (float)ws.nlvl[m[0]];
    if (c < 4) {
      float bx[4], tb[4], sg[4], gr[4];
      entry_boxes(d, p, lab + m[4] * 5, m[0], m[1], m[2], m[3], bx, tb, sg);
      ciou_terms(bx, tb, gr);
      // d(px)/d(tx) = 2 s (1-s);  d(pw)/d(tw) = 2 * pw * (1-s)   [pw = (2s)^2 * anchor]
      const float dp = c < 2 ? 2.0f * sg[c] * (1.0f - sg[c]) : 2.0f * bx[c] * (1.0f - sg[c]);
      atomicAdd(g + c, (gout4[0] + gout4[1]) * 0.05f * (-gr[c] / nl) * dp);
    } else {
      c -= 4;  // class index
      const int tc = (int)lab[m[4] * 5];
      const float w = (gout4[0] + gout4[3]) * (0.5f * d.C / 80.0f) / (nl * d.C);
      atomicAdd(g + 5 + c, w * (sig(p[5 + c]) - (c == tc ? 1.f : 0.f)));
    }
  }
}

size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t total_rows(const plyolo_yolov7_desc* d) {
  size_t r = 0;
  for (int l = 0; l < d->nlevels; ++l) r += (size_t)d->B * d->lvl_h[l] * d->lvl_w[l];
  return r;
}

V7Ws carve(const plyolo_yolov7_desc* d, void* workspace, size_t* used, size_t* zero_off, size_t* zero_bytes) {
  unsigned char* p = (unsigned char*)workspace;
  const size_t B = d->B, cap = d->cand_cap, rows = total_rows(d);
  const size_t nblk = (rows * d->na + 255) / 256;
  size_t off = 0;
  V7Ws ws;
  ws.ngt = (int*)(p + off); off += al256(B * 4);
  ws.ncand = (int*)(p + off); off += al256(B * 4);
  ws.cand = (int*)(p + off); off += al256(B * cap * 5 * 4);
  ws.cbox = (float*)(p + off); off += al256(B * cap * 4 * 4);
  ws.cS = (float*)(p + off); off += al256(B * cap * 4);
  ws.coff = (long long*)(p + off); off += al256(B * cap * 8);
  ws.rows = (float*)(p + off); off += al256(B * 4 * 2 * cap * 4);
  ws.selcnt = (int*)(p + off); off += al256(B * cap * 4);
  ws.selgt = (int*)(p + off); off += al256(B * cap * 4);
  ws.nmatch = (int*)(p + off); off += al256(B * 4);
  ws.match = (int*)(p + off); off += al256(B * cap * 6 * 4);
  ws.partial = (float*)(p + off); off += al256((nblk + 2 * B * 8) * 4);     // obj partials + V7_NP (box, cls) pairs per image
  *zero_off = off;
  ws.nlvl = (int*)(p + off); off += 256;
  ws.tobj = (float*)(p + off); off += al256(rows * d->na * 4);
  *zero_bytes = off - *zero_off;
  *used = off;
  return ws;
}

int check_desc(const plyolo_yolov7_desc& d) {
  PLY_CHECK_ARG(d.nlevels == V7_NL && d.na == V7_NA, "yolov7_loss: %d levels x %d anchors expected", V7_NL, V7_NA);
  PLY_CHECK_ARG(d.M >= 1 && d.M <= V7_MAXM, "yolov7_loss: 1..%d label rows per image (got %d)", V7_MAXM, d.M);
  PLY_CHECK_ARG(d.cand_cap >= 5 * V7_NA * V7_NL * d.M, "yolov7_loss: cand_cap must be >= 45*M = %d", 5 * V7_NA * V7_NL * d.M);
  return 0;
}

}  // namespace

using plyolo::submit;

extern "C" {

size_t plyolo_yolov7_workspace(const plyolo_yolov7_desc* d) {
  size_t used, zo, zb;
  carve(d, nullptr, &used, &zo, &zb);
  return used;
}

int plyolo_yolov7_loss_fwd(const plyolo_yolov7_desc* dp, const float* raw, const float* labels, float* losses, void* workspace,
                           size_t ws_bytes, void* stream) {
  const plyolo_yolov7_desc d = *dp;
  if (int rc = check_desc(d)) return rc;
  size_t need, zo, zb;
  const V7Ws ws = carve(&d, workspace, &need, &zo, &zb);
  PLY_CHECK_ARG(ws_bytes >= need, "yolov7_loss: workspace too small (%zu < %zu)", ws_bytes, need);
  const size_t total = total_rows(&d) * d.na;
  const int nblk = (int)((total + 255) / 256);
  plyolo::annotate("yolov7_loss_fwd", 0.0, (double)total * (5 + d.C) * 4.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipError_t e = plyolo::fill_async((unsigned char*)workspace + zo, 0, zb, s);
    if (e != hipSuccess) return e;
    const size_t rows_lds = (size_t)d.cand_cap * 2 * 4;
    if (hipError_t el = plyolo::ensure_dynamic_lds((const void*)k_v7_rows, rows_lds); el != hipSuccess) return el;
    hipLaunchKernelGGL(k_v7_cand, dim3(d.B), dim3(256), 0, s, d, raw, labels, ws);
    hipLaunchKernelGGL(k_v7_cdec, dim3((d.cand_cap + 63) / 64, d.B), dim3(64), 0, s, d, raw, ws);
    hipLaunchKernelGGL(k_v7_rows, dim3(d.M, d.B), dim3(64), rows_lds, s, d, raw, labels, ws);
    hipLaunchKernelGGL(k_v7_match, dim3(d.B), dim3(256), 0, s, d, raw, labels, ws);
    hipLaunchKernelGGL(k_v7_obj<false>, dim3(nblk), dim3(256), 0, s, d, raw, ws, total, (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(k_v7_pos, dim3(V7_NP, d.B), dim3(256), 0, s, d, raw, labels, ws, nblk);
    hipLaunchKernelGGL(k_v7_final, dim3(1), dim3(256), 0, s, d, ws, nblk, losses);
    return hipGetLastError();
  });
}

int plyolo_yolov7_loss_bwd(const plyolo_yolov7_desc* dp, const float* raw, const float* labels, const float* gout, float* draw,
                           void* workspace, size_t ws_bytes, void* stream) {
  const plyolo_yolov7_desc d = *dp;
  if (int rc = check_desc(d)) return rc;
  size_t need, zo, zb;
  const V7Ws ws = carve(&d, workspace, &need, &zo, &zb);
  PLY_CHECK_ARG(ws_bytes >= need, "yolov7_loss: workspace too small (%zu < %zu)", ws_bytes, need);
  const size_t rows = total_rows(&d), total = rows * d.na;
  const int nblk = (int)((total + 255) / 256);
  const size_t dbytes = rows * (size_t)d.na * (5 + d.C) * 4;
  plyolo::annotate("yolov7_loss_bwd", 0.0, (double)dbytes);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    if (total < (1ull << 31) && ((uintptr_t)draw & 15) == 0) {
      hipLaunchKernelGGL(k_v7_bwd_dense, dim3(2048), dim3(256), 0, s, d, raw, ws, (unsigned)total, gout, draw);
    } else {
      hipError_t e = plyolo::fill_async(draw, 0, dbytes, s);
      if (e != hipSuccess) return e;
      hipLaunchKernelGGL(k_v7_obj<true>, dim3(nblk), dim3(256), 0, s, d, raw, ws, total, gout, draw);
    }
    hipLaunchKernelGGL(k_v7_bwd_pos, dim3(32, d.B), dim3(256), 0, s, d, raw, labels, ws, gout, draw);
    return hipGetLastError();
  });
}

// test / diagnostics: copy the matched entries of image b (rows of 6 ints: level, anchor, gj, gi, gt row, last)
int plyolo_yolov7_matched(const plyolo_yolov7_desc* dp, const void* workspace, int32_t* counts_dev, int32_t* entries_dev, void* stream) {
  const plyolo_yolov7_desc d = *dp;
  size_t need, zo, zb;
  const V7Ws ws = carve(&d, (void*)workspace, &need, &zo, &zb);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipError_t e = hipMemcpyAsync(counts_dev, ws.nmatch, (size_t)d.B * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) return e;
    return hipMemcpyAsync(entries_dev, ws.match, (size_t)d.B * d.cand_cap * 6 * 4, hipMemcpyDeviceToDevice, s);
  });
}

}  // extern "C"
