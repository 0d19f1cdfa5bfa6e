// Eval post-processing on gfx950: class score/argmax + confidence filter, ordered
// compaction, stable score sort, IoU suppression bitmask (one 64-bit word per wave64
// column block) and the greedy scan -- everything on the device, one launch sequence
// for the whole batch, no per-image host round trips.
//
// Replaces postprocess() (reference models/evaluators/postprocess.py:7-48) and the
// un-vendored torchvision.ops.batched_nms / nms it calls (:30-41).  torchvision rule
// restated: stable descending score order; suppress j>i when
// inter/(area_i+area_j-inter) > thr (strict); batched = coordinate trick
// (boxes + cls*(max_coord+1), fp32) when 4*n <= numel_threshold, else per class.
#include "common.h"

namespace {

struct NmsWs {
  float* conf;     // [B,A]
  int* cls;        // [B,A]
  int* cand_idx;   // [B,cap]
  int* ncand;      // [B]
  float* sdet;     // [B,cap,6] candidates in sorted order (original coords)
  float* nbox;     // [B,cap,4] boxes used for the IoU test (offset in trick mode)
  unsigned long long* mask;  // [B,cap,words]
  int cap, words;
};

size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

NmsWs carve(int B, int A, int cap, void* workspace, size_t* used) {
  unsigned char* p = (unsigned char*)workspace;
  size_t off = 0;
  NmsWs w;
  w.cap = cap;
  w.words = (cap + 63) / 64;
  w.conf = (float*)(p + off); off += align256((size_t)B * A * 4);
  w.cls = (int*)(p + off); off += align256((size_t)B * A * 4);
  w.cand_idx = (int*)(p + off); off += align256((size_t)B * cap * 4);
  w.ncand = (int*)(p + off); off += align256((size_t)B * 4);
  w.sdet = (float*)(p + off); off += align256((size_t)B * cap * 24);
  w.nbox = (float*)(p + off); off += align256((size_t)B * cap * 16);
  w.mask = (unsigned long long*)(p + off); off += align256((size_t)B * cap * w.words * 8);
  *used = off;
  return w;
}

// class max (first index wins ties, like torch.max) and confidence -- postprocess.py:18-20
__global__ void k_score(const float* pred, int B, int A, int C, float* conf, int* cls) {
  const size_t ba = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (ba >= (size_t)B * A) return;
  const float* r = pred + ba * (5 + C);
  float best = r[5];
  int arg = 0;
  for (int c = 1; c < C; ++c) {
    const float v = r[5 + c];
    if (v > best) { best = v; arg = c; }
  }
  conf[ba] = r[4] * best;
  cls[ba] = arg;
}

// ordered compaction of anchors with conf >= thr, truncated to cap (anchor order kept, :23-25)
__global__ __launch_bounds__(256) void k_compact(const float* conf, int A, float thr, int cap, int* cand_idx, int* ncand) {
  const int b = blockIdx.x;
  const int tid = threadIdx.x;
  const int per = (A + 255) / 256;
  const int a0 = tid * per, a1 = min(A, a0 + per);
  int cnt = 0;
  for (int a = a0; a < a1; ++a) cnt += conf[(size_t)b * A + a] >= thr ? 1 : 0;
  __shared__ int sc[256];
  sc[tid] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // Hillis-Steele inclusive scan
    const int v = tid >= o ? sc[tid - o] : 0;
    __syncthreads();
    sc[tid] += v;
    __syncthreads();
  }
  int pos = sc[tid] - cnt;
  for (int a = a0; a < a1; ++a)
    if (conf[(size_t)b * A + a] >= thr) {
      if (pos < cap) cand_idx[(size_t)b * cap + pos] = a;
      ++pos;
    }
  if (tid == 255) ncand[b] = min(sc[255], cap);
}

DEVINL unsigned orderable(float c) {
  const unsigned u = __float_as_uint(c);
  return (u & 0x80000000u) ? ~u : (u ^ 0x80000000u);
}

// One workgroup per image: stable descending score sort (bitonic on (score, position) keys in
// LDS), gather into sorted order, max coordinate, class offsets.  FROM_PRED: candidates come
// from the compaction of `pred`; otherwise from an explicit [B,n_max,6] box list.
template <bool FROM_PRED>
__global__ __launch_bounds__(1024) void k_sort(const float* src, int A, int C, int n_max, const int* nbox_in, NmsWs w, int npow2,
                                                int class_agnostic, int numel_threshold) {
  extern __shared__ __align__(16) unsigned char smem[];
  unsigned long long* key = (unsigned long long*)smem;
  __shared__ float red[16];
  __shared__ float s_max;
  const int b = blockIdx.x, tid = threadIdx.x;
  int n = FROM_PRED ? w.ncand[b] : min(nbox_in[b], w.cap);
  if (!FROM_PRED && tid == 0) w.ncand[b] = n;
  // effective sort size: next pow2 >= n (<= npow2)
  int N = 1;
  while (N < n) N <<= 1;
  if (N > npow2) N = npow2;
  for (int i = tid; i < N; i += 1024) {
    unsigned long long k = ~0ull;
    if (i < n) {
      float sc;
      if (FROM_PRED) sc = w.conf[(size_t)b * A + w.cand_idx[(size_t)b * w.cap + i]];
      else sc = src[((size_t)b * n_max + i) * 6 + 4];
      k = ((unsigned long long)(~orderable(sc)) << 32) | (unsigned)i;
    }
    key[i] = k;
  }
  __syncthreads();
  for (int k = 2; k <= N; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < N; i += 1024) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const bool asc = (i & k) == 0;
          const unsigned long long x = key[i], y = key[ixj];
          if ((x > y) == asc) { key[i] = y; key[ixj] = x; }
        }
      }
      __syncthreads();
    }
  float mx = -INFINITY;
  for (int i = tid; i < n; i += 1024) {
    const int pos = (int)(unsigned)(key[i] & 0xffffffffull);
    float d[6];
    if (FROM_PRED) {
      const int a = w.cand_idx[(size_t)b * w.cap + pos];
      const float* r = src + ((size_t)b * A + a) * (5 + C);
      d[0] = r[0]; d[1] = r[1]; d[2] = r[2]; d[3] = r[3];
      d[4] = w.conf[(size_t)b * A + a];
      d[5] = (float)w.cls[(size_t)b * A + a];
    } else {
      const float* r = src + ((size_t)b * n_max + pos) * 6;
#pragma unroll
      for (int q = 0; q < 6; ++q) d[q] = r[q];
    }
    float* o = w.sdet + ((size_t)b * w.cap + i) * 6;
#pragma unroll
    for (int q = 0; q < 6; ++q) o[q] = d[q];
    mx = fmaxf(fmaxf(mx, fmaxf(d[0], d[1])), fmaxf(d[2], d[3]));
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0) red[tid >> 6] = mx;
  __syncthreads();
  if (tid == 0) {
    float m = red[0];
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    s_max = m;
  }
  __syncthreads();
  const bool trick = !class_agnostic && (4 * n <= numel_threshold);
  const float step = s_max + 1.0f;
  for (int i = tid; i < n; i += 1024) {
    const float* d = w.sdet + ((size_t)b * w.cap + i) * 6;
    const float off = trick ? d[5] * step : 0.0f;
    *(f32x4*)(w.nbox + ((size_t)b * w.cap + i) * 4) = f32x4{d[0] + off, d[1] + off, d[2] + off, d[3] + off};
  }
}

// suppression bitmask: word (i, cb) bit j set  <=>  box cb*64+j (j-th of the column block, > i) is suppressed by i
constexpr int NMS_LDS_N = 1024;   // images with at most this many candidates run entirely in one workgroup (k_nms_lds)

__global__ __launch_bounds__(64) void k_mask(NmsWs w, float thr, int class_agnostic, int numel_threshold) {
  const int b = blockIdx.y;
  const int n = w.ncand[b];
  const int nb = (n + 63) / 64;
  const bool vanilla = !class_agnostic && (4 * n > numel_threshold);
  __shared__ __align__(16) float cb_box[64][4];
  __shared__ float cb_cls[64], cb_area[64];
  const int lane = threadIdx.x;
  for (int t = blockIdx.x; t < nb * nb; t += gridDim.x) {
    const int rb = t / nb, cb = t - rb * nb;
    if (cb < rb) continue;
    const int j0 = cb * 64, i = rb * 64 + lane;
    __syncthreads();
    // the column block's boxes AND this lane's own row are requested together (one round trip; the row used to wait behind the barrier)
    const int jc = j0 + lane < n ? j0 + lane : n - 1, ic = i < n ? i : n - 1;
    const f32x4 v = *(const f32x4*)(w.nbox + ((size_t)b * w.cap + jc) * 4);
    const float vcls = w.sdet[((size_t)b * w.cap + jc) * 6 + 5];
    const f32x4 me = *(const f32x4*)(w.nbox + ((size_t)b * w.cap + ic) * 4);
    const float my_cls = w.sdet[((size_t)b * w.cap + ic) * 6 + 5];
    *(f32x4*)cb_box[lane] = v;
    cb_cls[lane] = vcls;
    cb_area[lane] = (v[2] - v[0]) * (v[3] - v[1]);
    __syncthreads();
    if (i < n) {
      const float my_area = (me[2] - me[0]) * (me[3] - me[1]);
      unsigned long long bits = 0ull;
      const int jn = min(64, n - j0);
      // every lane walks all columns of the block (uniform trip count: the reads of four columns are in flight together);
      // the diagonal block keeps only j > lane
#pragma unroll 4
      for (int j = 0; j < jn; ++j) {
        const f32x4 o = *(const f32x4*)cb_box[j];
        const float xx1 = fmaxf(me[0], o[0]), yy1 = fmaxf(me[1], o[1]);
        const float xx2 = fminf(me[2], o[2]), yy2 = fminf(me[3], o[3]);
        const float iw = fmaxf(0.0f, xx2 - xx1), ih = fmaxf(0.0f, yy2 - yy1);
        const float inter = iw * ih;
        const float ovr = inter / (my_area + cb_area[j] - inter);
        const bool same = !vanilla || (cb_cls[j] == my_cls);
        if (same && ovr > thr) bits |= 1ull << j;
      }
      if (rb == cb) bits &= lane == 63 ? 0ull : ~0ull << (lane + 1);
      w.mask[((size_t)b * w.cap + i) * w.words + cb] = bits;
    }
  }
}

// greedy scan, one wave per image; removed-set lives in registers (3 words per lane -> 12288 boxes)
__global__ __launch_bounds__(64) void k_scan(NmsWs w, int max_det, float* det, int* count) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = w.ncand[b];
  if (n <= NMS_LDS_N) return;
  const int nb = (n + 63) / 64;
  unsigned long long rem0 = 0ull, rem1 = 0ull, rem2 = 0ull;
  int kept = 0;
  for (int i = 0; i < n && kept < max_det; ++i) {
    const int wd = i >> 6;
    const int slot = wd >> 6, owner = wd & 63;
    const unsigned long long mine = slot == 0 ? rem0 : (slot == 1 ? rem1 : rem2);
    const unsigned long long word = __shfl(mine, owner);
    if ((word >> (i & 63)) & 1ull) continue;
    // keep box i
    if (lane < 6) det[((size_t)b * max_det + kept) * 6 + lane] = w.sdet[((size_t)b * w.cap + i) * 6 + lane];
    ++kept;
    const unsigned long long* row = w.mask + ((size_t)b * w.cap + i) * w.words;
    // words below the diagonal block were never written: only read cb >= wd
    if (lane < nb && lane >= wd) rem0 |= row[lane];
    if (lane + 64 < nb && lane + 64 >= wd) rem1 |= row[lane + 64];
    if (lane + 128 < nb && lane + 128 >= wd) rem2 |= row[lane + 128];
  }
  if (lane == 0) count[b] = kept;
}

// Greedy scan of one image with its whole suppression matrix ([n][n/64] words, n <= 1024: 128 KB) staged in LDS.
// The global-memory scan above walks the boxes one by one and pays a dependent L2 round trip (~0.5 us) for every KEPT
// box -- 138 us for 1000 boxes / 300 kept, 73 % of the batch.  Here the matrix k_mask wrote is copied into LDS once
// (coalesced), a block of 64 candidates is resolved with register shuffles only (its diagonal word per lane), and the
// rows of its survivors are OR-ed into the removed-set of the later blocks from LDS, lanes across the words.  Same
// greedy order, same results.  (Computing the mask itself here, on one CU per image, measured 2x SLOWER than the
// chip-wide k_mask: 1 M IoU evaluations per image are ~160 us of VALU time for a single CU.)
// value of lane l (wave-uniform l) as a scalar: v_readlane instead of the LDS-crossbar shuffle, which put ~100 cycles into
// every step of the scan's dependent chain
DEVINL unsigned long long readlane_u64(unsigned long long v, int l) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, l);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(1024) void k_nms_lds(NmsWs w, int max_det, float* det, int* count) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n = w.ncand[b];
  if (n > NMS_LDS_N) return;
  const int nb = (n + 63) / 64;
  extern __shared__ __align__(16) unsigned char nms_smem[];
  unsigned long long* smask = (unsigned long long*)nms_smem;              // [n][nb]
  // the bit matrix -> LDS, four words per thread in flight (a rolled load -> store loop was 16 dependent round trips in front of
  // the scan for 1000 candidates); words below the diagonal block were never written: read anyway (allocated), replaced by 0
  constexpr int CPB = 4;
  const int total = n * nb;
  for (int id0 = tid; id0 < total; id0 += 1024 * CPB) {
    unsigned long long v[CPB];
#pragma unroll
    for (int k = 0; k < CPB; ++k) {
      const int id = min(id0 + k * 1024, total - 1);
      const int i = id / nb, cb = id - i * nb;
      const unsigned long long word = w.mask[((size_t)b * w.cap + i) * w.words + cb];
      v[k] = cb >= (i >> 6) ? word : 0ull;
    }
#pragma unroll
    for (int k = 0; k < CPB; ++k)
      if (id0 + k * 1024 < total) smask[id0 + k * 1024] = v[k];
  }
  __shared__ unsigned long long s_keep[NMS_LDS_N / 64];   // survivors of every 64-candidate block
  __shared__ int s_base[NMS_LDS_N / 64];                  // survivors in front of the block
  if (tid < NMS_LDS_N / 64) { s_keep[tid] = 0ull; s_base[tid] = 0; }
  __syncthreads();
  if (tid < 64) {
    const int lane = tid;
    unsigned long long rem = 0ull;                // lane c: removed bits of block c (nb <= 16 words)
    int kept = 0;
    for (int wd = 0; wd < nb && kept < max_det; ++wd) {
      unsigned long long remw = readlane_u64(rem, wd);
      const int row = wd * 64 + lane;
      const unsigned long long diag = row < n ? smask[(size_t)row * nb + wd] : 0ull;
      const int jn = min(64, n - wd * 64);
      unsigned long long keep = 0ull;
      const int before = kept;
      // walk the candidates that are still alive (one step per SURVIVOR, not per candidate): lowest alive bit -> kept, its
      // diagonal word removes the ones it suppresses
      unsigned long long alive = ~remw & (jn == 64 ? ~0ull : ((1ull << jn) - 1ull));
      while (alive && kept < max_det) {
        const int j = __ffsll((long long)alive) - 1;
        keep |= 1ull << j;
        ++kept;
        alive &= ~(readlane_u64(diag, j) | (1ull << j));
      }
      if (lane == 0) { s_keep[wd] = keep; s_base[wd] = before; }
      if (lane > wd && lane < nb) {               // their rows suppress candidates of the later blocks (four rows in flight)
        unsigned long long k2 = keep;
        while (k2) {
          int jj[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            jj[q] = k2 ? __ffsll((long long)k2) - 1 : -1;
            k2 &= k2 - 1ull;      // 0 stays 0
          }
          unsigned long long r4[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) r4[q] = smask[(size_t)(wd * 64 + (jj[q] < 0 ? jj[0] : jj[q])) * nb + lane];
          rem |= (r4[0] | r4[1]) | (r4[2] | r4[3]);
        }
      }
    }
    if (lane == 0) count[b] = kept;
  }
  __syncthreads();
  // survivors -> output rows, in order, by the whole workgroup at once (copied inside the scan every 64-block waited for its own
  // round trip to the candidate rows: 16 of them in the one wave's dependent chain)
  if (tid < n) {
    const unsigned long long keep = s_keep[tid >> 6];
    const int lane = tid & 63;
    if ((keep >> lane) & 1ull) {
      const int pos = s_base[tid >> 6] + __popcll(keep & ((1ull << lane) - 1ull));
      const float* src = w.sdet + ((size_t)b * w.cap + tid) * 6;
      float* dst = det + ((size_t)b * max_det + pos) * 6;
#pragma unroll
      for (int k = 0; k < 6; ++k) dst[k] = src[k];
    }
  }
}

// ---- evaluation formatting (postprocess.py:95-138): per detection  boxes /= scale (IN PLACE, like the reference),
// xyxy -> (x1, y1, w, h), packed rows (x1, y1, x2, y2, w, h, score, class) of every image of the batch for ONE device->host copy
__global__ void k_format(const plyolo_fmt_image* imgs, int B, float* out) {
  const int b = blockIdx.y;
  const plyolo_fmt_image im = imgs[b];
  for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < im.n; r += gridDim.x * blockDim.x) {
    float* d = im.det + (size_t)r * im.ld;
    const float x1 = d[0] / im.scale, y1 = d[1] / im.scale, x2 = d[2] / im.scale, y2 = d[3] / im.scale;
    d[0] = x1; d[1] = y1; d[2] = x2; d[3] = y2;
    float* o = out + (size_t)(im.row0 + r) * 8;
    o[0] = x1; o[1] = y1; o[2] = x2; o[3] = y2; o[4] = x2 - x1; o[5] = y2 - y1; o[6] = d[4]; o[7] = d[5];
  }
}

int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

hipError_t run_nms_tail(const plyolo_nms_desc& d, const NmsWs& w, float* det, int32_t* count, hipStream_t s) {
  // every image takes exactly one of the two paths (decided on the device from its candidate count)
  const size_t lds = (size_t)NMS_LDS_N * (NMS_LDS_N / 64) * 8;
  if (hipError_t e = plyolo::ensure_dynamic_lds((const void*)k_nms_lds, lds); e != hipSuccess) return e;
  hipLaunchKernelGGL(k_mask, dim3(512, d.B), dim3(64), 0, s, w, d.nms_thre, d.class_agnostic, d.numel_threshold);
  hipLaunchKernelGGL(k_nms_lds, dim3(d.B), dim3(1024), lds, s, w, d.max_det, det, count);
  if (w.cap > NMS_LDS_N) hipLaunchKernelGGL(k_scan, dim3(d.B), dim3(64), 0, s, w, d.max_det, det, count);
  return hipGetLastError();
}

template <bool FROM_PRED>
hipError_t run_sort(const plyolo_nms_desc& d, const float* src, int n_max, const int* nbox, const NmsWs& w, hipStream_t s) {
  const int np2 = next_pow2(w.cap);
  const size_t lds = (size_t)np2 * 8;
  auto kern = k_sort<FROM_PRED>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL(kern, dim3(d.B), dim3(1024), lds, s, src, d.A, d.C, n_max, nbox, w, np2, d.class_agnostic, d.numel_threshold);
  return hipGetLastError();
}

}  // namespace

using plyolo::submit;

extern "C" {

size_t plyolo_postprocess_workspace(const plyolo_nms_desc* d) {
  size_t used;
  const int cap = d->max_nms < d->A ? d->max_nms : d->A;
  carve(d->B, d->A, cap, nullptr, &used);
  return used;
}

int plyolo_postprocess(const plyolo_nms_desc* dp, const float* pred, float* det, int32_t* count, int32_t* ncand, void* workspace,
                       size_t ws_bytes, void* stream) {
  const plyolo_nms_desc d = *dp;
  const int cap = d.max_nms < d.A ? d.max_nms : d.A;
  PLY_CHECK_ARG(cap <= 12288, "postprocess: at most 12288 boxes per image enter NMS (got %d)", cap);
  size_t need;
  const NmsWs w = carve(d.B, d.A, cap, workspace, &need);
  PLY_CHECK_ARG(ws_bytes >= need, "postprocess: workspace too small (%zu < %zu)", ws_bytes, need);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_score, dim3((unsigned)cdivz((size_t)d.B * d.A, 256)), dim3(256), 0, s, pred, d.B, d.A, d.C, w.conf, w.cls);
    hipLaunchKernelGGL(k_compact, dim3(d.B), dim3(256), 0, s, w.conf, d.A, d.conf_thre, cap, w.cand_idx, w.ncand);
    hipError_t e = run_sort<true>(d, pred, 0, nullptr, w, s);
    if (e != hipSuccess) return e;
    e = run_nms_tail(d, w, det, count, s);
    if (e != hipSuccess) return e;
    if (ncand) e = hipMemcpyAsync(ncand, w.ncand, (size_t)d.B * 4, hipMemcpyDeviceToDevice, s);
    return e;
  });
}

int plyolo_format_detections(const plyolo_fmt_image* imgs_dev, int B, int max_rows, float* out, void* stream) {
  PLY_CHECK_ARG(imgs_dev && out && B > 0 && max_rows > 0, "format_detections: bad arguments");
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_format, dim3((unsigned)cdiv(max_rows, 128), B), dim3(128), 0, s, imgs_dev, B, out);
    return hipGetLastError();
  });
}

int plyolo_batched_nms(const plyolo_nms_desc* dp, const float* boxes, int n_max, const int32_t* nbox, float* det, int32_t* count,
                       void* workspace, size_t ws_bytes, void* stream) {
  plyolo_nms_desc d = *dp;
  d.A = n_max;
  const int cap = d.max_nms < n_max ? d.max_nms : n_max;
  PLY_CHECK_ARG(cap <= 12288, "batched_nms: at most 12288 boxes per image (got %d)", cap);
  size_t need;
  const NmsWs w = carve(d.B, d.A, cap, workspace, &need);
  PLY_CHECK_ARG(ws_bytes >= need, "batched_nms: workspace too small (%zu < %zu)", ws_bytes, need);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipError_t e = run_sort<false>(d, boxes, n_max, nbox, w, s);
    if (e != hipSuccess) return e;
    return run_nms_tail(d, w, det, count, s);
  });
}

}  // extern "C"
