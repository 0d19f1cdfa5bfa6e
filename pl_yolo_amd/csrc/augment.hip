// GPU input pipeline: the per-image part of the reference's train / val transforms as ONE launch per
// batch -- HSV colour jitter, horizontal mirror, letterbox resize (bilinear), pad with 114, HWC uint8 BGR -> CHW fp32.
//
// Replaces, per image, augment_hsv + _mirror + preproc of reference models/data/augmentation/data_augments.py:88-133 (the
// DataLoader workers run them with OpenCV on the CPU: coco.py:85; 6 workers cannot feed > 10 k img/s).  OpenCV is not
// available in the build container, so its 8-bit algorithms are RESTATED here and parity against cv2 itself is UNPINNED:
//   cv2.resize(INTER_LINEAR), 8-bit: source coordinate (d + 0.5) * (src / dst) - 0.5, left tap clamped to [0, src-1] (weight
//     moved onto the clamped tap), coefficients rounded to 11-bit fixed point, result (S.b + 2^21) >> 22  (imgproc resize.cpp:
//     HResizeLinear / VResizeLinear with FixedPtCast<int, uchar, 22>; the x86 SIMD variant rounds the intermediate once more
//     and may differ by one level);
//   cv2.cvtColor BGR2HSV, 8-bit: v = max, s = (diff * sdiv[v] + 2^11) >> 12, h from the 12-bit division tables, H in [0, 180)
//     (color_hsv.simd.hpp RGB2HSV_b); HSV2BGR, 8-bit: float sector arithmetic, saturate_cast<uchar>(x * 255) (HSV2RGB_b);
//   the jitter itself: lut_hue = ((x * r0) % 180), lut_sat = clip(x * r1, 0, 255), lut_val = clip(x * r2, 0, 255), truncated
//     to uint8 (data_augments.py:118-121).
// The HIP kernel and oracle/augment.py implement exactly the same integer arithmetic: their outputs are compared bit for bit.
//
// Second half of the file: the pixel work of mosaic / random affine / mixup (reference models/data/mosaic_detection.py:60-247,
// 277-344; oracle/mosaic.py) -- four letterboxed images pasted around a random centre of a 2H x 2W canvas (k_mosaic4),
// cv2.warpAffine(INTER_LINEAR, BORDER_CONSTANT) with OpenCV's fixed-point coordinates (10 fraction bits per matrix product,
// 5-bit bilinear weights, (sum + 2^14) >> 15; imgproc/imgwarp.cpp, restated), letterbox + pad, and the 0.5 / 0.5 blend with a
// mirrored, zero-padded, cropped second image.  The random decisions and the label arithmetic stay on the host (data.py).
#include "common.h"

namespace {

DEVINL int sat_i(double v) {   // cv::saturate_cast<int>(double) = cvRound: nearest, ties to even
  return (int)__double2int_rn(v);
}
DEVINL int sdiv_tab(int i) { return i == 0 ? 0 : sat_i((255 << 12) / (1.0 * i)); }
DEVINL int hdiv_tab180(int i) { return i == 0 ? 0 : sat_i((180 << 12) / (6.0 * i)); }

// cv2 BGR2HSV (8-bit, H in [0,180))
DEVINL void bgr2hsv(int b, int g, int r, int* hh, int* ss, int* vv) {
  int v = b > g ? b : g; v = v > r ? v : r;
  int vmin = b < g ? b : g; vmin = vmin < r ? vmin : r;
  const int diff = v - vmin;
  const int vr = v == r ? -1 : 0, vg = v == g ? -1 : 0;
  const int s = (diff * sdiv_tab(v) + (1 << 11)) >> 12;
  int h = (vr & (g - b)) + (~vr & ((vg & (b - r + 2 * diff)) + ((~vg) & (r - g + 4 * diff))));
  h = (h * hdiv_tab180(diff) + (1 << 11)) >> 12;
  h += h < 0 ? 180 : 0;
  *hh = h; *ss = s; *vv = v;
}
DEVINL int sat_u8(float v) {   // saturate_cast<uchar>(float): cvRound then clamp
  int i = __float2int_rn(v);
  return i < 0 ? 0 : (i > 255 ? 255 : i);
}
// cv2 HSV2BGR (8-bit)
DEVINL void hsv2bgr(int H, int S, int V, int* b, int* g, int* r) {
#pragma clang fp contract(off)   // OpenCV (and the numpy oracle) round every product: no fused multiply-add here
  float h = (float)H, s = S * (1.f / 255.f), v = V * (1.f / 255.f);
  float fb, fg, fr;
  if (s == 0.f) {
    fb = fg = fr = v;
  } else {
    h *= 6.f / 180.f;
    if (h < 0.f) do h += 6.f; while (h < 0.f);
    else if (h >= 6.f) do h -= 6.f; while (h >= 6.f);
    int sector = (int)floorf(h);
    h -= sector;
    if ((unsigned)sector >= 6u) { sector = 0; h = 0.f; }
    float tab[4];
    tab[0] = v;
    tab[1] = v * (1.f - s);
    tab[2] = v * (1.f - s * h);
    tab[3] = v * (1.f - s * (1.f - h));
    const int sd[6][3] = {{1, 3, 0}, {1, 0, 2}, {3, 0, 1}, {0, 2, 1}, {0, 1, 3}, {2, 1, 0}};
    fb = tab[sd[sector][0]]; fg = tab[sd[sector][1]]; fr = tab[sd[sector][2]];
  }
  *b = sat_u8(fb * 255.f); *g = sat_u8(fg * 255.f); *r = sat_u8(fr * 255.f);
}
// the three look-up tables of augment_hsv, evaluated for one value (numpy: int16 * float64, float mod, truncation to uint8)
DEVINL int lut_hue(int x, double r0) { double t = fmod((double)x * r0, 180.0); if (t < 0) t += 180.0; return (int)t & 255; }
DEVINL int lut_clip(int x, double rr) { double t = (double)x * rr; t = t < 0.0 ? 0.0 : (t > 255.0 ? 255.0 : t); return (int)t; }

DEVINL void jitter(const plyolo_aug_image& im, int* b, int* g, int* r) {
  int h, s, v;
  bgr2hsv(*b, *g, *r, &h, &s, &v);
  hsv2bgr(lut_hue(h, im.hgain), lut_clip(s, im.sgain), lut_clip(v, im.vgain), b, g, r);
}

// source pixel (after the optional colour jitter and mirror), channel c of BGR
DEVINL void src_px(const plyolo_aug_image& im, int y, int x, int* bgr) {
  const int xs = im.flip ? im.w - 1 - x : x;
  const unsigned char* p = im.src + ((size_t)y * im.w + xs) * 3;
  bgr[0] = p[0]; bgr[1] = p[1]; bgr[2] = p[2];
  if (im.hsv) jitter(im, &bgr[0], &bgr[1], &bgr[2]);
}

// one thread per output pixel (all 3 channels)
__global__ void k_preproc(const plyolo_aug_image* imgs, int B, int OH, int OW, float* out) {
  const int b = blockIdx.y;
  const plyolo_aug_image im = imgs[b];
  const int dw = im.dw, dh = im.dh;   // int(img.shape[1] * r), int(img.shape[0] * r), computed by the host in float64
  const double sx_scale = (double)im.w / dw, sy_scale = (double)im.h / dh;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < OH * OW; idx += gridDim.x * blockDim.x) {
    const int oy = idx / OW, ox = idx - oy * OW;
    float v[3] = {114.f, 114.f, 114.f};
    if (oy < dh && ox < dw) {
#pragma clang fp contract(off)
      float fx = (float)((ox + 0.5) * sx_scale - 0.5), fy = (float)((oy + 0.5) * sy_scale - 0.5);
      int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
      fx -= x0; fy -= y0;
      if (x0 < 0) { x0 = 0; fx = 0.f; }
      if (x0 >= im.w - 1) { x0 = im.w - 1; fx = 0.f; }
      if (y0 < 0) { y0 = 0; fy = 0.f; }
      if (y0 >= im.h - 1) { y0 = im.h - 1; fy = 0.f; }
      const int x1 = x0 + 1 < im.w ? x0 + 1 : x0, y1 = y0 + 1 < im.h ? y0 + 1 : y0;
      const int ax1 = __float2int_rn(fx * 2048.f), ax0 = __float2int_rn((1.f - fx) * 2048.f);
      const int by1 = __float2int_rn(fy * 2048.f), by0 = __float2int_rn((1.f - fy) * 2048.f);
      int p00[3], p01[3], p10[3], p11[3];
      src_px(im, y0, x0, p00); src_px(im, y0, x1, p01); src_px(im, y1, x0, p10); src_px(im, y1, x1, p11);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int r0 = p00[c] * ax0 + p01[c] * ax1, r1 = p10[c] * ax0 + p11[c] * ax1;   // horizontal pass (x 2^11)
        const int q = (r0 * by0 + r1 * by1 + (1 << 21)) >> 22;                           // vertical pass + rounding
        v[c] = (float)(q < 0 ? 0 : (q > 255 ? 255 : q));
      }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(((size_t)b * 3 + c) * OH + oy) * OW + ox] = v[c];   // swap (2, 0, 1): CHW, BGR order kept
  }
}

// cv2.resize(INTER_LINEAR) of a plain uint8 HWC image, one output pixel (the arithmetic of k_preproc without jitter / mirror)
DEVINL void resize_px(const unsigned char* src, int h, int w, double sx_scale, double sy_scale, int oy, int ox, int* bgr) {
#pragma clang fp contract(off)
  float fx = (float)((ox + 0.5) * sx_scale - 0.5), fy = (float)((oy + 0.5) * sy_scale - 0.5);
  int x0 = (int)floorf(fx), y0 = (int)floorf(fy);
  fx -= x0; fy -= y0;
  if (x0 < 0) { x0 = 0; fx = 0.f; }
  if (x0 >= w - 1) { x0 = w - 1; fx = 0.f; }
  if (y0 < 0) { y0 = 0; fy = 0.f; }
  if (y0 >= h - 1) { y0 = h - 1; fy = 0.f; }
  const int x1 = x0 + 1 < w ? x0 + 1 : x0, y1 = y0 + 1 < h ? y0 + 1 : y0;
  const int ax1 = __float2int_rn(fx * 2048.f), ax0 = __float2int_rn((1.f - fx) * 2048.f);
  const int by1 = __float2int_rn(fy * 2048.f), by0 = __float2int_rn((1.f - fy) * 2048.f);
  const unsigned char* p00 = src + ((size_t)y0 * w + x0) * 3;
  const unsigned char* p01 = src + ((size_t)y0 * w + x1) * 3;
  const unsigned char* p10 = src + ((size_t)y1 * w + x0) * 3;
  const unsigned char* p11 = src + ((size_t)y1 * w + x1) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int r0 = p00[c] * ax0 + p01[c] * ax1, r1 = p10[c] * ax0 + p11[c] * ax1;
    const int q = (r0 * by0 + r1 * by1 + (1 << 21)) >> 22;
    bgr[c] = q < 0 ? 0 : (q > 255 ? 255 : q);
  }
}

struct Mosaic4 { plyolo_mosaic_tile t[4]; };

// one thread per canvas pixel: 114 outside the four quadrant rectangles, the letterboxed source pixel inside
__global__ void k_mosaic4(const Mosaic4 m, int CH, int CW, unsigned char* canvas) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < CH * CW; idx += gridDim.x * blockDim.x) {
    const int y = idx / CW, x = idx - y * CW;
    int v[3] = {114, 114, 114};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const plyolo_mosaic_tile& t = m.t[k];
      if (x >= t.lx1 && x < t.lx2 && y >= t.ly1 && y < t.ly2)
        resize_px(t.src, t.h, t.w, (double)t.w / t.dw, (double)t.h / t.dh, t.sy1 + (y - t.ly1), t.sx1 + (x - t.lx1), v);
    }
    unsigned char* o = canvas + (size_t)idx * 3;
    o[0] = (unsigned char)v[0]; o[1] = (unsigned char)v[1]; o[2] = (unsigned char)v[2];
  }
}

struct Affine6 { double m[6]; };   // the INVERTED matrix (dst -> src), as cv::warpAffine uses it

__global__ void k_warp_affine(const unsigned char* src, int SH, int SW, const Affine6 a, unsigned char* dst, int DH, int DW, int border) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < DH * DW; idx += gridDim.x * blockDim.x) {
    const int y = idx / DW, x = idx - y * DW;
    long long X, Y;
    {
#pragma clang fp contract(off)   // every product and sum rounded, as in OpenCV's scalar code and in the numpy oracle
      const long long adelta = __double2ll_rn(a.m[0] * (double)x * 1024.0), bdelta = __double2ll_rn(a.m[3] * (double)x * 1024.0);
      const long long X0 = __double2ll_rn((a.m[1] * (double)y + a.m[2]) * 1024.0) + 16, Y0 = __double2ll_rn((a.m[4] * (double)y + a.m[5]) * 1024.0) + 16;
      X = (X0 + adelta) >> 5;
      Y = (Y0 + bdelta) >> 5;
    }
    long long sxl = X >> 5, syl = Y >> 5;
    sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);     // saturate_cast<short>
    syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
    const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const bool y0ok = sy >= 0 && sy < SH, y1ok = sy + 1 >= 0 && sy + 1 < SH, x0ok = sx >= 0 && sx < SW, x1ok = sx + 1 >= 0 && sx + 1 < SW;
    const unsigned char* p00 = src + ((size_t)(y0ok ? sy : 0) * SW + (x0ok ? sx : 0)) * 3;
    const unsigned char* p01 = src + ((size_t)(y0ok ? sy : 0) * SW + (x1ok ? sx + 1 : 0)) * 3;
    const unsigned char* p10 = src + ((size_t)(y1ok ? sy + 1 : 0) * SW + (x0ok ? sx : 0)) * 3;
    const unsigned char* p11 = src + ((size_t)(y1ok ? sy + 1 : 0) * SW + (x1ok ? sx + 1 : 0)) * 3;
    unsigned char* o = dst + (size_t)idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int v00 = y0ok && x0ok ? p00[c] : border, v01 = y0ok && x1ok ? p01[c] : border;
      const int v10 = y1ok && x0ok ? p10[c] : border, v11 = y1ok && x1ok ? p11[c] : border;
      const int q = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
      o[c] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
  }
}

// cv2.warpPerspective(INTER_LINEAR, BORDER_CONSTANT): OpenCV's WarpPerspectiveInvoker restated -- destination blocks of 64 columns,
// X0 = M0*x_block + M1*y + M2 in doubles, per pixel W = W0 + M6*x1, W = W ? 32/W : 0, X = cvRound(clamp((X0 + M0*x1)*W)); the
// integer / 5-bit fraction split and the 15-bit bilinear weights are the remap warpAffine shares.
struct Persp9 { double m[9]; };    // the INVERTED 3x3 matrix (dst -> src)

__global__ void k_warp_perspective(const unsigned char* src, int SH, int SW, const Persp9 a, unsigned char* dst, int DH, int DW, int border) {
  const int bw = DW < 64 ? DW : 64;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < DH * DW; idx += gridDim.x * blockDim.x) {
    const int y = idx / DW, x = idx - y * DW;
    long long X, Y;
    {
#pragma clang fp contract(off)   // every product and sum rounded, as in OpenCV's scalar code and in the numpy oracle
      const double xb = (double)((x / bw) * bw), x1 = (double)(x % bw), yd = (double)y;
      const double X0 = a.m[0] * xb + a.m[1] * yd + a.m[2];
      const double Y0 = a.m[3] * xb + a.m[4] * yd + a.m[5];
      const double W0 = a.m[6] * xb + a.m[7] * yd + a.m[8];
      double W = W0 + a.m[6] * x1;
      W = W != 0.0 ? 32.0 / W : 0.0;
      const double lo = -2147483648.0, hi = 2147483647.0;
      const double fX = fmax(lo, fmin(hi, (X0 + a.m[0] * x1) * W));
      const double fY = fmax(lo, fmin(hi, (Y0 + a.m[3] * x1) * W));
      X = __double2ll_rn(fX);
      Y = __double2ll_rn(fY);
    }
    long long sxl = X >> 5, syl = Y >> 5;
    sxl = sxl < -32768 ? -32768 : (sxl > 32767 ? 32767 : sxl);     // saturate_cast<short>
    syl = syl < -32768 ? -32768 : (syl > 32767 ? 32767 : syl);
    const int sx = (int)sxl, sy = (int)syl, fx = (int)(X & 31), fy = (int)(Y & 31);
    const int w00 = (32 - fy) * (32 - fx) * 32, w01 = (32 - fy) * fx * 32, w10 = fy * (32 - fx) * 32, w11 = fy * fx * 32;
    const bool y0ok = sy >= 0 && sy < SH, y1ok = sy + 1 >= 0 && sy + 1 < SH, x0ok = sx >= 0 && sx < SW, x1ok = sx + 1 >= 0 && sx + 1 < SW;
    const unsigned char* p00 = src + ((size_t)(y0ok ? sy : 0) * SW + (x0ok ? sx : 0)) * 3;
    const unsigned char* p01 = src + ((size_t)(y0ok ? sy : 0) * SW + (x1ok ? sx + 1 : 0)) * 3;
    const unsigned char* p10 = src + ((size_t)(y1ok ? sy + 1 : 0) * SW + (x0ok ? sx : 0)) * 3;
    const unsigned char* p11 = src + ((size_t)(y1ok ? sy + 1 : 0) * SW + (x1ok ? sx + 1 : 0)) * 3;
    unsigned char* o = dst + (size_t)idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int v00 = y0ok && x0ok ? p00[c] : border, v01 = y0ok && x1ok ? p01[c] : border;
      const int v10 = y1ok && x0ok ? p10[c] : border, v11 = y1ok && x1ok ? p11[c] : border;
      const int q = (v00 * w00 + v01 * w01 + v10 * w10 + v11 * w11 + (1 << 14)) >> 15;
      o[c] = (unsigned char)(q < 0 ? 0 : (q > 255 ? 255 : q));
    }
  }
}

// dst [OH, OW, 3]: the resized image in the top-left dh x dw, `pad` elsewhere
__global__ void k_resize_pad(const unsigned char* src, int h, int w, int dh, int dw, unsigned char* dst, int OH, int OW, int pad) {
  const double sx_scale = (double)w / dw, sy_scale = (double)h / dh;
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < OH * OW; idx += gridDim.x * blockDim.x) {
    const int y = idx / OW, x = idx - y * OW;
    int v[3] = {pad, pad, pad};
    if (y < dh && x < dw) resize_px(src, h, w, sx_scale, sy_scale, y, x, v);
    unsigned char* o = dst + (size_t)idx * 3;
    o[0] = (unsigned char)v[0]; o[1] = (unsigned char)v[1]; o[2] = (unsigned char)v[2];
  }
}

// out = uint8(0.5 * origin + 0.5 * crop), crop = the (mirrored) other image padded with ZEROS to at least th x tw, cut at (y_off, x_off)
__global__ void k_mixup_blend(const unsigned char* origin, int th, int tw, const unsigned char* other, int bh, int bw, int flip, int x_off,
                              int y_off, unsigned char* out) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < th * tw; idx += gridDim.x * blockDim.x) {
    const int y = idx / tw, x = idx - y * tw;
    const int py = y + y_off, px = x + x_off;
    const bool in = py < bh && px < bw;
    const unsigned char* q = other + ((size_t)(in ? py : 0) * bw + (in ? (flip ? bw - 1 - px : px) : 0)) * 3;
    const unsigned char* o = origin + (size_t)idx * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) out[(size_t)idx * 3 + c] = (unsigned char)(int)(0.5f * (float)o[c] + 0.5f * (float)(in ? q[c] : 0));
  }
}

// ---- rounding cut-out (reference models/data/augmentation/cutout_round.py:6-55)
// per rectangle r (half-open, inside the image) the three channel sums of its uint8 pixels: one workgroup per rectangle, exact integers --
// the host divides by the pixel count in float64, which is what numpy's mean of a uint8 strip does (:21-31)
__global__ __launch_bounds__(256) void k_rect_sums(const unsigned char* img, int W, const int* rects, unsigned long long* sums) {
  const int y0 = rects[blockIdx.x * 4 + 0], y1 = rects[blockIdx.x * 4 + 1], x0 = rects[blockIdx.x * 4 + 2], x1 = rects[blockIdx.x * 4 + 3];
  const int rw = x1 - x0, n = (y1 - y0) * rw;
  unsigned long long a[3] = {0ull, 0ull, 0ull};
  for (int i = threadIdx.x; i < n; i += 256) {
    const int y = y0 + i / rw, x = x0 + i % rw;
    const unsigned char* q = img + ((size_t)y * W + x) * 3;
    a[0] += q[0]; a[1] += q[1]; a[2] += q[2];
  }
  __shared__ unsigned long long red[3][256];
#pragma unroll
  for (int c = 0; c < 3; ++c) red[c][threadIdx.x] = a[c];
  __syncthreads();
  for (int st = 128; st > 0; st >>= 1) {
    if ((int)threadIdx.x < st) {
#pragma unroll
      for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + st];
    }
    __syncthreads();
  }
  if (threadIdx.x < 3) sums[blockIdx.x * 3 + threadIdx.x] = red[threadIdx.x][0];
}

struct Holes { plyolo_rect r[PLYOLO_MAX_HOLES]; int n; double fill[3]; double mix, keep; };

// every pixel of the holes' bounding box walks the holes IN ORDER: inside one, v = uint8(mix * fill + keep * v) -- two float64 products
// and their sum, each rounded (numpy evaluates `mixup * cut + (1 - mixup) * img` term by term: no fused multiply-add), truncated by
// the store into the uint8 image (:52-53); a later hole blends over what an earlier one left
__global__ void k_cutout_holes(unsigned char* img, int W, Holes hs, int bx0, int by0, int bw, int bh) {
  for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < bw * bh; idx += gridDim.x * blockDim.x) {
    const int y = by0 + idx / bw, x = bx0 + idx % bw;
    unsigned char* q = img + ((size_t)y * W + x) * 3;
    int v[3] = {q[0], q[1], q[2]};
    bool hit = false;
    for (int k = 0; k < hs.n; ++k) {
      if (x >= hs.r[k].x1 && x < hs.r[k].x2 && y >= hs.r[k].y1 && y < hs.r[k].y2) {
        hit = true;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const double t = __dadd_rn(__dmul_rn(hs.mix, hs.fill[c]), __dmul_rn(hs.keep, (double)v[c]));
          v[c] = (t >= 0.0 && t < 256.0) ? (int)t : 0;     // in range for every finite fill colour; a NaN fill (an empty strip) stores 0
        }
      }
    }
    if (hit) { q[0] = (unsigned char)v[0]; q[1] = (unsigned char)v[1]; q[2] = (unsigned char)v[2]; }
  }
}

}  // namespace

using plyolo::submit;

extern "C" {

int plyolo_preproc_batch(const plyolo_aug_image* imgs_dev, int B, int out_h, int out_w, float* out, void* stream) {
  PLY_CHECK_ARG(imgs_dev && out && B > 0 && out_h > 0 && out_w > 0, "preproc_batch: bad arguments");
  plyolo::annotate("preproc_batch", 0.0, (double)B * out_h * out_w * 3 * 5.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_preproc, dim3((unsigned)cdiv(out_h * out_w, 256), B), dim3(256), 0, s, imgs_dev, B, out_h, out_w, out);
    return hipGetLastError();
  });
}

int plyolo_mosaic4(const plyolo_mosaic_tile* tiles_host, int canvas_h, int canvas_w, unsigned char* canvas, void* stream) {
  PLY_CHECK_ARG(tiles_host && canvas && canvas_h > 0 && canvas_w > 0, "mosaic4: bad arguments");
  Mosaic4 m;
  for (int k = 0; k < 4; ++k) {
    const plyolo_mosaic_tile& t = tiles_host[k];
    PLY_CHECK_ARG(t.src && t.h > 0 && t.w > 0 && t.dh > 0 && t.dw > 0, "mosaic4: tile %d has no image", k);
    PLY_CHECK_ARG(t.lx1 >= 0 && t.ly1 >= 0 && t.lx2 <= canvas_w && t.ly2 <= canvas_h && t.lx1 <= t.lx2 && t.ly1 <= t.ly2,
                  "mosaic4: tile %d leaves the canvas", k);
    PLY_CHECK_ARG(t.sx1 >= 0 && t.sy1 >= 0 && t.sx1 + (t.lx2 - t.lx1) <= t.dw && t.sy1 + (t.ly2 - t.ly1) <= t.dh,
                  "mosaic4: tile %d shows more than its resized image holds", k);
    m.t[k] = t;
  }
  plyolo::annotate("mosaic4", 0.0, (double)canvas_h * canvas_w * 3 * 2.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_mosaic4, dim3((unsigned)cdiv(canvas_h * canvas_w, 256)), dim3(256), 0, s, m, canvas_h, canvas_w, canvas);
    return hipGetLastError();
  });
}

int plyolo_warp_affine_u8(const unsigned char* src, int sh, int sw, const double* inv6_host, unsigned char* dst, int dh, int dw, int border,
                          void* stream) {
  PLY_CHECK_ARG(src && dst && inv6_host && sh > 0 && sw > 0 && dh > 0 && dw > 0 && border >= 0 && border <= 255, "warp_affine_u8: bad arguments");
  Affine6 a;
  for (int i = 0; i < 6; ++i) a.m[i] = inv6_host[i];
  plyolo::annotate("warp_affine_u8", 0.0, (double)dh * dw * 3 * 5.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_warp_affine, dim3((unsigned)cdiv(dh * dw, 256)), dim3(256), 0, s, src, sh, sw, a, dst, dh, dw, border);
    return hipGetLastError();
  });
}

int plyolo_warp_perspective_u8(const unsigned char* src, int sh, int sw, const double* inv9_host, unsigned char* dst, int dh, int dw, int border,
                               void* stream) {
  PLY_CHECK_ARG(src && dst && inv9_host && sh > 0 && sw > 0 && dh > 0 && dw > 0 && border >= 0 && border <= 255, "warp_perspective_u8: bad arguments");
  Persp9 a;
  for (int i = 0; i < 9; ++i) a.m[i] = inv9_host[i];
  plyolo::annotate("warp_perspective_u8", 0.0, (double)dh * dw * 3 * 5.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_warp_perspective, dim3((unsigned)cdiv(dh * dw, 256)), dim3(256), 0, s, src, sh, sw, a, dst, dh, dw, border);
    return hipGetLastError();
  });
}

int plyolo_resize_pad_u8(const unsigned char* src, int h, int w, int dh, int dw, unsigned char* dst, int out_h, int out_w, int pad, void* stream) {
  PLY_CHECK_ARG(src && dst && h > 0 && w > 0 && dh > 0 && dw > 0 && dh <= out_h && dw <= out_w && pad >= 0 && pad <= 255,
                "resize_pad_u8: bad arguments");
  plyolo::annotate("resize_pad_u8", 0.0, (double)out_h * out_w * 3 * 5.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_resize_pad, dim3((unsigned)cdiv(out_h * out_w, 256)), dim3(256), 0, s, src, h, w, dh, dw, dst, out_h, out_w, pad);
    return hipGetLastError();
  });
}

int plyolo_mixup_blend_u8(const unsigned char* origin, int th, int tw, const unsigned char* other, int bh, int bw, int flip, int x_off, int y_off,
                          unsigned char* out, void* stream) {
  PLY_CHECK_ARG(origin && other && out && th > 0 && tw > 0 && bh > 0 && bw > 0 && x_off >= 0 && y_off >= 0, "mixup_blend_u8: bad arguments");
  plyolo::annotate("mixup_blend_u8", 0.0, (double)th * tw * 3 * 3.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_mixup_blend, dim3((unsigned)cdiv(th * tw, 256)), dim3(256), 0, s, origin, th, tw, other, bh, bw, flip, x_off, y_off, out);
    return hipGetLastError();
  });
}

int plyolo_rect_sums_u8(const unsigned char* img, int H, int W, const int* rects_dev, int n, unsigned long long* sums_dev, void* stream) {
  PLY_CHECK_ARG(img && rects_dev && sums_dev && H > 0 && W > 0 && n > 0, "rect_sums_u8: bad arguments");
  plyolo::annotate("rect_sums_u8", 0.0, 0.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_rect_sums, dim3((unsigned)n), dim3(256), 0, s, img, W, rects_dev, sums_dev);
    return hipGetLastError();
  });
}

int plyolo_cutout_holes_u8(unsigned char* img, int H, int W, const plyolo_rect* holes_host, int n, const double* fill3_host, double mixup,
                           void* stream) {
  PLY_CHECK_ARG(img && holes_host && fill3_host && H > 0 && W > 0 && n > 0 && n <= PLYOLO_MAX_HOLES, "cutout_holes_u8: bad arguments (at most %d holes per call)", PLYOLO_MAX_HOLES);
  Holes hs{};
  hs.n = n;
  int bx0 = W, by0 = H, bx1 = 0, by1 = 0;
  for (int k = 0; k < n; ++k) {
    const plyolo_rect r = holes_host[k];
    PLY_CHECK_ARG(r.x1 >= 0 && r.y1 >= 0 && r.x2 <= W && r.y2 <= H && r.x1 <= r.x2 && r.y1 <= r.y2, "cutout_holes_u8: hole %d outside the image", k);
    hs.r[k] = r;
    if (r.x1 < r.x2 && r.y1 < r.y2) {
      bx0 = r.x1 < bx0 ? r.x1 : bx0; by0 = r.y1 < by0 ? r.y1 : by0;
      bx1 = r.x2 > bx1 ? r.x2 : bx1; by1 = r.y2 > by1 ? r.y2 : by1;
    }
  }
  if (bx1 <= bx0 || by1 <= by0) return 0;      // only empty holes
  for (int c = 0; c < 3; ++c) hs.fill[c] = fill3_host[c];
  hs.mix = mixup;
  hs.keep = 1.0 - mixup;                        // (1 - mixup), evaluated once in float64 as python does
  const int bw = bx1 - bx0, bh = by1 - by0;
  plyolo::annotate("cutout_holes_u8", 0.0, (double)bw * bh * 6.0);
  return submit(stream, [=](hipStream_t s) -> hipError_t {
    hipLaunchKernelGGL(k_cutout_holes, dim3((unsigned)cdiv(bw * bh, 256)), dim3(256), 0, s, img, W, hs, bx0, by0, bw, bh);
    return hipGetLastError();
  });
}

}  // extern "C"
