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

}  // extern "C"
