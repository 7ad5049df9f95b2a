// Context-network (mmseg HRNet, reference raft3d.py:109-160, configs/models/codd.py:44-74) helper kernels: bilinear resize of the
// HRModule fuse layers / ResizeConcatConv (torch F.interpolate semantics), add(+relu), and the multi-tensor copy the frame
// graph uses instead of torch.cat.  HBM-bound elementwise work; the convolutions of the network run on the conv family.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// HRNet helpers: bilinear resize (torch F.interpolate semantics) and add(+relu).
// ------------------------------------------------------------------------------------------------
// ``extra`` (optional, contiguous [B, C, Ho, Wo]): added to the accumulation base before the blend is added --
// out = relu?((out + extra) + v) / (extra + v): the "+ x_i" term of an HRModule fuse layer without its own launch
// (the same two rounded additions as an add_relu launch followed by this one)
__global__ void resize_bilinear_kernel(const float* __restrict__ in, int C, int Hi, int Wi, int Ho, int Wo, int ac,
                                       float* __restrict__ out, int out_ctot, int out_coff, int accumulate, int relu,
                                       long long total, const float* __restrict__ extra) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % Wo);
  long long t = e / Wo;
  const int y = (int)(t % Ho); t /= Ho;
  const int c = (int)(t % C);
  const int b = (int)(t / C);
  float sy, sx;
  if (ac) {
    sy = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) * (float)y : 0.f;
    sx = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) * (float)x : 0.f;
  } else {
    sy = fmaxf(((float)Hi / (float)Ho) * ((float)y + 0.5f) - 0.5f, 0.f);
    sx = fmaxf(((float)Wi / (float)Wo) * ((float)x + 0.5f) - 0.5f, 0.f);
  }
  const int y0 = min((int)sy, Hi - 1), x0 = min((int)sx, Wi - 1);
  const int y1 = min(y0 + 1, Hi - 1), x1 = min(x0 + 1, Wi - 1);
  const float ly = sy - (float)y0, lx = sx - (float)x0;
  const float* p = in + ((size_t)b * C + c) * Hi * Wi;
  float v = (1.f - ly) * ((1.f - lx) * p[y0 * Wi + x0] + lx * p[y0 * Wi + x1]) +
            ly * ((1.f - lx) * p[y1 * Wi + x0] + lx * p[y1 * Wi + x1]);
  float* o = out + ((size_t)b * out_ctot + out_coff + c) * Ho * Wo + (size_t)y * Wo + x;
  if (extra) {
    const float xe = extra[e];
    v += accumulate ? __fadd_rn(*o, xe) : xe;
  } else if (accumulate) {
    v += *o;
  }
  if (relu) v = fmaxf(v, 0.f);
  *o = v;
}

extern "C" int codd_resize_bilinear(const float* in, int B, int C, int Hi, int Wi, int Ho, int Wo, int align_corners,
                                    float* out, int out_ctot, int out_coff, int accumulate, int relu, void* stream) {
  if (!in || !out) return CODD_EINVAL;
  const long long total = (long long)B * C * Ho * Wo;
  resize_bilinear_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, C, Hi, Wi, Ho, Wo, align_corners, out,
                                                                            out_ctot, out_coff, accumulate, relu, total, nullptr);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_resize_bilinear_add(const float* in, int B, int C, int Hi, int Wi, int Ho, int Wo, int align_corners,
                                        float* out, int out_ctot, int out_coff, int accumulate, int relu,
                                        const float* extra, void* stream) {
  if (!in || !out || !extra) return CODD_EINVAL;
  const long long total = (long long)B * C * Ho * Wo;
  resize_bilinear_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(in, C, Hi, Wi, Ho, Wo, align_corners, out,
                                                                            out_ctot, out_coff, accumulate, relu, total, extra);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

__global__ void add_relu_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n, int relu,
                                float* __restrict__ y) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  float v = a[e] + (b ? b[e] : 0.f);
  if (relu) v = fmaxf(v, 0.f);
  y[e] = v;
}

extern "C" int codd_add_relu(const float* a, const float* b, long long n, int relu, float* y, void* stream) {
  if (!a || !y) return CODD_EINVAL;
  add_relu_kernel<<<cdiv(n, 256), 256, 0, (hipStream_t)stream>>>(a, b, n, relu, y);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// Up to 8 tensor copies in one launch (the recurrent-state write-back at the end of the captured frame: five
// dependent 8-us launches otherwise).  Element counts and addresses must be multiples of 4 floats / 16 bytes.
struct CopyMany {
  const float4* src[8];
  float4* dst[8];
  long long end[8];  // exclusive prefix ends, in float4 units
  int count;
};
__global__ void copy_many_kernel(const CopyMany c) {
  const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= c.end[c.count - 1]) return;
  int k = 0;
  long long base = 0;
#pragma unroll
  for (int q = 0; q < 7; ++q)
    if (q < c.count - 1 && e >= c.end[q]) { k = q + 1; base = c.end[q]; }
  c.dst[k][e - base] = c.src[k][e - base];
}
extern "C" int codd_copy_many(const float* const* src, float* const* dst, const long long* n, int count, void* stream) {
  if (!src || !dst || !n || count < 1 || count > 8) return CODD_EINVAL;
  CopyMany c;
  long long tot = 0;
  for (int k = 0; k < count; ++k) {
    if (!src[k] || !dst[k] || n[k] < 0 || (n[k] & 3) || ((uintptr_t)src[k] & 15) || ((uintptr_t)dst[k] & 15))
      return CODD_EINVAL;
    c.src[k] = (const float4*)src[k];
    c.dst[k] = (float4*)dst[k];
    tot += n[k] / 4;
    c.end[k] = tot;
  }
  for (int k = count; k < 8; ++k) { c.src[k] = nullptr; c.dst[k] = nullptr; c.end[k] = tot; }
  c.count = count;
  if (tot == 0) return CODD_OK;
  copy_many_kernel<<<cdiv(tot, 256), 256, 0, (hipStream_t)stream>>>(c);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

