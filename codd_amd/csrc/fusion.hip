// Fusion cue / blend kernels (reference model/fusion/fusion.py:168-318, 383-394).  HBM-bound fp32.
#include "common.h"
#include "se3.h"

// ------------------------------------------------------------------------------------------------
// 1/4-resolution cues.  Thread = low-res pixel.  corr_feat [B,31,h,w]:
//   0..8   feature cross correlation   <feat_curr, feat_warp(y+2(ky-1), x+2(kx-1))> / sqrt(CF)
//   9..16  self correlation of feat_curr (centre tap dropped), 17..24 the same for feat_warp
//   25..27 cost_curr(k=-1,0,1) = sum_c |fea_l - warp(fea_r, pc/4 + k)|,  28..30 cost_warp
// and the sub-sampled disparities pc, pw (pred[.., 1::4, 1::4]).
// ------------------------------------------------------------------------------------------------
// channel slices (waves) per 64-pixel group: 8, or 4 when the 3 P^2 + 6 partial sums of 8 slices would not fit the LDS
#define CUES_NS (P >= 5 ? 4 : 8)
// P = corr_cfg.patch_size (nn.Unfold(kernel P, padding P-1, dilation 2): tap (ky, kx) looks at (y + 2ky - (P-1),
// x + 2kx - (P-1)); the self correlations drop tap P*P/2); ds = Fusion.ds_scale.  Channels of corr_feat:
//   [0, P2) cross | [P2, 2P2-1) self(curr) | [2P2-1, 3P2-2) self(warp) | 3 cost_curr | 3 cost_warp      (P2 = P*P)
template <int P>
__global__ __launch_bounds__(512) void fusion_cues_lr_kernel(
    const float* __restrict__ pred_curr, const float* __restrict__ pred_warp, const float* __restrict__ feat_curr,
    const float* __restrict__ feat_warp, const float* __restrict__ fea_l, const float* __restrict__ fea_r, int H, int W,
    int ds, int CF, int CS, float* __restrict__ corr, float* __restrict__ dsub, int dsub_ctot, int dsub_coff) {
  // One thread per low-res pixel would be 540 waves at 960x576, each walking CF + CS channels of dependent gathers
  // (104 us, latency bound): the channels are dealt over CUES_NS waves per pixel group instead and the partial
  // sums combined through LDS in slice order.
  constexpr int P2 = P * P, NA = 3 * P2 + 6;
  extern __shared__ float red_[];  // [CUES_NS][NA][64]
  float(*red)[NA][64] = (float(*)[NA][64])red_;
  const int h = H / ds, w = W / ds, N = h * w, so = ds / 2 - 1;
  const int lane = threadIdx.x, slice = threadIdx.y;
  const int x = blockIdx.x * 64 + lane, y = blockIdx.y, b = blockIdx.z;
  const bool ok = x < w;
  const int xc = ok ? x : w - 1;
  const int pix = y * w + xc;
  const size_t fr_idx = (size_t)b * H * W + (size_t)(ds * y + so) * W + (ds * xc + so);
  const float pc = pred_curr[fr_idx], pw = pred_warp[fr_idx];
  float acc[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) acc[k] = 0.f;
  // stereo matching costs: 4 taps serve the three offsets (xs_k = xs_0 - k)
  const float inv_ds = 1.f / (float)ds;
  const float disp[2] = {pc * inv_ds, pw * inv_ds};
  const float* flb = fea_l + (size_t)b * CS * N + pix;
  const float* frb = fea_r + (size_t)b * CS * N + (size_t)y * w;
  float xsv[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const float xs = (float)xc - disp[s];
    xsv[s] = xs;
    float f0 = floorf(xs);
    const float a = xs - f0;
    f0 = fminf(fmaxf(f0, -4.f), (float)w + 4.f);
    const int i0 = (int)f0;
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int c = slice; c < CS; c += CUES_NS) {
      const float lv = flb[(size_t)c * N];
      const float* rr = frb + (size_t)c * N;
      float t[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int xi = i0 - 1 + q; t[q] = ((unsigned)xi < (unsigned)w) ? rr[xi] : 0.f; }
      const float w1 = a, w0 = 1.f - a;
      c0 += fabsf(lv - (w0 * t[2] + w1 * t[3]));  // k = -1
      c1 += fabsf(lv - (w0 * t[1] + w1 * t[2]));  // k = 0
      c2 += fabsf(lv - (w0 * t[0] + w1 * t[1]));  // k = +1
    }
    acc[3 * P2 + 3 * s] = c0; acc[3 * P2 + 1 + 3 * s] = c1; acc[3 * P2 + 2 + 3 * s] = c2;
  }
  // pixel-to-patch feature correlations: acc[0..P2) cross, [P2..2P2) self(curr), [2P2..3P2) self(warp)
  const float* fc = feat_curr + (size_t)b * CF * N;
  const float* fw = feat_warp + (size_t)b * CF * N;
  for (int c = slice; c < CF; c += CUES_NS) {
    const float kc = fc[(size_t)c * N + pix], kw_ = fw[(size_t)c * N + pix];
#pragma unroll
    for (int k = 0; k < P2; ++k) {
      const int yy = y + 2 * (k / P) - (P - 1), xx = xc + 2 * (k % P) - (P - 1);
      if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) {
        const float mc = fc[(size_t)c * N + yy * w + xx], mw = fw[(size_t)c * N + yy * w + xx];
        acc[k] += kc * mw;
        acc[P2 + k] += kc * mc;
        acc[2 * P2 + k] += kw_ * mw;
      }
    }
  }
#pragma unroll
  for (int k = 0; k < NA; ++k) red[slice][k][lane] = acc[k];
  __syncthreads();
  if (!ok) return;
  // every wave finalises its share of the sums (slice order: deterministic)
  float* out = corr + (size_t)b * (3 * P2 + 4) * N + pix;
  const float nrm = 1.f / sqrtf((float)CF), sc = 1.f / ((float)CS / 24.f);
  for (int k = slice; k < NA; k += CUES_NS) {
    float v = 0.f;
#pragma unroll
    for (int q = 0; q < CUES_NS; ++q) v += red[q][k][lane];
    if (k < P2) out[(size_t)k * N] = v * nrm;
    else if (k < 2 * P2) { const int t = k - P2; if (t != P2 / 2) out[(size_t)(P2 + (t < P2 / 2 ? t : t - 1)) * N] = v * nrm; }
    else if (k < 3 * P2) { const int t = k - 2 * P2; if (t != P2 / 2) out[(size_t)(2 * P2 - 1 + (t < P2 / 2 ? t : t - 1)) * N] = v * nrm; }
    else {
      const float xs = xsv[(k - 3 * P2) / 3];
      out[(size_t)(3 * P2 - 2 + (k - 3 * P2)) * N] = (xs != xs) ? xs : v * sc;
    }
  }
  if (slice == 0) {
    dsub[((size_t)b * dsub_ctot + dsub_coff) * N + pix] = pc;
    dsub[((size_t)b * dsub_ctot + dsub_coff + 1) * N + pix] = pw;
  }
}

template <int P>
static int launch_cues_lr(const float* pred_curr, const float* pred_warp, const float* feat_curr, const float* feat_warp,
                          const float* fea_l, const float* fea_r, int B, int H, int W, int ds, int CF, int CS,
                          float* corr_feat, float* dsub, int dsub_ctot, int dsub_coff, hipStream_t s) {
  const size_t lds = (size_t)CUES_NS * (3 * P * P + 6) * 64 * sizeof(float);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)fusion_cues_lr_kernel<P>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  dim3 grid(cdiv(W / ds, 64), H / ds, B);
  fusion_cues_lr_kernel<P><<<grid, dim3(64, CUES_NS), lds, s>>>(pred_curr, pred_warp, feat_curr, feat_warp, fea_l, fea_r, H,
                                                               W, ds, CF, CS, corr_feat, dsub, dsub_ctot, dsub_coff);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_fusion_cues_lr(const float* pred_curr, const float* pred_warp, const float* feat_curr,
                                   const float* feat_warp, const float* fea_l, const float* fea_r, int B, int H, int W,
                                   int patch, int ds, int CF, int CS, float* corr_feat, float* dsub, int dsub_ctot,
                                   int dsub_coff, void* stream) {
  if (!pred_curr || !pred_warp || !feat_curr || !feat_warp || !fea_l || !fea_r || !corr_feat || !dsub) return CODD_EINVAL;
  if (ds < 2 || (ds & 1) || (H % ds) || (W % ds)) return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
#define X(P)                                                                                                       \
  if (patch == P)                                                                                                  \
    return launch_cues_lr<P>(pred_curr, pred_warp, feat_curr, feat_warp, fea_l, fea_r, B, H, W, ds, CF, CS, corr_feat, \
                             dsub, dsub_ctot, dsub_coff, s);
  X(3) X(5)
#undef X
  return CODD_EUNSUPPORTED;  // patch sizes 3 and 5 are instantiated
}

// ------------------------------------------------------------------------------------------------
// Full-resolution cues: corr_feat_fr [B,3P2+5,H,W] =
//   [0,P2) |pc - pw~(nb)|, [P2,2P2-1) |pc - pc~(nb)| (tap P2/2 dropped), [2P2-1,3P2-2) |pw - pw~(nb)|,
//   3 flow_warp, 1 (pw > 0), 3 confidence_warp      (nb = P x P taps, dilation 2, zero pad)
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ void fusion_cues_fr_kernel(const float* __restrict__ pc_, const float* __restrict__ pw_,
                                      const float* __restrict__ flow_warp, const float* __restrict__ conf_warp, int H,
                                      int W, float* __restrict__ out) {
  constexpr int P2 = P * P;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const size_t N = (size_t)H * W, pix = (size_t)y * W + x;
  const float* pcb = pc_ + (size_t)b * N;
  const float* pwb = pw_ + (size_t)b * N;
  const float pc = pcb[pix], pw = pwb[pix];
  float* o = out + (size_t)b * (3 * P2 + 5) * N + pix;
#pragma unroll
  for (int k = 0; k < P2; ++k) {
    const int yy = y + 2 * (k / P) - (P - 1), xx = x + 2 * (k % P) - (P - 1);
    const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    const float mc = in ? pcb[(size_t)yy * W + xx] : 0.f, mw = in ? pwb[(size_t)yy * W + xx] : 0.f;
    o[(size_t)k * N] = fabsf(pc - mw);
    if (k != P2 / 2) {
      const int kk = k < P2 / 2 ? k : k - 1;
      o[(size_t)(P2 + kk) * N] = fabsf(pc - mc);
      o[(size_t)(2 * P2 - 1 + kk) * N] = fabsf(pw - mw);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    o[(size_t)(3 * P2 - 2 + c) * N] = flow_warp[((size_t)b * 3 + c) * N + pix];
    o[(size_t)(3 * P2 + 2 + c) * N] = conf_warp[((size_t)b * 3 + c) * N + pix];
  }
  o[(size_t)(3 * P2 + 1) * N] = pw > 0.f ? 1.f : 0.f;
}

extern "C" int codd_fusion_cues_fr(const float* pred_curr, const float* pred_warp, const float* flow_warp,
                                   const float* conf_warp, int B, int H, int W, int patch, float* out, void* stream) {
  if (!pred_curr || !pred_warp || !flow_warp || !conf_warp || !out) return CODD_EINVAL;
  dim3 grid(cdiv(W, 256), H, B);
  hipStream_t s = (hipStream_t)stream;
  if (patch == 3) fusion_cues_fr_kernel<3><<<grid, 256, 0, s>>>(pred_curr, pred_warp, flow_warp, conf_warp, H, W, out);
  else if (patch == 5) fusion_cues_fr_kernel<5><<<grid, 256, 0, s>>>(pred_curr, pred_warp, flow_warp, conf_warp, H, W, out);
  else return CODD_EUNSUPPORTED;
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// The whole full-resolution forget branch in ONE kernel (reference fusion.py:123-132, 243-318, 383-394): cues ->
// forget_head = Conv1x1(nc -> 16) -> Conv3x3(16 -> 8, zero padding) -> Conv1x1(8 -> 1) -> sigmoid.  The head has no
// non-linearity before the sigmoid, so it IS one 3x3 convolution of the cue map with the merged weights
//     W_eff[k] = w2 . W1[:, :, k] . W0   (1 x nc per tap k),   beta_k = w2 . W1[:, :, k] . b0,   c0 = w2 . b1 + b2,
// where tap k only contributes (its weights AND its bias beta_k) if the neighbour lies inside the image -- the zero
// padding of the 3x3 acts on the 16-channel map, bias included.  Host code merges the weights (fp64), this kernel
// evaluates   wr(p) = sigmoid(c0 + sum_{k, p + k inside} (W_eff[k] . cues(p + k) + beta_k)):
// the nc-channel cue tensor (70.8 MB at 960x576), the 16- and the 8-channel maps never exist; HBM traffic is the 8
// input planes once + one output plane (22 MB with the blend).
// Workgroup = 16 x 16 outputs; pc / pw tiles with halo 1 + (P - 1) in LDS; every position of the 18 x 18 halo-1 region
// computes its nc cues in registers and leaves its nine per-tap dot products in LDS.
// ------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void fusion_forget_kernel(const float* __restrict__ pc_, const float* __restrict__ pw_,
                                                            const float* __restrict__ flow_warp,
                                                            const float* __restrict__ conf_warp, int H, int W,
                                                            const float* __restrict__ weff, float* __restrict__ wr) {
  constexpr int P2 = P * P, NC = 3 * P2 + 5, HALO = P, TS = 16 + 2 * HALO, RS = 18;  // cue reach P - 1, + 1 for the 3x3
  __shared__ float spc[TS * TS], spw[TS * TS];
  __shared__ float sd[9][RS * RS];
  __shared__ __attribute__((aligned(16))) float sw[9 * NC + 12];  // W_eff [9][NC] | beta[9] | c0
  const int tid = threadIdx.x;
  int bx_, by_, b;  // neighbouring tiles share their halo (22 x 22 of 16 x 16 pixels): keep them on one XCD's L2
  codd_xcd_block(bx_, by_, b);
  const int y0 = by_ * 16, x0 = bx_ * 16;
  const size_t N = (size_t)H * W;
  const float* pcb = pc_ + (size_t)b * N;
  const float* pwb = pw_ + (size_t)b * N;
  for (int e = tid; e < 9 * NC + 10; e += 256) sw[e] = weff[e];
  for (int e = tid; e < TS * TS; e += 256) {
    const int yy = y0 - HALO + e / TS, xx = x0 - HALO + e % TS;
    const bool in = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W;
    const size_t a = in ? (size_t)yy * W + xx : 0;
    const float c = pcb[a], w_ = pwb[a];
    spc[e] = in ? c : 0.f;
    spw[e] = in ? w_ : 0.f;
  }
  __syncthreads();
  for (int q = tid; q < RS * RS; q += 256) {
    const int ry = q / RS, rx = q - ry * RS;
    const int y = y0 - 1 + ry, x = x0 - 1 + rx;
    float d[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) d[k] = 0.f;
    if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
      const int ly = ry - 1 + HALO, lx = rx - 1 + HALO;  // position inside the pc / pw tiles
      const float pc = spc[ly * TS + lx], pw = spw[ly * TS + lx];
      float cue[NC];
#pragma unroll
      for (int k = 0; k < P2; ++k) {
        const int o = (ly + 2 * (k / P) - (P - 1)) * TS + lx + 2 * (k % P) - (P - 1);
        const float mc = spc[o], mw = spw[o];
        cue[k] = fabsf(pc - mw);
        if (k != P2 / 2) {
          const int kk = k < P2 / 2 ? k : k - 1;
          cue[P2 + kk] = fabsf(pc - mc);
          cue[2 * P2 - 1 + kk] = fabsf(pw - mw);
        }
      }
      const size_t pix = (size_t)y * W + x;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        cue[3 * P2 - 2 + c] = flow_warp[((size_t)b * 3 + c) * N + pix];
        cue[3 * P2 + 2 + c] = conf_warp[((size_t)b * 3 + c) * N + pix];
      }
      cue[3 * P2 + 1] = pw > 0.f ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        float a = sw[9 * NC + k];  // beta_k
#pragma unroll
        for (int c = 0; c < NC; ++c) a = fmaf(sw[k * NC + c], cue[c], a);
        d[k] = a;
      }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) sd[k][q] = d[k];
  }
  __syncthreads();
  const int ty = tid >> 4, tx = tid & 15;
  const int y = y0 + ty, x = x0 + tx;
  if (y >= H || x >= W) return;
  float v = sw[9 * NC + 9];  // c0
#pragma unroll
  for (int k = 0; k < 9; ++k) v += sd[k][(ty + k / 3) * RS + tx + k % 3];  // tap k reads position p + (k/3 - 1, k%3 - 1)
  wr[(size_t)b * N + (size_t)y * W + x] = 1.f / (1.f + expf(-v));
}

extern "C" int codd_fusion_forget(const float* pred_curr, const float* pred_warp, const float* flow_warp,
                                  const float* conf_warp, int B, int H, int W, int patch, const float* weff, float* wr,
                                  void* stream) {
  if (!pred_curr || !pred_warp || !flow_warp || !conf_warp || !weff || !wr || B < 1 || H < 1 || W < 1) return CODD_EINVAL;
  dim3 grid(cdiv(W, 16), cdiv(H, 16), B);
  hipStream_t s = (hipStream_t)stream;
  if (patch == 3) fusion_forget_kernel<3><<<grid, 256, 0, s>>>(pred_curr, pred_warp, flow_warp, conf_warp, H, W, weff, wr);
  else if (patch == 5) fusion_forget_kernel<5><<<grid, 256, 0, s>>>(pred_curr, pred_warp, flow_warp, conf_warp, H, W, weff, wr);
  else return CODD_EUNSUPPORTED;
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Blend (reference fusion.py:344-355 nearest x4 up-sampling of the fusion weights, :383-394).
// ------------------------------------------------------------------------------------------------
__global__ void fusion_blend_kernel(const float* __restrict__ pc_, const float* __restrict__ pw_,
                                    const float* __restrict__ wf_lr, const float* __restrict__ wr_, int H, int W,
                                    int ds, float* __restrict__ fused, float* __restrict__ wf_out,
                                    float* __restrict__ wr_out, long long total) {
  long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int x = (int)(e % W);
  long long t = e / W;
  const int y = (int)(t % H);
  const int b = (int)(t / H);
  const float pc = pc_[e], pw = pw_[e];
  const float valid = pw > 0.f ? 1.f : 0.f;
  const float wf = wf_lr[((size_t)b * (H / ds) + y / ds) * (W / ds) + x / ds] * valid;
  const float wr = wr_[e] * valid;
  fused[e] = pc * (1.f - wf * wr) + pw * wf * wr;
  wf_out[e] = wf;
  wr_out[e] = wr;
}

extern "C" int codd_fusion_blend(const float* pred_curr, const float* pred_warp, const float* wf_lr,
                                 const float* wr_logit_sig, int B, int H, int W, int ds, float* fused, float* wf_out,
                                 float* wr_out, void* stream) {
  if (!pred_curr || !pred_warp || !wf_lr || !wr_logit_sig || !fused || !wf_out || !wr_out || ds < 1) return CODD_EINVAL;
  const long long total = (long long)B * H * W;
  fusion_blend_kernel<<<cdiv(total, 256), 256, 0, (hipStream_t)stream>>>(pred_curr, pred_warp, wf_lr, wr_logit_sig, H, W,
                                                                        ds, fused, wf_out, wr_out, total);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// On-device disparity metrics (reference model/codd.py:456-471, utils/metric.py:9-17,40-54):
// per frame, over the valid mask lo < gt < hi of the cropped region [0,h) x [0,w):
//   epe = mean |pred - gt|,  thN = mean (|pred - gt| > thr)
// and the sequence meters accumulate the per-frame means (AverageMeter semantics):
//   meters[0] += epe, meters[1] += th, meters[2] += 1   (only when the mask is non-empty)
// Two launches, no atomics, no host sync: block partials in fp64, then a single-block finish.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disp_metrics_partial_kernel(const float* __restrict__ pred,
                                                                   const float* __restrict__ gt, int W, int h, int w,
                                                                   float lo, float hi, float thr, long long HW,
                                                                   double* __restrict__ partial) {
  __shared__ double red[3][4];
  const int b = blockIdx.y;
  const long long n = (long long)h * w;
  double se = 0.0, st = 0.0, sc = 0.0;
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int y = (int)(e / w), x = (int)(e - (long long)y * w);
    const size_t idx = (size_t)b * HW + (size_t)y * W + x;
    const float g = gt[idx];
    if (g > lo && g < hi) {
      const float err = fabsf(pred[idx] - g);
      se += (double)err; st += err > thr ? 1.0 : 0.0; sc += 1.0;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); st += __shfl_xor(st, o, 64); sc += __shfl_xor(sc, o, 64); }
  if (lane == 0) { red[0][wave] = se; red[1][wave] = st; red[2][wave] = sc; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* p = partial + ((size_t)b * gridDim.x + blockIdx.x) * 3;
    for (int k = 0; k < 3; ++k) p[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}
__global__ void disp_metrics_finish_kernel(const double* __restrict__ partial, int nblk, int B,
                                           double* __restrict__ meters) {
  // one wave: lane l adds partials l, l + 64, ... in index order, then a butterfly (fixed order: deterministic)
  if (blockIdx.x != 0 || threadIdx.x >= 64) return;
  const int lane = threadIdx.x;
  for (int b = 0; b < B; ++b) {
    double se = 0.0, st = 0.0, sc = 0.0;
    for (int i = lane; i < nblk; i += 64) { const double* p = partial + ((size_t)b * nblk + i) * 3; se += p[0]; st += p[1]; sc += p[2]; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { se += __shfl_xor(se, o, 64); st += __shfl_xor(st, o, 64); sc += __shfl_xor(sc, o, 64); }
    if (lane == 0 && sc > 0.0) { meters[0] += se / sc; meters[1] += st / sc; meters[2] += 1.0; }
  }
}

__global__ void timestamp_kernel(long long* slot) { *slot = wall_clock64(); }
extern "C" int codd_timestamp(long long* slot, void* stream) {
  if (!slot) return CODD_EINVAL;
  timestamp_kernel<<<1, 1, 0, (hipStream_t)stream>>>(slot);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

extern "C" int codd_disp_metrics(const float* pred, const float* gt, int B, int H, int W, int h, int w, float lo,
                                 float hi, float thr, double* scratch, double* meters, void* stream) {
  if (!pred || !gt || !scratch || !meters || h > H || w > W || h < 1 || w < 1) return CODD_EINVAL;
  const int nblk = 128;
  hipStream_t s = (hipStream_t)stream;
  disp_metrics_partial_kernel<<<dim3(nblk, B), 256, 0, s>>>(pred, gt, W, h, w, lo, hi, thr, (long long)H * W, scratch);
  CODD_LAUNCH_CHECK();
  disp_metrics_finish_kernel<<<1, 64, 0, s>>>(scratch, nblk, B, meters);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Temporal metrics (reference model/codd.py:473-521, utils/metric.py:19-37, utils/warp.py:69-92):
// the current frame's (gt, pred, valid mask) are pulled back to the previous frame with the previous
// frame's GT flow (nearest sampling, zeros outside), and over mask_prev & mask_warp & mask_curr
//   tepe = |(pred_w - pred_prev) - (gt_w - gt_prev)|,  rel = tepe / (|gt_w - gt_prev| + 1e-3)
// meters[0..3] += mean tepe, mean (tepe > 3), mean rel, mean (rel > 1); meters[4] += 1 (non-empty);
// meters[5] += mean |flow|, meters[6] += 1.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tepe_partial_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                           const float* __restrict__ pred_prev,
                                                           const float* __restrict__ gt_prev,
                                                           const float* __restrict__ flow,
                                                           const float* __restrict__ gt_mask,
                                                           const float* __restrict__ gt2_prev, int W, int h, int w, float lo,
                                                           float hi, float bf, long long HW, double* __restrict__ partial) {
  // gt_mask: the map the CURRENT frame's validity is taken from (the reference substitutes a constant in-range map when
  // a frame has no disparity ground truth at all, model/codd.py:478-486); gt2_prev: second-frame disparity in the previous
  // frame's coordinates, used instead of the flow-warped ground truth when the data set provides it (:497-499)
  __shared__ double red[6][4];
  const int b = blockIdx.y;
  const long long n = (long long)h * w;
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int y = (int)(e / w), x = (int)(e - (long long)y * w);
    const size_t idx = (size_t)b * HW + (size_t)y * W + x;
    const float fx = flow[(size_t)b * 2 * HW + (size_t)y * W + x], fy = flow[(size_t)b * 2 * HW + HW + (size_t)y * W + x];
    s[5] += (double)sqrtf(fx * fx + fy * fy);
    const float gp = gt_prev[idx];
    const bool mprev = gp > lo && gp < hi;
    // nearest sample of the cropped [h,w] maps at (x + fx, y + fy), align_corners = True
    const float sx = nearbyintf((float)x + fx), sy = nearbyintf((float)y + fy);
    if (!(sx >= 0.f && sx <= (float)(w - 1) && sy >= 0.f && sy <= (float)(h - 1))) continue;
    const size_t sidx = (size_t)b * HW + (size_t)((int)sy) * W + (int)sx;
    float gw = gt[sidx];
    const float pw = pred[sidx], gmw = gt_mask[sidx];
    // mask of the current frame (disp range & |flow| < BF) at the SAMPLED and at the UNWARPED position
    const float fxs = flow[(size_t)b * 2 * HW + (size_t)((int)sy) * W + (int)sx];
    const float fys = flow[(size_t)b * 2 * HW + HW + (size_t)((int)sy) * W + (int)sx];
    const bool mw = gmw > lo && gmw < hi && sqrtf(fxs * fxs + fys * fys) < bf;
    const float gc = gt_mask[idx];
    bool mc = gc > lo && gc < hi && sqrtf(fx * fx + fy * fy) < bf;
    if (gt2_prev) { gw = gt2_prev[idx]; mc = mc && gw > 0.f; }
    if (!(mprev && mw && mc)) continue;
    const float dgt = gw - gp;
    const float te = fabsf((pw - pred_prev[idx]) - dgt);
    const float rel = te / (fabsf(dgt) + 1e-3f);
    s[0] += (double)te; s[1] += te > 3.f ? 1.0 : 0.0; s[2] += (double)rel; s[3] += rel > 1.f ? 1.0 : 0.0; s[4] += 1.0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    if (lane == 0) red[k][wave] = s[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* p = partial + ((size_t)b * gridDim.x + blockIdx.x) * 6;
    for (int k = 0; k < 6; ++k) p[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}
__global__ void tepe_finish_kernel(const double* __restrict__ partial, int nblk, int B, double npix,
                                   double* __restrict__ meters) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int b = 0; b < B; ++b) {
    double s[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < nblk; ++i) for (int k = 0; k < 6; ++k) s[k] += partial[((size_t)b * nblk + i) * 6 + k];
    if (s[4] > 0.0) { for (int k = 0; k < 4; ++k) meters[k] += s[k] / s[4]; meters[4] += 1.0; }
    meters[5] += s[5] / npix; meters[6] += 1.0;
  }
}

extern "C" int codd_tepe_metrics(const float* pred, const float* gt, const float* pred_prev, const float* gt_prev,
                                 const float* flow_prev, const float* gt_mask, const float* gt2_prev, int B, int H, int W,
                                 int h, int w, float lo, float hi, float bf, double* scratch, double* meters,
                                 void* stream) {
  if (!pred || !gt || !pred_prev || !gt_prev || !flow_prev || !scratch || !meters || h > H || w > W) return CODD_EINVAL;
  const int nblk = 128;
  hipStream_t s = (hipStream_t)stream;
  tepe_partial_kernel<<<dim3(nblk, B), 256, 0, s>>>(pred, gt, pred_prev, gt_prev, flow_prev, gt_mask ? gt_mask : gt,
                                                    gt2_prev, W, h, w, lo, hi, bf, (long long)H * W, scratch);
  CODD_LAUNCH_CHECK();
  tepe_finish_kernel<<<1, 64, 0, s>>>(scratch, nblk, B, (double)h * w, meters);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Input pre-processing (reference datasets/transforms.py:147-161 Pad(size_divisor=64, reflect),
// :373-427 Normalize(mean, std, to_rgb), datasets/formating.py:65-85): uint8 HWC BGR image ->
// fp32 CHW RGB, (x - mean) / std, reflect-padded (cv2.BORDER_REFLECT_101) on the bottom / right.
// ------------------------------------------------------------------------------------------------
__global__ void preprocess_kernel(const unsigned char* __restrict__ img, int h, int w, int bgr, float m0, float m1,
                                  float m2, float s0, float s1, float s2, int H, int W, float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const int sx = x < w ? x : 2 * (w - 1) - x, sy = y < h ? y : 2 * (h - 1) - y;  // reflect without edge repeat
  const unsigned char* p = img + ((size_t)sy * w + sx) * 3;
  const float c0 = bgr ? p[2] : p[0], c1 = p[1], c2 = bgr ? p[0] : p[2];
  const size_t N = (size_t)H * W, o = (size_t)y * W + x;
  out[o] = (c0 - m0) / s0; out[N + o] = (c1 - m1) / s1; out[2 * N + o] = (c2 - m2) / s2;
}
extern "C" int codd_preprocess(const unsigned char* img, int h, int w, int bgr, const float* mean, const float* stdv,
                               int H, int W, float* out, void* stream) {
  if (!img || !mean || !stdv || !out || H < h || W < w || H - h >= h || W - w >= w) return CODD_EINVAL;
  preprocess_kernel<<<dim3(cdiv(W, 256), H), 256, 0, (hipStream_t)stream>>>(img, h, w, bgr, mean[0], mean[1], mean[2],
                                                                            stdv[0], stdv[1], stdv[2], H, W, out);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Ablation plug-ins (reference model/fusion/others.py:40-168, model/motion/others.py:11-66).
// ------------------------------------------------------------------------------------------------
// mode 0: KalmanFusion with the reference's constant gain K (its P is never updated);
// mode 1: GTFusion -- pick the estimate closer to the ground truth (gt [B,1,hg,wg], zero outside).
__global__ void fusion_select_kernel(int mode, const float* __restrict__ cur, const float* __restrict__ warp,
                                     const float* __restrict__ gt, int H, int W, int hg, int wg, float K,
                                     float* __restrict__ out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const size_t i = ((size_t)b * H + y) * W + x;
  const float c = cur[i], w = warp[i];
  float v;
  if (mode == 0) {
    v = w + K * (c - w);
    if (w <= 0.f) v = c;
    if (fabsf(w - c) > 1.f) v = c;
  } else {
    const float g = (y < hg && x < wg) ? gt[((size_t)b * hg + y) * wg + x] : 0.f;
    const float d = fabsf(c - g) - fabsf(w - g);
    v = d < -1.f ? c : (d > 1.f ? w : (c + w) / 2.f);
    if (w <= 0.f) v = c;
    if (!(g > 0.f)) v = c;
  }
  out[i] = v;
}
extern "C" int codd_fusion_select(int mode, const float* cur, const float* warp, const float* gt, int B, int H,
                                  int W, int hg, int wg, float K, float* out, void* stream) {
  if (!cur || !warp || !out || (mode == 1 && !gt) || mode < 0 || mode > 1) return CODD_EINVAL;
  fusion_select_kernel<<<dim3(cdiv(W, 256), H, B), 256, 0, (hipStream_t)stream>>>(mode, cur, warp, gt, H, W, hg, wg,
                                                                                   K, out);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// GTMotion: move the previous frame's image / disparity (full resolution) and features (1/4 resolution)
// with the ground-truth flow (nearest sampling, zeros outside; occluded pixels and pixels whose source lies
// outside become 0); disparity additionally minus the GT disparity change.  Quarter-resolution features use
// the FULL-resolution flow sampled at [2::4, 2::4], unscaled, exactly like the reference.
__global__ void gt_motion_full_kernel(const float* __restrict__ img, const float* __restrict__ disp,
                                      const float* __restrict__ flow, const float* __restrict__ dchange,
                                      const unsigned char* __restrict__ occ, int H, int W, int hg, int wg,
                                      float* __restrict__ img_w, float* __restrict__ disp_w,
                                      float* __restrict__ flow3, float* __restrict__ conf) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= W) return;
  const size_t HW = (size_t)H * W, i = (size_t)y * W + x;
  const bool in = y < hg && x < wg;
  const size_t gi = in ? ((size_t)b * hg + y) * wg + x : 0;
  const size_t ghw = (size_t)hg * wg;
  const float fx = in ? flow[(size_t)b * 2 * ghw + (size_t)y * wg + x] : 0.f;
  const float fy = in ? flow[(size_t)b * 2 * ghw + ghw + (size_t)y * wg + x] : 0.f;
  const float dc = in ? dchange[gi] : 0.f;
  const bool oc = in && occ[gi] != 0;
  const float sx = nearbyintf((float)x + fx), sy = nearbyintf((float)y + fy);
  const bool ok = sx >= 0.f && sx <= (float)(W - 1) && sy >= 0.f && sy <= (float)(H - 1) && !oc;
  const size_t s = ok ? (size_t)((int)sy) * W + (int)sx : 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) img_w[((size_t)b * 3 + c) * HW + i] = ok ? img[((size_t)b * 3 + c) * HW + s] : 0.f;
  disp_w[(size_t)b * HW + i] = ok ? disp[(size_t)b * HW + s] - dc : 0.f;
  flow3[((size_t)b * 3 + 0) * HW + i] = fx; flow3[((size_t)b * 3 + 1) * HW + i] = fy; flow3[((size_t)b * 3 + 2) * HW + i] = dc;
#pragma unroll
  for (int c = 0; c < 3; ++c) conf[((size_t)b * 3 + c) * HW + i] = 1.f;
}
__global__ void gt_motion_feat_kernel(const float* __restrict__ feat, const float* __restrict__ flow, int C, int Hq,
                                      int Wq, int hg, int wg, float* __restrict__ feat_w) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y, b = blockIdx.z;
  if (x >= Wq) return;
  const int fy_ = 2 + 4 * y, fx_ = 2 + 4 * x;
  const bool in = fy_ < hg && fx_ < wg;
  const size_t ghw = (size_t)hg * wg;
  const float fx = in ? flow[(size_t)b * 2 * ghw + (size_t)fy_ * wg + fx_] : 0.f;
  const float fy = in ? flow[(size_t)b * 2 * ghw + ghw + (size_t)fy_ * wg + fx_] : 0.f;
  const float sx = nearbyintf((float)x + fx), sy = nearbyintf((float)y + fy);
  const bool ok = sx >= 0.f && sx <= (float)(Wq - 1) && sy >= 0.f && sy <= (float)(Hq - 1);
  const size_t hw = (size_t)Hq * Wq, i = (size_t)y * Wq + x, s = ok ? (size_t)((int)sy) * Wq + (int)sx : 0;
  for (int c = 0; c < C; ++c) feat_w[((size_t)b * C + c) * hw + i] = ok ? feat[((size_t)b * C + c) * hw + s] : 0.f;
}
extern "C" int codd_gt_motion(const float* img_prev, const float* disp_prev, const float* feat_prev, int C,
                              const float* gt_flow, const float* gt_disp_change, const unsigned char* gt_flow_occ,
                              int B, int H, int W, int hg, int wg, float* img_warp, float* feat_warp, float* conf,
                              float* disp_warp, float* flow3, void* stream) {
  if (!img_prev || !disp_prev || !feat_prev || !gt_flow || !gt_disp_change || !gt_flow_occ || !img_warp ||
      !feat_warp || !conf || !disp_warp || !flow3 || hg > H || wg > W || (H & 3) || (W & 3))
    return CODD_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  gt_motion_full_kernel<<<dim3(cdiv(W, 256), H, B), 256, 0, s>>>(img_prev, disp_prev, gt_flow, gt_disp_change,
                                                                 gt_flow_occ, H, W, hg, wg, img_warp, disp_warp, flow3,
                                                                 conf);
  CODD_LAUNCH_CHECK();
  gt_motion_feat_kernel<<<dim3(cdiv(W / 4, 64), H / 4, B), 64, 0, s>>>(feat_prev, gt_flow, C, H / 4, W / 4, hg, wg,
                                                                       feat_warp);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}

// ------------------------------------------------------------------------------------------------
// Scene-flow metric columns (reference model/codd.py:519-575; utils/misc.py:12-36): over the crop [0,h) x [0,w) and
// the mask  lo < gt_disp_prev < hi  &  |gt_flow_prev| < bf  &  |gt_disp_change| < bf  (& not occluded, when `occ`)
//   depth1 = clip(bf / pred_disp_prev, 0, bf);   est = project(Ts * X0) - project(X0), X0 = inv_project(depth1),
//   est.z *= bf   (inverse depth -> disparity);   gt = (flow_x, flow_y, disp_change)
//   meters[0] += #mask, [1] += sum |est - gt|_2 (3-D), [2] += sum |est - gt|_2 (2-D), [3] += #(3-D < 1), [4] += #(2-D < 1)
// Block partials in fp64, single-block finish in a fixed order: deterministic, no host sync.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sceneflow_partial_kernel(
    const float* __restrict__ Ts, const float* __restrict__ pred_prev, const float* __restrict__ gt_prev,
    const float* __restrict__ flow, const float* __restrict__ dchange, const unsigned char* __restrict__ occ, int W,
    int h, int w, float lo, float hi, float bf, float fx, float fy, float cx, float cy, long long HW,
    double* __restrict__ partial) {
  __shared__ double red[5][4];
  const int b = blockIdx.y;
  const long long n = (long long)h * w;
  double s[5] = {0, 0, 0, 0, 0};
  for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += (long long)gridDim.x * 256) {
    const int y = (int)(e / w), x = (int)(e - (long long)y * w);
    const size_t idx = (size_t)b * HW + (size_t)y * W + x;
    const float gp = gt_prev[idx];
    const float fu = flow[(size_t)b * 2 * HW + (size_t)y * W + x], fv = flow[(size_t)b * 2 * HW + HW + (size_t)y * W + x];
    const float dc = dchange[idx];
    if (!(gp > lo && gp < hi && sqrtf(fu * fu + fv * fv) < bf && fabsf(dc) < bf)) continue;
    if (occ && occ[idx]) continue;
    const float d1 = fminf(fmaxf(bf / pred_prev[idx], 0.f), bf);  // bf / 0 = inf -> bf
    const V3 X0 = inv_project(d1, x, y, fx, fy, cx, cy);
    const V3 X1 = se3_act(se3_load(Ts + idx * 7), X0);
    const V3 a = project(X1, fx, fy, cx, cy), c = project(X0, fx, fy, cx, cy);
    const float ex = (a.x - c.x) - fu, ey = (a.y - c.y) - fv, ez = (a.z - c.z) * bf - dc;
    const float e2 = sqrtf(ex * ex + ey * ey), e3 = sqrtf(ex * ex + ey * ey + ez * ez);
    s[0] += 1.0; s[1] += (double)e3; s[2] += (double)e2; s[3] += e3 < 1.f ? 1.0 : 0.0; s[4] += e2 < 1.f ? 1.0 : 0.0;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s[k] += __shfl_xor(s[k], o, 64);
    if (lane == 0) red[k][wave] = s[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* p = partial + ((size_t)b * gridDim.x + blockIdx.x) * 5;
    for (int k = 0; k < 5; ++k) p[k] = red[k][0] + red[k][1] + red[k][2] + red[k][3];
  }
}
__global__ void sceneflow_finish_kernel(const double* __restrict__ partial, int n, double* __restrict__ meters) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  for (int k = 0; k < 5; ++k) {
    double a = 0.0;
    for (int i = 0; i < n; ++i) a += partial[(size_t)i * 5 + k];
    meters[k] += a;
  }
}

extern "C" int codd_sceneflow_metrics(const float* Ts, const float* pred_prev, const float* gt_disp_prev,
                                      const float* gt_flow_prev, const float* gt_disp_change,
                                      const unsigned char* gt_flow_occ, int B, int H, int W, int h, int w, float lo,
                                      float hi, float bf, float fx, float fy, float cx, float cy, double* scratch,
                                      double* meters, void* stream) {
  if (!Ts || !pred_prev || !gt_disp_prev || !gt_flow_prev || !gt_disp_change || !scratch || !meters || h > H || w > W ||
      h < 1 || w < 1 || B < 1)
    return CODD_EINVAL;
  const int nblk = 128;
  hipStream_t s = (hipStream_t)stream;
  sceneflow_partial_kernel<<<dim3(nblk, B), 256, 0, s>>>(Ts, pred_prev, gt_disp_prev, gt_flow_prev, gt_disp_change,
                                                         gt_flow_occ, W, h, w, lo, hi, bf, fx, fy, cx, cy,
                                                         (long long)H * W, scratch);
  CODD_LAUNCH_CHECK();
  sceneflow_finish_kernel<<<1, 64, 0, s>>>(scratch, nblk * B, meters);
  CODD_LAUNCH_CHECK();
  return CODD_OK;
}
