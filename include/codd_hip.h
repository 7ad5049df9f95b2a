/*
 * codd_hip.h -- C ABI of libcodd_hip.so, the MI355X (gfx950) kernel library behind CODD's
 * per-frame stereo -> motion -> fusion forward path.
 *
 * The reference (facebookresearch/CODD) has no FFI of its own: it is pure Python on torch ops
 * plus three un-vendored CUDA dependencies (lietorch / lietorch_extras, pytorch3d, mmseg).  Each
 * entry point below therefore replaces a *call site* of the reference (cited file:line, paths
 * relative to the reference root) and is what a ctypes / pybind stub in the reference's plug-in
 * classes would bind.
 *
 * Conventions
 *   - all tensors are fp32, NCHW-contiguous device memory owned by the caller (PyTorch-ROCm);
 *   - a "view" (ptr, ctot, coff) addresses channels [coff, coff+C) of a buffer that physically
 *     holds `ctot` channels per batch item -- producers write straight into concatenated
 *     buffers instead of torch.cat;
 *   - every function enqueues on `stream` (hipStream_t as void*) and returns immediately:
 *     0 on success, a negative CODD_E* code on bad arguments, a positive hipError_t otherwise;
 *   - no allocation, no host synchronisation, nothing read from the process environment: safe under hipGraph
 *     capture and callable concurrently on distinct streams.  The ONLY process-wide state is the small option table
 *     behind codd_set_option() below; every option defaults to the shipped configuration and the host library sets
 *     one only where a test asks for it.
 */
#ifndef CODD_HIP_H
#define CODD_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define CODD_ABI_VERSION 12

#define CODD_OK 0
#define CODD_EINVAL (-1)
#define CODD_EUNSUPPORTED (-2)

/* Options (ABI v12; they replace the CODD_* environment switches the library used to read -- VERDICT r5 item 7).
 *   CODD_OPT_GN_Q4       neighbours per workgroup of the Gauss-Newton builder (default 192, >= 16): the grouping of its
 *                        partial sums, i.e. another fp32 summation order of the same normal equations; the parity tests
 *                        use 256 as a re-association probe.  Changes codd_se3_gn_scratch(): set it before sizing scratch.
 *   CODD_OPT_GN_BUILDER  5 (default): the pixel's own embedding in LDS (125 VGPRs); 3: in registers (157 VGPRs) -- the
 *                        bit-identity reference of the shipped builder.
 * codd_set_option returns the previous value, or CODD_EINVAL for an unknown key / out-of-range value; not thread-safe
 * against concurrent launches (set options before launching). */
enum { CODD_OPT_GN_Q4 = 0, CODD_OPT_GN_BUILDER = 1, CODD_OPT_COUNT = 2 };
int codd_set_option(int key, int value);
int codd_get_option(int key);

/* activation codes for the conv epilogue */
enum { CODD_ACT_NONE = 0, CODD_ACT_LRELU02 = 1, CODD_ACT_RELU = 2, CODD_ACT_SIGMOID = 3,
       CODD_ACT_TANH = 4, CODD_ACT_MISH = 5, CODD_ACT_RELU_CH0 = 6 };

typedef struct {
  const float* ptr;
  int ctot; /* channels physically present per batch item */
  int coff; /* first channel of the view */
} codd_view;

/* A split-bf16 activation tensor (codd_split_bf16 layout: [b][plane hi|lo][octet][hp][wp][8 bf16], image pixel
 * (y, x) at (y + bt, x + bl), planes = 2 for terms 3 / 1 for terms 1) as the DESTINATION of a producer kernel: the
 * producer writes the image interior of channels [8*o8, ...) only; border and padding channels stay as they are
 * (zero in a persistent buffer). */
/* `terms` of a record tensor / a layout-2 convolution: the operand format of the MFMA convolution family.
 *   CODD_TERMS_BF16  (1)   one plane of bf16 records, v_mfma_f32_16x16x32_bf16
 *   CODD_TERMS_SPLIT (3)   two planes (hi | lo) of bf16 records, three MFMAs per product: the fp32-grade default
 *   CODD_TERMS_F16   (16)  one plane of IEEE fp16 records, v_mfma_f32_16x16x32_f16 (ABI v10): 11 mantissa bits instead of
 *                          bf16's 8 at the same MFMA rate -- the reference's own reduced-precision hook is fp16
 *                          (auto_fp16, model/codd.py:37,128, enabled by inference.py:120-122).  Range: |x| > 65504
 *                          saturates to +-inf as in the reference's .half() path; the host policy uses it only for
 *                          RAFT3D's feature encoder and update block (O(1) activations). */
#define CODD_TERMS_BF16 1
#define CODD_TERMS_SPLIT 3
#define CODD_TERMS_F16 16
/*   CODD_TERMS_SPLIT_F16 (48)  two planes (hi | lo) of IEEE-fp16 records, three v_mfma_f32_16x16x32_f16 per product (ABI v11):
 *                          x = hi + lo to 22 significant bits, products to ~2^-22 -- the split scheme at fp32's own grade
 *                          (bf16 split: 16 bits, 2^-17), same MFMA rate.  fp16's RANGE applies: |x| > 65504 becomes +-inf and
 *                          lo parts below 6e-8 vanish (absolute error <= 3e-8 per operand); the host keeps it to stages with
 *                          O(1)-O(1e3) activations and the parity tests watch for non-finite values. */
#define CODD_TERMS_SPLIT_F16 48
#define CODD_TERMS_OK(t) ((t) == 1 || (t) == 3 || (t) == 16 || (t) == 48)
#define CODD_TERMS_PLANES(t) (((t) == 3 || (t) == 48) ? 2 : 1)
#define CODD_TERMS_IS_F16(t) ((t) == 16 || (t) == 48)
typedef struct codd_xs_view {
  void* ptr;
  int c8, hp, wp, bt, bl, o8, terms;
} codd_xs_view;

/* ---------------------------------------------------------------------------------------------
 * Convolution family (MFMA implicit GEMM).  Three kernel families behind one entry point, selected by `layout`:
 *   0 / 1  v_mfma_f32_16x16x4_f32 (exact fp32 k-ordered fma chain);
 *   2      v_mfma_f32_16x16x32_bf16 on split-bf16 operands: terms = 3 evaluates every product as
 *          a_hi b_hi + a_hi b_lo + a_lo b_hi (fp32 accumulate, ~16 mantissa bits per product: the fp32-grade
 *          default), terms = 1 is plain bf16 operands / fp32 accumulate (reference auto_fp16 hook,
 *          model/codd.py:37,128; BASELINE.json configs[4]).
 * Replaces every nn.Conv2d / nn.ConvTranspose2d call of the hot path: HITUNet (backbone.py:8-88),
 * TileInitialization convs (initialization.py:60-156,186-190), TileUpdate* / PostTileUpdate /
 * FinalTileUpdate (propagation.py:89-333), BasicEncoder (blocks/extractor.py:119-199),
 * BasicUpdateBlock + ConvGRU (raft3d.py:44-106, blocks/gru.py:9-35), HRNet + ResizeConcatConv
 * (raft3d.py:109-160), Fusion heads (fusion.py:74-138).
 *
 *   v   = sum_{ci,ky,kx} W[co,ci,ky,kx] * in[ci, oy*sy - pad_t + ky*dil_y, ox*sx - pad_l + kx*dil_x]
 *   v  += bias[co] + res1 + res2          (each optional)
 *   v   = act(v)
 *   out = v + post                        (optional)
 * The input is the channel concatenation of in0 (C0 channels) and in1 (C1 channels, may be 0).
 * `wpacked` is the weight tensor re-laid-out by codd_conv2d_pack_weights for (mb, ck).
 * store_mode 1 = ConvTranspose2d(k=2, s=2) expressed as a 1x1 conv with 4*Cout outputs
 * (co' = (a*2+b)*Cout + co is scattered to out[co][2y+a][2x+b]).
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  codd_view in0, in1;
  int C0, C1;
  int B, Hin, Win;
  const float* wpacked;
  const float* bias; /* [Cout] or NULL */
  codd_view res1, res2, post; /* ptr == NULL when absent */
  float* out;
  int out_ctot, out_coff;
  int Cout, Hout, Wout;
  int kh, kw, sy, sx, pad_t, pad_l, dil_y, dil_x;
  int act;
  int store_mode;
  int mb;  /* 16-channel output blocks per workgroup (1, 2 or 4) */
  int npb; /* 16-pixel blocks per wave (1, 2 or 4) */
  int nw;  /* waves (= 16-pixel tile rows) per workgroup: 0 or 4 (default), or 2 / 8 / 9 (npb 1 only) */
  int ck;  /* input channels staged per LDS chunk (multiple of 4) */
  int layout; /* 0: weights packed by codd_conv2d_pack_weights; 1: quad layout (codd_conv2d_pack_weights_quad;
                 ck 16 or 32, x-stride <= 2, 16-byte aligned rows); 2: split-bf16 kernel (weights packed by
                 codd_conv2d_pack_weights_bf16 for (mb, ck, terms); here nw = tile rows, npb = 16-pixel units per
                 tile row (1 | 2), ck a multiple of 8) */
  int terms;  /* layout 2: 1 = bf16 operands, 3 = split-bf16 (hi/lo) operands, 16 = fp16 operands, 48 = split-fp16 (CODD_TERMS_*) */
  int pgw, cgw; /* layout 2: wave grid of a workgroup (pixel-unit groups x channel-block groups), mb % cgw == 0 */
  /* layout 2: the input (channel concatenation of C0 + C1 channels; in0 / in1 are not read) re-laid-out by
   * codd_split_bf16 with borders (pad_t, pad_l): [B][plane][xs_c8 octets][xs_hp][xs_wp][8] bf16 */
  const void* xs;
  int xs_c8, xs_hp, xs_wp;
  int xs_bt, xs_bl; /* borders the split tensor was made with (>= pad_t, pad_l: one split input can serve several
                       convolutions of the same activation, e.g. a 3x3 and a dilated 3x3) */
  int xs_o8;        /* first channel octet of this conv's input inside the split tensor (channel-slice views) */
  /* layout 2, optional: write the result as split-bf16 records straight into the NEXT convolution's input tensor
   * (same layout as xs, borders (xso_bt, xso_bl), first octet xso_o8, xso_terms = the consumer's terms, which must equal `terms`) instead of
   * fp32 NCHW `out` (then unused; plain convolutions only: no residual / post operand, no deconv).  The border and
   * the octets past the output channels are NOT written: the caller keeps the tensor zero there. */
  void* xso;
  int xso_c8, xso_hp, xso_wp, xso_bt, xso_bl, xso_o8, xso_terms;
  /* layout 2: 0 | 1 = every accumulator tile is owned by one consumer wave; 2 = by a PAIR of waves on the same SIMD
   * that take alternate k-steps of every chunk and add their partial sums through LDS before the epilogue (two
   * independent MFMA streams per SIMD: one wave's LDS / barrier stalls are filled by the other) */
  int ksplit;
  /* layout 2: TWO tap sets over one input tile (the ConvGRU's conv*1 + conv*2 pairs, blocks/gru.py:12-33: a 3x3 and a
   * dilated 3x3 convolution of the same tensor whose results are only ever used summed).  dil2 > 0: kh = 2 * kh0 weight
   * rows; rows [0, kh0) are the taps of a kh0 x kw convolution with dilation dil2 (both axes), rows [kh0, kh) those of
   * one with dilation (dil_y, dil_x) >= dil2; both are centred on the same pixel and (pad_t, pad_l) belongs to the
   * larger one.  One launch, one read of the input tile, one accumulator.  0: plain convolution. */
  int dil2;
  /* layout 2: ConvGRU gate epilogues on channel-quad fp32 tensors ("c4": [B][C/4][H*W][4], private to the update
   * block).  Views name c4 tensors here (ctot = their channel count, coff a multiple of 4); `act` is ignored.
   *   0  none
   *   1  out_c4[co] = acc + bias                                                      (no xso)
   *   2  Cout = 3G:  s = acc + bias + res1_c4[co]
   *        co in [0, G):   out_c4[co]     = sigmoid(s + res2_c4[co])                                 (z)
   *        co in [G, 2G):  xso[co - G]    = sigmoid(s + res2_c4[co]) * post_c4[co - G]   (records)   (r * h)
   *        co in [2G, 3G): out_c4[co - G] = s                                                        (q's input stream)
   *   3  Cout = G:   q = tanh(acc + bias + res1_c4[G + co]),  z = res1_c4[co],  h = post_c4[co]:
   *        h' = (1 - z) h + z q  ->  out_c4[co] (may be the post tensor: in place) and xso[co] (records)
   * (reference blocks/gru.py:17-34 with the gate inputs summed by the producer, raft3d.py:92-106). */
  int gate;
} codd_conv_params;

int codd_conv2d(const codd_conv_params* p, void* stream);
/* n <= 4 INDEPENDENT convolutions as one launch (exact-fp32 quad layout, layout = 1, npb = 1, nw = 4, one common mb of
 * 1 | 2): the resolution branches of an mmseg HRModule (configs/models/codd.py:44-74) are launch-bound ~10 us layers
 * that do not depend on each other.  CODD_EUNSUPPORTED when a job does not fit that class (launch them one by one). */
int codd_conv2d_multi(const codd_conv_params* p, int n, void* stream);
/* layout 2 only: CODD_OK if codd_conv2d would accept this launch configuration (tile, chunk depth, wave grid: LDS
 * and register-staging limits of the instantiated kernels), CODD_EUNSUPPORTED otherwise.  Launches nothing and reads
 * no pointer field: callers probe candidate configurations before packing weights for them. */
int codd_conv2d_check(const codd_conv_params* p);

/* quad layout (layout = 1): four input channels innermost, [cog][chunk][tap][c/4][co][4] */
long long codd_conv2d_packed_size_quad(int Cout, int Cin, int kh, int kw, int mb, int ck);
int codd_conv2d_pack_weights_quad(const float* w, float* wpacked, int Cout, int Cin, int kh, int kw, int mb,
                                  int ck, void* stream);

/* Activation re-layout for layout 2: fp32 NCHW views (in0 | in1 concatenated) -> split-bf16 records
 * xs[b][plane hi|lo][octet][yp][xp][8 bf16]; pixel (y, x) sits at (y + bt, x + bl); the border and the channels past
 * C0 + C1 are zero, so that every halo tile of the convolution is in-bounds row segments of 16-byte records (the
 * kernel's LDS-DMA copies them verbatim).  A conv needs bt >= pad_t, bl >= pad_l, 8 * (c8 - xs_o8) >= ceil(Cin / ck) * ck
 * and hp >= max(bt + H, bt - pad_t + (tiles_y * th - 1) * sy + (kh - 1) * dil_y + 1), wp likewise (th x 16*npb = its tile).
 * terms = 1: hi plane only; terms = 16: one plane of fp16 records. */
long long codd_split_bf16_bytes(int B, int c8, int hp, int wp, int terms);
int codd_split_bf16(codd_view in0, int C0, codd_view in1, int C1, int B, int H, int W, int bt, int bl,
                    int c8, int hp, int wp, int terms, void* xs, void* stream);

/* split-bf16 layout (layout = 2): [cog][chunk][plane hi|lo][k-step][g][co][8 bf16]; element (co, ci, tap) is read
 * from w[co*co_stride + ci*ci_stride + tap] and scaled (as codd_conv2d_pack_weights_ex). */
long long codd_conv2d_packed_bytes_bf16(int Cout, int Cin, int kh, int kw, int mb, int ck, int terms);
int codd_conv2d_pack_weights_bf16(const float* w, void* wpacked, int Cout, int Cin, int kh, int kw, int mb, int ck,
                                  int terms, long long co_stride, long long ci_stride, float scale, void* stream);

/* number of floats of the packed weight buffer for (Cout, Cin, kh, kw, mb, ck) */
long long codd_conv2d_packed_size(int Cout, int Cin, int kh, int kw, int mb, int ck);
/* w: [Cout][Cin][kh][kw] device pointer -> wpacked (device).  Runs on `stream`. */
int codd_conv2d_pack_weights(const float* w, float* wpacked, int Cout, int Cin, int kh, int kw,
                             int mb, int ck, void* stream);
/* generalised: element (co, ci, tap) is read from w[co*co_stride + ci*ci_stride + tap] and scaled. */
int codd_conv2d_pack_weights_ex(const float* w, float* wpacked, int Cout, int Cin, int kh, int kw,
                                int mb, int ck, long long co_stride, long long ci_stride, float scale,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * ROLLING-WINDOW convolutions (exact fp32, v_mfma_f32_16x16x4_f32) for HITNet's large-map, few-channel stride-1
 * layers (round 4): one launch runs
 *   mode 0   y = actA(conv3x3_A(x) + bA)
 *   mode 1   y = actB(conv3x3_B(actA(conv3x3_A(x) + bA)) + bB [+ x])      BasicBlock (reference
 *            model/stereo/hitnet/propagation.py:103-121 incl. the enclosing LeakyReLU, :119-121) and the two trailing
 *            3x3 layers of a U-Net merge (backbone.py:31-39)
 *   mode 2   y = actB(conv3x3_B(actA(conv1x1_A(cat[in0, in1]) + bA)) + bB)  head of a merge / PostTileUpdate
 *            (backbone.py:24-30, propagation.py:255-258)
 * with C = 16 | 32 channels out of every stage, "same" zero padding per stage, x = cat[in0, in1] (C0 + C1 = C for
 * modes 0 / 1, <= 64 for mode 2).  A workgroup walks a 64-column strip down `rh` rows; weights stay in registers, the
 * intermediate of a pair stays in LDS (csrc/conv_roll.hip).  Weights are packed per stage by codd_roll_pack_weights
 * (w: [Cout][Cin][k][k] fp32, Cout <= C, output channels past Cout are zero; codd_roll_packed_size floats).
 * Only channels [0, cout_store) of the last stage are written.
 * --------------------------------------------------------------------------------------------- */
typedef struct {
  codd_view in0, in1;
  int C0, C1;
  int B, H, W;
  int C;                 /* 16 | 32 */
  int mode;              /* 0 | 1 | 2 */
  const float* wA;       /* packed weights of stage A (16-byte aligned) */
  const float* bA;       /* bias of stage A or NULL */
  const float* wB;       /* stage B (modes 1, 2) */
  const float* bB;
  int actA, actB;        /* CODD_ACT_* */
  int residual;          /* mode 1: add the chain input before actB */
  float* out;            /* NCHW, channels [out_coff, out_coff + cout_store) of a buffer with out_ctot channels */
  int out_ctot, out_coff, cout_store;
  int rh;                /* output rows per workgroup (launch granularity; any value >= 1 gives the same result) */
} codd_roll_params;
long long codd_roll_packed_size(int C, int k, int Cin);
int codd_roll_pack_weights(const float* w, float* packed, int C, int Cout, int Cin, int k, void* stream);
int codd_conv_roll(const codd_roll_params* p, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Stereo (HITNetMF)
 * --------------------------------------------------------------------------------------------- */
/* Tile cost volume + first arg-min, fused (never materialises cv):
 *   cost[y,x] = min_d sum_c |L[c,y,x] - R~[c,y,4x-d]|, d in [0,D), R~ = 0 outside [0,Wr)
 * replaces calc_init_disp + torch.min (initialization.py:18-45, 167-183).
 * L [B,C,Ht,Wt], R [B,C,Ht,Wr]; cost -> view channel 0 of `cost` buffer, disparity (as float)
 * -> channel 0 of `disp` buffer; dx, dy (channels 1,2 of the hypothesis) are zeroed when
 * zero_dxdy != 0 (initialization.py:192-208). */
int codd_tile_costvol_argmin(const float* L, const float* R, int B, int C, int Ht, int Wt, int Wr, int D,
                             float* cost, int cost_ctot, int cost_coff,
                             float* disp, int disp_ctot, int disp_coff, int zero_dxdy, void* stream);

/* Local slanted-plane correlation (TileWarping.forward, propagation.py:61-86) fused with the
 * ||fea_l||_1 unshuffle (propagation.py:157,207).  For each of `nhyp` (1 or 2) plane sets
 * (d,dx,dy = channels 0..2 of hyp{0,1}) writes 64 channels [fea(16) | cv(k=-1,0,1)(48)] into
 * out{0,1}.  fl, fr: [B,C,4Ht,4Wt]. */
int codd_tile_warp_cost(const float* fl, const float* fr, int B, int C, int Ht, int Wt,
                        codd_view hyp0, codd_view hyp1, int nhyp,
                        float* out0, float* out1, void* stream);

/* Plane up-sampling of a 16-channel hypothesis (propagation.py:10-32): channel 0 =
 * (d + (j-(s-1)/2) dx + (i-(s-1)/2) dy) * scale on an s-times finer grid (s = 2), channels
 * 1..15 nearest.  in [B,16,h,w] view -> out view [B,16,2h,2w]. */
int codd_hyp_upsample(codd_view in, int B, int h, int w, float scale,
                      float* out, int out_ctot, int out_coff, void* stream);

/* Hypothesis selection of TileUpdate (propagation.py:225-240): upd [B,34,h,w] = lastconv output;
 * cur / prev: 16-channel views; out [B,16,h,w] view. */
int codd_hyp_select(const float* upd, codd_view cur, codd_view prev, int B, int h, int w,
                    float* out, int out_ctot, int out_coff, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Motion (Motion + RAFT3D)
 * --------------------------------------------------------------------------------------------- */
/* InstanceNorm2d (affine=False, eps=1e-5) + optional residual + optional ReLU
 * (blocks/extractor.py:9-58,119-199):  y = relu?( (x - mean_c)/sqrt(var_c + eps) [+ res] ).
 * stats: 8-byte aligned scratch of 128*B*C floats (fp64 partial moments, <= 32 parts per plane). */
int codd_instnorm(const float* x, int B, int C, int HW, float* stats, const float* res, int relu,
                  float* y, void* stream);
/* ... with the result written as split-bf16 records (the next convolution's input form, codd_xs_view: image interior
 * only) and, when y != NULL, as fp32 too:  v = norm(x); relu ? max(v,0); res ? (v += res; relu_after_res ? max(v,0)).
 * The second ReLU is the residual block's (blocks/extractor.py:52-58): InstanceNorm -> ReLU -> + x -> ReLU -> re-layout
 * in one launch.  C must be a multiple of 8. */
int codd_instnorm_xs(const float* x, int B, int C, int H, int W, float* stats, const float* res, int relu,
                     int relu_after_res, float* y, codd_xs_view xs, void* stream);

/* All-pairs correlation pyramid (CorrBlock.__init__/corr, blocks/corr.py:28-45,56-62):
 * lvl0[n1,n2] = <f1[:,n1], f2[:,n2]> / 16, lvl_{i+1} = avg_pool2d(lvl_i, 2) over (y2,x2), computed by
 * linearity as <f1, avg_pool^i(f2)> / 16 so the 300 MB level-0 volume is never re-read.
 * f1,f2 [B,D,h,w]; lvl_i [B,h*w,(h>>i)*(w>>i)]. */
int codd_allpairs_corr(const float* f1, const float* f2, int B, int D, int h, int w,
                       float* lvl0, float* lvl1, float* lvl2, float* lvl3, float* scratch, void* stream);
/* floats of `scratch` codd_allpairs_corr needs (packed f1 + pooled f2 levels) */
long long codd_allpairs_corr_scratch(int B, int D, int h, int w);
/* avg_pool2d(kernel 2, stride 2, floor): in [BC,h,w] -> out [BC,h/2,w/2] */
int codd_avgpool2(const float* in, int BC, int h, int w, float* out, void* stream);

/* Pyramid lookup (lietorch_extras.corr_index_forward, call site blocks/corr.py:10-18,47-54):
 * out[b, l*49 + i*7 + j, y, x] = bilinear(lvl_l[b,y,x,:,:], (cx/2^l - 3 + i, cy/2^l - 3 + j)), zero
 * outside.  coords [B,h,w,3] (x,y,*) with stride `cstride` floats per pixel. */
int codd_corr_lookup(const float* lvl0, const float* lvl1, const float* lvl2, const float* lvl3,
                     const float* coords, int cstride, int B, int h, int w, float* out, void* stream);

/* Per-iteration geometry (raft3d.py:225-240; projective_ops.py:11-52; sampler_ops.py:9-28;
 * SE3.log): from T [B,h,w,7], depth1/depth2 [B,h,w] at 1/8 res and K8 = (fx,fy,cx,cy)/8:
 *   xyz    [B,h,w,3] = project(T * inv_project(depth1))
 *   minfo  [B,9,h,w] = clamp([xy - grid, 10*log(T), 10*(bilinear(1/depth2, xy) - xyz.z)], +-50)
 */
int codd_raft_geometry(const float* T, const float* depth1, const float* depth2, int B, int h, int w,
                       float fx, float fy, float cx, float cy, float* xyz, float* minfo, void* stream);

/* Dense SE3 Gauss-Newton step (se3_field.step_inplace, se3_field.py:150-170 =
 * lietorch_extras.se3_build_inplace + damping + cholesky6x6_forward + SE3.exp(dx) * Ts).
 * ae [B,32,h,w] (raw head output; /8 applied inside), delta/weight [B,3,h,w]; target = xyz + delta.
 * T is updated in place.  Hb: scratch of codd_se3_gn_scratch(B,h,w,radius) floats (per-workgroup partial
 * normal equations, combined in a fixed order -> deterministic, + the packed neighbour records). */
long long codd_se3_gn_scratch(int B, int h, int w, int radius);
int codd_se3_gn_step(float* T, const float* ae, int ae_c, const float* xyz, const float* delta,
                     const float* weight, const float* depth1, int B, int h, int w,
                     float fx, float fy, float cx, float cy, int radius, float lm, float ep,
                     float* Hb, void* stream);

/* The same step with the three 1x1 heads of the update block folded into the record packing (raft3d.py:59-61,
 * 100-104: ae = conv1x1(256->32), delta = conv1x1(256->3), weight = sigmoid(conv1x1(256->3))).
 * hidden: the 768 post-ReLU hidden channels [ae | delta | weight groups of 256] in split-bf16 form (written by the
 * fused 3x3 head convolution).  head_w: the 38 x 256 head weights as MFMA A operands,
 *   [32 blocks][plane hi|lo][lane 0..63][8 bf16]  (64 KB, 16-byte aligned): block t*8+s (t = 0, 1; s = 0..7) holds ae
 *   rows 16t + lane%16, channels 32s + 8*(lane/16) .. +8; block 16+s (s = 0..15) holds row lane%16 of
 *   [delta0..2 | weight0..2 | zeros] at hidden channel 256 + 32s + 8*(lane/16) .. +8 (zero where the row's head does
 *   not read that channel group); hi = bf16_rne(w), lo = bf16_rne(w - hi).  head_b [38] fp32 (ae, delta, weight).
 * weight_out [B,3,h,w] receives the confidence weights (up-sampled by the caller after the last update). */
int codd_se3_gn_step_heads(float* T, codd_xs_view hidden, const void* head_w, const float* head_b,
                           const float* xyz, const float* depth1, int B, int h, int w, float fx, float fy,
                           float cx, float cy, int radius, float lm, float ep, float* weight_out, float* Hb,
                           void* stream);

/* Convex 8x up-sampling (se3_field.cvx_upsample, se3_field.py:173-186) of `dim` channels.
 * mode 0: data [B,h,w,dim] -> out [B,8h,8w,dim]             (generic)
 * mode 1: data = SE3 field [B,h,w,7]: out = exp(cvx(log(T)))  (upsample_se3, :189-192)
 * mode 2: data [B,dim,h,w] -> out [B,dim,8h,8w]              (weight, raft3d.py:271-273) */
int codd_cvx_upsample(const float* data, const float* mask, int B, int h, int w, int dim, int mode,
                      float* out, void* stream);
/* upsample_se3 (mode 1) of T [B,h,w,7] and the convex up-sampling (mode 2) of weight [B,3,h,w] with the same mask in one
 * pass (raft3d.py:267-273): T_out [B,8h,8w,7], weight_out [B,3,8h,8w]. */
int codd_cvx_upsample_se3_weight(const float* T, const float* weight, const float* mask, int B, int h, int w,
                                 float* T_out, float* weight_out, void* stream);

/* disparity -> depth (motion.py:154-165): depth = clip(bf / (disp + 1e-5), 0, 210). */
int codd_disp_to_depth(const float* disp, long long n, float bf, float* depth, void* stream);
/* out[b][y][x] = in[b][oy + step*y][ox + step*x], out is [B][ceil((H-oy)/step)][ceil((W-ox)/step)]: RAFT3D's
 * 1/8-resolution depth samples `depth[:, 3::8, 3::8]` (reference model/motion/raft3d/raft3d.py:213-216). */
int codd_subsample(const float* in, int B, int H, int W, int oy, int ox, int step, float* out, void* stream);

/* Forward splat of the previous state into the current frame (Motion.transform_and_project,
 * motion.py:82-130 = pytorch3d PointsRasterizer(K=8) + AlphaCompositor) fused with
 * induced_flow (projective_ops.py:55-68) for the full-resolution call.
 *   T [B,Hs,Ws,7] sampled at (oy + ds*y, ox + ds*x) of a [B,HT,WT,7] field, depth likewise;
 *   feat channels = concat(featA [CA], flow(3, computed when with_flow), featB [CB]);
 *   out [B,C,H,W], zout [B,1,H,W] (nearest z, 0 when empty) or disparity when bf > 0:
 *   disp = bf/(z+1e-5), > W -> 0 (motion.py:190-193).
 * scratch: 16-byte aligned, codd_splat_scratch(B, H, W, radius) ints (per-pixel counters / list offsets, EXACT-SIZE
 * candidate lists -- a point covers at most (2 ceil(R) + 1)^2 pixel centres, R = radius min(H,W) / (2H) px, which
 * bounds them: no candidate is ever dropped -- and one float4 (u, v, z, valid) per source point). */
long long codd_splat_scratch(int B, int H, int W, float radius);
int codd_splat(const float* T, const float* depth, int HT, int WT, int oy, int ox, int ds,
               const float* featA, int CA, const float* featB, int CB, int with_flow,
               int B, int H, int W, float fx, float fy, float cx, float cy, float radius,
               float bf, float* out, float* zout, int* scratch, void* stream);

/* induced_flow (projective_ops.py:55-68): out [B,H,W,3] = project(T*X0) - project(X0). */
int codd_induced_flow(const float* T, const float* depth, int B, int H, int W,
                      float fx, float fy, float cx, float cy, float* out, void* stream);

/* context split (raft3d.py:183-186): net = tanh(x[:, :128]), inp = relu(x[:, 128:512]). */
int codd_context_split(const float* x, int B, int hw, float* net, float* inp, void* stream);

/* SE3.Identity field (raft3d.py:173): T [npix, 7] = (0,0,0, 0,0,0,1). */
int codd_se3_identity(float* T, long long npix, void* stream);

/* bilinear resize (HRNet fuse layers, align_corners = 0; ResizeConcatConv, align_corners = 1):
 * out view += / = resize(in).  accumulate: 0 overwrite, 1 add; relu applied after. */
int codd_resize_bilinear(const float* in, int B, int C, int Hi, int Wi, int Ho, int Wo, int align_corners,
                         float* out, int out_ctot, int out_coff, int accumulate, int relu, void* stream);
/* as codd_resize_bilinear, plus `extra` (contiguous [B, C, Ho, Wo], out_ctot == C, out_coff == 0 expected by the index):
 * out = relu?((out + extra) + blend) when accumulate, (extra + blend) otherwise -- the "+ x_i" term of an mmseg HRModule
 * fuse layer (configs/models/codd.py:44-74) folded into the neighbouring up-sampling term, with the roundings of the two
 * separate launches. */
int codd_resize_bilinear_add(const float* in, int B, int C, int Hi, int Wi, int Ho, int Wo, int align_corners,
                             float* out, int out_ctot, int out_coff, int accumulate, int relu, const float* extra,
                             void* stream);

/* y = relu?(a + b) elementwise (HRNet fuse sums), n floats. */
int codd_add_relu(const float* a, const float* b, long long n, int relu, float* y, void* stream);
/* dst[k][0..n[k]) = src[k][0..n[k]) for k < count <= 8 in ONE launch (the recurrent-state write-back at the end of a
 * captured frame); every n[k] a multiple of 4, every pointer 16-byte aligned. */
int codd_copy_many(const float* const* src, float* const* dst, const long long* n, int count, void* stream);

/* Diagnostics: *slot = the device's constant-rate wall clock (100 MHz ticks) when this one-thread launch runs -- a
 * marker inside a captured frame that does not perturb the replay the way a profiler does (tools/frame_marks.py). */
int codd_timestamp(long long* slot, void* stream);

/* ConvGRU gate fusions (blocks/gru.py:17-34).  t1, t2: the two gate convolutions of a gate pair
 * (3x3 and dilated 3x3, bias included); inp / cor / mot [B,384,hw]: the three input streams.
 *   zr [B,256,hw] = sigmoid(t1 + t2 + (inp+cor+mot)[:, :256]);  rh [B,128,hw] = r * h
 *   hout [B,128,hw] = (1 - z) h + z tanh(t1 + t2 + (inp+cor+mot)[:, 256:]) */
int codd_gru_gate_zr(const float* t1, const float* t2, const float* inp, const float* cor,
                     const float* mot, const float* h, int B, int hw, float* zr, float* rh, void* stream);
int codd_gru_gate_q(const float* t1, const float* t2, const float* inp, const float* cor,
                    const float* mot, const float* zr, const float* h, int B, int hw, float* hout,
                    void* stream);


/* The same gates, writing what the following convolutions read directly in their input form (no fp32 tensor, no
 * re-layout pass between gate and convolution):
 *   zr_xs: z [B,128,hw] fp32 (only z is read again) and r*h as split records into rh_xs (128 channels);
 *   q_xs:  z as written by zr_xs; the new hidden state as fp32 hout AND as split records into h_xs. */
int codd_gru_gate_zr_xs(const float* t1, const float* t2, const float* inp, const float* cor, const float* mot,
                        const float* h, int B, int H, int W, float* z, codd_xs_view rh_xs, void* stream);
int codd_gru_gate_q_xs(const float* t1, const float* t2, const float* inp, const float* cor, const float* mot,
                       const float* z, const float* h, int B, int H, int W, float* hout, codd_xs_view h_xs,
                       void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fusion
 * --------------------------------------------------------------------------------------------- */
/* 1/ds-resolution cues (fusion.py:200-241, 168-198, 243-318), patch = corr_cfg.patch_size (3 or 5; P2 = patch^2),
 * ds = Fusion.ds_scale (even): corr_feat [B, 3 P2 + 4, H/ds, W/ds] =
 * [feat cross (P2) | feat self curr (P2-1) | feat self warp (P2-1) | cost_curr (3) | cost_warp (3)]  (31 at patch 3)
 * and the sub-sampled disparities pc, pw (fuse(), fusion.py:331-342) written to channels
 * (dsub_coff, dsub_coff+1) of dsub. */
int codd_fusion_cues_lr(const float* pred_curr, const float* pred_warp, const float* feat_curr,
                        const float* feat_warp, const float* fea_l, const float* fea_r,
                        int B, int H, int W, int patch, int ds, int CF, int CS, float* corr_feat,
                        float* dsub, int dsub_ctot, int dsub_coff, void* stream);

/* Full-resolution cues (fusion.py:243-318): corr_feat_fr [B, 3 P2 + 5, H, W] = [|disp cross| (P2) |
 * |disp self curr| (P2-1) | |disp self warp| (P2-1) | flow_warp (3) | pred_warp>0 (1) | conf_warp (3)]  (32 at patch 3). */
int codd_fusion_cues_fr(const float* pred_curr, const float* pred_warp, const float* flow_warp,
                        const float* conf_warp, int B, int H, int W, int patch, float* out, void* stream);

/* The full-resolution forget branch in one launch (fusion.py:123-132, 243-318): the cue map of codd_fusion_cues_fr fed
 * through forget_head = Conv1x1(nc->16), Conv3x3(16->8, pad 1), Conv1x1(8->1), Sigmoid without materialising the cue
 * tensor or the 16- / 8-channel maps.  The head is linear up to the sigmoid, so the caller passes it MERGED:
 *   weff = [ W_eff[9][nc] | beta[9] | c0 ],  W_eff[k] = w2.W1[:,:,k].W0,  beta_k = w2.W1[:,:,k].b0,  c0 = w2.b1 + b2
 * (nc = 3 patch^2 + 5; tap k = ky*3 + kx of the 3x3).  A tap contributes W_eff[k].cues(p+k) + beta_k only where p+k lies
 * inside the image (the 3x3's zero padding acts on the biased 16-channel map).  wr [B,1,H,W] = sigmoid(...). */
int codd_fusion_forget(const float* pred_curr, const float* pred_warp, const float* flow_warp,
                       const float* conf_warp, int B, int H, int W, int patch, const float* weff, float* wr,
                       void* stream);

/* Blend (fusion.py:383-394): wf = up4(wf_lr) * (pw>0); wr *= (pw>0);
 * fused = pc*(1-wf*wr) + pw*wf*wr. */
int codd_fusion_blend(const float* pred_curr, const float* pred_warp, const float* wf_lr,
                      const float* wr_logit_sig, int B, int H, int W, int ds,
                      float* fused, float* wf_out, float* wr_out, void* stream);

/* On-device disparity metrics (model/codd.py:456-471; utils/metric.py:9-17,40-54): over the mask
 * lo < gt < hi of the crop [0,h) x [0,w) of [B,1,H,W] maps, adds the frame's EPE, its > thr rate and
 * 1 to meters[0..2] (fp64, AverageMeter semantics) when the mask is non-empty.  No host sync.
 * scratch: 3*128*B doubles. */
int codd_disp_metrics(const float* pred, const float* gt, int B, int H, int W, int h, int w,
                      float lo, float hi, float thr, double* scratch, double* meters, void* stream);

/* Temporal metrics (model/codd.py:473-521; utils/metric.py:19-37; utils/warp.py:69-92): pulls the current
 * (gt, pred, mask) back with the previous frame's GT flow [B,2,H,W] (nearest, zeros) and adds the
 * frame's mean TEPE, (TEPE>3) rate, relative TEPE, (rel>1) rate and 1 to meters[0..4] (when the joint mask
 * is non-empty), mean |flow| and 1 to meters[5..6].  scratch: 6*128*B doubles.  No host sync.  gt_mask (or NULL = gt): the map the current frame's validity mask is computed from -- the reference substitutes a
 * constant in-range map for frames without any disparity ground truth (model/codd.py:478-486); gt2_prev (or NULL):
 * ground-truth second-frame disparity in the previous frame's coordinates, used instead of the flow-warped gt and
 * requiring gt2_prev > 0 (:497-499). */
int codd_tepe_metrics(const float* pred, const float* gt, const float* pred_prev, const float* gt_prev,
                      const float* flow_prev, const float* gt_mask, const float* gt2_prev, int B, int H, int W,
                      int h, int w, float lo, float hi, float bf, double* scratch, double* meters, void* stream);

/* Scene-flow metric columns (model/codd.py:519-575; utils/misc.py:12-36, 62-77): over the crop [0,h) x [0,w) of
 * [B,*,H,W] maps and the mask lo < gt_disp_prev < hi & |gt_flow_prev| < bf & |gt_disp_change| < bf (& gt_flow_occ == 0
 * when given), with est = induced_flow(Ts, clip(bf / pred_prev, 0, bf), K) and est.z * bf (inverse depth ->
 * disparity): meters[0..4] += count, sum of 3-D end-point errors, sum of 2-D ones, #(3-D < 1 px), #(2-D < 1 px) -- the
 * reference's count / epe2d_scene_flow / epe2d_optical_flow / 1px_scene_flow / 1px_optical_flow accumulators.
 * Ts [B,H,W,7]; gt_flow_prev [B,2,H,W]; gt_flow_occ: bytes [B,1,H,W] or NULL.  scratch: 5*128*B doubles. */
int codd_sceneflow_metrics(const float* Ts, const float* pred_prev, const float* gt_disp_prev,
                           const float* gt_flow_prev, const float* gt_disp_change, const unsigned char* gt_flow_occ,
                           int B, int H, int W, int h, int w, float lo, float hi, float bf, float fx, float fy,
                           float cx, float cy, double* scratch, double* meters, void* stream);

/* Input pre-processing (datasets/transforms.py:147-161,373-427; formating.py:65-85): uint8 HWC image
 * (device) -> fp32 CHW RGB, (x - mean)/std, reflect-padded bottom/right to [3,H,W].  mean/stdv: host
 * pointers to 3 floats in RGB order. */
int codd_preprocess(const unsigned char* img, int h, int w, int bgr, const float* mean, const float* stdv,
                    int H, int W, float* out, void* stream);

/* Ablation plug-ins.  codd_fusion_select: mode 0 = KalmanFusion (model/fusion/others.py:124-153; constant
 * gain K = Q/(Q+R), the reference never updates P), mode 1 = GTFusion (:54-86; gt [B,1,hg,wg], zero-padded).
 * cur, warp, out: [B,1,H,W]. */
int codd_fusion_select(int mode, const float* cur, const float* warp, const float* gt, int B, int H,
                       int W, int hg, int wg, float K, float* out, void* stream);
/* GTMotion (model/motion/others.py:17-61): nearest warp of the previous image [B,3,H,W], disparity [B,H,W]
 * and features [B,C,H/4,W/4] by the ground-truth flow [B,2,hg,wg] (zero-padded), disparity minus
 * gt_disp_change [B,1,hg,wg], occluded (gt_flow_occ [B,1,hg,wg] bytes != 0) or out-of-image sources -> 0.
 * Outputs = the 5-entry memory: img_warp [B,3,H,W], feat_warp, conf (ones) [B,3,H,W], disp_warp [B,1,H,W],
 * flow3 = [flow, disp change] [B,3,H,W]. */
int codd_gt_motion(const float* img_prev, const float* disp_prev, const float* feat_prev, int C,
                   const float* gt_flow, const float* gt_disp_change, const unsigned char* gt_flow_occ,
                   int B, int H, int W, int hg, int wg, float* img_warp, float* feat_warp, float* conf,
                   float* disp_warp, float* flow3, void* stream);

/* codd_raft_geometry + codd_corr_lookup in one launch (the lookup workgroups project their own pixels; the
 * level-0 workgroups also write xyz [B,h,w,3] and minfo [B,9,h,w]).  Same results as the two separate calls. */
int codd_raft_geometry_lookup(const float* T, const float* depth1, const float* depth2, const float* lvl0,
                              const float* lvl1, const float* lvl2, const float* lvl3, int B, int h, int w,
                              float fx, float fy, float cx, float cy, float* xyz, float* minfo, float* out,
                              void* stream);
/* The same launch writing its two tensor results -- the 9 motion-info channels and the 196 correlation features --
 * directly as split-bf16 records (the inputs of the flow / correlation encoder convolutions); xyz stays fp32. */
int codd_raft_geometry_lookup_xs(const float* T, const float* depth1, const float* depth2, const float* lvl0,
                                 const float* lvl1, const float* lvl2, const float* lvl3, int B, int h, int w,
                                 float fx, float fy, float cx, float cy, float* xyz, codd_xs_view minfo_xs,
                                 codd_xs_view corr_xs, void* stream);

int codd_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CODD_HIP_H */
