/*
 * spi_hip.h -- C ABI of libspi_hip.so: the MI355X (gfx950) kernels behind SPI's inversion hot path.
 *
 * Every entry point takes raw device pointers + explicit sizes and a HIP stream, never allocates,
 * keeps no global device state, launches asynchronously on `stream`, and returns 0 on success or a
 * negative SPI_ERR_* code (spi_last_error() gives the text).  Tensors are fp32 and densely packed
 * in the stated row-major shape unless an argument says otherwise: the `_t` entry points take a dtype
 * (SPI_DTYPE_F32 / _F16, _F64 where stated) and, for upfirdn2d, strides -- the plugins' own dispatch
 * (bias_act.cpp:81, upfirdn2d.cpp:67, filtered_lrelu.cpp:151) -- and spi_conv_desc.act_dtype makes the
 * convolutions' activation tensors fp16 (the reference's use_fp16 blocks).
 *
 * Each function cites the reference interface it replaces (paths relative to the FeiiYin/SPI tree):
 * the three JIT-built CUDA plugins under eg3d/torch_utils/ops (bias_act.cpp:36, upfirdn2d.cpp:20,
 * filtered_lrelu.cpp:20,217) and the PyTorch-level renderer / conv / optimiser code that has no
 * native counterpart there.  INTEGRATION.md shows the reference-side binding for each.
 */
#ifndef SPI_HIP_H
#define SPI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* spi_stream_t;           /* hipStream_t */

#define SPI_OK               0
#define SPI_ERR_BAD_ARG     -1        /* invalid size / null pointer / unsupported combination */
#define SPI_ERR_UNSUPPORTED -2        /* valid request this build has no kernel for */
#define SPI_ERR_LAUNCH      -3        /* hipGetLastError() != hipSuccess after the launch */

#define SPI_ABI_VERSION 13   /* 2: spi_raymarch_bwd gained d_color_scale, spi_triplane_decode_bwd_sorted gained d_rgb_scale
                             * 3: spi_conv_desc gained workspace / workspace_bytes (Winograd path), spi_conv2d_workspace_bytes
                             * 4: + spi_sample_from_planes_fwd / _bwd (additive)
                             * 5: + contextual / roi_align / adam_pred / filtered_lrelu_fused   6: + spi_affine_fwd / _bwd   7: + spi_decoder_gains (additive)
                             * 8: + spi_bias_act_t / spi_upfirdn2d_t: the plugin entry points with a dtype (fp32 / fp16) and strides (additive)
                             * 9: spi_conv_desc gained out_zeroed (in what was padding after dw_zeroed: 0 = the behaviour of 8), + spi_conv2d_out_accumulates
                             * 10: + spi_affine_multi_fwd / _bwd, spi_modulate_multi_fwd / _bwd (additive)
                             * 11: + spi_conv2d_plan (additive); spi_conv_desc gained act_dtype (fp16 activation tensors; appended: 0 = the behaviour of 10),
                             *     + spi_upfirdn2d_fused_t / spi_tail_bwd_t / spi_chan_dot_t / spi_seg_flags_t / spi_filtered_lrelu_t,
                             *     SPI_DTYPE_F64 for spi_bias_act_t / spi_upfirdn2d_t (additive)
                             * 12: + spi_tail_bwd_dot_t (additive); spi_conv2d_workspace_bytes / workspace of fp16 activation tensors select the direct fp16 kernels
                             * 13: + spi_conv_wino_f4_set (additive); the Winograd workspace of a >= 256^2 layer is 2.25x larger (36 instead of 16 frequencies) */
int         spi_abi_version(void);
int         spi_sizeof_conv_desc(void);   /* sizeof(spi_conv_desc) of THIS build: bindings assert it against their own struct */
const char* spi_last_error(void);     /* thread-local, valid until the next failing call */

/* activation ids: same numbering as bias_act.py:22-32 `cuda_idx` */
enum { SPI_ACT_LINEAR = 1, SPI_ACT_RELU, SPI_ACT_LRELU, SPI_ACT_TANH, SPI_ACT_SIGMOID, SPI_ACT_ELU,
       SPI_ACT_SELU, SPI_ACT_SOFTPLUS, SPI_ACT_SWISH };

/* ------------------------------------------------------------------------------------------------
 * Volumetric renderer (eg3d/training/volumetric_rendering/*.py -- pure PyTorch in the reference)
 * ---------------------------------------------------------------------------------------------- */

/* RaySampler.forward, ray_sampler.py:24-63.  cam2world [N,16], intrinsics [N,9] -> ray_o, ray_d [N,res*res,3]. */
int spi_ray_sampler(const float* cam2world, const float* intrinsics, int N, int res,
                    float* ray_o, float* ray_d, spi_stream_t stream);

/* sample_stratified scalar branch, renderer.py:188-190: t = start + (k + xi) * (end-start)/(S-1).
 * xi, depths: [n_rays, S]. */
int spi_coarse_depths(const float* xi, int64_t n_rays, int S, float ray_start, float ray_end,
                      float* depths, spi_stream_t stream);

/* planes [NP, C, H, W] <-> channels-last [NP, H, W, C] (one 128-B line per texel when C = 32). */
int spi_nchw_to_nhwc(const float* src, float* dst, int NP, int C, int H, int W, spi_stream_t stream);
int spi_nhwc_to_nchw(const float* src, float* dst, int NP, int C, int H, int W, spi_stream_t stream);

/* sample_from_planes on its own (renderer.py:55-65; replaces the three F.grid_sample calls of :62-64 for a decoder that is not
 * the OSG MLP -- ImportanceRenderer takes any decoder callable, renderer.py:88,142-148):
 *   planes_nhwc [N,3,H,W,32], coords [N,P,3] -> out [N,3,P,32], plane k sampled at (x,y), (x,z), (z,x) of 2/box_warp * coords,
 *   bilinear, zeros padding, align_corners = False.  _bwd ACCUMULATES d_out [N,3,P,32] into d_planes_nhwc (pre-zeroed by the
 *   caller) with global atomics.  Not on the SPI hot path (the fused kernels below are). */
int spi_sample_from_planes_fwd(const float* planes_nhwc, const float* coords, int N, int64_t P, int H, int W, float box_warp, float* out,
                               spi_stream_t stream);
int spi_sample_from_planes_bwd(const float* d_out, const float* coords, int N, int64_t P, int H, int W, float box_warp, float* d_planes_nhwc,
                               spi_stream_t stream);

/* sample_from_planes + OSGDecoder.forward fused: renderer.py:55-65 + triplane.py:123-135.
 *   planes_nhwc [N,3,H,W,32];  points are either explicit `coords` [N,P,3] (ray_o = NULL) or
 *   ray_o/ray_d [N,M,3] + depths [N,M,S] with P = M*S.
 *   w1t [32,64] (= (W1 * weight_gain)^T), b1 [64], w2 [33,64], b2 [33]: decoder weights ALREADY
 *   multiplied by their weight_gain / bias_gain (networks_stylegan2.py:115-120).
 *   out: rgb / sigma rows.  Plain mode (out_S = 0): rgb [N,P,32], sigma [N,P].  Rays mode with
 *   out_S > 0: the point (ray r, sample k) goes to row r*out_S + out_off + k, so the coarse and
 *   fine passes fill one [R, Sc+Sf, .] buffer (the torch.cat of renderer.py:158-160 never happens).
 *   rgb may be NULL: only the densities are evaluated (depth-only rendering). */
int spi_triplane_decode_fwd(const float* planes_nhwc, const float* coords, const float* ray_o,
                            const float* ray_d, const float* depths, const float* w1t, const float* b1,
                            const float* w2, const float* b2, int N, int64_t P, int S, int H, int W,
                            float box_warp, int out_S, int out_off, float* rgb, float* sigma,
                            spi_stream_t stream);

/* Backward of the above.  d_planes_nhwc [N,3,H,W,32] is ACCUMULATED into (zero it first);
 * if dump_act != NULL (decoder weights need grads) the kernel writes per-point rows
 * [f(32) | h(64) | d_pre1(64) | d_y(33) | pad(7)] = 200 floats so the caller can form
 * dW1 = d_pre1^T f, dW2 = d_y^T h with a plain GEMM.  (Actual dump layout is column-major:
 * [193][N*P] with rows f 0..31, h 32..95, d_pre1 96..159, d_y 160..192.)
 * d_rgb / d_sigma use the same (out_S, out_off) row mapping as the forward outputs. */
int spi_triplane_decode_bwd(const float* planes_nhwc, const float* coords, const float* ray_o,
                            const float* ray_d, const float* depths, const float* w1t, const float* b1,
                            const float* w2, const float* b2, const float* d_rgb, const float* d_sigma,
                            int N, int64_t P, int S, int H, int W, float box_warp, int out_S, int out_off,
                            float* d_planes_nhwc, float* dump_act, spi_stream_t stream);

/* Same backward for the render path, tiled for scatter locality: the 64 x S points of every 8x8 patch of
 * neighbouring rays (ray m = row * ray_w + col) are ordered by DEPTH (uniform depth bins, no sort: each ray's
 * samples are already ascending) and visited 256 at a time -- thin slabs of the view frustum whose plane-gradient
 * contributions are pre-summed in a 16x16-texel LDS window per plane before touching HBM.  Sample (r, k) has depth
 * depths_sorted[r, k] and colour/density/gradient row r*S + perm[r, k] (perm may be NULL = identity).
 * The decoder runs on the fp32 matrix cores and its weight gradients are fused in: with dw1 != NULL the call
 * OVERWRITES dw1 [64,32], db1 [64], dw2 [33,64], db2 [33] (gradients wrt the gained weights w1t^T, b1, w2, b2);
 * dw1 == NULL (all four) = decoder frozen.  d_rgb == NULL = the colour gradient is zero (depth-only loss): the colour
 * layer is skipped.  d_rgb_scale == NULL: d_rgb is the per-sample gradient [N*M*S,32] (row mapping as above).  d_rgb_scale
 * != NULL (what spi_raymarch_bwd's d_color_scale output is for): d_rgb is PER RAY [N*M,32] and the gradient of sample row i
 * of ray r is d_rgb[r,:] * d_rgb_scale[i] -- the [N*M*S,32] tensor never exists.  colors (required with d_rgb): the forward's colour rows
 * [N*M*S,32] as spi_triplane_decode_fwd wrote them; the colour layer's sigmoid is read back ((c + 0.001) / 1.002) instead of
 * recomputed.  ray_active (optional, int32 [N*M] from
 * spi_raymarch_bwd): rays flagged 0 are dropped before the tiles are formed.  `workspace` must hold spi_triplane_decode_bwd_sorted_ws(...) floats
 * (weight fragments + per-wave partial sums; contents undefined afterwards).  d_planes_nhwc is accumulated into. */
int spi_triplane_decode_bwd_sorted(const float* planes_nhwc, const float* ray_o, const float* ray_d,
                                   const float* depths_sorted, const int32_t* perm, const float* w1t,
                                   const float* b1, const float* w2, const float* b2, const float* d_rgb,
                                   const float* d_rgb_scale, const float* colors, const float* d_sigma, int N, int M, int S,
                                   int ray_w, int H, int W,
                                   float box_warp, float* d_planes_nhwc, float* workspace, float* dw1, float* db1,
                                   float* dw2, float* db2, const int32_t* ray_active, spi_stream_t stream);
int64_t spi_triplane_decode_bwd_sorted_ws(int N, int M, int S, int ray_w);

/* Decoder weight gradients from a dump written by spi_triplane_decode_bwd (rows f | h | d_pre1 | d_y, each
 * `cols` long): dw1 [64,32], db1 [64], dw2 [33,64], db2 [33] wrt the gained weights (overwritten). */
int spi_decoder_wgrad(const float* dump, int64_t cols, float* dw1, float* db1, float* dw2, float* db2,
                      spi_stream_t stream);

/* min / max over a depth tensor (ray_marcher.py:50 clamps to the GLOBAL range).  out[2] = {min,max}. */
int spi_minmax(const float* x, int64_t n, float* out2, spi_stream_t stream);


/* The OSG decoder's parameters (FC 32 -> 64: w1 [64,32], b1 [64]; FC 64 -> 33: w2 [33,64], b2 [33]) times their FullyConnectedLayer gains
 * (networks_stylegan2.py:114-127) in one launch.  transpose_w1 = 1 writes w1 as [32,64] (the operand layout of the decode kernels),
 * 0 keeps [64,32] (scaling the gradients on the way back). */
int spi_decoder_gains(const float* w1, const float* b1, const float* w2, const float* b2, float g_w1, float g_b1, float g_w2, float g_b2,
                      float* o_w1, float* o_b1, float* o_w2, float* o_b2, int transpose_w1, spi_stream_t stream);
/* MipRayMarcher2.run_forward, ray_marcher.py:25-57.  One launch = R rays of S sorted samples.
 *   colors [R,S_store,C] (C = 32), densities [R,S_store] with S <= S_store <= 256 rows kept per ray;
 *   depths [R,S] sorted; perm (optional, int32 [R,S]): sample k of ray r is row perm[r,k] of
 *   colors/densities (identity if NULL) -- this folds unify_samples' three gathers
 *   (renderer.py:157-167) into the march.
 *   clamp2 -> {min,max} from spi_minmax.  rgb [R,C] may be NULL (coarse pass wants weights only).
 *   out: rgb [R,C] (scaled to [-1,1]), depth [R], weights [R,S-1], wsum [R] (any may be NULL). */
int spi_raymarch_fwd(const float* colors, const float* densities, const float* depths,
                     const int32_t* perm, const float* clamp2, int64_t R, int S, int S_store, int C,
                     int white_back, float* rgb, float* depth, float* weights, float* wsum,
                     spi_stream_t stream);

/* Backward: d_rgb [R,C] (NULL = 0: only the depth map is differentiated -- colors / d_colors are then not touched
 * and may be NULL), d_depth [R] (NULL = 0), d_weights [R,S-1] (NULL = 0) ->
 * d_colors [R,S,C] and / or d_color_scale [R,S], d_densities [R,S], written through perm like the forward reads.
 * The colour-row gradient is d_rgb[r,:] * (w[r,k-1] + w[r,k]): d_color_scale receives that per-sample scalar alone
 * (4 B instead of 128 B per sample; spi_triplane_decode_bwd_sorted rebuilds the rows from d_rgb).  Either output may be NULL.
 * ray_active (optional, int32 [R]): set to 0 for rays whose incoming gradient is exactly zero -- their rows of d_colors /
 * d_color_scale / d_densities are then NOT written (all zero by definition); pass the same array to
 * spi_triplane_decode_bwd_sorted. */
int spi_raymarch_bwd(const float* colors, const float* densities, const float* depths,
                     const int32_t* perm, const float* clamp2, const float* d_rgb, const float* d_depth,
                     const float* d_weights, int64_t R, int S, int S_store, int C, int white_back,
                     float* d_colors, float* d_color_scale, float* d_densities, int32_t* ray_active, spi_stream_t stream);

/* sample_importance + sample_pdf, renderer.py:194-253.  depths [R,S], weights [R,S-1], u [R,Sf]
 * -> fine depths [R,Sf]: in draw order like the reference (sort_out = 0) or ascending per ray (sort_out = 1;
 * same multiset -- what the renderer uses so the later merge reads two monotone streams). */
int spi_importance_sample(const float* depths, const float* weights, const float* u, int64_t R, int S,
                          int Sf, float* fine, int sort_out, spi_stream_t stream);

/* unify_samples' sort, renderer.py:157-163: concat coarse [R,Sc] + fine [R,Sf], ascending stable
 * sort.  out: sorted depths [R,Sc+Sf], perm int32 [R,Sc+Sf] (index into the concatenation). */
int spi_merge_sort_depths(const float* coarse, const float* fine, int64_t R, int Sc, int Sf,
                          float* sorted, int32_t* perm, spi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * StyleGAN2 operator layer (eg3d/torch_utils/ops/*.cu)
 * ---------------------------------------------------------------------------------------------- */

/* bias_act.cpp:36 `bias_act(x,b,xref,yref,dy,grad,dim,act,alpha,gain,clamp)`; kernel bias_act.cu:27-151.
 *   grad = 0: y = clamp(act(x + b) * gain)        (x = input)
 *   grad = 1: dx = dy * d/dx                        (x = dy; xref/yref = saved input/output)
 *   grad = 2: second-order term                     (x = d_dx; dy = first-order dy)
 * b (len sizeB, may be NULL) is indexed (i / stepB) % sizeB.  clamp < 0 disables clamping. */
int spi_bias_act(const float* x, const float* b, const float* xref, const float* yref, const float* dy,
                 float* y, int64_t n, int sizeB, int64_t stepB, int grad, int act, float alpha,
                 float gain, float clamp, spi_stream_t stream);

/* Backward of a layer tail  y = clamp(act(z + noise*strength + bias) * gain)  (SynthesisLayer.forward,
 * networks_stylegan2.py:320-329) in ONE pass over dy [N,C,HW]:
 *   dz      = dy * d y/d z                (act in {linear, relu, lrelu}; y = saved output; y = NULL: dz = dy, dz may be NULL)
 *   d_bias[c]   += sum_{n,hw} dz           (NULL to skip; CALLER zeroes)
 *   d_pixsum[hw] += sum_{n,c} dz           (NULL to skip; CALLER zeroes; d_noise = d_pixsum * strength)
 *   d_strength[0] += sum_hw d_pixsum[hw] * noise[hw]   (noise [HW] and d_strength together, NULL to skip; needs d_pixsum; CALLER zeroes)
 * replaces bias_act(grad=1) + two full-tensor torch reductions of the reference's autograd graph. */
int spi_tail_bwd(const float* dy, const float* y, float* dz, float* d_bias, float* d_pixsum, const float* noise,
                 float* d_strength, int N, int C, int64_t HW, int act, float alpha, float gain, float clamp,
                 spi_stream_t stream);

/* out[r] += sum_p a[r,p] * b'[r,p] for r < rows (= N*C), p < HW  (out: CALLER zeroes).  With act != 0, b is a layer
 * OUTPUT y = clamp(act(z + noise*noise_gain + bias)*gain) and b' is the reconstructed conv result z (act in {linear,
 * lrelu}; clamped pixels carry a zero gradient in `a`, so their irrecoverable z does not matter); act == 0: b' = b.
 * Used for the style gradient of a modulated conv whose weights are frozen (SPI stage 1):
 *   d s_i = <x_i, dx_i> / s_i  -  s_i * gain^2 * sum_o d_o^2 <dz_o, z_o> sum_t W[o,i,t]^2
 * -- two per-channel dot products and a GEMV instead of the O*I*k*k weight-gradient GEMM. */
int spi_chan_dot(const float* a, const float* b, float* out, int64_t rows, int C, int64_t HW, const float* bias,
                 const float* noise, const float* noise_gain, int act, float alpha, float gain, spi_stream_t stream);

/* flags[n, s] = 1 if any x[n, c, 16 s .. 16 s + 15] != 0 over all channels c (flat pixel index over H*W), else 0;
 * flags holds N * ceil(HW / 16) entries.  Feeds spi_conv_desc.dy_seg_flags. */
#define SPI_SEG_PIXELS 16
int spi_seg_flags(const float* x, int32_t* flags, int N, int C, int64_t HW, spi_stream_t stream);

/* Last step of the frozen-weight style gradient (see spi_chan_dot):
 *   ds[s,i] = A_i / st[s,i]  -  st[s,i] * gain^2 * sum_o dcoef[s,o]^2 * C_o * ww[o,i],     A_i = sum_n a[n,i],  C_o = sum_n cv[n,o]
 * a [N,I] = <x_i, dx_i>, cv [N,O] = <dz_o, z_o> (NULL without demodulation: second term dropped), st [NS,I] styles (NS = N, or 1
 * = one style row shared by the batch: then the sums run over n), dcoef [NS,O], ww [O,I] = sum_t W[o,i,t]^2.  |st| <= 1e-20 gives 0
 * for the first term.  One launch instead of ~15 elementwise / reduction / GEMV launches per layer. */
int spi_style_grad(const float* a, const float* cv, const float* st, const float* dcoef, const float* ww, float* ds, int N, int NS,
                   int I, int O, float style_gain, spi_stream_t stream);

/* upfirdn2d.cpp:20 `upfirdn2d(x,f,upx,upy,downx,downy,padx0,padx1,pady0,pady1,flip,gain)`.
 *   x [N,C,inH,inW] (dense NCHW), f [fH,fW]; y [N,C,outH,outW] with the reference's output-size rule.
 * Optional fused epilogue (NULL / act = 0 disables): y = bias_act(y + noise[outH,outW]*noise_gain[0], bias[C]). */
int spi_upfirdn2d(const float* x, const float* f, float* y, int N, int C, int inH, int inW, int fH,
                  int fW, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0,
                  int pady1, int flip, float gain, int outH, int outW,
                  const float* noise, const float* noise_gain, const float* bias, int act, float alpha,
                  float act_gain, float clamp, spi_stream_t stream);

/* The same two plugin entry points with the tensor type as an argument, as the reference instantiates them (bias_act.cpp:81 and
 * upfirdn2d.cpp:67 dispatch over half / float / double; its fp16 super-resolution blocks call them with half tensors in channels_last
 * layout, networks_stylegan2.py:423-436).  dtype: SPI_DTYPE_F32 or SPI_DTYPE_F16 (every tensor argument of the call has that type, the
 * FIR filter `f` stays float32 as in upfirdn2d.cpp:27); half tensors are computed in fp32 and rounded once at the store, like
 * InternalType<half> in bias_act.cu:14-16.  Anything else returns SPI_ERR_UNSUPPORTED (-2): the caller falls back to its own path, the
 * contract filtered_lrelu's rc = -1 has in the reference.
 * spi_upfirdn2d_t takes element strides {n, c, h, w} of x and y (NULL = dense NCHW): dense NCHW and channels_last are accepted, in any
 * combination, like the plugin's "non-overlapping and dense" rule (upfirdn2d.cpp:23).  No fused epilogue on this entry point. */
enum { SPI_DTYPE_F32 = 0, SPI_DTYPE_F16 = 1, SPI_DTYPE_F64 = 2 };   /* F64: spi_bias_act_t / spi_upfirdn2d_t only (the plugins' double instantiation) */
int spi_bias_act_t(const void* x, const void* b, const void* xref, const void* yref, const void* dy, void* y, int64_t n, int sizeB,
                   int64_t stepB, int grad, int act, float alpha, float gain, float clamp, int dtype, spi_stream_t stream);
int spi_upfirdn2d_t(const void* x, const float* f, void* y, int N, int C, int inH, int inW, const int64_t* x_strides,
                    const int64_t* y_strides, int fH, int fW, int upx, int upy, int downx, int downy, int padx0, int padx1, int pady0,
                    int pady1, int flip, float gain, int outH, int outW, int dtype, spi_stream_t stream);

/* fp16 ACTIVATION TENSORS through a whole use_fp16 block (ABI 11; networks_stylegan2.py:421-436: x = x.to(float16), every conv / FIR / bias_act of
 * the block reads and writes half tensors).  `dtype` is the element type of the activation arguments (void*); bias, noise, per-channel and
 * per-pixel sums and every accumulator stay fp32; a value is rounded to fp16 once, when it is stored.  dtype = SPI_DTYPE_F32 is the untyped
 * entry point of the same name.  spi_conv_desc.act_dtype does the same for the convolutions.
 *   spi_upfirdn2d_fused_t  spi_upfirdn2d with its fused layer tail (fp16: the 4x4 filter, up = down = 1, images >= 100 px -- the FIR after a
 *                          stride-2 transposed conv and its adjoint; other shapes: spi_upfirdn2d_t + spi_bias_act_t, SPI_ERR_UNSUPPORTED here)
 *   spi_tail_bwd_t         spi_tail_bwd on fp16 dy / y / dz          spi_chan_dot_t   spi_chan_dot on fp16 a / b
 *   spi_seg_flags_t        spi_seg_flags of an fp16 gradient */
int spi_upfirdn2d_fused_t(const void* x, const float* f, void* y, int N, int C, int inH, int inW, int fH, int fW, int upx, int upy,
                          int downx, int downy, int padx0, int padx1, int pady0, int pady1, int flip, float gain, int outH, int outW,
                          const float* noise, const float* noise_gain, const float* bias, int act, float alpha, float act_gain,
                          float clamp, int dtype, spi_stream_t stream);
int spi_tail_bwd_t(const void* dy, const void* y, void* dz, float* d_bias, float* d_pixsum, const float* noise, float* d_strength, int N, int C,
                   int64_t HW, int act, float alpha, float gain, float clamp, int dtype, spi_stream_t stream);
/* spi_tail_bwd_t plus, from the same pass, zdot[n*C + c] += sum_hw dz[n,c,hw] * z[n,c,hw] (zdot [N*C]: CALLER zeroes), z = the conv result reconstructed
 * from the saved output y exactly as spi_chan_dot(act != 0) reconstructs its b' (inverse activation / gain, minus z_bias[c] and z_noise[hw] *
 * z_noise_gain[0]; NULL = none): the second of the two dot products of the frozen-weight style gradient (networks_stylegan2._ModConvFrozen;
 * the reference's autograd computes the full weight gradient for it, networks_stylegan2.py:71-88) without re-reading dz and y (ABI 12). */
int spi_tail_bwd_dot_t(const void* dy, const void* y, void* dz, float* d_bias, float* d_pixsum, const float* noise, float* d_strength, int N, int C,
                       int64_t HW, int act, float alpha, float gain, float clamp, const float* z_bias, const float* z_noise, const float* z_noise_gain,
                       float* zdot, int dtype, spi_stream_t stream);
int spi_chan_dot_t(const void* a, const void* b, float* out, int64_t rows, int C, int64_t HW, const float* bias, const float* noise,
                   const float* noise_gain, int act, float alpha, float gain, int dtype, spi_stream_t stream);
int spi_seg_flags_t(const void* x, int32_t* flags, int N, int C, int64_t HW, int dtype, spi_stream_t stream);
/* spi_filtered_lrelu on fp16 tensors (filtered_lrelu.cpp:151,265: the plugin is dispatched for half too): x [N,C,inH,inW], b [C] and y are
 * `dtype` tensors, the filters and the intermediate `tmp` (N*C*midH*midW floats) stay fp32 = the plugin's internal type for half. */
int spi_filtered_lrelu_t(const void* x, const float* fu, const float* fd, const void* b, float* tmp, void* y, int N, int C,
                         int inH, int inW, int fuH, int fuW, int fdH, int fdW, int up, int down, int px0, int px1, int py0, int py1,
                         float gain, float slope, float clamp, int flip, int outH, int outW, int dtype, spi_stream_t stream);

/* filtered_lrelu.cpp:20 `filtered_lrelu(x,fu,fd,b,si,up,down,px0,px1,py0,py1,sx,sy,gain,slope,clamp,flip,writeSigns)`
 * forward without sign tensors: bias -> up-FIR(gain up^2) -> lrelu*gain, clamp -> down-FIR.
 * tmp must hold N*C*midH*midW floats (mid = upsampled+filtered size). */
int spi_filtered_lrelu(const float* x, const float* fu, const float* fd, const float* b, float* tmp,
                       float* y, int N, int C, int inH, int inW, int fuH, int fuW, int fdH, int fdW,
                       int up, int down, int px0, int px1, int py0, int py1, float gain, float slope,
                       float clamp, int flip, int outH, int outW, spi_stream_t stream);

/* The same operator in ONE launch (what filtered_lrelu.cu:119-1105 does: up-FIR, activation and down-FIR fused through shared memory; no
 * upsampled tensor in memory, no `tmp`).  mode 0: no sign tensor; mode 1: also WRITE the reference's bit-packed signs of the upsampled
 * samples (uint8 [N*C, sH, sW/4], 2 bits per sample: 1 = negative, 2 = clamped; sH >= midH, sW >= midW, sW % 4 == 0 -- the reference rounds
 * sW to 16); mode 2 (gradient pass, filtered_lrelu.py:246-262): READ them at (col + sx, row + sy) instead of applying lrelu / clamp.
 * N*C <= 65535; filters up to 256 taps whose windows fit 48 KB of LDS (12-tap filters at up = down = 2 use 31 KB). */
int spi_filtered_lrelu_fused(const float* x, const float* fu, const float* fd, const float* b, uint8_t* signs, float* y, int N,
                             int C, int inH, int inW, int fuH, int fuW, int fdH, int fdW, int up, int down, int px0, int px1,
                             int py0, int py1, float gain, float slope, float clamp, int flip, int mode, int sH, int sW, int sx,
                             int sy, int outH, int outW, spi_stream_t stream);

/* filtered_lrelu.cpp:217 `filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns) -> so`: the activation stage alone, IN PLACE on the
 * upsampled tensor x [NC, xH, xW], with the reference's bit-packed sign tensor (uint8 [NC, sH, sW/4]: 2 bits per element, 1 = negative,
 * 2 = clamped; sW a multiple of 4 -- the reference rounds it to 16).  mode 0: x = clamp(lrelu(x * gain)); mode 1: the same and WRITE the
 * signs; mode 2 (gradient pass): x = x * gain * (slope where the sign read at (col + sx, row + sy) says negative, 0 where clamped, 1 outside
 * the sign tensor).  clamp < 0 = none.  The caller owns both tensors (the reference allocates `so` inside, filtered_lrelu.cpp:246). */
int spi_filtered_lrelu_act(float* x, uint8_t* signs, int64_t NC, int xH, int xW, int sH, int sW, int sx, int sy,
                           float gain, float slope, float clamp, int mode, spi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Dense convolutions on the matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 * Replaces conv2d_gradfix.conv2d / conv_transpose2d with groups = batch as modulated_conv2d uses
 * them (networks_stylegan2.py:85-88, conv2d_resample.py:114-136) and the plain convs of the VGG losses.
 * ---------------------------------------------------------------------------------------------- */

typedef struct spi_conv_desc {
    int N, I, O;              /* batch, in / out channels                                              */
    int H, W;                 /* input spatial size                                                    */
    int kh, kw;               /* kernel size (1 or 3)                                                  */
    int pad;                  /* zero padding (stride-1 mode)                                          */
    int transposed;           /* 0: correlation, stride 1; 1: conv_transpose2d stride 2, padding 0     */
    int flip;                 /* 1: spatially flip the kernel (true convolution)                       */
    int w_tap_major;          /* 0: weights [O, I, kh, kw] (PyTorch);  1: [O, kh, kw, I] (channels innermost:  */
                              /*    contiguous slab loads and contiguous weight-gradient writes)        */
    int compute_f16;          /* arithmetic of the three passes; tensors stay fp32 in memory, accumulation is fp32:     */
                              /*  0: exact fp32 MFMA (v_mfma_f32_32x32x2_f32)                                            */
                              /*  1: operands rounded to fp16, fp16 MFMA (the reference's use_fp16 super-resolution     */
                              /*     blocks)                                                                             */
                              /*  2 / 3: "split bf16": each fp32 operand is cut into 2 / 3 bf16 pieces when it is staged */
                              /*     in LDS and the 3 / 6 significant piece products run on v_mfma_f32_32x32x16_bf16     */
                              /*     (error ~2^-16 / ~2^-23 per product; bf16 keeps the fp32 exponent range)             */
                              /*  modes 1-3 need whole 16-channel chunks (I % 16 == 0) and a grid that does not need     */
                              /*  split-K; otherwise the exact fp32 kernels run                                          */
    int64_t w_batch_stride;   /* elements between per-sample weights; 0 = weights shared by the batch  */
    /* fused epilogue of the forward (all optional): y = clamp(act(acc + noise*noise_gain + bias)*gain) */
    const float* bias;        /* [O] */
    const float* noise;       /* [OH,OW] */
    const float* noise_gain;  /* [1] device scalar */
    int act; float alpha, gain, clamp;
    /* backward passes only (optional, NULL = dense): spi_seg_flags() of the output gradient `dy`.  Masked losses leave whole
     * regions of dy exactly zero; dgrad then skips output tiles whose receptive field holds no flagged segment (writing
     * zeros) and wgrad reduces over the flagged 16-pixel slabs only -- the results are those of the dense kernels up to
     * the order of the fp32 sums. */
    const int32_t* dy_seg_flags;
    /* forward only (optional, NULL = dense): spi_seg_flags()-style map over the OUTPUT pixels [N, ceil(OH*OW/16)]: output tiles
     * that hold no flagged segment are not computed (zeros are written).  For consumers that read the output only inside a
     * known region (masked losses): flag that region, dilated by what the layers in between need. */
    const int32_t* out_seg_flags;
    int dw_zeroed;            /* spi_conv2d_wgrad only: 1 = the caller already zeroed dw (saves the memset launch when dw is a slice of
                               * a buffer that was cleared together with other small gradients) */
    int out_zeroed;           /* spi_conv2d_fwd / _dgrad only: 1 = the caller already zeroed the output.  Matters for the launches that ACCUMULATE
                               * into it (split-K implicit GEMM of the 4^2..32^2 layers, channel-split Winograd: spi_conv2d_out_accumulates says
                               * which): they skip their own fill launch -- a caller that clears all accumulators of an iteration with one fill
                               * saves ~40 launches per generator pass.  Ignored by the launches that overwrite. */
    /* optional scratch memory (device, 16-byte aligned).  With at least spi_conv2d_workspace_bytes(d, pass) bytes the 3x3 / stride-1 /
     * pad-1 forward and data-gradient passes of large layers run as Winograd F(2x2, 3x3) and the weight-gradient pass as F(3x3, 2x2) -- fp32
     * operands, fp32 accumulation, 2.25x fewer MFMAs; the result differs from the direct sum by a few fp32 roundings (what cuDNN runs for
     * the reference's fp32 3x3 convs).  The weight-gradient pass needs no scratch: its spi_conv2d_workspace_bytes is a nominal 16.
     * NULL / too small: the implicit-GEMM kernels run.  The workspace holds the transformed weights of THIS call only.  Offered for
     * compute_f16 = 0 and 3 (the 6-product split asks for fp32-equivalent products, which fp32 Winograd delivers faster on these layers).
     * fp16 activation tensors (act_dtype = SPI_DTYPE_F16, compute_f16 = 1): the same opt-in selects the DIRECT fp16 kernels for 3x3 / stride-1 / pad-1
     * layers -- forward / dgrad (output channels a multiple of 128, reduction channels a multiple of 16, at least 128 tiles of 16 x 32 pixels): the
     * workspace receives the weights converted to fp16 in their LDS layout (`workspace_ready` applies); the forward of a stride-2 TRANSPOSED 3x3
     * conv (output channels a multiple of 128, input channels of 32) and its data gradient (a stride-2 conv of the gradient; input channels a multiple
     * of 128) likewise; weight gradient (128 | O, 64 | I, 32 | W,
     * 4 | H, no dy_seg_flags): the workspace receives one partial sum per workgroup and dw is OVERWRITTEN with their sum (deterministic); a smaller
     * non-null workspace still selects the kernel, which then adds into the zeroed dw with fp32 atomics.  Same products and fp32 accumulation as the
     * implicit GEMM, another summation order. */
    void* workspace;
    int64_t workspace_bytes;
    /* element type of the ACTIVATION tensors of all three passes (x, y, dy, dx): SPI_DTYPE_F32 (0, default) or SPI_DTYPE_F16 -- fp16 only
     * together with compute_f16 = 1 and channel counts that are multiples of 16: the reference's use_fp16 blocks keep their activations
     * in half precision in memory (networks_stylegan2.py:421-436; conv2d on half tensors accumulates in fp32 and rounds once).  Weights, weight
     * gradients, bias and noise stay fp32. */
    int act_dtype;
    /* forward / dgrad Winograd (and direct fp16) passes only: 1 = `workspace` still holds the transformed weights an EARLIER call of the same pass wrote for the same
     * `w` (same layout flags) -- the weight-transform launch is skipped.  For frozen weights that are convolved again and again (the VGG feature
     * extractors of the losses: 36 such launches per stage-2 iteration); the caller keeps one workspace per (weight tensor, pass) alive. */
    int workspace_ready;
} spi_conv_desc;
/* weight layout: [O, I, kh, kw] (or [O, kh, kw, I] with w_tap_major) in both modes
 * (transposed: out[o,2y+ky,2x+kx] += x[i,y,x] * w[o,i,ky,kx]).
 * output size: stride-1: H + 2*pad - kh + 1;  transposed: 2*H + kh - 2  (= 2H+1 for 3x3).          */
/* bytes of workspace with which pass (0 forward, 1 dgrad, 2 wgrad) takes its Winograd path (fp16 activation tensors: its direct fp16 kernel); 0 = the
 * pass has none for this shape */
int64_t spi_conv2d_workspace_bytes(const spi_conv_desc* d, int pass);
/* Minimal-filtering tile of the fp32 Winograd forward / dgrad path (process-wide, host logic only; call it BEFORE sizing workspaces: the size differs).
 * 1 (default, or SPI_CONV_WINO_F4 unset): F(4x4, 3x3) -- 36 multiplications per 4x4 output tile and channel pair, 4x fewer than the direct sum --
 * where its 16 x 32-pixel blocks fill the chip (H % 16 == 0, W % 32 == 0, H W >= 256^2 with >= 128 reduction channels or H W >= 128^2 with >= 256, >= 256 blocks: the
 * generator's 128^2 (batched) / 256^2 / 512^2 layers), F(2x2, 3x3) elsewhere.
 * 0 (or SPI_CONV_WINO_F4=0): F(2x2, 3x3) everywhere.  Both are fp32 throughout; F(4x4) differs from the direct sum by ~1e-6 of the tensor's range
 * instead of ~2e-7 (what cuDNN's fp32 Winograd, the algorithm the reference runs through torch.nn.functional.conv2d at conv2d_gradfix.py:50-56, does too). */
void spi_conv_wino_f4_set(int on);
/* 1 if pass (0 forward, 1 dgrad) of `d` -- with the workspace `d` carries -- accumulates into its output through atomics (and therefore clears
 * it first unless d->out_zeroed), 0 if it overwrites, < 0 on a bad descriptor.  Host logic only: no launch, no device access. */
int spi_conv2d_out_accumulates(const spi_conv_desc* d, int pass);
/* How pass (0 forward, 1 dgrad, 2 wgrad) of `d` -- with the workspace `d` carries -- would be launched: out8 = {path (0 implicit GEMM,
 * 1 Winograd, 2 direct fp16), block tile rows (out channels), block tile columns (pixels; wgrad: weight columns), number of K ranges (split-K / channel
 * split; wgrad: pixel ranges), workgroups of the launch, threads per workgroup, 0, 0}.  Host logic only (tools/igemm_shapes.py: the per-shape
 * efficiency table); 0 or a negative error code. */
int spi_conv2d_plan(const spi_conv_desc* d, int pass, int32_t* out8);
int spi_conv2d_fwd  (const spi_conv_desc* d, const float* x, const float* w, float* y, spi_stream_t stream);
int spi_conv2d_dgrad(const spi_conv_desc* d, const float* dy, const float* w, float* dx, spi_stream_t stream);
/* dw has the layout/batching of w; with shared weights the batch is summed. */
int spi_conv2d_wgrad(const spi_conv_desc* d, const float* x, const float* dy, float* dw, spi_stream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Losses / optimiser helpers
 * ---------------------------------------------------------------------------------------------- */

/* rotate(), spi/utils/rotate.py:56-116: depth-guided warp of the source view into the target view.
 *   tgt_cam, src_cam [N,25]; tgt_depth, src_depth [N,dres,dres] (bilinearly resized to res inside);
 *   src_image [N,3,res,res]; src_mask [N,res,res] or NULL.  out: warp_rgb [N,3,res,res], warp_mask [N,res,res].
 *   src_cam_inv [N,16]: inverse of the source cam2world (host-side torch.inverse). */
int spi_rotate_warp(const float* tgt_cam, const float* src_cam_inv, const float* src_cam,
                    const float* tgt_depth, const float* src_depth, const float* src_image,
                    const float* src_mask, int N, int res, int dres, float eps,
                    float* warp_rgb, float* warp_mask, spi_stream_t stream);

/* Weight modulation / demodulation of modulated_conv2d (networks_stylegan2.py:62-69, fused path) in one pass:
 *   v[n,o,i,t] = weight[o,i,t] * styles[n,i] * style_gain;  d[n,o] = rsqrt(sum_{i,t} v^2 + 1e-8) (1 if !demodulate);
 *   w_out[n,o,t,i] = v * d     -- written TAP-MAJOR ([N,O,kh*kw,I]), the layout spi_conv2d_* take with w_tap_major = 1.
 * weight [O,I,T] (T = kh*kw, PyTorch layout), styles [N,I], dcoef [N,O] (output, kept for the backward pass). */
int spi_modulate_fwd(const float* weight, const float* styles, float* w_out, float* dcoef, int N, int O, int I,
                     int T, int demodulate, float style_gain, spi_stream_t stream);
/* Adjoint: g = d loss / d w_out [N,O,T,I] -> d_weight [O,I,T] (may be NULL: weights frozen, stage 1) and
 * d_styles [N,I] (accumulated with atomics: the CALLER zeroes it). */
int spi_modulate_bwd(const float* weight, const float* styles, const float* dcoef, const float* g, float* d_weight,
                     float* d_styles, int N, int O, int I, int T, int demodulate, float style_gain,
                     spi_stream_t stream);
/* The weight modulation of ALL layers of a synthesis network in one launch each way (every layer's styles are known before its first
 * convolution runs: spi_affine_multi_fwd).  jobs: HOST array of n_jobs <= SPI_MODULATE_MAX_JOBS descriptors (copied into the launch); the
 * fields are the arguments of spi_modulate_fwd / spi_modulate_bwd for one layer; N (style rows) is common.  _bwd: d_styles is ADDED to
 * (caller zeroes), d_weight may be NULL. */
#define SPI_MODULATE_MAX_JOBS 32
typedef struct {
    const float* weight; const float* styles; float* w_out; float* dcoef;      /* _fwd: dcoef may be NULL iff !demodulate */
    const float* g; float* d_weight; float* d_styles;                          /* _bwd */
    float style_gain; int O, I, T, demodulate;
} spi_modulate_job;
int spi_modulate_multi_fwd(const spi_modulate_job* jobs, int n_jobs, int N, spi_stream_t stream);
int spi_modulate_multi_bwd(const spi_modulate_job* jobs, int n_jobs, int N, spi_stream_t stream);


/* Noise regulariser of the stage-1 projectors (mirror_projector.py:106-116, w_plus_projector.py likewise):
 *   reg = sum_bufs sum_levels  mean(x * roll(x,1,W))^2 + mean(x * roll(x,1,H))^2,   x -> avg_pool2d(x,2) while size > 8
 * for T square noise buffers in ONE launch (one block per buffer walks its pyramid); the reference's autograd graph is
 * ~1200 tiny launches per optimisation step.
 *   bufs: device array of T buffer pointers; res: device int32 [T] (edge lengths, powers of two, <= max_res)
 *   pyramid: scratch [T * max_res^2 / 2] (pooled levels, kept for the backward); means [T*8*2]; loss [1] (CALLER zeroes). */
int spi_noise_reg_fwd(const float* const* bufs, const int32_t* res, int T, int max_res, float* pyramid, float* means,
                      float* loss, spi_stream_t stream);
/* d reg / d buf_t written to grads + goff[t] (int64 element offsets, device), scaled by the device scalar gout[0]. */
int spi_noise_reg_bwd(const float* const* bufs, const int32_t* res, int T, int max_res, const float* pyramid,
                      const float* means, const float* gout, float* grads, const int64_t* goff, float* gpyramid,
                      spi_stream_t stream);
/* After the optimiser step (mirror_projector.py:127-131): buf -= mean(buf); buf *= rsqrt(mean(buf^2)), all T in one launch. */
int spi_noise_renorm(float* const* bufs, const int32_t* res, int T, spi_stream_t stream);

/* LPIPS tail, lpips.py:43-65 + utils.py:6-8: per layer, out[n] += mean_hw( sum_c lin[c] *
 * (fx/(|fx|+1e-10) - fy/(|fy|+1e-10))^2 ).  fx, fy [N,C,HW]. */
int spi_lpips_layer_fwd(const float* fx, const float* fy, const float* lin, int N, int C, int64_t HW,
                        float* out, spi_stream_t stream);
/* d_out [N] -> d_fx [N,C,HW] (fy treated as constant). */
int spi_lpips_layer_bwd(const float* fx, const float* fy, const float* lin, const float* d_out, int N,
                        int C, int64_t HW, float* d_fx, spi_stream_t stream);

/* Contextual-loss chain of BoxCXLoss, bbox_cx_loss.py:93-129 (compute_cosine_distance's `1 - sim`, compute_relative_distance,
 * compute_cx, and the max / mean of compute_cx_loss before its -log), fused row-wise:
 *   out[b] = mean_j max_i cx[b,i,j],  cx = softmax_j((1 - rel)/band_width),  rel = clamp((1-sim)/(min_j(1-sim) + 1e-5), -10, 10)
 *   sim [B,P1,P2] cosine matrix (the torch.bmm of the normalised features stays with the caller); out [B].
 * Saved for the backward: row_min / row_sum [B,P1] (min_j dist, sum_j w), row_argmin [B,P1], col_argmax [B,P2] (int32).
 * workspace: spi_contextual_workspace_bytes(B,P1,P2) bytes of device scratch.  P2 <= 16384.  Exact ties of a minimum / maximum give
 * their gradient to the first index (torch.amin / amax split it evenly). */
int64_t spi_contextual_workspace_bytes(int B, int P1, int P2);
int spi_contextual_fwd(const float* sim, int B, int P1, int P2, float band_width, float* out, float* row_min,
                       float* row_sum, int32_t* row_argmin, int32_t* col_argmax, void* workspace, spi_stream_t stream);
/* d_out [B] -> d_sim [B,P1,P2] (every element written). */
int spi_contextual_bwd(const float* sim, const float* d_out, int B, int P1, int P2, float band_width,
                       const float* row_min, const float* row_sum, const int32_t* row_argmin,
                       const int32_t* col_argmax, float* d_sim, spi_stream_t stream);

/* torchvision.ops.roi_align(x, boxes, output_size = O) with spatial_scale = 1, sampling_ratio = -1, aligned = False (BoxCXLoss,
 * bbox_cx_loss.py:64-76), ONE box per image: x [N,C,H,W], boxes [N,4] (x1,y1,x2,y2) in device memory -> out [N,C,O,O].
 * _bwd: dy [N,C,O,O] -> dx [N,C,H,W] (the call zeroes dx). */
int spi_roi_align_fwd(const float* x, const float* boxes, float* out, int N, int C, int H, int W, int O, spi_stream_t stream);
int spi_roi_align_bwd(const float* boxes, const float* dy, float* dx, int N, int C, int H, int W, int O, spi_stream_t stream);

/* torch.optim.Adam (no amsgrad, no weight decay) over a list of tensors in one launch.
 *   ptrs: device array of 4*T pointers {param, grad, exp_avg, exp_avg_sq} per tensor; sizes: device int64 [T].
 *   step is the 1-based step count used for bias correction. */
int spi_adam_multi(void* const* ptrs, const int64_t* sizes, int T, int64_t max_size, float lr, float beta1,
                   float beta2, float eps, int step, spi_stream_t stream);

/* Affine (style) layers at inversion batch sizes: FullyConnectedLayer with activation 'linear', networks_stylegan2.py:95-127, as matrix-vector
 * products (N <= 8 rows).  y [N,O] = b [O] (NULL = 0) + gain * x [N,I] W[O,I]^T; in_features I a multiple of 4. */
int spi_affine_fwd(const float* x, const float* w, const float* b, float gain, float* y, int N, int I, int O, spi_stream_t stream);
/* g [N,O] -> dx [N,I] = gain * g W and / or dw [O,I] = gain * g^T x in one launch (either output may be NULL;
 * w is read for dx only, x for dw only).  The bias gradient is the column sum of g (caller's). */
int spi_affine_bwd(const float* g, const float* x, const float* w, float gain, float* dx, float* dw, int N, int I, int O, spi_stream_t stream);
/* ALL affine layers of a synthesis network in one launch each way (SynthesisNetwork.forward, networks_stylegan2.py:499-515: every SynthesisLayer /
 * ToRGBLayer maps its row of ws through its own FullyConnectedLayer -- 20 matrix-vector products of 5 us in the backbone, 6 per super-resolution
 * call).  jobs: HOST array of n_jobs <= SPI_AFFINE_MAX_JOBS descriptors (copied into the launch).  Row n of a job's input is x + n * x_row_stride
 * (rows of ws [N, L, I]: x = ws + l * I, x_row_stride = L * I).
 *   _fwd: y [N,O] = b + gain * x W^T per job.
 *   _bwd: per job g [N,O] (NULL: the job is skipped) -> dx_acc[n * x_row_stride ..] += gain * g W (atomic: jobs that read the same row of ws add
 *         into the same place; the caller zeroes; NULL = not wanted) and dw [O,I] = gain * g^T x (overwritten; NULL = not wanted). */
#define SPI_AFFINE_MAX_JOBS 32
typedef struct {
    const float* x; const float* w; const float* b;      /* input rows, weight [O,I], bias [O] or NULL */
    float* y;                                             /* _fwd output [N,O] */
    const float* g; float* dx_acc; float* dw;             /* _bwd */
    float gain; int O;
} spi_affine_job;
int spi_affine_multi_fwd(const spi_affine_job* jobs, int n_jobs, int N, int I, int64_t x_row_stride, spi_stream_t stream);
int spi_affine_multi_bwd(const spi_affine_job* jobs, int n_jobs, int N, int I, int64_t x_row_stride, spi_stream_t stream);
/* spi_adam_multi predicated on a DEVICE byte: *skip != 0 -> nothing is written.  The reference tests `loss_lpips <= threshold` on the
 * host before optimizer.step() (rot_bbox_cx_coach.py:148-151); a loop that enqueues iterations ahead of that read leaves the decision in
 * device memory and lets this launch honour it, so the parameters end exactly where the reference's break leaves them. */
int spi_adam_multi_pred(void* const* ptrs, const int64_t* sizes, int T, int64_t max_size, float lr, float beta1,
                        float beta2, float eps, int step, const unsigned char* skip, spi_stream_t stream);
/* The same launch with its step-dependent scalars in DEVICE memory: hyper = {lr, 1 - beta1^step, sqrt(1 - beta2^step)} (3 floats).
 * A stage-1 step captured in a HIP graph replays this launch unchanged; the host refreshes `hyper` (one 12-byte copy) before
 * each replay. */
int spi_adam_multi_dev(void* const* ptrs, const int64_t* sizes, int T, int64_t max_size, const float* hyper,
                       float beta1, float beta2, float eps, spi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SPI_HIP_H */
