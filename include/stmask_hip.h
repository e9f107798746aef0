/*
 * stmask_hip.h -- flat C ABI of libstmask_hip.so: hand-written gfx950 (MI355X / CDNA4) kernels for
 * STMask's per-frame inference hot path.
 *
 * This is the drop-in boundary.  The reference (MinghanLi/STMask) is pure Python; on this path it binds
 * four un-vendored third-party CUDA extensions and a handful of torch op chains.  Each entry point below
 * names the reference interface it replaces (file:line under the reference tree).  The Python shims in
 * stmask_amd/ (dcn_v2.DCN, mmcv.ops.DeformConv2d / roi_align, spatial_correlation_sample, layers.*) call
 * these through ctypes with tensor.data_ptr() and the current HIP stream; INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed from the caller (no allocation, no retention);
 *   - tensors are dense, row-major, fp32 unless stated; NCHW like the reference (proto / masks NHWC / NHW
 *     exactly as the reference lays them out);
 *   - calls only ENQUEUE work on `stream` (a hipStream_t passed as void*; NULL = default stream) and never
 *     synchronise; counts that the reference obtains through a device->host sync are written to device
 *     memory (`*_count`) so the caller decides when to read them;
 *   - return value: STM_OK (0) or a negative STM_E* code; stm_last_error_string() (thread-local) explains it.
 *     Nothing throws, nothing aborts;
 *   - process-wide state, all of it listed here: the thread-local error string; one fp16-range flag POINTER per device
 *     (stm_planar_set_range_flag: kernels launched on device d raise the flag registered on d); a once-read cache of the
 *     STM_* environment switches (stm_debug_reload_tunables makes the next call read them again); the
 *     per-device "dynamic LDS size reserved" marks of the convolution kernels.  Nothing else: no allocation, no stream, no
 *     tensor is retained.  Calls from different host threads are safe as long as they do not race on the same output memory;
 *     the error string is per thread.
 *   - STM_ABI_VERSION changes whenever a struct of this header changes size or an entry point changes signature or disappears
 *     (2: stm_conv_geom gained out_fmt_plus1, stm_conv2d_nhwc_f32 was removed, stm_struct_bytes and
 *     stm_debug_reload_tunables are declared;
 *      3: stm_conv_geom gained win_h / win_w / win_y0 / win_x0 and ph / pw may be negative (window launches), struct stm_conv_window and
 *     stm_conv2d_planar_windows_f32; the kx-reuse narrow-layer entries stm_conv_kxr_packed_bytes / stm_conv_pack_weights_kxr_f32 /
 *     stm_conv2d_planar_kxr_f32; stm_conv2d_planar_dual_f32; the fused stem stm_stem_packed_weight_bytes / stm_stem_pack_weights_f32 /
 *     stm_stem_fused_f32; the bottleneck chain stm_chain_tail_weight_bytes(_proj) / stm_chain_pack_tail(_proj)_f32 /
 *     stm_bottleneck_chain(_proj)_f32; stm_detect_cc_logits_f32; stm_corr_patch_nhwc_f32 / stm_roi_align_planes_nhwc_f32).
 *     stm_struct_bytes(which) lets a client check its layout of EVERY struct of this header against the library.
 *     Added since without a version change (new entry points only): stm_conv2d_planar_windows_pool_f32, stm_temporal_pool_fc_f32,
 *     stm_debug_launch_count;
 *      4: those three entry points are part of the version now (a library that lacks them must not pass for this ABI), plus the fused
 *     deformable convolution stm_deform_conv_fused_planar_f32 / stm_deform_conv_fused_planar_supported; stm_debug_launch_count(1); the batched per-class Fast NMS
 *     stm_fast_nms_batched_f32 / stm_fast_nms_batched_workspace_bytes; the host-side stm_rle_strings_host).
 */
#ifndef STMASK_HIP_H_
#define STMASK_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STM_ABI_VERSION 4

enum stm_status {
    STM_OK = 0,
    STM_EINVAL = -1,       /* bad dimension / unsupported combination of arguments */
    STM_ENULL = -2,        /* required pointer is NULL */
    STM_ELAUNCH = -3,      /* HIP reported an error when enqueuing */
    STM_EWORKSPACE = -4,   /* workspace missing or too small (see *_workspace_bytes) */
    STM_EUNSUPPORTED = -5  /* valid in the reference API but outside the hot path (e.g. groups != 1) */
};

typedef void* stm_stream_t; /* hipStream_t */

int stm_version(void);
const char* stm_last_error_string(void);
/* sizeof of the library's own view of a header struct: which = 0 stm_deform_geom, 1 stm_conv_geom, 2 stm_conv_window (passed as an
 * array: a wrong size is a wrong stride), 3 stm_head_layout; 0 for anything else */
size_t stm_struct_bytes(int which);
/* Diagnostics: the STM_* environment switches are read once per process; this makes the next call read them again (tests, A/B
 * scripts).  Not needed in production. */
void stm_debug_reload_tunables(void);
/* Diagnostics: launches of an optional kernel form since the process started (tests check that the form they compare really ran):
 * which = 0 conv_planar_kx3_kernel (kx-reuse staging of the 256 x 128 ring tiles), 1 dcn_fused_kernel (the fused deformable
 * convolution); -1 for anything else. */
long long stm_debug_launch_count(int which);

/* ---------------------------------------------------------------------------------------------------
 * Deformable convolution.
 * Replaces: dcn_v2.DCN.forward -> _backend.dcn_v2_forward (backbone.py:5,21-26,45; modulated, bias)
 *           mmcv.ops.DeformConv2d.forward -> ext_module.deform_conv_forward
 *                                           (Featurealign.py:3,27-31,72; no mask, no bias)
 *
 *   y[b,o,ho,wo] = bias[o] + sum_{c,i,j} w[o,c,i,j] * m[b,g,k,ho,wo] *
 *                  bilinear(x[b,c], ho*sh - ph + i*dh + dy, wo*sw - pw + j*dw + dx)        k = i*kw + j
 *   offset: channel (g*2K + 2k) = dy, (g*2K + 2k + 1) = dx;  batch stride off_bstride (elements)
 *   mask  : channel (g*K + k); NULL -> v1 (no modulation);  batch stride mask_bstride (elements);
 *           mask_is_logit != 0 -> sigmoid applied inside the kernel (DCN feeds the raw
 *           conv_offset_mask output: offset = om, mask = om + 2K*Ho*Wo, both with batch stride 3K*Ho*Wo,
 *           which fuses dcn_v2's chunk / cat / sigmoid)
 *   a sample contributes iff -1 < y < H and -1 < x < W; out-of-range corners read 0.
 *
 * stm_deform_im2col_f32 writes cols[B][C*K][Ho*Wo] (row c*K + k).
 * stm_deform_conv_fwd_f32 = im2col into `workspace` + fp32 MFMA GEMM with fused bias / optional ReLU.
 * ------------------------------------------------------------------------------------------------- */
typedef struct stm_deform_geom {
    int B, C, H, W;      /* input  [B,C,H,W] */
    int kh, kw, sh, sw;  /* kernel, stride */
    int ph, pw, dh, dw;  /* padding, dilation */
    int dg;              /* deformable groups (C % dg == 0) */
    int Ho, Wo;          /* output spatial size (must match the conv arithmetic) */
} stm_deform_geom;

int stm_deform_im2col_f32(const float* x, const float* offset, int64_t off_bstride, const float* mask,
                          int64_t mask_bstride, int mask_is_logit, float* cols, const stm_deform_geom* g,
                          int variant, stm_stream_t stream);
/* variant: 0 = auto, 1 = direct gather (global loads), 2 = LDS-staged input tiles (4 or 8 channels per workgroup),
 *          3 = LDS-staged, bilinear coefficients computed once per workgroup and reused across channel quads */

size_t stm_deform_conv_workspace_bytes(const stm_deform_geom* g);

int stm_deform_conv_fwd_f32(const float* x, const float* offset, int64_t off_bstride, const float* mask,
                            int64_t mask_bstride, int mask_is_logit, const float* weight /*[O,C,kh,kw]*/,
                            const float* bias /*[O] or NULL*/, float* y /*[B,O,Ho,Wo]*/, int O, int relu,
                            const stm_deform_geom* g, void* workspace, size_t workspace_bytes,
                            stm_stream_t stream);

/* Plain fp32 GEMM on the MFMA pipe (exact fp32 fmaf chain per output):
 *   Cmat[b][M,N] = A[M,K] * Bmat[b][K,N] (+ bias[m]) (ReLU optional), b < batch, A shared by the batch.
 * Used for the column-buffer GEMM above; exported for tests and profiling. */
int stm_gemm_bias_f32(const float* A, const float* Bmat, const float* bias, float* Cmat, int M, int N, int K,
                      int batch, int64_t b_bstride, int64_t c_bstride, int relu, stm_stream_t stream);
/* Same, with scratch for deterministic split-K (small M x N grids are split along K so that >= ~256 workgroups run;
 * partial sums are reduced in a fixed order).  workspace may be NULL (then no split). */
size_t stm_gemm_workspace_bytes(int M, int N, int batch);
int stm_gemm_bias_ws_f32(const float* A, const float* Bmat, const float* bias, float* Cmat, int M, int N, int K,
                         int batch, int64_t b_bstride, int64_t c_bstride, int relu, void* workspace,
                         size_t workspace_bytes, stm_stream_t stream);

/* FeatureAlign "ali" offsets (Featurealign.py:46-69): loc [B,4,H,W] -> offset [B,2*kh*kw,H,W].
 * fp32, reference operand order, canonical exp (bit-exact against the oracle). */
int stm_fcb_ali_offsets_f32(const float* loc, float* offset, int B, int H, int W, int kh, int kw,
                            stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Spatial correlation, kernel_size = 1, stride = 1, padding = 0.
 * Replaces: spatial_correlation_sampler.spatial_correlation_sample (track_to_segment_head.py:4,53-59)
 *           and, through scale / leaky_slope, the reference's follow-up `/ C` and leaky_relu_(0.1)
 *           (track_to_segment_head.py:60-62).  Pass scale = 1, leaky_slope = 1 for the raw sampler.
 *   out[b,i,j,y,x] = act(scale * sum_c f1[b,c,y,x] * f2[b,c,y+(i-P/2)*dil, x+(j-P/2)*dil])   [B,P,P,H,W]
 * ------------------------------------------------------------------------------------------------- */
int stm_corr_patch_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int P,
                       int dil, float scale, float leaky_slope, stm_stream_t stream);
/* the same values channels-last: out[b,y,x,i*P+j], [B,H,W,out_ld] with out_ld >= P*P (channels past P*P are not written) -- the
 * layout stm_roi_align_planes_nhwc_f32 gathers from.  in_nhwc != 0: f1 / f2 are channels-last too, [B,H,W,C] (what the trunk's
 * fp32 outputs are; same arithmetic, same bits) */
int stm_corr_patch_nhwc_f32(const float* f1, const float* f2, float* out, int B, int C, int H, int W, int P, int dil,
                            float scale, float leaky_slope, int out_ld, int in_nhwc, stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * RoIAlign, average pooling.
 * Replaces: mmcv.ops.roi_align -> ext_module.roi_align_forward (track_to_segment_head.py:6,86).
 *   feat [B,C,H,W], rois [n,5] = (batch, x1, y1, x2, y2) -> out [n,C,PH,PW].
 * ------------------------------------------------------------------------------------------------- */
int stm_roi_align_avg_f32(const float* feat, const float* rois, float* out, int B, int C, int H, int W, int n,
                          int PH, int PW, float spatial_scale, int sampling_ratio, int aligned,
                          stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Box decoding.  Replaces: layers.box_utils.decode (box_utils.py:238-283), bit-exact vs the oracle.
 * ------------------------------------------------------------------------------------------------- */
int stm_decode_boxes_f32(const float* loc, const float* priors, float* boxes, int64_t n,
                         stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Candidate generation.  Replaces: generate_candidate (TF_utils.py:54-82): decode + "max foreground
 * confidence > thresh" + ordered boolean compaction.
 *   loc [N,4], priors [N,4], conf [N,ncls] (soft-maxed) ->
 *   keep_idx [N] (ascending prior indices, first *count valid), cand_box [N,4] rows compacted likewise,
 *   count (device int).  One launch per frame; `batch` frames are processed by `batch` workgroups
 *   (inputs / outputs strided by N rows, priors shared, count[batch]).
 * ------------------------------------------------------------------------------------------------- */
int stm_generate_candidates_f32(const float* loc, const float* priors, const float* conf, int N, int ncls,
                                float thresh, int batch, int64_t* keep_idx, float* cand_box, int* count,
                                stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Cross-class Fast NMS.  Replaces: Detect_TF.cc_fast_nms (detection_TF.py:85-134) == Detect.cc_fast_nms
 * (detection.py:139-187) incl. jaccard / intersect (box_utils.py:37-88).
 *   conf [K,ncls] candidate rows (column 0 = background), boxes [K,4], centerness [K] or NULL.
 *   score = max_{c>=1} conf * centerness; order: score desc, ties -> lower row first; top_k; IoU upper
 *   triangle; column max; keep <= iou_thr.
 *   Outputs (capacity top_k each): idx_out (candidate row, int64), cls_out (argmax + 1, int64),
 *   score_out, box_out [top_k,4]; count_out (device int).
 *   k_dev: optional device pointer holding the real K (<= K) -- lets the whole chain run without a host
 *   sync.  `batch` independent problems are strided by K rows (outputs by top_k rows).
 *   Indices are bit-exact against the oracle.  K <= 16384 (one workgroup sorts in LDS); beyond: STM_EUNSUPPORTED,
 *   use stm_cc_fast_nms_ws_f32.
 * ------------------------------------------------------------------------------------------------- */
int stm_cc_fast_nms_f32(const float* conf, const float* boxes, const float* centerness, int K, int ncls,
                        const int* k_dev, float iou_thr, int top_k, int batch, int64_t* idx_out,
                        int64_t* cls_out, float* score_out, float* box_out, int* count_out,
                        stm_stream_t stream);

/* The same for ANY number of candidate rows (736x1280 frames have 58 860 priors, and a weakly trained head can pass most of
 * them): the per-row scores go to `workspace` (stm_cc_fast_nms_workspace_bytes(K, batch) = 4*K*batch + 256 bytes) and the NMS
 * workgroup, when K exceeds the 16 384 keys its LDS sort holds, first radix-selects exactly the top_k best (score desc, row
 * asc) candidates -- the result is the one sorting everything gives (detection_TF.py:93).  No k_dev form. */
size_t stm_cc_fast_nms_workspace_bytes(int K, int batch);
int stm_cc_fast_nms_ws_f32(const float* conf, const float* boxes, const float* centerness, int K, int ncls,
                           float iou_thr, int top_k, int batch, int64_t* idx_out, int64_t* cls_out,
                           float* score_out, float* box_out, int* count_out, void* workspace,
                           size_t workspace_bytes, stm_stream_t stream);

/* Per-class Fast NMS.  Replaces: Detect_TF.fast_nms (detection_TF.py:136-204) == Detect.fast_nms
 * (detection.py:211-263).  Outputs have capacity max_det.  workspace: stm_fast_nms_workspace_bytes. */
size_t stm_fast_nms_workspace_bytes(int K, int ncls, int top_k);
/* Fused generate_candidate + cross-class Fast NMS (STMask.py:317-320 chain) with no host round trip.
 *   loc [batch,N,4], priors [N,4], conf [batch,N,ncls] soft-maxed, centerness [batch,N] or NULL ->
 *   idx_out [batch,top_k] = PRIOR index of every detection (gather mask_coeff / track rows with it),
 *   cls / score / box likewise, count_out [batch].  Up to 16384 candidates per frame. */
size_t stm_detect_cc_workspace_bytes(int N, int batch);
int stm_detect_cc_f32(const float* loc, const float* priors, const float* conf, const float* centerness, int N,
                      int ncls, float conf_thresh, float iou_thr, int top_k, int batch, int64_t* idx_out,
                      int64_t* cls_out, float* score_out, float* box_out, int* count_out, void* workspace,
                      size_t workspace_bytes, stm_stream_t stream);
/* the same with conf = the raw class logits [batch, N, ncls]: the softmax of STMask.py:314 (F.softmax(pred_outs['conf'], -1)) is taken
 * per row inside the candidate pass (max over all classes, sum of exp(x - max) in class order) instead of by a separate pass over the
 * whole tensor; classes (argmax) and the keep / NMS logic are unchanged */
int stm_detect_cc_logits_f32(const float* loc, const float* priors, const float* conf_logits, const float* centerness, int N,
                      int ncls, float conf_thresh, float iou_thr, int top_k, int batch, int64_t* idx_out,
                      int64_t* cls_out, float* score_out, float* box_out, int* count_out, void* workspace,
                      size_t workspace_bytes, stm_stream_t stream);

int stm_fast_nms_f32(const float* conf, const float* boxes, const float* centerness, int K, int ncls,
                     const int* k_dev, float iou_thr, int top_k, float conf_thresh, int max_det,
                     int64_t* idx_out, int64_t* cls_out, float* score_out, float* box_out, int* count_out,
                     void* workspace, size_t workspace_bytes, stm_stream_t stream);
/* The same for B frames in ONE launch pair (ABI 4; the batched clip pipeline's use_cross_class_nms = False path): frame b reads the confidence rows
 * conf + b * conf_bstride (ncls floats each), the COMPACTED candidate boxes boxes + b * 4 K (float4 rows, as stm_generate_candidates_f32 writes them),
 * centerness + b * cen_bstride and its candidate count k_dev[b] (device, <= K = the capacity of a frame's candidate arrays).  row_index (optional,
 * [B][K] int64): candidate row i of frame b is row row_index[b][i] of conf / centerness -- the candidate pass's keep list, so that the [N, ncls]
 * confidence rows need not be gathered; idx_out then holds those row numbers (prior indices).  Outputs [B][max_det] (+ box [B][max_det][4]),
 * count_out [B].  Each frame's LDS sort covers its own count, not K.  Scores, order and ties as the one-frame entry (which is this one with B = 1). */
size_t stm_fast_nms_batched_workspace_bytes(int K, int ncls, int top_k, int B);
int stm_fast_nms_batched_f32(const float* conf, int64_t conf_bstride, const int64_t* row_index, const float* boxes, const float* centerness,
                             int64_t cen_bstride, int K, int ncls, const int* k_dev, float iou_thr, int top_k, float conf_thresh, int max_det,
                             int B, int64_t* idx_out, int64_t* cls_out, float* score_out, float* box_out, int* count_out,
                             void* workspace, size_t workspace_bytes, stm_stream_t stream);

/* Pairwise box IoU.  Replaces: layers.box_utils.jaccard (box_utils.py:60-88), 2-D form, bit-exact. */
int stm_jaccard_f32(const float* a, int na, const float* b, int nb, float* out, stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Prototype linear combination + sigmoid + crop.
 * Replaces: generate_mask (mask_utils.py:111-128) + crop / sanitize_coordinates (box_utils.py:298-364).
 *   proto [h,w,m] (NHWC as the reference permutes it, STMask.py:236), coeff [n,m], boxes [n,4] relative
 *   x1y1x2y2 or NULL (no crop) -> out [n,h,w] soft masks (already in the reference's permuted layout).
 *   apply_tanh: mask_proto_coeff_activation (config.py:447).  n_dev: optional device count (<= n).
 *   row_proto: optional device int[n]: row i uses prototype set proto + row_proto[i]*h*w*m, so the rows of
 *   several frames (clips) share one launch; NULL -> every row uses proto.
 * ------------------------------------------------------------------------------------------------- */
int stm_lincomb_sigmoid_crop_f32(const float* proto, const float* coeff, const float* boxes, float* out, int h,
                                 int w, int m, int n, int apply_tanh, const int* n_dev, const int* row_proto,
                                 stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Binary mask IoU.  Replaces: mask_iou (box_utils.py:435-447) on m.gt(thr).float() inputs
 * (track_TF.py:85,107).  m1 [n1,hw], m2 [n2,hw] SOFT masks (binarised `> thr` here) -> out [n1,n2].
 * workspace: bit-packed masks, stm_mask_iou_workspace_bytes(n1, n2, hw).
 * ------------------------------------------------------------------------------------------------- */
size_t stm_mask_iou_workspace_bytes(int n1, int n2, int hw);
int stm_mask_iou_f32(const float* m1, int n1, const float* m2, int n2, int hw, float thr, float* out,
                     void* workspace, size_t workspace_bytes, stm_stream_t stream);
/* Batched-clip form: group1 [n1], group2 [n2] (int32 -- the clip of each mask; any order is correct, rows sorted by clip as
 * stmask_amd.pipeline keeps them let whole workgroups skip): out[i][j] = IoU if group1[i] == group2[j], else 0 without reading
 * the pair (track_TF matches a detection only against the tracked instances of its own clip). */
int stm_mask_iou_grouped_f32(const float* m1, int n1, const float* m2, int n2, int hw, float thr, float* out, const int* group1,
                             const int* group2, void* workspace, size_t workspace_bytes, stm_stream_t stream);
/* The same two kernels without the re-read of the soft masks: stm_lincomb_sigmoid_crop_bits_f32 also writes the binarised mask
 * (value > bits_thr) as 64-pixel words [n][ceil(h*w / 64)] (bit k of word j = pixel 64 j + k), stm_mask_iou_bits_f32 takes two
 * such tables (mask_utils.py:111-128 + box_utils.py:435-447 on m.gt(0.5): the matching of track_TF.py:104-110). */
int stm_lincomb_sigmoid_crop_bits_f32(const float* proto, const float* coeff, const float* boxes, float* out, int h, int w, int m,
                                      int n, int apply_tanh, const int* n_dev, const int* row_proto, uint64_t* bits, float bits_thr,
                                      stm_stream_t stream);
int stm_mask_iou_bits_f32(const uint64_t* bits1, int n1, const uint64_t* bits2, int n2, int hw, float* out, const int* group1,
                          const int* group2, stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Fused dense-conv epilogue, in place: y = act(y + bias[c] (+ residual)).
 * Replaces: the BatchNorm / ReLU / residual-add passes around every trunk convolution (backbone.py:38-58,
 * make_net.py ReLUs, prediction_head_FC.py extras) once BN is folded into the conv weights (stmask_amd/fuse.py).
 *   channel of element i = (i / inner) % C; inner = H*W for NCHW, 1 for NHWC.  relu != 0 -> max(., 0).
 * ------------------------------------------------------------------------------------------------- */
int stm_bias_act_f32(float* y, const float* bias, const float* residual, int64_t n, int C, int64_t inner,
                     int relu, stm_stream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Output stage (next row after the hot path): mask leg of postprocess_ytbvis (output_utils.py:85-106).
 * Replaces: masks[:, :crop_h, :crop_w] -> F.interpolate(bilinear, align_corners=False) to out_h x out_w -> gt(0.5)
 *           -> per-mask .cpu() -> pycocotools.mask.encode (COCO RLE of the column-major binary image).
 *   masks [n, mh, mw] soft masks -> counts [n, max_runs] uint32 run lengths (first run = zeros, possibly 0),
 *   n_runs [n] (true number of runs; > max_runs means the row overflowed).  The 5-bit string packing of COCO RLE
 *   (maskApi.c rleToString) is a few hundred bytes per mask and is done by the caller on the host.
 * ------------------------------------------------------------------------------------------------- */
size_t stm_mask_rle_workspace_bytes(int n, int out_h, int out_w, int max_runs);
/* HOST function (no device work): COCO compressed RLE strings (maskApi.c rleToString) of n masks' run lengths -- counts [n][ld] as
 * stm_mask_resize_rle_f32 leaves them, copied to the host; row i holds n_runs[i] runs -- into out [n][out_ld] bytes, lengths in out_len [n].
 * STM_EINVAL if a string needs more than out_ld bytes (out_len[i] then says how many).  ABI 4. */
int stm_rle_strings_host(const uint32_t* counts, int ld, const int* n_runs, int n, char* out, int out_ld, int* out_len);
int stm_mask_resize_rle_f32(const float* masks, int n, int mh, int mw, int crop_h, int crop_w, int out_h, int out_w,
                            float thr, uint32_t* counts, int max_runs, int* n_runs, void* workspace,
                            size_t workspace_bytes, stm_stream_t stream);

/* ---- dense convolution on the bf16 matrix cores with fp32-equivalent accuracy (row f4 and the dense part of a2-a5) ----
 * Replaces torch.nn.Conv2d (+ folded eval BatchNorm + ReLU + residual add) as the reference uses it in
 * Bottleneck.forward (backbone.py:38-58), FPN.forward (FPN.py:68-108), the proto-net (make_net.py:5-59) and the
 * PredictionModule_FC tower (prediction_head_FC.py:146-195).  Activations are NHWC fp32 (a torch channels_last tensor
 * is exactly this); every fp32 operand is split into `planes` bf16 terms (3: six products, error ~2^-24, the default;
 * 2: three products, error ~2^-16) and accumulated in fp32.  Input channels must be a multiple of 32. */
typedef struct stm_conv_geom {
    int B, H, W, C;       /* input  [B, H, W, C] (NHWC) */
    int Ho, Wo, Cout;     /* output [B, Ho, Wo, Cout] */
    int kh, kw, sh, sw, ph, pw;
    int x_ld, out_ld, res_ld; /* elements between consecutive pixels of x / out / residual; 0 = dense (groups*C / Cout / Cout) */
    int planes;           /* planes of the format: 3 (fmt 0; 2 = its looser three-product mode), 2 (fmt 1), 1 (fmt 2) */
    /* --- the fields below are read by stm_conv2d_planar_f32 only; all zero = one dense ungrouped image batch --- */
    int groups;           /* grouped convolution: C input channels and Cout/groups output channels PER GROUP (0 = 1);
                             Cout/groups must be a multiple of 128 when groups > 1 */
    int n_levels;         /* > 0: the pixel axis is the concatenation of n_levels (<= 8) image batches of different
                             sizes -- the FPN levels a shared prediction head runs over in ONE launch.  Level l holds
                             pixels [lvl_start[l], lvl_start[l+1]) = B images of lvl_h[l] x lvl_w[l]; stride 1 and
                             "same" padding only; B/H/W/Ho/Wo are then ignored */
    int lvl_start[9], lvl_h[8], lvl_w[8];
    long long x_plane_stride, out_plane_stride, res_plane_stride; /* elements between bf16 planes; 0 = dense (slabs*np*32) */
    int x_np, out_np, res_np; /* pixels per channel slab of the planar input / output / residual buffers (0 = exactly the
                             pixels of this launch): lets a layer read from / write into a slice of a larger buffer */
    int group_cout[8];    /* grouped layers whose groups are zero-padded to a common width: real output channels of group i
                             (0 = all Cout/groups); the matrix-core tiles that would only multiply padding are skipped */
    int fmt;              /* plane format: 0 = three bf16 planes (six MFMA products per fp32 product; any fp32 range),
                             1 = two fp16 planes, x = h + l / 2048 with the low plane stored scaled so that it keeps its
                             11 bits (three products: half the matrix work, same fp32-level error; 22 bits of x for
                             6.1e-5 <= |x| <= 65504, absolute error < 1.5e-11 below.  Beyond 65504 a value has no
                             representation: see stm_planar_set_range_flag);
                             planes must then be 2 and the weights packed with stm_conv_pack_weights_fmt_f32;
                             2 = ONE fp16 plane: genuine fp16 activations and weights, one MFMA product, fp32 accumulation,
                             bias / residual / ReLU in fp32 (BASELINE config 5, "fp16 MFMA backbone convs"; error ~2^-11
                             per product, stated tolerance 2e-3 of sum |x w|); planes = 1.  Plane 0 of a fmt-1 tensor IS
                             the fmt-2 tensor of the same values */
    float out_scale;      /* fmt 1 / 2: 1 / wscale of the packed weights (0 = 1) */
    int tile_n;           /* output-channel tile the weights were packed for: 0 / 128 (stm_conv_pack_weights_f32) or 64
                             (stm_conv_pack_weights_tiled_f32): 128 x 64 tiles, two workgroups per CU -- narrow layers
                             (few output channels) and layers with few pixel tiles */
    int out_fmt_plus1;    /* 0: out_planes in `fmt`; k + 1: out_planes in format k.  Only fmt 2 -> out format 1 is
                             supported (the last fp16 layer of a ResNet stage hands both planes to the fp32-equivalent FPN) */
    int win_h, win_w, win_y0, win_x0; /* (ABI 3) win_w > 0: WINDOW launch of stm_conv2d_planar_f32 -- the launch computes only the
                             Ho x Wo outputs starting at (win_y0, win_x0) of every win_h x win_w output image and writes them to rows
                             b * win_h * win_w + (win_y0 + oy) * win_w + win_x0 + ox of the output tensors (out_np / the fp32 matrix
                             cover B * win_h * win_w rows).  Ho / Wo are free and ph / pw may be negative: output (oy, ox) reads input
                             (oy * sh - ph + ky, ox * sw - pw + kx).  With the kernel cut to the taps that can be inside the image this
                             runs the border classes of a "same" convolution without their zero taps (TemporalNet's 3x3 layers on 7x7
                             RoI maps: 361 of 441 tap-pixels are real).  No residual, no levels, no second source. */
} stm_conv_geom;

/* One layer as several window launches in ONE grid (csrc/conv_bf16x.hip, CLS instantiation): window i = the Ho x Wo outputs at (y0, x0) of every
 * g->win_h x g->win_w output image, computed with its own kh x kw sub-kernel (packed_weights[i], packed with stm_conv_pack_weights_fmt_f32 at
 * tile_n 128 and ONE common wscale, g->out_scale = 1 / wscale) and padding ph / pw (may be negative: output (oy, ox) of the window reads input
 * (oy - ph + ky, ox - pw + kx)).  g: B, H, W, C, Cout, sh = sw = 1, fmt 1, planes 2, win_h / win_w, x_np / out_np / strides as usual (the
 * outputs have B * win_h * win_w rows); its kh / kw / ph / pw / Ho / Wo / win_y0 / win_x0 are ignored.  At most 9 windows.
 * Use: the border classes of a "same" 3x3 convolution on small maps (TemporalNet, track_to_segment_head.py:10-37: three 3x3 / pad-1 layers on
 * 7x7 RoI maps) -- rows {0}, {1..H-2}, {H-1} x columns likewise, each with the taps that can be inside the map: 361 of 441 tap-pixels are
 * multiplied instead of 441, and the results are the same sums (bit-equal to the padded launch). */
typedef struct stm_conv_window { int kh, kw, ph, pw, Ho, Wo, y0, x0; } stm_conv_window;
int stm_conv2d_planar_windows_f32(const void* x_planes, const void* const* packed_weights, const stm_conv_window* windows, int n_windows,
                                  const float* bias, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu, stm_stream_t stream);

/* The same window set with ReLU and an average pool over each whole output image folded into the epilogue -- TemporalNet's conv3 -> ReLU ->
 * AvgPool2d((7, 7)) (track_to_segment_head.py:30-33) without the per-pixel tensor.  pool_fix [B][Cout] (unsigned 64-bit, 32.32 fixed point, 8-byte
 * aligned) receives, ADDED to what it holds, the sum over the image's win_h x win_w pixels of relu(conv + bias): zero it before the first use;
 * stm_temporal_pool_fc_f32 zeroes what it consumes.  The accumulation is integer (each workgroup converts its fp32 partial sum exactly), so the
 * result does not depend on the order in which workgroups finish.  A partial sum of 2^32 or more (or NaN) raises the device's range flag.
 * g as for stm_conv2d_planar_windows_f32, Cout a multiple of 128; the out_* fields are ignored. */
int stm_conv2d_planar_windows_pool_f32(const void* x_planes, const void* const* packed_weights, const stm_conv_window* windows, int n_windows,
                                       const float* bias, unsigned long long* pool_fix, const stm_conv_geom* g, stm_stream_t stream);

/* TemporalNet's tail (track_to_segment_head.py:33-37): mean[b][c] = pool_fix[b][c] / 2^32 / npix, y[b][o] = bias[o] + sum_c mean[b][c] weight[o][c]
 * for o < n_out -- weight [n_out][C] row-major is the reference's fc and fc_coeff stacked (4 + 32 rows), bias [n_out] or NULL.  Rows o < n_first
 * go to out [n][n_first], the others to out2 [n][n_out - n_first] (out2 NULL: everything to out [n][n_out]).  pooled_out [n][C] (optional)
 * receives the means.  clear != 0 zeroes the rows of pool_fix it has read.  One fixed summation order: run-to-run identical. */
int stm_temporal_pool_fc_f32(unsigned long long* pool_fix, int n, int C, int npix, const float* weight, const float* bias, int n_out, int n_first,
                             float* out, float* out2, float* pooled_out, int clear, stm_stream_t stream);

/* Two-source 1x1 convolution on the planar kernel: y = W [x1 ; x2 (stride s2)] + bias (+ residual) (+ ReLU) -- the last 1x1 convolution of
 * a ResNet stage's first bottleneck and its projection shortcut (backbone.py:38-58: out = bn3(conv3(out)); out += downsample(x); relu)
 * as ONE product over the concatenated input channels, W = [W3 | Wds] (packed with stm_conv_pack_weights_fmt_f32 as a [Cout, C1 + C2, 1, 1]
 * tensor), bias = b3 + bds.  g: B, H = Ho, W = Wo of the output, kh = kw = 1, stride 1, no padding, C = C1 + C2, fmt 1 or 2, x_np /
 * x_plane_stride for x_planes (which holds the C1 = C - C2 channels, one pixel per output pixel).  x2_planes: C2 channels of B images of
 * H2 x W2 (x2_np pixels per slab, 0 = B * H2 * W2; x2_plane_stride elements between planes, 0 = dense), read at stride s2:
 * Ho = (H2 - 1) / s2 + 1.  The projection's output tensor (written once and read back as the residual) never exists. */
int stm_conv2d_planar_dual_f32(const void* x_planes, const void* x2_planes, int C2, int H2, int W2, int s2, long long x2_np,
                               long long x2_plane_stride, const void* packed_weight, const float* bias, const float* residual_f32,
                               const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu,
                               void* workspace, size_t workspace_bytes, stm_stream_t stream);

/* ---- narrow-output stride-1 convolution with kx-reuse (csrc/conv_kxr.hip) ------------------------------------------------------
 * The same convolution for the layers with FEW output channels per group (<= 64 real ones, <= 4 groups), stride 1, "same" padding,
 * kw = 3 or 5, plane formats 1 / 2: the shared head's output layers (prediction_head_FC.py:146-195: conf / centerness + bbox / mask
 * groups of 41 / 5 / 32 channels over three kernel shapes), the DCN offset / mask convolutions at stride 1 (backbone.py:20-26), the
 * 64 -> 64 3x3 convolutions of ResNet layer1 (backbone.py:38-58).  On stm_conv2d_planar_f32's 128 x 64 tiles these layers run at the
 * L2 -> LDS staging rate; here each run of 256 flat pixels is staged once per (channel slab, ky) and serves all kw taps, the weight
 * tiles are as wide as the group's real channels (rounded up to 16), and results are stored straight from the accumulators.
 * Geometry: stm_conv_geom with C = input channels PER GROUP, Cout = groups * (output row stride of a group), group_cout[i] = real
 * channels of group i (0 = all), n_levels / lvl_* or B, H, W as for stm_conv2d_planar_f32, fmt, out_scale, x_np / x_plane_stride /
 * out_ld / out_np / out_plane_stride; tile_n, res_* and planes are ignored.  Every 16-channel tile of a group is written whole:
 * channels [group_cout[i], 16 * ceil(group_cout[i] / 16)) of a group receive bias-only values.  No residual input.
 * Weights: stm_conv_pack_weights_kxr_f32 of the OIHW tensor [Cout, C, kh, kw] (its own layout; wscale as stm_conv_pack_weights_fmt_f32). */
size_t stm_conv_kxr_packed_bytes(const struct stm_conv_geom* g);
int stm_conv_pack_weights_kxr_f32(const float* weight, void* packed, const struct stm_conv_geom* g, float wscale, stm_stream_t stream);
int stm_conv2d_planar_kxr_f32(const void* x_planes, const void* packed_weight, const float* bias, float* out_f32, void* out_planes,
                              const struct stm_conv_geom* g, int relu, stm_stream_t stream);

/* ---- bottleneck chain for the 64-channel blocks of ResNet layer1 (csrc/conv_chain.hip; reference backbone.py:38-58) ---------------
 *   mid2 = relu(conv2_3x3(mid1) + b2);  y = relu(conv3_1x1(mid2) + b3 + x);  z = relu(next_conv1_1x1(y) + b1_next)   (BatchNorm folded)
 * as one launch in plane format 1 (fp16 x 2): mid2 and y feed the next product from the accumulators, the 3x3's input, the identity
 * shortcut and the outputs cross HBM once.  mid1 / z: [B, H, W, 64] planes, x / y: [B, H, W, 256] planes, all dense
 * ([2][C/32][B*H*W][32]).  w2_packed = stm_conv_pack_weights_kxr_f32 of conv2's [64, 64, 3, 3] (C = 64, Cout = 64, one group, fmt 1);
 * tail_packed = stm_chain_pack_tail_f32 of conv3's [256, 64] and (or NULL) the next block's conv1 [64, 256], each times its power-of-two
 * weight scale; out_scale* = 1 / that scale.  z_planes may be NULL (then b1_next and the conv1 half of the tail are unused).
 * Arithmetic = the three stm_conv2d_planar_f32 calls it replaces up to the order of the fp32 sums inside one 32-channel K-slab.
 * The _proj forms are a stage's FIRST block at stride 1 (projection shortcut, backbone.py:52-53): x0 = the block's 64-channel input
 * [B, H, W, 64] planes, y = relu(conv3(mid2) + proj(x0) + b3) with b3 = conv3's bias + the projection's and both weight matrices
 * ([256, 64] each) under ONE scale wscale3 -- the product stm_conv2d_planar_dual_f32 computes, chained the same way. */
size_t stm_chain_tail_weight_bytes(void);
size_t stm_chain_tail_weight_bytes_proj(void);
int stm_chain_pack_tail_f32(const float* w3, const float* w1_next, void* packed, float wscale3, float wscale1, stm_stream_t stream);
int stm_chain_pack_tail_proj_f32(const float* w3, const float* wds, const float* w1_next, void* packed, float wscale3, float wscale1,
                                 stm_stream_t stream);
int stm_bottleneck_chain_f32(const void* mid1_planes, const void* x_planes, void* y_planes, void* z_planes, const void* w2_packed,
                             const void* tail_packed, const float* b2, const float* b3, const float* b1_next, float out_scale2,
                             float out_scale3, float out_scale1, int B, int H, int W, stm_stream_t stream);
int stm_bottleneck_chain_proj_f32(const void* mid1_planes, const void* x0_planes, void* y_planes, void* z_planes, const void* w2_packed,
                                  const void* tail_packed, const float* b2, const float* b3, const float* b1_next, float out_scale2,
                                  float out_scale3, float out_scale1, int B, int H, int W, stm_stream_t stream);

/* bytes of the packed (pre-split, pre-tiled) weight image; 0 on bad arguments */
size_t stm_conv_packed_weight_bytes(int Cout, int Cin, int kh, int kw, int planes);
/* weight [Cout, Cin, kh, kw] fp32 (torch OIHW, contiguous) -> packed image; done once per layer */
int stm_conv_pack_weights_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int planes,
                              stm_stream_t stream);
/* the same with an explicit output-channel tile width (64 or 128); planar convolution only */
size_t stm_conv_packed_weight_bytes_tiled(int Cout, int Cin, int kh, int kw, int planes, int tile_n);
int stm_conv_pack_weights_tiled_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int planes,
                                    int tile_n, stm_stream_t stream);
/* plane-format aware forms: fmt 0 = bf16 x 3 (as above), fmt 1 = fp16 x 2 / fmt 2 = fp16 x 1 with the weights multiplied by
 * the power of two `wscale` (bring max |w| to ~2^10; pass 1 / wscale as stm_conv_geom.out_scale) */
int stm_conv_pack_weights_fmt_f32(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int tile_n, int fmt,
                                  float wscale, stm_stream_t stream);
int stm_split_planes_fmt_f32(const float* x, void* planes, int64_t n_pixels, int C, int fmt, stm_stream_t stream);

/* Planar form of the same convolution: activations travel between layers ALREADY split, as bf16 planes in
 * channel-slab-major order [planes][C/32][pixels][32]: the 32-channel slab of one pixel is one 64-byte line and
 * consecutive pixels are consecutive lines (a K-slab's activation tile, and its shifted re-reads by the other taps, are
 * dense in memory and in L2); element (plane p, pixel i, channel c) sits at
 * ((p * plane_stride) + ((c/32) * np + i) * 32 + c%32) * 2 bytes; fp32 value = sum of the planes, exactly.  Staging a K-slab is then pure LDS-DMA -- each element is split once, by its producer's epilogue, instead of
 * kh*kw*(Cout/128) times by its consumers.  stm_split_bf16_planes_f32 enters the format from an fp32 NHWC tensor (C % 32 == 0);
 * the convolution writes fp32 NHWC (out_f32), three planes (out_planes), or both; the residual may be given in either
 * form.  g->planes selects how many input planes take part in the products (3 or 2); outputs always carry three. */
int stm_split_bf16_planes_f32(const float* x /* [n_pixels][C] fp32 */, void* planes /* [3][C/32][n_pixels][32] */,
                              int64_t n_pixels, int C, stm_stream_t stream);
int stm_conv2d_planar_f32(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                          const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu,
                          stm_stream_t stream);
/* the same with a scratch buffer: launches whose grid would leave most CUs idle over a long K (few pixels, many input
 * channels) are split along K, the parts' fp32 partial sums go through `workspace` and a finishing kernel adds them and
 * runs the epilogue; splitk * pixels * ceil(Cout / tile_n) * tile_n * 4 bytes are needed, otherwise (or with NULL) the
 * launch is not split */
int stm_conv2d_planar_ws_f32(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                             const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu,
                             void* workspace, size_t workspace_bytes, stm_stream_t stream);

/* ---- tracker bookkeeping of the batched clip pipeline (rows a11, a15, a16 for many clips at once) ----------------------------
 * The reference keeps one Track_TF per video and spends ~140 tiny torch launches and 4 host syncs per frame on row gathers,
 * concatenations and score arithmetic (track_TF.py:50-181, TF_utils.py:12-51,99-120, detection_TF.py:120-134).  With the
 * state of all clips concatenated (rows sorted by clip) each of those chains is one launch here; fp32 arithmetic in the
 * reference's operand order, bit-equal to the torch chains (tests/test_gpu_tracker.py).
 *
 * stm_gather_detections_f32: the Fast-NMS survivors of stm_detect_cc_f32 (idx / cls / score / box [B,top_k(,4)], count [B] on
 *   the device) -> concatenated detection rows: box, class, score, mask_coeff[idx], track[idx], centerness[idx], clip id.  D =
 *   sum(count) as the host read it (rows past D are not written).
 * stm_shift_rois_f32: RoIs of CandidateShift: (clip, sanitize_coordinates(box) in feature-map pixels) (box_utils.py:320-337).
 * stm_shift_apply_f32 (in place): box = decode(loc_shift, center_size(box)), coeff += coeff_shift, score *= decay
 *   (TF_utils.py:40-48; box_utils.py:25-35,238-283).
 * stm_match_scores_f32: compute_comp_scores + argmax (TF_utils.py:99-120, track_TF.py:104-129) over [dummy | prev rows of the
 *   same clip]: cos [D,Pn] raw embedding dot products, mask_iou [D,Pn]; box IoU and class equality are computed here;
 *   coeff4 = cfg.match_coeff (host floats); match[d] = 0 (new) or 1 + prev row.  prev_offsets [B+1]: clip row ranges.
 *   stm_match_scores_embed_f32: the same from the embedding tables det_track [D,E] / prev_track [Pn,E] (track_TF.py:99-101's
 *   matrix product restricted to the same-clip pairs, one fmaf chain over E per pair) instead of a precomputed cos.
 * stm_gather_rows2: out_t[r] = plan[r] < n_a ? a_t[plan[r]] : b_t[plan[r] - n_a] for up to 8 row tensors in one launch (the
 *   tracker update cat(prev, det).index_select(plan), track_TF.py:132-156); row_bytes multiples of 4.
 * stm_pack_tracked_f32: keep rule (tracked <= max_age, more than one mask pixel > 0.5, score > thr: track_TF.py:158-165) and
 *   the fixed-shape output [B, top_k, cols] = (box 4, score, class, object id, valid, mask_coeff ...), zero padded.
 *   keep_ws: n_rows ints of scratch. */
int stm_gather_detections_f32(const int64_t* idx, const int64_t* cls, const float* score, const float* box, const int* count,
                              const float* mask_coeff, const float* track, const float* centerness, int B, int top_k, int N,
                              int mask_dim, int embed_dim, int D, float* out_box, int64_t* out_cls, float* out_score,
                              float* out_coeff, float* out_track, float* out_centerness, int* out_clip, stm_stream_t stream);
int stm_shift_rois_f32(const float* box, const int* clip, float* rois, int n, int feat_h, int feat_w, stm_stream_t stream);
int stm_shift_apply_f32(const float* loc_shift, const float* coeff_shift, float* box, float* coeff, float* score, int n,
                        int mask_dim, float score_decay, stm_stream_t stream);
int stm_match_scores_f32(const float* cos, const float* mask_iou, const float* det_box, const float* prev_box,
                         const float* det_score, const int64_t* det_cls, const int64_t* prev_cls, const int* det_clip,
                         const int* prev_offsets, int D, int Pn, const float* coeff4, float dummy_iou, int* match,
                         stm_stream_t stream);
int stm_match_scores_embed_f32(const float* det_track, const float* prev_track, int embed_dim, const float* mask_iou,
                               const float* det_box, const float* prev_box, const float* det_score, const int64_t* det_cls,
                               const int64_t* prev_cls, const int* det_clip, const int* prev_offsets, int D, int Pn,
                               const float* coeff4, float dummy_iou, int* match, stm_stream_t stream);
int stm_gather_rows2(const void* const* a_rows, const void* const* b_rows, void* const* out_rows, const int* row_bytes,
                     int n_tensors, const int* plan, int n_rows, int n_a, stm_stream_t stream);
int stm_pack_tracked_f32(const float* mask, const float* score, const int* tracked, const int* offsets, const float* box,
                         const int64_t* cls, const float* mask_coeff, int n_rows, int hw, int B, int top_k, int cols,
                         int mask_dim, int max_age, float score_thr, int* keep_ws, float* out, stm_stream_t stream);
/* the same with the keep rule's pixel count taken from the masks' bit words (stm_lincomb_sigmoid_crop_bits_f32: [n_rows][words],
 * bit = value > 0.5): the soft masks are not read */
int stm_pack_tracked_bits_f32(const uint64_t* mask_bits, int words, const float* score, const int* tracked, const int* offsets,
                              const float* box, const int64_t* cls, const float* mask_coeff, int n_rows, int B, int top_k,
                              int cols, int mask_dim, int max_age, float score_thr, int* keep_ws, float* out,
                              stm_stream_t stream);

/* ---- `_f16` entry points: the genuine-fp16 convolution path of BASELINE config 5 ("fp16 MFMA backbone convs") -----------------
 * Replaces, for the ResNet backbone of a half-precision deployment, the same reference layers as the fp32-equivalent entries
 * above (backbone.py:38-58 Bottleneck 1x1 / 3x3 / downsample, backbone.py:20-26,45 DCN).  One fp16 plane per tensor
 * ([1][C/32][pixels][32] _Float16: plane format 2), weights as one fp16 plane scaled by the power of two `wscale`, ONE
 * v_mfma_f32_16x16x32_f16 product per reference product, fp32 accumulation, fp32 bias / residual / ReLU in the epilogue.
 * Stated tolerance: |y - y_fp64| <= 1e-3 * sum |x w| (each operand rounded to 11 bits).  g->fmt / g->planes are ignored (forced
 * to 2 / 1); g->out_fmt_plus1 = 2 makes the layer write BOTH planes of the fp16x2 format (hand-over to a fp32-equivalent
 * consumer).  Range guard as for fmt 1 (stm_planar_set_range_flag). */
int stm_split_planes_f16(const float* x, void* planes, int64_t n_pixels, int C, stm_stream_t stream);
int stm_conv_pack_weights_f16(const float* weight, void* packed, int Cout, int Cin, int kh, int kw, int tile_n, float wscale,
                              stm_stream_t stream);
int stm_conv2d_planar_f16(const void* x_planes, const void* packed_weight, const float* bias, const float* residual_f32,
                          const void* residual_planes, float* out_f32, void* out_planes, const stm_conv_geom* g, int relu,
                          void* workspace, size_t workspace_bytes, stm_stream_t stream);
int stm_dcn_sample_planar_f16(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np,
                              long long out_plane_stride, const stm_deform_geom* g, stm_stream_t stream);

/* ---- frame pre-processing on the device (row f3) ------------------------------------------------------------------
 * Replaces the host chain of eval.py:703-717 (evaluate_single): mmcv.imresize(im, (w, h)) [cv2.resize INTER_LINEAR on
 * the uint8 HWC image] -> (im - MEANS) / STD [numpy float64] -> mmcv.impad_to_multiple(im, 32) -> permute(2,0,1).float().
 * img [n, H0, W0, 3] uint8 (channel order as read), out [n, 3, Hp, Wp] fp32 (zero outside [h, w]); mean/std: 3 doubles
 * in the image's channel order.  mode 0: no normalisation; 1: (v-mean)/std; 2: v-mean; 3: v/255
 * (cfg.backbone.transform.{normalize, subtract_means, to_float}).  Bit-exact against oracle orc_preprocess_u8. */
int stm_preprocess_u8_f32(const uint8_t* img, float* out, int n, int H0, int W0, int h, int w, int Hp, int Wp,
                          const double* mean, const double* stdv, int mode, stm_stream_t stream);

/* ---- head output assembly (row a5 / f4) -----------------------------------------------------------------------------
 * Replaces the cat / view / tanh / F.normalize tail of PredictionModule_FC.forward (prediction_head_FC.py:168-195) for the
 * planar head: small[k] [pixels, small_ld] holds, per kernel shape k, the conf | centerness+bbox | mask groups (group_pad
 * channels each, real ones first), trk[k] [pixels, trk_ld] the track embedding; pixel axis = levels concatenated, level l =
 * B images of lvl_hw[l] pixels starting at lvl_start[l].  Outputs (N = K * sum lvl_hw): conf [B,N,n_cls], loc [B,N,4],
 * mask [B,N,mask_dim], track [B,N,embed_dim] L2-normalised, centerness [B,N,1] = tanh, the latter in the reference's
 * (level, k, pixel) order (it concatenates centerness along H), the others in (level, pixel, k) order.  One prior per
 * kernel shape (num_priors == 1, as every STMask config has). */
typedef struct stm_head_layout {
    int B, K, n_levels, n_cls, mask_dim, embed_dim, group_pad, small_ld, trk_ld;
    int lvl_start[8], lvl_hw[8];
} stm_head_layout;
int stm_head_assemble_f32(const float* const* small, const float* const* trk, const stm_head_layout* layout, float* conf,
                          float* loc, float* mask, float* track, float* centerness, stm_stream_t stream);

/* ---- deformable sampling into the planar format (row a1, inference graph) ------------------------------------------
 * The im2col half of dcn_v2.DCN (backbone.py:20-26,45) for activations that already live in the planar graph: x is fp32
 * NHWC [B,H,W,C] (C = 128 / 256 / 512), offset_mask the raw conv_offset_mask output pixel-major [B*Ho*Wo, om_ld] (18
 * offsets dy,dx per tap, then 9 mask logits; sigmoid applied here), the columns are written as bf16 planes
 * [3][9C/32][out_np][32] with K index = tap*C + channel, i.e. the input of stm_conv2d_planar_f32 as a 1x1 convolution over
 * 9C channels with the DCN weight reordered to [O][tap][C].  Sampling arithmetic and border rule are those of
 * stm_deform_im2col_f32, value for value. */
int stm_dcn_sample_planar_f32(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np,
                              long long out_plane_stride, const stm_deform_geom* g, stm_stream_t stream);
/* F.interpolate(x, size=(Ho, Wo), mode="bilinear", align_corners=False) of an fp32 NHWC tensor x [B][H][W][C], written
 * directly as planes [P][C/32][B*Ho*Wo][32] (fmt as in stm_conv_geom) for the next planar convolution: the proto-net's
 * InterpolateModule (make_net.py:31-40, yolact's x2 upsample between its convolutions).  C % 32 == 0. */
int stm_resize_bilinear_planes_f32(const float* x, void* planes, int B, int H, int W, int C, int Ho, int Wo, int fmt,
                                   stm_stream_t stream);

/* ResNet stem tail, backbone.py:73 `maxpool(relu(bn1(conv1(x))))` after BN folding: x = the raw convolution output, fp32 NHWC
 * [B][H][W][C]; planes [P][C/32][B*Ho*Wo][32] = relu(maxpool3x3/s2/p1(x) + bias[c]) (exactly relu-bias-then-pool: both are
 * monotone per channel), Ho = (H-1)/2+1, Wo = (W-1)/2+1.  bias may be NULL.  C % 32 == 0. */
int stm_bias_relu_maxpool_planes_f32(const float* x, const float* bias, void* planes, int B, int H, int W, int C, int fmt,
                                     stm_stream_t stream);

/* CandidateShift's RoI features (TF_utils.py:30-39: feats = relu(cat([corr, T2S_prev, T2S], 1)); mmcv roi_align(feats, rois,
 * 7, spatial_scale 1, sampling_ratio 0, aligned) -- bbox_feat_extractor) as planes for TemporalNet's first convolution
 * (track_to_segment_head.py:20): t2s_prev / t2s fp32 NHWC [B][H][W][C1], corr fp32 NCHW [B][Cc][H][W], rois [n][5] =
 * (image, x1, y1, x2, y2).  planes [P][Cpad/32][n*PH*PW][32], Cpad = 2*C1 + Cc rounded up to 32, pixel = (roi, py, px),
 * channel order [T2S_prev | T2S | corr | zeros] -- the consumer's weights are permuted to match.  Same arithmetic as
 * stm_roi_align_avg_f32 on the concatenated map. */
int stm_roi_align_planes_f32(const float* t2s_prev, const float* t2s, const float* corr, const float* rois, void* planes, int B,
                             int H, int W, int C1, int Cc, int n, int PH, int PW, int fmt, stm_stream_t stream);
/* the same with the correlation volume channels-last, [B][H][W][corr_ld] as stm_corr_patch_nhwc_f32 writes it (corr_ld a multiple
 * of 4, >= Cc rounded up to 8; 0 = the NCHW form above): a sample's Cc displacement channels are then 4 cache lines instead of Cc */
int stm_roi_align_planes_nhwc_f32(const float* t2s_prev, const float* t2s, const float* corr, int corr_ld, const float* rois,
                                  void* planes, int B, int H, int W, int C1, int Cc, int n, int PH, int PW, int fmt,
                                  stm_stream_t stream);

/* Stem entry (backbone.py:73 / resnet conv1: kh x kw convolution, stride s, on a Cin-channel frame with kw * Cin <= 32): x fp32
 * NHWC [B][H][W][Cin] -> planes R [P][1][B*H*Wo][32], Wo = (W + 2 pw - kw) / sw + 1, R[b][y][ox][j] = x[b][y][sw*ox - pw + j / Cin]
 * [j % Cin] for j < kw * Cin (zero outside the frame and for j >= kw * Cin).  The stem is then the (kh x 1), stride (sh, 1),
 * padding (ph, 0) planar convolution over R with weights w'[o][j][ky][0] = w[o][j % Cin][ky][j / Cin]. */
int stm_stem_rows_planes_f32(const float* x, void* planes, int B, int H, int W, int Cin, int kw, int sw, int pw, int fmt,
                             stm_stream_t stream);

/* The whole ResNet stem in one kernel (backbone.py:61-75: conv1 7x7 / stride 2 / padding 3 on a 3-channel frame, eval BatchNorm
 * folded into weight and bias, ReLU, MaxPool2d(3, 2, 1)): x fp32 NHWC [B][H][W][3] -> planes [P][Cout/32][B*Hp*Wp][32] in format
 * out_fmt, Hp = ((H - 1) / 2) / 2 + 1 (conv then pool, floor mode).  Cout = 64, fmt 1 or 2 (fmt 2 may write out_fmt 1).  No
 * intermediate tensor: the input patch of a tile of pooled pixels is split into fp16 planes in LDS, the conv positions under the
 * pool windows are computed on the matrix cores (three products per reference product in fmt 1) and pooled from LDS.
 * Weights: stm_stem_pack_weights_f32 of the OIHW tensor [64, 3, 7, 7] * wscale (a power of two); out_scale = 1 / wscale. */
size_t stm_stem_packed_weight_bytes(int Cout, int fmt);
int stm_stem_pack_weights_f32(const float* weight, void* packed, int Cout, int fmt, float wscale, stm_stream_t stream);
int stm_stem_fused_f32(const float* x, const void* packed_weight, const float* bias, void* out_planes, int B, int H, int W, int Cout,
                       int fmt, int out_fmt, float out_scale, stm_stream_t stream);

/* fp16 plane formats (fmt 1, fmt 2) range guard.  A value with |x| > 65504 (or inf / nan) has no fp16 plane representation and
 * would poison the following layers silently (inf - inf = nan, and a ReLU epilogue maps nan to 0).  Every producer of
 * fp16 planes (stm_split_planes_fmt_f32, the stm_conv2d_planar_* epilogues, stm_dcn_sample_planar_fmt_f32) therefore
 * writes 1 to *device_flag when it meets one.  The flag is sticky: the caller zeroes it, reads it with whatever
 * device-to-host read it does anyway, and must discard the results if it is set.  NULL (the initial state) disables
 * the guard.  The registration is PER DEVICE: it applies to the device that is current when this is called, and kernels
 * launched on device d raise device d's flag (a process driving several GPUs registers one flag on each). */
int stm_planar_set_range_flag(int* device_flag);

/* General form: x is pixel-major with a pixel stride of x_ld floats (>= C: a channel slice of a wider NHWC tensor is
 * fine), has_mask = 1 the modulated form above (offsets then mask logits, dcn_v2), has_mask = 0 mmcv's DeformConv2d as
 * FeatureAlign uses it (Featurealign.py:44-58; 2K offsets per pixel, 3x3 / 3x5 / 5x3 taps, C = 256).  The columns of the
 * g->B * Ho * Wo output pixels are written at pixels [out_pixel_offset, ...) of planes with out_np pixels per slab, so the
 * FPN levels of a shared head can share one column buffer.  Tap-major K order: k * C + c. */
int stm_deform_sample_planar_f32(const float* x, int x_ld, const float* offsets, int om_ld, int has_mask, void* planes, int out_np,
                                 int out_pixel_offset, long long out_plane_stride, const stm_deform_geom* g, int fmt,
                                 stm_stream_t stream);

/* plane-format aware form (fmt as in stm_conv_geom: 0 = bf16 x 3, 1 = fp16 x 2) */
int stm_dcn_sample_planar_fmt_f32(const float* x, const float* offset_mask, int om_ld, void* planes, int out_np,
                                  long long out_plane_stride, const stm_deform_geom* g, int fmt, stm_stream_t stream);

/* ---- fused deformable convolution for the planar graph (rows a1 / a7): sampler + plane split + matrix product in ONE kernel ----
 * Replaces the pair stm_deform_sample_planar_f32 (columns as planes, 78 % of a DCN layer's bytes) + stm_conv2d_planar_f32 over
 * taps*C channels, i.e. dcn_v2.DCN.forward (backbone.py:20-26,45; has_mask = 1: 18 offsets then 9 mask logits per pixel, sigmoid
 * applied here, bias, the bottleneck's ReLU) and mmcv.ops.DeformConv2d as FeatureAlign calls it (Featurealign.py:27-31,72;
 * has_mask = 0, 3x3 / 3x5 / 5x3, no bias).  x fp32 pixel-major [B*H*W, x_ld >= C]; offsets fp32 [B*Ho*Wo, om_ld]; packed_weight =
 * stm_conv_pack_weights_fmt_f32 of the ORIGINAL [Cout][C][kh][kw] weight at tile_n 128 in format fmt (1: fp16 x 2, 2: fp16 x 1),
 * out_scale = 1 / its wscale; the output leaves as planes [P][Cout/32][out_np][32] in out_fmt (= fmt, or 1 under fmt 2) at pixels
 * [out_pixel_offset, out_pixel_offset + B*Ho*Wo).  The sampled values are those of stm_deform_sample_planar_f32 bit for bit (same
 * corner weights, same blend, same split); the products and their order per K-slab are those of stm_conv2d_planar_f32, with the K
 * axis ordered channel-slab outer / tap inner.  No column buffer exists: algorithmic bytes = input + offsets + output + weights
 * (fused form).  One deformable group, <= 15 taps (9 with mask), C % 64 == 0, Cout % 128 == 0
 * (stm_deform_conv_fused_planar_supported says so without raising an error). */
int stm_deform_conv_fused_planar_supported(const stm_deform_geom* g, int Cout, int has_mask, int fmt);
int stm_deform_conv_fused_planar_f32(const float* x, int x_ld, const float* offsets, int om_ld, int has_mask, const void* packed_weight,
                                     const float* bias, void* out_planes, int out_np, int out_pixel_offset, long long out_plane_stride,
                                     int Cout, int relu, float out_scale, const stm_deform_geom* g, int fmt, int out_fmt,
                                     stm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* STMASK_HIP_H_ */
