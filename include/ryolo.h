/*
 * include/ryolo.h -- C ABI of libryolo_hip.so, the MI355X (gfx950) hot path of rotated-YOLOv3.
 *
 * Drop-in boundary.  The reference has exactly one FFI on this path, the pybind11 module `r_nms`
 * (utils/nms/src/rotate_polygon_nms.cpp:7-16, built by utils/nms/setup.py:4-13).  Everything else on the path
 * is reached through PyTorch operator calls (cuDNN conv / BN / PReLU in model/models.py:55-66, elementwise
 * decode in model/models.py:198-221).  Each entry point below names the reference interface it replaces.
 *
 * Conventions
 *   - plain C: pointers are DEVICE pointers unless a parameter says "host"; sizes are element counts;
 *     `stream` is a hipStream_t passed as void* (NULL = the null stream).  No torch types.
 *   - every function only ENQUEUES work on `stream` and returns; nothing is allocated inside (the caller
 *     provides workspaces whose size the *_workspace_bytes functions report), so calls are hipGraph-capturable.
 *   - return value: 0 = RYOLO_OK, otherwise a negative RYOLO_E* code (ryolo_strerror gives the text).
 *     There is NO CPU fallback: without a gfx950 device the launch fails and the error is returned.
 */
#ifndef RYOLO_H
#define RYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RYOLO_OK 0
#define RYOLO_EINVAL -1   /* bad argument (null pointer, negative size, unsupported shape) */
#define RYOLO_ELAUNCH -2  /* HIP reported an error while enqueuing */
#define RYOLO_ETOOBIG -3  /* n above RYOLO_RNMS_MAX_BOXES */

#define RYOLO_RNMS_MAX_BOXES 262144

const char *ryolo_strerror(int code);
/* ABI version; bumped when a signature changes. */
int ryolo_abi_version(void);
/* "RYOLO_BUILD_ID=" + 16 hex digits: a hash of the library's sources, headers and compiler flags (__graft_entry__.source_id()). */
const char *ryolo_build_id(void);
/* The library's six tuning switches (dispatch selections for tests and A/B timing; none of them changes results beyond fp32 summation
 * order): RYOLO_CONV3X3 = mp | mq, RYOLO_CONV1X1 = igemm, RYOLO_RNMS_TILES = 1 | 2 (column tiles per wave of the NMS mask kernel),
 * RYOLO_MQ_KORDER = 0 | 1, RYOLO_BN_REDUCE_TILES = 0, RYOLO_STEM_DGRAD = 0..3.  Each is read from the environment once, at its first use; this call sets (value) or clears (NULL / "") one in
 * the running process.  Not synchronised with launches in flight on other threads.  RYOLO_EINVAL: unknown name. */
int ryolo_set_tuning(const char *name, const char *value);

/* ------------------------------------------------------------------------------------------------
 * Rotated NMS  -- replaces r_nms(dets, threshold) of utils/nms/src/rotate_polygon_nms.cpp:7-12
 *                 (-> nms_cuda, rotate_polygon_nms_kernel.cu:323-384; tile kernel :262-308; IoU :22-260).
 *
 *   dets        [n, >=6] float32 rows (cx, cy, w, h, angle_rad, score); `row_stride` = floats between rows
 *               (a column slice dc[:, :6] of an [n,8] tensor is passed as-is with row_stride 8, cf.
 *               utils/nms/nms.py:64; the reference compacts it with index_select, kernel.cu:328).
 *   thr         IoU threshold, suppression iff IoU > thr (strict, kernel.cu:301).
 *   keep_out    [n] int64, receives the kept ORIGINAL row indices in ascending order (kernel.cu:380-383).
 *   num_keep    [1] int32, receives K.
 *   workspace   ryolo_rnms_workspace_bytes(n) bytes of device scratch (contents irrelevant on entry).
 *
 * Semantics: sort by score descending (stable: ties keep the lower index first); box j is suppressed iff a
 * kept box i earlier in that order has devRotateIoU(box_i, box_j) > thr, box_i as first argument.
 * Bit-exact against oracle/riou_oracle.c (which is pinned to the reference arithmetic, tests/golden).
 * The reference's blocking D2H copy of the whole n x n/64 bit matrix and its host scan (kernel.cu:352-376)
 * are replaced by an on-device scan; only K (4 bytes) ever needs to reach the host.
 * Calls of more than 20 416 boxes run part of their work on a second, library-owned stream (the scan of the first block rows beside
 * the IoU kernel of the last ones); HIP events order it inside the call's position on `stream` -- for the caller the call is still
 * "enqueued on `stream`": everything enqueued on `stream` before it is visible to it, everything after it sees its results.
 */
size_t ryolo_rnms_workspace_bytes(int n);
int ryolo_rnms(const float *dets, int n, int row_stride, float thr, int64_t *keep_out, int32_t *num_keep,
               void *workspace, size_t workspace_bytes, void *stream);

/* Segmented rotated NMS: `num_segments` independent sets stored back to back in `dets` (segment s = rows
 * [seg_offsets[s], seg_offsets[s+1]), device int32[num_segments+1]), each ALREADY sorted by score descending -- the
 * order in which the greedy scan visits a set (utils/nms/nms.py:57-66 sorts each image's class before calling r_nms).
 * One launch of each kernel covers all sets (a batch of images x classes).  keep_flags[i] = 1 iff row i survives in
 * its set.  max_seg_len >= the longest segment (host value; sizes the grid and the workspace).  Same arithmetic and
 * the same bit-exact bar as ryolo_rnms. */
size_t ryolo_rnms_segmented_workspace_bytes(int m, int num_segments, int max_seg_len);
int ryolo_rnms_segmented(const float *dets, int m, int row_stride, const int32_t *seg_offsets, int num_segments,
                         int max_seg_len, float thr, unsigned char *keep_flags, void *workspace, size_t workspace_bytes,
                         void *stream);

/* Measurement hook (bench.py's nms.roofline): while `device_counter` is non-NULL every subsequent ryolo_rnms /
 * ryolo_rnms_segmented call adds the number of box pairs whose exact polygon IoU it evaluated (pairs that survive the
 * bounding-circle reject) to *device_counter (uint64, zeroed by the caller).  NULL switches it off.  Not thread safe. */
void ryolo_rnms_count_pairs(uint64_t *device_counter);

/* ------------------------------------------------------------------------------------------------
 * Rotated IoU -- the arithmetic of devRotateIoU (kernel.cu:251-260) exposed directly; replaces the
 * per-pair Python/shapely loop of skew_bbox_iou (utils/utils.py:290-320) used by test.py:146.
 *   ryolo_riou_pairs : out[i]        = IoU(b1[i], b2[i])           i < n
 *   ryolo_riou_matrix: out[i*n2 + j] = IoU(b1[i], b2[j])           i < n1, j < n2
 * Rows are (cx, cy, w, h, angle_rad, ...) float32 with the given row strides (floats).
 */
int ryolo_riou_pairs(const float *b1, int stride1, const float *b2, int stride2, int n, float *out, void *stream);
int ryolo_riou_matrix(const float *b1, int n1, int stride1, const float *b2, int n2, int stride2, float *out,
                      void *stream);

/* Rotated IoU of the EVALUATION path -- replaces skew_bbox_iou (utils/utils.py:290-320: get_rotated_coors :702-725 + skewiou
 * :663-699, a per-pair Python loop over shapely polygon intersections in fp64) as called by test.py:146 for mAP matching.
 * Corners in fp64 with get_rotated_coors' rotation matrix, exact convex-polygon intersection (fp64 Sutherland-Hodgman clip),
 * inter / (area1 + area2 - inter), 0 when either area or the union is 0.  Differs from ryolo_riou_* (the NMS kernel's fp32
 * arithmetic, kept bit for bit) exactly where that arithmetic is not the geometric IoU: e.g. IoU(A, A) is 1 here.
 * Tolerance vs oracle/poly_iou.py: 1e-6 absolute. */
int ryolo_skew_iou_pairs(const float *b1, int stride1, const float *b2, int stride2, int n, float *out, void *stream);
int ryolo_skew_iou_matrix(const float *b1, int n1, int stride1, const float *b2, int n2, int stride2, float *out,
                          void *stream);


/* ------------------------------------------------------------------------------------------------
 * Convolution block -- replaces the operator chain the reference builds per `convolutional` cfg block,
 * nn.Conv2d -> nn.BatchNorm2d -> nn.PReLU (model/models.py:49-66), plus the `shortcut` add
 * (model/models.py:281-282) and the nearest `upsample` + `route` concat write (model/models.py:93-94,
 * :269-278) when they consume this block's output:
 *
 *     y = upsample( act( conv(x, W) * scale[c] + shift[c] ) + residual )
 *
 * Activations are NHWC bf16; a tensor may be a channel slice of a wider buffer (pixel stride `*_cstride`
 * elements >= its channel count), which is how route/concat outputs are written in place.
 * scale/shift: fp32 [ryolo_conv_cpad(Cout)] (eval-mode BatchNorm folded: scale = gamma/sqrt(var+eps),
 * shift = beta - mean*scale, cf. utils/torch_utils.py:45-69; or scale = 1, shift = bias for the head convs),
 * padded with zeros.  Arithmetic: bf16 x bf16 products accumulated in fp32 on MFMA; scale/shift/act in
 * fp32; rounded to bf16; the residual is added in fp32 and rounded to bf16 again (the value a layer-by-layer
 * bf16 execution of the reference produces).  Tolerance vs an fp32 reference on the same bf16 inputs: 2 bf16 ulp.
 * All pointers 16-byte aligned; Cin, Cout and the channel strides multiples of 8; ksize 1 (pad 0) or 3
 * (Cin a power of two or a multiple of 64); upsample 1 or 2.
 */
#define RYOLO_ACT_LINEAR 0
#define RYOLO_ACT_LEAKY 1 /* x > 0 ? x : slope * x  -- LeakyReLU(0.1) and the reference's PReLU(1) (models.py:63-66) */
#define RYOLO_ACT_MISH 2  /* x * tanh(softplus(x))  -- extension named by the north star, not in the reference */

typedef struct ryolo_conv_desc {
    int N, H, W;        /* input batch / height / width */
    int Cin, Cout;      /* channels as seen by the kernel (first layer: Cin padded 3 -> 8) */
    int ksize, stride, pad;
    int in_cstride, out_cstride, res_cstride; /* pixel strides in elements */
    int act;            /* RYOLO_ACT_* */
    float slope;
    int upsample;       /* 1, or 2: y has spatial size 2Ho x 2Wo, every result written to its 2x2 block */
    int tile;           /* low byte: 0 = auto; 1 = 128x128 (8 waves), 2 = 256x64, 3 = 256x32, 4 = 256x128 3-stage, 6 / 7 = 128x128 with
                         * 8 waves as 4x2 / 4 waves as 2x2 (pixels x channels per workgroup); 8 / 11 / 14 = the 256-channel
                         * multi-phase tile of conv_mp.hip with 256 / 192 / per-shape pixel rows, 9 = the 128 x 256 tile of
                         * conv_mq.hip; 12 = the stem kernel of conv_stem.hip (3x3, 32 -> 64 channels only), 13 = the weight-stationary 1x1 kernel of conv_pw.hip (two 4-wave workgroups per CU; same bits as 8 / 11); auto picks between them per launch
                         * (environment RYOLO_CONV3X3=mp|mq forces one family).  Test / tuning bits: 0x100 general address
                         * path, 0x200 never the persistent grid, 0x800 persistent grid also for 3x3;
                         * ryolo_conv2d_wgrad: 0x2000 the two-stage square tile instead of the three-stage tiles, 0x4000 the
                         * transposed 128 x 256 tile, bits 16+ : forced split count */
} ryolo_conv_desc;

/* bytes of the packed bf16 weight image [cpad(Cout)][kpad(ksize*ksize*Cin_pad)] (+ zero tail) */
size_t ryolo_conv_packed_weight_bytes(int Cout, int Cin_pad, int ksize);
/* w_oihw: fp32 [Cout][Cin][ksize][ksize] (the nn.Conv2d.weight layout, models.py:55) -> packed image */
int ryolo_conv_pack_weights(const float *w_oihw, int Cout, int Cin, int ksize, int Cin_pad, void *packed,
                            void *stream);
int ryolo_conv2d_bn_act(const ryolo_conv_desc *desc /* host */, const void *x, const void *w_packed,
                        const float *scale, const float *shift, const void *residual /* may be NULL */, void *y,
                        void *stream);
/* Two consecutive `convolutional` blocks (model/models.py:49-66, twice) in ONE launch when the tensor between them has no other
 * reader (inference): y = block_second(block_first(x)) [+ x when shortcut_from_input: the `shortcut` of models.py:281-282 whose
 * `from` is the first block's input].  The intermediate tensor is computed into LDS, rounded to bf16 as if it had been stored, and
 * never reaches HBM; the result is bit-identical to two ryolo_conv2d_bn_act calls.  Served pair (ryolo_conv_pair_supported):
 * Darknet-53 layers 2-4 (1x1 64->32, then 3x3 32->64 + shortcut).
 * Descriptors as for ryolo_conv2d_bn_act; second->out_cstride is y's pixel stride; no residual operand besides the shortcut. */
int ryolo_conv_pair_supported(const ryolo_conv_desc *first, const ryolo_conv_desc *second, int shortcut_from_input);
int ryolo_conv2d_bn_act_pair(const ryolo_conv_desc *first, const ryolo_conv_desc *second, const void *x, const void *w_first,
                             const float *scale_first, const float *shift_first, const void *w_second, const float *scale_second,
                             const float *shift_second, int shortcut_from_input, void *y, void *stream);
/* Which kernel ryolo_conv2d_bn_act (with_statistics: ryolo_conv2d_bn_act_stats) would launch for this descriptor on the current
 * device -- a dry run of the dispatch, nothing is enqueued.  For benchmarks and profiles (the per-kernel tables name the kernel
 * that actually ran); -1 for an invalid descriptor. */
#define RYOLO_CONV_KERNEL_MP256 1   /* conv_mp.hip, 256 pixels x 256 channels, one 8-wave workgroup per CU */
#define RYOLO_CONV_KERNEL_MP192 2   /* conv_mp.hip, 192 x 256 */
#define RYOLO_CONV_KERNEL_MQ 3      /* conv_mq.hip, 128 x 256, two 4-wave workgroups per CU */
#define RYOLO_CONV_KERNEL_DIRECT8 4 /* first layer (C_in 3 -> 8), fragments straight from global memory */
#define RYOLO_CONV_KERNEL_STEM 5    /* conv_stem.hip: 3x3, 32 -> 64 channels, stride 1 / 2: input patch staged once, filter in registers */
#define RYOLO_CONV_KERNEL_PW 6      /* conv_pw.hip: 1x1 stride 1, the filter slice in registers, rows through an LDS ring */
#define RYOLO_CONV_KERNEL_STEM0 8   /* conv_stem.hip: the first layer's forward with its input patch staged in LDS (statistics passes keep DIRECT8) */
#define RYOLO_CONV_KERNEL_STEM_DGRAD 7 /* conv_stem.hip: the data gradient of a 3x3 32 -> 64 stem layer (stride 2: all four parity classes) in one launch */
#define RYOLO_CONV_KERNEL_STEM64 11 /* conv_stem.hip: 3x3 / 1, 64 -> 128: input patch staged once, the filter split over the waves' registers */
#define RYOLO_CONV_KERNEL_MQ128 9   /* conv_mq.hip, 128 pixels x 128 channels (round 5): layers / data gradients with C_out % 256 != 0 */
#define RYOLO_CONV_KERNEL_MQ64 10   /* conv_mq.hip, 64 pixels x 128 channels: short tile lists (1x1 layers at 19^2 / 38^2) */
#define RYOLO_CONV_KERNEL_IGEMM 16  /* + tile code of conv.hip's 128x128 / 256x64 / 256x32 ... tiles */
int ryolo_conv_kernel_choice(const ryolo_conv_desc *desc, int with_residual, int with_statistics);
/* The same dry run for the data gradient of `forward_desc` (ryolo_conv2d_dgrad; stride 2: the choice of the last parity class)
 * and for its weight gradient (ryolo_conv2d_wgrad): 32 / 64 / 128 = the square two-stage tile, 256 / 257 / 258 / 259 / 260 = the
 * three-stage tile 256x128 / 128x256 / 128x(3 taps x 64) / 128x128 / 64x128 (c_out x c_in), RYOLO_WGRAD_KERNEL_TAPS + v = the stem's per-tap kernels.
 * bench.py names the kernels of its in-run train-step table through them. */
#define RYOLO_WGRAD_KERNEL_TAPS 1000
int ryolo_conv_dgrad_kernel_choice(const ryolo_conv_desc *forward_desc, int with_bn_reduce /* ryolo_conv2d_dgrad_bnreduce's choice */);
int ryolo_conv_wgrad_kernel_choice(const ryolo_conv_desc *forward_desc);
/* layout converters at the model boundary: the reference feeds NCHW fp32 images (train.py:236, detect.py:209) */
int ryolo_nchw_f32_to_nhwc_bf16(const float *x, int N, int C, int H, int W, int Cpad, void *y, void *stream);
int ryolo_nhwc_bf16_to_nchw_f32(const void *x, int N, int C, int H, int W, int cstride, float *y, void *stream);


/* Decode + score filter + compaction (no `io` tensor): the rows of ryolo_yolo_decode's `io` that the first half of
 * non_max_suppression keeps (utils/nms/nms.py:33-48: class_conf/class = max over io[6:], score = io[5]*class_conf,
 * score > conf_thres, w,h > min_wh, all finite).  Survivors are appended to cand[capacity][8] =
 * (x, y, w, h, angle, score, class_conf, class) in arbitrary order, cand_row[slot] = image * io_rows_per_image +
 * io_row_offset + row (sort on it to get the reference's order); *counter (zeroed by the caller, shared by the heads of one
 * forward) counts survivors and may exceed `capacity` (then the tail was dropped: re-run with more room). */
int ryolo_yolo_decode_filter(const void *head, int head_cstride, int bs, int ny, int nx, int na, int no,
                             const float *anchors, float stride, float context_factor, int arc, float conf_thres,
                             float min_wh, long long io_rows_per_image, long long io_row_offset, float *cand,
                             long long *cand_row, int *counter, int capacity, void *stream);

/* ------------------------------------------------------------------------------------------------
 * YOLO head decode -- replaces YOLOLayer.forward (model/models.py:183-227) and create_grids
 * (model/model_utils.py:16-35).  `head`: NHWC bf16 output of the last 1x1 conv, channel = a*no + k, no = nc+6.
 *   io [bs, io_rows_per_image, no] fp32: rows io_row_offset + (a*ny + y)*nx + x receive the decoded
 *        (x, y, w, h, angle, obj, cls...) in pixels (the three heads write into one tensor: the torch.cat of
 *        models.py:298 becomes a row offset); may be NULL when p is given and na*no and the head stride are multiples
 *        of 8 (the training forward only needs p);
 *   p  [bs, na, ny, nx, no] fp32 (may be NULL): the raw head values, the "training output" of models.py:189-194.
 *   anchors [na][3] fp32 = (w_px, h_px, angle_rad) (utils/parse_config.py:6-31 rows selected by the yolo mask);
 *   stride = img_size / grid (model_utils.py:20); context_factor = hyp['context_factor'] (models.py:207-208);
 *   arc: 0 'default*' (sigmoid obj+cls), 1 '*BCE*', 2 '*CE*' (models.py:210-218).
 */
int ryolo_yolo_decode(const void *head, int head_cstride, int bs, int ny, int nx, int na, int no,
                      const float *anchors, float stride, float context_factor, int arc, float *io,
                      long long io_rows_per_image, long long io_row_offset, float *p, void *stream);

/* A YOLO head in one launch (inference): the last 1x1 conv of a head (model/models.py:49-66 without BatchNorm: scale = 1, shift =
 * bias, linear) and YOLOLayer.forward (models.py:183-227) on its values.  The conv's results are rounded to bf16 -- the head tensor
 * ryolo_conv2d_bn_act would have stored -- into an on-chip tile and decoded there with the arithmetic of ryolo_yolo_decode; io / p
 * are bit-identical to conv + decode, and the head tensor (186 MB for the 76^2 head at bs 32) is never written or re-read.
 * Served (ryolo_conv_head_decode_supported, which runs the launch's own configuration test on the current device): C_in 256, 512 or 1024
 * (the three heads of yolov3.cfg), na*no <= 512, no 7 or 8.  Arguments as for the two calls it replaces. */
int ryolo_conv_head_decode_supported(const ryolo_conv_desc *desc, int na, int no);
int ryolo_conv_head_decode(const ryolo_conv_desc *desc, const void *x, const void *w_packed, const float *scale, const float *shift,
                           const float *anchors, int na, int no, float stride, float context_factor, int arc, float *io,
                           long long io_rows_per_image, long long io_row_offset, float *p, void *stream);

/* NHWC bf16 helpers for cfg graphs whose shortcut / upsample / route / maxpool cannot be fused into a conv
 * epilogue (model/models.py:79-94, :269-282).  Channel counts and strides multiples of 8. */
int ryolo_add_nhwc(const void *a, int a_cstride, const void *b, int b_cstride, void *y, int y_cstride,
                   long long npix, int C, void *stream);
int ryolo_upsample_nhwc(const void *x, int x_cstride, void *y, int y_cstride, int N, int H, int W, int C, int scale,
                        void *stream); /* scale 1 = slice copy */
int ryolo_maxpool_nhwc(const void *x, int x_cstride, void *y, int y_cstride, int N, int H, int W, int C, int ksize,
                       int stride, void *stream);


/* ------------------------------------------------------------------------------------------------
 * Training step of the conv block -- replaces autograd + cuDNN/ATen for nn.Conv2d / nn.BatchNorm2d (batch
 * statistics) / nn.PReLU under loss.backward() (train.py:268-282, model/models.py:49-66).
 *
 *   forward   z = conv(x, W) [+ bias]            ryolo_conv2d_bn_act_stats with scale = 1, shift = bias|0, act linear;
 *                                                with stat_part it also emits partial sums of z and z^2 per channel.
 *                                                STATISTICS REQUIRE shift == 0 (a BatchNorm conv has no bias): the rows of
 *                                                the last pixel tile past N*Ho*Wo are summed too, and they are exact zeros
 *                                                only then; a bias conv needs no statistics and passes stat_part = NULL
 *             mean, invstd, scale, shift         ryolo_bn_finalize (biased variance, eps; running stats with momentum)
 *             y = act(z*scale + shift) [+ res]   ryolo_bn_act_fwd
 *   backward  dz, dgamma, dbeta, dslope          ryolo_bn_act_bwd   (dz = scale*(g - mean(g) - xhat*mean(g*xhat)))
 *             dx (+)= conv^T(dz, W)              ryolo_conv2d_dgrad (stride 1: flipped filter; stride 2: 4 parity classes)
 *             dW  += sum_pix dz (x) x            ryolo_conv2d_wgrad (MFMA over pixels, split-K, fp32 partial tiles)
 */
/* Layer 0 (3x3 / stride 1 / pad 1, 3 -> 32 channels on the 8-channel padded NHWC input; ryolo_conv0_recompute_supported) trains
 * WITHOUT its conv output: z0 is four times the size of x and costs 27 MACs per value, so the engine recomputes it instead of
 * storing and re-reading it (4.9 GB less HBM traffic per bs-64 step):
 *   statistics            ryolo_conv2d_bn_act_stats with y = NULL (sums only; scale = 1, shift = 0)  ->  ryolo_bn_finalize
 *   y = act(BN(z0))       ryolo_conv0_bn_act_fwd   (batch statistics folded into scale / shift; `slope`: device scalar, PReLU)
 *   dz0, dgamma, ...      ryolo_conv0_bn_bwd       (reduce pass and apply pass both recompute z0 from x; same arguments and results
 *                                                   as ryolo_bn_act_bwd on the stored z0; workspace zeroed by the caller or
 *                                                   workspace_is_zero = 0; it is left zeroed)
 * Replaces nn.Conv2d + nn.BatchNorm2d + nn.PReLU of module_list[0] under autograd (model/models.py:49-66). */
int ryolo_conv0_recompute_supported(const ryolo_conv_desc *desc);
int ryolo_conv0_bn_act_fwd(const ryolo_conv_desc *desc, const void *x, const void *w_packed, const float *scale, const float *shift,
                           int act, const float *slope, void *y, void *stream);
size_t ryolo_conv0_bn_bwd_workspace_bytes(void);
int ryolo_conv0_bn_bwd(const ryolo_conv_desc *desc, const void *x, const void *w_packed, const void *dy, int dy_cstride,
                       const float *scale, const float *shift, const float *mean, const float *invstd, int act, const float *slope,
                       void *dz, int dz_cstride, float *dgamma, float *dbeta, float *dslope, void *workspace, size_t workspace_bytes,
                       int workspace_is_zero, void *stream);
/* Layer 0's WHOLE backward in one pass over dy (csrc/conv0_bwd.hip): the weight gradient of a layer without a data gradient is linear
 * in dz, so dz = scale * (g - S1/M - xhat * S2/M) never has to exist --
 *     dW[co][k] = scale_co * (G - (S1/M) Sx - (S2 invstd / M) (Z - mean Sx)),  G = sum_p g x, Z = sum_p z x, Sx = sum_p x
 * and one kernel that recomputes z from x accumulates S1, S2, S3 (slope), G, Z, Sx; two small kernels finish.  Replaces
 * ryolo_conv0_bn_bwd + ryolo_conv2d_wgrad for that layer (8.6 GB -> 1.9 GB of traffic at bs 64 / 608^2).  dgamma / dbeta / dslope are
 * accumulated into, grad_oihw ([32][cin_real][3][3] fp32) accumulated into or overwritten; workspace: any contents.  Partial rows
 * are per workgroup and summed in a fixed order: bit-reproducible.  Autograd of /root/reference/model/models.py:49-66, layer 0. */
size_t ryolo_conv0_bn_bwd_wgrad_workspace_bytes(void);
int ryolo_conv0_bn_bwd_wgrad(const ryolo_conv_desc *desc, const void *x, const void *w_packed, const void *dy, int dy_cstride,
                             const float *scale, const float *shift, const float *mean, const float *invstd, int act, const float *slope,
                             float *dgamma, float *dbeta, float *dslope, float *grad_oihw, int cin_real, int accumulate, void *workspace,
                             size_t workspace_bytes, void *stream);
int ryolo_conv_stat_rows(const ryolo_conv_desc *desc);
int ryolo_conv2d_bn_act_stats(const ryolo_conv_desc *desc, const void *x, const void *w_packed, const float *scale,
                              const float *shift, const void *residual, void *y,
                              double *stat_part /* fp64 [ryolo_conv_stat_rows][2][cpad(Cout)], ZEROED by the caller, or NULL */,
                              void *stream);
/* ryolo_bn_finalize reads the partial sums of channels [0, C) and writes zeros back over them, so one scratch buffer
 * that starts zeroed can serve every conv of a step without a memset per layer. */
int ryolo_bn_finalize(double *stat_part, int rows, int cpad, int C, long long count, float eps, float momentum,
                      const float *gamma, const float *beta, float *mean, float *invstd, float *scale, float *shift,
                      float *running_mean /* may be NULL */, float *running_var, void *stream);
int ryolo_bn_act_fwd(const void *z, int z_cstride, const float *scale, const float *shift, int act,
                     const float *slope /* device scalar or NULL */, const void *residual, int res_cstride, void *y,
                     int y_cstride, long long npix, int C, void *stream);
size_t ryolo_bn_act_bwd_workspace_bytes(long long npix, int C);
int ryolo_bn_act_bwd(const void *z, int z_cstride, const void *dy, int dy_cstride, const float *scale, const float *shift,
                     const float *mean, const float *invstd, int act, const float *slope, void *dz, int dz_cstride,
                     long long npix, int C, float *dgamma, float *dbeta, float *dslope, void *workspace,
                     size_t workspace_bytes, void *stream);
size_t ryolo_conv_packed_dgrad_bytes(int Cout, int Cin, int ksize, int stride);
int ryolo_conv_dgrad_tap_table(int ksize, int stride, int *host_out /* host int[72] */);
int ryolo_conv_pack_weights_dgrad(const float *w_oihw, int Cout, int Cin, int ksize, int stride, void *packed,
                                  const int *taps_table /* device int[72], a copy of ryolo_conv_dgrad_tap_table's output */,
                                  void *stream);
int ryolo_conv2d_dgrad(const ryolo_conv_desc *forward_desc, const void *dz, int dz_cstride, const void *packed_dgrad,
                       const float *ones, const float *zeros /* fp32 [cpad(Cin)] */, void *dx, int accumulate, void *stream);
/* The first pass of a block's BatchNorm / activation backward folded into the launch that PRODUCES its dy (autograd runs them as
 * separate ops: model/models.py:61-62 is a BatchNorm2d node, :281-282 the shortcut add whose gradient feeds it).  In a residual
 * chain the final gradient of a 3x3 block's output is written by the data gradient of the NEXT block's 1x1 conv (accumulating into
 * the chain's running gradient); ryolo_conv2d_dgrad_bnreduce is that data gradient and additionally reads the 3x3 block's conv
 * output z and leaves the per-channel partial sums of ryolo_bn_act_bwd's reduce pass in `part` ([rows][3][C_in] fp32, any
 * contents on entry), so that ryolo_bn_act_bwd_reduced only has to finalise and apply: the 3x3 block's z and dy are read once
 * less per step.  Activation leaky / PReLU only.  Kernels that carry the reduce: for 1x1 stride-1 layers with whole 128-channel
 * tiles the persistent 2x2 tile or conv_pw.hip (one row of `part` per workgroup); for the stride-1 layers whose plain data gradient
 * runs on a one-tile-per-workgroup tile (3x3 with C_in <= 128: the 128 x 128 and 256 x 64 tiles) that tile (one row per pixel tile;
 * ryolo_bn_act_bwd_reduced folds more than 2048 rows to 64 before it finalises -- workspace >= (3 + 192) * C floats for that).
 * ryolo_conv2d_dgrad_bnreduce_rows: 0 = this conv's data gradient cannot carry the reduce (stride 2, a layer conv_mq / conv_mp
 * or the persistent narrow tiles serve, ragged channel tiles ...) -- use the two separate calls; otherwise the rows of `part`. */
int ryolo_conv2d_dgrad_bnreduce_rows(const ryolo_conv_desc *forward_desc);
int ryolo_conv2d_dgrad_bnreduce(const ryolo_conv_desc *forward_desc, const void *dz, int dz_cstride, const void *packed_dgrad,
                                const float *ones, const float *zeros, void *dx, int accumulate,
                                const void *z /* conv output of the block that produced this conv's input */, int z_cstride,
                                const float *scale, const float *shift, const float *mean, const float *invstd /* its BatchNorm */,
                                const float *slope /* device scalar */, float *part /* [rows][3][C_in] */, void *stream);
int ryolo_bn_act_bwd_reduced(const void *z, int z_cstride, const void *dy, int dy_cstride, const float *scale, const float *shift,
                             const float *mean, const float *invstd, int act /* RYOLO_ACT_LEAKY */, const float *slope, void *dz,
                             int dz_cstride, long long npix, int C, float *dgamma, float *dbeta, float *dslope, const float *part,
                             int rows, void *workspace /* 3*C floats */, size_t workspace_bytes, void *stream);
size_t ryolo_conv_wgrad_workspace_bytes(const ryolo_conv_desc *forward_desc);
/* (measurement) the two launches of ryolo_conv2d_wgrad as separate calls -- the tile kernel that writes the split-K partials, then the
 * reduce into grad_oihw -- so that a caller bracketing calls with events times them apart.  Same arguments; partials then reduce ==
 * ryolo_conv2d_wgrad. */
int ryolo_conv2d_wgrad_partials(const ryolo_conv_desc *forward_desc, const void *x, const void *dz, int dz_cstride, int Cin_real,
                                float *grad_oihw, int accumulate, void *workspace, size_t workspace_bytes, void *stream);
int ryolo_conv2d_wgrad_reduce(const ryolo_conv_desc *forward_desc, const void *x, const void *dz, int dz_cstride, int Cin_real,
                              float *grad_oihw, int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* All split-K reduces of a backward segment as ONE launch (round 5).  ryolo_conv2d_wgrad_partials leaves a layer's partial tiles in ITS OWN
 * workspace (ryolo_conv_wgrad_workspace_bytes each; they must all stay alive until the batch has run); ryolo_conv_wgrad_reduce_job_fill
 * describes the reduce ryolo_conv2d_wgrad_reduce would launch for that layer as one job (returns its block count, 0 on a bad argument); the
 * caller lays the jobs out back to back (block_begin / block_end running sums), uploads the table once and calls
 * ryolo_conv_wgrad_reduce_batch(device_jobs, njobs, total_blocks) where the gradients are due -- the same bits as the per-layer reduces.
 * Replaces: autograd's per-layer weight-gradient accumulation behind /root/reference/train.py:268-282. */
typedef struct ryolo_wgrad_reduce_job {
    const float *part;   /* the layer's partial tiles [S][Cout_pad][Kpad] (device) */
    float *g;            /* fp32 OIHW gradient (device) */
    int S, Cout, Cin_real, Cin_k, ks, Kpad, Cout_pad, accumulate;
    int kind;            /* 0 one element per thread, 1 four split quarters per workgroup, 2 transposing 3x3 variant, 3 = 1 with four
                            input channels per thread and a quarter's loads all in flight, 4 = 2 with every load of a workgroup in flight (3, 4: the
                            batched launch's own forms, the same bits as 1 and 2) */
    int block_begin, block_end;
    int wide;            /* kinds 3, 4: several four-channel groups / (c_out, 64 c_in) units per workgroup (set by job_fill; 0 = one) */
} ryolo_wgrad_reduce_job;
int ryolo_conv_wgrad_reduce_job_fill(ryolo_wgrad_reduce_job *host_job, const ryolo_conv_desc *forward_desc, int Cin_real, const void *workspace,
                                     float *grad_oihw, int accumulate);
int ryolo_conv_wgrad_reduce_batch(const ryolo_wgrad_reduce_job *device_jobs, int njobs, int total_blocks, void *stream);
int ryolo_conv2d_wgrad(const ryolo_conv_desc *forward_desc, const void *x, const void *dz, int dz_cstride, int Cin_real,
                       float *grad_oihw /* fp32 [Cout][Cin_real][k][k] */, int accumulate, void *workspace,
                       size_t workspace_bytes, void *stream);
int ryolo_upsample2x_bwd(const void *dy, int dy_cstride, void *dx, int dx_cstride, int N, int H, int W, int C,
                         int accumulate, void *stream);
int ryolo_pgrad_to_nhwc(const float *pgrad, int bs, int na, int ny, int nx, int no, void *out, int out_cstride,
                        void *stream);

/* All weight packs of a training step in one launch.  A job packs one layout of one conv's fp32 OIHW weights:
 * kind 0 = the forward layout of ryolo_conv_pack_weights, kind 1 = one class of ryolo_conv_pack_weights_dgrad, kind 2 = one
 * x-fused stride-2 class (3x3 stride-2 convs with C_in in {32, 64}: two extra images behind the four classic ones, used by
 * ryolo_conv2d_dgrad when the input gradient is dense and its width even -- both column parities in one launch).
 * ryolo_conv_pack_job_fill (host) writes the 1 + (0 | 1 | 4 | 6) jobs of a conv into host_jobs (room for 7) and returns how many; the
 * caller concatenates all convs' jobs, turns each job's [0, block_end) into a running [block_begin, block_end) range,
 * uploads the array once and calls ryolo_conv_pack_batch(device_jobs, njobs, total_blocks) every step. */
typedef struct ryolo_pack_job {
    const void *src;   /* fp32 OIHW weights (device) */
    void *dst;         /* packed bf16 destination (device) */
    int kind, Cout, Cin, KS, Cin_pad, ntaps, Kpad, rows;
    int khs[9], kws[9];
    int block_begin, block_end;
} ryolo_pack_job;
int ryolo_conv_pack_job_fill(ryolo_pack_job *host_jobs, const float *w_oihw, int Cout, int Cin, int ksize, int stride,
                             int Cin_pad, void *packed_fwd, void *packed_dgrad);
int ryolo_conv_pack_batch(const ryolo_pack_job *device_jobs, int njobs, int total_blocks, void *stream);

/* ---------------------------------------------------------------------------------------------- training loss
 * One head of compute_loss (model/loss.py:266-367, 'default' arcs) and its gradient, no host synchronisation:
 *   lobj  = obj * mean_cells BCE(p[...,5], tobj; pos_weight obj_pw)                       (loss.py:346-348, all cells)
 *   lreg  = reg * (mean SmoothL1(sigmoid(p_xy), t_xy) + 2 mean SmoothL1(atan(p_a) + anchor_a, t_a)
 *                  + giou * mean(1 - wh_iou(t_wh, min(exp(p_wh), 1e3) * anchor_wh)))       (loss.py:314-323)
 *   lcls  = cls * mean BCE(p[...,6:], onehot; pos_weight cls_pw)          when nc > 1      (loss.py:329-333)
 * over the candidates (a, t) of the [na, NT] grid with w[a*NT + t] > 0 (build_targets, loss.py:161-258, in the
 * fixed-shape form of model/loss_static.py).  Per target t: image b, cell (gj, gi), class, cell offset txy, size twh and
 * angle ta in grid units.  npos = device scalar sum(w).  items[0..2] (lobj, lcls, lreg, already weighted) are
 * ACCUMULATED (the caller zeroes them once for all heads); dp [same shape as p] is overwritten with d(loss)/dp.
 * bitmap: ryolo_yolo_loss_bitmap_bytes(bs*na*ny*nx) bytes, zeroed by the caller (one bit per cell: several candidates
 * may share a cell, its objectness target is set once).  fp32; sums by atomics (order not fixed).                      */
/* build_targets (model/loss.py:161-258) over padded targets, all heads in one launch, no host synchronisation.
 * tpad [NT,7] rows (img, cls, x, y, w, h, angle) normalised, valid [NT] u8.  Per head h (host arrays of device pointers):
 * ng[h] = float[2] grid size (nx, ny), anchor_vec[h] = float[na,3]; outputs w[h] float[na,NT] (1 = positive candidate),
 * idx[h] int64[4,NT] = (image, class, gj, gi), box[h] float[5*NT] = txy [NT,2] | twh [NT,2] | ta [NT] in grid units (the
 * arrays ryolo_yolo_loss takes),
 * npos[h] float[1] (zeroed by the caller) += number of positives.  Every quirk of the reference is kept (cumulative
 * context rescale per head, angle gate on the last head's anchors, first-maximum / smallest-angle fallback). */
#define RYOLO_MAX_HEADS 4
int ryolo_build_targets(const float *tpad, const unsigned char *valid, int NT, int nheads, int na, const float *const *ng,
                        const float *const *anchor_vec, float iou_t, float ang_t, float context_factor, float *const *w,
                        long long *const *idx, float *const *box, float *const *npos, void *stream);
size_t ryolo_yolo_loss_bitmap_bytes(long long cells);
int ryolo_yolo_loss(const float *p, int bs, int na, int ny, int nx, int no, int nc, const float *w, int NT,
                    const long long *b, const long long *gj, const long long *gi, const long long *cls, const float *txy,
                    const float *twh, const float *ta, const float *anchor_vec /* [na,3] */, const float *npos, float giou,
                    float reg_w, float cls_w, float cls_pw, float obj_w, float obj_pw,
                    int iou_mode /* 0: wh_iou (reference, loss.py:322); 1: rotated IoU of the decoded box (riou) */,
                    unsigned *bitmap, float *dp, float *items /* [>=3] */, void *stream);

/* The same loss for a head that lives in the training engine: reads the head the conv wrote (`head`, bf16 NHWC, channel =
 * a*no + k; `p` is its fp32 [bs,na,ny,nx,no] copy, gathered by the positives) and writes d(loss)/d(head) as bf16 NHWC
 * (`head_grad`) -- the layout the head conv's backward consumes -- instead of an fp32 dp + a layout pass.  dp_sparse: fp32
 * [bs,na,ny,nx,no], ALL ZERO on entry and on exit (scratch for the positives' atomics; touched cells are re-zeroed).
 * bitmap as above (zeroed by the caller).  C = na*no and both channel strides must be multiples of 8. */
int ryolo_yolo_loss_nhwc(const void *head, int head_cstride, const float *p, int bs, int na, int ny, int nx, int no, int nc,
                         const float *w, int NT, const long long *b, const long long *gj, const long long *gi,
                         const long long *cls, const float *txy, const float *twh, const float *ta, const float *anchor_vec,
                         const float *npos, float giou, float reg_w, float cls_w, float cls_pw, float obj_w, float obj_pw,
                         int iou_mode, unsigned *bitmap, float *dp_sparse, void *head_grad, int head_grad_cstride, float *items,
                         void *stream);
/* buf[npix][C] (bf16, pixel stride cstride) *= g[0] unless g[0] == 1: the upstream gradient of loss.backward() applied to a
 * head gradient produced at loss time; returns after one scalar load in the usual g == 1 case. */
/* The other arcs of compute_loss (model/loss.py:284-286 focal wrappers -- the reference's usage note train.py:380 is
 * `--arc Fdefault` --, :350-360 unified heads, model/models.py:210-218): the same two functions with
 *   arc      RYOLO_ARC_FOCAL (every criterion but the IoU term is wrapped: loss *= (1.000001 - exp(-loss))^fl_gamma per element)
 *            | at most one of RYOLO_ARC_UBCE (BCE over the class logits of all cells, mean over cells*nc, in items[0]; no separate
 *            objectness / class-at-positives terms) and RYOLO_ARC_UCE (cross entropy over (background, classes) = logits 5..5+nc of
 *            all cells, mean over cells, in items[1]);  arc = 0 is the 'default' arc of the functions above;
 *   bitmap   ryolo_yolo_loss_bitmap_bytes_arc(cells, nc, arc) bytes, zeroed by the caller (uBCE keeps one more bit per (cell, class)).
 * no <= 32 for arc != 0. */
#define RYOLO_ARC_FOCAL 1
#define RYOLO_ARC_UBCE 2
#define RYOLO_ARC_UCE 4
size_t ryolo_yolo_loss_bitmap_bytes_arc(long long cells, int nc, int arc);
int ryolo_yolo_loss_arc(const float *p, int bs, int na, int ny, int nx, int no, int nc, const float *w, int NT,
                        const long long *b, const long long *gj, const long long *gi, const long long *cls, const float *txy,
                        const float *twh, const float *ta, const float *anchor_vec, const float *npos, float giou, float reg_w,
                        float cls_w, float cls_pw, float obj_w, float obj_pw, int iou_mode, int arc, float fl_gamma, unsigned *bitmap,
                        float *dp, float *items, void *stream);
int ryolo_yolo_loss_nhwc_arc(const void *head, int head_cstride, const float *p, int bs, int na, int ny, int nx, int no, int nc,
                             const float *w, int NT, const long long *b, const long long *gj, const long long *gi,
                             const long long *cls, const float *txy, const float *twh, const float *ta, const float *anchor_vec,
                             const float *npos, float giou, float reg_w, float cls_w, float cls_pw, float obj_w, float obj_pw,
                             int iou_mode, int arc, float fl_gamma, unsigned *bitmap, float *dp_sparse, void *head_grad,
                             int head_grad_cstride, float *items, void *stream);
int ryolo_scale_bf16_if(const float *g, void *buf, int cstride, long long npix, int C, void *stream);

/* Rotated IoU of n box pairs (cx, cy, w, h, angle; angle convention of get_rotated_coors, utils/utils.py:702-725) and its
 * gradient with respect to the FIRST box: iou [n], grad [n,5] (may be NULL).  The value is the polygon IoU of
 * skewiou (utils/utils.py:663-699) in fp32 (IoU(A, A) = 1); the reference has no backward for it -- this is the kernel of
 * the build's `riou` loss (hyp['riou'] = 1: lreg's wh_iou term becomes giou * mean(1 - riou(pbox, tbox))).
 * One pair per lane, boundary-integral form (csrc/riou_grad.h).  Degenerate boxes (w or h <= 0): 0, zero gradient. */
int ryolo_riou_loss_pairs(const float *pbox, const float *tbox, int n, float *iou, float *grad, void *stream);

/* ---------------------------------------------------------------------------------------------- optimizer
 * The SGD step of train.py:70-83 (momentum, nesterov, per-group weight decay) for every parameter tensor in one launch.
 * jobs: device array, one per tensor (fp32 p / grad / momentum buffer, n elements, param-group index, `first` = the
 * momentum buffer is uninitialised (first step: buf = d), [block_begin, block_end) = its workgroup range);
 * group_hparams: device float[groups][4] = (lr, momentum, weight_decay, gradient scale; 0 = no scaling -- data-parallel
 * runs pass 1/world instead of dividing the all-reduced gradient in a separate pass).  Same arithmetic as torch.optim.SGD. */
typedef struct ryolo_sgd_job {
    void *p;
    const void *g;
    void *buf;
    long long n;
    int group, first, block_begin, block_end;
} ryolo_sgd_job;
int ryolo_sgd_step(const ryolo_sgd_job *device_jobs, int njobs, int total_blocks, const float *group_hparams, int nesterov,
                   void *stream);

#ifdef __cplusplus
}
#endif
#endif /* RYOLO_H */
