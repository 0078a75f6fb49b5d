/*
 * od_boxes.h -- C ABI of libod_boxes.so: the detection-head box pipeline on MI355X (gfx950).
 *
 * The reference (Kotaro-Kuroda/object_detection v0.3.5) has no FFI: it is pure Python and reaches
 * this arithmetic through torchvision 0.16.2 objects it constructs in
 * object_detection/retinanet/retinanet_model.py:19-61 and calls in object_detection/train.py:62,84 and
 * object_detection/predict.py:61,72.  Each entry point below names the torchvision function(s) it
 * replaces and the reference line that puts them on the path.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter comment says "host";
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it, nothing synchronises;
 *   - no entry point allocates or frees device memory; scratch comes in through `workspace`
 *     (size from the matching *_workspace_bytes function, 256-byte aligned);
 *   - return value: 0 = ok; < 0 = OD_ERR_* (bad argument / documented limit / workspace too small);
 *     > 0 = a hipError_t from the runtime.  Nothing throws or exits;
 *   - boxes are fp32 xyxy, AoS [.,4]; indices/labels returned to callers are int64 like torchvision's;
 *   - preconditions for bit-exact matching: finite coordinates, x2 >= x1 and y2 >= y1 for GT boxes,
 *     anchors with positive area (what torchvision's forward asserts for targets).
 */
#ifndef OD_BOXES_H
#define OD_BOXES_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OD_MAX_LEVELS 8
#define OD_MAX_BASE_ANCHORS 16
#define OD_MAX_TOPK 8192        /* topk_candidates limit of od_select_candidates */
#define OD_MAX_NMS_BOXES 65536  /* boxes per image limit of od_batched_nms */

#define OD_CODER_BOXCODER 0     /* BoxCoder(weights=(1,1,1,1)).decode_single  (RetinaNet) */
#define OD_CODER_LINEAR 1       /* BoxLinearCoder(normalize_by_size=True).decode (FCOS)   */

#define OD_NMS_TRICK 0          /* _batched_nms_coordinate_trick */
#define OD_NMS_VANILLA 1        /* _batched_nms_vanilla          */
#define OD_NMS_AUTO_CPU 2       /* batched_nms rule for device.type == "cpu": numel > 4000 -> vanilla */
#define OD_NMS_AUTO_GPU 3       /* batched_nms rule otherwise:               numel > 20000 -> vanilla */
/* Threshold rule, or-ed into the mode.  torchvision's two native kernels disagree for an IoU that equals
 * float(iou_threshold): csrc/ops/cpu/nms_kernel.cpp promotes the fp32 IoU to double and compares with the double
 * threshold (the default here); csrc/ops/cuda/nms_kernel.cu compares the fp32 IoU with the threshold rounded to
 * float (this flag).  The reference runs on "cuda" when available (train.py:24, predict.py:25). */
#define OD_NMS_THRESHOLD_FLOAT 0x100
/* Or-ed into the mode: take the chip-wide bit-matrix form of the NMS (the shape of csrc/ops/cuda/nms_kernel.cu: an
 * n x ceil(n/64) matrix of suppression words computed by all compute units, then one ordered pass) when the call
 * qualifies -- num_seg*seg_stride <= OD_NMS_MATRIX_MAX_BOXES and a workspace of od_batched_nms_matrix_workspace_bytes.
 * Otherwise the flag is ignored.  Same results; faster when thousands of boxes are KEPT (a plain nms over a few thousand
 * boxes), slower for the pipeline's call, which keeps a few hundred and stops early. */
#define OD_NMS_PREFER_MATRIX 0x200
#define OD_NMS_MATRIX_MAX_BOXES 8192

/* Library version (major*10000 + minor*100 + patch). */
int od_version(void);

/*
 * Anchor generation.  Replaces torchvision.models.detection.anchor_utils.AnchorGenerator.grid_anchors
 * / forward (built at reference retinanet_model.py:22-27,42-46; called inside model(...) at
 * train.py:62,84, predict.py:61).  Output order (level, y, x, a); anchors = int32 shift + base.
 *   grid_h, grid_w, stride_h, stride_w, num_base : host int32[num_levels]
 *   base_anchors : host fp32 [sum(num_base), 4] -- the rounded cell anchors of every level
 *   out          : fp32 [n_total, 4], 16-byte aligned;  n_total = sum(grid_h*grid_w*num_base)
 */
int od_anchor_generate(const int32_t* grid_h, const int32_t* grid_w, const int32_t* stride_h,
                       const int32_t* stride_w, const int32_t* num_base, const float* base_anchors,
                       int32_t num_levels, float* out, int64_t n_total, void* stream);

/*
 * Fused box_iou + Matcher.  Replaces, per image, torchvision.ops.boxes.box_iou(gt, anchors) followed
 * by torchvision.models.detection._utils.Matcher(high, low, allow_low_quality_matches)(iou) -- the
 * loop body of RetinaNet.compute_loss (reference trigger train.py:62 with the thresholds of
 * configs/config_all.py:10-11 forwarded at train.py:144).  The [M,N] matrix is never materialised.
 *   anchors             : fp32 [n,4] (anchor_batch_stride = 0, shared by all images) or
 *                         [batch, n, 4] (anchor_batch_stride = n), 16-byte aligned
 *   gt_boxes            : fp32 [gt_offsets[batch], 4], images concatenated, 16-byte aligned
 *   gt_offsets          : int32 [batch+1], CSR offsets into gt_boxes (device)
 *   m_total             : gt_offsets[batch] as known to the host (sizes the workspace)
 *   matched             : int64 [batch, n]: matched GT index, -1 below `low`, -2 between thresholds;
 *                         an image with zero GT boxes gets all -1 (compute_loss's empty-target guard)
 *   workspace           : od_match_retinanet_workspace_bytes(m_total, batch, n) bytes, 256-byte aligned; scratch
 *                         without meaning between calls (nothing in it has to be initialised).  Calls that share a
 *                         workspace must not overlap in time.
 */
size_t od_match_retinanet_workspace_bytes(int64_t m_total, int32_t batch, int64_t n);
int od_match_retinanet(const float* anchors, int64_t anchor_batch_stride, int64_t n, const float* gt_boxes,
                       const int32_t* gt_offsets, int64_t m_total, int32_t batch, float high, float low,
                       int32_t allow_low_quality, int64_t* matched, void* workspace, size_t workspace_bytes,
                       void* stream);
/*
 * The same call with a hint about the order of the anchor list: anchor_grids (host) holds num_grids triples
 * (first, row_len, rows) saying that anchors [first, first + row_len * rows) are `rows` rows of `row_len` anchors with
 * neighbouring rows spatially adjacent -- a pyramid level as torchvision's AnchorGenerator.grid_anchors emits it
 * (order y, x, a: row_len = grid_w * anchors per location, rows = grid_h; anchor_utils.py of torchvision 0.16.2).
 * The kernels then form their 256-anchor tiles from 4 rows x 64 anchors instead of 256 consecutive ones wherever a grid
 * has at least 4 rows of at least 64 anchors and at least 16384 anchors in all (a row length that is not a multiple of
 * 64, or a row count that is not a multiple of 4, makes the last tiles of a row / of the grid overlap their neighbours;
 * grids for which that would cost more than a quarter more tiles are tiled consecutively), which tightens the tiles'
 * bounding boxes.  Results do not depend on the hint.  Grids must be ascending, disjoint and inside [0, n) (OD_ERR_ARG).
 */
int od_match_retinanet_grids(const float* anchors, int64_t anchor_batch_stride, int64_t n, const int64_t* anchor_grids,
                             int32_t num_grids, const float* gt_boxes, const int32_t* gt_offsets, int64_t m_total,
                             int32_t batch, float high, float low, int32_t allow_low_quality, int64_t* matched,
                             void* workspace, size_t workspace_bytes, void* stream);

/*
 * Tuning knob of od_match_retinanet[_grids] (process-wide, like the profile switches below; no counterpart in the
 * reference): the number of workgroups per image that share the low-quality restore of the two-launch path.  Every such
 * workgroup holds a whole compute unit.  Many of them make the kernel short when the GPU has nothing else to do
 * (default 16: a training step, where the losses wait for the matching); few of them leave the compute units to the
 * kernels of other streams when matching runs beside other work (3 measured best for 16 images next to the inference
 * pipeline).  Results do not depend on it.  groups in [2, 64]; 0 restores the default.  Returns the previous setting
 * (0 = default) or OD_ERR_ARG.
 */
int od_match_restore_groups(int32_t groups);

/*
 * FCOS centre-sampling matcher.  Replaces the matching loop of torchvision FCOS.compute_loss
 * (reference trigger retinanet_model.py:55-60 + train.py:62).  n_first_level / n_last_level are
 * num_anchors_per_level[0] / [-1] (lower bound 0 on the first level, upper bound inf on the last).
 * Same anchors / gt layout as od_match_retinanet.  matched: int64 [batch, n], -1 = unmatched.
 */
int od_match_fcos(const float* anchors, int64_t anchor_batch_stride, int64_t n, int64_t n_first_level,
                  int64_t n_last_level, const float* gt_boxes, const int32_t* gt_offsets, int32_t batch,
                  float center_sampling_radius, int64_t* matched, void* stream);

/*
 * Pairwise IoU matrix.  Replaces torchvision.ops.boxes.box_iou (public op; used by callers that want
 * the matrix itself).  out: fp32 [m, n] row-major, row = boxes1 index.
 */
int od_box_iou(const float* boxes1, int64_t m, const float* boxes2, int64_t n, float* out, void* stream);

/*
 * Matcher on an explicit quality matrix.  Replaces torchvision Matcher.__call__ /
 * set_low_quality_matches_ for callers that hold a matrix (model.proposal_matcher(matrix)).
 *   quality : fp32 [m, n] row-major (m = GT, n = predictions), m >= 1, n >= 1
 *   workspace : od_matcher_workspace_bytes(m, n)
 */
size_t od_matcher_workspace_bytes(int64_t m, int64_t n);
int od_matcher(const float* quality, int64_t m, int64_t n, float high, float low, int32_t allow_low_quality,
               int64_t* matches, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Box decoding + clipping.  Replaces BoxCoder.decode_single / BoxLinearCoder.decode followed
 * (when clip_h > 0) by torchvision.ops.boxes.clip_boxes_to_image.  exp() is the library's specified
 * fp32 algorithm (Cody-Waite reduction + degree-6 polynomial, one rounding per operation, no FMA);
 * rel_codes, boxes, out: fp32 [n,4].
 */
int od_decode_boxes(const float* rel_codes, const float* boxes, int64_t n, int32_t coder,
                    float bbox_xform_clip, float clip_h, float clip_w, float* out, void* stream);

/*
 * Score filter + top-k + decode + clip for every (image, level).  Replaces the inner loop of
 * RetinaNet.postprocess_detections / FCOS.postprocess_detections up to and including
 * clip_boxes_to_image (reference trigger train.py:84, predict.py:61; knobs predict.py:44-45).
 *   cls_logits[l]      : host array of device pointers, level l tensor [batch, level_size[l], num_classes]
 *   cls_batch_stride[l]: host int64, elements between consecutive images of level l
 *   bbox_ctrness       : NULL (RetinaNet: score = sigmoid(cls)) or like cls_logits with 1 value per
 *                        anchor (FCOS: score = sqrt(sigmoid(cls) * sigmoid(ctrness)))
 *   bbox_regression[l] : [batch, level_size[l], 4];  anchors[l]: [level_size[l], 4] (stride 0) or batched
 *   image_hw           : int32 [batch, 2] (height, width) device -- clip limits
 *   outputs, padded per (image, level) to `topk` slots, slot = level*topk + rank:
 *     cand_boxes fp32 [batch, L*topk, 4]; cand_scores fp32 [batch, L*topk]; cand_labels int32;
 *     cand_index int64 (flat index anchor*num_classes + class inside the level; may be NULL);
 *     cand_count int32 [batch, L].  Candidates of a level are ordered by (score desc, flat index asc).
 */
size_t od_select_candidates_workspace_bytes(const int64_t* level_size, int32_t num_levels, int32_t batch,
                                            int32_t num_classes);
int od_select_candidates(const float* const* cls_logits, const int64_t* cls_batch_stride,
                         const float* const* bbox_ctrness, const int64_t* ctr_batch_stride,
                         const float* const* bbox_regression, const int64_t* reg_batch_stride,
                         const float* const* anchors, const int64_t* anchor_batch_stride,
                         const int64_t* level_size, int32_t num_levels, int32_t batch, int32_t num_classes,
                         const int32_t* image_hw, float score_thresh, int32_t topk, int32_t coder,
                         float bbox_xform_clip, float* cand_boxes, float* cand_scores, int32_t* cand_labels,
                         int64_t* cand_index, int32_t* cand_count, void* workspace, size_t workspace_bytes,
                         void* stream);

/*
 * The same, reading the head's convolution outputs in their own channel-major layout, which removes the
 * permute + reshape copy of torchvision RetinaNetClassificationHead.forward / RetinaNetRegressionHead.forward
 * (`view(N,-1,K,H,W).permute(0,3,4,1,2).reshape(N,-1,K)`; heads built at reference retinanet_model.py:47-54):
 *   cls_logits[l]      : [batch, A_l * num_classes, plane_size[l]]  (plane = H*W of the level, A_l = anchors per cell)
 *   bbox_ctrness[l]    : [batch, A_l, plane_size[l]] or NULL;   bbox_regression[l]: [batch, A_l * 4, plane_size[l]]
 *   level_size[l] = plane_size[l] * A_l;  anchors and all outputs as above -- flat indices, order and results are
 *   identical to od_select_candidates on the permuted tensors.
 */
int od_select_candidates_planar(const float* const* cls_logits, const int64_t* cls_batch_stride,
                                const float* const* bbox_ctrness, const int64_t* ctr_batch_stride,
                                const float* const* bbox_regression, const int64_t* reg_batch_stride,
                                const float* const* anchors, const int64_t* anchor_batch_stride,
                                const int64_t* level_size, const int64_t* plane_size, int32_t num_levels,
                                int32_t batch, int32_t num_classes, const int32_t* image_hw, float score_thresh,
                                int32_t topk, int32_t coder, float bbox_xform_clip, float* cand_boxes,
                                float* cand_scores, int32_t* cand_labels, int64_t* cand_index, int32_t* cand_count,
                                void* workspace, size_t workspace_bytes, void* stream);

/*
 * Class-batched NMS for a batch of images.  Replaces torchvision.ops.boxes.batched_nms
 * (_batched_nms_coordinate_trick / _batched_nms_vanilla) + the native torchvision::nms op +
 * `keep[:max_keep]` + the gather of boxes/scores/labels (tail of postprocess_detections).
 * Greedy order = stable descending score (ties: lower index first); box j is suppressed by a kept
 * box i when fp32 inter/(area_i+area_j-inter), promoted to double, is > iou_threshold (with OD_NMS_THRESHOLD_FLOAT
 * in `mode`: when the fp32 quotient is > (float)iou_threshold).
 *   boxes [batch, num_seg*seg_stride, 4], scores, labels (int32, NULL = one class): padded segments;
 *   seg_count int32 [batch, num_seg]: valid entries at the head of each segment (device).
 *   segments_sorted: nonzero if every segment is already in (score desc, index asc) order.
 *   Indices written to keep_idx count valid entries only (the index a torch.cat of the segments
 *   would give).  Outputs: keep_idx int64 [batch, max_keep] (may be NULL), out_boxes fp32
 *   [batch, max_keep, 4], out_scores fp32, out_labels int64 (NULL allowed), out_count int32 [batch].
 *   Rows past out_count are zero-filled.  Limits: num_seg*seg_stride <= OD_MAX_NMS_BOXES, num_seg <= 64.
 *   Up to 8 sorted segments are merged lazily, strip by strip; unsorted input is ranked first (O(n^2) comparisons).
 */
size_t od_batched_nms_workspace_bytes(int32_t batch, int32_t num_seg, int32_t seg_stride);
/* workspace that lets OD_NMS_PREFER_MATRIX take effect (0: the call does not qualify); includes the size above */
size_t od_batched_nms_matrix_workspace_bytes(int32_t batch, int32_t num_seg, int32_t seg_stride);
int od_batched_nms(const float* boxes, const float* scores, const int32_t* labels, const int32_t* seg_count,
                   int32_t num_seg, int32_t seg_stride, int32_t segments_sorted, int32_t batch,
                   double iou_threshold, int32_t mode, int32_t max_keep, int64_t* keep_idx, float* out_boxes,
                   float* out_scores, int64_t* out_labels, int32_t* out_count, void* workspace,
                   size_t workspace_bytes, void* stream);

/*
 * od_batched_nms for callers that cannot allocate device memory or carry a workspace: code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc.  No counterpart in the reference.  Same arguments, results
 * and error codes as od_batched_nms, without the workspace: the kernels' scratch memory is a device array of
 * OD_NMS_NOALLOC_SCRATCH_BYTES inside the library (one copy per GPU, held until the process ends).  The call allocates
 * and frees nothing; the device that `stream` belongs to must be the current device.
 *   limit: od_batched_nms_workspace_bytes(batch, num_seg, seg_stride) <= OD_NMS_NOALLOC_SCRATCH_BYTES (e.g. 16 images
 *   of 8192 boxes), otherwise OD_ERR_LIMIT (-2).
 *   All calls on one GPU share the array: calls that can overlap in time on one GPU (two streams, two threads) race
 *   and must be serialised by the caller.
 *   The first call on a device looks up the array's address on that device (hipGetSymbolAddress), which is not a stream
 *   operation: make one call per device outside any stream capture before recording the entry point into a HIP graph.
 */
#define OD_NMS_NOALLOC_SCRATCH_BYTES (8u << 20)
int od_batched_nms_noalloc(const float* boxes, const float* scores, const int32_t* labels, const int32_t* seg_count,
                           int32_t num_seg, int32_t seg_stride, int32_t segments_sorted, int32_t batch,
                           double iou_threshold, int32_t mode, int32_t max_keep, int64_t* keep_idx, float* out_boxes,
                           float* out_scores, int64_t* out_labels, int32_t* out_count, void* stream);

/*
 * Message of the detections all-gather: the padded outputs of od_batched_nms for the local images, packed into
 * one fp32 row per image so that one collective moves them.  No counterpart in the reference, which runs one
 * process on one device (train.py:24, predict.py:25); this is the only exchange of the image-sharded pipeline.
 *   boxes fp32 [batch, max_det, 4], scores fp32 [batch, max_det], labels int64 [batch, max_det] (values below 2^24
 *   are exact in fp32), counts int32 [batch]  ->  message fp32 [batch, 6*max_det + 1]: max_det rows
 *   (x1, y1, x2, y2, score, label) followed by the count.
 */
int od_pack_detections(const float* boxes, const float* scores, const int64_t* labels, const int32_t* counts,
                       int32_t batch, int32_t max_det, float* message, void* stream);

/*
 * Tail of the reference's prediction loop for a whole batch in one launch.  Replaces, per image, the sequence at
 * reference predict.py:70-76: `boxes.mul_(ratio)` (ratio = [w/width, h/height, w/width, h/height] as fp32),
 * `nms(boxes, scores, 0.3)` (torchvision.ops.nms, class-agnostic, on the rescaled boxes), `boxes.to(torch.int32)`
 * and the three gathers `boxes[keep]`, `scores[keep]`, `labels[keep]`.
 *   boxes fp32 [batch, max_det, 4], scores fp32 [batch, max_det], labels int64 [batch, max_det], counts int32 [batch]:
 *     the padded detections as od_batched_nms writes them (any score order is accepted; the greedy order is the
 *     stable descending one, as torchvision's kernel sorts);  ratio fp32 [batch, 4] (device), 16-byte aligned
 *   mode: 0 = threshold rule of torchvision's CPU kernel, OD_NMS_THRESHOLD_FLOAT = of its CUDA kernel
 *   scaled_boxes fp32 [batch, max_det, 4] or NULL: the rescaled boxes of the valid detections in their given order
 *     (what mul_ leaves in the prediction dictionary); may be the same memory as `boxes`
 *   out_boxes int32 [batch, max_det, 4] (fp32 -> int32 conversion truncates towards zero like Tensor.to; values
 *     outside the int32 range saturate, NaN becomes 0), out_scores fp32, out_labels int64, keep_idx int64 or NULL
 *     (positions inside the image's detection list, as nms returns them), out_count int32 [batch].
 *   Rows past out_count are zero-filled.  Limit: max_det <= 1024 (OD_ERR_LIMIT beyond).
 *   workspace: od_predict_tail_workspace_bytes(batch, max_det), 256-byte aligned.  Above 128 detections per image the
 *   suppression matrix of an image is computed by several workgroups, which leave it there for the last one to finish.
 */
#define OD_MAX_PREDICT_TAIL 1024
size_t od_predict_tail_workspace_bytes(int32_t batch, int32_t max_det);
int od_predict_tail(const float* boxes, const float* scores, const int64_t* labels, const int32_t* counts,
                    const float* ratio, int32_t batch, int32_t max_det, double iou_threshold, int32_t mode,
                    float* scaled_boxes, int32_t* out_boxes, float* out_scores, int64_t* out_labels,
                    int64_t* keep_idx, int32_t* out_count, void* workspace, size_t workspace_bytes, void* stream);

/*
 * Classification loss of the RetinaNet head, fused with its target construction.  Replaces torchvision
 * RetinaNetClassificationHead.compute_loss = per image sigmoid_focal_loss(logits[valid], target[valid], "sum")
 * / max(1, #foreground), averaged over the batch (head built at reference retinanet_model.py:47-54, loss taken at
 * train.py:62-63).  target[a, c] = 1 iff matched[a] >= 0 and gt_labels[matched[a]] == c; anchors with
 * matched == -2 are ignored.
 *   cls_logits : [batch, n, num_classes] of element type `dtype` (OD_DTYPE_*: fp32, or the fp16 / bf16 a head
 *                produces under autocast -- the reference trains with use_amp, config_all.py:16, train.py:61),
 *                images batch_stride elements apart; arithmetic is fp32 for every element type
 *   matched    : int64 [batch, n] as written by od_match_retinanet;  gt_labels: int64 [gt_offsets[batch]]
 *   forward  -> loss fp32 [1] (mean over images), per_image_loss fp32 [batch], num_foreground int32 [batch]
 *   backward -> grad_logits [batch, n, num_classes] of the same element type (contiguous; 16-bit values rounded once,
 *               to nearest even) = grad_loss[0] * d loss / d cls_logits, from the forward's num_foreground
 * normalization: OD_LOSS_NORM_PER_IMAGE as above; OD_LOSS_NORM_BATCH is the FCOS classification loss
 *   (torchvision FCOSHead.compute_loss: sigmoid_focal_loss over the whole batch, "sum", / max(1, #foreground of the
 *   batch); FCOS built at reference retinanet_model.py:56-60); num_foreground[b] then holds the batch total.
 * Floating point: agrees with the fp32 torch evaluation to about 1e-6 relative (tests use 1e-5).
 */
#define OD_LOSS_NORM_PER_IMAGE 0 /* mean over images of sum_b / max(1, #fg_b)   (RetinaNet heads) */
#define OD_LOSS_NORM_BATCH 1     /* sum over the batch / max(1, #fg of the batch) (FCOS head)       */
#define OD_DTYPE_F32 0
#define OD_DTYPE_F16 1          /* IEEE half    */
#define OD_DTYPE_BF16 2         /* bfloat16     */
size_t od_focal_loss_workspace_bytes(int64_t n, int32_t num_classes, int32_t batch);
int od_focal_loss_forward(const void* cls_logits, int32_t dtype, int64_t batch_stride, int64_t n, int32_t num_classes,
                          const int64_t* matched, const int64_t* gt_labels, const int32_t* gt_offsets, int32_t batch,
                          float alpha, float gamma, int32_t normalization, float* loss, float* per_image_loss,
                          int32_t* num_foreground, void* workspace, size_t workspace_bytes, void* stream);
int od_focal_loss_backward(const void* cls_logits, int32_t dtype, int64_t batch_stride, int64_t n, int32_t num_classes,
                           const int64_t* matched, const int64_t* gt_labels, const int32_t* gt_offsets, int32_t batch,
                           float alpha, float gamma, int32_t normalization, const int32_t* num_foreground,
                           const float* grad_loss, void* grad_logits, void* stream);

/*
 * Regression loss of the RetinaNet head.  Replaces torchvision RetinaNetRegressionHead.compute_loss =
 * per image _box_loss(type, box_coder, anchors[fg], gt_boxes[matched[fg]], bbox_regression[fg]) / max(1, #fg) over the
 * foreground anchors fg = matched >= 0, averaged over the batch.  The reference selects type "giou"
 * (reference retinanet_model.py:53: generalized_box_iou_loss of the decoded boxes, eps = fp32 machine epsilon);
 * "l1" (torchvision's default) and "smooth_l1" compare the regression with BoxCoder.encode_single(gt, anchor).
 *   bbox_regression: [batch, n, 4] of element type `dtype` (OD_DTYPE_*), images batch_stride elements apart, rows
 *                    aligned to four elements; arithmetic is fp32
 *   anchors        : fp32 [n, 4]; images anchor_batch_stride elements apart, 0 = one table shared by all images
 *   matched        : int64 [batch, n];  gt_boxes: fp32 [gt_offsets[batch], 4];  weights: 4 host floats (BoxCoder)
 *   forward  -> loss fp32 [1], per_image_loss fp32 [batch], num_foreground int32 [batch]
 *   backward -> grad_regression [batch, n, 4] of the same element type (contiguous; zero rows for anchors that are
 *               not foreground)
 * Floating point: fp32 in the operation order of the torch evaluation; tests use 1e-5 relative.
 */
#define OD_BOX_LOSS_L1 0
#define OD_BOX_LOSS_SMOOTH_L1 1
#define OD_BOX_LOSS_GIOU 2
size_t od_box_loss_workspace_bytes(int64_t n, int32_t batch);
int od_box_loss_forward(const void* bbox_regression, int32_t dtype, int64_t batch_stride, const float* anchors,
                        int64_t anchor_batch_stride, int64_t n, const int64_t* matched, const float* gt_boxes,
                        const int32_t* gt_offsets, int32_t batch, int32_t loss_type, float beta, const float* weights,
                        float bbox_xform_clip, float eps, float* loss, float* per_image_loss, int32_t* num_foreground,
                        void* workspace, size_t workspace_bytes, void* stream);
int od_box_loss_backward(const void* bbox_regression, int32_t dtype, int64_t batch_stride, const float* anchors,
                         int64_t anchor_batch_stride, int64_t n, const int64_t* matched, const float* gt_boxes,
                         const int32_t* gt_offsets, int32_t batch, int32_t loss_type, float beta, const float* weights,
                         float bbox_xform_clip, float eps, const int32_t* num_foreground, const float* grad_loss,
                         void* grad_regression, void* stream);

/*
 * Regression and centre-ness losses of the FCOS head.  Replaces, in torchvision FCOSHead.compute_loss (FCOS built at
 * reference retinanet_model.py:56-60), generalized_box_iou_loss(BoxLinearCoder.decode(bbox_regression, anchors)[fg],
 * gt[fg], "sum") and binary_cross_entropy_with_logits(bbox_ctrness[fg], ctrness_target[fg], "sum"), both divided by
 * max(1, #foreground of the batch); ctrness_target = sqrt(min(l,r)/max(l,r) * min(t,b)/max(t,b)) of
 * BoxLinearCoder.encode(anchors, gt).  fg = matched >= 0.
 *   bbox_regression [batch, n, 4], bbox_ctrness [batch, n] of element type dtype; matched int64 [batch, n]
 *   forward  -> loss_regression, loss_ctrness fp32 [1]; num_foreground int32 [batch] (the batch total, repeated)
 *   backward -> grad_regression [batch, n, 4], grad_ctrness [batch, n] (contiguous, same element type)
 *   workspace: od_box_loss_workspace_bytes(n, batch)
 */
int od_fcos_box_loss_forward(const void* bbox_regression, const void* bbox_ctrness, int32_t dtype,
                             int64_t reg_batch_stride, int64_t ctr_batch_stride, const float* anchors,
                             int64_t anchor_batch_stride, int64_t n, const int64_t* matched, const float* gt_boxes,
                             const int32_t* gt_offsets, int32_t batch, float eps, float* loss_regression,
                             float* loss_ctrness, int32_t* num_foreground, void* workspace, size_t workspace_bytes,
                             void* stream);
int od_fcos_box_loss_backward(const void* bbox_regression, const void* bbox_ctrness, int32_t dtype,
                              int64_t reg_batch_stride, int64_t ctr_batch_stride, const float* anchors,
                              int64_t anchor_batch_stride, int64_t n, const int64_t* matched, const float* gt_boxes,
                              const int32_t* gt_offsets, int32_t batch, float eps, const int32_t* num_foreground,
                              const float* grad_loss_regression, const float* grad_loss_ctrness, void* grad_regression,
                              void* grad_ctrness, void* stream);

/*
 * Image side of the model's input transform in one pass.  Replaces torchvision GeneralizedRCNNTransform.forward's
 * normalize -> resize (torch.nn.functional.interpolate, bilinear, align_corners=False, recompute_scale_factor=True)
 * -> batch_images (zero padding); the reference configures it through get_model(min_size=size, max_size=size)
 * (reference train.py:142-144).
 *   images[i]   : host array of device pointers, image i fp32 [channels, image_h[i], image_w[i]] contiguous
 *   resized_h/w : host int32, size of image i after the resize (chosen by the caller exactly as torchvision does)
 *   mean, std   : host float [channels], channels <= 4
 *   out         : fp32 [batch, channels, out_h, out_w]; image i occupies the top-left resized_h[i] x resized_w[i]
 *                 corner, the rest is written as zero
 * Floating point: every operation rounded separately in the order of the torch evaluation; tests use 1e-5.
 */
int od_preprocess_images(const float* const* images, const int32_t* image_h, const int32_t* image_w,
                         const int32_t* resized_h, const int32_t* resized_w, int32_t batch, int32_t channels,
                         const float* mean, const float* std, float* out, int32_t out_h, int32_t out_w, void* stream);

/*
 * Per-kernel timing aid.  No counterpart in the reference (it has no profiling hooks); used by
 * bench.py to measure a kernel's own duration on the stream it runs on.  od_profile_select makes the
 * library launch that kernel (or all kernels) with a start and a stop event that receive the dispatch's own
 * begin / end time stamps (hipExtLaunchKernel) -- the duration a kernel trace reports;
 * od_profile_read waits for those events, returns the summed elapsed milliseconds and the number of
 * launches since the previous read, and resets the kernel's counters.
 */
#define OD_PROFILE_OFF (-1)
#define OD_PROFILE_ALL (-2)
#define OD_KERNEL_ANCHOR 0
#define OD_KERNEL_MATCH_ASSIGN 1    /* od_match_retinanet: argmax / thresholds / per-GT maxima */
#define OD_KERNEL_MATCH_RESTORE 2   /* od_match_retinanet: low-quality-match restore from entries */
#define OD_KERNEL_MATCH_FCOS 3
#define OD_KERNEL_BOX_IOU 4
#define OD_KERNEL_MATCHER 5         /* both kernels of od_matcher */
#define OD_KERNEL_DECODE 6
#define OD_KERNEL_FILTER 7          /* od_select_candidates: score filter stream          */
#define OD_KERNEL_SELECT 8          /* od_select_candidates: top-k + decode + clip        */
#define OD_KERNEL_NMS 9
#define OD_KERNEL_MATCH_HEAVY 10     /* od_match_retinanet: cooperative kernel for GT-heavy tiles */
#define OD_KERNEL_MATCH_INIT 11     /* od_match_retinanet: start values of the row maxima and tile counters */
#define OD_KERNEL_FOCAL_FWD 12      /* od_focal_loss_forward: the streaming kernel               */
#define OD_KERNEL_FOCAL_BWD 13
#define OD_KERNEL_BOX_LOSS_FWD 14   /* od_box_loss_forward: the streaming kernel                 */
#define OD_KERNEL_BOX_LOSS_BWD 15
#define OD_KERNEL_PREPROCESS 16
#define OD_KERNEL_SAMPLE 17         /* od_select_candidates: sampled score bound per (image, level)   */
#define OD_KERNEL_FILTER_REDO 18    /* od_select_candidates: second filter pass (normally exits at once) */
#define OD_KERNEL_PACK 19           /* od_pack_detections */
#define OD_KERNEL_NMS_ORDER 20      /* od_batched_nms: global order of unsorted candidates */
#define OD_KERNEL_PREDICT_TAIL 21   /* od_predict_tail */
#define OD_KERNEL_COUNT 22
int od_profile_select(int32_t kernel_id);
/* Time only every `every`-th launch of a selected kernel (default 1): a timed launch costs the host tens of
 * microseconds, which a throughput measurement should not pay on every step. */
int od_profile_stride(int32_t every);
int od_profile_read(int32_t kernel_id, double* total_ms, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* OD_BOXES_H */
