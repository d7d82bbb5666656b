/* libsylph_hip.so -- C ABI of the MI355X-native Sylph few-shot detection inference path.
 *
 * The reference (facebookresearch/sylph-few-shot-detection) is 100 % Python and has no FFI: its
 * "ABI" for this path is nn.Module.forward.  Each entry point below names the reference call it
 * replaces (paths relative to the reference root).  Plain pointers and sizes only; device pointers
 * are raw HIP device addresses (e.g. torch.Tensor.data_ptr()), the caller owns every buffer it
 * passes in, the library owns only the weights and workspace held by the context.
 *
 * Conventions
 *   - return value: 0 on success, non-zero on error; sylph_last_error() gives the message.
 *   - one context per process / GPU (the reference runs one process per GPU,
 *     tools/train_net.py:98-106); kernels are enqueued on the stream given to sylph_set_stream
 *     (default: the NULL stream).  No internal threads.  Calls do not synchronise unless stated.
 *   - "current batch": sylph_preprocess / sylph_import_pyramid select the batch (B, H, W) the
 *     following stage calls operate on.
 */
#ifndef SYLPH_HIP_H
#define SYLPH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sylph_ctx sylph_ctx;

enum { SYLPH_F32 = 0, SYLPH_BF16 = 1, SYLPH_F32S = 2 };

/* Subset of the yacs config the path reads (sylph/runner/adet_configs.py:25-61,
 * sylph/runner/default_configs.py:43-167). */
typedef struct sylph_config {
  int resnet_depth;        /* MODEL.RESNETS.DEPTH: 50 | 101 | 152 */
  int stride_in_1x1;       /* MODEL.RESNETS.STRIDE_IN_1X1 */
  int num_cls_convs;       /* MODEL.FCOS.NUM_CLS_CONVS */
  int num_box_convs;       /* MODEL.FCOS.NUM_BOX_CONVS */
  int nlevels;             /* len(MODEL.FCOS.FPN_STRIDES) == 5 */
  int strides[8];          /* MODEL.FCOS.FPN_STRIDES */
  float pixel_mean[3];     /* MODEL.PIXEL_MEAN */
  float pixel_std[3];      /* MODEL.PIXEL_STD */
  int size_divisibility;   /* backbone.size_divisibility (32) */
  int use_scale;           /* MODEL.FCOS.USE_SCALE */
  int cond_use_bias;       /* MODEL.META_LEARN.CODE_GENERATOR.USE_BIAS (CondConvBasic use_bias) */
  float pre_nms_thresh;    /* MODEL.FCOS.INFERENCE_TH_TEST */
  int pre_nms_topk;        /* MODEL.FCOS.PRE_NMS_TOPK_TEST */
  float nms_thresh;        /* MODEL.FCOS.NMS_TH */
  int post_nms_topk;       /* MODEL.FCOS.POST_NMS_TOPK_TEST */
  int thresh_with_ctr;     /* MODEL.FCOS.THRESH_WITH_CTR */
  int quality_mode;        /* MODEL.FCOS.BOX_QUALITY: 0 ["ctrness"], 1 ["iou"], 2 ["ctrness","iou"] */
  int cg_tower_layers;     /* len(CODE_GENERATOR.TOWER_LAYERS) (each ["GN","ReLU"]) */
  int cg_has_bias;         /* len(CODE_GENERATOR.BIAS_LAYER) != 0 */
  int cg_bias_l2_norm;     /* CODE_GENERATOR.BIAS_L2_NORM */
  int cg_post_norm;        /* CODE_GENERATOR.POST_NORM == "GN" */
  int cg_conv_l2_norm;     /* CODE_GENERATOR.CONV_L2_NORM */
  int cg_use_weight_scale; /* CODE_GENERATOR.USE_WEIGHT_SCALE */
  float prior_prob;        /* MODEL.FCOS.PRIOR_PROB */
  int cand_cap;            /* per (image, level) candidate capacity of the decode scan (0 = default: every location x class score of the largest level up to 262144 slots, else 1/8 of them, at most 4 M) */
  /* ROIEncoder variant (sylph/runner/default_configs.py:149-167; LVIS ROI-Encoder yaml) */
  int cg_type;             /* CODE_GENERATOR.NAME: 0 "CodeGenerator", 1 "ROIEncoder" */
  int tok_num_conv;        /* TOKENIZER.NUM_CONV (conv3x3 + GN + ReLU, CONV_DIM 256, NORM "GN") */
  int tok_num_fc;          /* TOKENIZER.NUM_FC (FC_DIM 256) */
  int enc_layers;          /* TRANSFORMER_ENCODER.LAYERS */
  int head_num_fc;         /* HEAD.NUM_FC */
  int head_fc_dim;         /* HEAD.FC_DIM (OUTPUT_DIM 256) */
  int cg_meta_bias;        /* CODE_GENERATOR.META_BIAS: the bias prior is the learned parameter
                              code_generator.code_generator_head.bias_value of the checkpoint (code_generator.py:422-425,856) */
  int cg_has_weight;       /* len(CODE_GENERATOR.WEIGHT_LAYER) != 0: learned per-shot weights, softmax over the shots of a class
                              (support_set_cls_weight: conv3x3 256->1 + global average pool; code_generator.py:583-613,766-777) */
  int cg_has_scale;        /* len(CODE_GENERATOR.SCALE_LAYER) != 0: per-class weight norm "cls_weight_norm"
                              (support_set_cls_scale: conv3x3 256->1 + global average pool; code_generator.py:615-645,976-993) */
  int num_share_convs;     /* MODEL.FCOS.NUM_SHARE_CONVS: layers of the shared tower in front of the cls / bbox towers (fcos.py:397,626) */
  int tower_norm;          /* MODEL.FCOS.NORM: 0 "GN" (and "NaiveGN": adet's NaiveGroupNorm is the same arithmetic), 1 "none": the
                              towers are (conv3x3 + bias, ReLU) x n, nn.Sequential conv index 2 i instead of 3 i (fcos.py:72-122,399) */
  int cg_tower_gn_mask;    /* CODE_GENERATOR.TOWER_LAYERS[i][0] == "GN"   -> bit i (sylph_config_default: all ones = ["GN", "ReLU"] layers); */
  int cg_tower_relu_mask;  /* CODE_GENERATOR.TOWER_LAYERS[i][1] == "ReLU" -> bit i.  A layer without norm / activation has no such module
                              in support_set_shared_tower, whose nn.Sequential indices advance per EXISTING module (code_generator.py:648-688) */
} sylph_config;

/* Fill cfg with the defaults of the COCO Meta-FCOS finetune yaml. */
void sylph_config_default(sylph_config* cfg);

/* Context.  dtype = SYLPH_BF16 (bf16 storage + MFMA, fp32 accumulate: the throughput mode), SYLPH_F32 (exact-fp32 MFMA: the reference
 * arithmetic, fcos_outputs.py:904-1028 within 1e-3 with identical NMS indices) or SYLPH_F32S (the parity mode at speed: fp32 storage
 * everywhere as in SYLPH_F32, every conv product as three bf16 MFMAs on operands split into bf16 hi + lo parts -- 2^-17 relative per
 * operand, the same 1e-3 bound). */
int sylph_ctx_create(int device_id, int dtype, sylph_ctx** out);
void sylph_ctx_destroy(sylph_ctx* ctx);
const char* sylph_last_error(void);
int sylph_set_stream(sylph_ctx* ctx, void* hip_stream);
int sylph_set_config(sylph_ctx* ctx, const sylph_config* cfg);

/* Weights: replaces DetectionCheckpointer(model).load (sylph/predictor.py:87-88).  name is the
 * reference state-dict key, data is host fp32 in the reference's tensor layout.
 * sylph_finalize_weights packs them for the kernels (K-major bf16/fp32, folded FrozenBN). */
int sylph_load_weight(sylph_ctx* ctx, const char* name, const float* data_host, const int64_t* shape, int ndim);
int sylph_finalize_weights(sylph_ctx* ctx);

/* MetaProposalNetwork.convert_batched_inputs_to_image_list
 * (sylph/modeling/meta_arch/meta_one_stage_detector.py:174-178): B device images, each (3,h,w) fp32
 * BGR 0-255 -> normalised, zero padded to a common size divisible by size_divisibility.
 * Host arrays: images_dev[B], heights[B], widths[B].  Returns the padded size.
 * bf16 mode: the normalisation is applied by the stem kernel of the next sylph_backbone_fpn call on the way into its LDS patch (same
 * arithmetic, bit-identical; no normalised copy of the batch is written) -- the images must stay valid and unchanged until that call
 * has been enqueued, on the same stream.  SYLPH_FUSE_PREPROCESS=0: a separate pass here. */
int sylph_preprocess(sylph_ctx* ctx, int B, const float* const* images_dev, const int* heights, const int* widths,
                     int* padded_h, int* padded_w);

/* Fused input pipeline: what SylphPredictor does on the host before the model call
 * (sylph/predictor.py:117-120,259-269: detectron2 ResizeShortestEdge -> ResizeTransform = PIL BILINEAR on the uint8 image,
 * RGB->BGR when INPUT.FORMAT is RGB, float CHW tensor) fused with convert_batched_inputs_to_image_list.  images_dev[b]:
 * device pointer to a (h, w, 3) uint8 HWC image (e.g. copied from a pinned host batch); new_heights/new_widths: the resize
 * targets.  Pillow's fixed-point two-pass resampling is reproduced bit for bit.  Makes the padded batch current. */
int sylph_preprocess_u8(sylph_ctx* ctx, int B, const unsigned char* const* images_dev, const int* heights, const int* widths,
                        const int* new_heights, const int* new_widths, int rgb_input, int* padded_h, int* padded_w);
/* Boundary/test entry: the normalised, padded network input of the current batch as (B,3,H,W) fp32 NCHW. */
int sylph_export_input(sylph_ctx* ctx, float* out_nchw_dev);

/* self.backbone(images.tensor) (meta_one_stage_detector.py:181,273): ResNet-FPN on the current batch. */
int sylph_backbone_fpn(sylph_ctx* ctx);

/* Boundary/test entry: make (B, padded H, W) the current batch and load its FPN pyramid from
 * nlevels device tensors (B,256,h_l,w_l) fp32 NCHW.  heights/widths: per-image unpadded sizes. */
int sylph_import_pyramid(sylph_ctx* ctx, int B, int padded_h, int padded_w, const int* heights, const int* widths,
                         const float* const* levels_nchw_dev);
/* Copy level `level` of the current pyramid to a (B,256,h_l,w_l) fp32 NCHW device tensor. */
int sylph_export_pyramid(sylph_ctx* ctx, int level, float* out_nchw_dev);

/* MetaFCOSHead.forward with support_set_per_class_code (sylph/modeling/meta_fcos/fcos.py:582-667;
 * CondConvBasic sylph/modeling/meta_fcos/head_utils.py:60-81).  cls_conv_dev: (N,256) fp32,
 * cls_bias_dev: (N) fp32 or NULL.  Results stay in the context (see sylph_export_head). */
int sylph_fcos_head(sylph_ctx* ctx, const float* cls_conv_dev, const float* cls_bias_dev, int N);
/* MetaFCOSHead.forward with support_set_per_class_code = None -> forward_base_train (fcos.py:543-578): the towers and the checkpoint's
 * OWN classifier `cls_logits` (nn.Conv2d(256, NUM_CLASSES, CLS_LOGITS_KERNEL_SIZE 1 or 3, padding k // 2), fcos.py:418-427) -- the base
 * detector (run_type None) and evaluation with the pretrained codes.  *num_classes receives NUM_CLASSES. */
int sylph_fcos_head_pretrained(sylph_ctx* ctx, int* num_classes);
/* logits (B,N,h,w), reg (B,4,h,w) = relu(scale_l*bbox_pred), ctr (B,1,h,w), iou (B,1,h,w); NULL skips. */
int sylph_export_head(sylph_ctx* ctx, int level, float* logits_nchw_dev, float* reg_nchw_dev, float* ctr_nchw_dev,
                      float* iou_nchw_dev);

/* Boundary/test entry, inverse of sylph_export_head: load level `level` of the head outputs of the current batch from
 * fp32 NCHW device tensors (logits (B,N,h,w), reg (B,4,h,w) already relu(scale*bbox_pred), ctr (B,1,h,w), iou (B,1,h,w);
 * NULL leaves that plane untouched), so that sylph_decode_nms can be driven with known head outputs (the
 * predict_proposals / ml_nms / detector_postprocess known-answer cases, fcos_outputs.py:904-1028). */
int sylph_import_head(sylph_ctx* ctx, int N, int level, const float* logits_nchw_dev, const float* reg_nchw_dev,
                      const float* ctr_nchw_dev, const float* iou_nchw_dev);

/* FCOSOutputs.predict_proposals + select_over_all_levels + detector_postprocess
 * (sylph/modeling/meta_fcos/fcos_outputs.py:743-812,904-1028; meta_one_stage_detector.py:288-296).
 * out_heights/out_widths (host, may be NULL = image size): the "height"/"width" of each input dict.
 * Device outputs, row-major [B][max_out][...]; counts_dev[B]; status_dev[1] (bit0 candidate
 * overflow, bit1 output truncated).  cand_dev: (level, location, class) ordinal of each detection. */
int sylph_decode_nms(sylph_ctx* ctx, const int* out_heights, const int* out_widths, int max_out, float* boxes_dev,
                     float* scores_dev, int* classes_dev, int* levels_dev, float* locations_dev, int* cand_dev,
                     int* counts_dev, int* status_dev);

/* CodeGenerator.forward / CodeGeneratorHead.forward_roi_align, eval branch
 * (sylph/modeling/code_generator/code_generator.py:924-1002): the current batch is the S support
 * images of ONE class; boxes_dev (S,4) fp32 XYXY (one gt box per image).  code_out_dev: 257 fp32 =
 * un-normalised cls_conv[256] ++ cls_bias[1].
 * With cg_type 1 this is ROIEncoder.forward (sylph/modeling/code_generator/roi_encoder.py:146-204,
 * S = EVAL_SHOT shots of one class): cls_conv[256] ++ cls_bias[1] (prior already added, no
 * normalisation step exists for this variant). */
int sylph_codegen(sylph_ctx* ctx, const float* boxes_dev, float* code_out_dev);
/* The same for SEVERAL classes in one batch (support-path throughput: C4 runs 866 classes x 5 shots through the R-101 backbone):
 * the current batch holds n_classes x shots support images, class k = images [k * shots, (k + 1) * shots); boxes_dev
 * (n_classes * shots, 4); codes_out_dev (n_classes, 257).  The reference computes one class per call
 * (meta_one_stage_detector.py:229-254); per class the arithmetic here is the same (ROIAlign, tower, GroupNorm and the shot mean
 * are per image / per class), only the launches are shared.  ROIEncoder (cg_type 1): shots = EVAL_SHOT; its encoder attends over
 * the class axis of a (classes, shots, C) tensor and sees one class per call at inference (roi_encoder.py:184-186), so a class of the
 * batch never meets another class's tokens here either: the codes are those of one call per class. */
int sylph_codegen_classes(sylph_ctx* ctx, const float* boxes_dev, int shots, float* codes_out_dev);
/* With cg_has_scale: the "cls_weight_norm" outputs of the last sylph_codegen[_classes] call (one fp32 per class), the factor
 * forward_normalize_code multiplies into the L2-normalised code (code_generator.py:838-840,987-993) -> pass them to
 * sylph_normalize_codes as weight_norm_dev. */
int sylph_codegen_weight_norm(sylph_ctx* ctx, float* weight_norm_out_dev);

/* Boundary/test entry: the ROIPooler call of the code generator alone (code_generator.py:341-348,928-930: box ->
 * level assignment -> ROIAlignV2 aligned, adaptive sampling, 7x7).  The current batch holds S images, boxes_dev (S,4)
 * one XYXY box per image; out_nchw_dev (S,256,7,7) fp32. */
int sylph_roi_align(sylph_ctx* ctx, const float* boxes_dev, float* out_nchw_dev);

/* CodeGeneratorHead.forward_normalize_code (code_generator.py:832-897): codes_dev (n,257) in place.
 * weight_norm_dev: (n) cls_weight_norm factors applied after the L2 normalisation (code_generator.py:838-840), or NULL. */
int sylph_normalize_codes(sylph_ctx* ctx, float* codes_dev, int n, const float* weight_norm_dev);

/* reduce_class_code + replace ordering (sylph/modeling/code_generator/utils.py:376-427; the cross-rank step of the
 * base-class "use all ground truths" path, sylph/evaluation/meta_learn_evaluation.py:118-254): rows_dev (n, row_ld) packed
 * chunk codes = cls_conv[256] | cls_bias | acc_weight | class id | valid | cls_weight_norm | has_weight_norm | payload;
 * the rows of a class are summed in row order (acc_weight in double).  divide_by_acc != 0: the cross-rank reduce (sums
 * divided by the accumulated weight when |1 - acc| > 1e-6, acc_weight := 1); == 0: the per-rank accumulation of
 * len/total_len-weighted chunk codes (meta_learn_evaluation.py:176-188; acc_weight := accumulated weight).
 * out_dev (num_classes, row_ld): row c = class id c (valid = 0 when no chunk of that class exists). */
int sylph_reduce_codes(sylph_ctx* ctx, const float* rows_dev, int n, int row_ld, float* out_dev, int num_classes,
                       int divide_by_acc);

/* The episode's one collective through the C ABI (MetaFCOSRunner._gather_class_code, sylph/runner/meta_fcos_runner.py:381-439,
 * which pickles dicts through all_gather_object): local_dev (n_local, 280) packed class-code rows of this rank (row layout of
 * sylph_reduce_codes + 16 name lanes) -> out_dev (world * capacity, 280): every rank's block in rank order on every rank, unused
 * rows zero (valid = 0).  ONE in-place ncclAllGather of equal-size blocks on the context's stream (RCCL over xGMI); capacity is the
 * statically known shard size ceil(n_classes / world), so there is no count exchange and no host read-back.  comm: an ncclComm_t --
 * from sylph_comm_init_rank below or the host's own RCCL communicator.  librccl is resolved at first use (dlopen): the library has
 * no link-time dependency on it.  The Python host side keeps torch.distributed (backend "nccl" = RCCL) as its default. */
int sylph_allgather_codes(sylph_ctx* ctx, void* comm, const float* local_dev, int n_local, int capacity, float* out_dev);
/* Communicator helpers for hosts without their own RCCL binding: rank 0 makes the 128-byte id (ncclGetUniqueId), hands it to the
 * other ranks over any channel, every rank calls init_rank (ncclCommInitRank on the context's device). */
int sylph_comm_unique_id(char* id_out_128);
int sylph_comm_init_rank(sylph_ctx* ctx, const char* id_128, int nranks, int rank, void** comm_out);
int sylph_comm_destroy(void* comm);

/* Primitive entries used by the kernel parity tests (F.conv2d / F.group_norm equivalents).
 * x: (B,C,H,W) fp32 NCHW device; w_host: (Cout,Cin,KH,KW) fp32 host; scale/shift host (Cout) or NULL;
 * residual: (B,Cout,Ho,Wo) device or NULL; y: (B,Cout,Ho,Wo) fp32 device. */
int sylph_conv2d(sylph_ctx* ctx, const float* x_nchw_dev, int B, int C, int H, int W, const float* w_host, int Cout,
                 int KH, int KW, int stride, int pad, const float* scale_host, const float* shift_host, int relu,
                 const float* residual_nchw_dev, float* y_nchw_dev);
int sylph_group_norm(sylph_ctx* ctx, const float* x_nchw_dev, int B, int H, int W, const float* gamma_host,
                     const float* beta_host, int relu, float* y_nchw_dev);

/* Kernel parity test entry for the dedicated ResNet stem kernels (bf16 contexts only): x (B,3,H,W) fp32 NCHW device,
 * already normalised; w_host (64,3,7,7), scale/shift host (64) = folded FrozenBN.  stem_out (B,64,H/2,W/2) =
 * relu(bn(conv7x7 s2 p3)), pool_out (B,64,H/4,W/4) = maxpool 3x3 s2 p1 of it; either may be NULL.
 * (detectron2 BasicStem; call site meta_one_stage_detector.py:181,273.) */
int sylph_stem_maxpool(sylph_ctx* ctx, const float* x_nchw_dev, int B, int H, int W, const float* w_host,
                       const float* scale_host, const float* shift_host, float* stem_out_nchw_dev, float* pool_out_nchw_dev);

/* Parity taps (test boundary; no reference counterpart -- the reference's modules are called one by one from Python, so its
 * tests can look at any intermediate; these entries give the parity tests the same view of the fused HIP graph).
 * sylph_set_debug_taps(1) before the first head call of a batch shape: every FCOS tower layer keeps its stored conv output in
 * its own buffer (same kernels and launches, other destination) instead of the two ping-pong buffers.
 * sylph_export_stage: output of ResNet stage `stage` (2..5 = res2..res5) of the last sylph_backbone_fpn as
 * (B, 256 << (stage - 2), h, w) fp32 NCHW (detectron2 ResNet.forward outputs; call site meta_one_stage_detector.py:181,273).
 * sylph_export_tower: conv output of layer `layer` of the cls (tower 0) / bbox (tower 1) tower on FPN level `level`
 * (sylph/modeling/meta_fcos/fcos.py:72-122,625-628) as (B,256,h_l,w_l) fp32 NCHW -- the value stored BEFORE GroupNorm when
 * the layer's GroupNorm is applied by its consumer -- and (coef_dev != NULL) that GroupNorm's per-(image, channel)
 * coefficients (a, b) with y = relu(a * x + b), as (B,256,2) fp32. */
int sylph_set_debug_taps(sylph_ctx* ctx, int on);
int sylph_export_stage(sylph_ctx* ctx, int stage, float* out_nchw_dev);
int sylph_export_tower(sylph_ctx* ctx, int tower, int layer, int level, float* y_nchw_dev, float* coef_dev);

/* Kernel parity entry: ONE detectron2 BottleneckBlock (1x1 -> 3x3 -> 1x1, FrozenBN folded to scale/shift, identity or
 * projection shortcut, ReLU) through the same launches sylph_backbone_fpn uses for such a block (fused kernels included).
 * x (B,Cin,H,W) fp32 NCHW device; w_host[4] = conv1 (mid,Cin,1,1), conv2 (mid,mid,3,3), conv3 (cout,mid,1,1), shortcut
 * (cout,Cin,1,1) or NULL (identity block); scale_host / shift_host[4]: per-output-channel FrozenBN scale and shift of the
 * same four convs (host).  y (B,cout,Ho,Wo) fp32 NCHW device.  (call site meta_one_stage_detector.py:181,273) */
int sylph_bottleneck(sylph_ctx* ctx, const float* x_nchw_dev, int B, int Cin, int H, int W, int stride, int mid, int cout,
                     const float* const* w_host, const float* const* scale_host, const float* const* shift_host,
                     float* y_nchw_dev);

/* Kernel parity entry: one FPN lateral as sylph_backbone_fpn launches it (detectron2 FPN.forward: lateral 1x1 conv + bias, plus
 * the nearest-2x upsampled level above, fused as a residual; call site meta_one_stage_detector.py:181,273).  x (B,C,H,W) fp32 NCHW
 * device; w_host (256,C,1,1), bias_host (256); top (B,256,H/2,W/2) device or NULL (fpn_lateral5); y (B,256,H,W) fp32 NCHW device. */
int sylph_fpn_lateral(sylph_ctx* ctx, const float* x_nchw_dev, int B, int C, int H, int W, const float* w_host, const float* bias_host,
                      const float* top_nchw_dev, float* y_nchw_dev);

/* Bytes of device memory currently held by the context (weights + workspace). */
int64_t sylph_device_bytes(sylph_ctx* ctx);

/* Measurement aid (no reference counterpart; the reference only logs wall-clock s/img,
 * sylph/evaluation/meta_learn_evaluation.py:392-463): when enabled, every launch of the MFMA
 * implicit-GEMM conv kernel is bracketed by HIP events on the launch stream.  sylph_profile_read
 * synchronises the stream and returns the summed kernel time (ms), the algorithmic FLOPs
 * (2*M*N*K of the logical problem) and the number of launches since the previous read. */
int sylph_profile_enable(sylph_ctx* ctx, int on);
/* Kernel micro-benchmark (tuning aid): time `iters` back-to-back launches of one conv layer
 * (B images of HxW, Cin->Cout, KxK, random bf16/fp32 operands) with HIP events; ms per launch and the
 * algorithmic FLOPs of one launch are returned. */
int sylph_bench_conv(sylph_ctx* ctx, int B, int H, int W, int Cin, int Cout, int K, int stride, int pad, int has_res,
                     int relu, int with_gn, int iters, float* ms_out, double* flops_out);
int sylph_profile_read(sylph_ctx* ctx, double* conv_ms, double* conv_flops, int64_t* conv_launches);
/* The same records grouped by kernel (consumes them like sylph_profile_read): names_out = max_kernels x 64 bytes (NUL-terminated
 * kernel names in first-launch order), ms / flops / launches per kernel; *n_out kernels were written. */
int sylph_profile_read_kernels(sylph_ctx* ctx, int max_kernels, char* names_out, double* ms_out, double* flops_out,
                               int64_t* launches_out, int* n_out);

#ifdef __cplusplus
}
#endif
#endif /* SYLPH_HIP_H */
