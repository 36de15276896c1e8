/*
 * opd_detr.h — C-ABI of the MI355X-native DETR person-detection forward pass (libopd_hip.so).
 *
 * The reference (Kizuna42/office-person-detection-vit) is 100 % Python and has no FFI: its Phase-2 detector is a
 * Python class wrapping Hugging Face `DetrForObjectDetection` (deleted `src/detection/vit_detector.py`, method map
 * in `coverage.json:1`; today's stand-in with the same surface is `src/detection/yolov8_detector.py:19-254`).  The
 * entry points below are what a ctypes binding of that class would call; each one names the reference interface it
 * replaces.  Plain pointers and sizes only — no torch / numpy types cross this boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative OPD_E* code; it never throws and never aborts.  The
 *     human-readable reason of the last failure on the calling thread is `opd_last_error()`.  The Python shim turns a
 *     non-zero code into `RuntimeError`, which preserves the reference's conventions: load failure ->
 *     RuntimeError("Failed to load ... model: ...") (`yolov8_detector.py:86-88`), inference errors re-raised
 *     (`:130-132`) and converted to an empty detection list by the phase (`src/pipeline/phases/detection.py:124-127`).
 *   - output buffers are caller-allocated; the library owns only the model, its workspace and its HIP stream.
 *   - single caller at a time per handle (the reference is strictly single-threaded).  DIFFERENT handles may
 *     be driven from different threads at once (throughput mode: several handles per GPU); the library serialises the one
 *     operation that needs it, hipGraph capture, internally.
 *   - tensors: logits [B][Q][C+1] f32, boxes [B][Q][4] f32 (cx,cy,w,h in [0,1]), encoder features [B][h*w][D] f32
 *     with h = ceil(H/32), w = ceil(W/32) (HF `DetrObjectDetectionOutput`: logits, pred_boxes,
 *     encoder_last_hidden_state; HF:models/detr/modeling_detr.py:1329-1443).
 */
#ifndef OPD_DETR_H
#define OPD_DETR_H

#include <stddef.h>
#include <stdint.h>

/* The product library is built with -fvisibility=hidden: exactly the functions declared here are exported. */
#if defined(__GNUC__)
#define OPD_API __attribute__((visibility("default")))
#else
#define OPD_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

#define OPD_OK 0
#define OPD_EINVAL (-1)   /* bad argument (null pointer, shape out of the configured maximum, ...) */
#define OPD_EIO (-2)      /* weight file missing / unreadable / malformed */
#define OPD_ESCHEMA (-3)  /* weight file lacks a tensor of the DETR state dict or has the wrong shape */
#define OPD_EHIP (-4)     /* a HIP runtime call failed (message carries hipGetErrorString) */
#define OPD_ENOMEM (-5)   /* device or host allocation failed */
#define OPD_ESTATE (-6)   /* call not valid in the handle's current state */

typedef struct opd_detr opd_detr; /* opaque model handle */

/* Input pixel formats accepted by the forward entry points. */
#define OPD_PIXELS_U8_BGR_HWC 0 /* the reference's frame: numpy uint8 [H][W][3], BGR (`yolov8_detector.py:90-96`) */
#define OPD_PIXELS_F32_NCHW 1   /* HF `pixel_values`: float32 [B][3][H][W], already normalised */

/* Where the caller's buffers live. */
#define OPD_MEM_HOST 0
#define OPD_MEM_DEVICE 1 /* pointers are HIP device pointers on the handle's device */
#define OPD_MEM_HOST_PIXELS_DEVICE_OUT 2 /* pixels in host memory, every output pointer a device pointer: a sharded caller uploads
                                          * camera frames and hands the records straight to an RCCL all-gather (sharding.py) */

typedef struct opd_config {
    int32_t struct_size; /* = sizeof(opd_config), for forward compatibility */
    int32_t max_batch;   /* workspace is sized for max_batch frames of max_height x max_width pixels, in either orientation:   */
    int32_t max_height;  /* a call may pass any H x W with H, W <= max(max_height, max_width) and H * W <= max_height * max_width */
    int32_t max_width;   /* (the HF size rule maps a portrait camera frame to about 1333 x 750)                                    */
    int32_t flags;       /* OPD_FLAG_* */
    int32_t reserved[3];
} opd_config;

#define OPD_FLAG_NO_GRAPH 1 /* launch kernels eagerly instead of replaying a captured hipGraph */
#define OPD_FLAG_BF16 4 /* bf16 instead of fp16 as the 16-bit operand type of every activation buffer and GEMM weight (the type BASELINE.json
                         * configs[1] names; same MFMA rate on gfx950, 8 mantissa bits instead of 11: the boxes drift more, see DESIGN.md
                         * section 3).  The decoder's linear layers keep their split fp16 operands (fp32-grade) in both modes. */
#define OPD_FLAG_MULTI_STREAM 2 /* this handle is one of several that keep batches in flight on one GPU: kernel choices are made for
                                 * throughput (a launch's last, partly filled round overlaps other handles' work) rather than for latency */

/* One detection record (32 bytes).  Boxes are (x1,y1,x2,y2) in pixels of the ORIGINAL frame, as produced by
 * HF `post_process_object_detection` (HF:models/detr/image_processing_detr.py:805-856). */
typedef struct opd_det {
    float x1, y1, x2, y2;
    float score;
    int32_t label;       /* argmax over the first C classes ("no object" excluded) */
    int32_t query_index; /* decoder query that produced it (`Detection.query_index`, src/models/data_models.py:38) */
    int32_t frame;       /* index of the frame inside the batch */
} opd_det;

/* Architecture facts read back from a loaded model. */
typedef struct opd_model_info {
    int32_t depths[4];
    int32_t d_model, heads, ffn_dim, encoder_layers, decoder_layers, num_queries, num_classes_plus1;
    int32_t max_batch, max_height, max_width;
    int32_t device_ordinal;
    int64_t weight_bytes_device;
    int64_t workspace_bytes_device;
} opd_model_info;

/* Replaces `ViTDetector.__init__` + `load_model()` (deleted vit_detector.py 36-44, 81-99; same contract as
 * `YOLOv8Detector.load_model`, yolov8_detector.py:70-88): parse a safetensors checkpoint carrying the HF
 * `DetrForObjectDetection` state dict (5.x or 4.x key names), fold every FrozenBN into its convolution in fp32
 * (HF:models/detr/modeling_detr.py:179-215), convert to the device layouts and upload to GPU `device_ordinal`. */
OPD_API int opd_detr_create(const opd_config* cfg, const char* weights_path, int device_ordinal, opd_detr** out);

/* Replaces `cleanup_resources` / detector release (`src/utils/memory_utils.py:33-41`). */
OPD_API void opd_detr_destroy(opd_detr* m);

/* A second handle on the SAME model: own stream, workspace and graph cache, the weights in HBM shared with `src` (read-only after
 * creation; freed when the last handle that uses them is destroyed, in any order).  For several batches in flight on one GPU
 * (`HipDetrDetector(streams=N)`, bench.py): no second parse of the checkpoint, one copy of the 83 MB of weights for L2 and the
 * Infinity Cache to hold instead of N.  No reference counterpart (one detector object per process there). */
OPD_API int opd_detr_clone(const opd_detr* src, opd_detr** out);

OPD_API int opd_detr_info(const opd_detr* m, opd_model_info* info);

/* Replaces `with torch.no_grad(): outputs = model(pixel_values, pixel_mask)` in `ViTDetector.detect_batch`
 * (deleted vit_detector.py 508-550; HF:models/detr/modeling_detr.py:1329-1443), including the BGR->RGB / 1/255 /
 * ImageNet mean-std step of `_preprocess_batch` (562-578) when `pixel_format == OPD_PIXELS_U8_BGR_HWC`.
 * All B frames share one size HxW (pixel_mask all ones).  `enc_features` may be NULL.
 * `mem_kind` says whether pixels AND outputs are host or device pointers.  Synchronous: results are complete on
 * return. */
OPD_API int opd_detr_forward(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W,
                     float* logits, float* boxes, float* enc_features);

/* The same call for a RAGGED batch: `model(pixel_values, pixel_mask)` with a padding mask, i.e. what HF's
 * `DetrImageProcessor.pad` + the mask paths of the model do when the frames of a batch differ in size
 * (HF:models/detr/image_processing_detr.py:639-668 zero-pad after normalisation + pixel_mask;
 * HF:models/detr/modeling_detr.py:283-289 nearest down-sampling of the mask, 294-368 position embedding from the mask's
 * cumulative sums, 402-427 additive key mask in encoder self-attention and decoder cross-attention).
 * `pixels` is the H x W canvas batch with every frame in its top-left corner; `valid_hw` = host [B][2] int32 (height, width)
 * of each frame inside the canvas (NULL or all (H, W): identical to opd_detr_forward).  Canvas pixels outside a frame are
 * ignored (written as zeros on the device).  `enc_features` covers the whole canvas map, padded positions included. */
OPD_API int opd_detr_forward_ragged(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W,
                            const int32_t* valid_hw, float* logits, float* boxes, float* enc_features);

/* Device-side resize: frames at CAMERA resolution in, the resize half of `_preprocess_batch` on the GPU.
 * `frames` = [B][h][w][3] uint8 BGR (host or device per `mem_kind`), all of one size; they are resized to H x W — the caller
 * computes (H, W) with the HF size rule (HF:image_transforms.py:206-242; `model_input_size` in the Python shim) — by a kernel
 * that is BIT-EXACT with Pillow's 8-bit bilinear resampler, i.e. with `DetrImageProcessor.resize`
 * (HF:models/detr/image_processing_detr.py:424-436), then run through the same path as opd_detr_forward.
 * Outputs follow `mem_kind`. */
OPD_API int opd_detr_forward_resized(opd_detr* m, const uint8_t* frames, int mem_kind, int B, int h, int w, int H, int W,
                             float* logits, float* boxes, float* enc_features);
/* The resize alone (host in, host out): [B][h][w][3] -> [B][out_h][out_w][3]; for parity tests against Pillow. */
OPD_API int opd_detr_resize_u8(opd_detr* m, const uint8_t* frames, int B, int h, int w, int out_h, int out_w, uint8_t* out);

/* Replaces `_postprocess_batch` part 1 = HF `post_process_object_detection` (deleted vit_detector.py 591-647;
 * HF:models/detr/image_processing_detr.py:805-856): softmax over C+1, max over the first C classes, cxcywh->xyxy,
 * scale by the ORIGINAL (height,width) of each frame, keep score > threshold.  Runs on the device on the logits and
 * boxes of the LAST forward of this handle.  `orig_hw` = [B][2] int32 host array (height, width) or NULL (= model
 * input size).  `out` (host) receives `counts[b]` records for frame b at out[b*Q ...]; records keep query order. */
OPD_API int opd_detr_postprocess(opd_detr* m, float threshold, const int32_t* orig_hw, opd_det* out, int32_t* counts);

/* One call = forward + device post-process, pixels in HBM when `mem_kind == OPD_MEM_DEVICE` (the benchmark's
 * timed region).  Equivalent to opd_detr_forward(..., NULL, NULL, NULL) followed by opd_detr_postprocess(...). */
OPD_API int opd_detr_detect(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W,
                    float threshold, const int32_t* orig_hw, opd_det* out, int32_t* counts);
/* With `mem_kind == OPD_MEM_DEVICE`, `out` ([B][Q] records) and `counts` ([B]) are DEVICE pointers too, so a sharded
 * caller can hand them straight to an RCCL all-gather; `orig_hw` is always a host array. */
/* Asynchronous submission for throughput-oriented callers: enqueues the forward and the post-process on the handle's stream
 * and returns with a `ticket` (0..3; at most 4 submissions may be outstanding per handle: a fifth returns OPD_ESTATE until the
 * oldest ticket has been waited for, and waiting twice for one ticket is OPD_ESTATE too); `out` / `counts` of a submission
 * are complete after opd_detr_wait(m, ticket) (`mem_kind` as in opd_detr_detect: host pixels are staged with an asynchronous
 * copy, host outputs travel through pinned memory and are delivered by opd_detr_wait; `pixels`, `out` and `counts` must stay
 * valid and untouched until that wait returns).  Work of consecutive submissions to ONE
 * handle runs in submission order; SEVERAL handles on one device (own stream, workspace and captured graph each) overlap:
 * the latency-bound tail of one batch fills the CUs the next batch's trunk leaves idle (+25 % frames/s with three handles at
 * batch 8). */
OPD_API int opd_detr_detect_async(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float threshold,
                          const int32_t* orig_hw, opd_det* out, int32_t* counts, int* ticket);
OPD_API int opd_detr_wait(opd_detr* m, int ticket);
/* Camera-resolution form (see opd_detr_forward_resized): resize on the device, detect, and scale the boxes back to the
 * camera frame size (h, w). */
OPD_API int opd_detr_detect_resized(opd_detr* m, const uint8_t* frames, int mem_kind, int B, int h, int w, int H, int W,
                            float threshold, opd_det* out, int32_t* counts);
/* The same for a LIST of host frames, one pointer per frame (what the reference hands a detector: detect_batch(frames: List[np.ndarray]),
 * deleted vit_detector.py:508; DetectorPort.detect(frames: Sequence[FrameDTO]), src/core/interfaces.py:30-34): frames[b] = [h][w][3] uint8
 * BGR, all of one size, each uploaded from where it lies -- no stacked copy of the batch on the host (25.6 MB at batch 8: 0.75 ms of a
 * 3.4-ms step).  (h, w) == (H, W): no resize.  mem_kind: OPD_MEM_HOST or OPD_MEM_HOST_PIXELS_DEVICE_OUT (the frames are host memory). */
OPD_API int opd_detr_detect_frames(opd_detr* m, const uint8_t* const* frames, int mem_kind, int B, int h, int w, int H, int W,
                                   float threshold, opd_det* out, int32_t* counts);
/* `detect_with_features(frame)` in ONE call (yolov8_detector.py:134-159: detect, then pool the encoder map under every detection's box,
 * src/tracking/feature_extractor.py:39-88): opd_detr_detect_frames on host frames + the appearance feature of every record of class `label`
 * (ROI mean-pool + L2 norm exactly as opd_detr_roi_features computes it for the record's box), pooled while the records are still on the
 * device -- one host wait instead of two.  features: [B][num_queries][d_model] fp32 (host), row = the record's query_index; rows of queries
 * without a record of that class are NOT written (whatever the buffer held).  Suppression (opd_person_nms) happens afterwards on the host: a suppressed record's row is simply unused. */
OPD_API int opd_detr_detect_frames_features(opd_detr* m, const uint8_t* const* frames, int B, int h, int w, int H, int W, float threshold,
                                            int label, opd_det* out, int32_t* counts, float* features);
/* Ragged-batch form (see opd_detr_forward_ragged); `valid_hw` and `orig_hw` are host arrays. */
OPD_API int opd_detr_detect_ragged(opd_detr* m, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W,
                           const int32_t* valid_hw, float threshold, const int32_t* orig_hw, opd_det* out, int32_t* counts);

/* Replaces `_postprocess_batch` part 2 (deleted vit_detector.py 591-647; `docs/plan.md:30`,
 * `config.yaml.disabled:38`): keep `label == person_label` (pass -1 to keep every class), greedy IoU-NMS in
 * descending score order at `nms_threshold` (pass >= 1 to disable).  Host-side, in place: compacts `dets[0..n)` and
 * returns the new count (>= 0) or a negative error. */
OPD_API int opd_person_nms(opd_det* dets, int n, int person_label, float nms_threshold);
/* The same for a whole batch in one call: frame f owns `dets[f * stride .. f * stride + counts[f])` (the fixed-slot layout
 * opd_detr_detect writes, stride = num_queries); every frame is compacted in place and `counts[f]` becomes its new count.
 * A negative `counts[f]` (padding slot of an uneven shard) is left alone.  Returns 0 or a negative error. */
OPD_API int opd_person_nms_batch(opd_det* dets, int32_t* counts, int n_frames, int stride, int person_label, float nms_threshold);

/* Page-locked host memory for frame batches handed over with OPD_MEM_HOST: the upload of such a buffer is one asynchronous DMA
 * instead of the runtime's staged copy of pageable memory.  Optional (any host pointer is accepted everywhere); the Python shim
 * stacks the caller's frames (`detect_batch(frames: list[np.ndarray])`, the reference's calling convention) directly into it. */
OPD_API int opd_host_alloc(size_t bytes, void** out);
OPD_API void opd_host_free(void* p);

/* "Next" row: replaces `SimilarityCalculator.compute_similarity_matrix` / `compute_distance_matrix`
 * (`src/tracking/similarity.py:42-220`), the tracker's cost matrix built right after the detect path:
 * similarity[i][j] = clip((aw * clip(f1_i . f2_j, -1, 1) + mw * IoU(box1_i, box2_j)) / (weights used), 0, 1); a row without
 * features (has == 0, or a NULL feature matrix) drops the appearance term and renormalises.  Boxes are (x, y, w, h).
 * All pointers are host pointers; `out` = [n1][n2] f32; `as_distance` != 0 returns 1 - similarity.  Runs on the device
 * `device_ordinal`; needs no model handle. */
OPD_API int opd_similarity_matrix(int device_ordinal, const float* feats1, const float* boxes1, const uint8_t* has1, int n1,
                          const float* feats2, const float* boxes2, const uint8_t* has2, int n2, int D,
                          double appearance_weight, double motion_weight, int as_distance, float* out);

/* Replaces `FeatureExtractor.extract_roi_features` on the DETR encoder map
 * (`src/tracking/feature_extractor.py:39-88`; deleted vit_detector.py 224-273): for each (x,y,w,h) box in original
 * pixels, mean-pool the last forward's encoder map of frame `frame` over the int-truncated, clamped ROI and
 * L2-normalise (x / (||x|| + 1e-8)).  `features` = host [n][D] f32.  Runs on the device. */
OPD_API int opd_detr_roi_features(opd_detr* m, int frame, const float* boxes_xywh, int n, int orig_h, int orig_w,
                          float* features);

/* Replaces `get_attention_map` / `_extract_attention_map` (deleted vit_detector.py 392-446, `coverage.json:1`; the surviving stand-in
 * returns None, `src/detection/yolov8_detector.py:243-254`; consumer: `Visualizer.draw_attention_map`, `src/visualization/visualizer.py:148-200`,
 * which takes a 1-D or 2-D array in [0, 1]).  The DETR-era source is gone, so the definition is this build's: the decoder's
 * cross-attention weights of layer `layer` (negative: counted from the last) of the LAST forward's frame `frame`, averaged over the
 * heads and over the `n_queries` query indices in `queries` (n_queries == 0: all queries).  `out` = host [fh * fw] f32, row-major over
 * the feature map of that forward (fh = ceil(H/32), fw = ceil(W/32)), summing to 1; `out_capacity` = floats the caller allocated: a
 * map larger than that is OPD_EINVAL, nothing is written.  Runs on the device from the q / k operands the forward left there, so the
 * call belongs to the caller that ran that forward: no other submission to this handle in between (single caller per handle). */
OPD_API int opd_detr_attention_map(opd_detr* m, int frame, int layer, const int32_t* queries, int n_queries, float* out, int out_capacity);

/* ---- multi-GPU: the path's one exchange step ------------------------------------------------------------------------------
 * The reference has no distributed mode; the port it reserves for batched detectors is `DetectorPort.detect(frames: Sequence[FrameDTO])`
 * (`src/core/interfaces.py:30-34`, home `src/adapters/__init__.py:1`).  Frames are independent, so the path shards by frame: one process
 * per GPU, each with its own detector handle, and ONE RCCL all-gather of fixed-size records per exchange (opd_det x num_queries per frame
 * slot + one int32 count per slot; count -1 = the slot holds no frame).  An `opd_comm` is a LANE: one detector handle's send / receive
 * buffers and events on a communicator that all lanes of the rank share (opd_comm_create makes the communicator and its first lane,
 * opd_comm_attach further lanes for further handles).  A step: forward -> post-process (writes the lane's send buffer) on the HANDLE's stream;
 * ncclAllGather -> copy to page-locked host memory on the COMMUNICATOR's own stream, which waits for the handle's stream through an event;
 * one host wait (opd_comm_wait).  All all-gathers of a rank are enqueued on that one stream in the order opd_comm_exchange was called: every
 * rank must call it in the same order across its lanes.  RCCL is resolved at run time (librccl.so.1); single-GPU callers never load it.
 * The 128-byte unique id travels from rank 0 to the other ranks by whatever launched them (file, socket, MPI, a torch.distributed store):
 * the library does no rendezvous.  ncclCommInitRank blocks until every rank has arrived: callers check opd_comm_available() on every rank
 * and agree on it first, and bound the set-up with a watchdog of their own (sharding.py, bench.py). */
typedef struct opd_comm opd_comm;
#define OPD_COMM_ID_BYTES 128
OPD_API int opd_comm_available(void);          /* OPD_OK when librccl could be resolved in this process: every rank checks BEFORE any rank enters the collective set-up */
OPD_API int opd_comm_unique_id(void* id128);   /* rank 0 only: ncclGetUniqueId */
/* collective over the `world` ranks (ncclCommInitRank): every rank passes the same id, its rank, and its OWN handle (device and stream) */
OPD_API int opd_comm_create(const void* id128, int rank, int world, opd_detr* m, opd_comm** out);
/* A further LANE on `parent`'s communicator for another detector handle of the same device (a rank that keeps several batches in flight):
 * own send / receive buffers and events, the SAME RCCL communicator.  All all-gathers of a rank are enqueued on the communicator's own
 * stream in the order opd_comm_exchange was called, so every rank must call it in the same order across its lanes.  Local, not collective. */
OPD_API int opd_comm_attach(opd_comm* parent, opd_detr* m, opd_comm** out);
/* Lanes may be destroyed in any order; the communicator goes with the last one.  Destroying the detector handle first is allowed: its lanes
 * then return OPD_ESTATE from every call and only opd_comm_destroy remains valid on them. */
OPD_API void opd_comm_destroy(opd_comm* c);
OPD_API int opd_comm_info(const opd_comm* c, int* rank, int* world);
/* A lane holds up to TWO exchanges: while one travels (issued, not yet waited for) the next may be begun, filled and issued; opd_comm_wait
 * delivers them oldest first; a third opd_comm_begin returns OPD_ESTATE.
 * One exchange = begin(slots) ; detect(slot0, frames ...) once or several times (a shard larger than max_batch goes in chunks) ; exchange ;
 * wait.  `slots` = frame slots per rank, the same number on every rank (an uneven shard leaves trailing slots at count -1).
 * opd_comm_detect = opd_detr_detect_async into the send buffer at slot0 (arguments as there); nothing synchronises before opd_comm_wait,
 * which delivers every rank's records: out_all [world][slots][num_queries], counts_all [world][slots] (host). */
OPD_API int opd_comm_begin(opd_comm* c, int slots);
OPD_API int opd_comm_detect(opd_comm* c, int slot0, const void* pixels, int pixel_format, int mem_kind, int B, int H, int W, float threshold,
                            const int32_t* orig_hw);
/* Device pointers of slot `slot0` of the send buffer (records [.][num_queries], counts [.]): for callers that fill the slots through another
 * entry point with device outputs (opd_detr_detect_resized / _ragged with OPD_MEM_HOST_PIXELS_DEVICE_OUT) on the SAME handle. */
OPD_API int opd_comm_buffers(opd_comm* c, int slot0, void** records, void** counts);
OPD_API int opd_comm_exchange(opd_comm* c);
OPD_API int opd_comm_wait(opd_comm* c, opd_det* out_all, int32_t* counts_all);

/* Device time (ms) of the last DETECT call (opd_detr_detect and its variants: forward + post-process) per stage, measured with HIP events on the handle's stream:
 * [0] preprocess+stem+pool, [1] stage1, [2] stage2, [3] stage3, [4] stage4, [5] projection+encoder, [6] decoder+heads,
 * [7] post-process.  Only filled when profiling is on: opd_detr_set_profiling(m, 1) = the forward is launched EAGERLY with an event pair
 * around every launch (opd_detr_kernel_times; the eager launches and their events add ~10 us per launch to the stage times), (m, 2) =
 * the stage marks are recorded INSIDE the replayed hipGraph: the stage times of the path as a caller gets it (no per-kernel times). */
OPD_API int opd_detr_set_profiling(opd_detr* m, int enabled);
OPD_API int opd_detr_stage_times(const opd_detr* m, float* ms8);
/* Per-kernel-class totals of the last profiled forward, from HIP event pairs recorded around EVERY launch on the
 * handle's stream: class 0 = conv_gemm_kernel on backbone convolutions (+ input projection), 1 = conv_gemm_kernel as
 * transformer linear layer, 2 = attention_kernel, 3 = reserved.  ms4[c] = summed device time, launches4[c] = number of
 * launches, flops4[c] = summed ALGORITHMIC FLOPs (2 x MAC) of those launches. */
OPD_API int opd_detr_kernel_times(const opd_detr* m, float* ms4, int32_t* launches4, double* flops4);
/* The same event pairs by KERNEL: one row per kernel instantiation of the last profiled forward (mode 1), longest first -- `name` as rocprofv3
 * prints it minus namespace and signature (e.g. "btail256_kernel<256, false>"), so a row can be checked against a kernel-trace summary;
 * launches, summed device time (ms) and summed algorithmic FLOPs (0 for launches that are not GEMM-shaped).  *count = rows available;
 * at most `capacity` are written. */
typedef struct opd_kernel_stat { char name[96]; int32_t launches; float ms; double flops; } opd_kernel_stat;
OPD_API int opd_detr_kernel_table(const opd_detr* m, opd_kernel_stat* out, int capacity, int* count);

/* Thread-local description of the last error returned on this thread ("" if none). */
OPD_API const char* opd_last_error(void);

/* Library / kernel build identification, e.g. "opd_hip 0.1 gfx950". */
OPD_API const char* opd_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OPD_DETR_H */
