"""``HipDetrDetector`` — the reference's Phase-2 detector surface on top of libopd_hip.so.

Drop-in for the class the pipeline instantiates in ``DetectionPhase.initialize``
(``src/pipeline/phases/detection.py:34-54``): same constructor keywords, attributes and methods as the DETR-era
``ViTDetector`` (deleted ``src/detection/vit_detector.py``; method map in ``coverage.json:1``; its stand-in with the
identical interface is ``src/detection/yolov8_detector.py:19-254``) and the same error conventions:

* ``detect*`` before ``load_model`` -> ``RuntimeError("Model not loaded. Call load_model() first.")``
  (``yolov8_detector.py:102-103``),
* load failure -> ``RuntimeError("Failed to load DETR model: ...")`` chained (``:86-88``),
* inference errors are logged and re-raised (``:130-132``) — the phase turns them into an empty list per frame
  (``detection.py:124-127``).

All arithmetic runs in hand-written HIP kernels behind the C-ABI of ``include/opd_detr.h``; there is no CPU or PyTorch
fallback — without the built library or without a GPU every call raises.
"""

from __future__ import annotations

from concurrent.futures import ThreadPoolExecutor
import ctypes as C
import logging
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _capi
from .data_models import Detection
from .feature_extractor import FeatureExtractor

logger = logging.getLogger(__name__)

PERSON_LABEL = 1  # COCO "person" in DETR's label space (tests/test_detection_phase.py:71 uses class_id=1)
DEFAULT_MODEL_NAME = "facebook/detr-resnet-50"


def model_input_size(height: int, width: int, shortest_edge: int = 800, longest_edge: int = 1333) -> Tuple[int, int]:
    """HF DETR size rule (HF:image_transforms.py:206-242): shortest edge -> 800 unless the longest would pass 1333."""
    size, raw_size = shortest_edge, None
    mn, mx = float(min(height, width)), float(max(height, width))
    if mx / mn * size > longest_edge:
        raw_size = longest_edge * mn / mx
        size = int(round(raw_size))
    if (height <= width and height == size) or (width <= height and width == size):
        return height, width
    if width < height:
        return (int(raw_size * height / width) if raw_size is not None else int(size * height / width)), size
    return size, (int(raw_size * width / height) if raw_size is not None else int(size * width / height))


def resize_frame(frame: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL bilinear resize of a uint8 HxWx3 frame, exactly what HF's image processor does on the host
    (HF:models/detr/image_processing_pil_detr.py:497-549).  Used for batches of mixed frame sizes and as the reference
    of the device-side resize (``opd_detr_*_resized``, bit-exact with this function)."""
    if frame.shape[0] == out_h and frame.shape[1] == out_w:
        return frame
    from PIL import Image

    return np.asarray(Image.fromarray(frame).resize((out_w, out_h), resample=Image.BILINEAR))


class HipDetrDetector:
    """DETR person detector running on one MI355X through libopd_hip.so."""

    def __init__(
        self,
        model_name: str = DEFAULT_MODEL_NAME,
        confidence_threshold: float = 0.5,
        device: Optional[str] = None,
        nms_threshold: float = 0.4,
        model_path: Optional[str] = None,
        iou_threshold: Optional[float] = None,
        max_batch: int = 8,
        max_size: Tuple[int, int] = (800, 1333),
        resize: bool = True,
        device_resize: bool = True,
        use_graph: bool = True,
        streams: int = 1,
        pinned_staging: bool = True,
        dtype: str = "fp16",
        frame_lists: bool = True,
    ):
        """
        Args mirror ``config.yaml.disabled:33-44`` (``model_name``, ``confidence_threshold``, ``nms_threshold``,
        ``device``, ``batch_size`` -> ``max_batch``).  ``model_path`` is a local ``.safetensors`` file with the HF
        ``DetrForObjectDetection`` state dict (or a directory holding ``model.safetensors``); hub loading by NAME is
        impossible offline, so ``model_name`` alone resolves only through ``$OPD_DETR_WEIGHTS``.
        ``iou_threshold`` is the keyword ``DetectionPhase.initialize`` passes (``src/pipeline/phases/detection.py:47-52``:
        ``model_path``, ``confidence_threshold``, ``device``, ``iou_threshold``): the same knob as ``nms_threshold``, which it
        overrides when given.
        ``device``: ``"hip"``, ``"hip:N"``, ``"cuda"``, ``"cuda:N"`` or None (= GPU 0).  ``"cpu"``/``"mps"`` are refused.
        ``streams``: detector handles (each with its own HIP stream and workspace; the weights are shared) that
        ``detect_batch`` keeps busy at once when a call spans several ``max_batch`` chunks; 1 = strictly serial.
        ``dtype``: 16-bit operand type of the device path: ``"fp16"`` (default: meets the 1e-3 box tolerance) or ``"bf16"``
        (``OPD_FLAG_BF16``: the type the DETR-era config's deployment target names; same speed, 8 mantissa bits: boxes drift ~4x more).
        ``pinned_staging``: stack the caller's frames into page-locked memory (``opd_host_alloc``) so that the upload is
        one DMA; False stacks into ordinary numpy memory.
        ``frame_lists``: hand a chunk of same-sized contiguous frames to the library as a list of pointers
        (``opd_detr_detect_frames``: each frame is uploaded from where it lies) instead of stacking it first; False always stacks.
        """
        self.model_name = model_name
        self.model_path = model_path
        self.confidence_threshold = confidence_threshold
        if iou_threshold is not None:
            nms_threshold = float(iou_threshold)
        self.nms_threshold = nms_threshold
        self.iou_threshold = nms_threshold  # the YOLO-era spelling of the same knob
        self.device = self._setup_device(device)
        self.max_batch = int(max_batch)
        self.max_size = (int(max_size[0]), int(max_size[1]))
        self.resize = resize
        self.device_resize = device_resize  # resize camera-resolution frames on the GPU (False: PIL on the host)
        self.use_graph = use_graph  # replay the forward as a captured hipGraph (False: launch every kernel eagerly)
        if dtype not in ("fp16", "bf16"):
            raise ValueError("dtype must be 'fp16' or 'bf16'")
        self.dtype = dtype
        self.streams = max(1, int(streams))
        self.model: Optional[int] = None  # opaque opd_detr* once loaded (handle 0)
        self._handles: List[int] = []  # all handles, handle 0 first
        self.pinned_staging = bool(pinned_staging)
        self.frame_lists = bool(frame_lists)
        self._staging: dict = {}  # handle slot -> (pinned pointer, capacity in bytes)
        self.feature_extractor = FeatureExtractor()
        self._lib = None
        self._info = None
        self._last_orig: List[Tuple[int, int]] = []
        logger.info(f"HipDetrDetector initialized with model: {model_name}")
        logger.info(f"Using device: {self.device}")
        logger.info(f"Confidence threshold: {confidence_threshold}")

    # ------------------------------------------------------------------------------------------------------------
    def _setup_device(self, device: Optional[str] = None) -> str:
        if device is None:
            return "hip:0"
        d = str(device).lower()
        if d in ("cpu", "mps"):
            raise ValueError(f"HipDetrDetector runs on an AMD GPU only; device={device!r} is not supported")
        if d in ("hip", "cuda"):
            return "hip:0"
        if d.startswith(("hip:", "cuda:")):
            return "hip:" + d.split(":", 1)[1]
        raise ValueError(f"unknown device {device!r}")

    @property
    def device_ordinal(self) -> int:
        return int(self.device.split(":")[1])

    def _resolve_weights(self) -> str:
        cand = self.model_path or os.environ.get("OPD_DETR_WEIGHTS")
        if cand is None:
            raise FileNotFoundError(
                f"no local weights for {self.model_name!r}: pass model_path=<.safetensors> or set OPD_DETR_WEIGHTS "
                "(hub download by name is not available)")
        if os.path.isdir(cand):
            cand = os.path.join(cand, "model.safetensors")
        if not os.path.exists(cand):
            raise FileNotFoundError(f"weight file not found: {cand}")
        return cand

    def load_model(self) -> None:
        """Parse the checkpoint natively, fold FrozenBN, upload to the GPU (``opd_detr_create``)."""
        try:
            lib = _capi.load_library()
            path = self._resolve_weights()
            cfg = _capi.OpdConfig(struct_size=C.sizeof(_capi.OpdConfig), max_batch=self.max_batch,
                                  max_height=self.max_size[0], max_width=self.max_size[1],
                                  flags=(0 if self.use_graph else _capi.OPD_FLAG_NO_GRAPH) | (_capi.OPD_FLAG_BF16 if self.dtype == "bf16" else 0) |
                                        (_capi.OPD_FLAG_MULTI_STREAM if self.streams > 1 else 0))
            self._lib = lib
            handle = C.c_void_p()
            rc = lib.opd_detr_create(C.byref(cfg), path.encode("utf-8"), self.device_ordinal, C.byref(handle))
            _capi.check(rc, "opd_detr_create")
            self._handles.append(handle.value)
            for _ in range(1, self.streams):   # further handles share the first one's weights in HBM
                clone = C.c_void_p()
                _capi.check(lib.opd_detr_clone(handle, C.byref(clone)), "opd_detr_clone")
                self._handles.append(clone.value)
            info = _capi.OpdModelInfo()
            _capi.check(lib.opd_detr_info(C.c_void_p(self._handles[0]), C.byref(info)), "opd_detr_info")
            self.model, self._info = self._handles[0], info
            logger.info(f"Model loaded: {path}")
        except Exception as e:
            self.close()
            logger.error(f"Failed to load model: {e}")
            raise RuntimeError(f"Failed to load DETR model: {e}") from e

    def close(self) -> None:
        # communicator lanes bound to this detector's handles go first (sharding.NativeExchange registers itself here); the library also
        # detaches any lane that outlives its handle (opd_detr_destroy), so the order is a courtesy, not a requirement
        for ex in list(getattr(self, "_exchanges", [])):
            ex.close()
        self._exchanges = []
        if self._lib is not None:
            for h in self._handles:
                self._lib.opd_detr_destroy(C.c_void_p(h))
            for ptr, _ in self._staging.values():
                self._lib.opd_host_free(C.c_void_p(ptr))
        self._staging = {}
        self._handles = []
        self.model = None

    def __del__(self):  # pragma: no cover - best effort
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------------------------------------------------
    def _require_model(self) -> None:
        if self.model is None:
            raise RuntimeError("Model not loaded. Call load_model() first.")

    def _stage_array(self, slot: int, shape: Tuple[int, ...]) -> np.ndarray:
        """uint8 array of ``shape`` to stack a batch into: page-locked memory owned by handle ``slot`` (grown on demand,
        freed in ``close``; valid until the slot's next batch), or plain numpy memory without ``pinned_staging``."""
        n = int(np.prod(shape))
        if not self.pinned_staging or self._lib is None:
            return np.empty(shape, np.uint8)
        ptr, cap = self._staging.get(slot, (None, 0))
        if cap < n:
            if ptr:
                self._lib.opd_host_free(C.c_void_p(ptr))
                self._staging.pop(slot)
            p = C.c_void_p()
            _capi.check(self._lib.opd_host_alloc(n, C.byref(p)), "opd_host_alloc")
            ptr, cap = p.value, n
            self._staging[slot] = (ptr, cap)
        return np.ctypeslib.as_array((C.c_uint8 * n).from_address(ptr)).reshape(shape)

    def _stack(self, slot: int, frames: Sequence[np.ndarray]) -> np.ndarray:
        batch = self._stage_array(slot, (len(frames),) + tuple(frames[0].shape))
        for i, f in enumerate(frames):
            batch[i] = f
        return batch

    def _preprocess_batch(self, frames: Sequence[np.ndarray], slot: int = 0):
        """Host part of ``_preprocess_batch`` (deleted vit_detector.py 562-578): validate, resize to the model size,
        stack to one contiguous uint8 [B,H,W,3] BGR block.  BGR->RGB, 1/255 and mean/std run on the device.

        Frames whose model-input sizes differ form a RAGGED batch: like HF's ``DetrImageProcessor.pad``
        (``image_processing_detr.py:639-668``) every frame sits in the top-left corner of a canvas of the batch-maximum
        size, and the per-frame valid sizes are returned so that the device applies the padding-mask paths.

        Frames of ONE size that need resizing are NOT resized here: the batch stays at camera resolution and the
        returned ``target`` = (H, W) tells the caller to use the device-side resize (bit-exact with PIL's bilinear,
        ``opd_detr_*_resized``); ``device_resize=False`` or mixed sizes keep the host PIL path.
        Returns (batch, original sizes, valid sizes [B,2] int32 or None, target (H, W) or None)."""
        if len(frames) == 0:
            raise ValueError("empty frame batch")
        orig, out = [], []
        for f in frames:
            if not isinstance(f, np.ndarray) or f.ndim != 3 or f.shape[2] != 3 or f.dtype != np.uint8:
                raise ValueError("frames must be uint8 numpy arrays of shape (H, W, 3) in BGR order")
            orig.append((int(f.shape[0]), int(f.shape[1])))
        if self.resize and self.device_resize and len(set(orig)) == 1:
            th, tw = model_input_size(orig[0][0], orig[0][1], self.max_size[0], self.max_size[1])
            if (th, tw) != orig[0]:
                return self._stack(slot, frames), orig, None, (th, tw)
        for f in frames:
            if self.resize:
                th, tw = model_input_size(f.shape[0], f.shape[1], self.max_size[0], self.max_size[1])
                f = resize_frame(f, th, tw)
            out.append(f)
        shapes = {o.shape for o in out}
        if len(shapes) == 1:
            return self._stack(slot, out), orig, None, None
        H, W = max(o.shape[0] for o in out), max(o.shape[1] for o in out)
        if max(H, W) > max(self.max_size) or H * W > self.max_size[0] * self.max_size[1]:
            raise ValueError(f"ragged batch canvas {H}x{W} exceeds the configured maximum {self.max_size} (either orientation)")
        canvas = self._stage_array(slot, (len(out), H, W, 3))
        canvas[...] = 0
        for i, o in enumerate(out):
            canvas[i, :o.shape[0], :o.shape[1]] = o
        valid = np.asarray([[o.shape[0], o.shape[1]] for o in out], dtype=np.int32)
        return canvas, orig, valid, None

    def forward_raw(self, frames: Sequence[np.ndarray], want_encoder: bool = True):
        """Model outputs for a batch of BGR frames: (logits [B,Q,C+1], pred_boxes [B,Q,4], encoder [B,hw,256] | None)."""
        self._require_model()
        batch, orig, valid, target = self._preprocess_batch(frames)
        B, H, W, _ = batch.shape
        if target is not None:
            H, W = target
        Q, ncls, D = self._info.num_queries, self._info.num_classes_plus1, self._info.d_model
        fh, fw = _feature_hw(H), _feature_hw(W)
        logits = np.empty((B, Q, ncls), np.float32)
        boxes = np.empty((B, Q, 4), np.float32)
        enc = np.empty((B, fh * fw, D), np.float32) if want_encoder else None
        if target is not None:
            rc = self._lib.opd_detr_forward_resized(C.c_void_p(self.model), batch.ctypes.data_as(C.c_void_p), _capi.OPD_MEM_HOST,
                                                    B, batch.shape[1], batch.shape[2], H, W, logits.ctypes.data_as(C.c_void_p),
                                                    boxes.ctypes.data_as(C.c_void_p),
                                                    enc.ctypes.data_as(C.c_void_p) if enc is not None else None)
            _capi.check(rc, "opd_detr_forward_resized")
            self._last_orig = orig
            return logits, boxes, enc
        rc = self._lib.opd_detr_forward_ragged(C.c_void_p(self.model), batch.ctypes.data_as(C.c_void_p),
                                               _capi.OPD_PIXELS_U8_BGR_HWC, _capi.OPD_MEM_HOST, B, H, W,
                                               valid.ctypes.data_as(C.c_void_p) if valid is not None else None,
                                               logits.ctypes.data_as(C.c_void_p), boxes.ctypes.data_as(C.c_void_p),
                                               enc.ctypes.data_as(C.c_void_p) if enc is not None else None)
        _capi.check(rc, "opd_detr_forward")
        self._last_orig = orig
        return logits, boxes, enc

    def _detect_records(self, frames: Sequence[np.ndarray], handle: Optional[int] = None):
        model = self.model if handle is None else handle
        B, Q = len(frames), self._info.num_queries
        recs = (_capi.OpdDet * (B * Q))()
        counts = (C.c_int32 * B)()
        self._detect_into(frames, model, C.addressof(recs), C.addressof(counts), on_device=False)
        return recs, counts, Q

    def _frame_list_target(self, frames: Sequence[np.ndarray]) -> Optional[Tuple[int, int]]:
        """(H, W) of the model input when the chunk can go down as a LIST of frame pointers (``opd_detr_detect_frames``: every frame
        uploaded from where it lies, no stacked copy): contiguous uint8 [h, w, 3] frames of ONE size, model size reached on the device."""
        f0 = frames[0]
        for f in frames:
            if not isinstance(f, np.ndarray) or f.dtype != np.uint8 or f.ndim != 3 or f.shape != f0.shape or f.shape[2] != 3 or not f.flags.c_contiguous:
                return None
        h, w = int(f0.shape[0]), int(f0.shape[1])
        if not self.resize:
            return (h, w)
        th, tw = model_input_size(h, w, self.max_size[0], self.max_size[1])
        return (th, tw) if ((th, tw) == (h, w) or self.device_resize) else None

    def _detect_into(self, frames: Sequence[np.ndarray], model: int, rec_ptr: int, cnt_ptr: int, on_device: bool) -> None:
        """One ``max_batch`` chunk of host frames -> ``[B][Q]`` ``opd_det`` records at ``rec_ptr`` and ``[B]`` counts at
        ``cnt_ptr``; ``on_device``: both are HIP device pointers on this detector's GPU (``OPD_MEM_HOST_PIXELS_DEVICE_OUT``)."""
        if len(frames) == 0:
            raise ValueError("empty frame batch")
        target = self._frame_list_target(frames) if self.frame_lists else None
        if target is not None:
            kind = _capi.OPD_MEM_HOST_PIXELS_DEVICE_OUT if on_device else _capi.OPD_MEM_HOST
            recs, counts = C.cast(C.c_void_p(rec_ptr), C.POINTER(_capi.OpdDet)), C.cast(C.c_void_p(cnt_ptr), C.POINTER(C.c_int32))
            ptrs = (C.c_void_p * len(frames))(*[f.ctypes.data for f in frames])
            rc = self._lib.opd_detr_detect_frames(C.c_void_p(model), ptrs, kind, len(frames), int(frames[0].shape[0]), int(frames[0].shape[1]),
                                                  target[0], target[1], float(self.confidence_threshold), recs, counts)
            _capi.check(rc, "opd_detr_detect_frames")
            self._last_orig = [(int(f.shape[0]), int(f.shape[1])) for f in frames]
            return
        batch, orig, valid, target = self._preprocess_batch(frames, self._handles.index(model))
        B, H, W, _ = batch.shape
        kind = _capi.OPD_MEM_HOST_PIXELS_DEVICE_OUT if on_device else _capi.OPD_MEM_HOST
        recs, counts = C.cast(C.c_void_p(rec_ptr), C.POINTER(_capi.OpdDet)), C.cast(C.c_void_p(cnt_ptr), C.POINTER(C.c_int32))
        if target is not None:   # camera-resolution batch: resize on the device, boxes come back in camera pixels
            rc = self._lib.opd_detr_detect_resized(C.c_void_p(model), batch.ctypes.data_as(C.c_void_p), kind,
                                                   B, H, W, target[0], target[1], float(self.confidence_threshold), recs, counts)
            _capi.check(rc, "opd_detr_detect_resized")
        else:
            hw = np.asarray(orig, dtype=np.int32).reshape(B, 2)
            rc = self._lib.opd_detr_detect_ragged(C.c_void_p(model), batch.ctypes.data_as(C.c_void_p),
                                                  _capi.OPD_PIXELS_U8_BGR_HWC, kind, B, H, W,
                                                  valid.ctypes.data_as(C.c_void_p) if valid is not None else None,
                                                  float(self.confidence_threshold), hw.ctypes.data_as(C.c_void_p), recs, counts)
            _capi.check(rc, "opd_detr_detect")
        self._last_orig = orig

    @property
    def num_queries(self) -> int:
        self._require_model()
        return int(self._info.num_queries)

    def detect_records_into(self, frames: Sequence[np.ndarray], records, counts) -> None:
        """Frame-sharded callers (``sharding.ShardedDetector``): detect ``frames`` (at most ``max_batch``) and write their fixed-size
        records / counts into caller-owned int32 buffers ``records`` ``[len(frames), Q, 8]`` and ``counts`` ``[len(frames)]``.
        The buffers are anything with ``data_ptr()`` / ``is_cuda`` / ``is_contiguous()`` (torch tensors: the library itself never
        imports torch); device buffers are written by the post-process kernel directly, so the records can go into a collective
        without touching the host."""
        self._require_model()
        if len(frames) == 0:
            return
        if len(frames) > self.max_batch:
            raise ValueError(f"{len(frames)} frames exceed max_batch = {self.max_batch}")
        if not (records.is_contiguous() and counts.is_contiguous()):
            raise ValueError("records / counts must be contiguous")
        self._detect_into(frames, self.model, int(records.data_ptr()), int(counts.data_ptr()), on_device=bool(records.is_cuda))

    def detect_records_at(self, frames: Sequence[np.ndarray], rec_ptr: int, cnt_ptr: int) -> None:
        """The same with raw DEVICE pointers on this detector's GPU (``sharding.NativeExchange``: slots of the send buffer of an
        ``opd_comm`` bound to handle 0, ``opd_comm_buffers``)."""
        self._require_model()
        if len(frames) > self.max_batch:
            raise ValueError(f"{len(frames)} frames exceed max_batch = {self.max_batch}")
        if len(frames):
            self._detect_into(frames, self.model, int(rec_ptr), int(cnt_ptr), on_device=True)

    def _postprocess_batch(self, recs, counts, Q: int) -> List[List[Detection]]:
        """``_postprocess_batch`` (deleted vit_detector.py 591-647): person filter + NMS (C-ABI), xyxy -> xywh, foot point."""
        rc = self._lib.opd_person_nms_batch(recs, counts, len(counts), Q, PERSON_LABEL, float(self.nms_threshold))   # in place
        _capi.check(rc, "opd_person_nms_batch")
        results: List[List[Detection]] = []
        for b in range(len(counts)):
            dets = []
            for i in range(int(counts[b])):
                r = recs[b * Q + i]
                bbox = (float(r.x1), float(r.y1), float(r.x2 - r.x1), float(r.y2 - r.y1))
                dets.append(Detection(bbox=bbox, confidence=float(r.score), class_id=PERSON_LABEL, class_name="person",
                                      camera_coords=self._get_foot_position(bbox), query_index=int(r.query_index)))
            results.append(dets)
        return results

    def detect_batch(self, frames: List[np.ndarray]) -> List[List[Detection]]:
        """Batched detection (deleted vit_detector.py 508-550; ``.kiro/specs/office-person-detection/design.md:646-653``)."""
        self._require_model()
        if len(frames) == 0:
            return []
        try:
            chunks = [frames[i:i + self.max_batch] for i in range(0, len(frames), self.max_batch)]
            if len(self._handles) > 1 and len(chunks) > 1:
                return self._detect_chunks_overlapped(chunks)
            out: List[List[Detection]] = []
            for chunk in chunks:
                recs, counts, Q = self._detect_records(chunk)
                out.extend(self._postprocess_batch(recs, counts, Q))
            return out
        except Exception as e:
            logger.error(f"Detection failed: {e}")
            raise

    def _detect_chunks_overlapped(self, chunks: List[List[np.ndarray]]) -> List[List[Detection]]:
        """Chunk k runs on handle ``k % streams``; one worker thread per handle drives its chunks in order.

        The C-ABI calls release the GIL and every handle owns its stream, so host preprocessing, uploads and the
        tail of one batch overlap the trunk of the next.  Each chunk's result is what the
        serial loop returns for it: the handles hold identical weights and the kernels are batch-invariant.
        """
        n = len(self._handles)
        results: List[Optional[List[List[Detection]]]] = [None] * len(chunks)

        def worker(j: int) -> None:
            for k in range(j, len(chunks), n):
                recs, counts, Q = self._detect_records(chunks[k], self._handles[j])
                results[k] = self._postprocess_batch(recs, counts, Q)

        with ThreadPoolExecutor(max_workers=n) as pool:
            for fut in [pool.submit(worker, j) for j in range(min(n, len(chunks)))]:
                fut.result()
        return [dets for r in results for dets in r]

    def detect(self, frame: np.ndarray) -> List[Detection]:
        """Single-frame detection (``yolov8_detector.py:90-132``)."""
        self._require_model()
        try:
            recs, counts, Q = self._detect_records([frame])
            dets = self._postprocess_batch(recs, counts, Q)[0]
            logger.debug(f"Detected {len(dets)} persons")
            return dets
        except Exception as e:
            logger.error(f"Detection failed: {e}")
            raise

    def detect_with_features(self, frame: np.ndarray) -> Tuple[List[Detection], np.ndarray]:
        """Detection + (N, 256) appearance features pooled from the DETR encoder map; assigns ``det.features``
        (``yolov8_detector.py:134-159``; deleted vit_detector.py 148-171, 224-273)."""
        self._require_model()
        target = self._frame_list_target([frame]) if (self.frame_lists and isinstance(frame, np.ndarray)) else None
        if target is not None and self._info.d_model == 256:
            # one C-ABI call: the records and the pooled feature of every person record come back behind one host wait
            Q, D = self._info.num_queries, self._info.d_model
            recs, counts = (_capi.OpdDet * Q)(), (C.c_int32 * 1)()
            feats = np.empty((1, Q, D), np.float32)
            ptrs = (C.c_void_p * 1)(frame.ctypes.data)
            rc = self._lib.opd_detr_detect_frames_features(C.c_void_p(self.model), ptrs, 1, int(frame.shape[0]), int(frame.shape[1]), target[0], target[1],
                                                           float(self.confidence_threshold), PERSON_LABEL, recs, counts,
                                                           feats.ctypes.data_as(C.POINTER(C.c_float)))
            _capi.check(rc, "opd_detr_detect_frames_features")
            self._last_orig = [(int(frame.shape[0]), int(frame.shape[1]))]
            detections = self._postprocess_batch(recs, counts, Q)[0]
            features = feats[0, [d.query_index for d in detections]] if detections else np.array([])
            for i, det in enumerate(detections):
                det.features = features[i]
            return detections, features
        detections = self.detect(frame)
        features = self.extract_features(frame, detections)
        for i, det in enumerate(detections):
            if i < len(features):
                det.features = features[i]
        return detections, features

    def extract_features(self, frame: np.ndarray, detections: List[Detection]) -> np.ndarray:
        """ROI mean-pool + L2 norm on the encoder map of the LAST forward (frame 0), on the device
        (``src/tracking/feature_extractor.py:39-88``).  Returns ``np.array([])`` when there is nothing to pool
        (``yolov8_detector.py:171-172``)."""
        self._require_model()
        if len(detections) == 0:
            return np.array([])
        boxes = np.asarray([d.bbox for d in detections], dtype=np.float32).reshape(-1, 4)
        feats = np.empty((len(detections), self._info.d_model), np.float32)
        for s in range(0, len(detections), 128):
            chunk = np.ascontiguousarray(boxes[s:s + 128])
            rc = self._lib.opd_detr_roi_features(C.c_void_p(self.model), 0, chunk.ctypes.data_as(C.c_void_p), len(chunk),
                                                 int(frame.shape[0]), int(frame.shape[1]),
                                                 feats[s:s + 128].ctypes.data_as(C.c_void_p))
            _capi.check(rc, "opd_detr_roi_features")
        return feats

    def _get_foot_position(self, bbox: Tuple[float, float, float, float]) -> Tuple[float, float]:
        """Centre of the bottom edge (``yolov8_detector.py:229-241``)."""
        x, y, w, h = bbox
        return (x + w / 2, y + h)

    def get_attention_map(self, frame: np.ndarray, layer_index: int = -1) -> Optional[np.ndarray]:
        """Attention map for visualisation (deleted vit_detector.py 392-446; ``Visualizer.draw_attention_map``,
        ``src/visualization/visualizer.py:148-200``, takes a 1-D or 2-D array in [0, 1]).  The DETR-era source is gone, so the
        definition is this build's: detect on ``frame``, then the decoder's cross-attention weights of layer ``layer_index``,
        averaged over the heads and over the queries of the kept person detections (all queries when there is none), as a
        ``(feature_h, feature_w)`` float32 array scaled so that its maximum is 1.  Like the reference's detector this call is
        single-threaded: the map is read from the state the ``detect`` inside it leaves on handle 0, so no other call on this
        detector may run between the two (the library refuses a map that does not fit the buffer sized here)."""
        self._require_model()
        dets = self.detect(frame)
        fh, fw = (_feature_hw(s) for s in self._model_hw(frame))
        q = np.asarray(sorted({d.query_index for d in dets if d.query_index is not None}), dtype=np.int32)
        out = np.empty((fh * fw,), np.float32)
        rc = self._lib.opd_detr_attention_map(C.c_void_p(self.model), 0, int(layer_index), q.ctypes.data_as(C.c_void_p) if len(q) else None,
                                              len(q), out.ctypes.data_as(C.c_void_p), int(out.size))
        _capi.check(rc, "opd_detr_attention_map")
        mx = float(out.max())
        return (out / mx if mx > 0 else out).reshape(fh, fw)

    def _model_hw(self, frame: np.ndarray) -> Tuple[int, int]:
        """Model-input size of a frame under this detector's resize policy."""
        if self.resize:
            return model_input_size(frame.shape[0], frame.shape[1], self.max_size[0], self.max_size[1])
        return int(frame.shape[0]), int(frame.shape[1])

    # ------------------------------------------------------------------------------------------------------------
    def set_profiling(self, enabled: bool) -> None:
        self._require_model()
        _capi.check(self._lib.opd_detr_set_profiling(C.c_void_p(self.model), int(enabled)), "opd_detr_set_profiling")

    def stage_times_ms(self) -> List[float]:
        self._require_model()
        ms = (C.c_float * 8)()
        _capi.check(self._lib.opd_detr_stage_times(C.c_void_p(self.model), ms), "opd_detr_stage_times")
        return [float(v) for v in ms]


def _feature_hw(n: int) -> int:
    """Spatial size after the stem (s2), max-pool (s2) and three stride-2 stages: five times ⌊(n-1)/2⌋+1."""
    for _ in range(5):
        n = (n - 1) // 2 + 1
    return n
