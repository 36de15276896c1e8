"""MI355X-native DETR person-detection forward pass (Phase-2 detector of office-person-detection-vit).

``HipDetrDetector`` keeps the reference's Python detector surface and calls through the C-ABI of
``include/opd_detr.h`` into hand-written HIP kernels (``csrc/``).  No CPU / PyTorch fallback exists.
"""

from .data_models import Detection  # noqa: F401
from .detector import HipDetrDetector, model_input_size  # noqa: F401
from .export import detections_to_coco, write_coco  # noqa: F401
from .feature_extractor import FeatureExtractor  # noqa: F401
from .similarity import SimilarityCalculator  # noqa: F401
from .tiling import TiledDetector  # noqa: F401

__all__ = ["Detection", "HipDetrDetector", "FeatureExtractor", "SimilarityCalculator", "TiledDetector", "detections_to_coco", "write_coco",
           "model_input_size"]
