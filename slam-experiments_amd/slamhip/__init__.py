"""slamhip — MI355X (gfx950) kernels for the descriptor-matching / reprojection
hot path of ViV99/slam-experiments, behind a ctypes C ABI (include/slamhip.h)."""
from ._lib import NO_MATCH_DIST, NO_MATCH_IDX, SlamHipBusy, SlamHipError, device_count, load  # noqa: F401
from .device import Context, DeviceBuffer, default_context, plan_describe  # noqa: F401
from .matching import (  # noqa: F401
    NORM_HAMMING,
    DeviceDescriptors,
    KeyframeDatabase,
    ResidentMatcher,
    Top2Table,
    as_descriptors,
    cross_check_arrays,
    knn2_device,
    knn2_device_batch,
    knn2_select_device,
    knn_match_arrays,
    knn_match_arrays_batch,
    knn_match_collection,
    match_arrays,
    ratio_test_arrays,
    split_image_index,
)
from .reproj import PoseOnlyProblem, ReprojProblem, build_linearization, poses_to_rt12  # noqa: F401
