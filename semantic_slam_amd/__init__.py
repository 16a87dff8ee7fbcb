"""semantic_slam_amd — MI355X-native hot path of hridaybavle/semantic_slam.

Host-side mirror (Python) of the reference's two hot call sites over the C-ABI of
``include/sslam.h`` (``libsslam_hip.so``, hand-written HIP for gfx950):

* :class:`GraphSLAM`              <- ``ps_graph_slam::GraphSLAM``           (reference include/ps_graph_slam/graph_slam.hpp:35-152)
* :class:`PointCloudSegmentation` <- ``point_cloud_segmentation``           (reference include/planar_segmentation/point_cloud_segmentation.h:8-184)

There is no CPU fallback: every compute entry point raises if the HIP library or a GPU is missing.
"""
from .graph_slam import GraphSLAM, GraphBatch, OptStats, SslamError  # noqa: F401
from ._lib import load_library, library_path, build_library  # noqa: F401

try:  # the frontend module is optional until its kernels are built
    from .segmentation import PointCloudSegmentation, SegParams  # noqa: F401
except ImportError:  # pragma: no cover
    pass
