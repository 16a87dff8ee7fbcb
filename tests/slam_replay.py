"""Shared by the orchestrator tests: drive the product (sslam_slam_* through semantic_slam_amd.SemanticGraphSLAM) and the oracle
(oracle/np_slam.py) through the same synthetic run."""
import numpy as np

from oracle import np_slam as S

# the values the reference ships (config/bucket_detector.yaml:26-27), not the constructor defaults
ODOM_STDDEV_X, ODOM_STDDEV_Q = 0.00667, 0.00001


def oracle_instance(**kw):
    kw.setdefault("const_stddev_x", ODOM_STDDEV_X)
    kw.setdefault("const_stddev_q", ODOM_STDDEV_Q)
    return S.SemanticGraphSlam(**kw)


def product_instance(**kw):
    from semantic_slam_amd.semantic_graph_slam import SemanticGraphSLAM, default_slam_params
    p = default_slam_params()
    p.const_stddev_x, p.const_stddev_q = ODOM_STDDEV_X, ODOM_STDDEV_Q
    for k, v in kw.items():
        setattr(p, k, v)
    return SemanticGraphSLAM(p)


def product_instance_with(segmentation, **kw):
    """the product with a frontend handle (the cloud path of the tick)"""
    from semantic_slam_amd.semantic_graph_slam import SemanticGraphSLAM, default_slam_params
    p = default_slam_params()
    p.const_stddev_x, p.const_stddev_q = ODOM_STDDEV_X, ODOM_STDDEV_Q
    for k, v in kw.items():
        setattr(p, k, v)
    return SemanticGraphSLAM(p, segmentation)


def planes_of(objs):
    from semantic_slam_amd.segmentation import Plane
    out = []
    for o in objs:
        p = Plane()
        for k in range(3):
            p.centroid_cam[k] = float(o["pose"][k])
        for k in range(4):
            p.normal_d[k] = float(o["normal"][k])
        p.class_id, p.plane_type = int(o["class_id"]), int(o["plane_type"])
        out.append(p)
    return out


def feed(system, ev, product):
    """one odometry sample through the callbacks; returns (became keyframe, tick ran)"""
    if ev.objects is not None:
        if product:
            system.setSegmentedObjects(planes_of(ev.objects))
        else:
            system.set_segmented_objects(ev.objects)
    kf = system.VIOCallback(ev.stamp, ev.odom) if product else system.vio(ev.stamp[0], ev.stamp[1], ev.odom)
    ran = system.run() if ev.run_after else False
    return bool(kf), bool(ran)
