"""MI355X-native bundle adjustment behind the reference's VO::Optimizer::optimize interface.

Sub-modules: hipba (ctypes binding of the C ABI), vo (Map/Frame/Feature/MapPoint/Optimizer mirror),
harness (ROS-free ba_test), scenes (synthetic workloads), sharded (multi-GPU driver).
"""
__all__ = ["hipba", "vo", "harness", "scenes"]
