"""MI355X-native PointNet++ SA/FP and DGCNN kNN/EdgeConv operators (drop-in for
UT-Team-Chun/Pointcloud-bridge's Highway_bridge/models/pointnet2_utils.py and DGCNN.py).

Python host code on PyTorch-ROCm; every operator calls hand-written gfx950 kernels through the
C ABI of include/pcb_hip.h (libpcb_hip.so, built by `python __graft_entry__.py build`).
There is no CPU or eager fallback: without the library or a GPU the operators raise.
"""
__version__ = "0.1.0"
