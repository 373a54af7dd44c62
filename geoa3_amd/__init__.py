"""geoa3_amd -- MI355X (gfx950) implementation of the GeoA3 inner attack loop.

    geoa3_amd.attack      attack(), attack_sharded(), AttackRunner        (Attacker/geoA3_attack.py)
    geoa3_amd.loss_utils  chamfer_loss, hausdorff_loss, curvature_loss... (Lib/loss_utils.py)
    geoa3_amd.ops         knn_points, knn_gather, nn1_pair, knn_planar... (pytorch3d.ops as used by the reference)
    geoa3_amd.pointnet    PointNet                                        (Model/PointNet.py)
    geoa3_amd.pointnet2   ext, PointNet2ClassificationSSG ...             (Model/pointnet2_ops_lib, Model/PointNetPP_ssg.py)
    geoa3_amd.data        ModelNet40 (.mat loader), synthetic generators  (Provider/modelnet10_instance250.py)
    geoa3_amd.distributed instance sharding over torch.distributed (RCCL)

All arithmetic runs in libgeoa3_hip.so (C ABI: include/geoa3_hip.h); there is no CPU fallback.
Submodules import torch lazily-enough that `import geoa3_amd` works without a GPU; using them needs one.
"""
__version__ = "0.1.0"
