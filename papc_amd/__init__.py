"""papc_amd -- the MI355X (gfx950) hot path of AgentMaker/PAPC behind the reference's op signatures.

Public surface mirrors the reference:
  papc_amd.functional  <- PAPC/models/layers/pointnet2_basic_layers.py free functions
  papc_amd.layers      <- PointNetSetAbstraction(.Msg)
  papc_amd.pillars     <- PFNLayer / PillarFeatureNet (pointpillars/models/bones/pillars.py)
  papc_amd.models      <- PointNet2_SSG_Clas / PointNet2_MSG_Clas / PointNet_Basic_Clas
The compute lives in libpapc_hip.so (hand-written HIP, C ABI in include/papc_hip.h).
"""
__version__ = "0.1.0"
