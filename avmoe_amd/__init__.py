"""avmoe_amd -- MI355X-native AVMoE adapter hot path (router + cross-modal / unimodal adapter experts).

Host side: Python on PyTorch-ROCm, mirroring the reference's MoEAdapter / ExpertAdapter module API
(AVMOE/AVE/nets/net_trans_v3.py:296-487).  Compute: hand-written HIP for gfx950 behind the C ABI in
include/avmoe.h (avmoe_amd/lib/libavmoe_hip.so).  There is no CPU or eager-PyTorch fallback."""
__version__ = "0.1.0"
