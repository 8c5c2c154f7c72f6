"""voicepuppet_amd - MI355X-native PixReferNet step + audio front-end (hot path of taylorlu/voicepuppet).

Python is the host (device memory, streams, torch.distributed); every kernel on the measured path is
hand-written HIP for gfx950 behind the C ABI in include/vp_hip.h (libvp_hip.so).  There is no CPU or
eager-PyTorch fallback: importing the ops without the built library raises.
"""
__version__ = "0.1.0"
