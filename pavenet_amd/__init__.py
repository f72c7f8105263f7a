"""MI355X-native PAVE-Net forward path."""
import os as _os

__version__ = '0.2.0'

# hipGraph replay: ROCm CLR's graph "AQL packet capture" (pre-built dispatch packets, on by default
# in ROCm 7) faults with a GPU memory-access error when a captured forward of >= 14 frames is
# replayed after a device-wide synchronize; with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 the same graph
# replays correctly (profiles/r02_graph_fault_probe.txt: one fresh process per runtime setting).
# The flag is read once, when the HIP runtime initialises, so it is set here -- on import, before
# the first HIP call -- unless the user has set it.
_pc = _os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')
if _pc is None:
    _os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'
    import torch as _torch
    GRAPH_REPLAY_SAFE = not _torch.cuda.is_initialized()   # too late if HIP is already up
else:
    GRAPH_REPLAY_SAFE = _pc == '0'
