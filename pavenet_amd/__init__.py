"""MI355X-native PAVE-Net forward path."""
import os as _os
import sys as _sys

__version__ = '0.3.0'

# hipGraph replay (opt-in, pavenet_amd/graph.py): ROCm CLR's graph "AQL packet capture" (pre-built
# dispatch packets, on by default in ROCm 7) faults with a GPU memory-access error when a captured
# forward of >= 14 frames is replayed after a device-wide synchronize; with
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 the same graph replays correctly
# (profiles/r02_graph_fault_probe.txt: one fresh process per runtime setting).  The runtime reads
# the flag once, when it initialises.  Policy:
#   * the user exported it: respected, nothing is changed ("0" = safe);
#   * unset, and `torch` has not been imported yet (so nothing of this process can have started HIP
#     through it): the flag is set to "0" here -- the ONE import-time change to the process
#     environment this package makes (README.md, "hipGraph replay");
#   * unset, torch already imported: the environment is left alone and large captures are refused
#     (the HIP runtime may already be up with the default, e.g. after torch.cuda.device_count()).
_pc = _os.environ.get('DEBUG_CLR_GRAPH_PACKET_CAPTURE')
if _pc is not None:
    GRAPH_REPLAY_SAFE = _pc == '0'
elif 'torch' not in _sys.modules:
    _os.environ['DEBUG_CLR_GRAPH_PACKET_CAPTURE'] = '0'
    GRAPH_REPLAY_SAFE = True
else:
    GRAPH_REPLAY_SAFE = False
