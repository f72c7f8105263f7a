"""pavenet_amd -- MI355X-native PAVE-Net forward path (hand-written HIP kernels behind a
C ABI + the host-side mirror of the reference's operator / registry surface)."""
__version__ = '0.1.0'
