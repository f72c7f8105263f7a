"""Builds pavenet_amd/lib/libpave_hip.so in-tree with hipcc for gfx950 (cross-compiles
without a GPU).  The .so is git-ignored but travels to the GPU box with the snapshot."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SOURCES = [os.path.join(_HERE, 'csrc', 'pave_kernels.hip'),
           os.path.join(_HERE, 'csrc', 'pave_gemm_split.hip'),
           os.path.join(_HERE, 'csrc', 'pave_enc_tile.hip')]
HEADERS = [os.path.join(_HERE, 'csrc', 'pave_internal.h'), os.path.join(ROOT, 'include', 'pave_hip.h')]
OUT = os.path.join(_HERE, 'lib', 'libpave_hip.so')


def build_native(force=False, verbose=False):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if not force and os.path.exists(OUT) and all(
            os.path.getmtime(OUT) >= os.path.getmtime(s) for s in SOURCES + HEADERS):
        return OUT
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-shared', '-fPIC',
           '-I' + os.path.join(ROOT, 'include'), '-o', OUT] + SOURCES
    if verbose:
        cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
    subprocess.check_call(cmd)
    return OUT


if __name__ == '__main__':
    print(build_native(force=True, verbose=True))
