"""Builds the native pieces in-tree (cross-compiles without a GPU); the .so files are git-ignored
but travel to the GPU box with the snapshot:

* ``pavenet_amd/lib/libpave_hip.so`` -- the C-ABI library (hipcc, gfx950);
* ``pavenet_amd/lib/libpave_hip_diag.so`` -- the same sources with ``-DPAVE_DIAG``: adds the
  kernel-form override and the timing-only ablation entry points (``pave_diag_*``) that tests/ and
  tools/ use through ``native.diag_build()``; the shipped library has neither;
* ``pavenet_amd/_ext.*.so`` -- the pybind module with mmcv._ext's ``ms_deform_attn_forward /
  _backward`` signatures on top of that C ABI (g++ against the installed torch headers).
"""
import os
import subprocess
import sysconfig

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
SOURCES = [os.path.join(_HERE, 'csrc', 'pave_kernels.hip'),
           os.path.join(_HERE, 'csrc', 'pave_gemm_split.hip'),
           os.path.join(_HERE, 'csrc', 'pave_enc_tile.hip'),
           os.path.join(_HERE, 'csrc', 'pave_gemm_dma.hip'),
           os.path.join(_HERE, 'csrc', 'pave_decoder.hip')]
HEADERS = sorted(os.path.join(_HERE, 'csrc', f) for f in os.listdir(os.path.join(_HERE, 'csrc')) if f.endswith('.h')) + \
          [os.path.join(ROOT, 'include', 'pave_hip.h')]
OUT = os.path.join(_HERE, 'lib', 'libpave_hip.so')
OUT_DIAG = os.path.join(_HERE, 'lib', 'libpave_hip_diag.so')
EXT_SOURCE = os.path.join(_HERE, 'csrc', 'pave_mmcv_ext.cpp')
EXT_OUT = os.path.join(_HERE, '_ext' + (sysconfig.get_config_var('EXT_SUFFIX') or '.so'))


def _fresh(out, deps):
    return os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(s) for s in deps)


def build_native(force=False, verbose=False, diag=True):
    """libpave_hip.so and (diag=True) libpave_hip_diag.so; one hipcc process per translation unit
    and flavour, all in parallel (the files are independent)."""
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    flavours = [('', [], OUT)] + ([('_diag', ['-DPAVE_DIAG=1'], OUT_DIAG)] if diag else [])
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    procs, links = [], []
    for suffix, defs, out in flavours:
        if not force and _fresh(out, SOURCES + HEADERS):
            continue
        objs = []
        for src in SOURCES:
            obj = os.path.join(_HERE, 'lib', os.path.basename(src).replace('.hip', suffix + '.o'))
            objs.append(obj)
            if not force and _fresh(obj, [src] + HEADERS):
                continue
            cmd = [hipcc, '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-c'] + defs + \
                  ['-I' + os.path.join(ROOT, 'include'), '-o', obj, src]
            if verbose and not suffix:
                cmd.insert(1, '-Rpass-analysis=kernel-resource-usage')
            procs.append((cmd, subprocess.Popen(cmd)))
        links.append((out, objs))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    for out, objs in links:
        subprocess.check_call([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out] + objs)
    return OUT


def build_ext(force=False):
    """The pybind module ``pavenet_amd._ext`` (mmcv._ext's two ms_deform_attn entry points,
    pybind.cpp:737-748) linked against libpave_hip.so."""
    if not force and _fresh(EXT_OUT, [EXT_SOURCE, OUT] + HEADERS):
        return EXT_OUT
    import torch
    from torch.utils.cpp_extension import include_paths, library_paths
    cmd = ['g++', '-O2', '-std=c++17', '-shared', '-fPIC', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           '-DTORCH_EXTENSION_NAME=_ext', '-DTORCH_API_INCLUDE_EXTENSION_H',
           f'-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}',
           '-I/opt/rocm/include', '-I' + os.path.join(ROOT, 'include'),
           '-I' + sysconfig.get_paths()['include']] + ['-I' + p for p in include_paths()] + \
          [EXT_SOURCE, '-o', EXT_OUT, '-L' + os.path.dirname(OUT), '-lpave_hip',
           '-Wl,-rpath,$ORIGIN/lib'] + \
          [a for p in library_paths() for a in ('-L' + p, '-Wl,-rpath,' + p)] + \
          ['-lc10', '-lc10_hip', '-ltorch_cpu', '-ltorch_hip', '-ltorch', '-ltorch_python']
    subprocess.check_call(cmd)
    return EXT_OUT


if __name__ == '__main__':
    print(build_native(force=True, verbose=True))
    print(build_ext(force=True))
