"""Experiment helper: builds variants of libpave_hip.so that differ only in -D switches of ONE GEMM
translation unit (pavenet_amd/lib/variants/libpave_hip_<name>.so; the other translation units are
taken from the regular build).  Flags containing PAVE_Q_ rebuild pave_gemm_dma.hip (the LDS-DMA
generation), any other flag pave_gemm_split.hip.  The timing-only ablations quoted in DESIGN.md 4.2
were temporary `#if defined(PAVE_Q_ABL)` edits of the kernel body built this way and timed with
tools/gemm_ab.py; they are not kept in the source.
    python tools/build_variants.py name:-DX=1,-DY=2 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pavenet_amd.build_native import build_native  # noqa: E402

LIB = os.path.join(ROOT, 'pavenet_amd', 'lib')


def main():
    build_native()
    os.makedirs(os.path.join(LIB, 'variants'), exist_ok=True)
    procs = []
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition(':')
        flags = [f for f in flags.split(',') if f]
        obj = os.path.join(LIB, 'variants', f'gemm_{name}.o')
        src = 'pave_gemm_split.hip'
        other = 'pave_gemm_dma_diag.o'
        if any('PAVE_Q_' in f for f in flags):   # switches of the DMA generation
            src, other = 'pave_gemm_dma.hip', 'pave_gemm_split_diag.o'
        cmd = ['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-fPIC', '-c',
               '-DPAVE_DIAG=1', '-I' + os.path.join(ROOT, 'include'), '-Rpass-analysis=kernel-resource-usage'] + flags + \
              ['-o', obj, os.path.join(ROOT, 'pavenet_amd', 'csrc', src)]
        procs.append((name, (obj, other), subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for name, (obj, other), p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            print(err[-3000:])
            raise SystemExit(f'variant {name} failed to build')
        # register / scratch use of the main kernel forms
        lines = err.splitlines()
        for i, ln in enumerate(lines):
            if 'Function Name' in ln and ('kernel_occ2ILi2ELi2ELb0ELi3ELb0ELi0E' in ln or
                                          'kernel_w8ILi2ELi2ELb0ELi3ELi0ELb1E' in ln or
                                          'kernel_occ2ILi2ELi2ELb0ELi3ELb0ELi1E' in ln or 'gemm_q_kernelILi4ELi0ELb0E' in ln):
                kn = 'q rows' if 'gemm_q' in ln else ('occ2 rows' if 'Li0EEEv' in ln and 'occ2' in ln else ('w8 ln' if 'w8' in ln else 'occ2 conv3x3'))
                info = [l.split(':')[-2].strip() + ':' + l.split(':')[-1].split('[')[0].strip()
                        for l in lines[i + 1:i + 9] if 'VGPRs' in l or 'Scratch' in l]
                print(f'{name:12s} {kn:14s} {" ".join(info)}')
        out = os.path.join(LIB, 'variants', f'libpave_hip_{name}.so')
        subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out,
                               obj, os.path.join(LIB, other), os.path.join(LIB, 'pave_kernels_diag.o'),
                               os.path.join(LIB, 'pave_enc_tile_diag.o')])
        print('built', out)


if __name__ == '__main__':
    main()
