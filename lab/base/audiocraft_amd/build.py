"""Build libacmi.so (the gfx950 HIP kernel library) in-tree with hipcc.

    python -m audiocraft_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels with the tree.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
SOURCES = ['acmi_core.hip', 'acmi_gemm.hip', 'acmi_attn.hip', 'acmi_lm.hip', 'acmi_rvq.hip', 'acmi_conv.hip', 'acmi_chroma.hip', 'acmi_audio.hip']
HEADERS = [os.path.join(ROOT, 'include', 'acmi.h'), os.path.join(CSRC, 'acmi_common.h'),
           os.path.join(CSRC, 'acmi_lm_internal.h')]
OUT = os.path.join(CSRC, 'libacmi.so')


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


def needs_build() -> bool:
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return OUT
    cmd = [_hipcc(), '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
           '-Wno-pass-failed', '-ffp-contract=off', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-o', OUT + '.tmp']
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(OUT + '.tmp', OUT)
    return OUT


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
