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
SOURCES = ['acmi_core.hip', 'acmi_gemm.hip', 'acmi_gemm_f32.hip', 'acmi_engine.hip', 'acmi_attn.hip', 'acmi_crossfold.hip', 'acmi_lm.hip', 'acmi_prefill.hip', 'acmi_rvq.hip', 'acmi_conv.hip', 'acmi_chroma.hip', 'acmi_audio.hip', 'acmi_diffusion.hip']
HEADERS = [os.path.join(ROOT, 'include', 'acmi.h'), os.path.join(CSRC, 'acmi_common.h'),
           os.path.join(CSRC, 'acmi_lm_internal.h')]
OUT = os.path.join(CSRC, 'libacmi.so')


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC=...)")


FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-pass-failed', '-ffp-contract=off']
# Kernarg preload (gfx950): the leading scalar / pointer arguments of a kernel arrive in SGPRs at wave start (up to 14
# dwords) instead of through scalar loads of a kernarg block that is cold on every launch of the decode chain; the
# kernels of these files are written for it (acmi_gemm.hip: TlHot; acmi_attn.hip: attn_decode_kernel).  Firmware without the feature runs the compiler's
# compatibility prologue, which loads the same registers.
_PRELOAD = ['-mllvm', '-amdgpu-kernarg-preload-count=14']
FILE_FLAGS = {'acmi_gemm.hip': _PRELOAD, 'acmi_gemm_f32.hip': _PRELOAD, 'acmi_attn.hip': _PRELOAD}
OBJDIR = os.path.join(CSRC, 'build')
FILE_DEPS = {'acmi_gemm_f32.hip': ['acmi_gemm.hip']}   # sources a translation unit #includes


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build() -> bool:
    return _stale(OUT, [os.path.join(CSRC, s) for s in SOURCES] + HEADERS)


def build(force: bool = False, verbose: bool = True, out: str = OUT, defines=()) -> str:
    """One object per translation unit (compiled in parallel, rebuilt only when the source or a header changed), then one
    link.  `out` / `defines`: A/B library variants for same-box timing (ACMI_LIB selects the .so at import)."""
    if not force and out == OUT and not needs_build():
        return OUT
    from concurrent.futures import ThreadPoolExecutor
    tag = '' if not defines else '_' + '_'.join(d.replace('=', '-') for d in defines)
    os.makedirs(OBJDIR, exist_ok=True)
    inc = ['-I' + os.path.join(ROOT, 'include'), '-I' + CSRC] + ['-D' + d for d in defines]

    def compile_one(src):
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + tag + '.o')
        path = os.path.join(CSRC, src)
        if force or _stale(obj, [path] + [os.path.join(CSRC, d) for d in FILE_DEPS.get(src, [])] + HEADERS):
            cmd = [_hipcc()] + FLAGS + FILE_FLAGS.get(src, []) + inc + ['-c', path, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=min(len(SOURCES), os.cpu_count() or 4)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out + '.tmp'] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    os.replace(out + '.tmp', out)
    return out


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
