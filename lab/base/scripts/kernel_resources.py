"""Per-kernel register / scratch / LDS report of the gfx950 code objects (no GPU needed).

    python scripts/kernel_resources.py [--fail-on-scratch]

Compiles every .hip file of audiocraft_amd/csrc to assembly (hipcc -S --cuda-device-only) and prints, for each
kernel, the VGPR / SGPR counts, the private segment (scratch) size and whether SGPRs were spilled to VGPR
lanes.  Kernels on the decode path must not use scratch: a kernel with a private segment starts its waves
more slowly, which showed up as +3 us per launch on the skinny GEMM (kernel notes).
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'audiocraft_amd', 'csrc')


def report(path: str):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, 'k.s')
        cmd = ['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-ffp-contract=off',
               '-Wno-pass-failed', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC, '-S', '--cuda-device-only',
               '-o', out, path]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read()
    rows = {}
    for m in re.finditer(r'^\s*\.set (\S+?)\.(num_vgpr|numbered_sgpr|private_seg_size), (\d+)', text, re.M):
        rows.setdefault(m.group(1), {})[m.group(2)] = int(m.group(3))
    for m in re.finditer(r'^\s*\.amdhsa_group_segment_fixed_size (\d+)', text, re.M):
        pass
    # SGPR spills show up as v_writelane in the kernel body
    bodies = re.split(r'^(\S+):\s*; @\S+$', text, flags=re.M)
    spills = {}
    for name, body in zip(bodies[1::2], bodies[2::2]):
        spills[name] = body.count('v_writelane_b32')
    return rows, spills


def demangle(names):
    try:
        out = subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt'], input='\n'.join(names), text=True,
                             capture_output=True, check=True).stdout.split('\n')
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--fail-on-scratch', action='store_true')
    args = ap.parse_args()
    bad = 0
    for path in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        rows, spills = report(path)
        names = demangle(list(rows))
        print(f'== {os.path.basename(path)}')
        for k, r in rows.items():
            short = re.sub(r'\(.*', '', names[k]).replace('void ', '')
            scratch = r.get('private_seg_size', 0)
            bad += scratch > 0
            print(f"  {short:60s} vgpr {r.get('num_vgpr', -1):3d}  sgpr {r.get('numbered_sgpr', -1):3d}  "
                  f"scratch {scratch:4d} B  sgpr-spill-writes {spills.get(k, 0)}")
    if args.fail_on_scratch and bad:
        sys.exit(f'{bad} kernel(s) use scratch')


if __name__ == '__main__':
    main()
