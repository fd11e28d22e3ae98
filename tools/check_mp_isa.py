"""Static check of conv_mp.hip's gfx950 code: the multi-phase K loop relies on COUNTED s_waitcnt vmcnt(N); a register spill
inside it (scratch_load / scratch_store are vector-memory operations, they enter the same in-order queue) would silently turn
those counts into races.  Compiles the unit with -save-temps and fails if any conv_mp_kernel instantiation has a scratch
instruction between the first and the last s_barrier of its innermost loop.  python tools/check_mp_isa.py [--keep]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "rotate-yolov3_amd", "csrc", "conv_mp.hip")


def kernels(asm):
    cur, name = None, None
    for line in asm.splitlines():
        m = re.match(r"^(_ZN\S*conv_mp_kernel\S*):", line)
        if m:
            name, cur = m.group(1), []
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                yield name, cur
                cur = None
            else:
                cur.append(line)


def check(lines):
    """returns (n_mfma_in_loop, n_scratch_in_loop): loop = the span between the first and last s_barrier that are followed /
    preceded by v_mfma within the innermost loop (Depth=2)."""
    # innermost loop: from the 'Inner Loop Header: Depth=2' label to the last backward branch to it; layout may rotate the
    # loop, so take the whole span that contains all s_barrier instructions adjacent to MFMA clusters
    idx_bar = [i for i, l in enumerate(lines) if re.match(r"\s*s_barrier", l)]
    idx_mfma = [i for i, l in enumerate(lines) if "v_mfma_" in l]
    if not idx_mfma:
        return 0, 0
    lo, hi = idx_mfma[0], idx_mfma[-1]
    bars = [i for i in idx_bar if lo - 400 < i < hi + 400]
    lo, hi = min(bars + [lo]), max(bars + [hi])
    scratch = [l.strip() for l in lines[lo:hi + 1] if re.match(r"\s*scratch_", l)]
    return len([i for i in idx_mfma if lo <= i <= hi]), scratch


def main():
    keep = "--keep" in sys.argv
    d = tempfile.mkdtemp(prefix="mp_isa_")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
           "-c", SRC, "-o", os.path.join(d, "conv_mp.o")]
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    asm = open(os.path.join(d, "conv_mp-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
    bad = 0
    n = 0
    for name, lines in kernels(asm):
        n += 1
        nm, scratch = check(lines)
        short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+conv_mp_kernelI", "conv_mp_kernel<", name).split("EEEv")[0]
        print("%-40s mfma in loop span %4d  scratch ops in loop span %d" % (short, nm, len(scratch)))
        if scratch:
            bad += 1
            for s in scratch[:4]:
                print("    ", s)
    if not keep:
        subprocess.run(["rm", "-rf", d])
    if n == 0:
        print("no conv_mp_kernel found")
        return 1
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
