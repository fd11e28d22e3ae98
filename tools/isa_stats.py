"""Per-kernel summary of a translation unit's gfx950 code: registers, scratch, and -- between the first and the last MFMA -- the
barriers, direct-to-LDS loads, waits (counted / vmcnt(0)) and scratch operations.
    python tools/isa_stats.py rotate-yolov3_amd/csrc/conv_pw.hip [kernel-name-substring]"""
import os
import re
import subprocess
import sys
import tempfile

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps"]


def compile_asm(src, workdir, extra=()):
    stem = os.path.basename(src)[:-4]
    subprocess.run(["hipcc"] + FLAGS + list(extra) + ["-c", os.path.abspath(src), "-o", os.path.join(workdir, stem + ".o")], check=True,
                   cwd=workdir, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(os.path.join(workdir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read()


def kernels(asm):
    """yields (mangled name, body lines, metadata text behind the body)"""
    parts = re.split(r"\n(_Z\S+):[^\n]*\n", asm)
    for i in range(1, len(parts) - 1, 2):
        chunk = parts[i + 1]
        if ".Lfunc_end" not in chunk or ".amdhsa_kernel" not in chunk:
            continue
        body, tail = chunk.split(".Lfunc_end", 1)
        yield parts[i], body.splitlines(), tail


def demangle_ints(name):
    return ",".join(re.findall(r"Li(\d+)E", name)) or "".join("1" if c == "1" else "0" for c in re.findall(r"Lb([01])E", name))


def stats(name, lines, tail):
    def meta(key):
        m = re.search(r"; %s: (\d+)" % key, tail)
        return int(m.group(1)) if m else -1
    mf = [k for k, l in enumerate(lines) if "v_mfma" in l]
    span = lines[mf[0]:mf[-1] + 1] if mf else []
    text = "\n".join(span)
    return {
        "vgpr": meta("NumVgprs"), "agpr": meta("NumAgprs"), "scratch": meta("ScratchSize"), "occupancy": meta("Occupancy"),
        "mfma": len(mf), "barriers": len([l for l in span if re.match(r"\s*s_barrier", l)]),
        "lds_dma": len([l for l in span if "buffer_load_dwordx4" in l and " lds" in l]),
        "vmcnt0": len([l for l in span if re.match(r"\s*s_waitcnt.*vmcnt\(0\)", l)]),
        "counted": sorted(set(int(x) for x in re.findall(r"s_waitcnt vmcnt\((\d+)\)", text)) - {0}),
        "scratch_ops": len([l for l in span if re.match(r"\s*scratch_", l)]),
    }


def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    d = tempfile.mkdtemp(prefix="isa_")
    asm = compile_asm(src, d)
    for name, lines, tail in kernels(asm):
        if filt not in name:
            continue
        s = stats(name, lines, tail)
        short = re.sub(r"^_ZN\d+_GLOBAL__N_1\d+", "", name)
        short = re.match(r"[A-Za-z_0-9]+?(?=I[LN]|E|$)", short).group(0) if re.match(r"[A-Za-z_0-9]+?(?=I[LN]|E|$)", short) else short[:40]
        print("%-28s <%s>  vgpr %3d agpr %3d scratch %4d occ %d | mfma %4d bar %3d dma %3d vmcnt0 %2d scratch-in-span %2d waits %s" % (
            short[:28], demangle_ints(name), s["vgpr"], s["agpr"], s["scratch"], s["occupancy"], s["mfma"], s["barriers"], s["lds_dma"],
            s["vmcnt0"], s["scratch_ops"], s["counted"]))
    subprocess.run(["rm", "-rf", d])


if __name__ == "__main__":
    main()
