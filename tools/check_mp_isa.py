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
    bad += check_conv_mq(d)
    bad += check_conv_pw(d)
    bad += check_wgrad_wide(d)
    bad += check_wgrad_reduce_batch(d)
    bad += check_conv0_bwd(d)
    bad += check_conv_stem(d)
    bad += check_store_data_hazard(d)
    if not keep:
        subprocess.run(["rm", "-rf", d])
    if n == 0:
        print("no conv_mp_kernel found")
        return 1
    return 1 if bad else 0


def check_conv_mq(d):
    """conv_mq.hip: the K loop between the first and the last MFMA of every instantiation of the family <GEN, VAR, CW, PF, BNRED> holds no
    scratch operation (spills share the in-order VMEM queue of the counted waits), only counted waits (phase 0: vmcnt(PF/4) /
    vmcnt(PF/4 + stores), phase 3: vmcnt(CW/16): never vmcnt(0)), 2 * (PF/4 + CW/16) direct-to-LDS loads, 8 * (CW/32) * (PF/2) MFMAs and
    ONE s_barrier per K tile; the counted waits carry exactly those values."""
    src = os.path.join(ROOT, "rotate-yolov3_amd", "csrc", "conv_mq.hip")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
           "-c", src, "-o", os.path.join(d, "conv_mq.o")]
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "conv_mq-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    bad, found, i = 0, 0, 0
    seen = set()
    while i < len(lines):
        m = re.match(r"^(_ZN\S*conv_mq_kernel\S*):", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        t = re.search(r"conv_mq_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELb([01])ELi(\d+)ELi(\d+)E", m.group(1))
        gen, var, cw, pf, bnred, fs, ko = [int(v) for v in t.groups()]
        seen.add((gen, cw, pf, bnred))
        if fs == 2:
            seen.add((gen, cw, pf, bnred, 'sweep2'))
        ppc, wph, per_kt = pf // 4, cw // 16, 8 * (cw // 32) * (pf // 2)
        nst = 0 if gen == 2 else (cw // 32) * pf
        idx = [k for k, l in enumerate(body) if "v_mfma_" in l]
        # the loop as laid out is rotated: phase 3's MFMAs open the block, its wait / barrier / chunk requests close it behind
        # phase 2's MFMAs -- take the tail up to the loop's backward branch
        hi = idx[-1]
        while hi + 1 < len(body) and hi < idx[-1] + 80 and not re.match(r"\s*s_c?branch", body[hi]):
            hi += 1
        # ... or, when the tail holds forward branches of its own (the channel-major K order's wrap tests), up to the innermost backward
        # branch that closes a loop around every MFMA
        labels = {mm.group(1): k for k, l in enumerate(body) for mm in [re.match(r"^(\.LBB\S+):", l)] if mm}
        back = [(k, labels[mm.group(1)]) for k, l in enumerate(body) for mm in [re.match(r"\s*s_c?branch\S*\s+(\.LBB\S+)", l)]
                if mm and mm.group(1) in labels and labels[mm.group(1)] <= idx[0] and k >= idx[-1]]
        if back:
            hi = max(hi, min(back)[0])
        # the loop may also be laid out un-rotated (phase 0's wait and chunk requests in front of the first MFMA): start at the inner loop's header
        lo = idx[0]
        heads = [k for k, l in enumerate(body[:idx[0]]) if "This Inner Loop Header" in l]
        if heads and idx[0] - heads[-1] < 2500 and not any("v_mfma_" in l for l in body[heads[-1]:idx[0]]):
            rotated = any(re.match(r"\s*s_waitcnt vmcnt\(", l) or (" lds" in l and "buffer_load" in l) for l in body[idx[-1]:hi + 1])
            if not rotated:
                lo = heads[-1]
        span = body[lo:hi + 1]
        nm = len(idx)
        kt = nm / float(per_kt)                           # K-tile bodies the compiler laid out
        scratch = [l.strip() for l in span if re.match(r"\s*scratch_", l)]
        full = [l.strip() for l in span if re.match(r"\s*s_waitcnt.*vmcnt\(0\)", l)]
        counted = [int(re.match(r"\s*s_waitcnt vmcnt\((\d+)\)", l).group(1)) for l in span if re.match(r"\s*s_waitcnt vmcnt\(([1-9]\d*)\)", l)]
        allowed = {ppc, wph, ppc + nst} if (nst and var == 0) else {ppc, wph}
        odd_waits = [c for c in counted if c not in allowed]
        bars = len([l for l in span if re.match(r"\s*s_barrier", l)])
        dma = len([l for l in span if "buffer_load_dwordx4" in l and " lds" in l])
        short = "conv_mq_kernel<gen %d, var %d, %d ch/wave, %d px frags%s%s>" % (gen, var, cw, pf, ", bnred" if bnred else "", (", sweep 2" if fs == 2 else "") + (", channel-major K" if ko else ""))
        print("%-62s mfma in loop span %4d  lds-dma %3d  counted waits %2d %s  vmcnt(0) %d  s_barrier %d  scratch %d" % (
            short, nm, dma, len(counted), sorted(set(counted)), len(full), bars, len(scratch)))
        found += 1
        if (scratch or full or nm % per_kt or len(counted) < 2 * kt or bars != kt or dma != 2 * (ppc + wph) * kt or
                (var == 0 and odd_waits)):
            bad += 1
        i = j
    if not found:
        print("no conv_mq_kernel found")
        return 1
    # the shipped family: the 256-channel tile (three epilogue kinds, both store orders).  The 128-channel tiles of 128 / 64 pixels (+ folded
    # reduce) are instantiated in the measurement build only since round 6 (the same loop checks apply to them when this script is pointed at
    # that build: every conv_mq_kernel found is checked)
    want = {(g, 64, 8, 0) for g in (0, 1, 2)} | {(0, 64, 8, 0, 'sweep2'), (1, 64, 8, 0, 'sweep2'), (2, 64, 8, 0, 'sweep2')}
    if not want <= seen:
        print("conv_mq: missing instantiations", sorted(want - seen))
        bad += 1
    return bad


def _pw_first_block_younger(kt, KT, RING, LPU, WL, NREQ, D):
    """mirror of conv_pw.hip pw_first_block_younger()"""
    after, seen = 0, False
    for u in range(RING - 1):
        if seen:
            after += LPU
        if u == kt:
            seen = True
    for j in range(min(D, KT)):
        if seen:
            after += WL
    for t in range(kt):
        if t == 0 and seen:
            after += NREQ
        if seen:
            after += LPU
        if t + RING - 1 == kt:
            seen = True
        if t + D < KT and seen:
            after += WL
    return after


def check_conv_pw(d):
    """conv_pw.hip: the counted waits are arithmetic on the ISSUE ORDER of the wave's vector-memory operations (fills = direct-to-LDS
    loads, filter loads, row-block requests, stores).  The compiler once hoisted filter loads in front of the prologue's fills and
    the wait for unit 0 allowed three operations too many in flight (a flaky wrong tile).  For every instantiation this checks, in
    the generated code: the prologue issues (RING-1) x LPU fills and THEN WD x WL filter loads; iteration kt of the first row block
    issues [NREQ requests (kt == 0)] [LPU fills] [WL filter loads while kt + WD < KT] in that order; the wait in front of its barrier
    is the number the model computes (clamped to 63); no vmcnt(0) and no scratch operation sits between the first and the last MFMA
    (MODE 4 excepted: its decode stage drains once per row block by design)."""
    src = os.path.join(ROOT, "rotate-yolov3_amd", "csrc", "conv_pw.hip")
    sfile = os.path.join(d, "conv_pw-hip-amdgcn-amd-amdhsa-gfx950.s")
    if not os.path.exists(sfile):
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
               "-c", src, "-o", os.path.join(d, "conv_pw.o")]
        subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(sfile).read().splitlines()
    bad, found, i = 0, 0, 0
    while i < len(lines):
        m = re.match(r"^(_ZN\S*conv_pw_kernelILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\S*):", lines[i])
        if not m:
            i += 1
            continue
        KT, NW, CF, PF, RING, MODE = [int(v) for v in m.groups()[1:]]
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        i = j
        found += 1
        LPU, WL, WD = (PF * 16 // 8) // NW, 2 * CF, min(3, KT)
        nstg = 0 if MODE == 4 else (PF if CF == 2 else PF // 2)
        NREQ = (nstg if MODE in (2, 3) else 0) + (nstg if MODE == 3 else 0)
        toks = []               # (kind, value): F fill, L other buffer load, S store, w wait, B barrier, m mfma
        for l in body:
            t = l.split(";")[0].strip()
            if t.startswith("s_barrier"):
                toks.append(("B", 0))
            elif t.startswith("s_waitcnt") and "vmcnt" in t:
                toks.append(("w", int(re.search(r"vmcnt\((\d+)\)", t).group(1))))
            elif t.startswith("buffer_load_dwordx4") and " lds" in t:
                toks.append(("F", 0))
            elif re.match(r"(buffer|global)_load", t):
                toks.append(("L", 0))
            elif re.match(r"(buffer|global)_store", t):
                toks.append(("S", 0))
            elif t.startswith("scratch_"):
                toks.append(("X", 0))
            elif "v_mfma" in t:
                toks.append(("m", 0))
        first_m = next(k for k, t in enumerate(toks) if t[0] == "m")
        # prologue = everything in front of the first barrier that FOLLOWS the first fill
        first_f = next(k for k, t in enumerate(toks) if t[0] == "F")
        bars = [k for k, t in enumerate(toks) if t[0] == "B" and k > first_f]
        pro = [t[0] for t in toks[first_f:bars[0]] if t[0] in "FL"]
        want_pro = ["F"] * ((RING - 1) * LPU) + ["L"] * (WD * WL)
        errs = []
        if pro != want_pro:
            errs.append("prologue issues %s, expected %d fills then %d filter loads" % ("".join(pro), (RING - 1) * LPU, WD * WL))
        for kt in range(KT):
            seg_end = bars[kt + 1] if kt + 1 < len(bars) else len(toks)
            seg = [t[0] for t in toks[bars[kt]:seg_end] if t[0] in "FL"]
            if kt == KT - 1:     # the last iteration is followed by the epilogue (scale / shift loads may appear): only the head matters
                seg = seg[:(NREQ if kt == 0 else 0) + LPU]
            want = ["L"] * (NREQ if kt == 0 else 0) + ["F"] * LPU + (["L"] * WL if kt + WD < KT else [])
            if seg[:len(want)] != want or (kt < KT - 1 and len(seg) != len(want)):
                errs.append("first row block, iteration %d issues %s, expected %s" % (kt, "".join(seg), "".join(want)))
            waits = [t[1] for t in toks[(bars[kt - 1] if kt else first_f):bars[kt]] if t[0] == "w"]
            model = min(63, _pw_first_block_younger(kt, KT, RING, LPU, WL, NREQ, WD))
            if not waits or waits[-1] != model:
                errs.append("first row block, iteration %d waits vmcnt(%s) in front of its barrier, the model says %d" % (kt, waits[-1] if waits else None, model))
        last_m = max(k for k, t in enumerate(toks) if t[0] == "m")
        span = toks[first_m:last_m + 1]
        if MODE != 4 and any(t == ("w", 0) for t in span):
            errs.append("vmcnt(0) between the MFMAs")
        if any(t[0] == "X" for t in span if True) and MODE != 3:
            errs.append("scratch operation between the MFMAs")
        print("conv_pw_kernel<%d,%d,%d,%d,%d,%d>  issue order and first-block waits %s" % (KT, NW, CF, PF, RING, MODE, "ok" if not errs else "BAD"))
        for e in errs:
            print("    ", e)
        bad += 1 if errs else 0
    if not found:
        print("no conv_pw_kernel found")
        return 1
    return bad


def check_conv0_bwd(d):
    """conv0_bwd.hip (layer 0's one-pass backward): properties of the generated code its speed depends on, each of which a
    first version lacked -- (1) the wave-uniform group offsets reach the buffer loads as SGPR operands: a divergent-looking wave index
    makes the compiler wrap EVERY load in a waterfall loop (v_readfirstlane + compare + branch; 0.66 -> 0.87 ms); (2) the request
    pipeline survives: no s_waitcnt vmcnt(0) between the first and the last MFMA (packed 2-byte loads, or `live &&` in every address,
    made the compiler wait for each load in turn); (3) no scratch; (4) the steady-state waits leave >= 16 loads in flight."""
    src = os.path.join(ROOT, "rotate-yolov3_amd", "csrc", "conv0_bwd.hip")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
           "-c", src, "-o", os.path.join(d, "conv0_bwd.o")]
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "conv0_bwd-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    bad, found, i = 0, 0, 0
    while i < len(lines):
        m = re.match(r"^(_ZN\S*conv0_bwd_fused_kernel\S*):", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = [l.split(";")[0].strip() for l in lines[i:j]]
        i = j
        found += 1
        idx = [k for k, l in enumerate(body) if l.startswith("v_mfma_")]
        span = body[idx[0]:idx[-1] + 1]
        waterfall = [l for l in span if l.startswith("v_readfirstlane")]
        full = [l for l in span if re.match(r"s_waitcnt.*vmcnt\(0\)", l)]
        scratch = [l for l in body if l.startswith("scratch_")]
        # the steady-state waits leave more than one whole group's requests (3 + 8 loads) in flight (the first / last trips wait for less)
        waits = sorted(set(int(v) for l in span for v in re.findall(r"vmcnt\((\d+)\)", l)))
        nload = sum(1 for l in span if l.startswith("buffer_load_ushort"))
        ok = not waterfall and not full and not scratch and waits and waits[-1] >= 16 and nload >= 24
        act = re.search(r"conv0_bwd_fused_kernelILi(\d+)E", m.group(1))
        print("conv0_bwd_fused_kernel<%s>  mfma %3d  waterfall %d  vmcnt(0) %d  scratch %d  counted waits %s  %s" % (
            act.group(1) if act else "?", len(idx), len(waterfall), len(full), len(scratch), waits, "ok" if ok else "BAD"))
        bad += 0 if ok else 1
    if not found:
        print("no conv0_bwd_fused_kernel found")
        return 1
    return bad


def check_conv_stem(d):
    """conv_stem.hip (halo forward kernels, the stem pair, the one-launch stem data gradients, layer 0's staged forward): their LDS patches
    arrive by direct-to-LDS loads and are double-buffered -- the next tile's patch is requested under this tile's MFMAs.  They do NOT count
    waits: a patch is consumed behind an explicit `s_waitcnt vmcnt(0)` + barrier.  What the generated code must keep (VERDICT r4 next #4c):
    (1) for every direct-to-LDS load, the next barrier in program order (wrapping to the tile loop's first barrier) has a full vmcnt(0)
    wait within the dozen instructions in front of it -- the compiler may move or merge waits, it must not drop this one;
    (2) no scratch operation between the first and the last MFMA (a spill there would sit between a patch's request and its wait)."""
    src = os.path.join(ROOT, "rotate-yolov3_amd", "csrc", "conv_stem.hip")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
           "-c", src, "-o", os.path.join(d, "conv_stem.o")]
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "conv_stem-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    bad, found, i = 0, 0, 0
    while i < len(lines):
        m = re.match(r"^(_Z\S+):", lines[i])
        if not m or "kernel" not in m.group(1):
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = [l.split(";")[0].strip() for l in lines[i:j]]
        i = j
        idx = [k for k, l in enumerate(body) if l.startswith("v_mfma_")]
        dma = [k for k, l in enumerate(body) if "buffer_load" in l and " lds" in l]
        if not idx or not dma:
            continue
        found += 1
        bars = [k for k, l in enumerate(body) if l.startswith("s_barrier")]
        full = [k for k, l in enumerate(body) if re.match(r"s_waitcnt.*vmcnt\(0\)", l)]
        scratch = [k for k in range(idx[0], idx[-1] + 1) if body[k].startswith("scratch_")]
        loop_bars = [b for b in bars if b > dma[0]] or bars
        unguarded = []
        for q in dma:
            nxt = [b for b in bars if b > q]
            b = nxt[0] if nxt else loop_bars[0]
            if not any(b - 12 <= w < b for w in full):
                unguarded.append((q, b))
        ok = not scratch and not unguarded and bars
        short = re.sub(r"^_ZN\d+ryolo_detail(\d+_GLOBAL__N_1)?\d+", "", m.group(1))[:44]
        print("%-46s mfma %3d  lds-dma %2d  barriers %d  vmcnt(0) %2d  unguarded patches %d  scratch in loop %d  %s" % (
            short, len(idx), len(dma), len(bars), len(full), len(unguarded), len(scratch), "ok" if ok else "BAD"))
        bad += 0 if ok else 1
    if found < 8:
        print("conv_stem.hip: only %d staged kernels found" % found)
        return 1
    return bad


def check_wgrad_wide(d):
    """train.hip's wgrad_wide_kernel: three LDS stages retired by `s_waitcnt vmcnt(NLD)`.  Its direct-to-LDS loads are inline
    assembly precisely because the compiler puts a full vmcnt(0) in front of LDS reads that follow loads it knows about; this
    fails if such a wait (or a spill) shows up between the kernel's MFMAs again."""
    src = os.path.join(ROOT, "rotate-yolov3_amd", "csrc", "train.hip")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
           "-c", src, "-o", os.path.join(d, "train.o")]
    subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = open(os.path.join(d, "train-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    bad, found = 0, 0
    i = 0
    while i < len(lines):
        m = re.match(r"^(_ZN\S*wgrad_wide_kernel\S*):", lines[i])
        if not m:
            i += 1
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]
        idx = [k for k, l in enumerate(body) if "v_mfma_" in l]
        # the loop body as laid out (rotated: MFMA block first, then the counted wait + barrier + the next stage's fills):
        # from just before the first MFMA to the last direct-to-LDS load that follows the MFMAs
        dmas = [k for k, l in enumerate(body) if "buffer_load_dwordx4" in l and " lds" in l]
        hi = max([k for k in dmas if k > idx[-1]] + [idx[-1]])
        lo = max(idx[0] - 40, 0)
        span = body[lo:hi + 1]
        waits = [l for l in span if re.match(r"\s*s_waitcnt vmcnt\(([1-9]\d*)\)", l)]
        full = [l.strip() for l in span if re.match(r"\s*s_waitcnt.*vmcnt\(0\)", l)]
        scratch = [l.strip() for l in span if re.match(r"\s*scratch_", l)]
        dma = len([k for k in dmas if lo <= k <= hi])
        print("%-44s mfma %3d  lds-dma in loop %2d  counted waits %d  vmcnt(0) in loop %d  scratch %d" % (
            m.group(1)[17:61], len(idx), dma, len(waits), len(full), len(scratch)))
        found += 1
        if full or scratch or not waits or dma == 0:
            bad += 1
        i = j
    if not found:
        print("no wgrad_wide_kernel found")
        return 1
    return bad


def check_wgrad_reduce_batch(d):
    """train.hip's wgrad_reduce_batch_kernel (check_wgrad_wide compiled the unit): the batched split-K reduce streams at the HBM roofline only
    while every load of a split quarter is in flight at once.  A first version with a branch around each load had a compiler-inserted
    `s_waitcnt vmcnt(0)` in front of EVERY load (one round trip per split).  This fails unless the kernel holds runs of >= 15 16-B buffer
    loads (job kind 3: the 16-split pass and the 4 x 4 / 2 x 8 group passes) and a run of >= 9 4-B buffer loads (kind 4) with no vmcnt
    wait inside the run, and no scratch anywhere."""
    lines = open(os.path.join(d, "train-hip-amdgcn-amd-amdhsa-gfx950.s")).read().splitlines()
    for i, l in enumerate(lines):
        m = re.match(r"^(_ZN\S*wgrad_reduce_batch_kernel\S*):", l)
        if not m:
            continue
        j = i + 1
        while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
            j += 1
        body = lines[i:j]

        def runs(pat):
            out, cur = [], 0
            for b in body:
                if re.match(pat, b):
                    cur += 1
                elif re.match(r"\s*s_waitcnt.*vmcnt", b) or re.match(r"\s*(s_cbranch|s_branch|s_barrier)", b) or re.match(r"^\.LBB", b):
                    if cur:
                        out.append(cur)
                    cur = 0
            if cur:
                out.append(cur)
            return sorted(out, reverse=True)
        r16, r4 = runs(r"\s*buffer_load_dwordx4 "), runs(r"\s*buffer_load_dword ")
        scratch = [b.strip() for b in body if re.match(r"\s*scratch_", b)]
        print("wgrad_reduce_batch_kernel   runs of 16-B buffer loads without a wait %s  of 4-B buffer loads %s  scratch %d" % (r16[:4], r4[:3], len(scratch)))
        ok = len([x for x in r16 if x >= 15]) >= 3 and r4 and r4[0] >= 9 and not scratch
        return 0 if ok else 1
    print("no wgrad_reduce_batch_kernel found")
    return 1


def _vregs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def store_data_distances(asm):
    """For every 12-/16-B vector-memory store: the number of wait states (instructions, s_nop N = N+1) before the first
    VALU instruction that overwrites one of its data VGPRs, looking 12 instructions ahead inside the basic block.
    Yields (kernel, wait_states, store, writer)."""
    cur, ins = None, []
    blocks = {}
    for l in asm.splitlines():
        m = re.match(r"^(_Z\S+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        if cur is None:
            continue
        t = l.split(";")[0].strip()
        if l.startswith(".Lfunc_end"):
            cur = None
        elif t and not t.startswith("."):
            blocks[cur].append(t)
    for k, ins in blocks.items():
        for i, t in enumerate(ins):
            if not re.match(r"(buffer|global|flat)_store_dwordx[34]\b", t):
                continue
            ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
            data = _vregs(ops[0]) if t.startswith("buffer") else _vregs(ops[1])
            ws = 0
            for u in ins[i + 1:i + 13]:
                m = re.match(r"s_nop (\d+)", u)
                if m:
                    ws += int(m.group(1)) + 1
                    continue
                if re.match(r"s_(cbranch|branch|endpgm|setpc)", u):
                    break
                if u.startswith("v_"):
                    parts = [o.strip() for o in u.split(None, 1)[1].split(",")]
                    wr = _vregs(parts[0])
                    if "permlane" in u and "swap" in u and len(parts) > 1:
                        wr |= _vregs(parts[1])
                    if wr & data:
                        yield k, ws, t, u
                        break
                ws += 1


def check_store_data_hazard(d):
    """Every translation unit of the library: a VALU write to the data registers of a 12-/16-B store must be at least two
    wait states behind the store.  The compiler guarantees one wait state only for stores WITHOUT a register soffset; gfx950
    corrupts the stored data (lanes 12-15 of each row of a data dword) when a register-soffset store is followed directly by
    the overwrite -- conv_common.h buffer_store16_soff() is the guarded form."""
    csrc = os.path.join(ROOT, "rotate-yolov3_amd", "csrc")
    bad = 0
    for unit in sorted(u for u in os.listdir(csrc) if u.endswith(".hip")):
        stem = unit[:-4]
        sfile = os.path.join(d, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        if not os.path.exists(sfile):
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-unused-value", "-save-temps",
                   "-I", os.path.join(ROOT, "include"), "-c", os.path.join(csrc, unit), "-o", os.path.join(d, stem + ".o")]
            subprocess.run(cmd, check=True, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        found = list(store_data_distances(open(sfile).read()))
        worst = min([w for _, w, _, _ in found], default=None)
        soff = [x for x in found if x[2].startswith("buffer") and re.match(r"s\d+", x[2].split(",")[3].split()[0])]
        print("%-14s wide stores whose data is rewritten within 12 instructions: %3d (register soffset: %3d)  min wait states %s" % (
            unit, len(found), len(soff), worst))
        for k, w, st, wr in found:
            if w < 2 and st.startswith("buffer"):
                bad += 1
                print("    STORE-DATA HAZARD in %s: %s -> %s (%d wait states)" % (k[:60], st, wr, w))
    return bad


if __name__ == "__main__":
    sys.exit(main())
