"""Derived table from the raw "PMC <kernel> <counter> n= avg= sum=" rows tools/pmc_cmd.sh prints (tools/rocpd_summary.py --pmc):
    python tools/pmc_table.py profiles/r06_pmc_wgrad_raw.txt [kernel-substring]
Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES
counts cycles summed over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs; SQ_LDS_IDX_ACTIVE counts LDS-array cycles summed over CUs."""
import re
import sys


def parse(path, filt=""):
    sections, cur, title = [], None, None
    for ln in open(path):
        if ln.startswith("#"):
            title = ln[1:].strip()
            cur = None
            continue
        m = re.match(r"PMC (.*?)\s+([A-Z][A-Z0-9_]+)\s+n=\s*(\d+) avg=([-+0-9.e]+) sum=", ln)
        if not m or filt not in m.group(1):
            continue
        k = re.sub(r"\(anonymous namespace\)::", "", m.group(1)).split("(")[0].replace("void ", "").strip()
        if cur is None or cur["title"] != title:
            cur = {"title": title, "k": {}}
            sections.append(cur)
        cur["k"].setdefault(k, {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
    return sections


def row(name, val, note=""):
    print("    %-46s %14s  %s" % (name, val, note))


def main():
    secs = parse(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
    for sec in secs:
        print("# " + (sec["title"] or ""))
        for k, c in sec["k"].items():
            g = lambda n: c[n][1] if n in c else None       # noqa: E731  per-launch average
            n = next(iter(c.values()))[0]
            print("  %s   (%d launches per pass)" % (k, n))
            gui = g("GRBM_GUI_ACTIVE")
            cyc = gui / 8.0 if gui else None                  # shader-engine cycles of one launch
            wc = g("SQ_WAVE_CYCLES")
            if cyc:
                row("elapsed (GRBM_GUI_ACTIVE / 8 XCDs)", "%.0f cyc" % cyc)
            if g("SQ_VALU_MFMA_BUSY_CYCLES") and cyc:
                row("MFMA pipes busy", "%.1f %%" % (100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc)), "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x elapsed)")
            if wc:
                for nm, lab in (("SQ_ACTIVE_INST_ANY", "wave time issuing an instruction"), ("SQ_WAIT_ANY", "wave time in s_waitcnt / barrier"),
                                ("SQ_WAIT_INST_ANY", "wave time waiting to issue (dependency / arbitration)"), ("SQ_WAIT_INST_LDS", "  of it: waiting on LDS"),
                                ("SQ_ACTIVE_INST_VALU", "wave time issuing VALU (incl. MFMA)"), ("SQ_ACTIVE_INST_SCA", "wave time issuing scalar"),
                                ("SQ_ACTIVE_INST_LDS", "wave time issuing LDS"), ("SQ_ACTIVE_INST_VMEM", "wave time issuing VMEM")):
                    if g(nm) is not None:
                        row(lab, "%.1f %%" % (100.0 * g(nm) / wc), nm + " / SQ_WAVE_CYCLES")
            waves = g("SQ_WAVES")
            if waves and wc and cyc:
                row("wave residency / elapsed", "%.2f" % (4.0 * wc / waves / cyc), "(4 x SQ_WAVE_CYCLES / SQ_WAVES) / elapsed")
            mf = g("SQ_INSTS_MFMA")
            if mf:
                if g("SQ_INSTS_VALU") is not None:
                    row("VALU instructions per MFMA (non-MFMA)", "%.2f" % ((g("SQ_INSTS_VALU") - mf) / mf))
                for nm, lab in (("SQ_INSTS_SALU", "SALU instructions per MFMA"), ("SQ_INSTS_LDS", "LDS instructions per MFMA"), ("SQ_INSTS_VMEM", "VMEM instructions per MFMA"),
                                ("SQ_INSTS_BRANCH", "branches per MFMA")):
                    if g(nm) is not None:
                        row(lab, "%.3f" % (g(nm) / mf))
            else:
                for nm, lab in (("SQ_INSTS_VALU", "VALU instructions"), ("SQ_INSTS_SALU", "SALU instructions"), ("SQ_INSTS_LDS", "LDS instructions"),
                                ("SQ_INSTS_VMEM", "VMEM instructions"), ("SQ_INSTS_BRANCH", "branches")):
                    if g(nm) is not None:
                        row(lab, "%.4g" % g(nm))
            if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
                row("active lanes per VALU issue", "%.1f of 64" % (g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU")),
                    "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU (64.0 on a kernel whose waves never diverge: wgrad_wide)")
            if g("SQ_LDS_IDX_ACTIVE") is not None and cyc:
                row("LDS array busy", "%.1f %%" % (100.0 * g("SQ_LDS_IDX_ACTIVE") / (256.0 * cyc)), "SQ_LDS_IDX_ACTIVE / (256 CUs x elapsed)")
            for nm in ("SQ_LDS_BANK_CONFLICT", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_DATA_FIFO_FULL", "SQ_LDS_CMD_FIFO_FULL", "SQ_VMEM_TA_ADDR_FIFO_FULL", "SQ_VALU_MFMA_COEXEC_CYCLES"):
                if g(nm) is not None:
                    row(nm, "%.4g" % g(nm), "per launch")


if __name__ == "__main__":
    main()
