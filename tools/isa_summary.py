#!/usr/bin/env python3
"""ISA summary of every tick kernel instantiation (no GPU needed): registers, spills, scratch, and WHERE the scratch
traffic and AGPR copies sit — inside a loop block (the stage loops, the Gram-Schmidt rounds) or in straight-line code.

    python tools/isa_summary.py [--out profiles/r02_isa_summary.md] [pattern]

For every inst_*.hip translation unit: hipcc -S --cuda-device-only -Rpass-analysis=kernel-resource-usage, then per
kernel: VGPR / AGPR / SGPR, VGPR+SGPR spill counts, scratch bytes per lane (compiler remarks), the number of
scratch_load/scratch_store and v_accvgpr_read/write instructions in the whole kernel and inside blocks the compiler
marks as loops (`; %bb... Loop Header/ in Loop:` annotations), listed per loop with its instruction count."""
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cgmres_cpp_amd", "csrc")
TMP = "/tmp/asm"


def compile_tu(src):
    os.makedirs(TMP, exist_ok=True)
    out = os.path.join(TMP, os.path.basename(src) + ".s")
    r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-S", "--cuda-device-only",
                        "-Rpass-analysis=kernel-resource-usage", "-o", out, src] + EXTRA, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    return open(out).read(), r.stderr


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.strip().split("\n")))


def remarks(rp):
    out = {}
    for m in re.finditer(r"Function Name: (\S+) \[.*?(?=Function Name:|\Z)", rp, re.S):
        blk = m.group(0)

        def g(key):
            mm = re.search(key + r": (\d+)", blk)
            return int(mm.group(1)) if mm else None
        out[m.group(1)] = dict(sgpr=g("TotalSGPRs"), vgpr=g("VGPRs"), agpr=g("AGPRs"), scratch=g(r"ScratchSize \[bytes/lane\]"),
                               sgpr_spill=g("SGPRs Spill"), vgpr_spill=g("VGPRs Spill"), occ=g(r"Occupancy \[waves/SIMD\]"))
    return out


def scan(body):
    """Counts of scratch and AGPR-copy instructions per loop nest level.
    The tick kernel is one big loop over the fused ticks (depth 1); depth 2 = code that runs once per Arnoldi
    iteration (Gram-Schmidt rounds, Hessenberg column) and the stage loops of the preamble; depth >= 3 = the stage
    loops inside the Arnoldi loop (the critical path).  Loops that contain v_trig_preop_f64 (library
    sincos inlined) or a call (library sincos out of line, lean kernels) are the redo of a chunk whose arguments left
    the fast trig range: counted separately as `slow`."""
    blocks, cur = [], None
    for line in body.split("\n"):
        t = line.strip()
        mm = re.match(r"^(\.LBB\d+_\d+):", t)
        if mm:
            cur = dict(label=mm.group(1)[2:], header=None, depth=0, ops=[])
            blocks.append(cur)
            continue
        if cur is None:
            continue
        if t.startswith(";"):
            d = re.search(r"Loop Header: Depth=(\d+)", t)
            if d:
                cur["header"], cur["depth"] = cur["label"], max(cur["depth"], int(d.group(1)))
            d = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", t)
            if d and int(d.group(2)) >= cur["depth"]:
                cur["header"], cur["depth"] = d.group(1), int(d.group(2))
            continue
        if not t or t.startswith("."):
            continue
        cur["ops"].append(t.split()[0])
    slow_headers = {b["header"] for b in blocks
                    if b["header"] and any(o in ("v_trig_preop_f64", "s_swappc_b64") for o in b["ops"])}
    levels = {k: dict(n=0, scratch=0, acc=0) for k in ("tick", "iter", "stage", "slow", "outside")}
    loops = {}
    for b in blocks:
        key = ("slow" if b["header"] in slow_headers else
               "outside" if b["depth"] == 0 else "tick" if b["depth"] == 1 else "iter" if b["depth"] == 2 else "stage")
        sc = sum(o.startswith("scratch_") for o in b["ops"])
        ac = sum(o.startswith("v_accvgpr") for o in b["ops"])
        lv = levels[key]
        lv["n"] += len(b["ops"]); lv["scratch"] += sc; lv["acc"] += ac
        if key == "stage":
            l = loops.setdefault(b["header"], dict(n=0, scratch=0, acc=0, ds=0, vmem=0))
            l["n"] += len(b["ops"]); l["scratch"] += sc; l["acc"] += ac
            l["ds"] += sum(o.startswith("ds_") for o in b["ops"])
            l["vmem"] += sum(o.startswith(("global_", "buffer_", "flat_")) for o in b["ops"])
    return levels, loops


EXTRA = []


def main():
    args = [a for a in sys.argv[1:]]
    out_path = None
    if "--out" in args:
        i = args.index("--out")
        out_path = args[i + 1]
        del args[i:i + 2]
    for a in list(args):
        if a.startswith("-D"):
            EXTRA.append(a)
            args.remove(a)
    pat = args[0] if args else "tick_wg_kernel"
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.startswith("inst_") and f.endswith(".hip"))
    with ThreadPoolExecutor(min(6, len(os.sched_getaffinity(0)))) as ex:
        res = list(ex.map(compile_tu, srcs))
    lines = ["# ISA summary of the tick kernels (gfx950, hipcc -O3; generated by tools/isa_summary.py)", "",
             "Per kernel: registers and spill counts from the compiler remarks, then WHERE scratch_load/store instructions and",
             "VGPR<->AGPR copies (v_accvgpr_*, one issue slot each) sit, by loop nest level: `stage` = the stage loops inside the",
             "Arnoldi loop (state / coefficient / costate sweeps: the critical path), `iter` = once per Arnoldi iteration",
             "(Gram-Schmidt rounds, Hessenberg column) and the preamble's stage loops, `tick` = once per control tick,",
             "`slow` = the library-sincos redo of a chunk of stages (arguments beyond the fast trig range; never taken in the benchmarks).",
             "Entries are `scratch/acc` instruction counts (static).", "",
             "| kernel | VGPR | AGPR | VGPR spill | SGPR spill | scratch B/lane | instrs | stage | iter | tick | slow |",
             "|---|---|---|---|---|---|---|---|---|---|---|"]
    detail = []
    for (asm, rp), src in zip(res, srcs):
        rm = remarks(rp)
        names = [m.group(1) for m in re.finditer(r"^(_Z\w+):", asm, re.M) if pat in m.group(1)]
        dm = demangle(names) if names else {}
        for name in names:  # (by name: a device FUNCTION ahead of a kernel must not swallow it)
            m = re.search(r"^" + re.escape(name) + r":.*?\.end_amdhsa_kernel", asm, re.S | re.M)
            if not m:
                continue
            lv, loops = scan(m.group(0))
            r = rm.get(name, {})
            short = re.sub(r"cgm::|void |\(cgm::WgParams<\w+>\)|\(cgm::LaneParams<\w+>\)", "", dm.get(name, name))
            cell = lambda k: f"{lv[k]['scratch']}/{lv[k]['acc']}"
            lines.append(f"| `{short}` | {r.get('vgpr')} | {r.get('agpr')} | {r.get('vgpr_spill')} | {r.get('sgpr_spill')} | "
                         f"{r.get('scratch')} | {sum(v['n'] for v in lv.values())} | {cell('stage')} | {cell('iter')} | "
                         f"{cell('tick')} | {cell('slow')} |")
            hot = {h: l for h, l in loops.items() if l["scratch"] or l["acc"]}
            if hot:
                detail.append(f"\n`{short}` — stage loops with scratch / AGPR copies: " +
                              "; ".join(f"{h}: {l['n']} instrs (ds {l['ds']}, vmem {l['vmem']}) scratch {l['scratch']} acc {l['acc']}"
                                        for h, l in hot.items()))
    text = "\n".join(lines + [""] + detail) + "\n"
    if out_path:
        open(os.path.join(ROOT, out_path) if not os.path.isabs(out_path) else out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main()
