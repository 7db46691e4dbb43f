#!/usr/bin/env python3
"""Issue-slot model of the headline tick kernel next to its measured phase times.

On this design the serial phases of a tick run on ONE wave per workgroup, and one wave issues one instruction of any
kind per ~4.4 cycles (tools/ubench_issue.hip).  The floor of a serial phase is therefore
    (instructions the critical wave executes) x 4.4 cycles,
which is what the stage loops were written against.  This tool takes
  * the static instruction count per stage of the two stage loops inside the Arnoldi loop, from the compiled ISA
    (state sweep: the depth-4 loop with the stage-table stores and no v_rndne_f64 = the rotation-mode stage loop, 2 stages
    per trip; costate sweep: the depth-3 loop with 16-byte LDS reads, 3 stages per trip), and
  * the measured shader cycles per phase from a phase-stamp log (tools/phase_stamps.py, diagnostic build on the GPU),
and writes profiles/<name>.json: per phase instructions/stage, floor cycles/stage, measured cycles/stage and their
ratio, plus the whole-tick view (share of the tick spent in the two sweeps, the tick if both ran at their floor).
    python tools/issue_model.py gpurun_out/r02_stamps_full.log profiles/r02_issue_model.json"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import isa_summary as isa  # noqa: E402

SLOT_CYCLES = 4.4
KERNEL = "_ZN3cgm14tick_wg_kernelINS_11PendulumDevIdEEdLi16ELi10ELb0ELi1EEEvNS_8WgParamsIT0_EE"  # (full plan, chunk-parallel costate)


def stage_loops(asm):
    m = re.search(r"^(" + KERNEL + r"):.*?\.end_amdhsa_kernel", asm, re.S | re.M)
    body = m.group(0)
    blocks, cur = [], None
    for line in body.split("\n"):
        t = line.strip()
        mm = re.match(r"^(\.LBB\d+_\d+):", t)
        if mm:
            cur = dict(label=mm.group(1), depth=0, ops=[], self_loop=False)
            blocks.append(cur)
            continue
        if cur is None or not t:
            continue
        if t.startswith(";"):
            d = re.findall(r"Depth=(\d+)", t)
            if d:
                cur["depth"] = max(cur["depth"], int(d[-1]))
            continue
        if t.startswith("."):
            continue
        cur["ops"].append(t)
        if t.startswith("s_cbranch") and t.split()[-1] == cur["label"]:
            cur["self_loop"] = True
    # Arnoldi loop = depth 2; inside it the state sweep sits in the chunk loop (depth 4), the costate sweep at depth 3
    out = {}
    for b in blocks:
        if not b["self_loop"]:
            continue
        ops = [o.split()[0] for o in b["ops"]]
        if any(o == "v_trig_preop_f64" for o in ops):
            continue
        if b["depth"] == 4 and sum(o.startswith("ds_write") for o in ops) >= 4 and \
                not any(o.startswith("v_rndne_f64") for o in ops):
            # the rotation-mode stage loop (no argument reduction); the fresh-evaluation loop (v_rndne_f64) is the fallback
            # (one u0 fetch per stage: the main loop does 4 stages per trip, its remainder loop 2 — keep the main one)
            st = sum(o.startswith("ds_read") for o in ops)
            if "state" not in out or out["state"]["mode"] != "rotation" or st > out["state"]["stages_per_trip"]:
                out["state"] = dict(label=b["label"], instructions_per_trip=len(ops), stages_per_trip=st,
                                    lds_ops=sum(o.startswith("ds_") for o in ops), mode="rotation")
        elif b["depth"] == 4 and any(o.startswith("v_rndne_f64") for o in ops) and "state" not in out:
            out["state"] = dict(label=b["label"], instructions_per_trip=len(ops), stages_per_trip=2,
                                lds_ops=sum(o.startswith("ds_") for o in ops), mode="fresh evaluation")
        elif b["depth"] == 3 and sum(o.startswith("ds_read_b128") for o in ops) >= 9:
            # the direct/particular lanes of the chunk-parallel sweep (wave 0): 3 coefficient pairs per stage, 3 stages per
            # trip (the transfer-matrix lanes of waves 1-3 fetch 2 pairs per stage: 6 reads per trip)
            out["costate"] = dict(label=b["label"], instructions_per_trip=len(ops), stages_per_trip=3,
                                  lds_ops=sum(o.startswith("ds_") for o in ops))
    return out


def parse_stamps(path):
    txt = open(path).read()
    head = re.search(r"ticks (\d+): shader cycles/tick (\d+), wall ([\d.]+) us/tick -> clock ([\d.]+) GHz", txt)
    ph = {m.group(1).strip(): (float(m.group(2)), float(m.group(4)))
          for m in re.finditer(r"^\s+(.+?)\s+(\d+) cyc/tick\s+([\d.]+)%\s+\((\d+) visits\)", txt, re.M)}
    return dict(cycles_per_tick=float(head.group(2)), clock_ghz=float(head.group(4))), ph


def main():
    stamps_log, out_path = sys.argv[1], sys.argv[2]
    dv, kmax = 50, 10
    asm, _ = isa.compile_tu(os.path.join(isa.CSRC, "inst_pendulum_f64.hip"))
    loops = stage_loops(asm)
    head, ph = parse_stamps(stamps_log)
    res = {"what": "issue-slot floor of the serial stage loops of tick_wg_kernel<PendulumDev<double>,double,16,10> vs the "
                   "measured shader cycles (diagnostic stamp build; the stamps themselves cost ~6 % of the tick)",
           "slot_cycles": SLOT_CYCLES, "slot_source": "tools/ubench_issue.hip: one instruction of any kind per ~4.4 cycles per wave",
           "stamps_log": os.path.basename(stamps_log), "measured_cycles_per_tick": head["cycles_per_tick"],
           "clock_ghz": head["clock_ghz"], "phases": {}}
    floor_total, meas_total = 0.0, 0.0
    n_wave0 = dv - 3 * (dv // 4)  # stages of the direct chunk = trips of wave 0 in the chunk-parallel costate sweep
    for name in ("state", "costate"):
        lp = loops[name]
        if name == "state":
            cyc, visits = ph["sweep phase 1 (state)"]
            stages = dv
        else:
            keys = [k for k in ph if k.startswith("costate") or k.startswith("sweep phase 3")]
            cyc, visits = sum(ph[k][0] for k in keys), max(ph[k][1] for k in keys)
            stages = n_wave0
        ips = lp["instructions_per_trip"] / lp["stages_per_trip"]
        floor = ips * SLOT_CYCLES
        meas = cyc / visits / stages
        res["phases"][name] = {"instructions_per_stage": ips, "floor_cycles_per_stage": floor,
                               "stages_on_the_critical_wave": stages,
                               "measured_cycles_per_stage": meas, "floor_over_measured": floor / meas,
                               "sweeps_per_tick_on_the_critical_path": visits, "share_of_tick": cyc / head["cycles_per_tick"],
                               "isa_loop": lp}
        if name == "costate" and "costate A: stage loop" in ph:
            # finer stamps (tools/phase_stamps.py ids 18-20): the stage loop alone — dv/4 common stages per sweep (3 per trip)
            lc, lv = ph["costate A: stage loop"]
            res["phases"][name]["stage_loop_alone"] = {
                "stages": dv // 4, "measured_cycles_per_stage": lc / lv / (dv // 4), "floor_over_measured": floor / (lc / lv / (dv // 4)),
                "note": "one stamp (~190 cycles) per sweep included; the rest of the phase is the prologue (flag, addresses, terminal "
                        "costate, first fetches), the two tail stages of the direct chunk, the boundary-record store, the barrier and "
                        "the combine phase B"}
        if name == "costate":
            res["phases"][name]["note"] = ("chunk-parallel sweep: wave 0 walks dv - 3*(dv/4) stages; the measured time also holds the "
                                           "barrier and the boundary/combine phase (~120 instructions per thread), which the per-stage floor does not count")
        floor_total += floor * stages * visits
        meas_total += cyc
    res["sweeps"] = {"share_of_tick": meas_total / head["cycles_per_tick"], "floor_cycles_per_tick": floor_total,
                     "measured_cycles_per_tick": meas_total, "floor_over_measured": floor_total / meas_total}
    res["tick_if_sweeps_ran_at_their_floor_cycles"] = head["cycles_per_tick"] - meas_total + floor_total
    res["critical_wave_sweep_instructions_per_tick"] = sum(
        res["phases"][n]["instructions_per_stage"] * res["phases"][n]["stages_on_the_critical_wave"] * res["phases"][n]["sweeps_per_tick_on_the_critical_path"]
        for n in res["phases"])
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
