#!/usr/bin/env python3
"""Condense the rocprofv3 CSVs written by tools/profile_bench.sh / tools/profile_any.sh into one JSON per kernel.
usage: tools/summarise_pmc.py gpurun_out/prof_<tag> <kernel-name-substring> <out.json> [algorithmic_bytes]

Durations: `trace_avg_seconds` is the kernel-trace pass's average (kernel_stats.csv); every PMC pass also carries its
own Start/End time stamps per dispatch, and `effective_clock_GHz` uses GRBM_GUI_ACTIVE and the duration OF THE SAME
PASS (round 2 divided the PMC pass's cycles by the trace pass's duration and got 2.96 GHz).  GRBM_GUI_ACTIVE is summed
over the 8 XCDs by rocprofv3, hence / 8."""
import collections
import csv
import glob
import json
import sys

prof, kname, outp = sys.argv[1], sys.argv[2], sys.argv[3]
alg_bytes = float(sys.argv[4]) if len(sys.argv) > 4 else None
ctr = {}         # kernel -> pass -> counter -> sum
disp = {}        # (kernel, pass, counter) -> dispatch ids
dur = {}         # (pass, kernel) -> {dispatch id: seconds}
PASSES = ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_wait", "pmc_lds", "pmc_lanes")
for d in PASSES:
    for f in glob.glob("%s/%s/*/*_counter_collection.csv" % (prof, d)):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            c = r["Counter_Name"]
            disp.setdefault((k, d, c), set()).add((f, r["Dispatch_Id"]))
            ctr.setdefault(k, {}).setdefault(d, collections.defaultdict(float))[c] += float(r["Counter_Value"])
            if "Start_Timestamp" in r:
                dur.setdefault((d, k), {})[(f, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
# per-launch averages: a counter row exists per dispatch (and per XCD/SE slice), so divide by dispatches
per_pass = {k: {d: {c: v / max(1, len(disp[(k, d, c)])) for c, v in cs.items()} for d, cs in ps.items()} for k, ps in ctr.items()}
out = {}
for k, ps in per_pass.items():
    flat = {}
    for d in PASSES:
        for c, v in ps.get(d, {}).items():
            flat.setdefault(c, v)          # a counter collected in two passes: the first pass's value (both are in per_pass)
    out[k] = flat
keys = [k for k in out if kname in k]
if not keys:
    sys.exit("no kernel matching %r among %s" % (kname, sorted(out)))
key = max(keys, key=lambda k: out[k].get("SQ_WAVE_CYCLES", out[k].get("FETCH_SIZE", 0)))
r = out[key]
trace = {}
for f in glob.glob("%s/trace/*/*kernel_stats.csv" % prof):
    for row in csv.DictReader(open(f)):
        trace[row["Name"].split("(")[0].replace("void ", "")] = row
tr = trace.get(key)
pass_avg = {d: sum(v.values()) / len(v) for (d, k), v in dur.items() if k == key and v}
fetch_b, write_b = r.get("FETCH_SIZE", 0) * 1024, r.get("WRITE_SIZE", 0) * 1024
sq_s = pass_avg.get("pmc_sq")
summary = {
    "command": "rocprofv3 --kernel-trace --stats | --pmc <one counter group per pass> --output-format csv -- python3 <program> "
               "(tools/profile_any.sh / tools/profile_bench.sh)",
    "kernel": key,
    "trace": None if tr is None else {"calls": int(tr["Calls"]), "avg_seconds": float(tr["AverageNs"]) * 1e-9,
                                      "min_seconds": float(tr["MinNs"]) * 1e-9, "max_seconds": float(tr["MaxNs"]) * 1e-9},
    "pmc_pass_avg_seconds": pass_avg,
    "per_launch_counters": r,
    "per_pass_counters": per_pass[key],
    "derived": {
        "FETCH_SIZE_bytes": fetch_b,
        "FETCH_SIZE_bytes_x2_gfx950_correction": 2 * fetch_b,   # MI355X_MICROARCH.md §HBM
        "WRITE_SIZE_bytes": write_b,
        "hbm_traffic_bytes_corrected": 2 * fetch_b + write_b,
    },
}
dv = summary["derived"]
if sq_s and r.get("GRBM_GUI_ACTIVE"):
    # two estimates of the clock from the SAME pass: GRBM_GUI_ACTIVE (summed over 8 XCDs; counts a little before and after
    # the kernel, so it reads high) and SQ_BUSY_CYCLES (summed over 32 shader engines; counts only while waves are resident,
    # so it reads low when an SE idles at the tail).  The VALU issue figure uses the lower one, i.e. is an upper bound.
    dv["clock_GHz_from_GRBM_GUI_ACTIVE"] = r["GRBM_GUI_ACTIVE"] / 8 / sq_s / 1e9
    if r.get("SQ_BUSY_CYCLES"):
        dv["clock_GHz_from_SQ_BUSY_CYCLES"] = r["SQ_BUSY_CYCLES"] / 32 / sq_s / 1e9
    dv["effective_clock_GHz"] = dv.get("clock_GHz_from_SQ_BUSY_CYCLES", dv["clock_GHz_from_GRBM_GUI_ACTIVE"])
    dv["effective_clock_source"] = "SQ_BUSY_CYCLES / 32 SEs / duration of the same PMC pass (pmc_sq)"
    if r.get("SQ_INSTS_VALU"):
        # instructions are a property of the launch; time: the un-perturbed kernel-trace average when there is one
        t_k = float(tr["AverageNs"]) * 1e-9 if tr is not None else sq_s
        dv["valu_wave_insts_per_clk_per_simd"] = r["SQ_INSTS_VALU"] / t_k / 1024 / (dv["effective_clock_GHz"] * 1e9)
        dv["valu_issue_frac_of_2.25_cycle_peak"] = dv["valu_wave_insts_per_clk_per_simd"] * 2.25
        # the same from ONE source — the SQ pass's own duration (counter passes perturb the kernel: 20-45 % slower)
        dv["valu_wave_insts_per_clk_per_simd_same_pass"] = r["SQ_INSTS_VALU"] / sq_s / 1024 / (dv["effective_clock_GHz"] * 1e9)
        dv["valu_issue_frac_same_pass"] = dv["valu_wave_insts_per_clk_per_simd_same_pass"] * 2.25
    if r.get("SQ_ACTIVE_INST_VALU"):
        # SQ_ACTIVE_INST_VALU counts quad-cycles the VALU pipe is occupied, summed over all SIMDs: x 4 / 1024 SIMDs = cycles
        # per SIMD, against the cycles of the same pass.  The saturation measure of a kernel whose mix is mostly 4-cycle forms.
        dv["valu_pipe_busy_frac"] = r["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (dv["effective_clock_GHz"] * 1e9 * sq_s)
        if r.get("SQ_INSTS_VALU"):
            dv["valu_cycles_per_instruction"] = r["SQ_ACTIVE_INST_VALU"] * 4 / r["SQ_INSTS_VALU"]
            # against the bound of the kernel's OWN mix: its instructions at the cycles per instruction they were measured to
            # hold the pipe, with the pipe never idle, take frac x the pass's time — the same number as the pipe-busy fraction,
            # named for what bench.py reports it as
            dv["valu_mix_bound_frac"] = dv["valu_pipe_busy_frac"]
            dv["valu_mix_bound_note"] = "SQ_INSTS_VALU x valu_cycles_per_instruction / 4 per SIMD / cycles of the pass (= pipe busy): 1.0 = the mix's own bound"
lanes = per_pass[key].get("pmc_lanes", {})
if lanes.get("SQ_THREAD_CYCLES_VALU") and lanes.get("SQ_ACTIVE_INST_VALU"):
    # lanes active per VALU cycle (tools/pmc_lanes.sh, round 3: 64.0 for full wavefronts, 37 for the generic kernel)
    dv["valu_lanes_active_of_64"] = lanes["SQ_THREAD_CYCLES_VALU"] / lanes["SQ_ACTIVE_INST_VALU"]
    dv["valu_lanes_active_source"] = "SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU of the pmc_lanes pass"
if r.get("SQ_WAVE_CYCLES"):
    wt = per_pass[key].get("pmc_wait", {})
    for c, name in (("SQ_WAIT_ANY", "wave_cycles_parked_frac"), ("SQ_WAIT_INST_ANY", "wave_cycles_waiting_to_issue_frac"),
                    ("SQ_ACTIVE_INST_ANY", "wave_cycles_issuing_frac")):
        if c in wt and wt.get("SQ_WAVE_CYCLES"):
            dv[name] = wt[c] / wt["SQ_WAVE_CYCLES"]
if alg_bytes and tr is not None:
    dv["algorithmic_bytes"] = alg_bytes
    dv["achieved_GBps_trace_avg"] = alg_bytes / (float(tr["AverageNs"]) * 1e-9) / 1e9
    dv["hbm_roofline_frac"] = dv["achieved_GBps_trace_avg"] / 8000.0
    dv["traffic_over_algorithmic"] = dv["hbm_traffic_bytes_corrected"] / alg_bytes
summary["all_kernels"] = out
json.dump(summary, open(outp, "w"), indent=1)
print(key, json.dumps({"trace": summary["trace"], "derived": dv}, indent=1))
