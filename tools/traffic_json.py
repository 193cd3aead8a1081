#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/pmc_pass.sh: HBM traffic (FETCH_SIZE x 2 + WRITE_SIZE, KB -> bytes; the
factor 2 is MI355X_MICROARCH.md's gfx950 correction: FETCH_SIZE counts 128-B requests at 64 B -- calibrated for this
kernel's 32-B random record gathers in round 6: tools/counter_calibration.py, profiles/r06_counter_calibration.txt) and issue-side figures of
the dominant kernel per launch, one entry per call size.
Per kernel VARIANT of the solve loop as well ("variants"), so that bench.py can put a variant's counter traffic next to the
algorithmic bytes of the same population of launches.
usage: tools/traffic_json.py <tag> <gpurun_out/pmc_320> <gpurun_out/pmc_2048> ...   (directory name ends in the call size; tag: r04)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
entries = []
tag = sys.argv[1]
stamps = {}
sp_ = os.path.join(ROOT, "profiles", f"{tag}_step_stamps.json")
if os.path.exists(sp_):
    stamps = json.load(open(sp_)).get("variants", {})
for spec in sys.argv[2:]:
    # <dir>[:robot[:grid[:shelf]]]   (default panda_5k, 128, table top): BASELINE configs[2] is fetch:128:shelf, configs[4] fetch_mobile:256:shelf
    parts = spec.split(":")
    d, robot_, grid_, shelf_ = parts[0], (parts[1] if len(parts) > 1 else "panda_5k"), int(parts[2]) if len(parts) > 2 else 128, len(parts) > 3 and parts[3] == "shelf"
    size = int(os.path.basename(d.rstrip("/")).split("_")[-1])
    js = json.load(open(os.path.join(d, "pmc_summary.json")))
    # the variant for the launches that fill the GPU (the one the roofline is quoted on); the variant for few instances in flight only if there is no other
    # (names as the C ABI's profile reports them: the crew variant behind an itemized launch is a row of its own)
    def norm(n):  # k_obstacle_gram<NP, PD, SWEEP, HOT> -> k_obstacle_gram<NP,PD>[crew]; others: blanks out
        import re
        m = re.match(r"k_obstacle_gram<(\d+), (\d+)(?:, (true|false))?(?:, (true|false))?>", n)
        if not m:
            return n.replace(", ", ",")
        return f"k_obstacle_gram<{m.group(1)},{m.group(2)}>" + ("crew" if m.group(3) == "true" else "") + ("" if m.group(4) in (None, "true") or m.group(3) == "true" else "general")
    # the solve loop's launches run the HOT variants; the general ones (init pass, value-only mode) are rows of their own
    js = {norm(n): v for n, v in js.items()}
    cands = [v for n, v in js.items() if (n.startswith("k_obstacle_gram<8,1>") or n.startswith("k_obstacle_gram<16,1>")) and "crew" not in n] or [v for n, v in js.items() if "k_obstacle_gram" in n]
    kname = [n for n, v in js.items() if v is max(cands, key=lambda v: v.get("launches", 0) * v.get("mean_us", 0.0))][0]
    k = max(cands, key=lambda v: v.get("launches", 0) * v.get("mean_us", 0.0))
    us = k["mean_us"]
    variants = {}
    for n_, v_ in js.items():
        if "k_obstacle_gram" in n_ or "k_lm_step" in n_:
            variants[n_.replace(", ", ",")] = {
                "launches": v_.get("launches"), "mean_launch_us_profiled": round(v_.get("mean_us", 0.0), 2),
                "fetch_bytes_per_launch": int(round(2 * v_.get("FETCH_SIZE", 0.0) * 1024)), "write_bytes_per_launch": int(round(v_.get("WRITE_SIZE", 0.0) * 1024)),
                "hbm_bytes_per_launch": int(round((2 * v_.get("FETCH_SIZE", 0.0) + v_.get("WRITE_SIZE", 0.0)) * 1024)),
                "l2_hit_rate": round(v_.get("TCC_HIT_sum", 0.0) / max(v_.get("TCC_REQ_sum", 1.0), 1.0), 3),
                "waves_waiting_frac": round(v_.get("SQ_WAIT_ANY", 0.0) / max(v_.get("SQ_WAVE_CYCLES", 1.0), 1.0), 3),
                "valu_insts_per_launch": int(v_.get("SQ_INSTS_VALU", 0)), "salu_insts_per_launch": int(v_.get("SQ_INSTS_SALU", 0)),
                "mfma_insts_per_launch": int(v_.get("SQ_INSTS_MFMA", 0)),
                # issue side: vector instructions x 4 cycles over the cycles of all 1024 SIMDs during the launch; waves of the launch
                # per SIMD; LDS bank-conflict cycles per LDS instruction
                "valu_issue_frac": round(4 * v_.get("SQ_INSTS_VALU", 0.0) / max(1024 * v_.get("mean_us", 0.0) * 1e-6 * 2.4e9, 1.0), 4),
                "waves_per_simd": round(v_.get("SQ_WAVES", 0.0) / 1024.0, 3),
                "lds_bank_conflict_cycles_per_lds_inst": round(v_.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v_.get("SQ_INSTS_LDS", 1.0), 1.0), 3)}
            st_ = stamps.get(n_.replace(", ", ","))
            if st_:  # stamped critical path of one workgroup (tools/step_stamps.py) over the launch's duration in the same ticks
                variants[n_.replace(", ", ",")]["critical_path_cycles"] = st_["critical_path_cycles"]
                variants[n_.replace(", ", ",")]["critical_path_over_launch"] = round(st_["critical_path_cycles"] / max(v_.get("mean_us", 0.0) * st_["ticks_per_us"], 1.0), 3)
    simd_cycles = 1024 * us * 1e-6 * 2.4e9  # 256 CUs x 4 SIMDs at 2.4 GHz
    entries.append({
        "robot": robot_, "grid": grid_, "shelf": bool(shelf_), "mode": "rounds", "instances_per_call": size, "slots": int(os.environ.get("GTO_SLOTS", "512")),
        "source": f"profiles/{tag}_{os.path.basename(d.rstrip('/'))}.txt (tools/pmc_pass.sh: rocprofv3 --kernel-trace --pmc, separate passes, bench.py --pipeline 1 "
                  f"--merged-launches-only with calls of {size} instances)",
        "kernel": kname, "variants": variants, "launches": k["launches"], "mean_launch_us_profiled": round(us, 2),
        "FETCH_SIZE_KB_mean_per_launch": round(k["FETCH_SIZE"], 1), "WRITE_SIZE_KB_mean_per_launch": round(k["WRITE_SIZE"], 1),
        "correction": "FETCH_SIZE doubled (a request is a 128-B line tallied at 64 B, for 32-B random gathers as for streaming reads; WRITE_SIZE exact for whole 64-B lines, 32 B per partial line: profiles/r06_counter_calibration.txt)",
        "hbm_bytes_per_launch": int(round((2 * k["FETCH_SIZE"] + k["WRITE_SIZE"]) * 1024)),
        "issue": {
            "fp64_valu_busy_frac": round(4 * k["SQ_ACTIVE_INST_VALU"] / simd_cycles, 3),
            "scalar_busy_frac": round(4 * k["SQ_ACTIVE_INST_SCA"] / simd_cycles, 3),
            "waves_waiting_frac": round(k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"], 3),
            "l2_hit_rate": round(k["TCC_HIT_sum"] / max(k["TCC_REQ_sum"], 1.0), 3),
            "valu_insts_per_launch": int(k["SQ_INSTS_VALU"]), "salu_insts_per_launch": int(k["SQ_INSTS_SALU"]),
            "lds_bank_conflict_cycles_per_lds_inst": round(k["SQ_LDS_BANK_CONFLICT"] / max(k["SQ_INSTS_LDS"], 1.0), 2),
            "how": "SQ_ACTIVE_INST_* x 4 (quad-cycles) / (1024 SIMDs x launch duration x 2.4 GHz); the kernel is bound by instruction "
                   "issue and latency, not by HBM"}})
out = {"note": "HBM traffic and issue-side figures of the dominant kernel per launch, by call size; bench.py quotes an entry only for the "
               "workload, solver mode and call size it was measured on", "entries": entries}
json.dump(out, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
