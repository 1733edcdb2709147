#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (gpurun_out/pmc_<tag>_*) per kernel: usage  tools/pmc_summary.py <tag> [out.md]"""
import collections, csv, glob, sys
tag = sys.argv[1]
out = []
def load(pat):
    fs = glob.glob(pat)
    return list(csv.DictReader(open(fs[0]))) if fs else []
C = collections.defaultdict(lambda: collections.defaultdict(list))
for name in ["sq1", "sq2", "fetch", "write", "grbm"]:
    for r in load(f"gpurun_out/pmc_{tag}_{name}/runc/*counter_collection.csv"):
        k = r["Kernel_Name"]
        if "ti::" not in k: continue
        C[k.split("(")[0].replace("void ti::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
D = collections.defaultdict(list)
for r in load(f"gpurun_out/pmc_{tag}_grbm/runc/*kernel_trace.csv"):
    if "ti::" in r["Kernel_Name"]:
        D[r["Kernel_Name"].split("(")[0].replace("void ti::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
out.append("| kernel | launches | ms (PMC run) | clock GHz | MFMA busy of SIMD-time | wave: MFMA-stall / parked / issuing | VALU inst per MFMA | HBM read MB (2x FETCH_SIZE KB) | HBM write MB |")
out.append("|---|---|---|---|---|---|---|---|---|")
for k, d in C.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    if "SQ_WAVE_CYCLES" not in m or k not in D: continue
    ms = sum(D[k]) / len(D[k])
    clk = m.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e-3) / 1e9
    simd_cycles = 1024 * ms * 1e-3 * clk * 1e9
    wc = m["SQ_WAVE_CYCLES"]
    out.append(f"| {k} | {len(D[k])} | {ms:.3f} | {clk:.2f} | {m['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles:.3f} | "
               f"{m['SQ_WAIT_INST_ANY'] / wc:.2f} / {m['SQ_WAIT_ANY'] / wc:.2f} / {m['SQ_ACTIVE_INST_ANY'] / wc:.2f} | "
               f"{m['SQ_INSTS_VALU'] / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f} | {2 * m.get('FETCH_SIZE', 0) / 1024:.1f} | {m.get('WRITE_SIZE', 0) / 1024:.1f} |")
# HBM traffic of the edge kernel per launch, launch-weighted over its variants, for bench.py's roofline.traffic
import json, os
tot = {"r": 0.0, "w": 0.0, "n": 0}
for k, d in C.items():
    if k.startswith(("painn_edge_kernel", "painn_pair_kernel")) and "FETCH_SIZE" in d and "WRITE_SIZE" in d:      # the message kernel, either layout
        n = len(d["FETCH_SIZE"])
        tot["r"] += 2 * sum(d["FETCH_SIZE"]) * 1024; tot["w"] += sum(d["WRITE_SIZE"]) * 1024; tot["n"] += n
if tot["n"] and os.environ.get("PMC_BATCH"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_sources_sha
    json.dump({"kernel": "message kernel (painn_pair_kernel / painn_edge_kernel)", "kernel_sources_sha": kernel_sources_sha(), "batch": int(os.environ["PMC_BATCH"]), "precision": os.environ.get("PMC_PRECISION", "f16x2"),
               "read_bytes_per_launch": tot["r"] / tot["n"], "write_bytes_per_launch": tot["w"] / tot["n"], "launches_measured": tot["n"],
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE (KB) doubled for 16-byte streaming reads (MI355X guide)",
               "source": f"gpurun_out/pmc_{tag}_fetch, pmc_{tag}_write"}, open("profiles/pmc_edge_traffic.json", "w"), indent=1)
txt = "\n".join(out)
print(txt)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(txt + "\n")
