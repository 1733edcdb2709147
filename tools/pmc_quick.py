#!/usr/bin/env python3
"""Per-kernel summary of the passes tools/pmc_quick.sh wrote: usage tools/pmc_quick.py OUTTAG [kernel-substring]"""
import collections, csv, glob, sys
tag = sys.argv[1]; want = sys.argv[2] if len(sys.argv) > 2 else "painn_edge_kernel"
C = collections.defaultdict(lambda: collections.defaultdict(list)); D = collections.defaultdict(list)
for d in glob.glob(f"gpurun_out/pq_{tag}_*/"):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ti::", "")
            C[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if d.rstrip("/").endswith("grbm"):
        for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                D[r["Kernel_Name"].split("(")[0].replace("void ti::", "")].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for k, d in sorted(C.items()):
    if want not in k or k not in D:
        continue
    m = {c: sum(v) / len(v) for c, v in d.items()}
    ms = sum(D[k]) / len(D[k]); clk = m.get("GRBM_GUI_ACTIVE", 0) / 8 / (ms * 1e-3) / 1e9
    simd = 1024 * ms * 1e-3 * clk * 1e9; wc = m.get("SQ_WAVE_CYCLES", 1); nm = max(m.get("SQ_INSTS_MFMA", 1), 1)
    g = lambda c: m.get(c, float("nan"))
    print(f"== {k[:90]}  launches {len(D[k])}  {ms:.3f} ms  clock {clk:.2f} GHz")
    print(f"   MFMA busy / SIMD-time {g('SQ_VALU_MFMA_BUSY_CYCLES') / simd:.3f}   MFMA+VALU co-exec / SIMD-time {g('SQ_VALU_MFMA_COEXEC_CYCLES') / simd:.3f}")
    print(f"   wave time: issue-stall {g('SQ_WAIT_INST_ANY') / wc:.2f}  parked {g('SQ_WAIT_ANY') / wc:.2f}  issuing {g('SQ_ACTIVE_INST_ANY') / wc:.2f}   (LDS issue-stall {g('SQ_WAIT_INST_LDS') / wc:.2f})")
    print(f"   active-inst share of wave time: VALU {g('SQ_ACTIVE_INST_VALU') / wc:.3f}  LDS {g('SQ_ACTIVE_INST_LDS') / wc:.3f}  VMEM {g('SQ_ACTIVE_INST_VMEM') / wc:.3f}  SALU {g('SQ_ACTIVE_INST_SCA') / wc:.3f}  MISC {g('SQ_ACTIVE_INST_MISC') / wc:.3f}")
    print(f"   per MFMA: VALU(incl MFMA) {g('SQ_INSTS_VALU') / nm:.2f}  trans {g('SQ_INSTS_VALU_TRANS') / nm:.3f}  LDS {g('SQ_INSTS_LDS') / nm:.2f}  SALU {g('SQ_INSTS_SALU') / nm:.2f}  VMEM {g('SQ_INSTS_VMEM') / nm:.3f}")
    print(f"   LDS: bank-conflict cycles / idx-active {g('SQ_LDS_BANK_CONFLICT') / max(g('SQ_LDS_IDX_ACTIVE'), 1):.3f}   idx-active / SIMD-time(x4 SIMDs share one LDS) {g('SQ_LDS_IDX_ACTIVE') / (simd / 4):.3f}")
    if "SQC_ICACHE_REQ" in m:
        print(f"   I-cache: req {g('SQC_ICACHE_REQ'):.3g}  hit rate {g('SQC_ICACHE_HITS') / max(g('SQC_ICACHE_REQ'), 1):.4f}  misses {g('SQC_ICACHE_MISSES'):.3g} (+dup {g('SQC_ICACHE_MISSES_DUPLICATE'):.3g})")
    if "SQ_IFETCH" in m:
        print(f"   ifetch: {g('SQ_IFETCH'):.3g} fetches, level-accum / fetch {g('SQ_IFETCH_LEVEL') / max(g('SQ_IFETCH'), 1):.1f};  ifetch level-accum / wave-cycles {g('SQ_IFETCH_LEVEL') / wc:.3f}   branches/MFMA {g('SQ_INSTS_BRANCH') / nm:.3f}")
    if "SQ_INST_LEVEL_LDS" in m:
        print(f"   latency (level-accum / inst): LDS {g('SQ_INST_LEVEL_LDS') / max(g('SQ_INSTS_LDS'), 1):.1f}  VMEM {g('SQ_INST_LEVEL_VMEM') / max(g('SQ_INSTS_VMEM'), 1):.1f}   fifo-full: LDS cmd {g('SQ_LDS_CMD_FIFO_FULL') / wc:.4f} data {g('SQ_LDS_DATA_FIFO_FULL') / wc:.4f}  TA addr {g('SQ_VMEM_TA_ADDR_FIFO_FULL') / wc:.4f} cmd {g('SQ_VMEM_TA_CMD_FIFO_FULL') / wc:.4f} wr-data {g('SQ_VMEM_WR_TA_DATA_FIFO_FULL') / wc:.4f}")
    if "FETCH_SIZE" in m:
        print(f"   HBM per launch: read {2 * m['FETCH_SIZE'] * 1024 / 1e9:.2f} GB (2 x FETCH_SIZE)  write {m.get('WRITE_SIZE', 0) * 1024 / 1e9:.2f} GB")
