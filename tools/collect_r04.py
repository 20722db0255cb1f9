#!/usr/bin/env python3
"""collect_r04.py <dir> — condense the rocprofv3 outputs of tools/profile_r04.sh into the small files kept under profiles/."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]
S = os.path.join(O, "summary"); os.makedirs(S, exist_ok=True)

def first(pat):
    g = sorted(glob.glob(os.path.join(O, pat), recursive=True))
    return g[0] if g else None

def kernel_stats(tag, out):
    f = first(f"{tag}/**/*kernel_stats.csv")
    if not f: return None
    rows = list(csv.DictReader(open(f)))
    with open(os.path.join(S, out), "w") as w:
        wr = csv.writer(w); wr.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            wr.writerow([r.get("Name"), r.get("Calls"), r.get("TotalDurationNs"), r.get("AverageNs"), r.get("Percentage"), r.get("MinNs"), r.get("MaxNs")])
    return rows

def counters(tag, match):
    f = first(f"{tag}/**/*counter_collection.csv")
    if not f: return {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if match not in k: continue
        k = k.split("(")[0][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); disp[k].add(r.get("Dispatch_Id"))
    return {k: dict(v, dispatches=len(disp[k])) for k, v in agg.items()}

kernel_stats("c3_stats", "r04_c3_kernel_stats.csv")
kernel_stats("c4_stats", "r04_c4_gram_kernel_stats.csv")
kernel_stats("c5_stats", "r04_c5_kernel_stats.csv")
sq = counters("c3_sq", "sweep_blk"); sq2 = counters("c3_sq2", "sweep_blk")
merged = {}
for d in (sq, sq2):
    for k, v in d.items(): merged.setdefault(k, {}).update(v)
for k, v in merged.items():
    wc = v.get("SQ_WAVE_CYCLES")
    if wc:
        v["derived"] = {"valu_active_frac_of_wave_cycles": v.get("SQ_ACTIVE_INST_VALU", 0) / wc, "wait_any_frac": v.get("SQ_WAIT_ANY", 0) / wc,
                        "wait_inst_any_frac": v.get("SQ_WAIT_INST_ANY", 0) / wc}
json.dump({"command": "rocprofv3 --pmc <set> -- python3 bench.py --config C3 --steps 1 --warmup 0 --no-cpu-baseline (two passes, tools/profile_r04.sh)",
           "notes": "sums over the dispatches of the kernel in the run (the 2^20-pattern sweep; the one-node winner re-solve is the <16,true> instantiation)",
           "kernels": merged}, open(os.path.join(S, "r04_c3_pmc_blk.json"), "w"), indent=1)
fe = counters("c3_fetch", "sweep_blk"); wr = counters("c3_write", "sweep_blk")
fk = sum(v.get("FETCH_SIZE", 0) for k, v in fe.items() if "Lb0" in k or "false" in k or True)
wk = sum(v.get("WRITE_SIZE", 0) for v in wr.values())
json.dump({"kernel": "sweep_blk_kernel<16, false> (+ the one-node re-solve)", "config": "C3", "fetch_size_kib": fk, "write_size_kib": wk,
           "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/profile_r04.sh); FETCH_SIZE doubled per "
                     "MI355X_MICROARCH.md §HBM (gfx950 reports 1/2 of streamed bytes)", "per_kernel": {"fetch": fe, "write": wr}},
          open(os.path.join(S, "r04_sweep_traffic.json"), "w"), indent=1)
g = counters("c4_pmc", "gram_kernel")
for k, v in g.items():
    if v.get("SQ_INSTS_VALU_MFMA_F64") and v.get("GRBM_GUI_ACTIVE"):
        n = v["dispatches"]
        kernel_cycles = v["GRBM_GUI_ACTIVE"] / 8.0                      # the counter is summed over the 8 XCDs
        v["derived"] = {"mfma_cycles_per_instruction": v["SQ_VALU_MFMA_BUSY_CYCLES"] / v["SQ_INSTS_VALU_MFMA_F64"],
                        "kernel_cycles_all_dispatches": kernel_cycles,
                        "mfma_utilisation": v["SQ_VALU_MFMA_BUSY_CYCLES"] / (kernel_cycles * 1024),   # 256 CUs x 4 SIMDs
                        "executed_flops_per_dispatch": v["SQ_INSTS_VALU_MFMA_F64"] / n * 2048}
json.dump({"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES -- python3 tools/gram_bench.py 1000000 512 16 3",
           "kernels": g}, open(os.path.join(S, "r04_c4_gram_pmc.json"), "w"), indent=1)
# beyond the register kernel: the deferred-update kernel's traffic per 2^18-pattern sweep (the script sweeps twice; the short calibration launch is its own instantiation)
ls = kernel_stats("l340_stats", "r04_d340_lazy_kernel_stats.csv")
lf, lw = counters("l340_fetch", "sweep_lazy_kernel<512, false>"), counters("l340_write", "sweep_lazy_kernel<512, false>")
if lf and lw:
    fk_ = sum(v.get("FETCH_SIZE", 0) for v in lf.values()); wk_ = sum(v.get("WRITE_SIZE", 0) for v in lw.values())
    n_ = max(v.get("dispatches", 1) for v in lf.values())
    avg = None
    for r in ls or []:
        if "sweep_lazy_kernel<512, false>" in r.get("Name", ""): avg = float(r["AverageNs"]) / 1e6
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/generic_timing.py 20000 340 18  (D = 340, K = 18: 2^18 patterns; "
                          "tools/profile_r04.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM)",
               "deferred_update_kernel": {"fetch_size_kib": fk_, "write_size_kib": wk_, "dispatches": n_, "hbm_bytes_per_sweep": (2 * fk_ + wk_) * 1024 / n_, "avg_ms": avg}},
              open(os.path.join(S, "r04_d340_traffic.json"), "w"), indent=1)
import shutil
for c in ("c2", "c3", "c4", "c5", "l340"):
    f = os.path.join(O, f"bench_{c}.json")
    if os.path.exists(f) and os.path.getsize(f) > 0:
        shutil.copy(f, os.path.join(S, f"r04_bench_{c}.json"))
print("summary in", S)
