#!/usr/bin/env python3
"""The counters tools/collect_wide_profiles.sh recorded for the first layer's product of the weight-streamed leg, per launch, as the JSON
bench.py quotes (profiles/rNN_wide_layer0_counters.json).   python tools/read_wide_pmc.py <dir with the csv files> <tag>
With a third argument D (3 or 1): the fused pass of the default network on 1024 features with D candidates instead
(rNN_widefused_pass<D>_*.csv -> rNN_widefused_pass<D>_counters.json)."""
import csv
import glob
import json
import os
import sys

d, tag = sys.argv[1], sys.argv[2]
KERNEL = "wide_gemm_kernel<8, 4, 2, 4"
GROUP = "wide"
LABEL = "wide_gemm_kernel<8,4,2,4,fp16-split>, 20k x 4096 x 256, 3 K-slices"
if len(sys.argv) > 3:
    KERNEL = "wide_gemm_kernel<2, 4, 8, 1, true, %s," % sys.argv[3]
    GROUP = "widefused_pass%s" % sys.argv[3]
    LABEL = "wide_gemm_kernel<2,4,8,1,fp16-split,D=%s> (fused pass), 100k x 1024, [50,5]" % sys.argv[3]


def mean_counter(path, name):
    vals = []
    with open(path, newline="") as fh:
        for row in csv.DictReader(fh):
            if KERNEL in row.get("Kernel_Name", "") and row.get("Counter_Name") == name:
                vals.append(float(row["Counter_Value"]))
    return sum(vals) / len(vals) if vals else None


out = {}
for path in sorted(glob.glob(os.path.join(d, "%s_%s_pmc_*.csv" % (tag, GROUP)))):
    with open(path, newline="") as fh:
        names = sorted({row.get("Counter_Name") for row in csv.DictReader(fh)} - {None})
    for name in names:
        v = mean_counter(path, name)
        if v is not None:
            out[name] = v
stats = os.path.join(d, "%s_%s_kernel_stats.csv" % (tag, GROUP))
dur_ns = None
if os.path.exists(stats):
    with open(stats, newline="") as fh:
        for row in csv.DictReader(fh):
            if KERNEL in row.get("Name", ""):
                dur_ns = float(row["AverageNs"])
res = {"kernel": LABEL, "kernel_ns_rocprof": dur_ns, "counters_per_launch": out,
       "units": "FETCH_SIZE / WRITE_SIZE in KiB; SQ counters cover 1/32 of the chip (8 CUs = 32 SIMDs) per dispatch: SQ_VALU_MFMA_BUSY_CYCLES in clock "
                "cycles, SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_BUSY_CYCLES in quad-cycles (chip-wide sums in this collection: see mfma_busy)"}
if "FETCH_SIZE" in out:
    # (gfx950: FETCH_SIZE tallies the 128-byte requests of wide streaming reads at 64 bytes - MI355X_MICROARCH.md, HBM section: doubled)
    res["traffic"] = 2.0 * out["FETCH_SIZE"] * 1024.0 + out.get("WRITE_SIZE", 0.0) * 1024.0
    res["traffic_note"] = "2 x FETCH_SIZE + WRITE_SIZE, bytes per launch (the guide's gfx950 correction for wide reads)"
if "SQ_VALU_MFMA_BUSY_CYCLES" in out and dur_ns:
    # (SQ_INSTS_MFMA of this collection is the WHOLE launch's count - 3 x rows x features x nodes / (16 x 16 x 32) - so the SQ counters
    # here are sums over the chip's 1024 SIMDs; GRBM_GUI_ACTIVE is the sum over the 8 XCDs of the cycles the launch took)
    per_simd = out["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0
    res["mfma_busy_cycles_per_simd"] = per_simd
    cycles = out["GRBM_GUI_ACTIVE"] / 8.0 if out.get("GRBM_GUI_ACTIVE") else dur_ns * 2.4
    res["kernel_cycles"] = cycles
    res["effective_clock_GHz"] = cycles / dur_ns
    res["mfma_busy"] = per_simd / cycles
    res["mfma_busy_note"] = "matrix-core busy cycles per SIMD / cycles of the launch (GRBM_GUI_ACTIVE / 8 XCDs)"
if out.get("SQ_WAVE_CYCLES"):
    res["wave_cycles_waiting"] = out.get("SQ_WAIT_ANY", 0.0) / out["SQ_WAVE_CYCLES"]
if "TCC_EA0_RDREQ_sum" in out and "TCC_EA0_RDREQ_DRAM_sum" in out and out["TCC_EA0_RDREQ_sum"]:
    res["read_requests_to_dram_share"] = out["TCC_EA0_RDREQ_DRAM_sum"] / out["TCC_EA0_RDREQ_sum"]
    res["read_requests_note"] = "TCC_EA0_RDREQ_DRAM counts every request that leaves the L2 for the memory side - the Infinity Cache sits behind that interface, so this counter does not tell its hits from HBM reads"
if "TCC_HIT_sum" in out and "TCC_MISS_sum" in out and (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]):
    res["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
print(json.dumps(res, indent=1))
