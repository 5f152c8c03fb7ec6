#!/usr/bin/env python
"""Matrix-core utilisation of the FFDNet convolution kernels from separate rocprofv3 --pmc passes (tools/profile_ffdnet_r3.sh).

The SQ counters are sampled (a fraction of the chip's waves) and their cycle units are not documented for gfx950, so the figure is
self-calibrating: the same pass holds SQ_INSTS_MFMA and SQ_VALU_MFMA_BUSY_CYCLES, and their ratio is the busy time one matrix
instruction is charged (32 cycles per v_mfma_f32_32x32x16_{f16,bf16} when the counter ticks per SIMD cycle).  The utilisation is

    mfma_busy = (algorithmic MFMA count x cycles charged per MFMA) / (kernel duration x clock x 1024 SIMDs)

with the clock taken from GRBM_GUI_ACTIVE / duration of the same launch, the MFMA count from the layer shapes (the sampled
SQ_INSTS_MFMA / SQ_WAVES x the launch's wave count must agree: `mfma_per_wave_sampled`) and the duration from the kernel trace."""
import csv, json, sys
pmc, stats_csv, out = sys.argv[1:4]
d = json.load(open(pmc))
dur = {}
for row in csv.DictReader(open(stats_csv)):
    name = row.get("Name", row.get("KernelName", ""))
    dur[name.replace("void ", "").replace("dpx::", "").split("(")[0]] = float(row.get("AverageNs", row.get("Average", 0.0)))
res = {}
for k, e in d.items():
    if "conv3x3" not in k:
        continue
    c = e["counters"]
    r = {"launches_sampled": e["launches_sampled"], "counters_per_launch": {n: round(v, 1) for n, v in sorted(c.items())}}
    ns = dur.get(k)
    if ns:
        r["avg_duration_us_kernel_trace"] = ns / 1e3
    if c.get("SQ_INSTS_MFMA") and c.get("SQ_VALU_MFMA_BUSY_CYCLES"):
        r["busy_cycles_charged_per_mfma"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_INSTS_MFMA"]
    if c.get("SQ_INSTS_MFMA") and c.get("SQ_WAVES"):
        r["mfma_per_wave_sampled"] = c["SQ_INSTS_MFMA"] / c["SQ_WAVES"]
    if c.get("GRBM_GUI_ACTIVE") and ns:
        r["effective_clock_GHz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / ns       # (the counter is summed over the 8 XCDs)
    if c.get("SQ_WAVE_CYCLES") and c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("SQ_WAVES"):
        # per sampled wave: matrix-pipe busy cycles / wave lifetime (SQ_WAVE_CYCLES ticks in quad-cycles: x4)
        r["mfma_busy_over_wave_lifetime"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * c["SQ_WAVE_CYCLES"])
        r["note_wave_lifetime"] = ("matrix-pipe busy cycles charged to the sampled waves / their summed lifetimes; with W waves resident "
                                   "per SIMD the pipe's utilisation is W x this figure (k_conv3x3_bf16: 8 waves per workgroup, one "
                                   "workgroup per CU = 2 waves per SIMD)")
        r["mfma_busy_frac_of_simd_cycles"] = 2.0 * r["mfma_busy_over_wave_lifetime"] if "bf16" in k else None
    for n in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if c.get(n) and c.get("SQ_WAVE_CYCLES"):
            r[n.lower() + "_share_of_wave_cycles"] = c[n] / c["SQ_WAVE_CYCLES"]
    if "hbm_traffic_bytes" in e:
        r["hbm_read_bytes_corrected"], r["hbm_write_bytes"] = e["hbm_read_bytes_corrected"], e["hbm_write_bytes"]
    res[k] = r
json.dump(res, open(out, "w"), indent=1, sort_keys=True)
for k, r in sorted(res.items()):
    print(k[:100], {a: (round(b, 3) if isinstance(b, float) else b) for a, b in r.items() if a not in ("counters_per_launch", "note_wave_lifetime")})
