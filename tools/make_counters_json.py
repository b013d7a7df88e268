#!/usr/bin/env python3
"""profiles/rNN_counters.json from the rocprofv3 passes of tools/profile_round.sh: per render kernel, per
launch, the VALU wave-instructions (SQ_INSTS_VALU), the HBM bytes (WRITE_SIZE exact for streaming stores and
float atomics; FETCH_SIZE x 2 for wide coalesced reads on gfx950 -- MI355X_MICROARCH.md, section HBM), the
average duration of the kernel-trace pass, and the VALU issue peak bench.py prices `achieved_valu_fraction`
against (tools/ubench/issue_probe2.hip).  bench.py reads this file; its `traffic` is marked from_profile.
    python tools/make_counters_json.py gpurun_out/prof_r02 profiles/r02_counters.json [size spheres aa]
"""
import collections
import csv
import json
import os
import sys

src, dst = sys.argv[1], sys.argv[2]
size, spheres, aa = (int(x) for x in (sys.argv[3:6] if len(sys.argv) >= 6 else (4096, 32, 1)))


def find(d, name):
    for root, _, files in os.walk(d):
        if name in files:
            return os.path.join(root, name)
    return None


def counters(sub):
    path = find(os.path.join(src, sub), "bench_counter_collection.csv")
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if path:
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"]
            name = "k_fwd" if "k_fwd" in k else "k_bwd" if "k_bwd" in k else "k_prep" if "k_prep" in k else \
                "k_finalize" if "k_finalize" in k else None
            if name:
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items()}


sq1, sq2, fetch, write = counters("pmc_sq1"), counters("pmc_sq2"), counters("pmc_fetch"), counters("pmc_write")
stats = {}
st = find(os.path.join(src, "trace"), "bench_kernel_stats.csv")
if st:
    for r in csv.DictReader(open(st)):
        for name in ("k_fwd", "k_bwd", "k_prep", "k_finalize"):
            if name in r["Name"]:
                stats[name] = float(r["AverageNs"]) * 1e-3
out = {
    "source": "rocprofv3 --pmc, separate passes of `python bench.py --steps 30 --warmup 5 --cpu-rows 0` (%s)" % src,
    "workload": {"size": size, "spheres": spheres, "aa": aa},
    # v_fmac_f32 stream, two waves per SIMD, VGPR operands: 2.38 cycles per wave-instruction per SIMD
    # (tools/ubench/issue_probe2.hip, s_memtime cycles); in-kernel clock from tools/stamp_bwd.py
    "valu_peak_inst_per_cycle_per_simd": 1.0 / 2.38, "simds": 1024, "clock_ghz": 2.2,
}
for name in ("k_prep", "k_fwd", "k_bwd", "k_finalize"):
    f_raw = fetch.get(name, {}).get("FETCH_SIZE", 0.0) * 1024.0        # rocprofv3 reports KB
    w = write.get(name, {}).get("WRITE_SIZE", 0.0) * 1024.0
    # reads: k_bwd streams 16-B-per-lane coalesced rows (x2 correction); k_fwd / k_prep read a few scattered KB (raw)
    f = f_raw * 2.0 if name == "k_bwd" else f_raw
    out[name] = {"avg_us_kernel_trace": stats.get(name), "valu_insts": sq1.get(name, {}).get("SQ_INSTS_VALU"),
                 "salu_insts": sq1.get(name, {}).get("SQ_INSTS_SALU"),
                 "wave_quad_cycles": sq1.get(name, {}).get("SQ_WAVE_CYCLES"),
                 "wait_any_quad_cycles": sq2.get(name, {}).get("SQ_WAIT_ANY"),
                 "valu_active_quad_cycles": sq2.get(name, {}).get("SQ_ACTIVE_INST_VALU"),
                 "fetch_bytes_raw": f_raw, "write_bytes": w, "hbm_bytes": f + w}
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps(out, indent=1))
