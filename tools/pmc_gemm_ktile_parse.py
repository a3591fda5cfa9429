"""Counters of tools/pmc_gemm_ktile.sh -> JSON: per-launch means for gemm256_kernel<2> (one launch = 32768 x 6144 x 1152,
3072 workgroups of 8 waves, 18 K-tiles each) and the derived per-wave, per-K-tile figures that say where a K-tile's
2545 cycles go.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* / SQ_INST_CYCLES_* count
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs; SQ_INST_LEVEL_* is the sum over
cycles of instructions in flight.   python tools/pmc_gemm_ktile_parse.py gpurun_out/gemm_ktile"""
import csv, glob, json, os, sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = sys.argv[1]
per = defaultdict(lambda: defaultdict(float))   # (file, dispatch) -> counter -> sum over instances
for path in glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(path)):
        if "gemm256_kernel" in r["Kernel_Name"]:
            per[(path, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
vals = defaultdict(list)
for (_p, _d), c in sorted(per.items(), key=lambda kv: (kv[0][0], int(kv[0][1]))):
    for k, v in c.items():
        vals[k].append(v)
c = {k: sum(v[len(v) // 4:]) / len(v[len(v) // 4:]) for k, v in vals.items()}  # skip warm-up launches
waves, ktiles = 3072 * 8, 18
wc = c["SQ_WAVE_CYCLES"]
frac = lambda k: c[k] / wc if k in c else None
d = {
    "wave_quad_cycles_per_wave": wc / waves,
    "wave_cycles_per_ktile_per_wave": 4 * wc / waves / ktiles,
    "share_of_wave_cycles": {
        "issuing (ACTIVE_INST_ANY)": frac("SQ_ACTIVE_INST_ANY"),
        "  of which VALU incl. MFMA issue": frac("SQ_ACTIVE_INST_VALU"),
        "  of which LDS": frac("SQ_ACTIVE_INST_LDS"),
        "  of which VMEM (LDS-DMA issue)": frac("SQ_ACTIVE_INST_VMEM"),
        "  of which scalar": frac("SQ_ACTIVE_INST_SCA"),
        "  of which misc (barrier, waitcnt, setprio)": frac("SQ_ACTIVE_INST_MISC"),
        "parked on s_waitcnt / s_barrier (WAIT_ANY)": frac("SQ_WAIT_ANY"),
        "issue stalled (WAIT_INST_ANY: MFMA dependency / pipe busy)": frac("SQ_WAIT_INST_ANY"),
        "  of which waiting to issue an LDS instruction": frac("SQ_WAIT_INST_LDS"),
    },
    "vmem_issue_quad_cycles_per_dma_piece": c.get("SQ_INST_CYCLES_VMEM_RD", 0) / c["SQ_INSTS_VMEM_RD"] if c.get("SQ_INSTS_VMEM_RD") else None,
    "dma_pieces_per_wave_per_ktile": c.get("SQ_INSTS_VMEM_RD", 0) / waves / ktiles,
    "mfma_per_wave_per_ktile": c.get("SQ_INSTS_MFMA", 0) / waves / ktiles,
    "valu_per_wave_per_ktile (incl. epilogue, spread over the K-tiles)": c.get("SQ_INSTS_VALU", 0) / waves / ktiles,
    "lds_insts_per_wave_per_ktile": c.get("SQ_INSTS_LDS", 0) / waves / ktiles,
    "mfma_busy_frac_at_held_clock": c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] * 128.0) if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c else None,
    "mfma_valu_coexec_frac_of_mfma_busy": c.get("SQ_VALU_MFMA_COEXEC_CYCLES", 0) / c["SQ_VALU_MFMA_BUSY_CYCLES"] if c.get("SQ_VALU_MFMA_BUSY_CYCLES") else None,
    "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT"),
    "lds_array_active_frac_of_kernel": c.get("SQ_LDS_IDX_ACTIVE", 0) / (c["SQ_CYCLES"]) if c.get("SQ_CYCLES") else None,
    "kernel_cycles_per_xcd": c.get("GRBM_GUI_ACTIVE", 0) / 8.0,
}
from esm_amd import _lib
print(json.dumps({"note": __doc__.split("\n\n")[0].replace("\n", " "), "build_id": _lib.lib().esmk_build_id().decode(),
                  "command": "bash tools/pmc_gemm_ktile.sh (rocprofv3 --pmc, four passes over tools/gemm_bench.py 12 ffn1_swiglu)",
                  "counters_per_launch": c, "derived": d}, indent=1))
