"""MFMA utilisation per kernel from a rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):
   python tools/pmc_mfma_util.py <counter_collection.csv>
SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs, GRBM_GUI_ACTIVE over its 8 XCDs, so
   busy fraction of the MFMA pipes = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024)
at the clock the chip actually held (cycles, not time)."""
import csv
import json
import sys
from collections import defaultdict


def short(name):
    for key in ("gemm288", "gemm256_kernel<2>", "gemm256", "attn_varlen", "qk_prep", "layernorm", "gemm_kernel", "embed", "cast", "pack_weight"):
        if key in name:
            return key
    return name[:40]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    per = defaultdict(lambda: defaultdict(float))
    calls = defaultdict(int)
    for r in rows:
        k = short(r["Kernel_Name"])
        per[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            calls[k] += 1
    tot_busy = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for v in per.values())
    tot_act = sum(v.get("GRBM_GUI_ACTIVE", 0.0) for v in per.values())
    out = {"note": __doc__.split("\n")[0], "kernels": {}}
    for k, v in sorted(per.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0)):
        act, busy = v.get("GRBM_GUI_ACTIVE", 0.0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if act <= 0:
            continue
        out["kernels"][k] = {"calls": calls[k], "share_of_active_cycles": round(act / tot_act, 4),
                             "mfma_busy_frac": round(busy / (act * 128.0), 4)}
    out["whole_run_mfma_busy_frac"] = round(tot_busy / (tot_act * 128.0), 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
