"""rocprofv3 --pmc passes over bench.py (tools/collect_profiles.sh) -> one JSON with per-kernel means per launch:
HBM-side bytes (FETCH_SIZE x 2 [gfx950: 16-B/lane streaming reads are tallied at half size, MI355X_MICROARCH.md HBM
section] + WRITE_SIZE, both reported in KiB), L2 hit rate, MFMA-busy fraction at the held clock.  The JSON carries the
build id of the library that ran, so bench.py can refuse figures that belong to another binary.
   python tools/pmc_bench_parse.py gpurun_out/prof"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    for key, lab in (("gemm256_kernel<2>", "gemm256<SWIGLU> (FFN up-projection)"), ("gemm256_kernel", "gemm256 (other epilogues)"),
                     ("gemm288_kernel<3, true>", "gemm288 + q statistics (QKV projection)"),
                     ("gemm288_kernel<2, true>", "gemm288 + q statistics (QKV projection)"),
                     ("gemm288", "gemm288 (out-proj / FFN down-projection; QKV where not fused)"),
                     ("attn_varlen_kernel<true>", "attn_varlen, q-fused"), ("attn_varlen_kernelILb1E", "attn_varlen, q-fused"),
                     ("attn_varlen", "attn_varlen"), ("qk_prep", "qk_prep"),
                     ("k_prep_kernel", "k_prep (k-only QK-LayerNorm + rotary, q statistics)"), ("layernorm_kernel", "add+layernorm"), ("gemm_kernel", "gemm 128-tile"),
                     ("splitk_reduce", "split-K reduce")):
        if key in name:
            return lab
    return None


def collect(root):
    per = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(os.path.join(root, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k:
                per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return per


def main():
    root = sys.argv[1]
    from esm_amd import _lib
    per = collect(root)
    out = {"note": __doc__.split("\n\n")[0].replace("\n", " "), "build_id": _lib.lib().esmk_build_id().decode(), "config": "c2",
           "command": "rocprofv3 --pmc <one counter group per pass> -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline",
           "kernels": {}}
    mean = lambda v: sum(v) / len(v) if v else None
    for k, c in per.items():
        f, w = mean(c.get("FETCH_SIZE")), mean(c.get("WRITE_SIZE"))
        hit, miss = mean(c.get("TCC_HIT_sum")), mean(c.get("TCC_MISS_sum"))
        busy, act = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES")), mean(c.get("GRBM_GUI_ACTIVE"))
        e = {"launches": len(c.get("GRBM_GUI_ACTIVE", []))}
        if f is not None and w is not None:
            e["hbm_bytes_per_launch"] = 2 * f * 1024 + w * 1024
            e["fetch_kib_raw"], e["write_kib_raw"] = f, w
        if hit is not None and miss is not None and hit + miss > 0:
            e["l2_hit_rate"] = hit / (hit + miss)
        if busy is not None and act:
            e["mfma_busy_frac"] = busy / (act * 128.0)  # busy summed over 1024 SIMDs, GUI_ACTIVE over 8 XCDs
            e["cycles_per_launch"] = act / 8.0
        out["kernels"][k] = e
    out["dominant_kernel"] = out["kernels"].get("gemm256<SWIGLU> (FFN up-projection)", {})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
