"""Reduce the rocprofv3 CSVs of scratch/round_profiles.sh to profiles/<tag>_counters.json (read by bench.py).

usage: python scratch/summarize_profiles.py <dir with the CSVs> [tag]

Per precision and per edge-kernel variant (GCL / coordinate): launches, average duration (kernel-stats run), HBM
bytes per launch from the FETCH_SIZE / WRITE_SIZE passes with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads: doubled;
both are in KiB), and the SQ activity fractions:
    (bf16x6: six bf16 MFMAs per product, precision number 2 in the kernel names)
    mfma_busy       = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)     matrix-pipe busy cycles per SIMD-cycle of the
                      kernel: the numerator is summed over the chip's 1024 SIMDs (it equals 64 x #MFMA for
                      v_mfma_f32_32x32x2_f32 and 32 x #MFMA for v_mfma_f32_32x32x16_bf16 - checked against the launch
                      geometry), SQ_BUSY_CYCLES is summed over the 32 shader engines, so SQ_BUSY_CYCLES / 32 = kernel
                      duration in shader cycles and 1024 / 32 = 32.
    clock_ghz       = (SQ_BUSY_CYCLES / 32) / duration of the same dispatch (kernel trace of the SQ pass): the chip
                      clocks to its power budget (fp32 kernel ~2.3 GHz, bf16x3 ~2.0 GHz under the profiler)
    SQ_INSTS_VALU / SQ_WAVE_CYCLES is reported as VALU instructions per wave quad-cycle
    wait_inst_frac  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES                    issue stalls (quad-cycles / quad-cycles)
    wait_any_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES                         parked in s_waitcnt / barrier
The edge-kernel mix of one forward is 2 GCL : 1 coordinate launch; the per-launch figures bench.py uses are that mix.
"""
import collections
import csv
import glob
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sys


def per_kernel(path):
    rows = collections.defaultdict(lambda: collections.defaultdict(list))
    if not os.path.exists(path):
        return rows
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return rows


def mean(v):
    return sum(v) / len(v) if v else None


def main():
    d = sys.argv[1]
    tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
    out = {"source": f"rocprofv3 passes of scratch/round_profiles.sh (raw CSVs: profiles/{tag}_*_pmc_*.csv, profiles/{tag}_*_sq.csv, "
                     f"profiles/{tag}_*_T50_kernel_stats.csv); bench.py --timesteps 3 (counters) / 50 (kernel stats), B=256 N=30 H=256 L=6",
           "shape": [256, 30, 256, 6],
           "kernel_source_sha256": __import__("bench").kernel_source_sha256(),
           # the library the passes ran on: bench.py replays these figures only while this very file is loaded
           "lib_sha256": __import__("hashlib").sha256(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                      "hierdiff_amd", "lib", "libhierdiff_hip.so"), "rb").read()).hexdigest(),
           "correction": "hbm bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64 B per 128-B request); "
                         "counters include Infinity-Cache hits",
           "edge_kernel": {}, "kernels": {}}
    for prec, pnum in (("fp32", 0), ("fp16x3", 3)):          # (rounds 2-5 also carried bf16x3 = 1 and bf16x6 = 2)
        stats = {}
        sp = os.path.join(d, f"{tag}_{prec}_T50_kernel_stats.csv")
        if os.path.exists(sp):
            for r in csv.DictReader(open(sp)):
                stats[r["Name"].split("(")[0].replace("void ", "").strip()] = (int(r["Calls"]), float(r["AverageNs"]) / 1e3,
                                                                              float(r["Percentage"]))
        fetch = per_kernel(os.path.join(d, f"{tag}_{prec}_pmc_FETCH_SIZE.csv"))
        write = per_kernel(os.path.join(d, f"{tag}_{prec}_pmc_WRITE_SIZE.csv"))
        sq = per_kernel(os.path.join(d, f"{tag}_{prec}_sq.csv"))
        sqdur = {}
        tp = os.path.join(d, f"{tag}_{prec}_sq_kernel_trace.csv")
        if os.path.exists(tp):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(tp)):
                acc[r["Kernel_Name"].split("(")[0].replace("void ", "").strip()].append(
                    (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
            sqdur = {k: mean(v) for k, v in acc.items()}
        names = sorted(set(stats) | set(fetch) | set(sq))
        kern = {}
        for k in names:
            e = {}
            if k in stats:
                e["calls"], e["avg_us"], e["pct_of_gpu_time"] = stats[k][0], round(stats[k][1], 2), stats[k][2]
            f, w = mean(fetch.get(k, {}).get("FETCH_SIZE", [])), mean(write.get(k, {}).get("WRITE_SIZE", []))
            if f is not None and w is not None:
                e["FETCH_SIZE_KiB"], e["WRITE_SIZE_KiB"] = round(f, 1), round(w, 1)
                e["hbm_bytes_per_launch"] = int((2 * f + w) * 1024)
            c = {n: mean(v) for n, v in sq.get(k, {}).items()}
            if c:
                e["sq"] = {n: round(v, 1) for n, v in c.items()}
                if c.get("SQ_BUSY_CYCLES"):
                    e["mfma_busy"] = round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (32.0 * c["SQ_BUSY_CYCLES"]), 4)
                    if k in sqdur:
                        e["sq_pass_avg_us"] = round(sqdur[k], 2)
                        e["clock_ghz"] = round(c["SQ_BUSY_CYCLES"] / 32.0 / (sqdur[k] * 1e3), 3)
                if c.get("SQ_WAVE_CYCLES"):
                    e["wait_inst_frac"] = round(c.get("SQ_WAIT_INST_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
                    e["wait_any_frac"] = round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4)
                    e["valu_insts_per_wave_quadcycle"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_WAVE_CYCLES"], 4)
            kern[k] = e
        out["kernels"][prec] = kern
        gcl = next((v for k, v in kern.items() if k.startswith(f"k_edge<256, false, {pnum}")), None)
        crd = next((v for k, v in kern.items() if k.startswith(f"k_edge<256, true, {pnum}")), None)
        if gcl and crd:
            mix = {}
            for key in ("hbm_bytes_per_launch", "mfma_busy", "wait_inst_frac", "wait_any_frac", "avg_us", "clock_ghz"):
                if key in gcl and key in crd:
                    v = (2 * gcl[key] + crd[key]) / 3.0
                    mix[key] = int(v) if key == "hbm_bytes_per_launch" else round(v, 4)
            out["edge_kernel"][prec] = mix
    path = os.path.join(d, f"{tag}_counters.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out["edge_kernel"], indent=1))
    for prec, kern in out["kernels"].items():
        print(f"---- {prec}")
        for k, e in sorted(kern.items(), key=lambda kv: -kv[1].get("pct_of_gpu_time", 0)):
            print(f"{k[:56]:56s} calls {e.get('calls', 0):6d} avg {e.get('avg_us', 0):8.1f} us {e.get('pct_of_gpu_time', 0):5.1f}%  "
                  f"hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:7.2f} MB  mfma_busy {e.get('mfma_busy', '-')}")
    print("wrote", path)


if __name__ == "__main__":
    main()
