"""Digest of a tools/collect_profiles.sh run -> the small files kept under profiles/ (run on either box)."""
import csv
import glob
import json
import os
import sys

ROUND = os.environ.get("CTO_ROUND", "round3")


def find(d, suffix):
    fs = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return fs[0] if fs else None


def main():
    out, tag = sys.argv[1], sys.argv[2]
    dst = os.path.join(out, "digest")
    os.makedirs(dst, exist_ok=True)
    for name in ("default", "short"):
        st = find(os.path.join(out, name), "kernel_stats.csv")
        if st:
            open(os.path.join(dst, ROUND + "_%s_%s_bench_kernel_stats.csv" % (tag, name)), "w").write(open(st).read())
        js = os.path.join(out, name + "_bench.json")
        if os.path.exists(js):
            lines = [l for l in open(js).read().split("\n") if l.startswith("{")]
            if lines:
                open(os.path.join(dst, ROUND + "_%s_%s_bench.json" % (tag, name)), "w").write(lines[-1] + "\n")
    # one step's launch timeline from the short run's kernel trace
    tr = find(os.path.join(out, "short"), "kernel_trace.csv")
    if tr:
        rows = list(csv.DictReader(open(tr)))
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        names = [r["Kernel_Name"] for r in rows]
        # a step = k_featurize_sites (tensor creation, one kernel since round 3) ... k_posterior with both networks in between; the last
        # such run of launches of the trace's timed region (the legs after it launch featurize / extraction kernels on their own)
        full = []
        for b in [i for i, n in enumerate(names) if "k_posterior" in n]:
            a = b
            while a > 0 and b - a < 16 and "k_featurize_sites" not in names[a]:
                a -= 1
            span = names[a:b + 1]
            if "k_featurize_sites" in names[a] and any("k_gru_layer" in n for n in span) and any("k_cvt_block" in n for n in span):
                full.append((a, b + 1))
        full = full[:max(1, len(full) // 2)]                 # the first half of them: warm-up + timed steps of the default model pair
        if full:
            a, b = full[-1]
            t0 = int(rows[a]["Start_Timestamp"])
            with open(os.path.join(dst, ROUND + "_%s_step_timeline.txt" % tag), "w") as f:
                f.write("# one step of bench.py (launch order): start us, duration us, kernel\n")
                for r in rows[a:b]:
                    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                    f.write("%9.1f %9.1f  %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:120]))
    # PMC passes
    kern = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        fn = find(os.path.join(out, "pmc_" + c), "counter_collection.csv")
        if not fn:
            continue
        acc = {}
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != c:
                continue
            acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            d = kern.setdefault(k, {})
            d[c + "_KB_mean_per_launch"] = round(sum(v) / len(v), 1)
            d["launches"] = len(v)
    if kern:
        note = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python bench.py --steps 3 "
                "--warmup 1 --no-cpu-baseline --no-e2e`, batch 4096; RAW counter values in KB.  WRITE_SIZE is exact on known byte counts "
                "(GRU layer 1 writes 135 168 KB); FETCH_SIZE reports HALF the bytes of coalesced reads on gfx950 (k_featurize_columns reads "
                "the 28.6 MB pack once and shows 14.8 MB): bench.py uses 2 x FETCH + WRITE, see profiles/README.md.")
        json.dump(dict(note=note, kernels=kern), open(os.path.join(dst, ROUND + "_%s_pmc_hbm_traffic.json" % tag), "w"), indent=1)
    # matrix-pipe occupancy per kernel
    vals = {}
    for d in glob.glob(os.path.join(out, "pmc_SQ_*")) + glob.glob(os.path.join(out, "pmc_GRBM*")):
        fn = find(d, "counter_collection.csv")
        if not fn or not os.path.isdir(d):
            continue
        for r in csv.DictReader(open(fn)):
            vals.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    mf = {}
    for k, v in vals.items():
        if "SQ_INSTS_MFMA" not in v or "SQ_WAVE_CYCLES" not in v:
            continue
        n_mfma = sum(v["SQ_INSTS_MFMA"]) / len(v["SQ_INSTS_MFMA"])
        wave_q = sum(v["SQ_WAVE_CYCLES"]) / len(v["SQ_WAVE_CYCLES"])
        if n_mfma <= 0:
            continue
        e = dict(SQ_INSTS_MFMA=round(n_mfma), SQ_WAVE_CYCLES_quad=round(wave_q),
                 mfma_issue_cycles_over_wave_cycles=round(n_mfma * 32.0 / (wave_q * 4.0), 4))
        for c in ("SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"):
            if c in v:
                e[c] = round(sum(v[c]) / len(v[c]))
        mf[k] = e
    if mf:
        note = ("rocprofv3 --kernel-trace --pmc passes (SQ_INSTS_MFMA SQ_WAVE_CYCLES; SQ_BUSY_CYCLES GRBM_GUI_ACTIVE) over `python bench.py "
                "--steps 3 --warmup 1 --no-cpu-baseline`.  v_mfma_f32_16x16x4_f32 occupies its SIMD's matrix pipe for 32 cycles; "
                "SQ_WAVE_CYCLES counts quad-cycles summed over waves, so mfma_issue_cycles_over_wave_cycles = 32 N_mfma / (4 SQ_WAVE_CYCLES) "
                "is the matrix-pipe occupancy seen by a wave (with two waves per SIMD, as in k_cvt_block, the pipe's own occupancy is "
                "up to twice that).")
        json.dump(dict(note=note, kernels=mf), open(os.path.join(dst, ROUND + "_%s_pmc_mfma.json" % tag), "w"), indent=1)
    print("digest:", sorted(os.listdir(dst)))


if __name__ == "__main__":
    main()
