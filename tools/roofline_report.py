"""Turn the output of tools/refresh_evidence.sh into the committed evidence of a round.
Usage: python tools/roofline_report.py gpurun_out/<tag> profiles r2
Writes profiles/<round>_roofline.md (one row per kernel / config: rocprofv3 average duration, algorithmic flop and bytes,
fraction of the binding peak, HBM traffic from the PMC passes over the algorithmic bytes), profiles/<round>_traffic.json (read by
bench.py), and copies the rocprofv3 summaries it used (kernel stats, per-kernel counters of the last dispatch)."""
import collections, csv, glob, json, os, shutil, sys

src, dst, rnd = sys.argv[1], sys.argv[2], sys.argv[3]
PEAKS = {"i8": (5033.0, "TOP/s int8 matrix (dense)"), "fp32": (157.3, "TFLOP/s f32 (vector = matrix)"), "fp64_mfma": (78.6, "TFLOP/s f64 matrix"), "hbm": (8000.0, "GB/s HBM")}
os.makedirs(dst, exist_ok=True)


def stats(name):
    """kernel name fragment -> (calls, average ns) from rocprofv3 --stats of run `name`"""
    out = {}
    for f in glob.glob(os.path.join(src, "prof", "**", f"{name}_kernel_stats.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            out[row["Name"]] = (int(row["Calls"]), float(row["AverageNs"]), float(row["Percentage"]))
        shutil.copy(f, os.path.join(dst, f"{rnd}_{name}_kernel_stats.csv"))
    return out


def counters(name, fragment):
    """counters of the LAST dispatch of the kernel whose name contains `fragment`, over the passes of run `name`"""
    best = {}
    for f in sorted(glob.glob(os.path.join(src, "pmc", "**", f"{name}_p*_counter_collection.csv"), recursive=True)):
        rows = collections.defaultdict(dict)
        for row in csv.DictReader(open(f)):
            if fragment not in row["Kernel_Name"]: continue
            d = rows[int(row["Dispatch_Id"])]
            d[row["Counter_Name"]] = d.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            d["_ns"] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"]); d["_vgpr"] = row["VGPR_Count"]; d["_lds"] = row["LDS_Block_Size"]
            d["_scratch"] = row["Scratch_Size"]; d["_grid"] = row["Grid_Size"]
        if rows:
            # a call that takes several launches of the kernel (ring epochs: 7 full grids and a remainder) is one "launch" of the table:
            # the last dispatch and those before it back to the previous dispatch of the SAME grid are summed (equal grids throughout: one dispatch)
            ids = sorted(rows)
            last = ids[-1]; span = [last]
            for i in reversed(ids[:-1]):
                if rows[i]["_grid"] == rows[last]["_grid"]: break
                span.append(i)
            if len(span) == len(ids): span = [last]              # (no earlier dispatch of that grid: not a pattern)
            for k in rows[last]:
                if k.startswith("_"): best[k] = rows[last][k] if k != "_ns" else sum(rows[i]["_ns"] for i in span)
                else: best[k] = sum(rows[i].get(k, 0.0) for i in span)
            best["_dispatches"] = len(span)
    return best


def find(st, fragment):
    hits = [(k, v) for k, v in st.items() if fragment in k]
    return max(hits, key=lambda kv: kv[1][0] * kv[1][1]) if hits else (None, None)


rows, summary = [], []
bench = json.load(open(os.path.join(src, "bench100.json")))
bench20 = json.load(open(os.path.join(src, "bench.json")))
for f in ("bench100.json", "bench.json"):
    shutil.copy(os.path.join(src, f), os.path.join(dst, f"{rnd}_final_{f}"))

# ---- headline
st = stats("headline")
fixed = str(bench.get("dtype", "")).startswith("i8")
frag = "fir_i8_" if fixed else "fir_mfma"
kname, (calls, avg_ns, pct) = find(st, frag)
c = counters("headline", kname)          # (the headline kernel by its full name: bench.py also runs config D — 32 channels — on the same kernel template)
spl = bench["roofline"]["algorithmic_bytes_per_launch"] / bench["roofline"]["bytes_per_sample"]
exec_flop = bench["roofline"]["flop_per_sample_executed"]
tf = spl * exec_flop / (avg_ns * 1e-9) / 1e12
traffic = None
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    traffic = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)          # KB -> bytes; gfx950 wide-read correction on the fetch side
rows.append(("headline: 8 ch 44.1k->48k, -4 (988x988 interp), 1M-frame calls", kname, calls, avg_ns, f"{exec_flop} executed (4T+3 = 3955 in the reference formulation)",
             bench["roofline"]["bytes_per_sample"], (f"{tf:.0f} TOP/s = {tf / 5033.0:.3f} of the dense int8 matrix peak (the same samples/s as 2 x Kpad f32 flop: {spl * 2048 / (avg_ns * 1e-9) / 1e12 / 157.3:.3f} of the f32 matrix peak)" if fixed
              else f"{tf:.1f} TFLOP/s = {tf / 157.3:.3f} of f32 matrix peak"), traffic, bench["roofline"]["algorithmic_bytes_per_launch"]))
tj = {"kernel": kname, "fixed_point": fixed, "workload": {"block_frames": bench["config"]["block_frames"], "channels": bench["config"]["channels_per_gpu"], "taps": 988, "filters": 988, "src": 44100, "dst": 48000},
      "FETCH_SIZE_KB": c.get("FETCH_SIZE"), "WRITE_SIZE_KB": c.get("WRITE_SIZE"), "fetch_correction": 2.0,
      "note": "gfx950 rocprofv3 FETCH_SIZE counts 64 B per 128-B request on wide coalesced reads (MI355X_MICROARCH.md, HBM): doubled; WRITE_SIZE as reported",
      "traffic_bytes_per_launch": traffic, "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
      "TCC_HIT_sum": c.get("TCC_HIT_sum"), "TCC_MISS_sum": c.get("TCC_MISS_sum"), "SQ_VALU_MFMA_BUSY_CYCLES": c.get("SQ_VALU_MFMA_BUSY_CYCLES"),
      "SQ_BUSY_CYCLES": c.get("SQ_BUSY_CYCLES"), "GRBM_GUI_ACTIVE": c.get("GRBM_GUI_ACTIVE"), "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_INSTS_MFMA": c.get("SQ_INSTS_MFMA"),
      "rocprofv3_avg_ns": avg_ns, "bench_hip_event_avg_ms": bench["roofline"]["avg_kernel_ms"]}
if c.get("SQ_VALU_MFMA_BUSY_CYCLES") and c.get("GRBM_GUI_ACTIVE"):
    # MFMA busy cycles are summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs
    tj["mfma_pipe_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / (c["GRBM_GUI_ACTIVE"] / 8), 3)
json.dump(tj, open(os.path.join(dst, f"{rnd}_traffic.json"), "w"), indent=1)
summary.append(("headline", kname, c))
for frag2, what in (("mfma_prepare", "its prepare launch"), ("i8_stage_kernel<true, true>", "its peak pass: per-slice peaks of history ++ input + the rows' digit planes"),
                    ("i8_stage_kernel<true, false>", "its quantise pass: block exponents, digit planes of history ++ input"), ("fir_mfma_stream", "the f32 kernel of the same run (bench.py's value_f32 leg)")):
    pk, v = find(st, frag2)
    if pk and pk != kname: rows.append((f"  ({what})", pk, v[0], v[1], "-", "-", "-", None, None))
# the f32 streaming kernel on the same workload (bench.py --kernel 6)
p6 = os.path.join(src, "bench_f32.json")
if os.path.exists(p6):
    b6 = json.load(open(p6)); shutil.copy(p6, os.path.join(dst, f"{rnd}_final_bench_f32.json"))
    st6 = stats("headline_f32"); k6, v6 = find(st6, "fir_mfma_stream")
    if k6:
        c6 = counters("headline_f32", k6)
        tf6 = spl * b6["roofline"]["flop_per_sample_executed"] / (v6[1] * 1e-9) / 1e12
        tr6 = int(c6["FETCH_SIZE"] * 1024 * 2 + c6["WRITE_SIZE"] * 1024) if ("FETCH_SIZE" in c6 and "WRITE_SIZE" in c6) else None
        rows.append((f"headline workload on the f32 streaming kernel (--kernel 6): {b6['value']} Msamples/s", k6, v6[0], v6[1], b6["roofline"]["flop_per_sample_executed"],
                     b6["roofline"]["bytes_per_sample"], f"{tf6:.1f} TFLOP/s = {tf6 / 157.3:.3f} of f32 matrix peak", tr6, b6["roofline"]["algorithmic_bytes_per_launch"]))
        summary.append(("headline_f32", k6, c6))

# ---- the other kernels
for case in ("fixed_D4", "fixed_D32", "matrix_B", "matrix_D4", "matrix_D32", "general_E", "general_P", "matrix_P", "general_A", "strict", "wide", "biquad", "biquad_serial", "decimate"):
    p = os.path.join(src, f"case_{case}.json")
    if not os.path.exists(p): continue
    try: info = json.loads(open(p).read().strip().splitlines()[-1])
    except Exception: continue
    st = stats(case)
    kname, v = find(st, info["kernel"])
    if not kname: continue
    calls, avg_ns, pct = v
    c = counters(case, info["kernel"])
    n = info["samples_per_launch"]
    peak, punit = PEAKS[info["peak"]]
    if info["peak"] == "hbm":
        ach = n * info["bytes_per_sample"] / (avg_ns * 1e-9) / 1e9; frac = f"{ach:.1f} GB/s = {ach / peak:.4f} of HBM peak"
    else:
        ach = n * info["flop_per_sample"] / (avg_ns * 1e-9) / 1e12; frac = f"{ach:.2f} {'TOP/s' if info['peak'] == 'i8' else 'TFLOP/s'} = {ach / peak:.3f} of {punit}"
    traffic = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024) if ("FETCH_SIZE" in c and "WRITE_SIZE" in c) else None
    rows.append((f"{case}: {info['Msamples_per_s']} Msamples/s end to end", kname, calls, avg_ns, info["flop_per_sample"], round(info["bytes_per_sample"], 3), frac, traffic,
                 int(n * info["bytes_per_sample"])))
    summary.append((case, kname, c))
    for k2, v2 in st.items():        # companions of the same call (commit / copy / prepare kernels)
        if k2 != kname and v2[2] >= 3.0 and "rocclr" not in k2:
            rows.append((f"  ({case}: other kernel, {v2[2]:.0f} % of GPU time)", k2, v2[0], v2[1], "-", "-", "-", None, None))

with open(os.path.join(dst, f"{rnd}_roofline.md"), "w") as f:
    f.write(f"# {rnd}: kernels against their rooflines (MI355X, rocprofv3; regenerate with tools/refresh_evidence.sh + tools/roofline_report.py)\n\n")
    f.write(f"bench.py (100 steps): **{bench['value']} Msamples/s** ({bench['ms_per_step']} ms/step; from cold clocks {bench['value_cold']}), "
            f"default run (20 steps): {bench20['value']}; roofline.frac {bench['roofline']['frac']} (HIP events {bench['roofline']['avg_kernel_ms']} ms per launch); "
            f"cpu_baseline {bench20.get('cpu_baseline', {}).get('value')} Msamples/s ({bench20.get('cpu_baseline', {}).get('kind')}, {bench20.get('cpu_baseline', {}).get('cores')} threads)\n\n")
    f.write("Peaks: int8 matrix 5033 TOP/s dense (2 x the bf16 rate), f32 vector/matrix 157.3 TFLOP/s, f64 matrix 78.6 TFLOP/s, HBM 8 TB/s (MI355X_MICROARCH.md).  HBM traffic = FETCH_SIZE x 2 (gfx950 wide-read "
            "correction) + WRITE_SIZE of the kernel's last dispatch, separate --pmc passes.\n\n")
    f.write("| config | kernel | launches | avg us (rocprofv3) | flop / sample | bytes / sample | achieved vs peak | HBM traffic / algorithmic bytes |\n|---|---|---|---|---|---|---|---|\n")
    for (cfg, k, calls, avg, fl, by, frac, traffic, alg) in rows:
        short = k.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0] if k else "-"
        tr = f"{traffic / 1e6:.1f} MB / {alg / 1e6:.1f} MB = {traffic / alg:.2f}" if traffic and alg else "-"
        f.write(f"| {cfg} | `{short}` | {calls} | {avg / 1e3:.1f} | {fl} | {by} | {frac} | {tr} |\n")
with open(os.path.join(dst, f"{rnd}_pmc_summary.txt"), "w") as f:
    for case, k, c in summary:
        f.write(f"== {case}: {k}\n   last dispatch {c.get('_ns', 0) / 1e3:.1f} us  grid {c.get('_grid')} vgpr {c.get('_vgpr')} lds {c.get('_lds')} scratch {c.get('_scratch')}\n")
        for name, v in sorted(c.items()):
            if not name.startswith("_"): f.write(f"   {name:28s} {v:.6g}\n")
for extra in ("configs.jsonl", "wide.jsonl", "host_api.txt", "art_timing.txt", "fixed_point_shapes.txt"):
    if os.path.exists(os.path.join(src, extra)): shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{rnd}_{extra}"))
print(open(os.path.join(dst, f"{rnd}_roofline.md")).read())
