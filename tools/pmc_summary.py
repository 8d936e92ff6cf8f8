"""Summarise rocprofv3 --pmc CSVs produced by tools/pmc.sh: for each kernel, the counters of its LAST dispatch (bench.py
pre-rolls the device for 200 ms before its steps, so the last launches run at steady-state clocks; the first ones do not)."""
import csv, glob, os, sys, collections
d = sys.argv[1]
best = {}
for f in sorted(glob.glob(os.path.join(d, "*counter_collection.csv"))):
    rows = collections.defaultdict(dict)
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"]
        key = "mfma" if "fir_mfma" in name else "general" if "fir_general" in name else "strict" if "fir_strict" in name else None
        if not key: continue
        did = (key, row["Dispatch_Id"])
        rows[did][row["Counter_Name"]] = float(row["Counter_Value"])
        rows[did]["_ns"] = int(row["End_Timestamp"]) - int(row["Start_Timestamp"]); rows[did]["_id"] = int(row["Dispatch_Id"])
        rows[did]["_grid"] = row["Grid_Size"]; rows[did]["_vgpr"] = row["VGPR_Count"]; rows[did]["_lds"] = row["LDS_Block_Size"]
    for (key, did), c in rows.items():
        cur = best.setdefault((key, os.path.basename(f)), c)
        if c["_id"] > cur["_id"]: best[(key, os.path.basename(f))] = c
for (key, f), c in sorted(best.items()):
    print(f"== {key}  [{f}]  duration {c['_ns']/1e3:.1f} us  grid {c['_grid']} vgpr {c['_vgpr']} lds {c['_lds']}")
    for k, v in sorted(c.items()):
        if not k.startswith("_"): print(f"   {k:28s} {v:.5g}")
