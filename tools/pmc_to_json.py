"""Sum rocprofv3 --pmc counter_collection CSVs per kernel: python3 tools/pmc_to_json.py <kernel substring> <dir>...
Prints {counter: value per dispatch (mean over the dispatches whose duration is within 2x of the longest)} for the
kernels whose name contains the substring, plus dispatch count and mean duration per pass directory."""
import csv
import glob
import json
import os
import sys


def load(pattern, d):
    per = {}   # dispatch id -> {counter: value, "_dur": ns, "_meta": ...}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if pattern not in row["Kernel_Name"]:
                    continue
                e = per.setdefault(row["Dispatch_Id"], {"_dur": int(row["End_Timestamp"]) - int(row["Start_Timestamp"]),
                                                        "_vgpr": row["VGPR_Count"], "_scratch": row["Scratch_Size"], "_lds": row["LDS_Block_Size"],
                                                        "_grid": row["Grid_Size"], "_wg": row["Workgroup_Size"], "_name": row["Kernel_Name"]})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
    return per


def summarise(pattern, d):
    per = load(pattern, d)
    if not per:
        return {}
    longest = max(e["_dur"] for e in per.values())
    keep = [e for e in per.values() if e["_dur"] * 2 >= longest]   # leave out warm-up launches of a different size
    out = {"dispatches": len(keep), "dur_ms": sum(e["_dur"] for e in keep) / len(keep) / 1e6}
    for k in keep[0]:
        if k.startswith("_"):
            continue
        out[k] = sum(e.get(k, 0.0) for e in keep) / len(keep)
    out["meta"] = {k[1:]: keep[0][k] for k in keep[0] if k.startswith("_") and k != "_dur"}
    return out


if __name__ == "__main__":
    pat = sys.argv[1]
    print(json.dumps({os.path.basename(os.path.normpath(d)): summarise(pat, d) for d in sys.argv[2:]}, indent=1))
