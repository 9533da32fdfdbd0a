"""Per-call kernel durations from a rocprofv3 --kernel-trace CSV, in launch order (the --stats summary averages coarse and fine
launches and forward and backward products of one kernel together).

    python tools/trace_calls.py <dir with *_kernel_trace.csv> [--last N] [--out file.json]

Prints the last N launches (default: one train chunk of the general-shape path = 80) as  name, µs, grid, LDS bytes."""
import argparse
import csv
import glob
import json
import os
import re

ap = argparse.ArgumentParser()
ap.add_argument("dir")
ap.add_argument("--last", type=int, default=80)
ap.add_argument("--out", default=None)
args = ap.parse_args()

files = sorted(glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True))
if not files:
    raise SystemExit(f"no *_kernel_trace.csv under {args.dir}")
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"knerf::(gen::)?", "", name)
    return name.split("(")[0].replace("void ", "")


out = []
for r in rows[-args.last:]:
    out.append({"kernel": short(r["Kernel_Name"]), "us": round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, 1),
                "grid": [int(r.get("Grid_Size_X", 0) or 0), int(r.get("Grid_Size_Y", 0) or 0), int(r.get("Grid_Size_Z", 0) or 0)],
                "wg": int(r.get("Workgroup_Size_X", 0) or 0), "lds": int(r.get("LDS_Block_Size", 0) or 0),
                "vgpr": int(r.get("VGPR_Count", 0) or 0)})
for o in out:
    print(f"{o['kernel']:<34} {o['us']:>9.1f} us  grid {o['grid']}  wg {o['wg']}  lds {o['lds']}  vgpr {o['vgpr']}")
if args.out:
    json.dump(out, open(args.out, "w"), indent=0)
