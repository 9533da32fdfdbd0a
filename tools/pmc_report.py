#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (tools/pmc.sh) per kernel: mean counter value per dispatch."""
import csv, glob, json, sys, collections
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
layout = sys.argv[2] if len(sys.argv) > 2 else None      # bench.py LAYOUT_TAG of the build that was profiled
skip = int(sys.argv[3]) if len(sys.argv) > 3 else None   # kbench --skip-dead-tiles of the profiled run (list-mode or contiguous backward kernels)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("knerf::", "")
        if not any(x in k for x in ("mlp_", "wgrad", "composite", "sample")):
            continue
        grid = int(r.get("Grid_Size", 0) or 0)
        agg[(k, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
rep = {}
if layout:
    rep["_layout"] = layout
if skip is not None:
    rep["_options"] = {"skip_dead_tiles": skip}
for (k, grid), cs in sorted(agg.items()):
    rep[f"{k} grid={grid}"] = {c: round(sum(v) / len(v), 1) for c, v in sorted(cs.items())}
    row = rep[f"{k} grid={grid}"]
    row["dispatches"] = len(next(iter(cs.values())))
    if "FETCH_SIZE" in cs and "WRITE_SIZE" in cs:
        # gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section); both in KiB.
        # wgrad's coarse and fine launches share one grid size: max = fine pass, min = coarse pass.
        row["hbm_bytes_per_launch"] = (2.0 * max(cs["FETCH_SIZE"]) + max(cs["WRITE_SIZE"])) * 1024.0
        row["hbm_bytes_per_launch_min"] = (2.0 * min(cs["FETCH_SIZE"]) + min(cs["WRITE_SIZE"])) * 1024.0
print(json.dumps(rep, indent=1))
