#!/usr/bin/env python3
"""Aggregates rocprofv3 --pmc CSVs (tools/pmc.sh) per kernel: mean counter value per dispatch.

    python tools/pmc_report.py <outdir> <layout tag> <skip_dead_tiles 0|1> [library that was profiled]

The summary records which kernels the counters belong to: `_kernel_digest` (keras_nerf_amd/build.py: source of the three big kernels
+ flags, from the record beside the profiled library) and `_lib_sha16`.  bench.py quotes a summary only when its digest equals that of
the library the bench itself loaded."""
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/pmc"
layout = sys.argv[2] if len(sys.argv) > 2 else None      # bench.py LAYOUT_TAG of the build that was profiled
skip = int(sys.argv[3]) if len(sys.argv) > 3 else None   # kbench --skip-dead-tiles of the profiled run (list-mode or contiguous backward kernels)
from keras_nerf_amd import _lib
info = _lib.build_info(sys.argv[4] if len(sys.argv) > 4 else None)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("knerf::", "")
        if not any(x in k for x in ("mlp_", "wgrad", "composite", "sample")):
            continue
        grid = int(r.get("Grid_Size", 0) or 0)
        agg[(k, grid)][r["Counter_Name"]].append(float(r["Counter_Value"]))
rep = {"_kernel_digest": info["kernel_digest"], "_lib_sha16": info["lib_sha16"], "_git_head_at_build": info["git_head_at_build"]}
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
