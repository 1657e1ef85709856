#!/usr/bin/env python3
"""tools/pmc_summary.py <dir> [bench args] — fold rocprofv3 counter_collection CSVs (one sub-directory per --pmc
pass, written by tools/pmc_collect.sh) into one JSON: per kernel, per counter, launches and the per-launch mean.
HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE reports half the bytes of 16 B/lane
streaming reads (MI355X_MICROARCH.md, HBM), WRITE_SIZE is exact for 16 B/lane stores."""
import collections
import csv
import glob
import json
import os
import sys


def main():
    d = sys.argv[1]
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(d, "p*", "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            if not any(x in name for x in ("k_scatter", "k_hist", "k_join", "k_part", "k_copy")):
                continue
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out = {}
    for k, cs in sorted(vals.items()):
        # Launches that did nothing are left out of the per-launch means: a kernel that finds its relation's overflow flag up returns
        # at once (the optimistic attempt on a skewed relation, the join behind it), and averaging those in would make a launch look
        # as if it moved less than its algorithmic bytes.  "Did nothing" = below 2 % of the largest launch of that kernel and counter.
        e = {}
        for c, v in sorted(cs.items()):
            top = max(v)
            kept = [x for x in v if top <= 0 or x >= 0.02 * top]
            e[c] = {"launches": len(kept), "per_launch": sum(kept) / len(kept), "empty_launches_dropped": len(v) - len(kept)}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            e["hbm_bytes_per_launch"] = (2 * e["FETCH_SIZE"]["per_launch"] + e["WRITE_SIZE"]["per_launch"]) * 1024
        if "SQ_LDS_IDX_ACTIVE" in e and "SQ_LDS_BANK_CONFLICT" in e and e["SQ_LDS_IDX_ACTIVE"]["per_launch"]:
            e["lds_bank_conflict_frac"] = e["SQ_LDS_BANK_CONFLICT"]["per_launch"] / e["SQ_LDS_IDX_ACTIVE"]["per_launch"]
        if "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]["per_launch"]:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
                if c in e:
                    e[c + "_over_WAVE_CYCLES"] = e[c]["per_launch"] / e["SQ_WAVE_CYCLES"]["per_launch"]
        out[k] = e
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icde2019-gpu-join_amd", "libhj.so")
    json.dump({"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(),
               "command": "rocprofv3 --pmc <group> --kernel-trace --output-format csv -- python3 bench.py --steps 1 --warmup 0 "
                          "--no-cpu-baseline --no-materialize --no-extras " + " ".join(sys.argv[2:]) +
                          " (one pass per counter group; FETCH_SIZE/WRITE_SIZE in KB, FETCH_SIZE doubled per MI355X_MICROARCH.md HBM)",
               "kernels": out}, sys.stdout, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
