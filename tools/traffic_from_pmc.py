"""Turns the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py` into profiles/<round>_traffic.json.

    python tools/traffic_from_pmc.py <workload> <kernel substring> <fetch csv> <write csv> <algorithmic bytes/launch> <out json>

FETCH_SIZE is doubled (gfx950 tallies 128-B read requests at 64 B, MI355X_MICROARCH.md), WRITE_SIZE taken as is;
both are in KB.  The mean is over the dispatches of the named kernel.
"""
import csv
import json
import os
import sys


def mean_kb(path, counter, kernel):
    vals = {}
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if kernel in r["Kernel_Name"] and r["Counter_Name"].startswith(counter):
                vals[r["Dispatch_Id"]] = vals.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return len(vals), sum(vals.values()) / max(len(vals), 1)


def main():
    workload, kernel, fetch_csv, write_csv, alg, out = sys.argv[1:7]
    nf, f = mean_kb(fetch_csv, "FETCH_SIZE", kernel)
    nw, w = mean_kb(write_csv, "WRITE_SIZE", kernel)
    data = {}
    if os.path.exists(out):
        with open(out) as fh:
            data = json.load(fh)
    data[workload] = {
        "kernel": kernel,
        "raw": {"FETCH_SIZE": {"dispatches": nf, "mean_kb_per_launch": f}, "WRITE_SIZE": {"dispatches": nw, "mean_kb_per_launch": w}},
        "fetch_bytes_per_launch_corrected": 2 * f * 1024,
        "write_bytes_per_launch": w * 1024,
        "traffic_bytes_per_launch": 2 * f * 1024 + w * 1024,
        "algorithmic_bytes_per_launch": float(alg),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of `python3 bench.py --steps 2 --warmup 1 "
                  "--no-cpu-baseline`; mean over the kernel's dispatches; FETCH_SIZE x2 per MI355X_MICROARCH.md, WRITE_SIZE as is; unit KB",
    }
    with open(out, "w") as fh:
        json.dump(data, fh, indent=1)
    print(json.dumps(data[workload], indent=1))


if __name__ == "__main__":
    main()
