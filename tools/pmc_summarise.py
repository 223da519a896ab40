#!/usr/bin/env python3
"""Summarise the FETCH_SIZE / WRITE_SIZE passes of tools/pmc_bench.sh into per-kernel HBM traffic.

Corrections follow /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the counters are in KiB;
on gfx950 FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream at 64 bytes, so the
read side is doubled (all kernels here read with dwordx4 per lane); WRITE_SIZE is taken as is.

usage: pmc_summarise.py gpurun_out/pmc_bench profiles/r01/pmc_traffic.json
"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def per_kernel(directory: str, counter: str):
    sums, counts = defaultdict(float), defaultdict(int)
    for path in glob.glob(f"{directory}/{counter}/**/*counter_collection.csv", recursive=True):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if row["Counter_Name"] != counter:
                    continue
                k = short(row["Kernel_Name"])
                sums[k] += float(row["Counter_Value"])
                counts[k] += 1
    return {k: sums[k] / counts[k] for k in sums}, dict(counts)


def main(src: str, dst: str) -> None:
    fetch, n_f = per_kernel(src, "FETCH_SIZE")
    write, n_w = per_kernel(src, "WRITE_SIZE")
    out = {"_note": "average per launch; bytes = KiB*1024; read side = FETCH_SIZE x2 (gfx950 "
                    "correction for dwordx4 streams), write side = WRITE_SIZE as reported",
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        if k.startswith("at::") or k.startswith("__amd"):
            continue
        rd = 2.0 * fetch.get(k, 0.0) * 1024.0
        wr = write.get(k, 0.0) * 1024.0
        out["kernels"][k] = {"launches": n_f.get(k, n_w.get(k, 0)),
                             "fetch_size_kib_raw": fetch.get(k), "write_size_kib_raw": write.get(k),
                             "read_bytes": rd, "write_bytes": wr, "hbm_bytes": rd + wr}
    with open(dst, "w") as fh:
        json.dump(out, fh, indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:45s} rd {v['read_bytes']/1e6:10.2f} MB  wr {v['write_bytes']/1e6:10.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
