"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) per kernel.

usage: python profiles/summarize_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

Corrections for gfx950 (MI355X): both counters are in KiB; FETCH_SIZE reports one half of
the bytes of wide coalesced streaming reads (8-16 B per lane), so it is doubled; WRITE_SIZE
is exact for such stores.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch.

The summary carries `_meta`: the hash of the device sources the profiled library was built from
(msm_amd._lib.kernel_source_hash) and, where git is at hand, the commit.  bench.py reports these bytes as
`roofline.traffic` only while that hash equals the one of the tree it runs from.
"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def per_kernel(path):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = per_kernel(sys.argv[1])
    write, _ = per_kernel(sys.argv[2])
    out = {}
    for k in sorted(fetch):
        if "msm::" not in k:
            continue
        short = k.split("msm::(anonymous namespace)::")[1].split("(")[0]
        f, w = fetch[k], write.get(k, 0.0)
        out[short] = dict(launches=nf[k], fetch_size_kib=round(f, 1), write_size_kib=round(w, 1),
                          hbm_bytes_per_launch=int((2 * f + w) * 1024))
    sys.path.insert(0, ROOT)
    from msm_amd._lib import kernel_source_hash
    meta = dict(kernel_source_hash=kernel_source_hash())
    try:
        meta["git_sha"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], capture_output=True, text=True,
                                         timeout=10).stdout.strip() or None
    except (OSError, subprocess.SubprocessError):
        meta["git_sha"] = None          # the GPU box holds a snapshot without .git
    json.dump(dict(out, _meta=meta), open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        print(f"{k:45s} {v['hbm_bytes_per_launch'] / 1e9:8.3f} GB/launch")


if __name__ == "__main__":
    main()
