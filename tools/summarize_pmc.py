#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs, as gpurun requires) per kernel.

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane)
coalesced read stream (MI355X_MICROARCH.md §HBM), so read bytes = 2 * FETCH_SIZE * 1024 for the glds-staged kernels;
WRITE_SIZE is taken at face value (uncalibrated).  Infinity-Cache hits are counted, so this is fabric traffic
(an upper bound on HBM traffic).
usage: summarize_pmc.py <fetch.csv> <write.csv> <out.json>
"""
import collections
import csv
import json
import sys


def agg(path):
    a = collections.defaultdict(lambda: [0, 0.0])
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0].strip()
            a[k][0] += 1
            a[k][1] += float(r["Counter_Value"])
    return a


def main():
    fetch, write = agg(sys.argv[1]), agg(sys.argv[2])
    out = {}
    for k in fetch:
        n, v = fetch[k]
        wn, wv = write.get(k, [0, 0.0])
        rd = 2.0 * v * 1024 / n
        wr = wv * 1024 / wn if wn else 0.0
        out[k] = dict(launches=n, fetch_kib_per_launch=v / n, write_kib_per_launch=(wv / wn if wn else 0.0),
                      read_bytes_per_launch_corrected=rd, write_bytes_per_launch=wr, traffic_bytes_per_launch=rd + wr)
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"])[:6]:
        print(f"{k[:50]:50s} launches={v['launches']:4d} traffic/launch={v['traffic_bytes_per_launch'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main()
