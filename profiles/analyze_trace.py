#!/usr/bin/env python
"""Group a rocprofv3 kernel-trace CSV by (kernel, grid): calls, total, mean, min duration."""
import collections
import csv
import sys


def main(path, top=40):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        short = name.split("svln12_GLOBAL__N_1")[-1][:58] if "svln" in name else name[:58]
        wg = int(r["Workgroup_Size_X"])
        key = (short, int(r["Grid_Size_X"]) // wg, r["Grid_Size_Y"], r["Grid_Size_Z"], wg)
        agg[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = sum(sum(v) for v in agg.values())
    print(f"total kernel time {tot / 1e6:.2f} ms")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
        print(f"{k[0]:60s} wgs({k[1]:>6d},{k[2]:>4s},{k[3]:>3s}) t{k[4]:>4d} calls {len(v):6d} tot_ms {sum(v) / 1e6:8.2f} "
              f"avg_us {sum(v) / len(v) / 1e3:8.2f} min {min(v) / 1e3:8.2f}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
