#!/usr/bin/env python
"""Per-phase GPU busy time of one steady turn from a rocprofv3 kernel-trace CSV of bench.py.
usage: turn_breakdown.py <kernel_trace.csv> [turn_index]   (a turn starts at a patchify launch)"""
import collections
import csv
import sys


def main(path, turn=10, detail=False):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "patchify" in r["Kernel_Name"]]
    seg = rows[idx[turn]:idx[turn + 1]]
    ph, stats, prev_end, start = "vision", collections.OrderedDict(), None, {}
    gaps, prev_nm = [], None
    for r in seg:
        nm, s, e = r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if ph == "vision" and "gather_rows" in nm:
            ph = "prefill"
        if ph == "prefill" and "gemv" in nm:
            ph = "decode"
        st = stats.setdefault(ph, [0, 0, 0, 0])
        st[0] += e - s
        st[3] += 1
        if prev_end is not None:
            st[1] += max(0, s - prev_end)
            gaps.append((max(0, s - prev_end), ph, prev_nm[:48], nm[:48]))
        prev_nm = nm
        start.setdefault(ph, s)
        st[2] = e - start[ph]
        prev_end = e
    tot = 0
    for k, v in stats.items():
        print(f"{k:8s} busy {v[0] / 1e6:7.3f} ms  gaps {v[1] / 1e6:6.3f} ms  span {v[2] / 1e6:7.3f} ms  kernels {v[3]}")
        tot += v[2]
    if detail:
        for p in stats:
            g = [x[0] for x in gaps if x[1] == p]
            small = sum(x for x in g if x < 3000)
            print(f"{p:8s} gaps < 3 us: {small / 1e3:8.1f} us in {sum(1 for x in g if x < 3000)}; >= 3 us: {(sum(g) - small) / 1e3:8.1f} us in {sum(1 for x in g if x >= 3000)}")
        for g in sorted(gaps, reverse=True)[:14]:
            print(f"  gap {g[0] / 1e3:7.1f} us  {g[1]:8s} {g[2]}  ->  {g[3]}")
    print(f"turn span {(int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6:.3f} ms, kernels {len(seg)}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 10, detail=len(sys.argv) > 3 and sys.argv[3] == "gaps")
