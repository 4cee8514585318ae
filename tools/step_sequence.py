#!/usr/bin/env python3
"""Kernel sequence of one replayed step out of a rocprofv3 --kernel-trace CSV: start offset, duration, queue, name.

  python tools/step_sequence.py <kernel_trace.csv> <n_kernels_back_from_the_end> [marker substring of the step's first kernel]
"""
import csv
import sys


def main(path, marker):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    rows.sort()
    idx = [i for i, r in enumerate(rows) if marker in r[3]]
    a, b = idx[-2], idx[-1]
    t0 = rows[a][0]
    qs = sorted({r[2] for r in rows[a:b]})
    for s, e, q, n in rows[a:b]:
        name = n.replace("void ", "").replace("sv::", "").split("(")[0][:60]
        print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}  q{qs.index(q)}  {name}")
    print(f"step span {(max(r[1] for r in rows[a:b]) - t0) / 1e3:.1f} us, {b - a} kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
