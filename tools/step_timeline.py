#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 --kernel-trace CSV: for the last replayed steps of each GAN step, the wall span of a
step, the time at least one kernel was running, the summed kernel time, and the per-queue idle gaps.

  python tools/step_timeline.py gpurun_out/tl/tl_kernel_trace.csv
"""
import csv
import sys
from collections import defaultdict


def main(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), int(r["Queue_Id"]), r["Kernel_Name"]))
    rows.sort()
    # a step starts at the D2D copy of the inputs that train_step issues before the graph launch
    starts = [i for i, r in enumerate(rows) if "copyBuffer" in r[3] or "direct_copy" in r[3]]
    # split into segments at gaps > 200 us (host work between phases) and at step boundaries (gaps > 8 us on all queues)
    segs, cur, last_end = [], [], None
    for r in rows:
        if last_end is not None and r[0] - last_end > 30_000:
            segs.append(cur)
            cur = []
        cur.append(r)
        last_end = max(last_end or 0, r[1])
    segs.append(cur)
    print(f"{len(rows)} dispatches, {len(segs)} bursts separated by > 30 us of idle GPU")
    big = [s for s in segs if len(s) >= 40]
    for name, group in (("last bursts", big[-6:]),):
        for s in group:
            t0, t1 = s[0][0], max(r[1] for r in s)
            busy, cur_end = 0, t0
            for a, b, q, n in s:
                if b > cur_end:
                    busy += b - max(a, cur_end)
                    cur_end = b
            ksum = sum(b - a for a, b, q, n in s)
            per_q = defaultdict(int)
            for a, b, q, n in s:
                per_q[q] += b - a
            print(f"  burst: {len(s):4d} kernels  span {(t1 - t0) / 1e3:8.1f} us  any-kernel-running {busy / 1e3:8.1f} us  "
                  f"sum {ksum / 1e3:8.1f} us  per queue {dict((q, round(v / 1e3, 1)) for q, v in per_q.items())}")
    # gap histogram inside the last big burst
    s = big[-1]
    gaps = []
    cur_end = s[0][1]
    for a, b, q, n in s[1:]:
        if a > cur_end:
            gaps.append((a - cur_end, n[:60]))
        cur_end = max(cur_end, b)
    gaps.sort(reverse=True)
    print("  idle gaps in the last burst:", len(gaps), "total", round(sum(g for g, _ in gaps) / 1e3, 1), "us; largest:")
    for g, n in gaps[:8]:
        print(f"    {g / 1e3:6.1f} us before {n}")
    byname = defaultdict(lambda: [0, 0])
    for a, b, q, n in s:
        k = n.split("(")[0][-50:]
        byname[k][0] += 1
        byname[k][1] += b - a
    print("  kernels of the last burst:")
    for k, (c, t) in sorted(byname.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"    {k:52s} {c:4d} {t / 1e3:8.1f} us")


if __name__ == "__main__":
    main(sys.argv[1])
