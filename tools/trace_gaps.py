"""Idle time between kernels of one step, from a `rocprofv3 --kernel-trace --output-format csv` trace.

    python tools/trace_gaps.py <dir with *_kernel_trace.csv> [first_kernel_of_a_step]

Takes the steady-state part of the trace (after the first third), splits it into steps at every launch of the
first kernel of a step (k_filter) and reports busy time, idle time and the largest gaps by predecessor kernel."""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    first = sys.argv[2] if len(sys.argv) > 2 else "k_filter"
    path = sorted(glob.glob(d + "/**/*kernel_trace.csv", recursive=True))[0]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[len(rows) // 3:]
    starts = [i for i, r in enumerate(rows) if first in r[2]]
    steps = [(starts[i], starts[i + 1]) for i in range(len(starts) - 1)]
    busy = idle = 0
    gap_after = defaultdict(list)
    for a, b in steps:
        for i in range(a, b):
            s, e, n = rows[i]
            busy += e - s
            g = rows[i + 1][0] - e
            idle += max(g, 0)
            gap_after[n.split("(")[0][:40]].append(g)
    n = len(steps)
    a, b = steps[len(steps) // 2]
    print("one step, in launch order (start offset us, duration us, gap to next us):")
    for i in range(a, b):
        st, en, nm = rows[i]
        print(f"  {(st - rows[a][0]) / 1e3:8.1f} {(en - st) / 1e3:7.1f} {(rows[i + 1][0] - en) / 1e3:6.1f}  {nm.split('(')[0][:60]}")
    print(f"{n} steps: busy {busy / n / 1e3:.1f} us, idle {idle / n / 1e3:.1f} us, kernels/step {sum(b - a for a, b in steps) / n:.1f}")
    for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:25]:
        print(f"  after {k:42s} mean gap {sum(v) / len(v) / 1e3:7.2f} us  x{len(v) / n:.1f}/step")


if __name__ == "__main__":
    main()
