"""tools/kernel_timeline.py <rocprofv3 ..._kernel_trace.csv> [t0_ms t1_ms]: the dispatches of a run in time order -- start (ms since
the first dispatch), duration, queue, kernel -- and, per kernel name, calls / total / average inside the window.  What
`--stats` cannot show: which kernels run beside each other and where a stream waits."""
import csv
import sys
from collections import defaultdict


def main():
    path = sys.argv[1]
    rows = list(csv.DictReader(open(path)))
    if not rows:
        return
    k_start = next(c for c in rows[0] if c.lower().startswith("start"))
    k_end = next(c for c in rows[0] if c.lower().startswith("end"))
    k_name = next(c for c in rows[0] if c.lower() in ("kernel_name", "name"))
    k_q = next((c for c in rows[0] if c.lower().startswith("queue")), None)
    ev = sorted((int(r[k_start]), int(r[k_end]), r.get(k_q, "") if k_q else "", r[k_name]) for r in rows)
    t_first = ev[0][0]
    lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
    hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e18
    qmap = {}
    agg = defaultdict(lambda: [0, 0.0])
    for s, e, q, name in ev:
        t = (s - t_first) / 1e6
        if t < lo or t > hi:
            continue
        qi = qmap.setdefault(q, len(qmap))
        short = name.split("(")[0][-70:]
        print("%10.3f + %8.3f ms  q%d  %s" % (t, (e - s) / 1e6, qi, short))
        agg[short][0] += 1; agg[short][1] += (e - s) / 1e6
    print("---- per kernel inside the window")
    for name, (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%9.3f ms  %5d calls  %9.3f ms avg  %s" % (tot, n, tot / n, name))


if __name__ == "__main__":
    main()
