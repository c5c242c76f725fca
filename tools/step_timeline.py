#!/usr/bin/env python
"""Timeline of ONE steady-state train step from a rocprofv3 rocpd sqlite file:  step_timeline.py results.db [step_index] [sgd_per_step]
A step = the span between the last optimizer launch (sgd_multi) of one step and of the next.  Prints, per HIP queue, the busy time and
first / last kernel; the time both / one / no queue is busy; the largest idle gaps of the busiest queue with the kernels around
them; and what runs in the last 2 ms (the tail the next step waits for)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 5
per = int(sys.argv[3]) if len(sys.argv) > 3 else 6
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("queue_id", "stream_id", "queue", "stream") if c in cols), None)
print("columns:", cols)
marks = [r[0] for r in cur.execute("select end from kernels where %s like '%%sgd_multi%%' order by start" % name_col)]
t0, t1 = marks[which * per - 1], marks[(which + 1) * per - 1]
rows = cur.execute("select %s, start, end, %s from kernels where start >= ? and end <= ? order by start" % (name_col, qcol or "0"), (t0, t1)).fetchall()
short = lambda n: re.sub(r"\(anonymous namespace\)::|void ", "", n)[:70]
print("step %d: %.3f ms, %d kernels" % (which, (t1 - t0) / 1e6, len(rows)))
queues = {}
for n, s, e, q in rows:
    queues.setdefault(q, []).append((s, e, n))
for q, ks in sorted(queues.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _ in ks)
    print("queue %s: %d kernels, busy %.3f ms, first at +%.3f ms (%s), last ends at +%.3f ms (%s)" % (
        q, len(ks), busy / 1e6, (ks[0][0] - t0) / 1e6, short(ks[0][2]), (max(e for _, e, _ in ks) - t0) / 1e6, short(ks[-1][2])))
# coverage: how long 0 / 1 / 2+ kernels are in flight
ev = sorted([(s, 1) for _, s, e, _ in rows] + [(e, -1) for _, s, e, _ in rows])
depth, last, cover = 0, t0, {}
for t, d in ev:
    cover[min(depth, 2)] = cover.get(min(depth, 2), 0) + (t - last)
    depth += d
    last = t
cover[0] = cover.get(0, 0) + (t1 - last)
print("in flight: none %.3f ms, one kernel %.3f ms, two or more %.3f ms" % tuple(cover.get(i, 0) / 1e6 for i in range(3)))
main = max(queues.items(), key=lambda kv: len(kv[1]))[1]
gaps = []
for (s0, e0, n0), (s1, e1, n1) in zip(main, main[1:]):
    if s1 - e0 > 3000:
        gaps.append((s1 - e0, e0, n0, n1))
print("busiest queue: %d gaps > 3 us, %.3f ms in all; total of all gaps %.3f ms" % (
    len(gaps), sum(g[0] for g in gaps) / 1e6, sum(max(0, b[0] - a[1]) for a, b in zip(main, main[1:])) / 1e6))
for g, at, n0, n1 in sorted(gaps, reverse=True)[:14]:
    print("  %7.1f us at +%.3f ms: after %s | before %s" % (g / 1e3, (at - t0) / 1e6, short(n0)[:44], short(n1)[:44]))
print("last 2 ms of the step:")
for n, s, e, q in rows:
    if e > t1 - 2_000_000 and e - s > 20_000:
        print("  q%s +%.3f .. +%.3f ms  %7.1f us  %s" % (q, (s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e3, short(n)))
