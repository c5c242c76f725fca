#!/usr/bin/env python
"""per-kernel totals from a rocprofv3 rocpd sqlite file: rocpd_stats.py results.db [steps] [rows] [name-substring grid_x]
-> markdown table; `--after NAME COUNT` (anywhere on the line) drops everything up to and including the COUNT-th launch of a
kernel whose name contains NAME -- e.g. `--after sgd_multi 18` skips 3 warm-up train steps (6 optimizer launches each), so
one-time work (operand caches, first-step allocations) stays out of the per-step table; with a name substring and a grid size (threads) also the average duration of exactly those launches
(bench.py's roofline leg launches the dominant kernel on ONE shape: grid_x = ceil(M/128)*ceil(K/128)*256)"""
import sqlite3, sys, re
after = None
if "--after" in sys.argv:
    i = sys.argv.index("--after")
    after = (sys.argv[i + 1], int(sys.argv[i + 2]))
    del sys.argv[i:i + 3]
by_grid = []
while "--by-grid" in sys.argv:   # `--by-grid NAME`: launches of kernels matching NAME grouped by grid size (one row per layer shape)
    i = sys.argv.index("--by-grid")
    by_grid.append(sys.argv[i + 1])
    del sys.argv[i:i + 2]
db = sqlite3.connect(sys.argv[1])
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
where = ""
if after is not None:
    marks = [r[0] for r in cur.execute("select start from kernels where %s like ? order by start" % name_col, ("%" + after[0] + "%",))]
    where = " where start > %d" % marks[after[1] - 1]
if steps <= 0:  # 0: count the steps in the window (one ce_fwd_kernel launch per train step)
    steps = float(cur.execute("select count(*) from kernels%s%s like '%%ce_fwd_kernel%%'" % (where, (" and " if where else " where ") + name_col)).fetchone()[0]) or 1.0
rows = cur.execute("select %s, count(*), sum(end - start), avg(end - start) from kernels%s group by %s order by 3 desc" % (name_col, where, name_col)).fetchall()
total = sum(r[2] for r in rows)
print("total kernel time %.1f ms = %.2f ms/step" % (total / 1e6, total / 1e6 / steps))
print("| kernel | calls | ms/step | avg us | % |\n|---|---|---|---|---|")
for name, n, tot, avg in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    print("| `%s` | %d | %.2f | %.1f | %.2f |" % (name[:90], n, tot / 1e6 / steps, avg / 1e3, 100.0 * tot / total))

if len(sys.argv) > 5:
    sub, gx = sys.argv[4], int(sys.argv[5])
    r = cur.execute("select count(*), avg(end - start), min(end - start), max(end - start) from kernels where %s like ? and grid_x = ?" % name_col,
                    ("%" + sub + "%", gx)).fetchone()
    print("\nlaunches of *%s* with grid_x=%d: n=%d avg %.1f us (min %.1f, max %.1f)" % (sub, gx, r[0], r[1] / 1e3, r[2] / 1e3, r[3] / 1e3))

for sub in by_grid:
    w2 = (where + " and " if where else " where ") + "%s like ?" % name_col
    rs = cur.execute("select grid_x, grid_y, grid_z, count(*), avg(end - start), sum(end - start) from kernels%s group by grid_x, grid_y, grid_z order by 6 desc" % w2,
                     ("%" + sub + "%",)).fetchall()
    print("\nlaunches of *%s* by grid (threads): x y z | per step | avg us | ms/step" % sub)
    for gx, gy, gz, n, avg, tot in rs[:40]:
        print("  %8d %5d %3d | %6.1f | %7.1f | %6.3f" % (gx, gy, gz, n / steps, avg / 1e3, tot / 1e6 / steps))
