"""Idle gaps of the GPU in a rocprofv3 --kernel-trace run (rocpd sqlite): for the last `steps` steps (delimited by
tps_grid_kernel, one per step) prints busy / idle time per step and the largest gaps with the kernels on either side.
python tools/gap_report.py <dir-or-db> [steps]"""
import glob, os, sqlite3, sys
src = sys.argv[1]
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[-1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]
ks = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if "tps_grid_kernel" in r[2]]
marks = marks[-(nsteps + 1):]
tot_busy = tot_idle = 0
gaps = []
for a, b in zip(marks[:-1], marks[1:]):
    seg = rows[a:b + 1]
    span = seg[-1][0] - seg[0][0]
    busy = 0; cur_end = seg[0][0]
    for s, e, n in seg[:-1]:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n))
        busy += max(0, e - max(s, cur_end)) if e > cur_end else 0
        if e > cur_end:
            cur_end = e; prev = n
    tot_busy += busy; tot_idle += span - busy
    ksum = sum(e - s for s, e, n in seg[:-1])        # > busy when kernels of two streams ran side by side
    print(f"step span {span / 1e6:7.3f} ms  busy {busy / 1e6:7.3f}  idle {(span - busy) / 1e6:6.3f}  kernel-time sum {ksum / 1e6:7.3f}  kernels {len(seg) - 1}")
print(f"mean busy {tot_busy / 1e6 / (len(marks) - 1):.3f} ms  idle {tot_idle / 1e6 / (len(marks) - 1):.3f} ms")
gaps.sort(reverse=True)
agg = {}
for g, p, n in gaps:
    k = (p[:60], n[:60]); agg.setdefault(k, [0, 0]); agg[k][0] += g; agg[k][1] += 1
for (p, n), (g, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:18]:
    print(f"{g / 1e3 / (len(marks) - 1):8.1f} us/step in {cnt / (len(marks) - 1):5.1f} gaps  after {p}  before {n}")
