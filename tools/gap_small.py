"""In-graph launch gaps: from a rocprofv3 --kernel-trace run (rocpd sqlite) sums, over the last `steps` steps (delimited by
tps_grid_kernel), the idle gaps between consecutive kernels that are shorter than 20 us (back-to-back launches inside a
graph replay or a tight eager sequence) and prints a histogram.  python tools/gap_small.py <dir-or-db> [steps]"""
import glob, os, sqlite3, sys
src = sys.argv[1]
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[-1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]
ks = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
marks = [i for i, r in enumerate(rows) if "tps_grid_kernel" in r[2]][-(nsteps + 1):]
seg = rows[marks[0]:marks[-1]]
n = len(marks) - 1
bins = [0.5, 1, 1.5, 2, 3, 5, 10, 20]
hist = [[0, 0.0] for _ in bins]
big = [0, 0.0]
cur_end = seg[0][1]
for s, e, name in seg[1:]:
    g = (s - cur_end) / 1e3
    if g > 0:
        for i, b in enumerate(bins):
            if g <= b:
                hist[i][0] += 1; hist[i][1] += g; break
        else:
            big[0] += 1; big[1] += g
    cur_end = max(cur_end, e)
print(f"{len(seg) / n:.0f} kernels per step")
lo = 0
for b, (cnt, tot) in zip(bins, hist):
    print(f"gaps {lo:>4} .. {b:<4} us: {cnt / n:7.1f} per step, {tot / n:8.1f} us per step")
    lo = b
print(f"gaps > 20 us       : {big[0] / n:7.1f} per step, {big[1] / n:8.1f} us per step (host-bound stretches under the profiler)")
