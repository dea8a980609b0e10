"""Per-kernel summary of a rocprofv3 `--kernel-trace` run (rocpd sqlite output of ROCm 7.2): writes the
`*_kernel_stats.csv` files kept under profiles/ (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs,
VGPRs, LDS bytes) and prints the top rows, optionally per step.  `python tools/prof_summary.py <dir-or-db> [out.csv] [steps]`"""
import csv, glob, os, sqlite3, subprocess, sys

src = sys.argv[1]
db = src if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*.db"), recursive=True))[-1]
out = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else None
steps = float(sys.argv[3]) if len(sys.argv) > 3 else None
c = sqlite3.connect(db)
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [x for x in t if x.startswith("rocpd_kernel_dispatch")][0]
ks = [x for x in t if x.startswith("rocpd_info_kernel_symbol")][0]
rows = list(c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start), "
                      f"s.arch_vgpr_count, max(d.group_segment_size) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"))
tot = sum(r[2] for r in rows)


def demangle(n):
    try:
        return subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", n.replace(".kd", "")], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


names = {r[0]: demangle(r[0]) for r in rows}
if out:
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "VGPRs", "LDSBytes"])
        for r in rows:
            w.writerow([names[r[0]], r[1], r[2], round(r[3], 1), round(100.0 * r[2] / tot, 4), r[4], r[5], r[6], r[7]])
print(f"total kernel time {tot / 1e6:.2f} ms, {sum(r[1] for r in rows)} launches" + (f"; per step: {tot / 1e6 / steps:.3f} ms, {sum(r[1] for r in rows) / steps:.0f} launches" if steps else ""))
for r in rows[:45]:
    per = f" {r[2] / 1e6 / steps:7.3f} ms/step {r[1] / steps:6.1f}/step" if steps else ""
    print(f"{names[r[0]][:100]:100s} {r[1]:6d} {r[2] / 1e6:8.2f} ms avg {r[3] / 1e3:7.1f} us {100.0 * r[2] / tot:5.2f}%{per}")
