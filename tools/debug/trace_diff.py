"""First lines of two parity_trace.py outputs whose checksums differ by more than a relative tolerance."""
import sys, re
a, b = open(sys.argv[1]).read().splitlines(), open(sys.argv[2]).read().splitlines()
tol = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-6
print(len(a), len(b), "lines")
shown = 0
for x, y in zip(a, b):
    if x == y:
        continue
    fx, fy = re.findall(r"[-+]?\d\.\d+e[-+]\d+", x), re.findall(r"[-+]?\d\.\d+e[-+]\d+", y)
    if x.split(" sum ")[0] != y.split(" sum ")[0]:
        print("STRUCTURE", x[:120], "|", y[:120]); shown += 1
    elif fx and fy:
        rel = max(abs(float(p) - float(q)) / max(abs(float(p)), abs(float(q)), 1e-30) for p, q in zip(fx, fy))
        if rel > tol:
            print(f"rel {rel:.2e}  {x[:110]}\n              {y[:110]}"); shown += 1
    if shown >= 25:
        break
