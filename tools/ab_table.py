"""Tabulate gpurun_out/ab_<tag>.txt (tools/ab_micro.sh): sp-kernel launch time per shape and library variant (rep1/rep2)."""
import re, collections, sys
txt = open(sys.argv[1]).read()
cur = None; data = collections.OrderedDict(); vars_ = []
for line in txt.splitlines():
    m = re.match(r"=== (\w+) \(rep (\d)\)", line)
    if m:
        cur = (m.group(1), m.group(2))
        if m.group(1) not in vars_: vars_.append(m.group(1))
        continue
    m = re.match(r"(OK |BAD) (nb=\d+ \d+->\d+@\d+):.*time old\s+([\d.]+) us .* sp\s+([\d.]+) us", line)
    if m: data.setdefault(m.group(2), {})[cur] = (float(m.group(3)), float(m.group(4)), m.group(1))
    m = re.match(r"(copy|bn_act_fwd|conv3x3_rw<8,1> 16->16)\s+([\d.]+) us", line)
    if m: data.setdefault(m.group(1), {})[cur] = (0.0, float(m.group(2)), "OK ")
print("shape".ljust(26), " ".join(v.rjust(13) for v in vars_))
for k, v in data.items():
    print(k.ljust(26), " ".join((f"{v.get((x, '1'), (0, 0, ''))[1]:5.1f}/{v.get((x, '2'), (0, 0, ''))[1]:5.1f}" + ("!" if any(v.get((x, r), (0, 0, 'OK '))[2] != 'OK ' for r in '12') else " ")).rjust(13) for x in vars_))
