for f in 0 1; do ARCO_FORCE_DIST=$f python tools/force_dist_check.py 4 2>/dev/null | grep FORCE_DIST > gpurun_out/fd_$f.json; done
python - <<'PY'
import json
a = json.loads(open("gpurun_out/fd_0.json").read().split("FORCE_DIST ",1)[1]); b = json.loads(open("gpurun_out/fd_1.json").read().split("FORCE_DIST ",1)[1])
for it,(x,y) in enumerate(zip(a["terms"], b["terms"])):
    print(it, {k: (x[k], y[k], abs(x[k]-y[k])) for k in x})
print(a["checksum"], b["checksum"], a["worst_repro"], b["worst_repro"], a["ms_per_step"], b["ms_per_step"], b["collectives_per_step"])
PY
