#!/bin/bash
# A/B on ONE box: the headline step with the 2-D heads' adjoint on fp32 atomics (1) / order-independent (2); then the gradient's
# reproducibility over 60 executions in both settings (tools/debug/self_consistency.py, graph trainer)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2 3; do
for det in 1 2; do
  echo "det=$det: $(ARCO_DET_SCATTER=$det python bench.py --steps 30 --warmup 5 --no_subs --no_cpu_baseline --k2_0_steps 0 --sustain_s 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d["ms_per_step"], d["value"])')"
done
done
for det in 1 2; do
  echo "det=$det reproducibility: $(ARCO_DET_SCATTER=$det python tools/debug/self_consistency.py 60 4 g 2>&1 | tail -3 | tr '\n' ' ' | cut -c1-300)"
done
