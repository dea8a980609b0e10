#!/bin/bash
# A/B on ONE box: the row-sparse backward of the wide 1x1 convolutions (ops.SPARSE_BWD) - dense dataflow sub-record, LA, LiTS f16
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
for rep in 1 2; do
for sb in 0 1; do
  echo "sparse_bwd=$sb dense dataflow: $(ARCO_SPARSE_BWD=$sb python bench.py --sub dropin_dense_dataflow --sub_steps 6 2>/dev/null | python -c 'import sys,json; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(d["ms_per_step"], d["wide_1x1_backward_calls_per_step"], d["loss_terms"])')"
  echo "sparse_bwd=$sb LA: $(ARCO_SPARSE_BWD=$sb GRAPH_TRAIN=1 CONV_MMA=f32x3 EQV_PASS=1 python tools/bench3d.py 2 2>&1 | tail -1 | cut -c1-130)"
  echo "sparse_bwd=$sb LiTS f16: $(ARCO_SPARSE_BWD=$sb GRAPH_TRAIN=1 CONV_MMA=f32x3 ACT_DTYPE=f16 EQV_PASS=1 python tools/bench3d.py 1 160 160 96 2>&1 | tail -1 | cut -c1-130)"
done
done
