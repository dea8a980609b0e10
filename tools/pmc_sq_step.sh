#!/bin/bash
# on the GPU box: SQ counters per kernel (wave cycles parked / issuing, MFMA busy) over the default 2-D step, the LA step and the
# f16-storage LiTS step -> gpurun_out/sq_<tag>.csv + a table on stdout
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out
C="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_VALU"
cd /tmp
rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sq2d -o x -- python3 $root/tools/prof_step.py 2 > $o/sq2d.log 2>&1
cp $(find /tmp/sq2d -name "*counter_collection.csv" | head -1) $o/sq_2d.csv
EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sq3d -o x -- python3 $root/tools/bench3d.py 2 > $o/sq3d.log 2>&1
cp $(find /tmp/sq3d -name "*counter_collection.csv" | head -1) $o/sq_la.csv
EQV_PASS=1 CONV_MMA=f32x3 GRAPH_TRAIN=1 ACT_DTYPE=f16 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/sq3h -o x -- python3 $root/tools/bench3d.py 1 160 160 96 > $o/sq3h.log 2>&1
cp $(find /tmp/sq3h -name "*counter_collection.csv" | head -1) $o/sq_lits_f16.csv
cd $root
python3 tools/pmc_sq_table.py gpurun_out/sq_2d.csv gpurun_out/sq_la.csv gpurun_out/sq_lits_f16.csv
