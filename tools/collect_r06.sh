#!/bin/bash
# on the GPU box: the round's evidence in one call -> gpurun_out/r06_*  (copied into profiles/ afterwards)
root=$(pwd); export TMPDIR=/tmp; o=$root/gpurun_out
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $o/r06_bench.json 2> $o/r06_bench.err
bash tools/prof_run.sh r06 40 > $o/r06_prof2d.txt 2>&1
bash tools/prof_run3d.sh r06 > $o/r06_prof3d.txt 2>&1
bash tools/prof_run3d.sh r06_lits "SHAPE=1 160 160 96" ACT_DTYPE=f16 > $o/r06_prof3d_lits.txt 2>&1
mkdir -p $o/r06_pmc_conv_rw $o/r06_pmc_gemm_sp
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_hf -o f -- python3 $root/tools/pmc_conv_rw.py > $o/r06_pmc_f.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_hw -o w -- python3 $root/tools/pmc_conv_rw.py > $o/r06_pmc_w.log 2>&1)
grep -h "conv3x3_rw_kernel" $(find /tmp/pmc5_hf -name "*counter_collection.csv" | head -1) > $o/r06_pmc_conv_rw/fetch_size_counter_collection.csv
grep -h "conv3x3_rw_kernel" $(find /tmp/pmc5_hw -name "*counter_collection.csv" | head -1) > $o/r06_pmc_conv_rw/write_size_counter_collection.csv
head -1 $(find /tmp/pmc5_hf -name "*counter_collection.csv" | head -1) > $o/r06_pmc_conv_rw/header.csv
# the dense 496-wide GEMM (gemm_sp_kernel): HBM traffic of one launch class
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_gf -o f -- python3 $root/tools/gemm_sp_bench.py pmc > $o/r06_pmc_gf.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc5_gw -o w -- python3 $root/tools/gemm_sp_bench.py pmc > $o/r06_pmc_gw.log 2>&1)
grep -h "gemm_sp_kernel\|igemm_kernel" $(find /tmp/pmc5_gf -name "*counter_collection.csv" | head -1) > $o/r06_pmc_gemm_sp/fetch_size_counter_collection.csv
grep -h "gemm_sp_kernel\|igemm_kernel" $(find /tmp/pmc5_gw -name "*counter_collection.csv" | head -1) > $o/r06_pmc_gemm_sp/write_size_counter_collection.csv
head -1 $(find /tmp/pmc5_gf -name "*counter_collection.csv" | head -1) > $o/r06_pmc_gemm_sp/header.csv
tail -c 800 $o/r06_bench.json; head -3 $o/r06_prof2d.txt | cut -c1-160; wc -l $o/r06_pmc_conv_rw/*.csv $o/r06_pmc_gemm_sp/*.csv
bash tools/prof_bench.sh r06 > $o/r06_prof_bench.txt 2>&1
python3 tools/step_timeline.py > $o/r06_timeline_final.txt 2>&1
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > $o/r06_gputests_final.log
python -c "import __graft_entry__ as g; g.smoke()" >> $o/r06_gputests_final.log 2>&1
cat $o/r06_gputests_final.log; tail -4 $o/r06_prof_bench.txt | cut -c1-300
