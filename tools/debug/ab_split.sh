# A/B of the fp32 -> 3 x bf16 split form (igemm_args.h) on one box: gemm micro-benchmark, then the headline step alternating
for v in "" _split_old; do
  echo "== lib '$v'"; ARCO_LIB=$PWD/arco_amd/lib/libarco_hip$v.so python tools/gemm_sp_bench.py 20 2>/dev/null | head -2 | cut -c1-190
done
for rep in 1 2; do for v in "" _split_old; do
  ARCO_LIB=$PWD/arco_amd/lib/libarco_hip$v.so python bench.py --steps 30 --warmup 5 --no_cpu_baseline --no_subs --k2_0_steps 0 --sustain_s 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lib \"$v\"', d['ms_per_step'])"
done; done
