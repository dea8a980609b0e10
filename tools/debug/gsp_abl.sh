# ablation builds of gemm_sp_kernel (tools/build_variant.sh gsp_<X> "-DGSP_..." gemm_sp.hip): what the loader / MFMA waves cost alone
for v in "" _gsp_NO_CONS _gsp_NO_PROD _gsp_NO_SPLIT; do
  echo "== variant '$v'"
  ARCO_LIB=$PWD/arco_amd/lib/libarco_hip$v.so timeout 300 python tools/gemm_sp_bench.py 20 2>/dev/null | sed -e 's/max err vs fp64/err/' | cut -c1-260 | head -${ROWS:-8}
done
