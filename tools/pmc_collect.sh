# PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) over the dominant GEMM shapes and the decode gate|up stream.
# usage (GPU box): bash tools/pmc_collect.sh  -> gpurun_out/pmc_{gemm,skinny}_{FETCH_SIZE,WRITE_SIZE}.json
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for tool in gemm skinny; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_x
    rocprofv3 --kernel-trace --pmc $ctr -d /tmp/pmc_x -o x -- python $R/tools/${tool}_pmc.py > /dev/null 2>&1
    python $R/tools/${tool}_pmc.py --parse $(find /tmp/pmc_x -name "*.db" | head -1) $ctr > $R/gpurun_out/pmc_${tool}_${ctr}.json
  done
done
head -c 1500 $R/gpurun_out/pmc_gemm_FETCH_SIZE.json; head -c 600 $R/gpurun_out/pmc_skinny_FETCH_SIZE.json
