mkdir -p gpurun_out/r2zz
timeout 900 python bench.py > gpurun_out/r2zz/bench.json 2> gpurun_out/r2zz/bench.err; cut -c1-200 gpurun_out/r2zz/bench.json
PMC_OUT=gpurun_out/r2zz/pmc_fe bash tools/pmc_frontend_valu.sh > gpurun_out/r2zz/pmc_fe.log 2>&1; tail -16 gpurun_out/r2zz/pmc_fe.log; cp gpurun_out/r2zz/pmc_fe/summary.txt gpurun_out/r2zz/frontend_pmc_valu_summary.txt; rm -rf gpurun_out/r2zz/pmc_fe
