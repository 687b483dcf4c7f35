mkdir -p gpurun_out/validate
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/validate/pytest.log; tail -4 gpurun_out/validate/pytest.log
bash tools/r2_profiles.sh r2c > gpurun_out/validate/profiles.log 2>&1; tail -5 gpurun_out/validate/profiles.log
cut -c1-400 gpurun_out/prof_r2c/bench.json
