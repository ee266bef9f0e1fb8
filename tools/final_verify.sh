cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
timeout 1500 python -m pytest tests/ -m gpu -q > gpurun_out/final/gputests.txt 2>&1
tail -3 gpurun_out/final/gputests.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.txt 2>&1; tail -2 gpurun_out/final/smoke.txt
timeout 1200 python bench.py > gpurun_out/final/bench.txt 2>&1; tail -c 600 gpurun_out/final/bench.txt
