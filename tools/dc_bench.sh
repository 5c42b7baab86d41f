set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 200 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread -k "dcfilter" > gpurun_out/dc_tests.log 2>&1 || { tail -20 gpurun_out/dc_tests.log; exit 1; }
tail -1 gpurun_out/dc_tests.log
for sa in 1 0; do
timeout -k 10 300 python bench.py --format SC16 --dcfilter --submit-ahead $sa --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/bench_sc16dc_sa$sa.json 2> gpurun_out/bench_sc16dc.err
done
timeout -k 10 300 python bench.py --submit-ahead 1 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_default_sa1.json 2>> gpurun_out/bench_sc16dc.err
timeout -k 10 300 python bench.py --submit-ahead 0 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_default_sa0.json 2>> gpurun_out/bench_sc16dc.err
