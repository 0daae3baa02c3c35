set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fin
make -C oracle -s
timeout -k 10 500 python bench.py --steps 64 --warmup 8 > gpurun_out/fin/bench.json 2> gpurun_out/fin/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fin/stats -o s -- python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/fin/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/fin/fetch -o f -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/fin/fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/fin/write -o w -- python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/fin/write.log 2>&1
cat gpurun_out/fin/bench.json
