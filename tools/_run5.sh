mkdir -p $GRAFT_REPO_ROOT/gpurun_out/profiles
python -m pytest tests/test_unet_gpu.py tests/test_gemm_gpu.py tests/test_train_gpu.py -x -q 2>&1 | tail -3
python tools/unet_train_one.py 512
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ut && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ut -- python3 $GRAFT_REPO_ROOT/tools/unet_train_one.py 512 > /tmp/prof_ut.txt 2> /tmp/prof_ut.err; cp $(ls /tmp/prof_ut/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/profiles/r05_bench_unet_train_b512_kernel_stats_v3.csv; cat /tmp/prof_ut.txt
cd $GRAFT_REPO_ROOT && python bench.py --steps 10 --warmup 3 > gpurun_out/bench_wip.json 2> gpurun_out/bench_wip.err; tail -c 300 gpurun_out/bench_wip.err
