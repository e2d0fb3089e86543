set -x
python -m pytest tests/test_istft_gpu.py tests/test_ref_docs_golden.py tests/test_unet_gpu.py tests/test_train_gpu.py tests/test_fullsize_properties_gpu.py -x -q > gpurun_out/t4.log 2>&1; tail -5 gpurun_out/t4.log
for occ in 3 2; do for late in 0 1; do echo "OCC=$occ LATE=$late"; AVSI_ISTFT_OCC=$occ AVSI_ISTFT_LATE=$late python tools/istft_time.py 4096; done; done > gpurun_out/istft_ab.log 2>&1; cat gpurun_out/istft_ab.log
python tools/clock_under_load.py 8192 > gpurun_out/clock.log 2>&1; cat gpurun_out/clock.log
python tools/lws_time.py 1024 > gpurun_out/lws_slots.log 2>&1; AVSI_LWS_DUO_SLOTS=128 python tools/lws_time.py 1024 >> gpurun_out/lws_slots.log 2>&1; AVSI_LWS_DUO_SLOTS=64 python tools/lws_time.py 1024 >> gpurun_out/lws_slots.log 2>&1; cat gpurun_out/lws_slots.log
python tools/unet_train_one.py 512
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ut && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ut -- python3 $GRAFT_REPO_ROOT/tools/unet_train_one.py 512 > /tmp/prof_ut.txt 2> /tmp/prof_ut.err; cp $(ls /tmp/prof_ut/*/*kernel_stats.csv | head -1) $GRAFT_REPO_ROOT/gpurun_out/profiles/r05_bench_unet_train_b512_kernel_stats_v2.csv; cat /tmp/prof_ut.txt
