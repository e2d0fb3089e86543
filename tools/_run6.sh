python -m pytest tests/test_exchange_gpu.py tests/test_recurrent_kernels_vs_oracle_gpu.py tests/test_blstm_gpu.py -x -q 2>&1 | tail -3
python tools/rec_fine_stamps.py 32
python tools/rec_fine_stamps.py 128
tools/copy_sweep > gpurun_out/copy_sweep.log 2>&1; grep -v "unroll 1 \|nt 1\|nt 2" gpurun_out/copy_sweep.log | tail -44
