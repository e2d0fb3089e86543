for na in 0 1; do echo "NOACK=$na"; AVSI_COOP_NOACK=$na python tools/rec_coop_time.py 250 32 64 128 256 2>&1 | grep Bp; done
AVSI_COOP_NOACK=1 python -m pytest tests/test_exchange_gpu.py tests/test_recurrent_kernels_vs_oracle_gpu.py tests/test_blstm_gpu.py tests/test_golden.py -x -q 2>&1 | tail -3
for na in 0 1; do echo "NOACK=$na"; AVSI_COOP_NOACK=$na python bench.py --batch 32 --steps 50 --warmup 10 --no-cpu-baseline --no-also | cut -c1-330; done
