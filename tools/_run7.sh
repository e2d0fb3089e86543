python tools/thresholds_sweep.py > gpurun_out/thresholds_box1.log 2>&1; cat gpurun_out/thresholds_box1.log
for i in 1 2; do python bench.py --mode unet --batch 32 --steps 50 --warmup 10 | cut -c1-260; done
python tools/unet_infer_loop.py 32 2>&1 | tail -3
