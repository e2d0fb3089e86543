#!/bin/bash
# Whole-step throughput of bench.py over batch sizes (the table in DESIGN.md section 5): bash tools/batch_sweep.sh
for b in 32 64 128 256 512 1024 2048 4096; do
    python bench.py --batch $b --steps 10 --warmup 3 --no-also --no-cpu-baseline 2>/dev/null | python tools/bench_line.py infer
done
for b in 8 32 128 256 1024; do
    python bench.py --mode train --batch $b --steps 10 --warmup 3 --no-also --no-cpu-baseline 2>/dev/null | python tools/bench_line.py train
done
