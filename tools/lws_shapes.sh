# python tools/lws_time.py B [U NW G] over a list of launch shapes (0 = the library's choice); writes gpurun_out/lws_shapes.log
set -e
run() { timeout -k 10 100 python tools/lws_time.py "$@" 2>&1 | grep "B=" >> gpurun_out/lws_shapes.log; }
rm -f gpurun_out/lws_shapes.log
for b in 1 8 32 100 256 1024; do run $b; done
run 1 1 1 1; run 4 4 1 1                     # one wave alone: ms / (102 sweeps x 250 frames) = time per frame
run 8 1 8 13; run 32 4 4 16; run 64 2 8 8; run 64 1 8 4; run 100 2 8 5; run 256 2 8 2; run 512 4 4 2; run 512 2 8 1
cat gpurun_out/lws_shapes.log
