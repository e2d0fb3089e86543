# python tools/lws_time.py B U NW G over launch shapes of the two-utterances-per-wave kernel
run() { AVSI_LWS_KERNEL=duo timeout -k 10 100 python tools/lws_time.py "$@" 2>&1 | grep "B="; }
for s in "1024 0 16 1" "1024 0 8 1" "512 0 16 1" "512 0 8 1" "512 0 8 2" "256 0 16 2" "256 0 16 1" "256 0 8 2" "256 0 8 4" "128 0 16 2" "128 0 16 4" "128 0 8 4" "2048 0 16 1"; do run $s || exit 1; done
