# python tools/lws_time.py B U NW G over launch shapes of the two-utterances-per-wave kernel (and the skewed one beside it)
run() { timeout -k 10 100 python tools/lws_time.py "$@" 2>&1 | grep "B="; }
export AVSI_LWS_KERNEL=duo
for s in "2048 0 16 1" "1024 0 16 1" "768 0 16 1" "512 0 16 1" "384 0 16 1" "384 0 16 2" "256 0 16 2" "256 0 8 2" "192 0 16 2" "192 0 8 4" "128 0 16 4" "128 0 8 4" "128 0 8 8" "64 0 8 8" "64 0 4 16" "64 0 16 4"; do run $s || exit 1; done
export AVSI_LWS_KERNEL=skew
for s in "384" "256" "192" "128" "64"; do run $s || exit 1; done
