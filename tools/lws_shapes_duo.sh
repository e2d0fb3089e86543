# python tools/lws_time.py B U NW G over launch shapes of the two-utterances-per-wave kernel (and the skewed one beside it)
run() { timeout -k 10 100 python tools/lws_time.py "$@" 2>&1 | grep "B="; }
export AVSI_LWS_KERNEL=duo
for s in "512 0 16 1" "384 0 16 1" "256 0 16 2" "192 0 16 2" "192 0 8 4" "128 0 16 4" "128 0 8 4" "128 0 16 2" "96 0 8 4" "96 0 16 4" "64 0 8 8" "64 0 8 4" "32 0 8 8" "32 0 4 16"; do run $s || exit 1; done
export AVSI_LWS_KERNEL=skew
for s in "192" "128" "96" "64" "32"; do run $s || exit 1; done
