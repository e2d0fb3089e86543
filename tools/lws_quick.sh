# python tools/lws_time.py B U NW G over a few launch shapes of the skewed kernel (0 = the library's choice)
set -e
run() { timeout -k 10 100 python tools/lws_time.py "$@" 2>&1 | grep "B="; }
for s in "1" "8" "32" "64" "100" "128" "160" "256" "257" "1024"; do run $s; done
