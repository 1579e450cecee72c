#!/bin/bash
# sample the shader clock and socket power while a command runs:  tools/clock_watch.sh <tag> <cmd...>   -> gpurun_out/clock_<tag>.txt
tag=$1; shift
mkdir -p gpurun_out
( while true; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 0.25; done ) > gpurun_out/clock_$tag.txt &
W=$!
"$@"
kill $W
