#!/bin/bash
# Kernel timelines of one 64-spp 1080p step and of one single-frame call (profiles/TAG_step_timeline.txt,
# profiles/TAG_one_frame_timeline.txt are copies of what this leaves): tools/timelines.sh TAG -> gpurun_out/TAG/
out=gpurun_out/${1:-tl}
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
FRAMES=64 TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/step -o wf -- python3 tools/one_frame.py > $out/step.log 2>&1 || exit 1
python3 tools/one_frame_trace.py $(find $out/step -name '*.db' | head -1) > $out/step_timeline.txt || exit 1
FRAMES=1 TRACE=1 timeout -k 10 300 rocprofv3 --kernel-trace -d $out/one -o wf -- python3 tools/one_frame.py > $out/one.log 2>&1 || exit 1
python3 tools/one_frame_trace.py $(find $out/one -name '*.db' | head -1) > $out/one_frame_timeline.txt || exit 1
rm -rf $out/step $out/one
