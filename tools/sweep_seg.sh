#!/bin/bash
# frames per segment of the time-sliced decode on the benchmark's own workload: launch and step times
cd $GRAFT_REPO_ROOT
for sg in ${SEGS:-84 100 125 143 167 200}; do
  DSR_VITERBI_SEG=$sg python bench.py --steps 6 --warmup 2 --no-cpu --no-verify 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(\"seg\", \"$sg\", \"value\", round(d[\"value\"],2), \"ms_per_step\", round(d[\"ms_per_step\"],1), \"launch_ms\", round(d[\"roofline\"][\"launch_ms\"],1), \"frac\", round(d[\"roofline\"][\"frac\"],4), \"serial_step\", round(d[\"config\"][\"serial_step_ms\"],1))"
done
