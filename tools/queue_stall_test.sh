#!/bin/bash
# The work queue's ring-overwrite hazard, exercised: a build in which one workgroup sits
# ~25 ms on a drawn ticket without reading its entry while the ring (shrunk to the smallest legal size) wraps under it
# must give the SAME trajectories as the shipped library -- the host rewrites a slot only once the entry it held has
# been answered.  GPU box; builds the 2->16-16-1-only variant first (about a minute).
#   tools/queue_stall_test.sh [ticket, default 3000] [loops, default 1100]
cd "$(dirname "$0")/.." || exit 1
t=${1:-3000}; loops=${2:-1100}
tools/build_variant.sh stall -DBORE_SHAPE_MASK=0x2 -DBORE_QUEUE_STALL_TEST=$t || exit 1
python tools/ab_engine.py --check --loops $loops --reps 2 --rounds 1 --steps 15 default stall
