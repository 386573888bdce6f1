#!/bin/bash
# round 6: config C3 as written through the library's own collective (a group of one member, one-rank RCCL communicator) against the HW queues a
# process gets and the member's commit stream (PT_AMD_GROUP_OWN_STREAMS=1: a stream of its own, as the first build of the round had it)
for q in 4 6 8; do
  for o in 0 1; do
    echo "== GPU_MAX_HW_QUEUES=$q PT_AMD_GROUP_OWN_STREAMS=$o"
    GPU_MAX_HW_QUEUES=$q PT_AMD_GROUP_OWN_STREAMS=$o python profiles/group_probe.py --only c3
  done
done
