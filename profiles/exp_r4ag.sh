# round 4: LATE against the build before, with larger chunks of the path pools (a chunk's worth of appends is the slack a late reservation's chunk installation has)
A="--per-iteration-sample 0 --repeats 5"
for cs in 11 13 15; do
  export PT_AMD_CHUNK_SHIFT=$cs
  echo "chunk shift $cs" >> gpurun_out/r4ag.txt
  bash profiles/ab_libs.sh "--steps 20 --warmup 5 $A" _prev "" >> gpurun_out/r4ag.txt
done
