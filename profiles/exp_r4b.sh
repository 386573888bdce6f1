# round-4 A/B: fixed-cost diet of k_bounce (base = HEAD of round 3; _a = segment look-up by ballot + first loads before the staging barrier +
# 64 tally shards; _b = _a + tile tickets), pipelined and with one batch in flight
mkdir -p gpurun_out/r4b
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7" _b _c > gpurun_out/r4b/ab_p2.txt 2>&1
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --pipeline 1" _b _c > gpurun_out/r4b/ab_p1.txt 2>&1
cat gpurun_out/r4b/ab_p2.txt gpurun_out/r4b/ab_p1.txt
