# round-4: resident workgroups per CU and batches in flight at the new step of 64 iterations
for p in 2 3; do for b in 5 6 7 8; do
PT_AMD_BLOCKS_PER_CU=$b python bench.py --steps 20 --warmup 5 --repeats 7 --cpu-spp 0 --per-iteration-sample 0 --pipeline $p 2>/dev/null | python profiles/line_fields.py "pipeline $p, $b workgroups per CU"
done; done
