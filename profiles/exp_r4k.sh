# round-4: batches in flight for the headline and for config C3 as written
for p in 2 3; do
python bench.py --steps 20 --warmup 5 --cpu-spp 0 --pipeline $p 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); c = d['value_c3_as_written']
print('pipeline $p: value %.1f (%.1f..%.1f) frac %.4f pipelined %.4f | c3 as written %.1f M paths/s, %.5f ms per iteration' % (d['value'], d['value_min'], d['value_max'], d['roofline']['frac'], d['roofline']['frac_pipelined'], c['value'], c['ms_per_iteration']))"
done
