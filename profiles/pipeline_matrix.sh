for rep in 1 2; do
for cfg in "2 32 40" "3 32 40" "2 64 20" "3 64 20" "2 128 10" "3 128 10"; do
set -- $cfg
python bench.py --steps $3 --warmup 5 --pipeline $1 --batch $2 --cpu-spp 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('pipeline $1 batch $2', d['value'])"
done; done
