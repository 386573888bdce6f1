# round-4: tickets drawn one at a time (_run1), in runs of two (_run2) and of four (default)
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _run1 ""
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _run1 ""
