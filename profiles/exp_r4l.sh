# round-4: the round's C2 builds on one box (suffixes = commits), pipelined and alone
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0" _base _66690a7 _e6b4c94 ""
bash profiles/ab_libs.sh "--steps 20 --warmup 5 --repeats 7 --per-iteration-sample 0 --pipeline 1" _base _66690a7 _e6b4c94 ""
