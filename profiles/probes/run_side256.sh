# side-stream blur on / off at the bench batch, interleaved on one box
for M in 64 100000 1 100000 64; do FB_ORB_SIDE_MIN=$M python bench.py --no-ba --cpu-sample 0 --steps 20 --no-single 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('sideMin', sys.argv[1], round(d['value']), round(d['ms_per_step'],3))" $M; done
