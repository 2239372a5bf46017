one() { python bench.py --no-ba --cpu-sample 0 --steps 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d['value']), 'b1', round(d['single_sequence']['b1']['ms_per_pair'],4), 'b8', round(d['single_sequence']['b8']['ms_per_pair'],4), 'pose', round(d['kernels_ms_per_step_single_stream']['k_pose_opt'],4))" $1; }
timeout -k 10 400 python -m pytest tests/test_pose_gpu.py tests/test_parity_sweeps_gpu.py tests/test_golden.py -m gpu -x -q 2>&1 | tail -3
FB_POSE_NT=448 timeout -k 10 400 python -m pytest tests/test_pose_gpu.py tests/test_parity_sweeps_gpu.py -m gpu -x -q -k "pose" 2>&1 | tail -3
one reg512; FB_POSE_NT=448 one split; one reg512; FB_POSE_NT=448 one split
