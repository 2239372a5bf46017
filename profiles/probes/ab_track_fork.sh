for i in 1 2; do
for v in fork nofork; do
  if [ $v = nofork ]; then export FB_TRACK_NO_FORK=1; else unset FB_TRACK_NO_FORK; fi
  python bench.py --no-ba --cpu-sample 0 --no-single 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); tc=d['track_chain']
print('$v', {k: [round(tc[k][n]['ms_per_step'],3) for n in ('dependent','dependent_pipelined','reference_keyframe_dependent')] for k in ('b1','b8','b256')})
"
done; done
