# Probe (round 3, verdict item 4b): the B = 256 step with the k_blur launch compiled out (results are wrong; only the time means
# something) = the upper bound of what "no blurred pyramid" could give before the describe kernel pays for blurring its patches.
set -e
FB_BUILD_DEFS="-DFB_ORB_ABLATE_BLUR" python3 -m fishbirdeyevisualslam_amd.build --force > /dev/null
python3 bench.py --no-ba --cpu-sample 0 --no-single --no-chain > gpurun_out/ablate_blur.json 2> gpurun_out/ablate_blur.err || true
python3 -m fishbirdeyevisualslam_amd.build --force > /dev/null
python3 bench.py --no-ba --cpu-sample 0 --no-single --no-chain > gpurun_out/ablate_blur_ref.json 2> gpurun_out/ablate_blur_ref.err || true
python3 -c "
import json
a=json.load(open('gpurun_out/ablate_blur.json')); b=json.load(open('gpurun_out/ablate_blur_ref.json'))
print('step without k_blur %.3f ms, with %.3f ms' % (a['ms_per_step'], b['ms_per_step']))"
