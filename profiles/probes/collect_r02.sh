# Collects the round-2 profile evidence on the GPU box (run from the repo root through gpurun); outputs under gpurun_out/r02/.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02; mkdir -p $O
CMD="python3 bench.py --serial --no-ba --cpu-sample 0 --no-single --steps 10"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
FB_FAST_DBG=20 python3 profiles/probes/fast_phase_timers.py > $O/fast_phase_timers.txt 2>&1
echo collected
