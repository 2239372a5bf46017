set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
CMD="python3 bench.py --serial --no-ba --cpu-sample 0 --no-single --no-chain --steps 10"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
python3 profiles/probes/rocpd_stats.py $O/kt/kt_results.db > $O/r03_bench_serial_b256_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
python3 profiles/probes/make_traffic_json.py $O/fetch/f_results.db $O/write/w_results.db 256 $O/r03_pmc_traffic.json > $O/r03_pmc_fetch_write_b256.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES SQ_BUSY_CYCLES -d $O/sq -o s -- $CMD > $O/sq.log 2>&1
python3 profiles/probes/pmc_summary.py $O/sq/s_results.db > $O/r03_pmc_sq_b256.txt 2>&1 || true
rm -rf $O/kt $O/fetch $O/write $O/sq
echo collected
