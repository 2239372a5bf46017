set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03; mkdir -p $O
CMD="python3 bench.py --serial --no-ba --cpu-sample 0 --no-single --no-chain --steps 10"
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $CMD > $O/kt.log 2>&1
python3 profiles/probes/rocpd_stats.py $O/kt/kt_results.db > $O/r03_bench_serial_b256_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f -- $CMD > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w -- $CMD > $O/write.log 2>&1
python3 profiles/probes/make_traffic_json.py $O/fetch/f_results.db $O/write/w_results.db 256 $O/r03_pmc_traffic.json > $O/r03_pmc_fetch_write_b256.txt 2>&1
rm -rf $O/kt $O/fetch $O/write
echo collected
