# Round-3 evidence of the final code (run from the repo root through gpurun); outputs under gpurun_out/r03f/.
# The configs[2] step traces (kernel trace + the FETCH_SIZE / WRITE_SIZE passes) are those of collect_r03.sh: the extractor,
# matcher and pose kernels of that step did not change afterwards.
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f; mkdir -p $O
python3 -m pytest tests -x -q -m gpu -s > $O/r03_gpu_tests.log 2>&1
python3 -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
tail -2 $O/r03_gpu_tests.log
python3 bench.py > $O/r03_bench_b256.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats -d $O/ba -o ba -- python3 profiles/probes/ba_run.py 7 > $O/ba.log 2>&1
python3 profiles/probes/rocpd_stats.py $O/ba/ba_results.db > $O/r03_local_ba_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/c1 -o c1 -- python3 profiles/probes/chain_run.py 1 60 > $O/chain_b1.log 2>&1
python3 profiles/probes/rocpd_stats.py $O/c1/c1_results.db > $O/r03_chain_b1_kernel_stats.csv
rocprofv3 --kernel-trace --stats -d $O/c256 -o c256 -- python3 profiles/probes/chain_run.py 256 10 > $O/chain_b256.log 2>&1
python3 profiles/probes/rocpd_stats.py $O/c256/c256_results.db > $O/r03_chain_b256_kernel_stats.csv
python3 profiles/probes/chain_ref_kernels.py 1 > $O/r03_chain_reference_keyframe_b1_kernels.txt 2>/dev/null
python3 profiles/probes/chain_ref_kernels.py 256 > $O/r03_chain_reference_keyframe_b256_kernels.txt 2>/dev/null
rm -rf $O/ba $O/c1 $O/c256
echo collected
