# SQ counters of the bandwidth-bound stages (bench_kernels.py): instruction counts, unit busy time, LDS conflicts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pmc_kernels}; mkdir -p $O
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS --output-format csv -d $O/a -- python3 bench_kernels.py > $O/a.log 2>&1 &&
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/b -- python3 bench_kernels.py > $O/b.log 2>&1 &&
python3 tools/summarize_pmc.py $O/a/*/*_counter_collection.csv $O/b/*/*_counter_collection.csv > $O/summary.txt
