cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_ngp; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/a -- python3 bench.py --model ngp --rays 65536 --steps 2 --warmup 1 --no-cpu --no-graph > $O/a.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/b -- python3 bench.py --model ngp --rays 65536 --steps 2 --warmup 1 --no-cpu --no-graph > $O/b.log 2>&1 &&
python3 tools/summarize_pmc.py $O/a/*/*_counter_collection.csv $O/b/*/*_counter_collection.csv > $O/summary.txt
