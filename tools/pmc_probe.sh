cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc1
mkdir -p $O
cd $R
timeout -k 5 60 rocprofv3 --list-avail > $O/avail.txt 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $O/p1 -o p1 --output-format csv -- python3 tools/one.py 64 1 render_mode=1 batch_paths=268435456 > $O/p1.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum -d $O/p2 -o p2 --output-format csv -- python3 tools/one.py 64 1 render_mode=1 batch_paths=268435456 > $O/p2.log 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/p3 -o p3 --output-format csv -- python3 tools/one.py 64 1 render_mode=1 batch_paths=268435456 > $O/p3.log 2>&1
echo done
