#!/bin/bash
# Round-1 evidence run on the GPU box: parity tests, bench line, rocprofv3 kernel stats, PMC passes.
# Each rocprofv3 pass is its own short run; counters never share a run with --kernel-trace/--stats.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r1
mkdir -p $O
cd $R
ONE="python3 tools/one.py 256 1"
timeout -k 10 300 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/pytest_gpu.log 2>&1 || { tail -5 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 300 python3 bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cat $O/bench.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- python3 bench.py --no-cpu-baseline > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- $ONE > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- $ONE > $O/write.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $O/sq1 -o p --output-format csv -- $ONE > $O/sq1.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM -d $O/sq2 -o p --output-format csv -- $ONE > $O/sq2.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d $O/tcc -o p --output-format csv -- $ONE > $O/tcc.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum -d $O/tcp -o p --output-format csv -- $ONE > $O/tcp.log 2>&1 || exit 1
echo profile_r1 done
