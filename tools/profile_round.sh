#!/bin/bash
# Evidence run (round given by ROUND, default r3) on the GPU box for one workload: bench line, rocprofv3 kernel stats of the same command,
# and one --pmc pass per counter group (never together with --kernel-trace / --stats).
#   tools/profile_round.sh <tag> <workload> <spp> <steps> <warmup>
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-c3}; WL=${2:-C3}; SPP=${3:-256}; STEPS=${4:-3}; WARM=${5:-1}
ROUND=${ROUND:-r3}
O=$R/gpurun_out/$ROUND/$TAG
mkdir -p $O
cd $R
B="python3 bench.py --workload $WL --spp $SPP --steps $STEPS --warmup $WARM --no-cpu-baseline --no-extras"
P="python3 bench.py --workload $WL --spp $SPP --steps 1 --warmup 1 --no-cpu-baseline --no-extras --no-roofline-replay"
timeout -k 10 600 $B > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $O/stats -o bench --output-format csv -- $B > $O/stats.log 2>&1 || { tail -5 $O/stats.log; exit 1; }
pass() { n=$1; shift; timeout -s KILL 300 rocprofv3 --pmc "$@" -d $O/$n -o p --output-format csv -- $P > $O/$n.log 2>&1 || { echo "pass $n failed"; tail -3 $O/$n.log; exit 1; }; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU
pass sq2 SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 tools/pmc_to_json.py "k_path<" $O/fetch $O/write $O/sq1 $O/sq2 $O/tcc $O/tcp $O/grbm > $O/pmc.json
cp $O/stats/*kernel_stats.csv $O/kernel_stats.csv 2>/dev/null
echo "profile $ROUND $TAG done"
